"""Seeded random contractions (random mode sets, mode orders, extents, element types, alpha / beta)
through the C ABI against the CPU oracle.  The generator favours the shapes that select different
kernels: vector and element layouts, role swaps, batch modes, K shorter and longer than a tile,
skinny M, split-K sized K.  CTC_FUZZ_CASES raises the number of cases (default 400), CTC_FUZZ_SEED draws another
sequence of them (for one-off campaigns beside the fixed suite)."""
import os
import random

import numpy as np
import pytest
import torch

import ctc_amd

pytestmark = pytest.mark.gpu

TOL = {"f64": 1e-12, "f32": 1e-5, "bf16": 8e-3}
CODE = {"f64": 0, "f32": 1, "bf16": 2}
LETTERS = "abcdefghij"
_SEED = int(os.environ.get("CTC_FUZZ_SEED", "0"), 0)  # 0: the suite's fixed sequences


def _case(rng):
    n_free_a = rng.choice([0, 1, 1, 2, 2, 3])
    n_free_b = rng.choice([0, 1, 1, 2, 2, 3])
    n_k = rng.choice([0, 1, 1, 1, 2, 2])
    n_batch = rng.choice([0, 0, 0, 1, 1, 2])
    if n_free_a + n_free_b + n_batch == 0:
        n_free_a = 1
    names = list(LETTERS[:n_free_a + n_free_b + n_k + n_batch])
    rng.shuffle(names)
    fa, rest = names[:n_free_a], names[n_free_a:]
    fb, rest = rest[:n_free_b], rest[n_free_b:]
    ks, bs = rest[:n_k], rest[n_k:]
    menu = rng.choice([[1, 2, 3, 5, 8], [4, 8, 12, 16], [7, 9, 16, 33], [2, 4, 64], [16, 24, 40]])
    ext = {c: rng.choice(menu) for c in names}
    style = rng.random()
    if ks and style < 0.15:
        ext[ks[0]] = rng.choice([256, 1000, 2048])   # long K: split along K
    elif fa and style < 0.3:
        ext[fa[0]] = rng.choice([200, 513])            # many tile rows
    elif fb and style < 0.45:
        ext[fb[0]] = rng.choice([4096, 5000])          # long N: skinny kernels when M is small
    a_modes, b_modes, c_modes = fa + ks + bs, fb + ks + bs, fa + fb + bs
    if not a_modes or not b_modes:
        return None
    for m in (a_modes, b_modes, c_modes):
        rng.shuffle(m)
    size = lambda m: int(np.prod([ext[c] for c in m])) if m else 1  # noqa: E731
    flops = size(fa) * size(fb) * size(ks) * size(bs)
    if max(size(a_modes), size(b_modes), size(c_modes)) > 6_000_000 or flops > 400_000_000:
        return None
    spec = "%s,%s->%s" % ("".join(a_modes), "".join(b_modes), "".join(c_modes))
    dtype = rng.choice(["f64", "f64", "f32", "f32", "bf16"])
    alpha = rng.choice([1.0, 1.0, -0.5, 2.0])
    beta = rng.choice([0.0, 0.0, 0.0, 1.0, -1.5])
    return spec, ext, dtype, alpha, beta


def _cases():
    rng = random.Random(0xC7C ^ _SEED)
    want = int(os.environ.get("CTC_FUZZ_CASES", "400"))
    out = []
    while len(out) < want:
        c = _case(rng)
        if c is not None:
            out.append(c)
    return out


def _to_dev(x, dtype, dev):
    if dtype == "bf16":
        return torch.from_numpy(x.view(np.int16)).to(dev).view(torch.bfloat16)
    return torch.from_numpy(x).to(dev)


def _to_host(t, dtype):
    if dtype == "bf16":
        return t.view(torch.int16).cpu().numpy().view(np.uint16)
    return t.cpu().numpy()


def test_random_contractions_against_oracle(cuda, oracle):
    kernels = {}
    for n, (spec, ext, dtype, alpha, beta) in enumerate(_cases()):
        ins, outm = spec.split("->")
        la, lb = ins.split(",")
        size = lambda m: int(np.prod([ext[c] for c in m])) if m else 1  # noqa: E731
        code = CODE[dtype]
        A = oracle.fill(size(la), 1000 + n, code)
        B = oracle.fill(size(lb), 2000 + n, code)
        C0 = oracle.fill(size(outm), 3000 + n, code)
        plan = ctc_amd.Plan(spec, ext, dtype)
        s = plan.summary()
        kernels[(s["kernel"], dtype)] = kernels.get((s["kernel"], dtype), 0) + 1
        dC = _to_dev(C0.copy(), dtype, cuda)
        plan.execute(_to_dev(A, dtype, cuda), _to_dev(B, dtype, cuda), dC, alpha=alpha, beta=beta)
        want = oracle.contract(spec, ext, A, B, C=C0.copy(), alpha=alpha, beta=beta, dtype=code)
        got = _to_host(dC, dtype)
        if dtype == "bf16":
            got, want = oracle.bf16_to_f32(got), oracle.bf16_to_f32(want)
        got, want = got.astype(np.float64), want.astype(np.float64)
        assert np.isfinite(got).all(), (n, spec, ext, dtype)
        scale = max(np.abs(want).max(), 1e-300)
        err = np.abs(got - want).max()
        assert err <= TOL[dtype] * scale, (n, spec, ext, dtype, alpha, beta, s["kernel"], err, scale)
    # the generator reached the GEMM kernel in every element type and the skinny kernels
    assert {k for k, _ in kernels} >= {"mfma_gemm"}, kernels
    assert all(("mfma_gemm", d) in kernels for d in ("f64", "f32", "bf16")), kernels


def test_random_permutations_are_bit_exact(cuda):
    """Seeded random mode permutations (rank 1..6, element sizes 2 / 4 / 8) against numpy.transpose."""
    rng = random.Random(0x9E37 ^ _SEED)
    done = 0
    want_cases = int(os.environ.get("CTC_FUZZ_CASES", "400"))
    while done < want_cases:
        rank = rng.choice([1, 2, 2, 3, 3, 4, 4, 5, 6])
        menu = rng.choice([[1, 2, 3, 5], [4, 8, 16], [7, 33, 64], [2, 100], [16, 20, 128]])
        shape = [rng.choice(menu) for _ in range(rank)]
        if int(np.prod(shape)) > 4_000_000:
            continue
        perm = list(range(rank))
        rng.shuffle(perm)
        tdt, ndt = rng.choice([(torch.int16, np.int16), (torch.float32, np.float32), (torch.float64, np.float64)])
        n = int(np.prod(shape))
        host = (np.arange(n, dtype=np.int64) * 2654435761 % 65521).astype(ndt).reshape(shape)
        x = torch.from_numpy(host).to(cuda)
        out = torch.full([shape[p] for p in perm], -1, dtype=tdt, device=cuda)
        ctc_amd.permute(x, perm, out=out)
        assert np.array_equal(out.cpu().numpy(), np.transpose(host, perm)), (shape, perm, ndt)
        done += 1


def test_random_shardings_against_oracle(cuda, oracle):
    """Random contractions cut into 2..4 slabs along a random mode (free: disjoint slabs of C written in
    place; contracted: partial results summed; batch: disjoint slabs), every slab executed through its
    own sub-plan with the full tensors' strides, against the oracle's unsharded result."""
    rng = random.Random(0x5AAD ^ _SEED)
    want_cases = max(20, int(os.environ.get("CTC_FUZZ_CASES", "400")) // 3)
    done = 0
    kinds = set()
    while done < want_cases:
        c = _case(rng)
        if c is None:
            continue
        spec, ext, dtype, alpha, _beta = c
        if dtype == "bf16":
            continue  # partial sums of a contracted mode would be rounded to bf16 before they are added
        ins, outm = spec.split("->")
        la, lb = ins.split(",")
        nr = rng.choice([2, 3, 4])
        modes = [m for m in ext if ext[m] % nr == 0 and ext[m] >= nr]
        if not modes:
            continue
        mode = rng.choice(modes)
        size = lambda m: int(np.prod([ext[x] for x in m])) if m else 1  # noqa: E731
        code = CODE[dtype]
        A = oracle.fill(size(la), 5000 + done, code)
        B = oracle.fill(size(lb), 6000 + done, code)
        want = oracle.contract(spec, ext, A, B, alpha=alpha, dtype=code).astype(np.float64)
        tdt = torch.float64 if dtype == "f64" else torch.float32
        dA, dB = torch.from_numpy(A).to(cuda), torch.from_numpy(B).to(cuda)
        full = ctc_amd.Plan(spec, ext, dtype)
        contracted = mode not in outm
        kinds.add("contracted" if contracted else "batch" if (mode in la and mode in lb) else "free")
        if contracted:
            acc = torch.zeros(size(outm), dtype=torch.float64, device=cuda)
            for r in range(nr):
                sh, oa, ob, oc = full.shard(mode, r, nr)
                part = torch.full((size(outm),), float("nan"), dtype=tdt, device=cuda)
                sh.execute(dA[oa:], dB[ob:], part[oc:], alpha=alpha)
                acc += part.double()
            got = acc.cpu().numpy()
        else:
            dC = torch.full((size(outm),), float("nan"), dtype=tdt, device=cuda)
            for r in range(nr):
                sh, oa, ob, oc = full.shard(mode, r, nr)
                sh.execute(dA[oa:], dB[ob:], dC[oc:], alpha=alpha)
            got = dC.cpu().numpy().astype(np.float64)
        assert np.isfinite(got).all(), (spec, ext, dtype, mode, nr)
        err = np.abs(got - want).max()
        assert err <= TOL[dtype] * max(np.abs(want).max(), 1e-300), (spec, ext, dtype, mode, nr, err)
        done += 1
    assert kinds >= {"free", "contracted"}, kinds


def _padded(modes, ext, rng):
    """Strides (in elements, per operand position) of a tensor whose modes are padded by random amounts,
    and the number of elements its buffer needs."""
    pad = {m: ext[m] + rng.choice([0, 0, 1, 2, 5]) for m in modes}
    strides, s = [0] * len(modes), 1
    for i in range(len(modes) - 1, -1, -1):
        strides[i] = s
        s *= pad[modes[i]]
    need = 1 + sum((ext[m] - 1) * st for m, st in zip(modes, strides))
    return strides, need


def test_random_strided_operands_against_oracle(cuda, oracle):
    """Padded (non-packed) strides on A, B and C and base pointers shifted off the 16-byte grid."""
    rng = random.Random(0x57D1DE ^ _SEED)
    want_cases = max(20, int(os.environ.get("CTC_FUZZ_CASES", "400")) // 3)
    done = 0
    while done < want_cases:
        c = _case(rng)
        if c is None:
            continue
        spec, ext, dtype, alpha, beta = c
        if dtype == "bf16":
            continue
        ins, outm = spec.split("->")
        la, lb = ins.split(",")
        sa, na = _padded(la, ext, rng)
        sb, nb = _padded(lb, ext, rng)
        sc, nc = _padded(outm, ext, rng)
        if max(na, nb, nc) > 8_000_000:
            continue
        code = CODE[dtype]
        shift = [rng.choice([0, 0, 1, 3]) for _ in range(3)]
        A = oracle.fill(na + shift[0], 7000 + done, code)
        B = oracle.fill(nb + shift[1], 8000 + done, code)
        C0 = oracle.fill(nc + shift[2], 9000 + done, code)
        want = oracle.contract(spec, ext, A[shift[0]:], B[shift[1]:], C=C0[shift[2]:].copy(), alpha=alpha, beta=beta,
                               dtype=code, strides_a=sa, strides_b=sb, strides_c=sc)
        plan = ctc_amd.Plan(spec, ext, dtype, strides_a=sa, strides_b=sb, strides_c=sc)
        dC = torch.from_numpy(C0.copy()).to(cuda)
        plan.execute(torch.from_numpy(A).to(cuda)[shift[0]:], torch.from_numpy(B).to(cuda)[shift[1]:], dC[shift[2]:],
                     alpha=alpha, beta=beta)
        got = dC.cpu().numpy()[shift[2]:].astype(np.float64)
        want = want.astype(np.float64)
        # elements in the padding are never touched; elements of C are compared where they exist
        assert np.array_equal(np.isnan(got), np.isnan(want))
        err = np.abs(got - want).max()
        assert err <= TOL[dtype] * max(np.abs(want).max(), 1e-300), (spec, ext, dtype, sa, sb, sc, shift, err)
        done += 1
