"""Parity of the HIP contraction path (through the C ABI) with the CPU oracle.

Tolerance: max |C_gpu - C_ref| <= tol * max |C_ref| with tol = 1e-12 (f64) and
1e-5 (f32).  The oracle accumulates sequentially in f64; the kernel accumulates
per K-tile inside the MFMA, so the results differ by summation order only."""
import numpy as np
import pytest
import torch

import ctc_amd

pytestmark = pytest.mark.gpu

TOL = {"f64": 1e-12, "f32": 1e-5}
TDT = {"f64": torch.float64, "f32": torch.float32}
CODE = {"f64": 0, "f32": 1}


def _operands(oracle, spec, ext, dtype, dev, seed_a=0xC7C0A, seed_b=0xC7C0B):
    ins, outm = spec.split("->")
    la, lb = ins.split(",")
    A = oracle.fill(int(np.prod([ext[c] for c in la])) if la else 1, seed_a, CODE[dtype])
    B = oracle.fill(int(np.prod([ext[c] for c in lb])) if lb else 1, seed_b, CODE[dtype])
    n_out = int(np.prod([ext[c] for c in outm])) if outm else 1
    return A, B, n_out, torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)


def _check(got, want, dtype):
    got = got.astype(np.float64)
    want = want.astype(np.float64)
    err = np.abs(got - want).max()
    scale = max(np.abs(want).max(), 1e-300)
    assert err <= TOL[dtype] * scale, "max err %.3e vs scale %.3e" % (err, scale)


def test_golden_cases(cuda, oracle, golden):
    """Every committed vector, HIP path against the stored numpy.einsum samples and the oracle."""
    for c in golden["cases"]:
        if c["dtype"] not in TOL:
            continue
        dt = c["dtype"]
        A, B, n_out, dA, dB = _operands(oracle, c["spec"], c["extents"], dt, cuda, c["seed_a"], c["seed_b"])
        dC = torch.full((n_out,), float("nan"), dtype=TDT[dt], device=cuda)
        ctc_amd.Plan(c["spec"], c["extents"], dt).execute(dA, dB, dC)
        got = dC.cpu().numpy()
        _check(got, oracle.contract(c["spec"], c["extents"], A, B, dtype=CODE[dt]), dt)
        scale = float.fromhex(c["max_abs"])
        for i, hx in c["samples"]:
            assert abs(float(got[i]) - float.fromhex(hx)) <= 2 * TOL[dt] * scale, (c["name"], i)


CASES = [
    # name, spec, extents, dtype
    ("ccsd_32", "aeim,ebmj->abij", dict(a=32, b=32, i=32, j=32, e=32, m=32), "f64"),
    ("ccsd_ragged_tiles", "aeim,ebmj->abij", dict(a=13, b=11, i=10, j=14, e=7, m=6), "f64"),
    ("ccsd_odd_scalar_loads", "aeim,ebmj->abij", dict(a=9, b=7, i=11, j=5, e=3, m=13), "f64"),
    ("plain_gemm_edges", "ak,kb->ab", dict(a=130, b=70, k=50), "f64"),
    ("gemm_tn", "ka,kb->ab", dict(a=96, b=64, k=40), "f64"),
    ("gemm_nt", "ak,bk->ab", dict(a=96, b=64, k=40), "f64"),
    ("gemm_c_transposed", "ak,kb->ba", dict(a=72, b=136, k=24), "f64"),
    ("k_smaller_than_tile", "ak,kb->ab", dict(a=64, b=64, k=2), "f64"),
    ("single_element", "ak,kb->ab", dict(a=1, b=1, k=1), "f64"),
    ("outer_product", "a,b->ab", dict(a=70, b=130), "f64"),
    ("matvec", "ak,k->a", dict(a=300, k=70), "f64"),
    ("batched", "zak,zkb->zab", dict(z=5, a=40, b=72, k=24), "f64"),
    ("batch_fastest", "akz,kbz->abz", dict(z=6, a=20, b=24, k=10), "f64"),
    ("six_index", "adij,bcdk->abcijk", dict(a=6, b=5, c=4, i=6, j=4, k=8, d=16), "f64"),
    ("adversarial", "dkca,bk->abcd", dict(a=32, b=8, c=12, d=32, k=8), "f32"),
    ("ccsd_f32", "aeim,ebmj->abij", dict(a=24, b=24, i=24, j=24, e=24, m=24), "f32"),
    ("ccsd_f32_scalar", "aeim,ebmj->abij", dict(a=6, b=7, i=5, j=9, e=3, m=10), "f32"),
    ("hadamard_batch_only", "ab,ab->ab", dict(a=33, b=20), "f64"),
    # few tiles, long K: every tile is cut along K and summed by the fix-up kernel
    ("splitk_small_tiles", "ak,kb->ab", dict(a=256, b=256, k=2048), "f64"),
    ("splitk_ragged", "ak,kb->ab", dict(a=250, b=201, k=2047), "f64"),
    ("splitk_f32", "ak,bk->ab", dict(a=192, b=320, k=4096), "f32"),
    ("splitk_permuted", "aeim,ebmj->abij", dict(a=12, b=12, i=12, j=12, e=40, m=40), "f64"),
    # skinny kernels (M <= 16, long N)
    ("skinny_transpose_ragged", "dkca,bk->abcd", dict(a=36, b=8, c=5, d=44, k=8), "f32"),
    ("skinny_transpose_f64_m3", "dkca,bk->abcd", dict(a=34, b=3, c=7, d=66, k=5), "f64"),
    ("skinny_transpose_m16", "dkca,bk->abcd", dict(a=64, b=16, c=2, d=64, k=12), "f32"),
    ("skinny_same_f32", "kn,mk->mn", dict(m=8, n=40000, k=8), "f32"),
    ("skinny_same_f64_m13", "kn,mk->mn", dict(m=13, n=8192, k=24), "f64"),
    ("skinny_same_batched", "zkn,zmk->zmn", dict(z=3, m=4, n=4096, k=6), "f64"),
    ("skinny_two_level_n", "kpq,mk->mpq", dict(m=5, p=70, q=72, k=9), "f64"),
]


@pytest.mark.parametrize("name,spec,ext,dtype", CASES, ids=[c[0] for c in CASES])
def test_against_oracle(cuda, oracle, name, spec, ext, dtype):
    A, B, n_out, dA, dB = _operands(oracle, spec, ext, dtype, cuda)
    dC = torch.full((n_out,), float("nan"), dtype=TDT[dtype], device=cuda)
    ctc_amd.Plan(spec, ext, dtype).execute(dA, dB, dC)
    _check(dC.cpu().numpy(), oracle.contract(spec, ext, A, B, dtype=CODE[dtype]), dtype)


def test_config3_adversarial_contraction_at_full_size(cuda, oracle):
    """The transpose-bound contraction at the size the measurements quote: abcd = dkca,bk fp32 with
    a = c = d = 128, b = k = 8 (64 MiB per large tensor, 0.27 Gflop: seconds for the oracle)."""
    spec, ext = "dkca,bk->abcd", dict(a=128, b=8, c=128, d=128, k=8)
    A, B, n_out, dA, dB = _operands(oracle, spec, ext, "f32", cuda)
    dC = torch.full((n_out,), float("nan"), dtype=torch.float32, device=cuda)
    plan = ctc_amd.Plan(spec, ext, "f32")
    assert plan.summary()["kernel"].startswith("skinny"), plan.summary()
    plan.execute(dA, dB, dC)
    _check(dC.cpu().numpy(), oracle.contract(spec, ext, A, B, dtype=CODE["f32"]), "f32")


def test_headline_contraction_at_extent_96_slabs(cuda, oracle):
    """abij = aeim,ebmj fp64 at all extents 96 (1.57e12 flop: far too much for the oracle as a whole):
    three C[a0, b0, :, :] slabs of the result are recomputed by the oracle from regenerated operands, and
    the launch must have taken the LDS-DMA kernel with K-tiles of 64-byte rows."""
    n = 96
    spec, ext = "aeim,ebmj->abij", dict(a=n, b=n, i=n, j=n, e=n, m=n)
    dA = torch.empty(n ** 4, dtype=torch.float64, device=cuda)
    dB = torch.empty(n ** 4, dtype=torch.float64, device=cuda)
    ctc_amd.fill(dA, 0xC7C0A)
    ctc_amd.fill(dB, 0xC7C0B)
    dC = torch.full((n ** 4,), float("nan"), dtype=torch.float64, device=cuda)
    plan = ctc_amd.Plan(spec, ext, "f64")
    assert plan.schedule()["dma_form"] == 1
    before = ctc_amd.get_option("launches_dma_gemm")
    plan.execute(dA, dB, dC)
    torch.cuda.synchronize()
    assert ctc_amd.get_option("launches_dma_gemm") == before + 1
    assert not bool(torch.isnan(dC).any())
    B = oracle.fill(n ** 4, 0xC7C0B).reshape(n, n, n, n)
    got = dC.reshape(n, n, n * n)
    for a0, b0 in ((0, 0), (n // 3, n // 2), (n - 1, n - 1)):
        A1 = oracle.fill(n ** 3, 0xC7C0A + a0 * n ** 3)  # A[a0] is n^3 consecutive elements of the stream
        want = oracle.contract(spec, dict(a=1, b=1, i=n, j=n, e=n, m=n), A1,
                               np.ascontiguousarray(B[:, b0:b0 + 1]).reshape(-1))
        _check(got[a0, b0].cpu().numpy(), want, "f64")


# Short contracted extent, many tiles: the epilogue (16-byte stores of C through the offset
# tables kept in LDS) carries as much weight as the K loop.  Ragged edges, K tails, every
# staging layout, batches.
SHORT_K_CASES = [
    ("short_k_nn", "ak,kb->ab", dict(a=1100, b=900, k=64), "f64"),
    ("short_k_nn_k_tail", "ak,kb->ab", dict(a=4100, b=2500, k=70), "f64"),
    ("short_k_tn", "ka,kb->ab", dict(a=1056, b=2304, k=48), "f64"),
    ("short_k_nt", "ak,bk->ab", dict(a=1000, b=1200, k=112), "f64"),
    ("short_k_c_transposed", "ak,kb->ba", dict(a=1300, b=1100, k=96), "f64"),
    ("short_k_batched", "zak,zkb->zab", dict(z=3, a=700, b=500, k=48), "f64"),
    ("short_k_six_index", "adij,bcdk->abcijk", dict(a=6, b=5, c=4, i=6, j=4, k=8, d=48), "f64"),
    ("short_k_scalar_loads", "aeim,bejm->abij", dict(a=9, b=7, i=11, j=6, e=5, m=13), "f64"),
    ("short_k_odd_n_element_stores", "ak,kb->ab", dict(a=700, b=333, k=40), "f64"),
    ("short_k_f32_nt", "ak,bk->ab", dict(a=2000, b=1500, k=100), "f32"),
    ("short_k_f32_tn", "ka,kb->ab", dict(a=1024, b=1152, k=36), "f32"),
    ("short_k_f32_n_not_multiple_of_4", "ak,kb->ab", dict(a=300, b=1002, k=36), "f32"),
]


@pytest.mark.parametrize("name,spec,ext,dtype", SHORT_K_CASES, ids=[c[0] for c in SHORT_K_CASES])
def test_short_k_against_oracle(cuda, oracle, name, spec, ext, dtype):
    A, B, n_out, dA, dB = _operands(oracle, spec, ext, dtype, cuda)
    dC = torch.full((n_out,), float("nan"), dtype=TDT[dtype], device=cuda)
    ctc_amd.Plan(spec, ext, dtype).execute(dA, dB, dC)
    _check(dC.cpu().numpy(), oracle.contract(spec, ext, A, B, dtype=CODE[dtype]), dtype)


def test_vector_stores_alpha_beta(cuda, oracle):
    spec, ext = "ak,kb->ab", dict(a=2300, b=1700, k=80)
    A, B, n_out, dA, dB = _operands(oracle, spec, ext, "f64", cuda)
    C0 = oracle.fill(n_out, 99)
    dC = torch.from_numpy(C0.copy()).to(cuda)
    plan = ctc_amd.Plan(spec, ext)
    assert plan.summary()["vec_c"] == 2
    plan.execute(dA, dB, dC, alpha=0.75, beta=-1.5)
    want = oracle.contract(spec, ext, A, B, C=C0.copy(), alpha=0.75, beta=-1.5)
    _check(dC.cpu().numpy(), want, "f64")


@pytest.mark.parametrize("force", ["64", "128"])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_c_base_off_the_16_byte_grid(cuda, oracle, monkeypatch, force, dtype):
    """C one element past an aligned address: the 16-byte store path must not be taken."""
    monkeypatch.setenv("CTC_TC_FORCE_TILE", force)
    spec, ext = "ak,kb->ab", dict(a=300, b=264, k=48)
    A, B, n_out, dA, dB = _operands(oracle, spec, ext, dtype, cuda)
    plan = ctc_amd.Plan(spec, ext, dtype)
    assert plan.summary()["vec_c"] > 1
    buf = torch.full((n_out + 1,), float("nan"), dtype=TDT[dtype], device=cuda)
    plan.execute(dA, dB, buf[1:])
    _check(buf[1:].cpu().numpy(), oracle.contract(spec, ext, A, B, dtype=CODE[dtype]), dtype)
    assert torch.isnan(buf[0])


def test_skinny_cases_use_the_skinny_kernels():
    kinds = {c[0]: ctc_amd.Plan(c[1], c[2], c[3]).summary()["kernel"] for c in CASES if c[0].startswith("skinny")}
    assert kinds["skinny_transpose_ragged"] == "skinny_transpose"
    assert kinds["skinny_transpose_f64_m3"] == "skinny_transpose"
    assert kinds["skinny_same_f32"] == "skinny_same"
    assert kinds["skinny_same_batched"] == "skinny_same"
    assert ctc_amd.Plan("aeim,ebmj->abij", dict(a=32, b=32, i=32, j=32, e=32, m=32)).summary()["kernel"] == "mfma_gemm"


@pytest.mark.parametrize("spec,ext,dtype", [
    ("dkca,bk->abcd", dict(a=36, b=8, c=5, d=44, k=8), "f32"),
    ("kn,mk->mn", dict(m=7, n=8192, k=10), "f64"),
])
def test_skinny_alpha_beta(cuda, oracle, spec, ext, dtype):
    A, B, n_out, dA, dB = _operands(oracle, spec, ext, dtype, cuda)
    C0 = oracle.fill(n_out, 99, CODE[dtype])
    dC = torch.from_numpy(C0.copy()).to(cuda)
    ctc_amd.Plan(spec, ext, dtype).execute(dA, dB, dC, alpha=0.5, beta=2.0)
    want = oracle.contract(spec, ext, A, B, C=C0.copy(), alpha=0.5, beta=2.0, dtype=CODE[dtype])
    _check(dC.cpu().numpy(), want, dtype)


def test_alpha_beta(cuda, oracle):
    spec, ext = "aeim,ebmj->abij", dict(a=12, b=10, i=8, j=6, e=4, m=8)
    A, B, n_out, dA, dB = _operands(oracle, spec, ext, "f64", cuda)
    C0 = oracle.fill(n_out, 99)
    dC = torch.from_numpy(C0.copy()).to(cuda)
    ctc_amd.Plan(spec, ext).execute(dA, dB, dC, alpha=0.75, beta=-1.5)
    want = oracle.contract(spec, ext, A, B, C=C0.copy(), alpha=0.75, beta=-1.5)
    _check(dC.cpu().numpy(), want, "f64")


def test_splitk_alpha_beta_and_repeat(cuda, oracle):
    spec, ext = "ak,kb->ab", dict(a=250, b=201, k=2047)
    A, B, n_out, dA, dB = _operands(oracle, spec, ext, "f64", cuda)
    C0 = oracle.fill(n_out, 77)
    want = oracle.contract(spec, ext, A, B, C=C0.copy(), alpha=1.5, beta=-0.25)
    plan = ctc_amd.Plan(spec, ext)
    plan.prepare()
    other = torch.cuda.Stream()
    for stream in (None, other, None):  # same plan from two streams: the workspace hand-over is ordered
        dC = torch.from_numpy(C0.copy()).to(cuda)
        torch.cuda.synchronize()
        plan.execute(dA, dB, dC, alpha=1.5, beta=-0.25, stream=stream)
        torch.cuda.synchronize()
        _check(dC.cpu().numpy(), want, "f64")


def test_beta_zero_ignores_nan_in_c(cuda, oracle):
    spec, ext = "ak,kb->ab", dict(a=40, b=24, k=16)
    A, B, n_out, dA, dB = _operands(oracle, spec, ext, "f64", cuda)
    dC = torch.full((n_out,), float("nan"), dtype=torch.float64, device=cuda)
    ctc_amd.Plan(spec, ext).execute(dA, dB, dC, beta=0.0)
    assert torch.isfinite(dC).all()


def test_padded_strides_and_unaligned_base(cuda, oracle):
    ext = dict(a=20, b=12, k=16)
    A = oracle.fill(20 * 18 + 1, 5)
    B = oracle.fill(16 * 12, 6)
    want = oracle.contract("ak,kb->ab", ext, A[1:], B, strides_a=[18, 1])
    dA = torch.from_numpy(A).to(cuda)
    dB = torch.from_numpy(B).to(cuda)
    dC = torch.zeros(20 * 12, dtype=torch.float64, device=cuda)
    plan = ctc_amd.Plan("ak,kb->ab", ext, strides_a=[18, 1])
    assert plan.summary()["layout_a"] == "k_vec"
    # base pointer shifted by one element: only 8-byte aligned, forces element loads
    plan.execute(dA[1:], dB, dC)
    _check(dC.cpu().numpy(), want, "f64")


def test_torch_tensor_front_end(cuda, oracle):
    A = torch.from_numpy(oracle.fill(6 * 5 * 4, 1)).to(cuda).reshape(6, 5, 4)
    B = torch.from_numpy(oracle.fill(5 * 7, 2)).to(cuda).reshape(5, 7)
    got = ctc_amd.contract("adc,db->abc", A, B).cpu().numpy()
    want = np.einsum("adc,db->abc", A.cpu().numpy(), B.cpu().numpy())
    assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()
    # non-contiguous view as an operand
    got = ctc_amd.contract("dac,db->abc", A.permute(1, 0, 2), B).cpu().numpy()
    assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()


def test_ttgt_path_matches_fused_path(cuda, oracle):
    spec, ext = "aeim,ebmj->abij", dict(a=12, b=10, i=8, j=6, e=4, m=8)
    A, B, n_out, dA, dB = _operands(oracle, spec, ext, "f64", cuda)
    plan = ctc_amd.Plan(spec, ext)
    ws = torch.empty(plan.ttgt_workspace_bytes(), dtype=torch.uint8, device=cuda)
    dC = torch.full((n_out,), float("nan"), dtype=torch.float64, device=cuda)
    plan.execute_ttgt(dA, dB, dC, ws)
    _check(dC.cpu().numpy(), oracle.contract(spec, ext, A, B), "f64")


def test_sharded_execution_free_and_contracted(cuda, oracle):
    spec, ext = "aeim,ebmj->abij", dict(a=8, b=6, i=6, j=8, e=4, m=6)
    A, B, n_out, dA, dB = _operands(oracle, spec, ext, "f64", cuda)
    want = oracle.contract(spec, ext, A, B)
    full = ctc_amd.Plan(spec, ext)
    dC = torch.full((n_out,), float("nan"), dtype=torch.float64, device=cuda)
    for r in range(4):
        sh, oa, ob, oc = full.shard("a", r, 4)
        sh.execute(dA[oa:], dB[ob:], dC[oc:])
    _check(dC.cpu().numpy(), want, "f64")
    acc = torch.zeros(n_out, dtype=torch.float64, device=cuda)
    for r in range(2):
        sh, oa, ob, oc = full.shard("e", r, 2)
        part = torch.empty(n_out, dtype=torch.float64, device=cuda)
        sh.execute(dA[oa:], dB[ob:], part[oc:])
        acc += part
    _check(acc.cpu().numpy(), want, "f64")


def test_device_generator_is_bit_identical(cuda, oracle):
    for dt, code in ((torch.float64, 0), (torch.float32, 1), (torch.bfloat16, 2)):
        x = torch.empty(10007, dtype=dt, device=cuda)
        ctc_amd.fill(x, 0xC7C0A, first=5)
        want = oracle.fill(10007, 0xC7C0A + 5, code)
        got = x.view(torch.int16).cpu().numpy().view(np.uint16) if code == 2 else x.cpu().numpy()
        assert np.array_equal(got, want)


def test_headline_size_properties(cuda, oracle):
    """N = 64 (BASELINE config 2): too big for the oracle in seconds, so check
    (i) a free-index slab of it against the oracle and (ii) linearity in A."""
    n = 64
    spec, ext = "aeim,ebmj->abij", dict(a=n, b=n, i=n, j=n, e=n, m=n)
    dA = torch.empty(n ** 4, dtype=torch.float64, device=cuda)
    dB = torch.empty(n ** 4, dtype=torch.float64, device=cuda)
    ctc_amd.fill(dA, 0xC7C0A)
    ctc_amd.fill(dB, 0xC7C0B)
    dC = torch.empty(n ** 4, dtype=torch.float64, device=cuda)
    plan = ctc_amd.Plan(spec, ext)
    plan.execute(dA, dB, dC)
    # (i) rows a = 5 and b in {0, 63}: C[5, b, :, :] recomputed by the oracle on the slab
    A = oracle.fill(n ** 4, 0xC7C0A).reshape(n, n, n, n)
    B = oracle.fill(n ** 4, 0xC7C0B).reshape(n, n, n, n)
    for b in (0, 63):
        sub = oracle.contract("aeim,ebmj->abij", dict(a=1, b=1, i=n, j=n, e=n, m=n),
                              np.ascontiguousarray(A[5:6]).reshape(-1),
                              np.ascontiguousarray(B[:, b:b + 1]).reshape(-1))
        got = dC.reshape(n, n, n, n)[5, b].cpu().numpy().reshape(-1)
        _check(got, sub, "f64")
    # (ii) linearity: contract(2A, B) == 2 contract(A, B) exactly (power-of-two scaling)
    dC2 = torch.empty_like(dC)
    plan.execute(dA * 2.0, dB, dC2)
    assert torch.equal(dC2, dC * 2.0)


def test_offsets_beyond_2_31_elements(cuda):
    """C with 2.6e9 elements (10.5 GB of f32): row and column offsets are 64-bit all the way to the
    stores.  Rows on both sides of the 2^31-element line are recomputed in f64."""
    a, b, k = 65536, 40000, 8
    dA = torch.empty(a * k, dtype=torch.float32, device=cuda)
    dB = torch.empty(k * b, dtype=torch.float32, device=cuda)
    ctc_amd.fill(dA, 0xC7C0A)
    ctc_amd.fill(dB, 0xC7C0B)
    dC = torch.empty(a * b, dtype=torch.float32, device=cuda)
    ctc_amd.Plan("ak,kb->ab", dict(a=a, b=b, k=k), "f32").execute(dA, dB, dC)
    A = dA.reshape(a, k).double()
    B = dB.reshape(k, b).double()
    for r in (0, 1, 53686, 53687, 53688, a - 1):   # 53687 * 40000 < 2^31 <= 53688 * 40000
        want = A[r] @ B
        got = dC[r * b:(r + 1) * b].double()
        assert (got - want).abs().max() <= 1e-5 * want.abs().max(), r
    del dC
    torch.cuda.empty_cache()


def test_schedule_does_not_depend_on_earlier_plans(cuda):
    """The split of the last round of tiles is computed from the resident workgroups of the kernel
    instantiation that will run; a small-tile plan scheduled first (four workgroups per CU) must not
    change what a large-tile plan gets (two per CU), in either element type."""
    big_ext = dict(a=56, b=56, i=56, j=56, e=56, m=56)       # 25 x 25 tiles of 128: one round of 512 + 113
    small_ext = dict(a=32, b=32, i=32, j=32, e=32, m=32)     # 64 x 64 tiles
    for dt in ("f64", "f32"):
        small = ctc_amd.Plan("aeim,ebmj->abij", small_ext, dt)
        assert tuple(small.summary()["tile"][:2]) == (64, 64)
        s_small = small.schedule()
        big = ctc_amd.Plan("aeim,ebmj->abij", big_ext, dt)
        assert tuple(big.summary()["tile"][:2]) == (128, 128)
        s = big.schedule()
        assert s["tile_rows"] * s["tile_cols"] == 625
        slots_big = s["whole_tiles"]  # one full round
        assert s["tail_tiles"] == 625 - slots_big and s["tail_tiles"] > 0 and s["split"] >= 2, (dt, s, s_small)
        assert slots_big % 256 == 0 and slots_big <= 512, (dt, s)   # two workgroups per CU at most for this tile


@pytest.mark.parametrize("spec,ext,dtype", [
    ("akz,kbz->abz", dict(z=6, a=20, b=24, k=10), "f64"),
    ("akz,kbz->abz", dict(z=16, a=130, b=72, k=40), "f32"),
    ("kaz,bzk->baz", dict(z=5, a=33, b=17, k=21), "f64"),       # z fastest in A and C only: fused path
    ("akyz,kbyz->abyz", dict(y=3, z=4, a=9, b=11, k=7), "f64"),  # two batch modes, both fastest
])
def test_batch_fastest_layouts_take_the_transposes(cuda, oracle, spec, ext, dtype):
    """A batch mode that is the unit-stride mode of A, B and C: ctc_tc_execute transposes explicitly
    through a workspace the plan owns (beta == 0) and falls back to the fused kernel for beta != 0;
    both must match the oracle, repeatedly and on two streams."""
    A, B, n_out, dA, dB = _operands(oracle, spec, ext, dtype, cuda)
    plan = ctc_amd.Plan(spec, ext, dtype)
    want = oracle.contract(spec, ext, A, B, alpha=1.5, dtype=CODE[dtype])
    s2 = torch.cuda.Stream()
    for rep in range(3):
        dC = torch.full((n_out,), float("nan"), dtype=TDT[dtype], device=cuda)
        if rep == 2:
            torch.cuda.synchronize()  # the fill ran on the default stream: s2 must not overtake it
            with torch.cuda.stream(s2):
                plan.execute(dA, dB, dC, alpha=1.5, stream=s2.cuda_stream)
            s2.synchronize()
        else:
            plan.execute(dA, dB, dC, alpha=1.5)
        _check(dC.cpu().numpy(), want, dtype)
    C0 = oracle.fill(n_out, 77, CODE[dtype])
    dC = torch.from_numpy(C0.copy()).to(cuda)
    plan.execute(dA, dB, dC, alpha=1.5, beta=-0.5)
    _check(dC.cpu().numpy(), oracle.contract(spec, ext, A, B, C=C0.copy(), alpha=1.5, beta=-0.5, dtype=CODE[dtype]), dtype)


@pytest.mark.parametrize("spec,ext", [
    ("zak,zkb->zab", dict(z=70000, a=5, b=6, k=4)),        # tiled kernel, batch slowest
    ("zkn,zmk->zmn", dict(z=66000, m=2, n=4096, k=3)),     # skinny kernel
])
def test_more_batch_entries_than_one_grid_holds(cuda, oracle, spec, ext):
    """The batch index is a grid dimension (65535 at most): larger batches run as several launches."""
    dtype = "f32"
    A, B, n_out, dA, dB = _operands(oracle, spec, ext, dtype, cuda)
    dC = torch.full((n_out,), float("nan"), dtype=TDT[dtype], device=cuda)
    plan = ctc_amd.Plan(spec, ext, dtype)
    plan.prepare()
    plan.execute(dA, dB, dC)
    _check(dC.cpu().numpy(), oracle.contract(spec, ext, A, B, dtype=CODE[dtype]), dtype)


def test_prepare_covers_the_explicit_transpose_path(cuda, oracle):
    """After prepare(), executing a plan that runs through explicit transposes takes no device memory."""
    spec, ext = "akz,kbz->abz", dict(z=16, a=96, b=80, k=64)
    A, B, n_out, dA, dB = _operands(oracle, spec, ext, "f64", cuda)
    dC = torch.full((n_out,), float("nan"), dtype=torch.float64, device=cuda)
    plan = ctc_amd.Plan(spec, ext, "f64")
    plan.prepare()
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    plan.execute(dA, dB, dC)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before
    _check(dC.cpu().numpy(), oracle.contract(spec, ext, A, B), "f64")
