"""Parity of the bf16 contraction path (bf16 operands and result, f32 accumulation inside
v_mfma_f32_16x16x32_bf16) with the CPU oracle (f64 accumulation, one rounding to bf16).

Tolerance: max |C_gpu - C_ref| <= 8e-3 * max |C_ref|.  One bf16 rounding of the result is
worth up to 2^-8 = 3.9e-3 of the element; the f32 accumulation error can move a sum across a
rounding boundary, so two correctly computed results can differ by one bf16 ulp."""
import numpy as np
import pytest
import torch

import ctc_amd

pytestmark = pytest.mark.gpu

TOL = 8e-3
BF16 = 2


def _to_dev(u16, dev):
    return torch.from_numpy(u16.view(np.int16)).to(dev).view(torch.bfloat16)


def _to_host(t):
    return t.view(torch.int16).cpu().numpy().view(np.uint16)


def _sizes(spec, ext):
    ins, outm = spec.split("->")
    la, lb = ins.split(",")
    n = lambda s: int(np.prod([ext[c] for c in s])) if s else 1  # noqa: E731
    return n(la), n(lb), n(outm)


def _check(oracle, got_u16, want_u16):
    got = oracle.bf16_to_f32(got_u16).astype(np.float64)
    want = oracle.bf16_to_f32(want_u16).astype(np.float64)
    assert np.isfinite(got).all()
    err = np.abs(got - want).max()
    scale = max(np.abs(want).max(), 1e-300)
    assert err <= TOL * scale, "max err %.3e vs scale %.3e" % (err, scale)


CASES = [
    # name, spec, extents, expected (layout_a, layout_b) or None
    ("nn_kvec_outer", "ak,kb->ab", dict(a=256, b=256, k=128), ("k_vec", "outer_vec")),
    ("nt_kvec_kvec", "ak,bk->ab", dict(a=200, b=136, k=72), ("k_vec", "k_vec")),
    ("tn_outer_outer", "ka,kb->ab", dict(a=136, b=264, k=40), ("outer_vec", "outer_vec")),
    ("tt_outer_kvec", "ka,bk->ab", dict(a=128, b=128, k=64), ("outer_vec", "k_vec")),
    ("ragged_k_tail", "ak,kb->ab", dict(a=130, b=72, k=200), ("k_vec", "outer_vec")),
    ("odd_everything_scalar", "ak,kb->ab", dict(a=67, b=45, k=77), ("scalar", "scalar")),
    ("odd_tn_scalar", "ka,kb->ab", dict(a=67, b=45, k=77), ("scalar", "scalar")),
    ("scalar_a_vec_b", "ak,kb->ab", dict(a=64, b=64, k=50), ("scalar", "outer_vec")),
    ("c_transposed", "ak,kb->ba", dict(a=72, b=136, k=24), None),
    ("k_shorter_than_a_vector", "ak,kb->ab", dict(a=64, b=64, k=2), None),
    ("single_element", "ak,kb->ab", dict(a=1, b=1, k=1), None),
    ("outer_product", "a,b->ab", dict(a=70, b=136), None),
    ("matvec", "ak,k->a", dict(a=300, k=72), None),
    ("batched", "zak,zkb->zab", dict(z=5, a=40, b=72, k=24), None),
    ("batch_fastest", "akz,kbz->abz", dict(z=6, a=20, b=24, k=10), None),
    ("ccsd_16", "aeim,ebmj->abij", dict(a=16, b=16, i=16, j=16, e=16, m=16), ("k_vec", "outer_vec")),
    ("ccsd_ragged", "aeim,ebmj->abij", dict(a=13, b=11, i=10, j=14, e=7, m=6), None),
    ("six_index", "adij,bcdk->abcijk", dict(a=6, b=5, c=4, i=6, j=8, k=8, d=16), None),
    ("adversarial_modes", "dkca,bk->abcd", dict(a=32, b=8, c=12, d=32, k=8), None),
    ("long_k", "ak,bk->ab", dict(a=128, b=128, k=4096), ("k_vec", "k_vec")),
    # fewer tiles than workgroup slots: every tile is cut along K, f32 partial sums, fix-up kernel
    ("splitk_medium", "ak,kb->ab", dict(a=512, b=384, k=4096), ("k_vec", "outer_vec")),
    ("splitk_ragged_k_tail", "ak,bk->ab", dict(a=250, b=201, k=3064), ("k_vec", "k_vec")),
    ("splitk_scalar", "ak,kb->ab", dict(a=130, b=67, k=2221), ("scalar", "scalar")),
    ("splitk_permuted", "aeim,ebmj->abij", dict(a=8, b=8, i=16, j=16, e=48, m=64), None),
    # batched launches: one list of tiles over the entries, cut along K like a single product's
    ("batched_splitk", "zak,zkb->zab", dict(z=3, a=200, b=264, k=4096), ("k_vec", "outer_vec")),
    ("batched_splitk_scalar", "zak,zbk->zab", dict(z=5, a=131, b=67, k=2221), ("scalar", "scalar")),
    ("batched_splitk_middle_mode", "azk,kzb->azb", dict(z=4, a=136, b=200, k=3072), None),
]


@pytest.mark.parametrize("name,spec,ext,layouts", CASES, ids=[c[0] for c in CASES])
def test_bf16_against_oracle(cuda, oracle, name, spec, ext, layouts):
    na, nb, nc = _sizes(spec, ext)
    A = oracle.fill(na, 0xC7C0A, BF16)
    B = oracle.fill(nb, 0xC7C0B, BF16)
    plan = ctc_amd.Plan(spec, ext, "bf16")
    s = plan.summary()
    assert s["kernel"] == "mfma_gemm" and tuple(s["tile"]) == (128, 128, 64)
    if layouts is not None:
        assert (s["layout_a"], s["layout_b"]) == layouts
    dC = torch.full((nc,), float("nan"), dtype=torch.bfloat16, device=cuda)
    plan.execute(_to_dev(A, cuda), _to_dev(B, cuda), dC)
    _check(oracle, _to_host(dC), oracle.contract(spec, ext, A, B, dtype=BF16))


# The 256 x 256 tile (512 threads, one workgroup per CU) is chosen on its own once a problem has
# enough tiles of that size for the chip; CTC_TC_FORCE_TILE=256 selects it for small problems.
BIG_TILE_CASES = [
    ("nn_ragged", "ak,kb->ab", dict(a=520, b=392, k=200), ("k_vec", "outer_vec")),
    ("nt_ragged", "ak,bk->ab", dict(a=264, b=648, k=72), ("k_vec", "k_vec")),
    ("tn_ragged", "ka,kb->ab", dict(a=392, b=520, k=136), ("outer_vec", "outer_vec")),
    ("tt_one_tile", "ka,bk->ab", dict(a=256, b=256, k=64), ("outer_vec", "k_vec")),
    ("smaller_than_a_tile", "ak,kb->ab", dict(a=40, b=72, k=24), ("k_vec", "outer_vec")),
    ("k_tail_shorter_than_a_tile", "ak,kb->ab", dict(a=264, b=264, k=104), None),
    ("splitk", "ak,kb->ab", dict(a=768, b=520, k=4096), ("k_vec", "outer_vec")),
    ("splitk_nt_k_tail", "ak,bk->ab", dict(a=300, b=264, k=3064), ("k_vec", "k_vec")),
    ("ccsd_24", "aeim,ebmj->abij", dict(a=24, b=24, i=24, j=24, e=16, m=24), None),
    ("batched", "zak,zkb->zab", dict(z=3, a=264, b=328, k=72), None),
    ("batched_splitk", "zak,zkb->zab", dict(z=3, a=520, b=264, k=4096), None),
]


@pytest.mark.parametrize("name,spec,ext,layouts", BIG_TILE_CASES, ids=[c[0] for c in BIG_TILE_CASES])
def test_bf16_big_tile_forced(cuda, oracle, monkeypatch, name, spec, ext, layouts):
    monkeypatch.setenv("CTC_TC_FORCE_TILE", "256")
    na, nb, nc = _sizes(spec, ext)
    A = oracle.fill(na, 0xC7C0A, BF16)
    B = oracle.fill(nb, 0xC7C0B, BF16)
    plan = ctc_amd.Plan(spec, ext, "bf16")
    s = plan.summary()
    assert s["kernel"] == "mfma_gemm" and tuple(s["tile"]) == (256, 256, 64), s
    if layouts is not None:
        assert (s["layout_a"], s["layout_b"]) == layouts
    C0 = oracle.fill(nc, 99, BF16)
    dC = _to_dev(C0.copy(), cuda)
    plan.execute(_to_dev(A, cuda), _to_dev(B, cuda), dC, alpha=2.0, beta=0.5)
    _check(oracle, _to_host(dC), oracle.contract(spec, ext, A, B, C=C0.copy(), alpha=2.0, beta=0.5, dtype=BF16))


def test_bf16_big_tile_forced_on_scalar_layout_stays_small(monkeypatch):
    monkeypatch.setenv("CTC_TC_FORCE_TILE", "256")
    s = ctc_amd.Plan("ak,kb->ab", dict(a=67, b=45, k=77), "bf16").summary()
    assert tuple(s["tile"]) == (128, 128, 64)


def test_bf16_big_tile_chosen_and_exact_on_small_integers(cuda):
    """16 x 12 tiles of 256: the planner takes the big tile by itself.  Integer operands make the
    result exact, so it must equal torch's f64 product bit for bit, in all four operand layouts."""
    m, n, k = 4096, 3072, 96
    g = torch.Generator().manual_seed(11)
    A = torch.randint(-2, 3, (m, k), generator=g).to(torch.float32)
    B = torch.randint(-1, 2, (k, n), generator=g).to(torch.float32)
    want = A.double() @ B.double()
    assert want.abs().max() < 256
    for spec, a, b in (("ak,kb->ab", A, B), ("ka,kb->ab", A.t().contiguous(), B),
                       ("ak,bk->ab", A, B.t().contiguous()), ("ka,bk->ab", A.t().contiguous(), B.t().contiguous())):
        plan = ctc_amd.Plan(spec, dict(a=m, b=n, k=k), "bf16")
        assert tuple(plan.summary()["tile"]) == (256, 256, 64), spec
        dA = a.to(cuda).to(torch.bfloat16).reshape(-1)
        dB = b.to(cuda).to(torch.bfloat16).reshape(-1)
        dC = torch.full((m * n,), float("nan"), dtype=torch.bfloat16, device=cuda)
        plan.execute(dA, dB, dC)
        assert torch.equal(dC.double().cpu().reshape(m, n), want), spec


def test_bf16_big_tile_unaligned_base_runs_on_small_tile(cuda, oracle, monkeypatch):
    """A plan made for the big tile whose operand base turns out unaligned at execution time is
    demoted to element loads, which only the 128 tile has."""
    monkeypatch.setenv("CTC_TC_FORCE_TILE", "256")
    spec, ext = "ak,kb->ab", dict(a=264, b=136, k=64)
    na, nb, nc = _sizes(spec, ext)
    A = oracle.fill(na + 1, 5, BF16)
    B = oracle.fill(nb, 6, BF16)
    want = oracle.contract(spec, ext, A[1:].copy(), B, dtype=BF16)
    plan = ctc_amd.Plan(spec, ext, "bf16")
    assert tuple(plan.summary()["tile"]) == (256, 256, 64)
    dC = torch.full((nc,), float("nan"), dtype=torch.bfloat16, device=cuda)
    plan.execute(_to_dev(A, cuda)[1:], _to_dev(B, cuda), dC)
    _check(oracle, _to_host(dC), want)


def test_bf16_golden_vector(cuda, oracle, golden):
    """The committed numpy.einsum samples of the bf16 case."""
    n = 0
    for c in golden["cases"]:
        if c["dtype"] != "bf16":
            continue
        na, nb, nc = _sizes(c["spec"], c["extents"])
        A = oracle.fill(na, c["seed_a"], BF16)
        B = oracle.fill(nb, c["seed_b"], BF16)
        dC = torch.full((nc,), float("nan"), dtype=torch.bfloat16, device=cuda)
        ctc_amd.Plan(c["spec"], c["extents"], "bf16").execute(_to_dev(A, cuda), _to_dev(B, cuda), dC)
        got = oracle.bf16_to_f32(_to_host(dC)).astype(np.float64)
        scale = float.fromhex(c["max_abs"])
        for i, hx in c["samples"]:
            assert abs(float(got[i]) - float.fromhex(hx)) <= TOL * scale, (c["name"], i)
        n += 1
    assert n >= 1


@pytest.mark.parametrize("spec,ext", [
    ("ak,kb->ab", dict(a=200, b=136, k=72)),            # 8-byte stores of C
    ("ak,kb->ab", dict(a=67, b=45, k=77)),              # element stores
    ("ak,kb->ab", dict(a=300, b=264, k=3072)),          # split along K: the fix-up kernel applies alpha and beta
])
def test_bf16_alpha_beta(cuda, oracle, spec, ext):
    na, nb, nc = _sizes(spec, ext)
    A = oracle.fill(na, 0xC7C0A, BF16)
    B = oracle.fill(nb, 0xC7C0B, BF16)
    C0 = oracle.fill(nc, 99, BF16)
    dC = _to_dev(C0.copy(), cuda)
    ctc_amd.Plan(spec, ext, "bf16").execute(_to_dev(A, cuda), _to_dev(B, cuda), dC, alpha=0.5, beta=-1.5)
    want = oracle.contract(spec, ext, A, B, C=C0.copy(), alpha=0.5, beta=-1.5, dtype=BF16)
    _check(oracle, _to_host(dC), want)


def test_bf16_unaligned_bases_fall_back_to_element_access(cuda, oracle):
    spec, ext = "ak,kb->ab", dict(a=136, b=72, k=64)
    na, nb, nc = _sizes(spec, ext)
    A = oracle.fill(na + 1, 5, BF16)
    B = oracle.fill(nb + 1, 6, BF16)
    want = oracle.contract(spec, ext, A[1:].copy(), B[1:].copy(), dtype=BF16)
    plan = ctc_amd.Plan(spec, ext, "bf16")
    assert plan.summary()["layout_a"] == "k_vec" and plan.summary()["vec_c"] == 8
    buf = torch.full((nc + 1,), float("nan"), dtype=torch.bfloat16, device=cuda)
    plan.execute(_to_dev(A, cuda)[1:], _to_dev(B, cuda)[1:], buf[1:])
    _check(oracle, _to_host(buf[1:].contiguous()), want)
    assert torch.isnan(buf[0])


def test_bf16_exact_on_small_integers(cuda):
    """Integer-valued operands: every product and sum is exact in f32 and the results fit bf16,
    so the result must equal the integer contraction bit for bit (catches operand-map mistakes
    that a tolerance would hide).  B is asymmetric on purpose."""
    m, n, k = 192, 160, 96
    g = torch.Generator().manual_seed(7)
    A = torch.randint(-2, 3, (m, k), generator=g).to(torch.float32)
    B = torch.randint(-1, 2, (k, n), generator=g).to(torch.float32)
    want = A.double() @ B.double()
    assert want.abs().max() < 256  # integers below 2^8 are exact in bf16
    for spec, a, b in (("ak,kb->ab", A, B), ("ka,kb->ab", A.t().contiguous(), B),
                       ("ak,bk->ab", A, B.t().contiguous()), ("ka,bk->ab", A.t().contiguous(), B.t().contiguous())):
        dA = a.to(cuda).to(torch.bfloat16).reshape(-1)
        dB = b.to(cuda).to(torch.bfloat16).reshape(-1)
        dC = torch.full((m * n,), float("nan"), dtype=torch.bfloat16, device=cuda)
        ctc_amd.Plan(spec, dict(a=m, b=n, k=k), "bf16").execute(dA, dB, dC)
        assert torch.equal(dC.double().cpu().reshape(m, n), want), spec


def test_bf16_linearity_at_size(cuda):
    """2048^3: contract(2A, B) == 2 contract(A, B) exactly (power-of-two scaling commutes with
    every rounding), and a row block recomputed in f64 by torch agrees within tolerance."""
    n = 2048
    dA = torch.empty(n * n, dtype=torch.bfloat16, device=cuda)
    dB = torch.empty(n * n, dtype=torch.bfloat16, device=cuda)
    ctc_amd.fill(dA, 0xC7C0A)
    ctc_amd.fill(dB, 0xC7C0B)
    plan = ctc_amd.Plan("ak,kb->ab", dict(a=n, b=n, k=n), "bf16")
    dC = torch.empty(n * n, dtype=torch.bfloat16, device=cuda)
    dC2 = torch.empty_like(dC)
    plan.execute(dA, dB, dC)
    plan.execute(dA * 2, dB, dC2)
    assert torch.equal(dC2, dC * 2)
    rows = dA.reshape(n, n)[100:132].double() @ dB.reshape(n, n).double()
    got = dC.reshape(n, n)[100:132].double()
    assert (got - rows).abs().max() <= TOL * rows.abs().max()


# Two operands read in 16-byte vectors, the 256 tile and a K that is a multiple of 32 with k offsets affine
# inside aligned groups of 32 take the ring kernel (K-tiles of 32 in four stages, LDS-DMA issued from
# assembly).  An operand is K-major ("k_vec": unit stride along K) or outer-major ("outer_vec": unit stride along
# its rows, read back through the transposing LDS loads).
RING_CASES = [
    ("one_k_tile", "ak,bk->ab", dict(a=264, b=648, k=32)),
    ("outer_b_one_k_tile", "ak,kb->ab", dict(a=264, b=648, k=32)),
    ("outer_b_ragged_edges", "ak,kb->ab", dict(a=520, b=392, k=96)),
    ("outer_b_splitk", "ak,kb->ab", dict(a=300, b=264, k=3072)),
    ("outer_a", "ka,bk->ab", dict(a=392, b=300, k=160)),
    ("outer_a_ragged_edges", "ka,bk->ab", dict(a=520, b=777, k=96)),
    ("outer_both", "ka,kb->ab", dict(a=392, b=520, k=224)),
    ("outer_both_splitk", "ka,kb->ab", dict(a=264, b=264, k=2048)),
    ("outer_b_headline_shape", "aeim,ebmj->abij", dict(a=12, b=10, e=3, i=24, j=32, m=32)),
    ("outer_b_two_contracted_modes", "aef,efb->ab", dict(a=392, b=264, e=3, f=64)),
    ("outer_both_batched", "zka,zkb->zab", dict(z=3, a=264, b=328, k=64)),
    ("outer_b_permuted_result", "ak,kb->ba", dict(a=264, b=328, k=64)),
    # k offsets affine inside aligned groups of 16 only: one table entry per half of a K-tile
    ("k16_kmajor_both", "aef,ebf->ab", dict(a=392, b=264, e=6, f=16)),
    ("k16_kmajor_both_48", "aef,ebf->ab", dict(a=300, b=520, e=2, f=48)),
    ("k16_headline_shape", "aeim,ebmj->abij", dict(a=12, b=10, e=4, i=24, j=32, m=16)),
    ("k16_headline_shape_80", "aeim,ebmj->abij", dict(a=5, b=6, e=2, i=56, j=48, m=80)),
    ("k16_outer_a", "efa,ebf->ab", dict(a=392, b=264, e=4, f=16)),
    ("k16_splitk", "aef,ebf->ab", dict(a=300, b=264, e=64, f=48)),
    ("three_k_tiles_ragged_edges", "ak,bk->ab", dict(a=520, b=392, k=96)),
    ("seven_k_tiles", "ak,bk->ab", dict(a=300, b=777, k=224)),
    ("splitk", "ak,bk->ab", dict(a=300, b=264, k=3072)),
    ("two_contracted_modes", "aef,bef->ab", dict(a=392, b=264, e=3, f=64)),
    ("modes_that_do_not_fuse", "aef,ebf->ab", dict(a=392, b=264, e=3, f=32)),
    ("batched", "zak,zbk->zab", dict(z=3, a=264, b=328, k=64)),
    ("permuted_result", "ak,bk->ba", dict(a=264, b=328, k=64)),
]


@pytest.mark.parametrize("name,spec,ext", RING_CASES, ids=[c[0] for c in RING_CASES])
def test_bf16_ring_kernel(cuda, oracle, monkeypatch, name, spec, ext):
    monkeypatch.setenv("CTC_TC_FORCE_TILE", "256")
    na, nb, nc = _sizes(spec, ext)
    A = oracle.fill(na, 0xC7C0A, BF16)
    B = oracle.fill(nb, 0xC7C0B, BF16)
    plan = ctc_amd.Plan(spec, ext, "bf16")
    s = plan.summary()
    assert tuple(s["tile"]) == (256, 256, 64) and "scalar" not in (s["layout_a"], s["layout_b"]), s
    if name.startswith("outer_"):
        assert "outer_vec" in (s["layout_a"], s["layout_b"]), s
    dC = torch.full((nc,), float("nan"), dtype=torch.bfloat16, device=cuda)
    before = ctc_amd.get_option("launches_bf16_ring")
    plan.execute(_to_dev(A, cuda), _to_dev(B, cuda), dC)
    assert ctc_amd.get_option("launches_bf16_ring") == before + 1
    want = oracle.contract(spec, ext, A, B, dtype=BF16)
    _check(oracle, _to_host(dC), want)
    # alpha and beta on top of that result (one more rounding: the fall-back epilogue)
    C0 = _to_host(dC).copy()
    plan.execute(_to_dev(A, cuda), _to_dev(B, cuda), dC, alpha=0.5, beta=-1.0)
    _check(oracle, _to_host(dC), oracle.contract(spec, ext, A, B, C=C0, alpha=0.5, beta=-1.0, dtype=BF16))


def test_bf16_ring_kernel_is_left_alone_when_it_does_not_apply(cuda, oracle, monkeypatch):
    monkeypatch.setenv("CTC_TC_FORCE_TILE", "256")
    for spec, ext in (("ak,bk->ab", dict(a=264, b=264, k=72)),      # K no multiple of 32
                      ("ak,kb->ab", dict(a=264, b=262, k=64)),      # an operand without 16-byte vectors
                      ("aeim,ebmj->abij", dict(a=12, b=12, e=4, i=24, j=24, m=24)),  # groups of 32 straddle m
                      ("aef,ebf->ab", dict(a=264, b=264, e=4, f=24))):  # groups of 32 straddle a mode of B
        na, nb, nc = _sizes(spec, ext)
        A = oracle.fill(na, 0xC7C0A, BF16)
        B = oracle.fill(nb, 0xC7C0B, BF16)
        dC = torch.full((nc,), float("nan"), dtype=torch.bfloat16, device=cuda)
        before = ctc_amd.get_option("launches_bf16_ring")
        ctc_amd.Plan(spec, ext, "bf16").execute(_to_dev(A, cuda), _to_dev(B, cuda), dC)
        assert ctc_amd.get_option("launches_bf16_ring") == before, spec
        _check(oracle, _to_host(dC), oracle.contract(spec, ext, A, B, dtype=BF16))


def test_bf16_ring_kernel_seeded_random(cuda, oracle, monkeypatch):
    """Random operand layouts (K-major / outer-major each), ragged tile edges, one to three contracted modes,
    alpha and beta on the 256 tile: the launches that qualify for the ring kernel must take it, and every
    result must match the oracle."""
    import os
    import random
    monkeypatch.setenv("CTC_TC_FORCE_TILE", "256")
    rng = random.Random(0xBF16)
    took = 0
    n_cases = int(os.environ.get("CTC_BF16_RING_FUZZ_CASES", "24"))
    for n in range(n_cases):
        ext = dict(a=8 * rng.randint(30, 70), b=8 * rng.randint(30, 70), k=32 * rng.randint(1, 6))
        ka, kb = ["k"], ["k"]
        if rng.random() < 0.5:  # a second contracted mode, slower than k in both operands
            ext["e"] = rng.randint(2, 3)
            ka, kb = ["e", "k"], ["e", "k"]
        a_modes = ["a"] + ka if rng.random() < 0.5 else ka + ["a"]
        b_modes = ["b"] + kb if rng.random() < 0.5 else kb + ["b"]
        c_modes = rng.choice(["ab", "ba"])
        spec = "%s,%s->%s" % ("".join(a_modes), "".join(b_modes), c_modes)
        alpha, beta = rng.choice([1.0, -0.5, 2.0]), rng.choice([0.0, 0.0, 1.0, -1.5])
        na, nb, nc = _sizes(spec, ext)
        A = oracle.fill(na, 100 + n, BF16)
        B = oracle.fill(nb, 200 + n, BF16)
        C0 = oracle.fill(nc, 300 + n, BF16)
        plan = ctc_amd.Plan(spec, ext, "bf16")
        s = plan.summary()
        assert tuple(s["tile"]) == (256, 256, 64), (spec, ext, s)
        dC = _to_dev(C0, cuda)
        before = ctc_amd.get_option("launches_bf16_ring")
        plan.execute(_to_dev(A, cuda), _to_dev(B, cuda), dC, alpha=alpha, beta=beta)
        took += ctc_amd.get_option("launches_bf16_ring") - before
        want = oracle.contract(spec, ext, A, B, C=C0.copy(), alpha=alpha, beta=beta, dtype=BF16)
        try:
            _check(oracle, _to_host(dC), want)
        except AssertionError as e:
            raise AssertionError("%s %s alpha %g beta %g layouts %s %s: %s" % (spec, ext, alpha, beta, s["layout_a"], s["layout_b"], e))
    assert took == n_cases, (took, n_cases)
