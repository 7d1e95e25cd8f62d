"""The fp64 / fp32 GEMM kernel that stages its operands by LDS-DMA (128 x 128 tiles, K-tiles of 8 fp64 or 16
fp32 contracted elements) against the CPU oracle and against the register-staged kernel, through the C ABI.  Both kernels add the products of
their K-tiles in k order into the same accumulators, so the two agree to the parity tolerance (1e-12 fp64 /
1e-5 fp32 of max |C|); index and table outputs are the planner's and are compared bit for bit elsewhere."""
import os

import numpy as np
import pytest
import torch

import ctc_amd

pytestmark = pytest.mark.gpu

TOL = {"f64": 1e-12, "f32": 1e-5}
TDT = {"f64": torch.float64, "f32": torch.float32}
CODE = {"f64": 0, "f32": 1}
NAMES = ("launches_classic", "launches_dma_gemm", "launches_stream", "launches_stream_f64_dma")


def _delta(fn):
    before = [ctc_amd.get_option(n) for n in NAMES]
    fn()
    return tuple(ctc_amd.get_option(n) - b for n, b in zip(NAMES, before))


def _operands(oracle, spec, ext, dev, dtype="f64"):
    ins, outm = spec.split("->")
    la, lb = ins.split(",")
    A = oracle.fill(int(np.prod([ext[c] for c in la])), 0xC7C0A, CODE[dtype])
    B = oracle.fill(int(np.prod([ext[c] for c in lb])), 0xC7C0B, CODE[dtype])
    return A, B, int(np.prod([ext[c] for c in outm])), torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)


def _check(got, want, dtype="f64"):
    got, want = got.astype(np.float64), want.astype(np.float64)
    err = np.abs(got - want).max()
    assert err <= TOL[dtype] * np.abs(want).max(), "max err %.3e vs scale %.3e" % (err, np.abs(want).max())


CASES = [
    # every operand layout pair, ragged edges in M and N, one K-tile to many (the request ring wraps when
    # a workgroup has fewer than three K-tiles)
    ("nn_kmajor_outer", "ak,kb->ab", dict(a=2500, b=2600, k=64)),
    ("nt_kmajor_kmajor", "ak,bk->ab", dict(a=2562, b=2430, k=40)),
    ("tn_outer_outer", "ka,kb->ab", dict(a=2600, b=2498, k=56)),
    ("tt_outer_kmajor", "ka,bk->ab", dict(a=2434, b=2690, k=72)),
    ("one_k_tile", "ak,kb->ab", dict(a=2500, b=2600, k=8)),
    ("two_k_tiles", "ka,kb->ab", dict(a=2500, b=2600, k=16)),
    ("three_k_tiles", "ak,bk->ab", dict(a=2500, b=2600, k=24)),
    ("five_k_tiles", "ka,bk->ab", dict(a=2500, b=2600, k=40)),
    # contracted modes that do not fuse: affine only inside groups of 8 (the headline contraction's shape)
    ("ccsd_three_by_24", "aeim,ebmj->abij", dict(a=52, b=50, i=50, j=52, e=3, m=24)),
    ("ccsd_uneven", "aeim,ebmj->abij", dict(a=50, b=44, i=52, j=60, e=5, m=16)),
    ("ccsd_other_layouts", "eaim,jbem->abij", dict(a=48, b=50, i=54, j=52, e=3, m=8)),
    # the last, partly filled round of tiles is cut along K: partial sums through the workspace
    ("split_k_tail", "ak,kb->ab", dict(a=1024, b=1152, k=4096)),
    ("split_k_everything", "ka,bk->ab", dict(a=256, b=384, k=32768)),
    # batch entries run as grid.y
    ("batched", "zak,zkb->zab", dict(z=3, a=1500, b=1400, k=32)),
    ("rank6", "adij,bcdk->abcijk", dict(a=6, b=20, c=20, i=20, j=22, k=8, d=24)),
    # K (or the k tables' affine groups) a multiple of 4 only: K-tiles of 4, one k-step each (from two rounds
    # of tiles on)
    ("k4_nn_kmajor_outer", "ak,kb->ab", dict(a=4100, b=4200, k=44)),
    ("k4_nt_kmajor_kmajor", "ak,bk->ab", dict(a=4164, b=4032, k=20)),
    ("k4_tn_outer_outer", "ka,kb->ab", dict(a=4200, b=4100, k=28)),
    ("k4_tt_outer_kmajor", "ka,bk->ab", dict(a=4036, b=4292, k=12)),
    ("k4_one_k_tile", "ak,kb->ab", dict(a=4100, b=4200, k=4)),
    ("k4_five_k_tiles", "ak,bk->ab", dict(a=4100, b=4200, k=20)),
    ("k4_ccsd_three_by_12", "aeim,ebmj->abij", dict(a=82, b=84, i=50, j=52, e=3, m=12)),
    # K a multiple of 8 but the k tables affine in groups of 4 only: two images of 32-byte rows per stage
    ("k4x2_ccsd_four_by_12", "aeim,ebmj->abij", dict(a=52, b=50, i=50, j=52, e=4, m=12)),
    ("k4x2_ccsd_other_layouts", "eaim,jbem->abij", dict(a=48, b=50, i=54, j=52, e=6, m=4)),
    ("k4x2_modes_do_not_fuse", "aef,feb->ab", dict(a=2600, b=2500, e=6, f=12)),
    ("k4x2_outer_operands", "efa,feb->ab", dict(a=2600, b=2500, e=6, f=12)),
    ("k4x2_split_k_tail", "aef,feb->ab", dict(a=1024, b=1152, e=340, f=12)),
    ("k4_split_k_tail", "ak,kb->ab", dict(a=4224, b=4480, k=1036)),
    # k tables affine in runs that are no multiple of 4: ragged K-tiles of 8 (the last tile of a run overlaps
    # its predecessor; the overlap is zeroed in the fragments)
    ("ragged_nn_kmajor_outer", "ak,kb->ab", dict(a=2500, b=2600, k=62)),
    ("ragged_nt_kmajor_kmajor", "ak,bk->ab", dict(a=2562, b=2430, k=30)),
    ("ragged_tn_outer_outer", "ka,kb->ab", dict(a=2600, b=2498, k=22)),
    ("ragged_tt_outer_kmajor", "ka,bk->ab", dict(a=2434, b=2690, k=14)),
    ("ragged_ccsd_three_by_22", "aeim,ebmj->abij", dict(a=52, b=50, i=50, j=52, e=3, m=22)),
    ("ragged_ccsd_other_layouts", "eaim,jbem->abij", dict(a=48, b=50, i=54, j=52, e=5, m=14)),
    ("ragged_split_k_tail", "aef,feb->ab", dict(a=1024, b=1152, e=150, f=22)),
    ("ragged_split_k_everything", "ka,bk->ab", dict(a=256, b=384, k=32774)),
    ("ragged_batched", "zak,zkb->zab", dict(z=3, a=1500, b=1400, k=38)),
]


# fp32: K-tiles of 16, or of 8 when K (or the k tables' affine groups) is a multiple of 8 only
F32_CASES = [
    ("k8_nn_kmajor_outer", "ak,kb->ab", dict(a=4100, b=4200, k=72)),
    ("k8_nt_kmajor_kmajor", "ak,bk->ab", dict(a=4164, b=4032, k=40)),
    ("k8_tn_outer_outer", "ka,kb->ab", dict(a=4200, b=4100, k=56)),
    ("k8_tt_outer_kmajor", "ka,bk->ab", dict(a=4036, b=4292, k=24)),
    ("k8_one_k_tile", "ak,kb->ab", dict(a=4100, b=4200, k=8)),
    ("k8_ccsd_three_by_24", "aeim,ebmj->abij", dict(a=82, b=84, i=50, j=52, e=3, m=24)),
    ("k8x2_ccsd_two_by_24", "aeim,ebmj->abij", dict(a=52, b=50, i=50, j=52, e=2, m=24)),
    ("k8x2_modes_do_not_fuse", "aef,feb->ab", dict(a=2600, b=2500, e=6, f=24)),
    ("k8x2_outer_operands", "efa,feb->ab", dict(a=2600, b=2500, e=6, f=24)),
    ("k8x2_split_k_tail", "aef,feb->ab", dict(a=1024, b=1152, e=170, f=24)),
    ("k8_split_k_tail", "ka,bk->ab", dict(a=4224, b=4480, k=1032)),
    ("nn_kmajor_outer", "ak,kb->ab", dict(a=2500, b=2600, k=64)),
    ("nt_kmajor_kmajor", "ak,bk->ab", dict(a=2564, b=2432, k=48)),
    ("tn_outer_outer", "ka,kb->ab", dict(a=2600, b=2500, k=80)),
    ("tt_outer_kmajor", "ka,bk->ab", dict(a=2436, b=2692, k=32)),
    ("one_k_tile", "ka,kb->ab", dict(a=2500, b=2600, k=16)),
    ("ccsd_three_by_32", "aeim,ebmj->abij", dict(a=52, b=50, i=50, j=52, e=3, m=32)),
    ("ccsd_other_layouts", "eaim,jbem->abij", dict(a=48, b=50, i=54, j=52, e=3, m=16)),
    ("split_k_tail", "ak,kb->ab", dict(a=1024, b=1152, k=4096)),
    ("split_k_everything", "ka,bk->ab", dict(a=256, b=384, k=32768)),
    ("batched", "zak,zkb->zab", dict(z=3, a=1500, b=1400, k=32)),
    ("rank6", "adij,bcdk->abcijk", dict(a=6, b=20, c=20, i=20, j=24, k=8, d=32)),
    # ragged K-tiles of 16
    ("ragged_nn_kmajor_outer", "ak,kb->ab", dict(a=2500, b=2600, k=28)),
    ("ragged_tt_outer_kmajor", "ka,bk->ab", dict(a=2436, b=2692, k=44)),
    ("ragged_nt_kmajor_kmajor", "ak,bk->ab", dict(a=2564, b=2432, k=60)),
    ("ragged_ccsd_three_by_28", "aeim,ebmj->abij", dict(a=52, b=50, i=50, j=52, e=3, m=28)),
    ("ragged_split_k_tail", "ka,kb->ab", dict(a=1024, b=1152, k=4092)),
]
ALL_CASES = [c + ("f64",) for c in CASES] + [c + ("f32",) for c in F32_CASES]


@pytest.mark.parametrize("name,spec,ext,dtype", ALL_CASES, ids=[c[3] + "_" + c[0] for c in ALL_CASES])
def test_dma_gemm_against_oracle_and_register_staged(cuda, oracle, name, spec, ext, dtype):
    A, B, n_out, dA, dB = _operands(oracle, spec, ext, cuda, dtype)
    plan = ctc_amd.Plan(spec, ext, dtype)
    s = plan.summary()
    assert s["kernel"] == "mfma_gemm" and s["tile"][:2] == (128, 128), s
    assert s["layout_a"] != "scalar" and s["layout_b"] != "scalar", s
    ctc_amd.set_option("stream_mode", 1)  # the persistent kernel is another test's business
    try:
        dC = torch.full((n_out,), float("nan"), dtype=TDT[dtype], device=cuda)
        took = _delta(lambda: plan.execute(dA, dB, dC, alpha=1.25))
        assert took == (0, 1, 0, 0), took
        want = oracle.contract(spec, ext, A, B, alpha=1.25, dtype=CODE[dtype])
        got = dC.cpu().numpy()
        assert not np.isnan(got).any()  # every element of C was stored
        _check(got, want, dtype)
        # beta != 0 on top of the first result
        want2 = 0.5 * want.astype(np.float64) + 0.75 * want.astype(np.float64) / 1.25
        plan.execute(dA, dB, dC, alpha=0.75, beta=0.5)
        _check(dC.cpu().numpy(), want2, dtype)
        ctc_amd.set_option("dma_gemm", 0)
        dC2 = torch.full((n_out,), float("nan"), dtype=TDT[dtype], device=cuda)
        took = _delta(lambda: plan.execute(dA, dB, dC2, alpha=1.25))
        assert took == (1, 0, 0, 0), took
        _check(dC2.cpu().numpy(), want, dtype)
    finally:
        ctc_amd.set_option("dma_gemm", 1)
        ctc_amd.set_option("stream_mode", 0)


# Stream-K shares: the K-tile iterations of the last one to two rounds of tiles dealt out in equal runs.  Each
# case names the schedule it must produce when shares are forced: tiles run whole, tiles shared out.
STREAM_K_CASES = [
    # fewer tiles than workgroup slots: shares shorter than a tile, several slabs per tile
    ("fewer_tiles_than_slots", "ak,kb->ab", dict(a=1024, b=1024, k=4096), "f64"),
    ("fewer_tiles_ragged_edges", "ka,bk->ab", dict(a=1000, b=900, k=4096), "f64"),
    # between one and two rounds: every share is longer than a tile, a tile is cut at most once
    ("one_round_and_a_bit", "ak,kb->ab", dict(a=3072, b=3200, k=1024), "f64"),
    ("one_round_and_a_bit_ragged_edges", "ak,bk->ab", dict(a=3100, b=3130, k=1000), "f64"),
    # two full rounds and a bit: the first round runs whole tiles
    ("two_rounds_and_a_bit", "ka,kb->ab", dict(a=3200, b=5632, k=512), "f64"),
    # K-tiles that are not 64-byte rows: two images of 32-byte rows (modes of 60: groups of 4), ragged K-tiles (runs of 14)
    ("rows_of_32_bytes", "aeim,ebmj->abij", dict(a=40, b=40, i=24, j=24, e=60, m=60), "f64"),
    ("ragged_k_tiles", "aef,feb->ab", dict(a=2560, b=2560, e=40, f=14), "f64"),
    ("f32_fewer_tiles", "ak,kb->ab", dict(a=1024, b=1152, k=4096), "f32"),
    ("f32_one_round_and_a_bit", "ka,kb->ab", dict(a=3072, b=3200, k=2048), "f32"),
]


@pytest.mark.parametrize("name,spec,ext,dtype", STREAM_K_CASES, ids=[c[0] for c in STREAM_K_CASES])
def test_stream_k_shares_against_oracle(cuda, oracle, name, spec, ext, dtype):
    """Forced stream-K against the oracle (alpha and beta), against the same plan with equal K chunks, and
    twice in a row bit for bit (the slabs are added in share order: no atomics, nothing depends on timing)."""
    A, B, n_out, dA, dB = _operands(oracle, spec, ext, cuda, dtype)
    C0 = oracle.fill(n_out, 0xC7C0C, CODE[dtype])
    plan = ctc_amd.Plan(spec, ext, dtype)
    ctc_amd.set_option("stream_mode", 1)
    ctc_amd.set_option("stream_k", 2)
    try:
        sch = plan.schedule()
        assert sch["dma_form"] > 0 and sch["stream_k_shares"] >= 2 and sch["tail_tiles"] > 0, sch
        assert sch["workspace_bytes"] == sch["stream_k_shares"] * 128 * 128 * (8 if dtype == "f64" else 4), sch
        dC = torch.from_numpy(C0.copy()).to(cuda)
        took = _delta(lambda: plan.execute(dA, dB, dC, alpha=0.75, beta=-1.5))
        assert took == (0, 1, 0, 0), took
        want = oracle.contract(spec, ext, A, B, C=C0.copy(), alpha=0.75, beta=-1.5, dtype=CODE[dtype])
        _check(dC.cpu().numpy(), want, dtype)
        dC1 = torch.full((n_out,), float("nan"), dtype=TDT[dtype], device=cuda)
        dC2 = torch.full((n_out,), float("nan"), dtype=TDT[dtype], device=cuda)
        plan.execute(dA, dB, dC1)
        plan.execute(dA, dB, dC2)
        assert not bool(torch.isnan(dC1).any())
        assert torch.equal(dC1, dC2)
        want1 = oracle.contract(spec, ext, A, B, dtype=CODE[dtype])
        _check(dC1.cpu().numpy(), want1, dtype)
        ctc_amd.set_option("stream_k", 1)
        assert plan.schedule()["stream_k_shares"] == 0
        dC3 = torch.full((n_out,), float("nan"), dtype=TDT[dtype], device=cuda)
        plan.execute(dA, dB, dC3)
        _check(dC3.cpu().numpy(), want1, dtype)
    finally:
        ctc_amd.set_option("stream_k", 0)
        ctc_amd.set_option("stream_mode", 0)


def _random_share_cases(count, seed=20240917):
    rng = np.random.RandomState(seed)
    specs = ("ak,kb->ab", "ak,bk->ab", "ka,kb->ab", "ka,bk->ab")
    out = []
    for i in range(count):
        dtype = "f32" if i % 4 == 3 else "f64"
        unit = 4 if dtype == "f32" else 2  # vector layouts need extents that are multiples of a 16-byte vector
        m = int(rng.randint(700, 6000)) // unit * unit
        n = int(rng.randint(700, 6000)) // unit * unit
        k = int(rng.randint(64, 384)) * 16
        out.append((specs[i % 4], dict(a=m, b=n, k=k), dtype, (1, 8, 16, 64)[(i // 4) % 4]))
    return out


# CTC_SK_FUZZ_CASES / CTC_SK_FUZZ_SEED: more shapes, another sequence (one-off campaigns beside the fixed suite)
_SHARE_CASES = _random_share_cases(int(os.environ.get("CTC_SK_FUZZ_CASES", "24")), int(os.environ.get("CTC_SK_FUZZ_SEED", "20240917"), 0))


@pytest.mark.parametrize("spec,ext,dtype,group", _SHARE_CASES,
                         ids=["%s_%dx%dx%d_%s_g%d" % (c[0].replace(",", "_").replace("->", "_"), c[1]["a"], c[1]["b"], c[1]["k"], c[2], c[3])
                              for c in _SHARE_CASES])
def test_stream_k_shares_random_shapes(cuda, spec, ext, dtype, group):
    """Seeded random shapes (tile counts from below one round of workgroup slots to several, ragged edges, all
    four layout pairs, every group size) with stream-K shares forced, against the same plan with equal K chunks:
    the two schedules add the same products in different orders, so they agree to the parity tolerance; and
    the shares leave no element of C unwritten."""
    m, n, k = ext["a"], ext["b"], ext["k"]
    dA = torch.empty(m * k, dtype=TDT[dtype], device=cuda)
    dB = torch.empty(n * k, dtype=TDT[dtype], device=cuda)
    ctc_amd.fill(dA, 0xC7C0A)
    ctc_amd.fill(dB, 0xC7C0B)
    plan = ctc_amd.Plan(spec, ext, dtype)
    if plan.summary()["tile"][:2] != (128, 128):
        pytest.skip("the planner takes the small tile for this shape")
    ctc_amd.set_option("stream_mode", 1)
    try:
        ctc_amd.set_option("stream_k", 1)
        want = torch.full((m * n,), float("nan"), dtype=TDT[dtype], device=cuda)
        plan.execute(dA, dB, want)
        ctc_amd.set_option("stream_k", 2)
        ctc_amd.set_option("stream_k_group", group)
        sch = plan.schedule()
        got = torch.full((m * n,), float("nan"), dtype=TDT[dtype], device=cuda)
        took = _delta(lambda: plan.execute(dA, dB, got, alpha=1.0))
        assert took == (0, 1, 0, 0), took
        torch.cuda.synchronize()
        assert not bool(torch.isnan(got).any()), sch
        scale = float(want.abs().max())
        assert float((got - want).abs().max()) <= TOL[dtype] * scale, sch
        tiles = sch["tile_rows"] * sch["tile_cols"]
        if tiles % 512:  # the chip's 512 workgroup slots: a ragged tile count is what shares are for
            assert sch["stream_k_shares"] >= 2, sch
    finally:
        ctc_amd.set_option("stream_k", 0)
        ctc_amd.set_option("stream_k_group", 0)
        ctc_amd.set_option("stream_mode", 0)


def test_schedule_query_reports_the_kernel_that_runs(cuda):
    """CTC_TC_Q_SCHEDULE describes the launch ctc_tc_execute makes: the LDS-DMA form for the plan's layouts
    and k tables, and the workspace that launch uses (checked through the no-allocation entry point, which
    refuses a launch that needs more than it is given: the fixed buffer must be large enough exactly when the
    reported bytes say so)."""
    cases = [
        ("aeim,ebmj->abij", dict(a=48, b=48, i=48, j=48, e=48, m=48), 1),   # K-tiles of 64-byte rows
        ("aef,feb->ab", dict(a=2304, b=2304, e=120, f=12), 2),               # runs of 12: groups of 4, two images
        ("aef,feb->ab", dict(a=2304, b=2304, e=120, f=14), 4),               # runs of 14: ragged K-tiles
        ("ak,kb->ab", dict(a=2304, b=2304, k=2305), 0),                      # odd K: element loads of A, register-staged
    ]
    for spec, ext, form in cases:
        plan = ctc_amd.Plan(spec, ext, "f64")
        plan.prepare()
        sch = plan.schedule()
        assert sch["dma_form"] == form, (spec, ext, sch)
        assert sch["workspace_bytes"] > 0 and sch["workspace_bytes_worst"] >= sch["workspace_bytes"], sch
        fits = sch["workspace_bytes"] <= ctc_amd.noalloc_scratch_bytes()
        assert fits or form == 0, sch  # the LDS-DMA kernel's schedules fit by construction; a register-staged one may not
        ins, outm = spec.split("->")
        la, lb = ins.split(",")
        num = lambda t: int(np.prod([ext[c] for c in t]))  # noqa: E731
        a = torch.zeros(num(la), dtype=torch.float64, device=cuda)
        b = torch.zeros(num(lb), dtype=torch.float64, device=cuda)
        c = torch.zeros(num(outm), dtype=torch.float64, device=cuda)
        if fits:
            took = _delta(lambda: plan.execute_noalloc(a, b, c))
            assert took == ((0, 1, 0, 0) if form else (1, 0, 0, 0)), (spec, took)
        else:
            with pytest.raises(ctc_amd.CtcError, match="needs %d bytes" % sch["workspace_bytes"]):
                plan.execute_noalloc(a, b, c)
    torch.cuda.synchronize()


def test_which_launches_take_the_dma_kernel(cuda):
    x = torch.zeros(4096 * 4096, dtype=torch.float64, device=cuda)
    c = torch.zeros(4096 * 4096, dtype=torch.float64, device=cuda)
    run = lambda spec, ext, a=x: lambda: ctc_amd.Plan(spec, ext, "f64").execute(a, x, c)  # noqa: E731
    big = dict(a=4096, b=4096)
    assert _delta(run("ak,kb->ab", dict(big, k=256))) == (0, 1, 0, 0)
    assert _delta(run("ak,kb->ab", dict(big, k=260))) == (0, 1, 0, 0)                 # K a multiple of 4: K-tiles of 4
    assert _delta(run("ak,kb->ab", dict(big, k=262))) == (0, 1, 0, 0)                 # K not a multiple of 4: ragged K-tiles of 8
    assert _delta(run("ak,kb->ab", dict(big, k=263))) == (1, 0, 0, 0)                 # odd K: no 16-byte vectors in A
    assert _delta(run("aef,feb->ab", dict(big, e=4, f=12))) == (0, 1, 0, 0)            # groups of 8 straddle a mode, of 4 do not
    assert _delta(run("aef,feb->ab", dict(big, e=4, f=10))) == (1, 0, 0, 0)            # groups of 4 straddle a mode; the last K-tile of a run of 10 would be 2 live elements of 8
    assert _delta(run("aef,feb->ab", dict(big, e=4, f=14))) == (0, 1, 0, 0)            # runs of 14: ragged K-tiles, 6 live elements of 8 in a run's last one
    assert _delta(run("aef,feb->ab", dict(big, e=40, f=6))) == (1, 0, 0, 0)            # runs shorter than a K-tile
    assert _delta(run("aef,feb->ab", dict(big, e=16, f=8))) == (0, 1, 0, 0)            # they do not
    assert _delta(run("ak,kb->ab", dict(a=512, b=512, k=256))) == (1, 0, 0, 0)         # small tile
    assert _delta(run("ak,kb->ab", dict(big, k=256), x[1:])) == (1, 0, 0, 0)           # unaligned base: element loads
    f32 = torch.zeros(4096 * 256, dtype=torch.float32, device=cuda)
    c32 = torch.zeros(4096 * 4096, dtype=torch.float32, device=cuda)
    assert _delta(lambda: ctc_amd.Plan("ak,kb->ab", dict(big, k=256), "f32").execute(f32, f32, c32)) == (0, 1, 0, 0)
    assert _delta(lambda: ctc_amd.Plan("ak,kb->ab", dict(big, k=248), "f32").execute(f32, f32, c32)) == (0, 1, 0, 0)  # K-tiles of 8
    assert _delta(lambda: ctc_amd.Plan("ak,kb->ab", dict(big, k=252), "f32").execute(f32, f32, c32)) == (0, 1, 0, 0)  # ragged K-tiles of 16
    assert _delta(lambda: ctc_amd.Plan("ak,kb->ab", dict(big, k=250), "f32").execute(f32, f32, c32)) == (1, 0, 0, 0)  # no 16-byte vectors in A
    ctc_amd.set_option("dma_gemm", 0)
    try:
        assert _delta(run("ak,kb->ab", dict(big, k=256))) == (1, 0, 0, 0)
    finally:
        ctc_amd.set_option("dma_gemm", 1)


def test_headline_contraction_at_extent_64_properties(cuda, oracle):
    """abij = aeim,ebmj at extent 64 (2 * 64^6 flops: too big for the oracle in a test): the LDS-DMA kernel
    and the register-staged kernel must agree to 1e-12, and slabs C[a0, b0, :, :] are recomputed by the
    oracle from the operand slices they depend on."""
    n = 64
    spec, ext = "aeim,ebmj->abij", dict(a=n, b=n, i=n, j=n, e=n, m=n)
    dA = torch.empty(n ** 4, dtype=torch.float64, device=cuda)
    dB = torch.empty(n ** 4, dtype=torch.float64, device=cuda)
    ctc_amd.fill(dA, 0xC7C0A)
    ctc_amd.fill(dB, 0xC7C0B)
    plan = ctc_amd.Plan(spec, ext, "f64")
    dC = torch.full((n ** 4,), float("nan"), dtype=torch.float64, device=cuda)
    assert _delta(lambda: plan.execute(dA, dB, dC)) == (0, 1, 0, 0)
    ctc_amd.set_option("dma_gemm", 0)
    try:
        dC2 = torch.full((n ** 4,), float("nan"), dtype=torch.float64, device=cuda)
        plan.execute(dA, dB, dC2)
    finally:
        ctc_amd.set_option("dma_gemm", 1)
    scale = float(dC2.abs().max())
    assert float((dC - dC2).abs().max()) <= TOL["f64"] * scale
    A = dA.cpu().numpy().reshape(n, n, n, n)
    B = dB.cpu().numpy().reshape(n, n, n, n)
    C = dC.cpu().numpy().reshape(n, n, n, n)
    for a0, b0 in ((0, 0), (17, 63), (63, 5)):
        sub = dict(a=1, b=1, i=n, j=n, e=n, m=n)
        want = oracle.contract(spec, sub, np.ascontiguousarray(A[a0:a0 + 1]).reshape(-1),
                               np.ascontiguousarray(B[:, b0:b0 + 1]).reshape(-1), dtype=0).reshape(n, n)
        assert np.abs(C[a0, b0] - want).max() <= TOL["f64"] * np.abs(want).max()


def test_seeded_random_contractions_on_large_tiles(cuda, oracle):
    """Random mode orders, extents, element types, alpha and beta on problems large enough for the
    128 x 128 tile (the generic random test draws small ones): whichever kernel the launch takes must
    match the oracle, and the LDS-DMA kernel must take a fair share of them."""
    import random
    rng = random.Random(0xD3A ^ int(os.environ.get("CTC_SK_FUZZ_SEED", "0"), 0) if "CTC_SK_FUZZ_SEED" in os.environ else 0xD3A)
    took_dma = 0
    n_cases = int(os.environ.get("CTC_DMA_FUZZ_CASES", "30"))
    for n in range(n_cases):
        dtype = rng.choice(["f64", "f64", "f32"])
        ext = dict(a=rng.choice([56, 60, 66]), i=rng.choice([48, 56, 64]),
                   b=rng.choice([56, 60, 66]), j=rng.choice([48, 56, 64]),
                   e=rng.choice([1, 2, 3, 5]), m=rng.choice([8, 16, 16, 24, 32, 40, 50, 14, 22, 28, 30, 44, 46]))
        la, lb, lc = list("aeim"), list("ebmj"), list("abij")
        for m in (la, lb, lc):
            rng.shuffle(m)
        if ext["e"] == 1 and rng.random() < 0.5:  # one contracted mode only
            la.remove("e")
            lb.remove("e")
            del ext["e"]
        spec = "%s,%s->%s" % ("".join(la), "".join(lb), "".join(lc))
        alpha, beta = rng.choice([1.0, -0.5, 2.0]), rng.choice([0.0, 0.0, 1.0, -1.5])
        size = lambda m: int(np.prod([ext[c] for c in m]))  # noqa: E731
        A = oracle.fill(size(la), 100 + n, CODE[dtype])
        B = oracle.fill(size(lb), 200 + n, CODE[dtype])
        C0 = oracle.fill(size(lc), 300 + n, CODE[dtype])
        plan = ctc_amd.Plan(spec, ext, dtype)
        dC = torch.from_numpy(C0.copy()).to(cuda)
        d = _delta(lambda: plan.execute(torch.from_numpy(A).to(cuda), torch.from_numpy(B).to(cuda), dC, alpha=alpha, beta=beta))
        took_dma += d[1]
        want = oracle.contract(spec, ext, A, B, C=C0.copy(), alpha=alpha, beta=beta, dtype=CODE[dtype])
        got = dC.cpu().numpy().astype(np.float64)
        err = np.abs(got - want.astype(np.float64)).max()
        scale = np.abs(want).max()
        assert err <= TOL[dtype] * scale, (n, spec, ext, dtype, alpha, beta, plan.summary()["tile"], d, err, scale)
    assert took_dma >= n_cases // 6, took_dma


# Batched launches whose tiles are scheduled as one list over all batch entries: the last round is cut along K
# (equal chunks or stream-K shares) although no single entry has a tail of its own.
BATCHED_TAIL_CASES = [
    ("fewer_tiles_than_slots", "zak,zkb->zab", dict(z=3, a=640, b=512, k=4096), "f64"),          # 60 tiles
    ("one_round_and_a_bit", "zak,zbk->zab", dict(z=5, a=1408, b=1300, k=1024), "f64"),            # 605 tiles
    ("two_rounds_and_a_bit", "zka,zkb->zab", dict(z=13, a=1000, b=1400, k=1536), "f64"),          # 1144 tiles
    ("batch_mode_in_the_middle", "azk,kzb->azb", dict(z=3, a=1280, b=1024, k=1280), "f64"),
    ("two_batch_modes", "yzak,yzkb->yzab", dict(y=2, z=3, a=1000, b=1100, k=384), "f64"),
    ("ragged_k_tiles", "zaef,zfeb->zab", dict(z=4, a=1280, b=1280, e=20, f=14), "f64"),
    ("f32_one_round_and_a_bit", "zak,zkb->zab", dict(z=5, a=1408, b=1300, k=2048), "f32"),
    ("f32_fewer_tiles", "zka,zbk->zab", dict(z=6, a=512, b=640, k=4096), "f32"),
]


@pytest.mark.parametrize("name,spec,ext,dtype", BATCHED_TAIL_CASES, ids=[c[3] + "_" + c[0] for c in BATCHED_TAIL_CASES])
@pytest.mark.parametrize("stream_k", [1, 2])
def test_batched_launch_balances_its_last_round(cuda, oracle, name, spec, ext, dtype, stream_k):
    A, B, n_out, dA, dB = _operands(oracle, spec, ext, cuda, dtype)
    C0 = oracle.fill(n_out, 0xC7C0C, CODE[dtype])
    plan = ctc_amd.Plan(spec, ext, dtype)
    s = plan.summary()
    assert s["kernel"] == "mfma_gemm" and s["tile"][:2] == (128, 128) and s["batch_count"] > 1, s
    ctc_amd.set_option("stream_mode", 1)
    ctc_amd.set_option("stream_k", stream_k)
    try:
        sch = plan.schedule()
        assert sch["dma_form"] > 0 and sch["batch_entries_in_one_tile_list"] == s["batch_count"], sch
        tiles = sch["tile_rows"] * sch["tile_cols"] * s["batch_count"]
        assert sch["whole_tiles"] + sch["tail_tiles"] == tiles, sch
        assert (sch["tail_tiles"] > 0) == (sch["workspace_bytes"] > 0), sch
        # the cases whose last round is worth cutting by the scheduler's costs (the others run whole tiles from one
        # list: a last round of at most one workgroup per CU is cheap, few K-tiles leave nothing to cut)
        cut = ("fewer_tiles_than_slots", "one_round_and_a_bit", "two_rounds_and_a_bit", "f32_one_round_and_a_bit", "f32_fewer_tiles")
        if name in cut:
            assert sch["tail_tiles"] > 0, sch
            assert stream_k == 2 or (sch["stream_k_shares"] == 0 and sch["split"] >= 2), sch
        if stream_k == 2 and name in cut + ("batch_mode_in_the_middle",):
            assert sch["stream_k_shares"] >= 2, sch  # forced wherever a tile has 64 K-tiles or more
        dC = torch.from_numpy(C0.copy()).to(cuda)
        took = _delta(lambda: plan.execute(dA, dB, dC, alpha=0.75, beta=-1.5))
        assert took == (0, 1, 0, 0), took
        want = oracle.contract(spec, ext, A, B, C=C0.copy(), alpha=0.75, beta=-1.5, dtype=CODE[dtype])
        _check(dC.cpu().numpy(), want, dtype)
        dC1 = torch.full((n_out,), float("nan"), dtype=TDT[dtype], device=cuda)
        dC2 = torch.full((n_out,), float("nan"), dtype=TDT[dtype], device=cuda)
        plan.execute(dA, dB, dC1)
        plan.execute(dA, dB, dC2)
        torch.cuda.synchronize()
        assert not bool(torch.isnan(dC1).any())
        assert torch.equal(dC1, dC2)
        _check(dC1.cpu().numpy(), oracle.contract(spec, ext, A, B, dtype=CODE[dtype]), dtype)
    finally:
        ctc_amd.set_option("stream_k", 0)
        ctc_amd.set_option("stream_mode", 0)


# The same on the register-staged kernels (the small tile; the large tile under element loads).
BATCHED_CLASSIC_CASES = [
    ("small_tile_long_k", "zak,zkb->zab", dict(z=6, a=200, b=180, k=2048), "f64", 64),            # 72 tiles, all cut
    ("small_tile_two_per_cu", "zak,zbk->zab", dict(z=5, a=640, b=576, k=1024), "f64", 64),        # 450 tiles
    ("small_tile_f32", "zka,zkb->zab", dict(z=7, a=256, b=192, k=4096), "f32", 64),               # 84 tiles
    ("large_tile_element_loads", "zak,zkb->zab", dict(z=3, a=1500, b=1400, k=1025), "f64", 128),  # odd K: A by elements
    ("small_tile_odd_everything", "zak,zkb->zab", dict(z=9, a=131, b=77, k=1501), "f64", 64),
]


@pytest.mark.parametrize("name,spec,ext,dtype,tile", BATCHED_CLASSIC_CASES, ids=[c[3] + "_" + c[0] for c in BATCHED_CLASSIC_CASES])
def test_batched_launch_on_register_staged_kernels(cuda, oracle, name, spec, ext, dtype, tile):
    A, B, n_out, dA, dB = _operands(oracle, spec, ext, cuda, dtype)
    C0 = oracle.fill(n_out, 0xC7C0C, CODE[dtype])
    plan = ctc_amd.Plan(spec, ext, dtype)
    s = plan.summary()
    assert s["kernel"] == "mfma_gemm" and s["tile"][:2] == (tile, tile) and s["batch_count"] > 1, s
    sch = plan.schedule()
    assert sch["dma_form"] == 0 and sch["batch_entries_in_one_tile_list"] == s["batch_count"], sch
    assert sch["whole_tiles"] + sch["tail_tiles"] == sch["tile_rows"] * sch["tile_cols"] * s["batch_count"], sch
    assert (sch["tail_tiles"] > 0) == (sch["workspace_bytes"] > 0), sch
    if name in ("small_tile_long_k", "small_tile_f32"):
        assert sch["tail_tiles"] > 0 and sch["split"] >= 2, sch
    dC = torch.from_numpy(C0.copy()).to(cuda)
    took = _delta(lambda: plan.execute(dA, dB, dC, alpha=0.75, beta=-1.5))
    assert took == (1, 0, 0, 0), took
    want = oracle.contract(spec, ext, A, B, C=C0.copy(), alpha=0.75, beta=-1.5, dtype=CODE[dtype])
    _check(dC.cpu().numpy(), want, dtype)
    dC1 = torch.full((n_out,), float("nan"), dtype=TDT[dtype], device=cuda)
    dC2 = torch.full((n_out,), float("nan"), dtype=TDT[dtype], device=cuda)
    plan.execute(dA, dB, dC1)
    plan.execute(dA, dB, dC2)
    torch.cuda.synchronize()
    assert not bool(torch.isnan(dC1).any())
    assert torch.equal(dC1, dC2)
    _check(dC1.cpu().numpy(), oracle.contract(spec, ext, A, B, dtype=CODE[dtype]), dtype)


def _random_batched_cases(count, seed=20250311):
    rng = np.random.RandomState(seed)
    specs = ("zak,zkb->zab", "zak,zbk->zab", "zka,zkb->zab", "zka,zbk->zab", "azk,kzb->azb")
    out = []
    for i in range(count):
        dtype = "f32" if i % 3 == 2 else "f64"
        z = int(rng.randint(2, 13))
        m, n = int(rng.randint(40, 900)), int(rng.randint(40, 900))
        k = int(rng.choice([rng.randint(200, 700), rng.randint(1000, 3000), rng.randint(4000, 9000)]))
        if i % 2 == 0:  # vector layouts: multiples of four everywhere
            m, n, k = m // 4 * 4, n // 4 * 4, k // 16 * 16
        out.append((specs[i % len(specs)], dict(z=z, a=m, b=n, k=k), dtype))
    return out


_BATCHED_RANDOM = _random_batched_cases(int(os.environ.get("CTC_BATCH_FUZZ_CASES", "30")), int(os.environ.get("CTC_BATCH_FUZZ_SEED", "20250311"), 0))


@pytest.mark.parametrize("spec,ext,dtype", _BATCHED_RANDOM,
                         ids=["%s_%dx%dx%dx%d_%s" % (c[0].replace(",", "_").replace("->", "_"), c[1]["z"], c[1]["a"], c[1]["b"], c[1]["k"], c[2]) for c in _BATCHED_RANDOM])
def test_batched_random_shapes_against_single_products(cuda, spec, ext, dtype):
    """Seeded random batches (2 - 12 entries of 40 - 900 rows and columns under K from 200 to 9000, even and odd
    sizes): the batched plan - one list of tiles, cut along K where the scheduler says so - against one plan per
    entry on that entry's slice of the operands.  The two add the same products in different orders."""
    z, m, n, k = ext["z"], ext["a"], ext["b"], ext["k"]
    ins, outm = spec.split("->")
    la, lb = ins.split(",")
    A = torch.empty(z * m * k, dtype=TDT[dtype], device=cuda)
    B = torch.empty(z * k * n, dtype=TDT[dtype], device=cuda)
    ctc_amd.fill(A, 0xC7C0A)
    ctc_amd.fill(B, 0xC7C0B)
    C0 = torch.empty(z * m * n, dtype=TDT[dtype], device=cuda)
    ctc_amd.fill(C0, 0xC7C0C)
    got = C0.clone()
    plan = ctc_amd.Plan(spec, ext, dtype)
    plan.execute(A, B, got, alpha=1.5, beta=0.25)
    # entry by entry through strided single products
    shape = lambda letters: [ext[c] for c in letters]  # noqa: E731
    At, Bt, Wt = A.view(shape(la)), B.view(shape(lb)), C0.clone().view(shape(outm))
    single = la.replace("z", "") + "," + lb.replace("z", "") + "->" + outm.replace("z", "")
    sa = lambda t, letters: [t.stride(i) for i, c in enumerate(letters) if c != "z"]  # noqa: E731
    one = ctc_amd.Plan(single, dict(a=m, b=n, k=k), dtype, sa(At, la), sa(Bt, lb), sa(Wt, outm))
    for e in range(z):
        ea, eb, ec = At.select(la.index("z"), e), Bt.select(lb.index("z"), e), Wt.select(outm.index("z"), e)
        one.execute(A[ea.storage_offset():], B[eb.storage_offset():], Wt.view(-1)[ec.storage_offset():], alpha=1.5, beta=0.25)
    torch.cuda.synchronize()
    want = Wt.reshape(-1)
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 4 * TOL[dtype] * scale, (plan.summary(), plan.schedule())
