"""The persistent deferred-store kernel (short K under a large C) against the CPU oracle and against
the one-tile-per-workgroup kernel, through the C ABI.  Both kernels add the products of a K-tile
inside the MFMA in the same order, but their K-tiles differ (32 vs 16 contracted elements), so the
two agree to the parity tolerance (1e-12 f64 / 1e-5 f32 of max |C|), not bit for bit."""
import numpy as np
import pytest
import torch

import ctc_amd

pytestmark = pytest.mark.gpu

TOL = {"f64": 1e-12, "f32": 1e-5}
TDT = {"f64": torch.float64, "f32": torch.float32}
CODE = {"f64": 0, "f32": 1}


@pytest.fixture()
def forced_stream():
    ctc_amd.set_option("stream_mode", 2)
    yield
    ctc_amd.set_option("stream_mode", 0)


def _operands(oracle, spec, ext, dtype, dev):
    ins, outm = spec.split("->")
    la, lb = ins.split(",")
    A = oracle.fill(int(np.prod([ext[c] for c in la])), 0xC7C0A, CODE[dtype])
    B = oracle.fill(int(np.prod([ext[c] for c in lb])), 0xC7C0B, CODE[dtype])
    return A, B, int(np.prod([ext[c] for c in outm])), torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)


def _check(got, want, dtype):
    got, want = got.astype(np.float64), want.astype(np.float64)
    err = np.abs(got - want).max()
    assert err <= TOL[dtype] * np.abs(want).max(), "max err %.3e vs scale %.3e" % (err, np.abs(want).max())


CASES = [
    # every operand layout pair (K-major / outer-major), ragged edges in M, N and K, one to many items per workgroup
    ("nn_kmajor_outer", "ak,kb->ab", dict(a=1300, b=1100, k=64), "f64"),
    ("nt_kmajor_kmajor", "ak,bk->ab", dict(a=700, b=900, k=96), "f64"),
    ("tn_outer_outer", "ka,kb->ab", dict(a=2048, b=1536, k=40), "f64"),
    ("tn_outer_outer_k64", "ka,kb->ab", dict(a=2050, b=1534, k=64), "f64"),
    ("tn_outer_outer_k80_five_k_tiles_of_16", "ka,kb->ab", dict(a=2050, b=1534, k=80), "f64"),
    ("tn_outer_outer_k112_many_items", "ka,kb->ab", dict(a=4100, b=3002, k=112), "f64"),
    ("tt_outer_kmajor", "ka,bk->ab", dict(a=520, b=260, k=130), "f64"),
    ("tt_outer_kmajor_k128", "ka,bk->ab", dict(a=520, b=262, k=128), "f64"),
    ("many_items_kmajor_kmajor_k96", "ak,bk->ab", dict(a=4100, b=3000, k=96), "f64"),
    ("k_tail_and_three_tiles", "ak,kb->ab", dict(a=384, b=128, k=70), "f64"),
    ("many_items_per_workgroup_two_k_tiles", "ak,kb->ab", dict(a=300, b=45000, k=64), "f64"),
    ("many_items_per_workgroup_three_k_tiles", "ka,bk->ab", dict(a=4100, b=3000, k=90), "f64"),
    ("many_items_f32", "ak,kb->ab", dict(a=2000, b=20000, k=64), "f32"),
    ("k_tail_kmajor_both", "ak,bk->ab", dict(a=260, b=130, k=50), "f64"),
    ("two_contracted_modes_that_fuse", "aef,efb->ab", dict(a=256, b=192, e=6, f=10), "f64"),
    ("two_contracted_modes_that_do_not_fuse", "aef,feb->ab", dict(a=256, b=192, e=6, f=10), "f64"),
    ("one_tile_only", "ak,kb->ab", dict(a=100, b=60, k=34), "f64"),
    ("many_k_tiles", "ak,kb->ab", dict(a=640, b=704, k=1000), "f64"),
    ("config4_shape_small", "adij,bcdk->abcijk", dict(a=3, b=6, c=8, i=8, j=10, k=12, d=64), "f64"),
    ("rank6_swapped_roles", "bcdk,adij->abcijk", dict(a=4, b=5, c=6, i=6, j=8, k=10, d=48), "f64"),
    ("f32_nn", "ak,kb->ab", dict(a=1500, b=1028, k=64), "f32"),
    ("f32_tn_k_tail", "ka,kb->ab", dict(a=1024, b=1152, k=100), "f32"),
    ("f32_config4_shape_small", "adij,bcdk->abcijk", dict(a=3, b=6, c=8, i=8, j=12, k=12, d=64), "f32"),
]


@pytest.mark.parametrize("dma,alpha", [(1, 1.0), (1, 1.25), (0, 1.25)], ids=["lds_dma", "lds_dma_scaled", "register_staged"])
@pytest.mark.parametrize("name,spec,ext,dtype", CASES, ids=[c[0] for c in CASES])
def test_persistent_kernel_against_oracle(cuda, oracle, forced_stream, name, spec, ext, dtype, dma, alpha):
    """fp64 has two forms of the persistent kernel (operands staged by LDS-DMA, with and without the
    multiplication by alpha, or through registers as in fp32); fp32 has the last only."""
    if dtype == "f32" and dma:
        pytest.skip("fp32 has no LDS-DMA form")
    ctc_amd.set_option("stream_f64_dma", dma)
    try:
        # the LDS-DMA form takes K a multiple of its K-tile only: other cases fall back to the
        # register-staged form, which the test then simply runs twice
        _run_case(cuda, oracle, spec, ext, dtype, alpha=alpha)
    finally:
        ctc_amd.set_option("stream_f64_dma", 1)


def test_which_kernel_runs(cuda, oracle):
    """Routing: the counters say which tiled-GEMM kernel a launch took."""
    def delta(fn):
        names = ("launches_classic", "launches_dma_gemm", "launches_stream", "launches_stream_f64_dma")
        before = [ctc_amd.get_option(n) for n in names]
        fn()
        d = [ctc_amd.get_option(n) - b for n, b in zip(names, before)]
        return (d[0] + d[1], d[2], d[3])  # one-tile-per-workgroup (either staging), persistent, persistent LDS-DMA
    x = torch.zeros(4096 * 96, dtype=torch.float64, device=cuda)
    c = torch.zeros(4096 * 4096, dtype=torch.float64, device=cuda)
    run = lambda spec, ext, **kw: lambda: ctc_amd.Plan(spec, ext, "f64").execute(x, x, c, **kw)  # noqa: E731
    ctc_amd.set_option("stream_mode", 2)
    try:
        assert delta(run("ak,kb->ab", dict(a=4096, b=4096, k=64))) == (0, 0, 1)
        assert delta(run("ak,kb->ab", dict(a=4096, b=4096, k=64), alpha=2.0)) == (0, 0, 1)      # alpha != 1: its own instantiation
        assert delta(run("ak,kb->ab", dict(a=4096, b=4096, k=80))) == (0, 1, 0)                 # K not a multiple of 32
        assert delta(run("ka,kb->ab", dict(a=4096, b=4096, k=80))) == (0, 0, 1)                 # outer-major pair: K-tiles of 16
        assert delta(run("ka,kb->ab", dict(a=4096, b=4096, k=72))) == (0, 1, 0)
        assert delta(run("aef,feb->ab", dict(a=4096, b=4096, e=8, f=8))) == (0, 1, 0)           # k offsets not one stride
        assert delta(run("ak,kb->ab", dict(a=4096, b=4096, k=64), beta=1.0)) == (1, 0, 0)       # beta != 0
        assert delta(run("ak,kb->ab", dict(a=4096, b=4096, k=32))) == (1, 0, 0)                 # one K-tile only
        ctc_amd.set_option("stream_mode", 0)
        assert delta(run("ak,kb->ab", dict(a=4096, b=4096, k=64))) == (1, 0, 0)                 # C too small to bother
        ctc_amd.set_option("stream_mode", 1)
        assert delta(run("ak,kb->ab", dict(a=4096, b=4096, k=64))) == (1, 0, 0)
    finally:
        ctc_amd.set_option("stream_mode", 0)


def test_automatic_routing_of_short_k(cuda):
    """Automatic mode, a C of 4096 tiles of 128 x 128: fp64 takes the persistent kernel only in its LDS-DMA
    form; what that form cannot run goes to the one-tile-per-workgroup kernels, never to the (slower)
    register-staged persistent form.  fp32 has the register-staged form only and takes it."""
    names = ("launches_classic", "launches_dma_gemm", "launches_stream", "launches_stream_f64_dma")

    def delta(fn):
        before = [ctc_amd.get_option(n) for n in names]
        fn()
        d = [ctc_amd.get_option(n) - b for n, b in zip(names, before)]
        return (d[0] + d[1], d[2], d[3])
    x = torch.zeros(8192 * 256, dtype=torch.float64, device=cuda)  # the largest operand below: 8192 x 256
    c = torch.zeros(8192 * 8192, dtype=torch.float64, device=cuda)
    views = {"f64": (x, c), "f32": (x.view(torch.float32), c.view(torch.float32))}  # zeros in either element type

    def run(spec, ext, dt="f64", **kw):
        return lambda: ctc_amd.Plan(spec, ext, dt).execute(views[dt][0], views[dt][0], views[dt][1], **kw)
    assert ctc_amd.get_option("stream_mode") == 0
    assert delta(run("ak,kb->ab", dict(a=8192, b=8192, k=64))) == (0, 0, 1)
    assert delta(run("ak,kb->ab", dict(a=8192, b=8192, k=64), alpha=1.25)) == (0, 0, 1)
    assert delta(run("ak,kb->ab", dict(a=8192, b=8192, k=40))) == (1, 0, 0)                  # K = 40: no whole K-tiles
    assert delta(run("ka,kb->ab", dict(a=8192, b=8192, k=40))) == (1, 0, 0)
    assert delta(run("ka,kb->ab", dict(a=8192, b=8192, k=80))) == (0, 0, 1)                  # K-tiles of 16
    assert delta(run("ak,kb->ab", dict(a=8192, b=8192, k=80))) == (1, 0, 0)                  # K-tiles of 32 only
    assert delta(run("aef,feb->ab", dict(a=8192, b=8192, e=8, f=8))) == (1, 0, 0)             # k offsets not one stride
    assert delta(run("ak,kb->ab", dict(a=8192, b=8192, k=256))) == (1, 0, 0)                 # K above the cut
    assert delta(run("ak,kb->ab", dict(a=8192, b=8192, k=64), "f32")) == (0, 1, 0)


def test_streaming_store_hint_needs_whole_sectors(cuda):
    """abcijk = adij,bcdk with free extents 42 (runs of C of 336 bytes: they end inside 64-byte sectors): stores with
    the streaming hint run at half the rate there, so the automatic choice leaves the persistent fp64 kernel (which
    always streams) for the one-tile-per-workgroup kernel; with extents 40 (320-byte runs) it takes the persistent one."""
    names = ("launches_classic", "launches_dma_gemm", "launches_stream", "launches_stream_f64_dma")
    for n, want in ((42, (1, 0, 0)), (40, (0, 0, 1))):
        ext = dict(a=1, b=n, c=n, i=n, j=n, k=n, d=64)
        plan = ctc_amd.Plan("adij,bcdk->abcijk", ext, "f64")
        A = torch.zeros(64 * n * n, dtype=torch.float64, device=cuda)
        B = torch.zeros(n * n * 64 * n, dtype=torch.float64, device=cuda)
        C = torch.empty(n ** 5, dtype=torch.float64, device=cuda)
        before = [ctc_amd.get_option(x) for x in names]
        plan.execute(A, B, C)
        torch.cuda.synchronize()
        d = [ctc_amd.get_option(x) - b for x, b in zip(names, before)]
        assert (d[0] + d[1], d[2], d[3]) == want, (n, d)
        assert float(C.abs().max()) == 0.0
        del A, B, C
        torch.cuda.empty_cache()


def _run_case(cuda, oracle, spec, ext, dtype, alpha=1.25):
    A, B, n_out, dA, dB = _operands(oracle, spec, ext, dtype, cuda)
    dC = torch.full((n_out,), float("nan"), dtype=TDT[dtype], device=cuda)
    plan = ctc_amd.Plan(spec, ext, dtype)
    s = plan.summary()
    assert s["kernel"] == "mfma_gemm" and s["layout_a"] != "scalar" and s["layout_b"] != "scalar" and s["vec_c"] > 1, s
    plan.execute(dA, dB, dC, alpha=alpha)
    want = oracle.contract(spec, ext, A, B, alpha=alpha, dtype=CODE[dtype])
    got = dC.cpu().numpy()
    assert not np.isnan(got).any()  # every element of C was stored
    _check(got, want, dtype)
    # the one-tile-per-workgroup kernel on the same plan
    ctc_amd.set_option("stream_mode", 1)
    dC2 = torch.full((n_out,), float("nan"), dtype=TDT[dtype], device=cuda)
    plan.execute(dA, dB, dC2, alpha=alpha)
    _check(dC2.cpu().numpy(), want, dtype)


def test_config4_slab_at_full_size_properties(cuda, oracle):
    """abcijk = adij,bcdk with free extents 40, d = 64, slab a = 5 of 40 (one GPU's share of eight;
    4.1 GB of C): far too big for the oracle, so (i) the automatic choice must take the persistent
    kernel and agree with the classic one to 1e-12, (ii) rows of C are recomputed by the oracle."""
    n, d, a = 40, 64, 5
    spec, ext = "adij,bcdk->abcijk", dict(a=a, b=n, c=n, i=n, j=n, k=n, d=d)
    dA = torch.empty(a * d * n * n, dtype=torch.float64, device=cuda)
    dB = torch.empty(n * n * d * n, dtype=torch.float64, device=cuda)
    ctc_amd.fill(dA, 0xC7C0A)
    ctc_amd.fill(dB, 0xC7C0B)
    nc = a * n ** 5
    dC = torch.full((nc,), float("nan"), dtype=torch.float64, device=cuda)
    dC2 = torch.full((nc,), float("nan"), dtype=torch.float64, device=cuda)
    plan = ctc_amd.Plan(spec, ext, "f64")
    ctc_amd.set_option("stream_mode", 0)
    plan.execute(dA, dB, dC)
    ctc_amd.set_option("stream_mode", 1)
    plan.execute(dA, dB, dC2)
    ctc_amd.set_option("stream_mode", 0)
    torch.cuda.synchronize()
    scale = float(dC2.abs().max())
    assert float((dC - dC2).abs().max()) <= 1e-12 * scale
    assert not bool(torch.isnan(dC).any())
    # C[a0, b0, c0, :, :, :] = sum_d A[a0, d, :, :] (x) B[b0, c0, d, :]
    A = oracle.fill(a * d * n * n, 0xC7C0A).reshape(a, d, n, n)
    B = oracle.fill(n * n * d * n, 0xC7C0B).reshape(n, n, d, n)
    for a0, b0, c0 in ((0, 0, 0), (4, 39, 17), (2, 20, 39)):
        want = oracle.contract("adij,bcdk->abcijk", dict(a=1, b=1, c=1, i=n, j=n, k=n, d=d),
                               np.ascontiguousarray(A[a0]).reshape(-1), np.ascontiguousarray(B[b0, c0]).reshape(-1))
        got = dC.reshape(a, n, n, n ** 3)[a0, b0, c0].cpu().numpy()
        _check(got, want, "f64")
