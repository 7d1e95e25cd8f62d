"""ctc_tc_execute_noalloc (the entry point that allocates nothing: split-K partial sums go to the
library's fixed per-GPU scratch buffer) against ctc_tc_execute, through the C ABI.

Both entry points launch the same kernels with the same schedule, so the results must agree bit
for bit; ctc_tc_execute itself is checked against the CPU oracle here as well (1e-12 / 1e-5 of
max |C|).  Two calls in a row, the second on a smaller contraction, with the result buffers
poisoned in between: stale partial sums of the first call must not reach the second."""
import numpy as np
import pytest
import torch

import ctc_amd

pytestmark = pytest.mark.gpu

SPEC = "aeim,ebmj->abij"
TOL = {"f64": 1e-12, "f32": 1e-5}
TDT = {"f64": torch.float64, "f32": torch.float32}
CODE = {"f64": 0, "f32": 1}


def _run_pair(cuda, oracle, n, dtype, check_oracle, k=None):
    k = k or n
    ext = dict(a=n, b=n, i=n, j=n, e=k, m=k)
    A = oracle.fill(n * n * k * k, 0xC7C0A, CODE[dtype])
    B = oracle.fill(n * n * k * k, 0xC7C0B, CODE[dtype])
    dA, dB = torch.from_numpy(A).to(cuda), torch.from_numpy(B).to(cuda)
    plan = ctc_amd.Plan(SPEC, ext, dtype)
    plan.prepare()
    sched = plan.schedule()
    # the case must really use scratch: tiles of the last round are cut along K
    assert sched["workspace_bytes"] > 0, sched
    assert sched["workspace_bytes"] <= ctc_amd.noalloc_scratch_bytes()
    want = torch.full((n ** 4,), float("nan"), dtype=TDT[dtype], device=cuda)
    got = torch.full((n ** 4,), float("nan"), dtype=TDT[dtype], device=cuda)
    plan.execute(dA, dB, want)
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    plan.execute_noalloc(dA, dB, got)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before  # no device memory was taken
    assert torch.equal(got, want)
    # beta != 0 reads C back: same comparison with an accumulate
    plan.execute(dA, dB, want, alpha=0.5, beta=-1.25)
    plan.execute_noalloc(dA, dB, got, alpha=0.5, beta=-1.25)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    if check_oracle:
        ref = oracle.contract(SPEC, ext, A, B, dtype=CODE[dtype]).astype(np.float64)
        ref = 0.5 * ref - 1.25 * ref
        err = np.abs(got.cpu().numpy().astype(np.float64) - ref).max()
        assert err <= 2 * TOL[dtype] * np.abs(ref).max()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_noalloc_matches_execute_twice_second_smaller(cuda, oracle, dtype):
    _run_pair(cuda, oracle, 48, dtype, check_oracle=False)  # 2304^3: large tile, last round cut along K
    _run_pair(cuda, oracle, 16, dtype, check_oracle=True, k=48)   # 256^2 x 2304: a few small tiles, all cut along K


def test_noalloc_needs_prepare_and_rejects_oversized_schedules(cuda, oracle):
    n = 16
    ext = dict(a=n, b=n, i=n, j=n, e=n, m=n)
    x = torch.zeros(n ** 4, dtype=torch.float64, device=cuda)
    plan = ctc_amd.Plan(SPEC, ext, "f64")
    with pytest.raises(ctc_amd.CtcError, match="ctc_tc_plan_prepare"):
        plan.execute_noalloc(x, x, x.clone())
    # 130 tiles of 128 x 128 under a long, odd K (element loads of A: the register-staged kernel, whose tiles
    # are all cut into equal K chunks, several per CU): more partial sums than the fixed buffer holds
    # must be refused, not truncated
    big = ctc_amd.Plan("ak,kb->ab", dict(a=1280, b=1664, k=16385), "f64")
    big.prepare()
    sch = big.schedule()
    assert sch["dma_form"] == 0 and sch["workspace_bytes"] > ctc_amd.noalloc_scratch_bytes(), sch
    a = torch.zeros(2560 * 16385, dtype=torch.float64, device=cuda)  # room for both shapes of this test
    c = torch.zeros(2560 * 1920, dtype=torch.float64, device=cuda)
    with pytest.raises(ctc_amd.CtcError, match="scratch"):
        big.execute_noalloc(a, a, c)
    big.execute(a, a, c)  # the allocating entry point still runs it
    torch.cuda.synchronize()
    # the same shape with an even K takes the LDS-DMA kernel, whose schedules never need more than the buffer holds
    fits = ctc_amd.Plan("ak,kb->ab", dict(a=2560, b=1920, k=16384), "f64")
    fits.prepare()
    sch = fits.schedule()
    assert sch["dma_form"] == 1 and 0 < sch["workspace_bytes_worst"] and sch["workspace_bytes"] <= ctc_amd.noalloc_scratch_bytes(), sch
    fits.execute_noalloc(a, a, c)
    torch.cuda.synchronize()


def test_noalloc_bf16_split_k(cuda, oracle):
    """bf16: fewer tiles than workgroup slots under a long K, so every tile is cut along K and the f32 partial
    sums go through the fixed scratch buffer; bit-equal to ctc_tc_execute, within the bf16 tolerance of the
    oracle (8e-3 of max |C|: one bf16 rounding of the result plus f32 accumulation)."""
    spec, ext = "ak,kb->ab", dict(a=512, b=384, k=8192)
    A = oracle.fill(512 * 8192, 0xC7C0A, 2)
    B = oracle.fill(8192 * 384, 0xC7C0B, 2)
    to_dev = lambda u16: torch.from_numpy(u16.view(np.int16)).to(cuda).view(torch.bfloat16)  # noqa: E731
    dA, dB = to_dev(A), to_dev(B)
    plan = ctc_amd.Plan(spec, ext, "bf16")
    plan.prepare()
    sch = plan.schedule()
    assert sch["split"] >= 2 and 0 < sch["workspace_bytes"] <= ctc_amd.noalloc_scratch_bytes(), sch
    want = torch.zeros(512 * 384, dtype=torch.bfloat16, device=cuda)
    got = torch.ones(512 * 384, dtype=torch.bfloat16, device=cuda)
    plan.execute(dA, dB, want)
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    plan.execute_noalloc(dA, dB, got)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    ref = oracle.bf16_to_f32(oracle.contract(spec, ext, A, B, dtype=2)).astype(np.float64)
    res = oracle.bf16_to_f32(got.view(torch.int16).cpu().numpy().view(np.uint16)).astype(np.float64)
    assert np.abs(res - ref).max() <= 8e-3 * np.abs(ref).max()
    # second call, smaller contraction, stale partial sums of the first in the buffer
    spec2, ext2 = "ak,bk->ab", dict(a=250, b=200, k=3064)
    A2, B2 = oracle.fill(250 * 3064, 0xC7C0A, 2), oracle.fill(200 * 3064, 0xC7C0B, 2)
    plan2 = ctc_amd.Plan(spec2, ext2, "bf16")
    plan2.prepare()
    want2 = torch.zeros(250 * 200, dtype=torch.bfloat16, device=cuda)
    got2 = torch.ones(250 * 200, dtype=torch.bfloat16, device=cuda)
    plan2.execute(to_dev(A2), to_dev(B2), want2)
    plan2.execute_noalloc(to_dev(A2), to_dev(B2), got2)
    torch.cuda.synchronize()
    assert torch.equal(got2.view(torch.int16), want2.view(torch.int16))
    ref2 = oracle.bf16_to_f32(oracle.contract(spec2, ext2, A2, B2, dtype=2)).astype(np.float64)
    res2 = oracle.bf16_to_f32(got2.view(torch.int16).cpu().numpy().view(np.uint16)).astype(np.float64)
    assert np.abs(res2 - ref2).max() <= 8e-3 * np.abs(ref2).max()
