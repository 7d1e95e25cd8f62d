"""Contractions whose operands would be read element by element (an odd extent on the unit-stride mode) under a
large problem: ctc_tc_execute copies A and B into padded K-major rows, multiplies those with the LDS-DMA kernel and
writes C in place.  Checked against the CPU oracle (1e-12 fp64, 1e-5 fp32, 8e-3 bf16 of max |C|), against the fused
kernel on the same plan (the "pad_transposes" option switched off), with alpha and beta, batch modes, a slab plan,
and twice in a row (stale workspace contents must not reach the second result)."""
import numpy as np
import pytest
import torch

import ctc_amd

pytestmark = pytest.mark.gpu

TOL = {"f64": 1e-12, "f32": 1e-5, "bf16": 8e-3}
CODE = {"f64": 0, "f32": 1, "bf16": 2}


def _dev(x, dtype, cuda):
    if dtype == "bf16":
        return torch.from_numpy(x.view(np.int16)).to(cuda).view(torch.bfloat16)
    return torch.from_numpy(x).to(cuda)


def _host(t, dtype, oracle):
    if dtype == "bf16":
        return oracle.bf16_to_f32(t.view(torch.int16).cpu().numpy().view(np.uint16)).astype(np.float64)
    return t.cpu().numpy().astype(np.float64)


def _want(oracle, spec, ext, A, B, C0, alpha, beta, dtype):
    w = oracle.contract(spec, ext, A, B, C=None if C0 is None else C0.copy(), alpha=alpha, beta=beta, dtype=CODE[dtype])
    return (oracle.bf16_to_f32(w) if dtype == "bf16" else w).astype(np.float64)


CASES = [
    # spec, extents, dtype: M, N >= 2048 (1024 below fp64) and K >= 1024 (512), an odd extent on a unit-stride mode
    ("aeim,ebmj->abij", dict(a=47, b=45, i=47, j=49, e=33, m=35), "f64"),
    ("aeim,ebmj->abij", dict(a=41, b=43, i=41, j=41, e=23, m=45), "f32"),
    ("aeim,ebmj->abij", dict(a=41, b=43, i=41, j=41, e=23, m=45), "bf16"),
    ("ak,kb->ab", dict(a=2049, b=2051, k=1025), "f64"),          # A K-major with odd rows, B outer-major with odd rows
    ("ka,bk->ab", dict(a=1601, b=1603, k=1537), "f32"),
    ("zaek,zkbe->zab", dict(z=3, a=1601, b=1603, e=23, k=25), "f32"),   # a batch mode, two contracted modes in different orders
    ("mia,jmb->abij", dict(a=41, b=43, i=41, j=41, m=1201), "f32"),     # C's unit-stride mode is free in A's partner: roles
]


@pytest.mark.parametrize("spec,ext,dtype", CASES, ids=["%s_%s" % (c[0].replace(",", "_").replace("->", "_"), c[2]) for c in CASES])
def test_padded_copies_against_oracle_and_fused_kernel(cuda, oracle, spec, ext, dtype):
    ins, outm = spec.split("->")
    la, lb = ins.split(",")
    num = lambda t: int(np.prod([ext[c] for c in t]))  # noqa: E731
    A, B = oracle.fill(num(la), 0xC7C0A, CODE[dtype]), oracle.fill(num(lb), 0xC7C0B, CODE[dtype])
    C0 = oracle.fill(num(outm), 0xC7C0C, CODE[dtype])
    plan = ctc_amd.Plan(spec, ext, dtype)
    s = plan.summary()
    assert s["transposes"] == "padded_copies" and "scalar" in (s["layout_a"], s["layout_b"]), s
    dA, dB = _dev(A, dtype, cuda), _dev(B, dtype, cuda)
    # beta = 0 over a poisoned C, twice (the second run finds the first one's copies and partial sums in the workspace)
    want = _want(oracle, spec, ext, A, B, None, 1.0, 0.0, dtype)
    for _ in range(2):
        dC = _dev(C0.copy(), dtype, cuda)
        if dtype != "bf16":
            dC.fill_(float("nan"))
        before = ctc_amd.get_option("padded_executions")
        plan.execute(dA, dB, dC)
        assert ctc_amd.get_option("padded_executions") == before + 1
        got = _host(dC, dtype, oracle)
        assert np.isfinite(got).all()
        assert np.abs(got - want).max() <= TOL[dtype] * np.abs(want).max()
    # alpha and beta: C is read and written in place by the GEMM
    dC = _dev(C0.copy(), dtype, cuda)
    plan.execute(dA, dB, dC, alpha=-0.5, beta=1.25)
    want2 = _want(oracle, spec, ext, A, B, C0, -0.5, 1.25, dtype)
    got2 = _host(dC, dtype, oracle)
    assert np.abs(got2 - want2).max() <= 2 * TOL[dtype] * np.abs(want2).max()
    # the fused kernel on the same plan (element loads) agrees to the same tolerance
    ctc_amd.set_option("pad_transposes", 0)
    try:
        dF = _dev(C0.copy(), dtype, cuda)
        before = ctc_amd.get_option("padded_executions")
        plan.execute(dA, dB, dF)
        assert ctc_amd.get_option("padded_executions") == before
        fused = _host(dF, dtype, oracle)
        assert np.abs(fused - want).max() <= TOL[dtype] * np.abs(want).max()
        # and back: the fused launch used the plan's workspace for its partial sums; the padded copies must be
        # cleared again before they are trusted
        ctc_amd.set_option("pad_transposes", 1)
        dC = _dev(C0.copy(), dtype, cuda)
        plan.execute(dA, dB, dC)
        got3 = _host(dC, dtype, oracle)
        assert np.abs(got3 - want).max() <= TOL[dtype] * np.abs(want).max()
    finally:
        ctc_amd.set_option("pad_transposes", 1)


def test_prepared_plan_allocates_nothing_and_slab_plans_take_the_path(cuda, oracle):
    n = 47
    ext = dict(a=2 * n, b=n, i=n, j=n, e=33, m=35)
    spec = "aeim,ebmj->abij"
    full = ctc_amd.Plan(spec, ext, "f64")
    A = oracle.fill(2 * n * 33 * n * 35, 1, 0)
    B = oracle.fill(33 * n * 35 * n, 2, 0)
    dA, dB = torch.from_numpy(A).to(cuda), torch.from_numpy(B).to(cuda)
    dC = torch.full((2 * n * n * n * n,), float("nan"), dtype=torch.float64, device=cuda)
    for rank in range(2):
        slab, oa, ob, oc = full.shard("a", rank, 2)
        assert slab.summary()["transposes"] == "padded_copies"
        slab.prepare()
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info()[0]
        slab.execute(dA[oa:], dB[ob:], dC[oc:])
        slab.execute(dA[oa:], dB[ob:], dC[oc:])
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free_before  # prepare() took everything the executions need
    want = oracle.contract(spec, ext, A, B, dtype=0)
    got = dC.cpu().numpy()
    assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()


def test_small_or_even_contractions_keep_the_fused_kernel():
    for spec, ext, dt in (("aeim,ebmj->abij", dict(a=48, b=48, i=48, j=48, e=32, m=36), "f64"),   # even extents: vector loads
                          ("aeim,ebmj->abij", dict(a=21, b=21, i=21, j=21, e=21, m=21), "f64"),   # odd but small
                          ("ak,kb->ab", dict(a=15, b=100001, k=1025), "f32")):                     # skinny
        assert ctc_amd.Plan(spec, ext, dt).summary()["transposes"] == "fused", (spec, ext)
    # explicit strides (not packed): the copies assume packed operands
    p = ctc_amd.Plan("ak,kb->ab", dict(a=2049, b=2051, k=1025), "f64", strides_a=[1027, 1])
    assert p.summary()["transposes"] == "fused"


def test_seeded_random_mode_orders_through_the_padded_copies(cuda, oracle):
    """Random mode orders of A, B and C (so: random transposes into the padded rows, role swaps, C layouts), odd
    extents, all three element types, alpha and beta.  Cases the planner leaves on the fused kernel (an operand
    order that happens to allow vectors) are checked all the same; most must take the padded copies."""
    import random
    rng = random.Random(0x0DD)
    took = 0
    n_cases = 14
    for n in range(n_cases):
        dtype = rng.choice(["f64", "f32", "f32", "bf16"])
        ext = dict(a=rng.choice([47, 49, 51]), i=rng.choice([47, 49, 51]), b=rng.choice([47, 49, 51]), j=rng.choice([47, 49, 51]),
                   e=rng.choice([21, 23, 25]), m=rng.choice([47, 49, 51]))
        la, lb, lc = list("aeim"), list("ebmj"), list("abij")
        for m in (la, lb, lc):
            rng.shuffle(m)
        spec = "%s,%s->%s" % ("".join(la), "".join(lb), "".join(lc))
        alpha, beta = rng.choice([1.0, -0.5, 2.0]), rng.choice([0.0, 0.0, 1.0, -1.5])
        num = lambda t: int(np.prod([ext[c] for c in t]))  # noqa: E731
        A, B = oracle.fill(num(la), 100 + n, CODE[dtype]), oracle.fill(num(lb), 200 + n, CODE[dtype])
        C0 = oracle.fill(num(lc), 300 + n, CODE[dtype])
        plan = ctc_amd.Plan(spec, ext, dtype)
        before = ctc_amd.get_option("padded_executions")
        dC = _dev(C0.copy(), dtype, cuda)
        plan.execute(_dev(A, dtype, cuda), _dev(B, dtype, cuda), dC, alpha=alpha, beta=beta)
        took += ctc_amd.get_option("padded_executions") - before
        want = _want(oracle, spec, ext, A, B, C0, alpha, beta, dtype)
        got = _host(dC, dtype, oracle)
        assert np.abs(got - want).max() <= 2 * TOL[dtype] * np.abs(want).max(), (n, spec, ext, dtype, alpha, beta, plan.summary()["transposes"])
    assert took >= n_cases // 2, took
