"""Planner (host side of libctc_tc.so) against the oracle's classification and, for the
execution view, against a numpy emulation of the kernel's table addressing.  No GPU."""
import itertools

import numpy as np
import pytest

import ctc_amd
from ctc_amd import _lib

SPECS = [
    ("adc,db->abc", dict(a=8, b=6, c=5, d=7)),
    ("aeim,ebmj->abij", dict(a=6, b=4, i=8, j=6, e=3, m=10)),
    ("dkca,bk->abcd", dict(a=16, b=8, c=4, d=12, k=8)),
    ("adij,bcdk->abcijk", dict(a=4, b=5, c=3, i=6, j=4, k=5, d=8)),
    ("zak,zkb->zab", dict(z=3, a=6, b=4, k=8)),
    ("akz,kbz->abz", dict(z=4, a=10, b=12, k=6)),
    ("a,b->ab", dict(a=17, b=9)),
    ("ak,k->a", dict(a=33, k=21)),
    ("ka,kb->ba", dict(a=20, b=12, k=18)),
    ("ab,ab->ab", dict(a=5, b=6)),
]


@pytest.mark.parametrize("spec,ext", SPECS)
def test_classification_matches_oracle_bit_exact(oracle, spec, ext):
    plan = ctc_amd.Plan(spec, ext)
    s, o = plan.summary(), oracle.classify(spec)
    assert s["names"] == o["names"]
    assert s["class_codes"] == o["class_codes"]
    for key in ("batch", "m", "n", "k", "perm_a", "perm_b", "perm_c"):
        assert s[key] == o[key], key


def test_classification_matches_golden(golden):
    for c in golden["cases"]:
        s = ctc_amd.Plan(c["spec"], c["extents"], c["dtype"]).summary()
        w = c["classification"]
        assert s["names"] == w["names"]
        assert {n: ctc_amd.CLASS_NAMES[k] for n, k in zip(s["names"], s["class_codes"])} == w["classes"]
        for key in ("batch", "m", "n", "k", "perm_a", "perm_b", "perm_c"):
            assert s[key] == w[key], (c["name"], key)


def _emulate(plan, A, B, n_out):
    """What the kernel computes, restated with numpy gathers over the plan's tables."""
    s = plan.summary()
    t = {q: plan.table(getattr(_lib, q)) for q in
         ("Q_ROW_A", "Q_K_A", "Q_K_B", "Q_COL_B", "Q_ROW_C", "Q_COL_C", "Q_BAT_A", "Q_BAT_B", "Q_BAT_C")}
    GA, GB = (B, A) if s["swapped"] else (A, B)
    C = np.zeros(n_out)
    for b in range(s["batch_count"]):
        Am = GA[t["Q_BAT_A"][b] + t["Q_ROW_A"][:, None] + t["Q_K_A"][None, :]]
        Bm = GB[t["Q_BAT_B"][b] + t["Q_K_B"][:, None] + t["Q_COL_B"][None, :]]
        C[t["Q_BAT_C"][b] + t["Q_ROW_C"][:, None] + t["Q_COL_C"][None, :]] = Am @ Bm
    return C


@pytest.mark.parametrize("spec,ext", SPECS)
def test_offset_tables_reproduce_the_contraction(oracle, spec, ext):
    plan = ctc_amd.Plan(spec, ext)
    s = plan.summary()
    ins, outm = spec.split("->")
    la, lb = ins.split(",")
    A = oracle.fill(int(np.prod([ext[c] for c in la])), 3)
    B = oracle.fill(int(np.prod([ext[c] for c in lb])), 4)
    n_out = int(np.prod([ext[c] for c in outm]))
    want = oracle.contract(spec, ext, A, B)
    got = _emulate(plan, A, B, n_out)
    assert np.allclose(got, want, rtol=0, atol=1e-12)
    assert s["M"] * s["N"] * s["batch_count"] == n_out
    # every C element is addressed exactly once
    rc, cc, bc = (plan.table(q) for q in (_lib.Q_ROW_C, _lib.Q_COL_C, _lib.Q_BAT_C))
    addr = (bc[:, None, None] + rc[None, :, None] + cc[None, None, :]).reshape(-1)
    assert np.array_equal(np.sort(addr), np.arange(n_out))


def test_execution_view_headline_contraction():
    n = 64
    plan = ctc_amd.Plan("aeim,ebmj->abij", dict(a=n, b=n, i=n, j=n, e=n, m=n))
    s = plan.summary()
    assert (s["M"], s["N"], s["K"], s["batch_count"]) == (n * n, n * n, n * n, 1)
    assert not s["swapped"]
    assert s["exec_k"] == "me" and s["exec_m"] == "ia" and s["exec_n"] == "jb"
    assert s["layout_a"] == "k_vec" and s["layout_b"] == "outer_vec"
    assert s["vec_a"] == 2 and s["vec_b"] == 2
    assert s["tile"] == (128, 128, 16)
    assert plan.flops == 2 * n ** 6


def test_bf16_tile_choice(monkeypatch):
    """bf16: 128 x 128 x 64 by default, 256 x 256 x 64 (512 threads) once the problem has 192 such
    tiles and both operands can be read in 16-byte vectors; CTC_TC_FORCE_TILE overrides the count."""
    small = ctc_amd.Plan("ak,kb->ab", dict(a=2048, b=2048, k=512), "bf16").summary()
    assert tuple(small["tile"]) == (128, 128, 64)
    big = ctc_amd.Plan("ak,kb->ab", dict(a=4096, b=3072, k=512), "bf16").summary()
    assert tuple(big["tile"]) == (256, 256, 64)
    odd = ctc_amd.Plan("ak,kb->ab", dict(a=4099, b=3073, k=511), "bf16").summary()  # element loads
    assert (odd["layout_a"], odd["layout_b"]) == ("scalar", "scalar") and tuple(odd["tile"]) == (128, 128, 64)
    monkeypatch.setenv("CTC_TC_FORCE_TILE", "128")
    assert tuple(ctc_amd.Plan("ak,kb->ab", dict(a=4096, b=3072, k=512), "bf16").summary()["tile"]) == (128, 128, 64)
    monkeypatch.setenv("CTC_TC_FORCE_TILE", "256")
    assert tuple(ctc_amd.Plan("ak,kb->ab", dict(a=512, b=512, k=64), "bf16").summary()["tile"]) == (256, 256, 64)
    # f64 plans are not touched by the bf16 rule
    monkeypatch.delenv("CTC_TC_FORCE_TILE")
    assert tuple(ctc_amd.Plan("ak,kb->ab", dict(a=4096, b=3072, k=512), "f64").summary()["tile"]) == (128, 128, 16)


def test_swap_puts_c_unit_stride_mode_in_n():
    # C's fastest mode 'a' belongs to A: the planner exchanges the operand roles
    s = ctc_amd.Plan("ka,kb->ba", dict(a=20, b=12, k=18)).summary()
    assert s["swapped"] and s["exec_n"][0] == "a"
    s = ctc_amd.Plan("dkca,bk->abcd", dict(a=16, b=8, c=4, d=12, k=8), "f32").summary()
    assert s["swapped"] and s["exec_n"][0] == "d"


def test_vector_width_needs_divisible_extents_and_strides():
    assert ctc_amd.Plan("ak,kb->ab", dict(a=8, b=8, k=7)).summary()["layout_a"] == "scalar"
    assert ctc_amd.Plan("ak,kb->ab", dict(a=8, b=8, k=8)).summary()["layout_a"] == "k_vec"
    assert ctc_amd.Plan("ak,kb->ab", dict(a=8, b=7, k=8)).summary()["layout_b"] == "scalar"
    s = ctc_amd.Plan("ak,kb->ab", dict(a=8, b=8, k=8), "f32").summary()
    assert s["vec_a"] == 4 and s["vec_b"] == 4
    assert ctc_amd.Plan("ak,kb->ab", dict(a=8, b=8, k=6), "f32").summary()["layout_a"] == "scalar"
    # padded leading dimension that breaks 16-byte alignment of the rows
    s = ctc_amd.Plan("ak,kb->ab", dict(a=8, b=8, k=8), strides_a=[9, 1]).summary()
    assert s["layout_a"] == "scalar"
    s = ctc_amd.Plan("ak,kb->ab", dict(a=8, b=8, k=8), strides_a=[10, 1]).summary()
    assert s["layout_a"] == "k_vec"


def test_strided_operands(oracle):
    ext = dict(a=6, b=4, k=8)
    A = oracle.fill(6 * 10, 5)  # rows padded to 10
    B = oracle.fill(8 * 4, 6)
    plan = ctc_amd.Plan("ak,kb->ab", ext, strides_a=[10, 1])
    got = _emulate(plan, A, B, 24)
    want = oracle.contract("ak,kb->ab", ext, A, B, strides_a=[10, 1])
    assert np.allclose(got, want, atol=1e-13)


def test_shard_free_and_contracted(oracle):
    ext = dict(a=8, b=4, i=4, j=6, e=6, m=2)
    spec = "aeim,ebmj->abij"
    full = ctc_amd.Plan(spec, ext)
    A = oracle.fill(8 * 6 * 4 * 2, 1)
    B = oracle.fill(6 * 4 * 2 * 6, 2)
    want = oracle.contract(spec, ext, A, B)
    # free index: each slab of C computed independently
    C = np.zeros(want.size)
    for r in range(4):
        sh, oa, ob, oc = full.shard("a", r, 4)
        assert ob == 0 and oa == r * 2 * 6 * 4 * 2 and oc == r * 2 * 4 * 4 * 6
        part = _emulate(sh, A[oa:], B[ob:], want.size - oc)
        n_slab = 2 * 4 * 4 * 6
        C[oc:oc + n_slab] = part[:n_slab]
    assert np.allclose(C, want, atol=1e-13)
    # contracted index: partial sums add up
    C = np.zeros(want.size)
    for r in range(3):
        sh, oa, ob, oc = full.shard("e", r, 3)
        assert oc == 0
        C += _emulate(sh, A[oa:], B[ob:], want.size)
    assert np.allclose(C, want, atol=1e-13)
    with pytest.raises(ctc_amd.CtcError):
        full.shard("a", 0, 3)
    with pytest.raises(ctc_amd.CtcError):
        full.shard("q", 0, 2)


def test_errors():
    for bad in ("ab,bc", "ab,bc->ad", "aa,ab->b", "ab,cd->abc", "a1,1b->ab"):
        with pytest.raises(ctc_amd.CtcError):
            ctc_amd.Plan(bad, dict(a=2, b=2, c=2, d=2))
    with pytest.raises(ctc_amd.CtcError):
        ctc_amd.Plan("ab,bc->ac", dict(a=2, b=2))       # missing extent
    with pytest.raises(ctc_amd.CtcError):
        ctc_amd.Plan("ab,bc->ac", dict(a=2, b=0, c=2))  # empty mode


def test_describe_mentions_every_decision():
    text = ctc_amd.Plan("aeim,ebmj->abij", dict(a=6, b=4, i=8, j=6, e=3, m=10)).describe()
    for word in ("free_m", "free_n", "contracted", "perm_a", "swapped", "tile", "layout_a"):
        assert word in text
