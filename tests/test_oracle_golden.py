"""The CPU oracle against the committed vectors (numpy.einsum outputs on generator inputs).

The reference tree holds no contraction vectors, so these pin the oracle's
restatement only; parity with the reference stays unpinned."""
import numpy as np
import pytest

CODE = {"f64": 0, "f32": 1, "bf16": 2}
# oracle accumulates in f64; f32/bf16 results are rounded once on store
RTOL = {"f64": 1e-13, "f32": 2e-7, "bf16": 8e-3}


def test_splitmix64_published_known_answer(oracle):
    # first three outputs of the published SplitMix64 stream seeded with 0
    # (state advances by the golden gamma before each mix)
    gamma = 0x9E3779B97F4A7C15
    assert oracle.splitmix64(0) == 0xE220A8397B1DCDAF
    assert oracle.splitmix64(gamma) == 0x6E789E6AA1B965F4
    assert oracle.splitmix64(2 * gamma & 0xFFFFFFFFFFFFFFFF) == 0x06C45D188009454F


def test_generator_vectors(oracle, golden):
    for seed_hex, v in golden["generator"].items():
        seed = int(seed_hex, 16)
        assert [hex(oracle.splitmix64(seed + k)) for k in range(4)] == v["splitmix64"]
        assert [float(x).hex() for x in oracle.fill(4, seed, 0)] == v["f64"]
        assert [float(x).hex() for x in oracle.fill(4, seed, 1)] == v["f32"]
        assert [int(x) for x in oracle.fill(4, seed, 2)] == v["bf16"]


def test_generator_range_and_offset(oracle):
    a = oracle.fill(4096, 7, 0)
    assert a.min() >= -1.0 and a.max() < 1.0
    # element k depends on seed + k only
    b = oracle.fill(4000, 7 + 96, 0)
    assert np.array_equal(a[96:], b)
    f = oracle.fill(4096, 7, 1)
    assert f.dtype == np.float32 and f.min() >= -1.0 and f.max() < 1.0


def test_bf16_rounding(oracle):
    x = np.array([1.0, 1.00390625, 1.005859375, -2.5, 0.0, 3.0e38], dtype=np.float32)
    h = oracle.f32_to_bf16(x)
    back = oracle.bf16_to_f32(h)
    assert back[0] == 1.0 and back[3] == -2.5 and back[4] == 0.0
    assert back[1] == 1.0           # tie rounds to even mantissa
    assert back[2] == 1.0078125     # above the tie rounds up


def _case_ids(golden):
    return [c["name"] for c in golden["cases"]]


def test_classification_vectors(oracle, golden):
    for c in golden["cases"]:
        got = oracle.classify(c["spec"])
        want = c["classification"]
        for key in ("names", "classes", "a", "b", "c", "batch", "m", "n", "k", "perm_a", "perm_b", "perm_c"):
            assert got[key] == want[key], (c["name"], key)


def test_contract_vectors(oracle, golden):
    for c in golden["cases"]:
        ext, dt = c["extents"], CODE[c["dtype"]]
        ins, outm = c["spec"].split("->")
        la, lb = ins.split(",")
        na = int(np.prod([ext[ch] for ch in la]))
        nb = int(np.prod([ext[ch] for ch in lb]))
        A = oracle.fill(na, c["seed_a"], dt)
        B = oracle.fill(nb, c["seed_b"], dt)
        C = oracle.contract(c["spec"], ext, A, B, dtype=dt)
        assert C.size == c["n_out"]
        Cf = oracle.bf16_to_f32(C).astype(np.float64) if c["dtype"] == "bf16" else C.astype(np.float64)
        scale = float.fromhex(c["max_abs"])
        tol = RTOL[c["dtype"]] * max(scale, 1e-300)
        for i, hx in c["samples"]:
            assert abs(Cf[i] - float.fromhex(hx)) <= tol, (c["name"], i)
        assert abs(Cf.sum() - float.fromhex(c["sum"])) <= tol * C.size
        assert abs(np.abs(Cf).sum() - float.fromhex(c["sum_abs"])) <= tol * C.size


def test_contract_alpha_beta_and_strides(oracle):
    ext = dict(a=5, b=4, k=6)
    A = oracle.fill(30, 11)
    B = oracle.fill(24, 12)
    C0 = oracle.fill(20, 13)
    want = 0.5 * np.einsum("ak,kb->ab", A.reshape(5, 6), B.reshape(6, 4)) - 2.0 * C0.reshape(5, 4)
    got = oracle.contract("ak,kb->ab", ext, A, B, C=C0.copy(), alpha=0.5, beta=-2.0)
    assert np.allclose(got.reshape(5, 4), want, rtol=0, atol=1e-14)
    # A stored transposed (k fastest -> a fastest) through explicit strides
    At = np.ascontiguousarray(A.reshape(5, 6).T).reshape(-1)
    got2 = oracle.contract("ak,kb->ab", ext, At, B, strides_a=[1, 5])
    assert np.allclose(got2.reshape(5, 4), np.einsum("ak,kb->ab", A.reshape(5, 6), B.reshape(6, 4)), atol=1e-14)


def test_contract_rejects_bad_specs(oracle):
    for bad in ("ab,bc", "ab,bc->ad", "aa,ab->b", "ab,cd->abc", "a1,1b->ab"):
        with pytest.raises(ValueError):
            oracle.classify(bad)


def test_permute_against_numpy(oracle):
    rng = np.random.RandomState(0)
    for shape, perm in (((3, 4, 5), (2, 0, 1)), ((7, 1, 6, 2), (3, 2, 1, 0)), ((8,), (0,)),
                        ((2, 3, 4, 5, 6), (0, 4, 1, 3, 2))):
        for dt in (np.float64, np.float32, np.uint16):
            a = (rng.rand(*shape) * 1000).astype(dt)
            got = oracle.permute(a, shape, perm)
            assert np.array_equal(got, np.ascontiguousarray(a.transpose(perm)).reshape(-1))
