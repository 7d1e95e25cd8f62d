"""Grouped small contractions (BASELINE config 5 shape: many ab = ak,kb with extents of a few tens)
through the C ABI, against the CPU oracle problem by problem."""
import numpy as np
import pytest
import torch

import ctc_amd

pytestmark = pytest.mark.gpu

TOL = {0: 1e-12, 1: 1e-5}
NP = {0: np.float64, 1: np.float32}


def _problem_set(seed, count, choices):
    # extents from splitmix64(seed + i), the same integer stream the generator uses
    from oracle import ref
    dims = []
    for i in range(count):
        x = ref.splitmix64(seed + i)
        dims.append((choices[x % len(choices)], choices[(x >> 8) % len(choices)], choices[(x >> 16) % len(choices)]))
    return dims


def _run(cuda, oracle, dims, code, alpha=1.0, beta=0.0):
    desc, na, nb, nc = ctc_amd.pack_small_problems(dims)
    A = oracle.fill(na, 0xC7C05, code)
    B = oracle.fill(nb, 0xC7C06, code)
    C0 = oracle.fill(nc, 0xC7C07, code)
    dA, dB, dC = (torch.from_numpy(x.copy()).to(cuda) for x in (A, B, C0))
    dP = torch.from_numpy(desc.view(np.uint8).copy()).to(cuda)
    ctc_amd.grouped_contract(dP, dA, dB, dC, alpha, beta)
    got = dC.cpu().numpy().astype(np.float64)
    for p, (m, n, k) in zip(desc, dims):
        a = A[p["off_a"]:p["off_a"] + m * k]
        b = B[p["off_b"]:p["off_b"] + k * n]
        c0 = C0[p["off_c"]:p["off_c"] + m * n].copy()
        want = oracle.contract("ak,kb->ab", dict(a=m, b=n, k=k), a, b, C=c0, alpha=alpha, beta=beta, dtype=code,
                               threads=1).astype(np.float64)
        err = np.abs(got[p["off_c"]:p["off_c"] + m * n] - want).max()
        assert err <= TOL[code] * max(np.abs(want).max(), 1e-300), (m, n, k, err)


@pytest.mark.parametrize("code", [0, 1])
def test_config5_extents(cuda, oracle, code):
    """The whole configuration: 4096 problems, extents drawn from {8, 16, 24, 32}, seed 0xC7C05, one launch."""
    _run(cuda, oracle, _problem_set(0xC7C05, 4096, [8, 16, 24, 32]), code)


@pytest.mark.parametrize("code", [0, 1])
def test_ragged_and_larger_extents(cuda, oracle, code):
    dims = [(1, 1, 1), (3, 5, 7), (17, 33, 9), (40, 50, 6), (64, 64, 64), (5, 70, 35), (33, 1, 2), (16, 16, 3)]
    _run(cuda, oracle, dims, code)
    _run(cuda, oracle, dims, code, alpha=-0.5, beta=1.25)


def test_empty_group_is_a_no_op(cuda):
    x = torch.zeros(4, dtype=torch.float64, device=cuda)
    p = torch.zeros(0, dtype=torch.uint8, device=cuda)
    ctc_amd.grouped_contract(p, x, x, x)
