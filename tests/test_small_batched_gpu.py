"""Batches of small products (zab = zak,zkb with M, N <= 32): the plan runs on the grouped kernel, one wave per
product, instead of the tiled kernels.  Against the CPU oracle, with alpha / beta, roles exchanged, and the shapes
that must NOT take it (larger products, a C stored along the batch mode; strided rows are fine: leading dimensions)."""
import numpy as np
import pytest
import torch

import ctc_amd

pytestmark = pytest.mark.gpu

TOL = {"f64": 1e-12, "f32": 1e-5}
CODE = {"f64": 0, "f32": 1}
TDT = {"f64": torch.float64, "f32": torch.float32}


def _run(cuda, oracle, spec, ext, dtype, expect_grouped, alpha=1.0, beta=0.0, strides=None):
    ins, outm = spec.split("->")
    la, lb = ins.split(",")
    strides = strides or {}

    def span(letters, st):
        if st is None:
            return int(np.prod([ext[c] for c in letters]))
        return 1 + sum((ext[c] - 1) * s for c, s in zip(letters, st))
    na, nb, nc = span(la, strides.get("a")), span(lb, strides.get("b")), span(outm, strides.get("c"))
    A, B, C0 = oracle.fill(na, 11, CODE[dtype]), oracle.fill(nb, 12, CODE[dtype]), oracle.fill(nc, 13, CODE[dtype])
    plan = ctc_amd.Plan(spec, ext, dtype, strides.get("a"), strides.get("b"), strides.get("c"))
    dC = torch.from_numpy(C0.copy()).to(cuda)
    before = ctc_amd.get_option("launches_grouped")
    plan.execute(torch.from_numpy(A).to(cuda), torch.from_numpy(B).to(cuda), dC, alpha=alpha, beta=beta)
    assert (ctc_amd.get_option("launches_grouped") - before == 1) == expect_grouped, plan.summary()
    want = oracle.contract(spec, ext, A, B, C=C0.copy(), alpha=alpha, beta=beta, dtype=CODE[dtype],
                           strides_a=strides.get("a"), strides_b=strides.get("b"), strides_c=strides.get("c"))
    got = dC.cpu().numpy().astype(np.float64)
    assert np.abs(got - want.astype(np.float64)).max() <= 2 * TOL[dtype] * np.abs(want).max()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("z", [513, 514, 515, 1023])
def test_four_small_products_per_wave_with_a_ragged_count(cuda, oracle, dtype, z):
    """Products of at most 16 x 16 under K <= 32 share a wave four at a time: batch counts that are no multiple of
    four (the last wave holds one to three), odd sizes, alpha / beta, every operand order."""
    _run(cuda, oracle, "zak,zkb->zab", dict(z=z, a=8, b=8, k=8), dtype, True, alpha=1.5, beta=-0.5)
    _run(cuda, oracle, "zka,zbk->zab", dict(z=z, a=5, b=16, k=31), dtype, True)
    _run(cuda, oracle, "zak,zbk->zab", dict(z=z, a=16, b=3, k=1), dtype, True, beta=2.0)
    _run(cuda, oracle, "zka,zkb->zab", dict(z=z, a=1, b=1, k=32), dtype, True)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_batches_of_small_products_take_the_grouped_kernel(cuda, oracle, dtype):
    _run(cuda, oracle, "zak,zkb->zab", dict(z=3000, a=16, b=16, k=16), dtype, True)
    _run(cuda, oracle, "zak,zkb->zab", dict(z=2000, a=12, b=20, k=28), dtype, True, alpha=-0.5, beta=1.25)
    _run(cuda, oracle, "zak,zkb->zab", dict(z=1500, a=7, b=9, k=5), dtype, True)          # odd everything
    _run(cuda, oracle, "zak,zkb->zab", dict(z=700, a=32, b=24, k=64), dtype, True, beta=-1.0)
    _run(cuda, oracle, "yzak,yzkb->yzab", dict(y=30, z=40, a=8, b=8, k=8), dtype, True)   # two batch modes
    # C's unit-stride mode belongs to the first operand: the roles are exchanged and the tables still describe
    # row-major products
    _run(cuda, oracle, "zkb,zak->zab", dict(z=1000, a=16, b=24, k=8), dtype, True)
    # the other storage orders of the operands: A as [k][m], B as [n][k], both; a batch mode in the middle
    _run(cuda, oracle, "zak,zbk->zab", dict(z=900, a=16, b=16, k=16), dtype, True, alpha=2.0, beta=0.5)
    _run(cuda, oracle, "zka,zkb->zab", dict(z=900, a=12, b=20, k=28), dtype, True)
    _run(cuda, oracle, "zka,zbk->zab", dict(z=800, a=7, b=9, k=5), dtype, True, beta=-1.0)
    _run(cuda, oracle, "azk,kzb->azb", dict(z=700, a=16, b=12, k=24), dtype, True)
    # small products under a longer K
    _run(cuda, oracle, "zak,zkb->zab", dict(z=600, a=16, b=16, k=1000), dtype, True, alpha=0.5, beta=0.5)
    _run(cuda, oracle, "zka,zbk->zab", dict(z=600, a=9, b=13, k=257), dtype, True)
    _run(cuda, oracle, "zak,zkb->zab", dict(z=600, a=24, b=24, k=96), dtype, dtype == "f64")


def test_shapes_that_keep_the_tiled_kernels(cuda, oracle):
    _run(cuda, oracle, "zak,zkb->zab", dict(z=600, a=40, b=40, k=16), "f64", False)       # larger than 32 x 32
    _run(cuda, oracle, "zak,zkb->abz", dict(z=600, a=16, b=16, k=16), "f64", False)       # C stored along the batch mode
    _run(cuda, oracle, "zak,zkb->zab", dict(z=100, a=16, b=16, k=16), "f64", False)       # a small batch
    _run(cuda, oracle, "zak,zkb->zab", dict(z=600, a=32, b=32, k=32), "f32", False)       # fp32 ties there
    ctc_amd.set_option("small_batched", 0)
    try:
        _run(cuda, oracle, "zak,zkb->zab", dict(z=3000, a=16, b=16, k=16), "f64", False)
    finally:
        ctc_amd.set_option("small_batched", 1)
