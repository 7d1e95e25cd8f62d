"""Contracted-mode sharding through the C layer on ONE GPU: a one-rank RCCL communicator runs the
real ncclReduceScatter / ncclAllReduce calls on the library's second stream, chunk by chunk behind
the GEMMs.  (More ranks: tests/test_sharded_gloo.py checks the same chunk plans and offsets with
world sizes 2-4 on the CPU.)  Tolerance 1e-12 of max |C| (fp64), 1e-5 (fp32)."""
import numpy as np
import pytest
import torch

import ctc_amd
from ctc_amd.sharded import Communicator, ShardedContraction

pytestmark = pytest.mark.gpu

SPEC = "aeim,ebmj->abij"


@pytest.fixture(scope="module")
def comm(cuda):
    torch.cuda.set_device(cuda)
    c = Communicator(0, 1)
    yield c
    del c


@pytest.mark.parametrize("reduce", ["reduce_scatter", "all_reduce"])
@pytest.mark.parametrize("dtype,code,tdt,tol", [("f64", 0, torch.float64, 1e-12), ("f32", 1, torch.float32, 1e-5)])
def test_one_rank_communicator_runs_the_pipelined_sum(cuda, oracle, comm, reduce, dtype, code, tdt, tol):
    ext = dict(a=12, b=10, i=6, j=8, e=9, m=7)
    A = oracle.fill(12 * 9 * 6 * 7, 0xC7C0A, code)
    B = oracle.fill(9 * 10 * 7 * 8, 0xC7C0B, code)
    want = oracle.contract(SPEC, ext, A, B, alpha=0.5, dtype=code).astype(np.float64)
    sc = ShardedContraction(SPEC, ext, dtype, "e", 0, 1, reduce=reduce, nchunks=3)
    assert sc.nchunks == 3
    sc.prepare()
    dA, dB = torch.from_numpy(A).to(cuda), torch.from_numpy(B).to(cuda)
    for use_comm in (comm, None):  # None: one rank without a communicator (the sum is a copy)
        dC = torch.full((sc.out_numel(),), float("nan"), dtype=tdt, device=cuda)
        for _ in range(2):  # twice: the second call reuses the partial buffer while nothing else is in flight
            sc.execute(dA, dB, dC, alpha=0.5, comm=use_comm)
        torch.cuda.synchronize()
        got = dC.cpu().numpy().astype(np.float64)
        assert np.abs(got - want).max() <= tol * np.abs(want).max()


def test_rejects_free_modes_and_short_workspaces(cuda, comm):
    ext = dict(a=4, b=4, i=4, j=4, e=4, m=4)
    with pytest.raises(ctc_amd.CtcError, match="CONTRACTED"):
        from ctc_amd import _lib
        import ctypes
        h = ctypes.c_void_p()
        names = "".join(ext.keys()).encode()
        arr = (ctypes.c_int64 * 6)(*ext.values())
        ctc_amd.host.check(_lib.load().ctc_tc_sharded_create(SPEC.encode(), 6, names, arr, 0, b"a", 0, 1, 2, 0, ctypes.byref(h)))
    sc = ShardedContraction(SPEC, ext, "f64", "e", 0, 1)
    x = torch.zeros(256, dtype=torch.float64, device=cuda)
    with pytest.raises(ValueError, match="scratch"):
        sc.execute(x, x, x.clone(), comm=comm, scratch=torch.zeros(8, dtype=torch.float64, device=cuda))
