"""World-size-2 runs of the sharding logic over gloo on the CPU.  The kernel launch is replaced
by a numpy emulation of the plan's offset tables (the same emulation tests/test_planner.py checks
against the oracle); everything else - slab extents, rank-local plans, the collective - is the
code the GPU path runs."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _emulated_compute(plan, A, B, C, alpha, beta):
    from ctc_amd import _lib
    s = plan.summary()
    t = {q: plan.table(getattr(_lib, q)) for q in
         ("Q_ROW_A", "Q_K_A", "Q_K_B", "Q_COL_B", "Q_ROW_C", "Q_COL_C", "Q_BAT_A", "Q_BAT_B", "Q_BAT_C")}
    a, b, c = A.numpy(), B.numpy(), C.numpy()
    ga, gb = (b, a) if s["swapped"] else (a, b)
    for z in range(s["batch_count"]):
        am = ga[t["Q_BAT_A"][z] + t["Q_ROW_A"][:, None] + t["Q_K_A"][None, :]]
        bm = gb[t["Q_BAT_B"][z] + t["Q_K_B"][:, None] + t["Q_COL_B"][None, :]]
        idx = t["Q_BAT_C"][z] + t["Q_ROW_C"][:, None] + t["Q_COL_C"][None, :]
        c[idx] = alpha * (am @ bm) + (beta * c[idx] if beta else 0.0)


def _worker(rank, world, port, spec, ext, mode, reduce, out_dir, nchunks=4):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ctc_amd.sharded import ShardedContraction
        from oracle import ref
        sc = ShardedContraction(spec, ext, "f64", mode, rank, world, reduce=reduce, nchunks=nchunks)
        ins, outm = spec.split("->")
        la, lb = ins.split(",")
        A_full = torch.from_numpy(ref.fill(int(np.prod([ext[c] for c in la])), 0xC7C0A))
        B_full = torch.from_numpy(ref.fill(int(np.prod([ext[c] for c in lb])), 0xC7C0B))
        A, B = sc.slab_of(A_full, la), sc.slab_of(B_full, lb)
        assert A.numel() == sc.local_numel("a") and B.numel() == sc.local_numel("b")
        n = sc.out_numel()
        if sc.kind == "contracted":
            assert n == (sc.full_c_numel() if reduce == "all_reduce" else sc.full_c_numel() // world)
        C = torch.full((n,), float("nan"), dtype=torch.float64)
        sc.execute(A, B, C, compute=_emulated_compute)
        np.save(os.path.join(out_dir, "n%d.npy" % rank), np.array([sc.nchunks]))
        np.save(os.path.join(out_dir, "c%d.npy" % rank), C.numpy())
    finally:
        dist.destroy_process_group()


_port = [0]


def _run(tmp_path, spec, ext, mode, reduce="reduce_scatter", world=2, nchunks=4):
    _port[0] += 1
    port = 29500 + (os.getpid() * 7 + _port[0]) % 2000
    mp.spawn(_worker, args=(world, port, spec, ext, mode, reduce, str(tmp_path), nchunks), nprocs=world, join=True)
    return [np.load(os.path.join(str(tmp_path), "c%d.npy" % r)) for r in range(world)]


SPEC = "aeim,ebmj->abij"
EXT = dict(a=4, b=6, i=3, j=4, e=6, m=2)


def _want(oracle):
    A = oracle.fill(4 * 6 * 3 * 2, 0xC7C0A)
    B = oracle.fill(6 * 6 * 2 * 4, 0xC7C0B)
    return oracle.contract(SPEC, EXT, A, B)


def test_free_index_slabs_need_no_collective(tmp_path, oracle):
    parts = _run(tmp_path, SPEC, EXT, "a")
    want = _want(oracle).reshape(4, 6, 3, 4)
    for r, got in enumerate(parts):
        assert np.allclose(got.reshape(2, 6, 3, 4), want[2 * r:2 * r + 2], atol=1e-13)


def test_free_index_in_the_middle_of_c(tmp_path, oracle):
    parts = _run(tmp_path, SPEC, EXT, "b")
    want = _want(oracle).reshape(4, 6, 3, 4)
    for r, got in enumerate(parts):
        assert np.allclose(got.reshape(4, 3, 3, 4), want[:, 3 * r:3 * r + 3], atol=1e-13)


def test_contracted_index_reduce_scatter(tmp_path, oracle):
    parts = _run(tmp_path, SPEC, EXT, "e", "reduce_scatter")
    want = _want(oracle)
    assert np.allclose(np.concatenate(parts), want, atol=1e-13)


def test_contracted_index_all_reduce(tmp_path, oracle):
    parts = _run(tmp_path, SPEC, EXT, "e", "all_reduce")
    want = _want(oracle)
    for got in parts:
        assert np.allclose(got, want, atol=1e-13)


@pytest.mark.parametrize("world,nchunks,want_chunks", [(2, 2, 2), (2, 4, 2), (4, 1, 1), (3, 2, 1)])
def test_contracted_index_pipelined_chunks_keep_contiguous_ownership(tmp_path, oracle, world, nchunks, want_chunks):
    """C's slowest mode a = 4: with p ranks the pipeline may cut each rank's rows into 4 / p / n pieces;
    whatever the cut, rank r must own the flat range r of C (p = 3 does not divide 4: one plain reduce-scatter)."""
    ext = dict(a=4, b=6, i=3, j=4, e=12, m=2)
    A = oracle.fill(4 * 12 * 3 * 2, 0xC7C0A)
    B = oracle.fill(12 * 6 * 2 * 4, 0xC7C0B)
    want = oracle.contract(SPEC, ext, A, B)
    parts = _run(tmp_path, SPEC, ext, "e", "reduce_scatter", world=world, nchunks=nchunks)
    assert int(np.load(os.path.join(str(tmp_path), "n0.npy"))[0]) == want_chunks
    assert np.allclose(np.concatenate(parts), want, atol=1e-13)


def test_contracted_index_all_reduce_in_chunks_with_the_slow_mode_in_b(tmp_path, oracle):
    """C's slowest mode comes from B here (role swap inside the chunk plans), all-reduce in 3 chunks."""
    spec, ext = "ek,bek->ba"[:0] or "eai,bem->baim", dict(a=3, b=6, i=2, m=4, e=4)
    A = oracle.fill(4 * 3 * 2, 0xC7C0A)
    B = oracle.fill(6 * 4 * 4, 0xC7C0B)
    want = oracle.contract(spec, ext, A, B)
    parts = _run(tmp_path, spec, ext, "e", "all_reduce", world=2, nchunks=3)
    assert int(np.load(os.path.join(str(tmp_path), "n0.npy"))[0]) == 3
    for got in parts:
        assert np.allclose(got, want, atol=1e-13)
    parts = _run(tmp_path, spec, ext, "e", "reduce_scatter", world=2, nchunks=3)
    assert np.allclose(np.concatenate(parts), want, atol=1e-13)


def test_rejects_uneven_split():
    from ctc_amd.sharded import ShardedContraction
    with pytest.raises(ValueError):
        ShardedContraction(SPEC, EXT, "f64", "i", 0, 2)
