"""Strong-scaling rehearsal on ONE GPU: the plan that rank 0 of 1 / 2 / 4 / 8 would run, timed alone.

For every contraction bench.py splits over the ranks (the headline contraction split along the free index
a, BASELINE config 4 split along a, the headline contraction split along the contracted index e - there the
chunk GEMMs without the collectives) one JSON line per world size with the slab's time and T(1) / T(slab):
the speed-up the N-GPU run reaches if nothing but the kernels costs time (no collective, no launch skew),
i.e. the ceiling of the scaling curve.  Every rank runs the same shape, so rank 0 stands for all.

  python tools/slab_rehearsal.py [headline|config4|contracted ...]  > profiles/rNN_slab_rehearsal.jsonl
"""
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import ctc_amd  # noqa: E402

PEAK = 78.6
SPEC = "aeim,ebmj->abij"
WORLDS = (1, 2, 4, 8)


def prewarm(fn, seconds=0.15):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        fn()
        torch.cuda.synchronize()


def time_plan(plan, A, B, C, iters):
    plan.prepare()
    prewarm(lambda: plan.execute(A, B, C))
    each = plan.time_each(A, B, C, iters=iters)
    return statistics.median(each), sum(each) / len(each)


def emit(case, world, ms_med, ms_avg, flops_rank, t1, extra):
    line = {"case": case, "n_ranks": world, "ms_median": round(ms_med, 4), "ms_avg": round(ms_avg, 4),
            "tflops_per_gpu": round(flops_rank / ms_med * 1e-9, 2), "frac_of_fp64_mfma_peak": round(flops_rank / ms_med * 1e-9 / PEAK, 4),
            "speedup_ceiling": round(t1 / ms_med, 3), "of_ideal": round(t1 / ms_med / world, 4)}
    line.update(extra)
    print(json.dumps(line), flush=True)


def headline(dev, n=96, iters=10):
    B = torch.empty(n ** 4, dtype=torch.float64, device=dev)
    ctc_amd.fill(B, 0xC7C0B)
    full = ctc_amd.Plan(SPEC, dict(a=n, b=n, i=n, j=n, e=n, m=n), "f64")
    t1 = None
    for world in WORLDS:
        plan, off_a, _, _ = full.shard("a", 0, world)
        a_loc = n // world
        A = torch.empty(a_loc * n ** 3, dtype=torch.float64, device=dev)
        C = torch.empty(a_loc * n ** 3, dtype=torch.float64, device=dev)
        ctc_amd.fill(A, 0xC7C0A, first=off_a)
        med, avg = time_plan(plan, A, B, C, iters)
        t1 = t1 or med
        sch = plan.schedule()
        emit("headline abij=aeim,ebmj fp64 extent %d, a split" % n, world, med, avg, plan.flops, t1,
             {"a_per_rank": a_loc, "tiles": sch["tile_rows"] * sch["tile_cols"], "whole_tiles": sch["whole_tiles"],
              "tail_tiles": sch["tail_tiles"], "split": sch["split"], "kernel": plan.gemm_kernel_name()[0]})
        del A, C
        torch.cuda.empty_cache()


def config4(dev, n=40, d=64, iters=6):
    spec = "adij,bcdk->abcijk"
    B = torch.empty(n * n * d * n, dtype=torch.float64, device=dev)
    ctc_amd.fill(B, 0xC7C0B)
    full = ctc_amd.Plan(spec, dict(a=n, b=n, c=n, i=n, j=n, k=n, d=d), "f64")
    names = ("launches_classic", "launches_dma_gemm", "launches_stream", "launches_stream_f64_dma")
    t1 = None
    for world in WORLDS:
        plan, off_a, _, _ = full.shard("a", 0, world)
        a_loc = n // world
        A = torch.empty(a_loc * d * n * n, dtype=torch.float64, device=dev)
        C = torch.empty(a_loc * n ** 5, dtype=torch.float64, device=dev)
        ctc_amd.fill(A, 0xC7C0A, first=off_a)
        before = [ctc_amd.get_option(x) for x in names]
        med, avg = time_plan(plan, A, B, C, iters if world > 1 else 4)
        took = [x[9:] for x, b in zip(names, before) if ctc_amd.get_option(x) > b]
        t1 = t1 or med
        emit("config4 abcijk=adij,bcdk fp64 free 40 d 64, a split", world, med, avg, plan.flops, t1,
             {"a_per_rank": a_loc, "c_GB_per_rank": round(a_loc * n ** 5 * 8e-9, 2),
              "c_write_TBps": round(a_loc * n ** 5 * 8 / med * 1e-9, 2), "took": took})
        del A, C
        torch.cuda.empty_cache()


def contracted(dev, n=96, iters=6, nchunks=4):
    from ctc_amd.sharded import ShardedContraction
    ext = dict(a=n, b=n, i=n, j=n, e=n, m=n)
    t1 = None
    for world in WORLDS:
        sc = ShardedContraction(SPEC, ext, "f64", "e", 0, world, reduce="reduce_scatter", nchunks=nchunks)
        A = torch.empty(sc.local_numel("a"), dtype=torch.float64, device=dev)
        B = torch.empty(sc.local_numel("b"), dtype=torch.float64, device=dev)
        partial = torch.empty(sc.workspace_numel(), dtype=torch.float64, device=dev)
        ctc_amd.fill(A, 0xC7C0A)
        ctc_amd.fill(B, 0xC7C0B)
        sc.prepare()

        def gemms():
            for c in range(sc.nchunks):
                plan, oa, ob, op, _, elems = sc.chunk(c)
                plan.execute(A[oa:], B[ob:], partial[op:op + elems])
        prewarm(gemms)
        each = []
        for _ in range(iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            gemms()
            e1.record()
            e1.synchronize()
            each.append(e0.elapsed_time(e1))
        med, avg = statistics.median(each), sum(each) / len(each)
        t1 = t1 or med
        emit("headline extent %d, contracted index e split, the %d chunk GEMMs of one step without the reduce-scatters" % (n, sc.nchunks),
             world, med, avg, 2 * n ** 6 // world, t1, {"e_per_rank": n // world, "k_per_rank": n * n // world})
        del A, B, partial, sc
        torch.cuda.empty_cache()


if __name__ == "__main__":
    which = sys.argv[1:] or ["headline", "config4", "contracted"]
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    for w in which:
        {"headline": headline, "config4": config4, "contracted": contracted}[w](dev)
