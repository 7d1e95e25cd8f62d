"""Many equal small products: the batched plan zab = zak,zkb (tiled GEMM kernel, batch as a grid dimension) against
the grouped kernel (one wave per problem) on the same data.  One JSON line per size."""
import json
import os
import statistics
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctc_amd  # noqa: E402


def timed(fn, iters=6, seconds=0.1):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        fn()
        torch.cuda.synchronize()
    each = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        each.append(e0.elapsed_time(e1))
    return statistics.median(each)


dev = torch.device("cuda:0")
for dt, tdt in (("f64", torch.float64), ("f32", torch.float32)):
    CASES = ((100000, 16, 16, 16), (100000, 8, 8, 8), (50000, 24, 24, 24), (30000, 32, 32, 32), (20000, 40, 40, 40), (10000, 48, 48, 48),
                       (8000, 64, 64, 64), (50000, 16, 16, 64), (50000, 32, 32, 8), (50000, 12, 20, 28))
    if os.environ.get("SMALL_LONG_K"):  # beyond the K the plan route takes
        CASES = ((50000, 16, 16, 128), (20000, 16, 16, 256), (20000, 32, 32, 128), (10000, 32, 32, 256), (50000, 8, 8, 128), (10000, 16, 16, 1024), (20000, 24, 24, 96))
    for z, m, n, k in CASES:
        A = torch.empty(z * m * k, dtype=tdt, device=dev)
        B = torch.empty(z * k * n, dtype=tdt, device=dev)
        C1 = torch.empty(z * m * n, dtype=tdt, device=dev)
        C2 = torch.empty(z * m * n, dtype=tdt, device=dev)
        ctc_amd.fill(A, 1)
        ctc_amd.fill(B, 2)
        plan = ctc_amd.Plan("zak,zkb->zab", dict(z=z, a=m, b=n, k=k), dt)
        g0 = ctc_amd.get_option("launches_grouped")
        t_plan = timed(lambda: plan.execute(A, B, C1))
        routed = ctc_amd.get_option("launches_grouped") > g0
        ctc_amd.set_option("small_batched", 0)
        t_tiled = timed(lambda: plan.execute(A, B, C1)) if routed else t_plan
        ctc_amd.set_option("small_batched", 1)
        desc, _, _, _ = ctc_amd.pack_small_problems([(m, n, k)] * z)
        d_desc = torch.from_numpy(desc.view(np.uint8)).to(dev)
        t_grp = timed(lambda: ctc_amd.grouped_contract(d_desc, A, B, C2))
        fl = 2.0 * z * m * n * k
        nbytes = (A.numel() + B.numel() + C1.numel()) * A.element_size()
        print(json.dumps({"dtype": dt, "batch": z, "m": m, "n": n, "k": k, "plan_takes_grouped_kernel": routed, "tiled_ms": round(t_tiled, 4), "plan_ms": round(t_plan, 4), "plan_tflops": round(fl / t_plan * 1e-9, 2),
                          "grouped_ms": round(t_grp, 4), "grouped_tflops": round(fl / t_grp * 1e-9, 2), "grouped_TBps": round(nbytes / t_grp * 1e-9, 2),
                          "plan_over_grouped_time": round(t_plan / t_grp, 2), "max_diff": float((C1 - C2).abs().max())}), flush=True)
        del A, B, C1, C2, d_desc
        torch.cuda.empty_cache()
