"""Batches of small products in the four storage orders of the operands (zak / zka, zkb / zbk): the plan on the
grouped kernel against the same plan on the tiled kernels (option small_batched 0).  One JSON line per case."""
import json
import os
import statistics
import sys
import time

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
    for z, m, n, k in ((100000, 16, 16, 16), (100000, 8, 8, 8), (100000, 12, 16, 32), (200000, 4, 4, 4), (50000, 24, 24, 24), (30000, 32, 32, 32), (50000, 12, 20, 28), (50000, 16, 16, 64)):
        for spec in ("zak,zkb->zab", "zak,zbk->zab", "zka,zkb->zab", "zka,zbk->zab"):
            A = torch.empty(z * m * k, dtype=tdt, device=dev)
            B = torch.empty(z * k * n, dtype=tdt, device=dev)
            C1 = torch.empty(z * m * n, dtype=tdt, device=dev)
            C2 = torch.empty(z * m * n, dtype=tdt, device=dev)
            ctc_amd.fill(A, 1)
            ctc_amd.fill(B, 2)
            plan = ctc_amd.Plan(spec, dict(z=z, a=m, b=n, k=k), dt)
            g0 = ctc_amd.get_option("launches_grouped")
            t_grp = timed(lambda: plan.execute(A, B, C1))
            routed = ctc_amd.get_option("launches_grouped") > g0
            ctc_amd.set_option("small_batched", 0)
            t_tiled = timed(lambda: plan.execute(A, B, C2))
            ctc_amd.set_option("small_batched", 1)
            fl = 2.0 * z * m * n * k
            nbytes = (A.numel() + B.numel() + C1.numel()) * A.element_size()
            print(json.dumps({"dtype": dt, "spec": spec, "batch": z, "m": m, "n": n, "k": k, "takes_grouped_kernel": routed, "tiled_ms": round(t_tiled, 4),
                              "plan_ms": round(t_grp, 4), "plan_tflops": round(fl / t_grp * 1e-9, 2), "plan_TBps": round(nbytes / t_grp * 1e-9, 2),
                              "tiled_over_plan_time": round(t_tiled / t_grp, 2), "max_diff": float((C1 - C2).abs().max())}), flush=True)
            del A, B, C1, C2
            torch.cuda.empty_cache()
