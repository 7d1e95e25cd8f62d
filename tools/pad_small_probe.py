"""Odd-sized products whose smaller edge lies between 512 and 1023, all four operand orders: fused kernel against the
padded K-major copies (run with CTC_TC_PAD_MIN_MN=512 to send them through the copies).  One JSON line per case."""
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctc_amd  # noqa: E402


def timed(fn, iters=8, seconds=0.1):
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
SHAPES = ((979, 2363, 6616), (601, 3001, 2001), (777, 777, 4097), (515, 5001, 1025), (901, 903, 8193), (1001, 999, 1501), (700, 1300, 3000), (650, 651, 12001))
for dt, tdt in (("f32", torch.float32), ("f64", torch.float64)):
    for m, n, k in SHAPES:
        for spec in ("ak,kb->ab", "ak,bk->ab", "ka,kb->ab", "ka,bk->ab"):
            A = torch.empty(m * k, dtype=tdt, device=dev)
            B = torch.empty(k * n, dtype=tdt, device=dev)
            C = torch.empty(m * n, dtype=tdt, device=dev)
            ctc_amd.fill(A, 1)
            ctc_amd.fill(B, 2)
            plan = ctc_amd.Plan(spec, dict(a=m, b=n, k=k), dt)
            plan.prepare()
            ms = timed(lambda: plan.execute(A, B, C))
            s = plan.summary()
            print(json.dumps({"dtype": dt, "spec": spec, "m": m, "n": n, "k": k, "tile": list(s["tile"])[:2], "layouts": [s["layout_a"], s["layout_b"]],
                              "transposes": s["transposes"], "ms": round(ms, 4), "tflops": round(plan.flops / ms * 1e-9, 1)}), flush=True)
            del A, B, C
            torch.cuda.empty_cache()
