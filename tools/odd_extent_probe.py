"""Odd extents (no 16-byte vectors are legal on the operands): the headline contraction through the padded copies
(the automatic choice for large problems) against the fused kernel with element loads on the same plan
("pad_transposes" option off).  One JSON line per (element type, extent).

  python tools/odd_extent_probe.py [extents ...]"""
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import ctc_amd  # noqa: E402

PEAK = {"f64": 78.6, "f32": 157.3, "bf16": 2500.0}


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


def main():
    extents = [int(x) for x in sys.argv[1:]] or [33, 37, 41, 45, 49, 55, 63, 65, 75, 81, 95]
    dev = torch.device("cuda:0")
    for dtype, tdt in (("f64", torch.float64), ("f32", torch.float32), ("bf16", torch.bfloat16)):
        for n in extents:
            ext = dict(a=n, b=n, i=n, j=n, e=n, m=n)
            A = torch.zeros(n ** 4, dtype=torch.float32, device=dev)
            B = torch.zeros(n ** 4, dtype=torch.float32, device=dev)
            ctc_amd.fill(A, 1)
            ctc_amd.fill(B, 2)
            A, B = A.to(tdt), B.to(tdt)
            C1 = torch.empty(n ** 4, dtype=tdt, device=dev)
            C2 = torch.empty(n ** 4, dtype=tdt, device=dev)
            plan = ctc_amd.Plan("aeim,ebmj->abij", ext, dtype)
            s = plan.summary()
            plan.prepare()
            flops = 2.0 * n ** 6
            ctc_amd.set_option("pad_transposes", 0)
            t_fused = timed(lambda: plan.execute(A, B, C1))
            ctc_amd.set_option("pad_transposes", 1)
            line = {"dtype": dtype, "extent": n, "automatic": s["transposes"], "tile": list(s["tile"]),
                    "fused_ms": round(t_fused, 4), "fused_tflops": round(flops / t_fused * 1e-9, 2),
                    "fused_frac_of_peak": round(flops / t_fused * 1e-9 / PEAK[dtype], 3)}
            if s["transposes"] == "padded_copies":
                t_pad = timed(lambda: plan.execute(A, B, C2))
                line.update({"padded_ms": round(t_pad, 4), "padded_tflops": round(flops / t_pad * 1e-9, 2),
                             "padded_frac_of_peak": round(flops / t_pad * 1e-9 / PEAK[dtype], 3),
                             "padded_over_fused": round(t_fused / t_pad, 3),
                             "max_rel_diff": float((C2.double() - C1.double()).abs().max() / C1.double().abs().max())})
            print(json.dumps(line), flush=True)
            del A, B, C1, C2, plan
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
