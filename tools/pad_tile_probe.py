"""Odd-extent fp32 contractions of middling size (150 - 220 large tiles): the planner's choice against a forced tile
(CTC_TC_FORCE_TILE=64 / 128; the padded K-major copies need the large tile).  One JSON line per case."""
import json
import os
import statistics
import sys
import time

import numpy as np
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
CASES = [("aeim,ebmj->abij", dict(a=41, b=43, i=41, j=41, e=23, m=45)), ("ka,bk->ab", dict(a=1601, b=1603, k=1537)),
         ("mia,jmb->abij", dict(a=41, b=43, i=41, j=41, m=1201)), ("ak,kb->ab", dict(a=1501, b=1503, k=1025)),
         ("aeim,ebmj->abij", dict(a=37, b=37, i=37, j=37, e=37, m=37)), ("aeim,ebmj->abij", dict(a=39, b=39, i=39, j=39, e=39, m=39)),
         ("aeim,ebmj->abij", dict(a=43, b=43, i=43, j=43, e=43, m=43)), ("ak,kb->ab", dict(a=1801, b=1803, k=2049))]
for dt, tdt in (("f32", torch.float32), ("f64", torch.float64)):
    for spec, ext in CASES:
        ins, outm = spec.split("->")
        la, lb = ins.split(",")
        num = lambda t: int(np.prod([ext[c] for c in t]))  # noqa: E731
        A = torch.empty(num(la), dtype=tdt, device=dev)
        B = torch.empty(num(lb), dtype=tdt, device=dev)
        C = torch.empty(num(outm), dtype=tdt, device=dev)
        ctc_amd.fill(A, 1)
        ctc_amd.fill(B, 2)
        plan = ctc_amd.Plan(spec, ext, dt)
        plan.prepare()
        ms = timed(lambda: plan.execute(A, B, C))
        s = plan.summary()
        print(json.dumps({"dtype": dt, "spec": spec, "ext": ext, "M": s["M"], "N": s["N"], "K": s["K"], "tile": list(s["tile"])[:2], "transposes": s["transposes"],
                          "ms": round(ms, 4), "tflops": round(plan.flops / ms * 1e-9, 1), "force": os.environ.get("CTC_TC_FORCE_TILE")}), flush=True)
