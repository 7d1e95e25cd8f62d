"""The CCSD-style contraction with its modes in other orders (operands K-major or outer-major, contracted modes in
different orders in A and B, C stored along other modes), fp64 and fp32, extent 64 and 72: which kernel each takes and
what it reaches.  One JSON line per case; outliers are what to look at."""
import itertools
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctc_amd  # noqa: E402

PEAK = {"f64": 78.6, "f32": 157.3}
SPECS = [
    "aeim,ebmj->abij",   # the headline
    "aeim,ebmj->ijab", "aeim,ebmj->jiba", "aeim,ebmj->aibj",
    "aime,bjme->abij", "aime,mebj->abij", "emai,mejb->abij", "iame,jbem->abij",
    "eaim,mbej->abij", "maei,jebm->abij", "aeim,bjem->baji", "eima,ebmj->abij",
]


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
names = ("launches_classic", "launches_dma_gemm", "launches_stream", "launches_stream_f64_dma")
extents = [int(x) for x in sys.argv[1:]] or [64, 72]
for dt, tdt in (("f64", torch.float64), ("f32", torch.float32)):
    for n in extents:
        ext = dict(a=n, b=n, i=n, j=n, e=n, m=n)
        A = torch.empty(n ** 4, dtype=tdt, device=dev)
        B = torch.empty(n ** 4, dtype=tdt, device=dev)
        C = torch.empty(n ** 4, dtype=tdt, device=dev)
        ctc_amd.fill(A, 1)
        ctc_amd.fill(B, 2)
        for spec in SPECS:
            plan = ctc_amd.Plan(spec, ext, dt)
            s = plan.summary()
            before = [ctc_amd.get_option(x) for x in names]
            ms = timed(lambda: plan.execute(A, B, C))
            took = [x[9:] for x, b in zip(names, before) if ctc_amd.get_option(x) > b]
            tf = plan.flops / ms * 1e-9
            print(json.dumps({"dtype": dt, "extent": n, "spec": spec, "tflops": round(tf, 1), "frac": round(tf / PEAK[dt], 3), "took": took,
                              "layouts": [s["layout_a"], s["layout_b"]], "vec_c": s["vec_c"], "swapped": s["swapped"], "transposes": s["transposes"]}), flush=True)
