"""Batched products zab = zak,zkb (and with the batch mode in other positions), fp64 / fp32: rate and kernel per
(batch count, size).  One JSON line per case."""
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctc_amd  # noqa: E402

PEAK = {"f64": 78.6, "f32": 157.3}
CASES = [("zak,zkb->zab", 4, 2048, 2048, 2048), ("zak,zkb->zab", 64, 512, 512, 512), ("zak,zkb->zab", 1024, 128, 128, 128),
         ("zak,zkb->zab", 4096, 64, 64, 64), ("zak,zkb->zab", 16, 1024, 1024, 256), ("zak,zkb->zab", 8, 1000, 1000, 1000),
         ("azk,kzb->azb", 64, 512, 512, 512), ("akz,kbz->abz", 64, 512, 512, 512), ("zak,zbk->zab", 64, 512, 512, 512),
         ("zak,zkb->abz", 64, 512, 512, 512), ("zak,zkb->zab", 3, 3000, 3000, 3000), ("zak,zkb->zab", 100000, 16, 16, 16)]


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
names = ("launches_classic", "launches_dma_gemm", "launches_stream", "launches_stream_f64_dma", "launches_grouped")
for dt, tdt in (("f64", torch.float64), ("f32", torch.float32)):
    for spec, z, m, n, k in CASES:
        A = torch.empty(z * m * k, dtype=tdt, device=dev)
        B = torch.empty(z * k * n, dtype=tdt, device=dev)
        C = torch.empty(z * m * n, dtype=tdt, device=dev)
        ctc_amd.fill(A, 1)
        ctc_amd.fill(B, 2)
        plan = ctc_amd.Plan(spec, dict(z=z, a=m, b=n, k=k), dt)
        s = plan.summary()
        before = [ctc_amd.get_option(x) for x in names]
        ms = timed(lambda: plan.execute(A, B, C))
        took = [x[9:] for x, b in zip(names, before) if ctc_amd.get_option(x) > b]
        tf = plan.flops / ms * 1e-9
        nbytes = (A.numel() + B.numel() + C.numel()) * A.element_size()
        print(json.dumps({"dtype": dt, "spec": spec, "batch": z, "M": m, "N": n, "K": k, "ms": round(ms, 4), "tflops": round(tf, 1),
                          "frac_mfma": round(tf / PEAK[dt], 3), "TBps": round(nbytes / ms * 1e-9, 2), "tile": list(s["tile"]), "took": took,
                          "layouts": [s["layout_a"], s["layout_b"]], "transposes": s["transposes"]}), flush=True)
        del A, B, C
        torch.cuda.empty_cache()
