"""Batched products of middling size (64^3 ... 256^3 under hundreds to thousands of batch entries) are streams:
a few flops per byte.  Rate and bytes per second per case; run under CTC_TC_C_STREAM=0 / 1 to compare plain stores
of C with the streaming hint.  One JSON line per case."""
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
CASES = ((1024, 128, 128, 128), (4096, 128, 128, 128), (4096, 64, 64, 64), (16384, 64, 64, 64), (512, 256, 256, 256), (2048, 256, 256, 64),
         (16, 1024, 1024, 256), (64, 1024, 1024, 128), (2048, 96, 96, 96), (1024, 200, 200, 40))
if os.environ.get("BSTORE_SWEEP"):
    # C from 32 MiB to 512 MiB at fixed product sizes
    CASES = tuple((z * f, e, e, e) for e, z in ((128, 256), (64, 1024), (96, 512), (32, 8192)) for f in (1, 2, 3, 4, 6, 8, 16)) + \
        ((1, 4096, 4096, 4096), (1, 4096, 4096, 512), (4, 2048, 2048, 256), (1, 3072, 3072, 128))
for dt, tdt in (("f64", torch.float64), ("f32", torch.float32)):
    for z, m, n, k in CASES:
        A = torch.empty(z * m * k, dtype=tdt, device=dev)
        B = torch.empty(z * k * n, dtype=tdt, device=dev)
        C = torch.empty(z * m * n, dtype=tdt, device=dev)
        ctc_amd.fill(A, 1)
        ctc_amd.fill(B, 2)
        plan = ctc_amd.Plan("zak,zkb->zab", dict(z=z, a=m, b=n, k=k), dt)
        plan.prepare()
        ms = timed(lambda: plan.execute(A, B, C))
        nbytes = (A.numel() + B.numel() + C.numel()) * A.element_size()
        s = plan.summary()
        print(json.dumps({"c_stream_env": os.environ.get("CTC_TC_C_STREAM"), "dtype": dt, "batch": z, "m": m, "n": n, "k": k, "ms": round(ms, 4),
                          "tflops": round(2.0 * z * m * n * k / ms * 1e-9, 1), "TBps": round(nbytes / ms * 1e-9, 2),
                          "C_MiB": C.numel() * C.element_size() >> 20, "tile": list(s["tile"])}), flush=True)
        del A, B, C
        torch.cuda.empty_cache()
