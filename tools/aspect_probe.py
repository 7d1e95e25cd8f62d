"""Plain products ab = ak,kb (and ak,bk) at extreme aspect ratios, fp64 / fp32: rate, kernel, schedule.  One JSON
line per case; the fraction of the MFMA peak and the bytes the operands move say which roof applies."""
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctc_amd  # noqa: E402

PEAK = {"f64": 78.6, "f32": 157.3}
SHAPES = [(256, 256, 1 << 20), (512, 512, 1 << 18), (128, 4096, 1 << 16), (4096, 128, 1 << 16), (32, 32, 1 << 22), (64, 8192, 8192),
          (8192, 64, 8192), (1024, 1024, 65536), (16384, 16384, 256), (16384, 16384, 512), (2048, 2048, 2048), (3000, 5000, 7000),
          (100, 100000, 1000), (100000, 100, 1000), (17, 100000, 4096)]


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
for dt, tdt in (("f64", torch.float64), ("f32", torch.float32)):
    for spec in ("ak,kb->ab", "ak,bk->ab"):
        for m, n, k in SHAPES:
            A = torch.empty(m * k, dtype=tdt, device=dev)
            B = torch.empty(k * n, dtype=tdt, device=dev)
            C = torch.empty(m * n, dtype=tdt, device=dev)
            ctc_amd.fill(A, 1)
            ctc_amd.fill(B, 2)
            plan = ctc_amd.Plan(spec, dict(a=m, b=n, k=k), dt)
            s = plan.summary()
            before = [ctc_amd.get_option(x) for x in names]
            ms = timed(lambda: plan.execute(A, B, C))
            took = [x[9:] for x, b in zip(names, before) if ctc_amd.get_option(x) > b]
            tf = plan.flops / ms * 1e-9
            nbytes = (A.numel() + B.numel() + C.numel()) * A.element_size()
            sch = plan.schedule() if s["kernel"] == "mfma_gemm" else {}
            print(json.dumps({"dtype": dt, "spec": spec, "M": m, "N": n, "K": k, "ms": round(ms, 4), "tflops": round(tf, 1), "frac_mfma": round(tf / PEAK[dt], 3),
                              "TBps": round(nbytes / ms * 1e-9, 2), "kernel": s["kernel"], "tile": list(s["tile"]), "took": took,
                              "split": sch.get("split"), "shares": sch.get("stream_k_shares"), "tail": sch.get("tail_tiles")}), flush=True)
            del A, B, C
            torch.cuda.empty_cache()
