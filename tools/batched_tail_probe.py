"""Batched products whose tile count is no multiple of the chip's workgroup slots: rate per shape.  Run under
CTC_TC_NO_FLAT_BATCH=1 (the batch as a grid dimension, no cut of the last round) and CTC_TC_NO_BATCHED_CHUNKS=1
(the planner's tile choice without cuts for batched products) to compare.  One JSON line per case."""
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctc_amd  # noqa: E402

PEAK = {"f64": 78.6, "f32": 157.3}


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
CASES = ((3, 3000, 3000, 3000), (5, 2000, 2000, 2000), (7, 1500, 1500, 4096), (3, 640, 512, 4096), (10, 1024, 1024, 1024), (12, 1024, 1024, 1024),
         (20, 768, 768, 2048), (6, 512, 640, 8192), (2, 4096, 4096, 1024), (9, 1280, 1152, 512), (33, 512, 512, 512), (3, 1024, 1024, 16384),
         (4, 2048, 2048, 2048), (64, 512, 512, 512), (100, 384, 384, 1024), (17, 896, 896, 896))
if os.environ.get("BATCHED_SMALL"):
    # batches of products small enough for the 64 x 64 tile (or close to the boundary between the tiles)
    CASES = ((8, 300, 300, 300), (16, 256, 256, 4096), (6, 200, 180, 2048), (32, 128, 128, 2048), (5, 640, 576, 1024), (100, 100, 100, 1000),
             (20, 384, 384, 384), (3, 500, 500, 8000), (2, 1000, 1000, 1000), (12, 192, 192, 16384), (50, 64, 64, 4096), (4, 700, 900, 1100))
if os.environ.get("PROBE_CASES"):  # "batch,m,n,k;batch,m,n,k;..."
    CASES = tuple(tuple(int(v) for v in c.split(",")) for c in os.environ["PROBE_CASES"].split(";"))
TYPES = (("f64", torch.float64), ("f32", torch.float32))
if os.environ.get("BATCHED_BF16"):
    TYPES = (("bf16", torch.bfloat16),)
    PEAK["bf16"] = 2500.0
    CASES = ((3, 3000, 3000, 3000), (5, 2048, 2048, 2048), (10, 1024, 1024, 1024), (12, 1024, 1024, 4096), (20, 768, 768, 2048), (6, 512, 640, 8192),
             (33, 512, 512, 512), (3, 1024, 1024, 16384), (4, 2048, 2048, 2048), (64, 512, 512, 512), (7, 1536, 1536, 4096), (2, 4096, 4096, 4096), (50, 256, 256, 4096))
for dt, tdt in TYPES:
    for z, m, n, k in CASES:
        A = torch.empty(z * m * k, dtype=tdt, device=dev)
        B = torch.empty(z * k * n, dtype=tdt, device=dev)
        C = torch.empty(z * m * n, dtype=tdt, device=dev)
        if dt == "bf16":
            A.copy_(torch.rand(A.numel(), dtype=torch.float32, device=dev) - 0.5)
            B.copy_(torch.rand(B.numel(), dtype=torch.float32, device=dev) - 0.5)
        else:
            ctc_amd.fill(A, 1)
            ctc_amd.fill(B, 2)
        plan = ctc_amd.Plan("zak,zkb->zab", dict(z=z, a=m, b=n, k=k), dt) if z > 1 else ctc_amd.Plan("ak,kb->ab", dict(a=m, b=n, k=k), dt)
        plan.prepare()
        ms = timed(lambda: plan.execute(A, B, C))
        s, sch = plan.summary(), plan.schedule()
        tf = 2.0 * z * m * n * k / ms * 1e-9
        print(json.dumps({"dtype": dt, "batch": z, "m": m, "n": n, "k": k, "ms": round(ms, 4), "tflops": round(tf, 1), "frac_mfma": round(tf / PEAK[dt], 3),
                          "tile": list(s["tile"])[:2], "tiles": sch["tile_rows"] * sch["tile_cols"] * z, "whole": sch["whole_tiles"], "cut": sch["tail_tiles"],
                          "split": sch["split"], "shares": sch["stream_k_shares"], "one_list": sch["batch_entries_in_one_tile_list"]}), flush=True)
        del A, B, C
        torch.cuda.empty_cache()
