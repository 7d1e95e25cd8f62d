"""BASELINE config 3 (abcd = dkca,bk, fp32, a = d = 128, b = k = 8) at c = 128 and c = 1024: achieved TB/s."""
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctc_amd  # noqa: E402


def timed(fn, iters=20, seconds=0.1):
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
for c in (128, 1024, 4096):
    ext = dict(a=128, b=8, c=c, d=128, k=8)
    A = torch.empty(128 * 8 * c * 128, dtype=torch.float32, device=dev)
    B = torch.empty(64, dtype=torch.float32, device=dev)
    C = torch.empty(128 * 8 * c * 128, dtype=torch.float32, device=dev)
    ctc_amd.fill(A, 1)
    ctc_amd.fill(B, 2)
    plan = ctc_amd.Plan("dkca,bk->abcd", ext, "f32")
    ms = timed(lambda: plan.execute(A, B, C))
    nbytes = (A.numel() + C.numel()) * 4
    print(json.dumps({"c": c, "kernel": plan.summary()["kernel"], "ms": round(ms, 4), "TBps": round(nbytes / ms * 1e-9, 2), "GB": round(nbytes * 1e-9, 2)}), flush=True)

# B and C contiguous along the same N mode: ab = ak,kb with a = k = 8
for n in (1 << 21, (1 << 21) + 4160, 1 << 24, (1 << 24) + 4160, 1 << 26, (1 << 26) + 4160):  # powers of two and not: row pitch against the channel interleave
    A = torch.empty(64, dtype=torch.float32, device=dev)
    B = torch.empty(8 * n, dtype=torch.float32, device=dev)
    C = torch.empty(8 * n, dtype=torch.float32, device=dev)
    ctc_amd.fill(A, 1)
    ctc_amd.fill(B, 2)
    plan = ctc_amd.Plan("ak,kb->ab", dict(a=8, k=8, b=n), "f32")
    ms = timed(lambda: plan.execute(A, B, C))
    nbytes = (B.numel() + C.numel()) * 4
    print(json.dumps({"n": n, "kernel": plan.summary()["kernel"], "ms": round(ms, 4), "TBps": round(nbytes / ms * 1e-9, 2), "GB": round(nbytes * 1e-9, 2)}), flush=True)
    del A, B, C
    torch.cuda.empty_cache()
