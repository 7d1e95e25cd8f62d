"""Seeded random products (single and batched, vector-friendly and odd sizes, fp64 and fp32) timed with the library as
it is; run twice - once with the A/B switches that bring the earlier scheduler and tile choice back - and compare.
  python tools/random_shapes_ab.py [count] [seed]   ->  one JSON line per shape"""
import json
import os
import statistics
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctc_amd  # noqa: E402


def timed(fn, iters=6, seconds=0.05):
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


count = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 20250312)
dev = torch.device("cuda:0")
specs = ("ak,kb->ab", "ak,bk->ab", "ka,kb->ab", "ka,bk->ab")
for i in range(count):
    dt = "f32" if i % 2 else "f64"
    tdt = torch.float32 if i % 2 else torch.float64
    z = 1 if rng.rand() < 0.5 else int(rng.choice([2, 3, 4, 6, 8, 12, 16, 32, 64]))
    lo, hi = (64, 3000) if z == 1 else (32, 1500)
    m, n = int(np.exp(rng.uniform(np.log(lo), np.log(hi)))), int(np.exp(rng.uniform(np.log(lo), np.log(hi))))
    k = int(np.exp(rng.uniform(np.log(64), np.log(12000))))
    if rng.rand() < 0.6:  # vector-friendly sizes
        m, n, k = max(16, m // 16 * 16), max(16, n // 16 * 16), max(16, k // 16 * 16)
    if z * (m * k + k * n + m * n) * (8 if dt == "f64" else 4) > 6e9:
        continue
    spec = specs[int(rng.randint(4))]
    ext = dict(a=m, b=n, k=k)
    if z > 1:
        ins, outm = spec.split("->")
        spec = ",".join("z" + t for t in ins.split(",")) + "->z" + outm
        ext["z"] = z
    la, lb = spec.split("->")[0].split(",")
    num = lambda t: int(np.prod([ext[c] for c in t]))  # noqa: E731
    A = torch.empty(num(la), dtype=tdt, device=dev)
    B = torch.empty(num(lb), dtype=tdt, device=dev)
    C = torch.empty(z * m * n, dtype=tdt, device=dev)
    ctc_amd.fill(A, 1)
    ctc_amd.fill(B, 2)
    plan = ctc_amd.Plan(spec, ext, dt)
    plan.prepare()
    ms = timed(lambda: plan.execute(A, B, C))
    s = plan.summary()
    print(json.dumps({"i": i, "dtype": dt, "spec": spec, "z": z, "m": m, "n": n, "k": k, "tile": list(s["tile"])[:2], "transposes": s["transposes"],
                      "layouts": [s["layout_a"], s["layout_b"]], "ms": round(ms, 4), "tflops": round(plan.flops / ms * 1e-9, 2)}), flush=True)
    del A, B, C
    torch.cuda.empty_cache()
