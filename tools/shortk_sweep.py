"""Time of the config-4 slab contraction abcijk = adij,bcdk (fp64, |C| 4.1 GB) as a function of the
contracted extent d: separates the store-bound part (d -> 0) from the MFMA part (slope)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctc_amd  # noqa: E402

dev = torch.device("cuda:0")
for d in [int(x) for x in (sys.argv[1:] or "4 16 32 64 128 256".split())]:
    e = dict(a=5, b=40, c=40, i=40, j=40, k=40, d=d)
    A = torch.empty(5 * d * 1600, dtype=torch.float64, device=dev)
    B = torch.empty(1600 * d * 40, dtype=torch.float64, device=dev)
    C = torch.empty(5 * 40 ** 5, dtype=torch.float64, device=dev)
    ctc_amd.fill(A, 1)
    ctc_amd.fill(B, 2)
    plan = ctc_amd.Plan("adij,bcdk->abcijk", e, "f64")
    ms = plan.time(A, B, C, warmup=3, iters=10)
    print(json.dumps({"d": d, "ms": round(ms, 4), "TFLOP/s": round(plan.flops / ms * 1e-9, 2),
                      "C_write_GB/s": round(C.numel() * 8 / ms * 1e-6, 1),
                      "mfma_ms_at_78.6": round(plan.flops / 78.6e9, 4)}))
