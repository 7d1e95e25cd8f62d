"""bf16 products with large results at several contracted extents: the store path of the epilogue shows at
small K.   python tools/bf16_shapes.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctc_amd  # noqa: E402

dev = torch.device("cuda:0")
for spec in ("ak,bk->ab", "ak,kb->ab"):
    for (m, n, k) in [(16384, 16384, 64), (16384, 16384, 256), (16384, 16384, 1024), (16384, 16384, 4096), (8192, 8192, 8192),
                      (4096, 4096, 4096), (2048, 2048, 2048)]:
        A = torch.empty(m * k, dtype=torch.bfloat16, device=dev)
        B = torch.empty(k * n, dtype=torch.bfloat16, device=dev)
        C = torch.empty(m * n, dtype=torch.bfloat16, device=dev)
        ctc_amd.fill(A, 1)
        ctc_amd.fill(B, 2)
        plan = ctc_amd.Plan(spec, dict(a=m, b=n, k=k), "bf16")
        for _ in range(5):
            plan.execute(A, B, C)
        ms = plan.time(A, B, C, warmup=2, iters=10)
        print(json.dumps({"spec": spec, "shape": [m, n, k], "tile": plan.summary()["tile"][0], "ms": round(ms, 4),
                          "TFLOP/s": round(2.0 * m * n * k / ms * 1e-9, 1),
                          "C_write_GB/s": round(m * n * 2 / ms * 1e-6, 1)}), flush=True)
