"""Diagnostic: the persistent kernel's slab time under one build variant (CTC_TC_LIB), d = 64 and 512."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctc_amd  # noqa: E402

n = 40
dev = torch.device("cuda:0")
ctc_amd.set_option("stream_mode", 2)
for d in (64, 128, 512):
    ext = dict(a=5, b=n, c=n, i=n, j=n, k=n, d=d)
    A = torch.empty(5 * d * n * n, dtype=torch.float64, device=dev)
    B = torch.empty(n * n * d * n, dtype=torch.float64, device=dev)
    C = torch.empty(5 * n ** 5, dtype=torch.float64, device=dev)
    ctc_amd.fill(A, 1)
    ctc_amd.fill(B, 2)
    plan = ctc_amd.Plan("adij,bcdk->abcijk", ext, "f64")
    ms = plan.time(A, B, C, warmup=3, iters=10)
    print(json.dumps({"lib": os.path.basename(os.environ.get("CTC_TC_LIB", "default")), "d": d, "ms": round(ms, 4),
                      "tflops": round(plan.flops / ms * 1e-9, 2)}), flush=True)
