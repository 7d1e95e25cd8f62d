"""Products with one or a few tiles of C and a very long K: the split along K is what fills the chip.
python tools/long_k.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctc_amd  # noqa: E402

dev = torch.device("cuda:0")
for dt, tdt in (("f64", torch.float64), ("f32", torch.float32), ("bf16", torch.bfloat16)):
    for (m, n, k) in [(48, 48, 1 << 20), (128, 128, 1 << 18), (256, 512, 1 << 16), (64, 64, 4096)]:
        A = torch.empty(m * k, dtype=tdt, device=dev)
        B = torch.empty(k * n, dtype=tdt, device=dev)
        C = torch.empty(m * n, dtype=tdt, device=dev)
        ctc_amd.fill(A, 1)
        ctc_amd.fill(B, 2)
        plan = ctc_amd.Plan("ak,bk->ab", dict(a=m, b=n, k=k), dt)
        for _ in range(3):
            plan.execute(A, B, C)
        ms = plan.time(A, B, C, warmup=2, iters=10)
        nbytes = (A.numel() + B.numel() + C.numel()) * A.element_size()
        print(json.dumps({"dtype": dt, "shape": [m, n, k], "split": plan.schedule()["split"], "ms": round(ms, 4),
                          "TFLOP/s": round(2.0 * m * n * k / ms * 1e-9, 2), "GB/s": round(nbytes / ms * 1e-6, 1)}), flush=True)
