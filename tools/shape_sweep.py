"""Plain products ab = ak,kb over rectangular shapes (fp64 by default): a table to spot shapes where the
tile choice or the split along K leaves the chip idle.   python tools/shape_sweep.py [f64|f32|bf16]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctc_amd  # noqa: E402

dt = sys.argv[1] if len(sys.argv) > 1 else "f64"
tdt = {"f64": torch.float64, "f32": torch.float32, "bf16": torch.bfloat16}[dt]
dev = torch.device("cuda:0")
sizes = [64, 256, 1024, 4096, 16384]
for m in sizes:
    for n in sizes:
        for k in sizes:
            flops = 2.0 * m * n * k
            if flops < 2e9 or flops > 3e12 or max(m * k, k * n, m * n) > 2 ** 28:
                continue
            A = torch.empty(m * k, dtype=tdt, device=dev)
            B = torch.empty(k * n, dtype=tdt, device=dev)
            C = torch.empty(m * n, dtype=tdt, device=dev)
            ctc_amd.fill(A, 1)
            ctc_amd.fill(B, 2)
            plan = ctc_amd.Plan("ak,kb->ab", dict(a=m, b=n, k=k), dt)
            iters = 20 if flops < 1e11 else 5
            for _ in range(iters):
                plan.execute(A, B, C)
            ms = plan.time(A, B, C, warmup=2, iters=iters)
            s = plan.summary()
            sch = plan.schedule() if s["kernel"] == "mfma_gemm" else {}
            print(json.dumps({"M": m, "N": n, "K": k, "kernel": s["kernel"], "tile": s["tile"][:2], "swapped": s["swapped"],
                              "split": sch.get("split"), "tail": sch.get("tail_tiles"), "ms": round(ms, 4),
                              "TFLOP/s": round(flops / ms * 1e-9, 2)}), flush=True)
            del A, B, C
