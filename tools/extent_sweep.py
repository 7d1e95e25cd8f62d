"""CCSD-style contraction abij = aeim,ebmj at a range of extents: where the tile choice and the
split along K hold the fp64 / fp32 rate and where they do not.   python tools/extent_sweep.py [f64|f32]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctc_amd  # noqa: E402

dt = sys.argv[1] if len(sys.argv) > 1 else "f64"
tdt = {"f64": torch.float64, "f32": torch.float32}[dt]
dev = torch.device("cuda:0")
sizes = [int(x) for x in sys.argv[2:]] or [16, 24, 32, 40, 48, 56, 64, 72, 80, 88, 96]
for n in sizes:
    ext = dict(a=n, b=n, i=n, j=n, e=n, m=n)
    A = torch.empty(n ** 4, dtype=tdt, device=dev)
    B = torch.empty(n ** 4, dtype=tdt, device=dev)
    C = torch.empty(n ** 4, dtype=tdt, device=dev)
    ctc_amd.fill(A, 1)
    ctc_amd.fill(B, 2)
    plan = ctc_amd.Plan("aeim,ebmj->abij", ext, dt)
    for _ in range(20 if n < 64 else 3):
        plan.execute(A, B, C)
    s = plan.summary()
    row = {"extent": n, "gemm": n * n, "tile": s["tile"]}
    # the last round of tiles cut into equal K chunks only, then the schedule chosen by cost (which may deal the
    # last one to two rounds out in stream-K shares), same process, same buffers
    for mode, label in ((1, "equal_chunks"), (2, "shares_forced"), (0, "by_cost")):
        ctc_amd.set_option("stream_k", mode)
        sch = plan.schedule() if s["kernel"] == "mfma_gemm" else {}
        ms = plan.time(A, B, C, warmup=3, iters=20 if n < 64 else 5)
        row[label] = {"ms": round(ms, 4), "TFLOP/s": round(plan.flops / ms * 1e-9, 2), "whole_tiles": sch.get("whole_tiles"),
                      "tail_tiles": sch.get("tail_tiles"), "split": sch.get("split"), "shares": sch.get("stream_k_shares")}
    ctc_amd.set_option("stream_k", 0)
    row["ms"], row["TFLOP/s"] = row["by_cost"]["ms"], row["by_cost"]["TFLOP/s"]
    print(json.dumps(row), flush=True)
