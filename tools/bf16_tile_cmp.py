"""The 128 and the 256 tile of the bf16 kernel (CTC_TC_FORCE_TILE) on mid-sized products: where the
planner's threshold (192 tiles of 256) should sit.   python tools/bf16_tile_cmp.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctc_amd  # noqa: E402

dev = torch.device("cuda:0")
shapes = [(2048, 2048, 2048), (2048, 2048, 8192), (2048, 4096, 4096), (2048, 4096, 16384), (3072, 3072, 4096), (1024, 4096, 16384),
          (1024, 1024, 16384), (4096, 2048, 2048), (2560, 2560, 8192), (3584, 3584, 3584)]
for (m, n, k) in shapes:
    A = torch.empty(m * k, dtype=torch.bfloat16, device=dev)
    B = torch.empty(k * n, dtype=torch.bfloat16, device=dev)
    C = torch.empty(m * n, dtype=torch.bfloat16, device=dev)
    ctc_amd.fill(A, 1)
    ctc_amd.fill(B, 2)
    row = {"shape": [m, n, k], "tiles256": ((m + 255) // 256) * ((n + 255) // 256)}
    for force in ("128", "256"):
        os.environ["CTC_TC_FORCE_TILE"] = force
        plan = ctc_amd.Plan("ak,bk->ab", dict(a=m, b=n, k=k), "bf16")
        for _ in range(5):
            plan.execute(A, B, C)
        ms = plan.time(A, B, C, warmup=2, iters=10)
        row["t" + force] = round(2.0 * m * n * k / ms * 1e-9, 1)
        row["split" + force] = plan.schedule()["split"]
    del os.environ["CTC_TC_FORCE_TILE"]
    row["default"] = ctc_amd.Plan("ak,bk->ab", dict(a=m, b=n, k=k), "bf16").summary()["tile"][0]
    print(json.dumps(row), flush=True)
