import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctc_amd
dev = torch.device("cuda:0")
for k in (8192, 8192 + 32, 8192 + 64, 8192 + 96, 8192 + 256, 7936):
    for spec in ("ak,bk->ab", "ak,kb->ab"):
        ext = dict(a=8192, b=8192, k=k)
        plan = ctc_amd.Plan(spec, ext, "bf16")
        A = torch.empty(8192 * k, dtype=torch.bfloat16, device=dev)
        B = torch.empty(8192 * k, dtype=torch.bfloat16, device=dev)
        C = torch.empty(8192 * 8192, dtype=torch.bfloat16, device=dev)
        ctc_amd.fill(A, 1); ctc_amd.fill(B, 2)
        for _ in range(5): plan.execute(A, B, C)
        ms = plan.time(A, B, C, warmup=2, iters=10)
        print(json.dumps({"spec": spec, "k": k, "ms": round(ms, 4), "TFLOP/s": round(plan.flops / ms * 1e-9, 1)}), flush=True)
