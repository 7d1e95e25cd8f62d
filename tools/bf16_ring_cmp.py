"""bf16 products and the headline contraction with K-major and outer-major operands; run once as is and once with
CTC_TC_NO_BF16_RING_OUTER=1 (outer-major operands on the two-buffer kernel) to compare.
   python tools/bf16_ring_cmp.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctc_amd  # noqa: E402

dev = torch.device("cuda:0")
cases = [(s, dict(a=n, b=n, k=n)) for n in (8192, 4096) for s in ("ak,bk->ab", "ak,kb->ab", "ka,bk->ab", "ka,kb->ab")]
cases += [("aeim,ebmj->abij", {m: n for m in "abeijm"}) for n in (64, 72, 80, 88, 96)]
for spec, ext in cases:
    plan = ctc_amd.Plan(spec, ext, "bf16")
    s = plan.summary()
    a_modes, rest = spec.split(",")
    b_modes, c_modes = rest.split("->")
    count = lambda modes: int(torch.tensor([ext[m] for m in modes]).prod())
    A = torch.empty(count(a_modes), dtype=torch.bfloat16, device=dev)
    B = torch.empty(count(b_modes), dtype=torch.bfloat16, device=dev)
    C = torch.empty(count(c_modes), dtype=torch.bfloat16, device=dev)
    ctc_amd.fill(A, 1)
    ctc_amd.fill(B, 2)
    before = ctc_amd.get_option("launches_bf16_ring")
    for _ in range(5):
        plan.execute(A, B, C)
    ring = ctc_amd.get_option("launches_bf16_ring") > before
    ms = plan.time(A, B, C, warmup=2, iters=10)
    print(json.dumps({"spec": spec, "extent": max(ext.values()), "layouts": [s["layout_a"], s["layout_b"]], "tile": s["tile"][0],
                      "ring": ring, "ms": round(ms, 4), "TFLOP/s": round(plan.flops / ms * 1e-9, 1)}), flush=True)
