"""Per-launch device time of the headline contraction over a long run (clock ramp and settling)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctc_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 120
ext = dict(a=n, b=n, i=n, j=n, e=n, m=n)
dev = torch.device("cuda:0")
A = torch.empty(n ** 4, dtype=torch.float64, device=dev)
B = torch.empty_like(A)
C = torch.empty_like(A)
ctc_amd.fill(A, 1)
ctc_amd.fill(B, 2)
plan = ctc_amd.Plan("aeim,ebmj->abij", ext, "f64")
plan.prepare()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
ev[0].record()
for i in range(iters):
    plan.execute(A, B, C)
    ev[i + 1].record()
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(iters)]
tf = [plan.flops / m * 1e-9 for m in ms]
t = 0.0
for i in range(iters):
    t += ms[i]
    if i < 8 or i % 10 == 0 or i == iters - 1:
        print("launch %3d  t=%7.1f ms  %.2f TFLOP/s" % (i, t, tf[i]))
