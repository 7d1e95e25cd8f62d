"""Workload for the HBM-traffic counter passes of the permutation kernels: a calibration copy of
known size, then the 128^4 fp32 reversal dcba <- abcd and the config-3 contraction abcd = dkca, bk
(c = 1024, beyond the Infinity Cache), three launches each.

  rocprofv3 --pmc FETCH_SIZE -- python3 tools/permute_traffic_probe.py   (and again with WRITE_SIZE)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctc_amd  # noqa: E402

dev = torch.device("cuda:0")
x = torch.empty((128, 128, 128, 128), dtype=torch.float32, device=dev)
y = torch.empty_like(x)
ctc_amd.fill(x.view(-1), 1)
for _ in range(2):
    y.copy_(x)
for _ in range(3):
    ctc_amd.permute(x, (3, 2, 1, 0), out=y)
ext = dict(a=128, b=8, c=1024, d=128, k=8)
A = torch.empty(128 * 8 * 1024 * 128, dtype=torch.float32, device=dev)
B = torch.empty(64, dtype=torch.float32, device=dev)
C = torch.empty(128 * 8 * 1024 * 128, dtype=torch.float32, device=dev)
ctc_amd.fill(A, 2)
ctc_amd.fill(B, 3)
plan = ctc_amd.Plan("dkca,bk->abcd", ext, "f32")
for _ in range(3):
    plan.execute(A, B, C)
torch.cuda.synchronize()
print("done")
