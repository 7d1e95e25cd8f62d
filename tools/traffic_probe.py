"""Workload for the HBM-traffic counter passes: a calibration copy of known size (its kernel
reads and writes exactly `n` bytes with 16-byte accesses) and the headline contraction.

  python tools/traffic_probe.py [extent] [launches]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctc_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
# calibration: permute kernel used as a plain row copy of 1 GiB (rank-2, identity-like fast mode)
x = torch.empty(1 << 27, dtype=torch.float64, device=dev)
y = torch.empty_like(x)
ctc_amd.fill(x, 1)
for _ in range(2):
    y.copy_(x)
ext = dict(a=n, b=n, i=n, j=n, e=n, m=n)
A = torch.empty(n ** 4, dtype=torch.float64, device=dev)
B = torch.empty(n ** 4, dtype=torch.float64, device=dev)
C = torch.empty(n ** 4, dtype=torch.float64, device=dev)
ctc_amd.fill(A, 0xC7C0A)
ctc_amd.fill(B, 0xC7C0B)
plan = ctc_amd.Plan("aeim,ebmj->abij", ext, "f64")
plan.prepare()
for _ in range(launches):
    plan.execute(A, B, C)
torch.cuda.synchronize()
print("done")
