"""Runs one contraction a few times so that rocprofv3 can attach counters to its kernel.

  python tools/prof_case.py six|ccsd96|ccsd64|ccsd96_f32|ccsd96_bf16|ccsd64_bf16|nt8k_bf16|nn8k_bf16|tn8k_bf16|tt8k_bf16 [iters]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctc_amd  # noqa: E402

CASES = {
    "six": ("adij,bcdk->abcijk", dict(a=5, b=40, c=40, i=40, j=40, k=40, d=64), "f64"),
    "ccsd96": ("aeim,ebmj->abij", dict(a=96, b=96, i=96, j=96, e=96, m=96), "f64"),
    "ccsd64": ("aeim,ebmj->abij", dict(a=64, b=64, i=64, j=64, e=64, m=64), "f64"),
    "ccsd96_f32": ("aeim,ebmj->abij", dict(a=96, b=96, i=96, j=96, e=96, m=96), "f32"),
    "ccsd96_bf16": ("aeim,ebmj->abij", dict(a=96, b=96, i=96, j=96, e=96, m=96), "bf16"),
    "ccsd64_bf16": ("aeim,ebmj->abij", dict(a=64, b=64, i=64, j=64, e=64, m=64), "bf16"),
    "nt8k_bf16": ("ak,bk->ab", dict(a=8192, b=8192, k=8192), "bf16"),
    "nn8k_bf16": ("ak,kb->ab", dict(a=8192, b=8192, k=8192), "bf16"),
    "tn8k_bf16": ("ka,bk->ab", dict(a=8192, b=8192, k=8192), "bf16"),
    "tt8k_bf16": ("ka,kb->ab", dict(a=8192, b=8192, k=8192), "bf16"),
}
spec, ext, dtype = CASES[sys.argv[1]]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
tdt = {"f64": torch.float64, "f32": torch.float32, "bf16": torch.bfloat16}[dtype]
ins, outm = spec.split("->")
la, lb = ins.split(",")


def vol(s):
    n = 1
    for c in s:
        n *= ext[c]
    return n


dev = torch.device("cuda:0")
A = torch.empty(vol(la), dtype=tdt, device=dev)
B = torch.empty(vol(lb), dtype=tdt, device=dev)
C = torch.empty(vol(outm), dtype=tdt, device=dev)
ctc_amd.fill(A, 0xC7C0A)
ctc_amd.fill(B, 0xC7C0B)
plan = ctc_amd.Plan(spec, ext, dtype)
plan.prepare()
for _ in range(iters):
    plan.execute(A, B, C)
torch.cuda.synchronize()
print("done", plan.summary()["tile"])
