"""Every permutation of a rank-4 fp32 tensor and a sample of rank-5 ones, sorted by achieved GB/s: the
slowest ones show which mode orders the tiling handles badly.   python tools/permute_sweep.py"""
import itertools
import json
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctc_amd  # noqa: E402

dev = torch.device("cuda:0")


def run(shape, perm, dtype=torch.float32):
    x = torch.empty(shape, dtype=dtype, device=dev)
    ctc_amd.fill(x.view(-1), 1)
    out = torch.empty([shape[p] for p in perm], dtype=dtype, device=dev)
    for _ in range(2):
        ctc_amd.permute(x, perm, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ctc_amd.permute(x, perm, out=out)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / 5
    return 2 * x.numel() * x.element_size() / ms * 1e-6


rows = []
shape4 = (96, 128, 80, 112)
for perm in itertools.permutations(range(4)):
    rows.append({"shape": shape4, "perm": perm, "GB/s": round(run(shape4, perm), 1)})
rng = random.Random(5)
shape5 = (24, 40, 32, 48, 56)
perms5 = list(itertools.permutations(range(5)))
for perm in rng.sample(perms5, 30):
    rows.append({"shape": shape5, "perm": perm, "GB/s": round(run(shape5, perm), 1)})
# extents that rule out 16-byte vectors on one side or both
for shape, perm in (((97, 128, 80, 111), (3, 2, 1, 0)), ((96, 128, 80, 111), (3, 2, 1, 0)), ((97, 128, 80, 112), (3, 2, 1, 0)),
                    ((1000, 3, 1000, 36), (3, 1, 2, 0)), ((4096, 4096), (1, 0)), ((128, 128, 128, 128), (3, 2, 1, 0)), ((128, 128, 128, 128), (0, 1, 3, 2)), ((16384, 1030), (1, 0)), ((8, 8, 8, 8, 8, 8, 8, 8), (7, 6, 5, 4, 3, 2, 1, 0))):
    rows.append({"shape": shape, "perm": perm, "GB/s": round(run(shape, perm), 1)})
rows.sort(key=lambda r: r["GB/s"])
for r in rows:
    print(json.dumps(r), flush=True)
