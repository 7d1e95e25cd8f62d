"""A few permutations timed several times each (median of 5 runs of 10 launches)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctc_amd  # noqa: E402
dev = torch.device("cuda:0")
CASES = [((128, 128, 128, 128), (3, 2, 1, 0)), ((128, 128, 128, 128), (0, 1, 3, 2)), ((4096, 4096), (1, 0)),
         ((96, 128, 80, 112), (0, 1, 3, 2)), ((96, 128, 80, 112), (3, 2, 1, 0)), ((97, 128, 80, 111), (3, 2, 1, 0)),
         ((1000, 3, 1000, 36), (3, 1, 2, 0))]
for shape, perm in CASES:
    x = torch.empty(shape, dtype=torch.float32, device=dev)
    ctc_amd.fill(x.view(-1), 1)
    out = torch.empty([shape[p] for p in perm], dtype=torch.float32, device=dev)
    for _ in range(3):
        ctc_amd.permute(x, perm, out=out)
    rates = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ctc_amd.permute(x, perm, out=out)
        e1.record(); e1.synchronize()
        rates.append(2 * x.numel() * 4 / (e0.elapsed_time(e1) / 10) * 1e-6)
    rates.sort()
    print(json.dumps({"shape": shape, "perm": perm, "GB/s": [round(r) for r in rates]}), flush=True)
