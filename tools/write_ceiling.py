"""Write-only and read-only HBM rates of simple streaming kernels (bounds for store-dominated contractions)."""
import json
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctc_amd  # noqa: E402


def time_ms(fn, warmup=2, iters=8):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


for gib in (1, 4, 16):
    n = gib * (1 << 30) // 8
    x = torch.empty(n, dtype=torch.float64, device="cuda:0")
    y = torch.empty(n, dtype=torch.float64, device="cuda:0")
    out = {"GiB": gib}
    ms = time_ms(lambda: x.zero_())
    out["torch_zero_GB/s"] = round(n * 8 / ms * 1e-6, 1)
    ms = time_ms(lambda: x.fill_(1.5))
    out["torch_fill_GB/s"] = round(n * 8 / ms * 1e-6, 1)
    ms = time_ms(lambda: ctc_amd.fill(x, 7))
    out["ctc_fill_GB/s"] = round(n * 8 / ms * 1e-6, 1)
    ms = time_ms(lambda: x.sum())
    out["torch_sum_read_GB/s"] = round(n * 8 / ms * 1e-6, 1)
    ms = time_ms(lambda: y.copy_(x))
    out["copy_rw_GB/s"] = round(2 * n * 8 / ms * 1e-6, 1)
    print(json.dumps(out))
    del x, y
