"""The headline contraction with its operands starting and its result ending in HOST memory: what a caller pays
who hands over host buffers (the ctc_contract tool does, after reading its files).  Not bench.py's `value`, which
starts with the operands resident in HBM.   python tools/pcie_inclusive.py [extent]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctc_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
dev = torch.device("cuda:0")
ext = dict(a=n, b=n, i=n, j=n, e=n, m=n)
plan = ctc_amd.Plan("aeim,ebmj->abij", ext, "f64")
for pinned in (False, True):
    hA = torch.rand(n ** 4, dtype=torch.float64, pin_memory=pinned)
    hB = torch.rand(n ** 4, dtype=torch.float64, pin_memory=pinned)
    hC = torch.empty(n ** 4, dtype=torch.float64, pin_memory=pinned)
    dA = torch.empty(n ** 4, dtype=torch.float64, device=dev)
    dB = torch.empty_like(dA)
    dC = torch.empty_like(dA)
    plan.prepare()
    times = []
    for it in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dA.copy_(hA, non_blocking=True)
        dB.copy_(hB, non_blocking=True)
        plan.execute(dA, dB, dC)
        hC.copy_(dC, non_blocking=True)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    ms = min(times[1:]) * 1e3
    kernel_ms = plan.time(dA, dB, dC, warmup=1, iters=5)
    print(json.dumps({"extent": n, "host_memory": "pinned" if pinned else "pageable", "bytes_over_pcie": 3 * 8 * n ** 4,
                      "ms_host_to_host": round(ms, 2), "TFLOP/s_host_to_host": round(plan.flops / ms * 1e-9, 2),
                      "ms_kernel_alone": round(kernel_ms, 3), "TFLOP/s_resident": round(plan.flops / kernel_ms * 1e-9, 2),
                      "PCIe_GB/s_if_copies_were_all": round(3 * 8 * n ** 4 / (ms - kernel_ms) * 1e-6, 1)}), flush=True)
