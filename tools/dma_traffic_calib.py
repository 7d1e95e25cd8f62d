"""Calibration of FETCH_SIZE for the LDS-DMA GEMM kernel's two access patterns: one 128 x 128 tile of C with
a very long K, cut along K over the whole chip, so that every operand byte crosses the L2's memory side
exactly once (operands far larger than every cache).  Three layout pairs:
    ak,bk  both operands K-major  (64-byte row pieces per K-tile)
    ka,kb  both outer-major       (whole 1 KB lines)
    ak,kb  one of each
Run under  rocprofv3 --pmc FETCH_SIZE ;  python tools/dma_traffic_calib.py summary <dir>  prints bytes
counted per byte read for each pair."""
import glob
import json
import os
import sqlite3
import sys

K = 1 << 20  # 128 x 2^20 fp64 = 1 GiB per operand


def run():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import ctc_amd
    dev = torch.device("cuda:0")
    A = torch.empty(128 * K, dtype=torch.float64, device=dev)
    B = torch.empty(128 * K, dtype=torch.float64, device=dev)
    C = torch.empty(128 * 128, dtype=torch.float64, device=dev)
    ctc_amd.fill(A, 1)
    ctc_amd.fill(B, 2)
    for spec in ("ak,bk->ab", "ka,kb->ab", "ak,kb->ab"):
        plan = ctc_amd.Plan(spec, dict(a=128, b=128, k=K), "f64")
        before = ctc_amd.get_option("launches_dma_gemm")
        for _ in range(3):
            plan.execute(A, B, C)
        torch.cuda.synchronize()
        print(spec, plan.summary()["tile"], "dma launches", ctc_amd.get_option("launches_dma_gemm") - before, flush=True)


def summary(directory):
    rows = {}
    for path in glob.glob(directory + "/**/*.db", recursive=True):
        c = sqlite3.connect(path)
        for name, cn, value in c.execute("select kernel_name, counter_name, value from counters_collection"):
            if cn == "FETCH_SIZE" and "gemm_dma_kernel" in name:
                key = name[name.index("gemm_dma_kernel"):].split("(")[0]
                rows.setdefault(key, []).append(float(value) * 1024.0)
    true_bytes = 2.0 * 128 * K * 8
    for key, v in sorted(rows.items()):
        mean = sum(v) / len(v)
        print(json.dumps({"kernel": key, "launches": len(v), "FETCH_SIZE_bytes": int(mean), "bytes_read": int(true_bytes),
                          "counted_per_byte": round(mean / true_bytes, 4)}))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "summary":
        summary(sys.argv[2])
    else:
        run()
