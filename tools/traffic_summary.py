"""Turns the two counter passes of tools/pmc_traffic.sh (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE,
rocpd sqlite output) into profiles/<name>: HBM-side bytes per launch of the headline contraction,
with the gfx950 read calibration taken from the copy kernel of known size in the same run.

  python tools/traffic_summary.py gpurun_out/traffic_FETCH_SIZE gpurun_out/traffic_WRITE_SIZE profiles/r01c_bench_traffic.json [extent]
"""
import glob
import json
import sqlite3
import sys


def counters(directory, counter):
    out = {"gemm": [], "fixup": [], "copy": [], "gemm_name": ""}
    for path in glob.glob(directory + "/**/*.db", recursive=True):
        c = sqlite3.connect(path)
        for name, cn, value in c.execute("select kernel_name, counter_name, value from counters_collection"):
            if cn != counter:
                continue
            if "splitk_fixup" in name or "sk_fixup" in name:
                out["fixup"].append(float(value))
            elif "gemm_kernel" in name or "gemm_dma_kernel" in name:
                out["gemm"].append(float(value))
                key = "gemm_dma_kernel" if "gemm_dma_kernel" in name else "gemm_kernel"
                out["gemm_name"] = name[name.index(key):].split("(")[0]
            elif "copyBuffer" in name:
                out["copy"].append(float(value))
    return out


def main():
    fetch_dir, write_dir, dest = sys.argv[1:4]
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 96
    f, w = counters(fetch_dir, "FETCH_SIZE"), counters(write_dir, "WRITE_SIZE")
    copy_bytes = float(1 << 30)  # tools/traffic_probe.py copies 1 GiB
    read_corr = copy_bytes / 1024.0 / (sum(f["copy"]) / len(f["copy"]))
    write_corr = copy_bytes / 1024.0 / (sum(w["copy"]) / len(w["copy"]))
    mean = lambda v: sum(v) / len(v) if v else 0.0  # noqa: E731
    read_raw = (mean(f["gemm"]) + mean(f["fixup"])) * 1024.0
    written = (mean(w["gemm"]) + mean(w["fixup"])) * 1024.0 * write_corr
    out = {
        "workload": "BASELINE configs[1], all extents %d (tools/traffic_probe.py under rocprofv3 --pmc, one counter per pass)" % n,
        "kernel": f["gemm_name"],
        "calibration": {"kernel": "__amd_rocclr_copyBuffer, 1 GiB read + 1 GiB written, 16-byte accesses",
                        "FETCH_SIZE_KB": f["copy"], "WRITE_SIZE_KB": w["copy"],
                        "read_correction": round(read_corr, 4), "write_correction": round(write_corr, 4)},
        "raw_KB": {"FETCH_SIZE_gemm": f["gemm"], "WRITE_SIZE_gemm": w["gemm"],
                   "FETCH_SIZE_fixup": f["fixup"], "WRITE_SIZE_fixup": w["fixup"]},
        "per_launch_bytes": {"read_corrected": int(read_raw * read_corr), "read_uncorrected": int(read_raw),
                             "written": int(written), "total_corrected": int(read_raw * read_corr + written)},
        "algorithmic_bytes": 3 * 8 * n ** 4,
        "note": "FETCH_SIZE on gfx950 reports half of the bytes of wide 16-byte-per-lane reads (the calibration copy "
                "shows it); the correction is applied to all of the contraction's reads, so the true value lies between "
                "the uncorrected and the corrected figure. The counters sit on the L2's memory side: Infinity Cache hits "
                "are included.",
    }
    with open(dest, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out["per_launch_bytes"]))


if __name__ == "__main__":
    main()
