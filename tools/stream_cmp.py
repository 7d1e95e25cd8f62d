"""Persistent deferred-store kernel vs the one-tile-per-workgroup kernel on short-K contractions.
One JSON line per (case, kernel)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import ctc_amd

PEAK = {"f64": 78.6, "f32": 157.3}
TDT = {"f64": torch.float64, "f32": torch.float32}


def run(name, spec, ext, dtype, iters=10, alpha=1.0):
    ins, outm = spec.split("->")
    la, lb = ins.split(",")
    num = lambda s: int(torch.tensor([ext[c] for c in s]).prod())  # noqa: E731
    dev = torch.device("cuda:0")
    A = torch.empty(num(la), dtype=TDT[dtype], device=dev)
    B = torch.empty(num(lb), dtype=TDT[dtype], device=dev)
    C = torch.empty(num(outm), dtype=TDT[dtype], device=dev)
    ctc_amd.fill(A, 0xC7C0A)
    ctc_amd.fill(B, 0xC7C0B)
    plan = ctc_amd.Plan(spec, ext, dtype)
    plan.prepare()
    names = ("launches_classic", "launches_dma_gemm", "launches_stream", "launches_stream_f64_dma")
    rates = {}
    for mode, label in ((1, "classic"), (2, "persistent"), (0, "automatic")):
        ctc_amd.set_option("stream_mode", mode)
        before = [ctc_amd.get_option(n) for n in names]
        ms = plan.time(A, B, C, warmup=3, iters=iters, alpha=alpha)
        took = [n[9:] for n, b in zip(names, before) if ctc_amd.get_option(n) > b]
        tf = plan.flops / ms * 1e-9
        rates[label] = tf
        line = {"case": name, "dtype": dtype, "alpha": alpha, "kernel": label, "took": took, "ms": round(ms, 4), "tflops": round(tf, 2),
                "frac_of_mfma_peak": round(tf / PEAK[dtype], 3),
                "c_write_TBps": round(C.numel() * C.element_size() / ms * 1e-9, 2)}
        if label == "automatic":
            line["vs_better_of_the_two"] = round(tf / max(rates["classic"], rates["persistent"]), 4)
        print(json.dumps(line), flush=True)
    ctc_amd.set_option("stream_mode", 0)
    del A, B, C
    torch.cuda.empty_cache()


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    n = 40
    if which == "routing":
        # the automatic choice against both kernels over the contracted extent and alpha
        for d in (40, 48, 64, 80, 96, 112, 128):
            for alpha in (1.0, 1.25):
                run("config4_slab_a5_d%d" % d, "adij,bcdk->abcijk", dict(a=5, b=n, c=n, i=n, j=n, k=n, d=d), "f64", alpha=alpha)
        sys.exit(0)
    run("config4_slab_a5_d64", "adij,bcdk->abcijk", dict(a=5, b=n, c=n, i=n, j=n, k=n, d=64), "f64")
    if which == "all":
        run("config4_whole_a40_d64", "adij,bcdk->abcijk", dict(a=40, b=n, c=n, i=n, j=n, k=n, d=64), "f64", iters=4)
        for d in (40, 128, 256, 512):
            run("config4_slab_a5_d%d" % d, "adij,bcdk->abcijk", dict(a=5, b=n, c=n, i=n, j=n, k=n, d=d), "f64")
        run("gemm_8000x64000x64", "ak,kb->ab", dict(a=8000, b=64000, k=64), "f64")
        run("gemm_nt_8192x8192x128", "ak,bk->ab", dict(a=8192, b=8192, k=128), "f64")
        run("gemm_tn_8192x8192x128", "ka,kb->ab", dict(a=8192, b=8192, k=128), "f64")
        run("config4_slab_f32", "adij,bcdk->abcijk", dict(a=5, b=n, c=n, i=n, j=n, k=n, d=64), "f32")
        run("gemm_f32_16384x16384x128", "ak,kb->ab", dict(a=16384, b=16384, k=128), "f32")
