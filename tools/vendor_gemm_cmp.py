"""This library beside the vendor stack on the same GPU: torch.matmul (rocBLAS / hipBLASLt underneath) on the
packed GEMM views, torch.einsum (transposes + batched GEMM + transpose) on the contractions themselves.
A measurement only: nothing in the library calls either.  One JSON line per case.

  python tools/vendor_gemm_cmp.py [f64] [f32] [bf16] [config4] [odd]
"""
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import ctc_amd  # noqa: E402

PEAK = {"f64": 78.6, "f32": 157.3, "bf16": 2500.0}
TDT = {"f64": torch.float64, "f32": torch.float32, "bf16": torch.bfloat16}
DEV = torch.device("cuda:0")


def timed(fn, iters=10, seconds=0.15):
    """Median of per-call event times after a fixed stretch of untimed load."""
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        fn()
        torch.cuda.synchronize()
    each = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        each.append(e0.elapsed_time(e1))
    return statistics.median(each)


def operand(numel, dtype, seed):
    x = torch.empty(numel, dtype=TDT[dtype], device=DEV)
    if dtype == "bf16":
        f = torch.empty(numel, dtype=torch.float32, device=DEV)
        ctc_amd.fill(f, seed)
        x.copy_(f)
    else:
        ctc_amd.fill(x, seed)
    return x


def line(case, dtype, who, ms, flops, extra=None):
    tf = flops / ms * 1e-9
    out = {"case": case, "dtype": dtype, "who": who, "ms": round(ms, 4), "tflops": round(tf, 2),
           "frac_of_mfma_peak": round(tf / PEAK[dtype], 4)}
    out.update(extra or {})
    print(json.dumps(out), flush=True)
    return tf


def worst_diff(got, want):
    return float((got.double() - want.double()).abs().max() / want.double().abs().max())


def gemm(dtype, m, n, k, b_kmajor, iters=10):
    """C[m,n] = A[m,k] B: B stored [k][n] (`ak,kb`) or [n][k] (`ak,bk`)."""
    spec = "ak,bk->ab" if b_kmajor else "ak,kb->ab"
    case = "gemm %dx%dx%d %s" % (m, n, k, spec)
    A, B = operand(m * k, dtype, 0xC7C0A), operand(k * n, dtype, 0xC7C0B)
    C = torch.empty(m * n, dtype=TDT[dtype], device=DEV)
    plan = ctc_amd.Plan(spec, dict(a=m, b=n, k=k), dtype)
    plan.prepare()
    flops = 2.0 * m * n * k
    ours = line(case, dtype, "libctc_tc", timed(lambda: plan.execute(A, B, C), iters), flops)
    A2 = A.view(m, k)
    B2 = B.view(n, k).t() if b_kmajor else B.view(k, n)
    out = torch.empty(m, n, dtype=TDT[dtype], device=DEV)
    vendor = line(case, dtype, "torch.matmul", timed(lambda: torch.matmul(A2, B2, out=out), iters), flops,
                  {"max_rel_diff_of_results": worst_diff(C.view(m, n), out)})
    print(json.dumps({"case": case, "dtype": dtype, "libctc_tc_over_torch": round(ours / vendor, 3)}), flush=True)


def contraction(dtype, spec, ext, iters=10):
    ins, outm = spec.split("->")
    la, lb = ins.split(",")
    num = lambda s: int(torch.tensor([ext[c] for c in s]).prod())  # noqa: E731
    shape = lambda s: [ext[c] for c in s]  # noqa: E731
    case = "%s %s" % (spec, ",".join("%s=%d" % kv for kv in ext.items()))
    A, B = operand(num(la), dtype, 0xC7C0A), operand(num(lb), dtype, 0xC7C0B)
    C = torch.empty(num(outm), dtype=TDT[dtype], device=DEV)
    plan = ctc_amd.Plan(spec, ext, dtype)
    plan.prepare()
    ours = line(case, dtype, "libctc_tc", timed(lambda: plan.execute(A, B, C), iters), plan.flops)
    At, Bt = A.view(shape(la)), B.view(shape(lb))
    res = [None]

    def vendor_fn():
        res[0] = torch.einsum(spec, At, Bt)
    # torch.einsum hands back a strided VIEW of its GEMM result (modes in GEMM order); the contraction asks for C
    # in the order the spec names, so the second line materialises it there, as this library's result is
    view = line(case, dtype, "torch.einsum (result left as a strided view of the GEMM's)", timed(vendor_fn, iters), plan.flops,
                {"max_rel_diff_of_results": worst_diff(C.view(shape(outm)), res[0]), "result_is_contiguous": res[0].is_contiguous()})

    def vendor_packed():
        res[0] = torch.einsum(spec, At, Bt).contiguous()
    vendor = line(case, dtype, "torch.einsum(...).contiguous() (result in the spec's order)", timed(vendor_packed, iters), plan.flops)
    print(json.dumps({"case": case, "dtype": dtype, "libctc_tc_over_torch": round(ours / vendor, 3),
                      "libctc_tc_over_torch_view": round(ours / view, 3)}), flush=True)
    del res, A, B, C
    torch.cuda.empty_cache()


if __name__ == "__main__":
    which = sys.argv[1:] or ["f64", "f32", "bf16", "config4", "odd"]
    torch.cuda.set_device(0)
    print(json.dumps({"torch": torch.__version__, "hip": torch.version.hip, "device": torch.cuda.get_device_name(0)}), flush=True)
    head = "aeim,ebmj->abij"
    for w in which:
        if w in ("f64", "f32"):
            for n in (64, 96):
                gemm(w, n * n, n * n, n * n, False)
                contraction(w, head, dict(a=n, b=n, i=n, j=n, e=n, m=n))
            gemm(w, 9216, 9216, 9216, True)
        elif w == "bf16":
            gemm(w, 8192, 8192, 8192, False)
            gemm(w, 8192, 8192, 8192, True)
            contraction(w, head, dict(a=96, b=96, i=96, j=96, e=96, m=96))
        elif w == "odd":
            for dt in ("f64", "f32", "bf16"):
                for n in (63, 75, 95):
                    contraction(dt, head, dict(a=n, b=n, i=n, j=n, e=n, m=n), iters=8)
        elif w == "config4":
            contraction("f64", "adij,bcdk->abcijk", dict(a=5, b=40, c=40, i=40, j=40, k=40, d=64), iters=6)
