"""Odd extents: the fused kernel (element loads) against what a pad-and-transpose path could reach:
both operands brought to K-major rows [.., K padded to a multiple of 8] (the LDS-DMA kernel qualifies), C written
in place through the offset tables.  Times the GEMM part on the padded operands and the two transposes (packed
output, which is what the permutation kernels produce today) separately.  One JSON line per extent."""
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import ctc_amd  # noqa: E402


def timed(fn, iters=8, seconds=0.1):
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


def main():
    dev = torch.device("cuda:0")
    for dtype, tdt, unit in (("f64", torch.float64, 8), ("f32", torch.float32, 16)):
        for n in (63, 65, 75, 81, 95):
            ext = dict(a=n, b=n, i=n, j=n, e=n, m=n)
            A = torch.empty(n ** 4, dtype=tdt, device=dev)
            B = torch.empty(n ** 4, dtype=tdt, device=dev)
            C = torch.empty(n ** 4, dtype=tdt, device=dev)
            ctc_amd.fill(A, 1)
            ctc_amd.fill(B, 2)
            fused = ctc_amd.Plan("aeim,ebmj->abij", ext, dtype)
            t_fused = timed(lambda: fused.execute(A, B, C))
            K = n * n
            Kp = (K + unit - 1) // unit * unit
            A2 = torch.zeros(n * n * Kp, dtype=tdt, device=dev)
            B2 = torch.zeros(n * n * Kp, dtype=tdt, device=dev)
            A2.view(n, n, Kp)[:, :, :K] = A.view(n, n, n, n).permute(0, 2, 1, 3).reshape(n, n, K)       # [a,i,(e,m)]
            B2.view(n, n, Kp)[:, :, :K] = B.view(n, n, n, n).permute(1, 3, 0, 2).reshape(n, n, K)       # [b,j,(e,m)]
            C2 = torch.empty(n ** 4, dtype=tdt, device=dev)
            gemm = ctc_amd.Plan("aik,bjk->abij", dict(a=n, b=n, i=n, j=n, k=Kp), dtype)
            names = ("launches_classic", "launches_dma_gemm")
            before = [ctc_amd.get_option(x) for x in names]
            t_gemm = timed(lambda: gemm.execute(A2, B2, C2))
            took = [x[9:] for x, b in zip(names, before) if ctc_amd.get_option(x) > b]
            diff = float((C2 - C).abs().max() / C.abs().max())
            At = torch.empty(n ** 4, dtype=tdt, device=dev)
            t_pa = timed(lambda: ctc_amd.permute(A.view(n, n, n, n), (0, 2, 1, 3), out=At.view(n, n, n, n)))
            t_pb = timed(lambda: ctc_amd.permute(B.view(n, n, n, n), (1, 3, 0, 2), out=At.view(n, n, n, n)))
            flops = 2.0 * n ** 6
            total = t_gemm + t_pa + t_pb
            print(json.dumps({"dtype": dtype, "extent": n, "fused_ms": round(t_fused, 4), "fused_tflops": round(flops / t_fused * 1e-9, 2),
                              "padded_gemm_ms": round(t_gemm, 4), "padded_gemm_tflops": round(flops / t_gemm * 1e-9, 2), "gemm_kernel": took,
                              "transpose_a_ms": round(t_pa, 4), "transpose_b_ms": round(t_pb, 4),
                              "transposes_TBps": round(2 * 2 * n ** 4 * A.element_size() / (t_pa + t_pb) * 1e-9, 2),
                              "pad_path_ms": round(total, 4), "pad_path_tflops": round(flops / total * 1e-9, 2),
                              "pad_path_over_fused": round(t_fused / total, 3), "max_rel_diff": diff}), flush=True)
            del A, B, C, A2, B2, C2, At
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
