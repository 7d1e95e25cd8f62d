"""Secondary measurements (BASELINE configs 3 and 5, plus kernel variants).  The headline
(config 2) lives in bench.py; this script prints one JSON line per measurement.

  python benchmarks/run_configs.py [permute] [adversarial] [f32] [bf16] [six] [small] [fastr|fastr_large] [sw] [ttgt]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import ctc_amd  # noqa: E402

HBM_PEAK_GBS = 8000.0
# dense MFMA peaks of MI355X per element type, TFLOP/s (fp64: public datasheet; the others: the guide's dense figures)
MFMA_PEAK_TFLOPS = {"f64": 78.6, "f32": 157.3, "bf16": 2500.0}


def roofline(bound, achieved, peak, unit):
    return {"bound": bound, "achieved": round(achieved, 3), "peak": peak, "unit": unit, "frac": round(achieved / peak, 4)}
DEV = torch.device("cuda:0")


def time_ms(fn, warmup=3, iters=10):
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


def permute_case(shape, perm, dtype=torch.float32, name=None):
    x = torch.empty(shape, dtype=dtype, device=DEV)
    ctc_amd.fill(x.view(-1), 0xC7C0A)
    out = torch.empty([shape[p] for p in perm], dtype=dtype, device=DEV)
    ms = time_ms(lambda: ctc_amd.permute(x, perm, out=out))
    nbytes = 2 * x.numel() * x.element_size()
    y = torch.empty_like(x)
    ms_copy = time_ms(lambda: y.copy_(x))
    print(json.dumps({"workload": name or "permute %s perm %s %s" % (list(shape), list(perm), dtype),
                      "ms": round(ms, 4), "GB/s": round(nbytes / ms * 1e-6, 1),
                      "frac_hbm_peak": round(nbytes / ms * 1e-6 / HBM_PEAK_GBS, 4),
                      "d2d_copy_GB/s": round(nbytes / ms_copy * 1e-6, 1),
                      "algorithmic_bytes": nbytes,
                      "roofline": roofline("hbm", nbytes / ms * 1e-6, HBM_PEAK_GBS, "GB/s")}))


def contraction_case(spec, ext, dtype, name, ttgt=False):
    tdt = {"f64": torch.float64, "f32": torch.float32, "bf16": torch.bfloat16}[dtype]
    ins, outm = spec.split("->")
    la, lb = ins.split(",")
    n = lambda s: int(torch.tensor([ext[c] for c in s]).prod()) if s else 1  # noqa: E731
    A = torch.empty(n(la), dtype=tdt, device=DEV)
    B = torch.empty(n(lb), dtype=tdt, device=DEV)
    C = torch.empty(n(outm), dtype=tdt, device=DEV)
    ctc_amd.fill(A, 0xC7C0A)
    ctc_amd.fill(B, 0xC7C0B)
    plan = ctc_amd.Plan(spec, ext, dtype)
    if ttgt:
        ws = torch.empty(plan.ttgt_workspace_bytes(), dtype=torch.uint8, device=DEV)
        ms = time_ms(lambda: plan.execute_ttgt(A, B, C, ws))
    else:
        ms = plan.time(A, B, C, warmup=3, iters=10)
    nbytes = (A.numel() + B.numel() + C.numel()) * A.element_size()
    s = plan.summary()
    # the roof that bounds the contraction: HBM when moving its operands once takes longer than its MFMAs
    t_mfma = plan.flops / (MFMA_PEAK_TFLOPS[dtype] * 1e9)
    t_hbm = nbytes / (HBM_PEAK_GBS * 1e6)
    roof = (roofline("hbm", nbytes / ms * 1e-6, HBM_PEAK_GBS, "GB/s") if t_hbm > t_mfma
            else roofline("mfma", plan.flops / ms * 1e-9, MFMA_PEAK_TFLOPS[dtype], "TFLOP/s"))
    print(json.dumps({"workload": name, "path": "ttgt" if ttgt else "fused", "ms": round(ms, 4), "roofline": roof,
                      "TFLOP/s": round(plan.flops / ms * 1e-9, 3), "GB/s": round(nbytes / ms * 1e-6, 1),
                      "frac_hbm_peak": round(nbytes / ms * 1e-6 / HBM_PEAK_GBS, 4),
                      "gemm": [s["M"], s["N"], s["K"], s["batch_count"]], "tile": s["tile"],
                      "layouts": [s["layout_a"], s["layout_b"]], "swapped": s["swapped"]}))


def main():
    what = set(sys.argv[1:]) or {"permute", "adversarial", "f32"}
    if "permute" in what:
        permute_case((128, 128, 128, 128), (3, 2, 1, 0), name="config3 pure permute dcba<-abcd fp32 128^4")
        permute_case((128, 128, 128, 128), (3, 2, 1, 0), torch.float64, name="permute dcba<-abcd fp64 128^4")
        permute_case((128, 128, 128, 128), (0, 1, 3, 2), name="permute abdc<-abcd fp32 128^4")
        permute_case((128, 128, 128, 128), (1, 0, 2, 3), name="permute bacd<-abcd fp32 128^4 (row copy)")
        permute_case((128, 8, 128, 128), (3, 1, 2, 0), name="permute akcd<-dkca fp32 (A of config 3)")
    if "adversarial" in what:
        ext = dict(a=128, b=8, c=128, d=128, k=8)
        contraction_case("dkca,bk->abcd", ext, "f32", "config3 adversarial abcd=dkca,bk fp32")
        contraction_case("dkca,bk->abcd", ext, "f32", "config3 adversarial abcd=dkca,bk fp32", ttgt=True)
        # same contraction with c = 1024: 1.07 GB moved, four times the 256 MiB Infinity Cache
        contraction_case("dkca,bk->abcd", dict(a=128, b=8, c=1024, d=128, k=8), "f32",
                         "config3 adversarial, c=1024 (beyond the Infinity Cache)")
    if "f32" in what:
        for n in (64, 96):
            contraction_case("aeim,ebmj->abij", dict(a=n, b=n, i=n, j=n, e=n, m=n), "f32", "ccsd fp32 extent %d" % n)
    if "bf16" in what:
        for n in (64, 96):
            contraction_case("aeim,ebmj->abij", dict(a=n, b=n, i=n, j=n, e=n, m=n), "bf16", "ccsd bf16 extent %d" % n)
        contraction_case("ak,bk->ab", dict(a=8192, b=8192, k=8192), "bf16", "plain GEMM bf16 8192^3, both operands K-major")
        contraction_case("ka,kb->ab", dict(a=8192, b=8192, k=8192), "bf16", "plain GEMM bf16 8192^3, both operands outer-major")
        contraction_case("aeim,ebmj->abij", dict(a=63, b=63, i=63, j=63, e=63, m=63), "bf16", "ccsd bf16 extent 63 (element loads)")
    if "six" in what:
        # config 4: abcijk = adij, bcdk, free extents 40, d = 64; one GPU's slab of the free index a (5 of 40)
        e = dict(a=5, b=40, c=40, i=40, j=40, k=40, d=64)
        contraction_case("adij,bcdk->abcijk", e, "f64", "config4 slab a=5/40: abcijk=adij,bcdk fp64 (|C| 4.1 GB)")
        e = dict(a=40, b=40, c=40, i=40, j=40, k=40, d=64)
        contraction_case("adij,bcdk->abcijk", e, "f64", "config4 whole on one GPU (|C| 32.8 GB)")
    if "small" in what:
        import numpy as np
        from oracle import ref
        count, choices = 4096, [8, 16, 24, 32]
        dims = []
        for i in range(count):
            x = ref.splitmix64(0xC7C05 + i)
            dims.append((choices[x % 4], choices[(x >> 8) % 4], choices[(x >> 16) % 4]))
        desc, na, nb, nc = ctc_amd.pack_small_problems(dims)
        flops = sum(2 * m * n * k for m, n, k in dims)
        dP = torch.from_numpy(desc.view(np.uint8).copy()).to(DEV)
        for tdt, name in ((torch.float64, "f64"), (torch.float32, "f32")):
            A = torch.empty(na, dtype=tdt, device=DEV)
            B = torch.empty(nb, dtype=tdt, device=DEV)
            C = torch.empty(nc, dtype=tdt, device=DEV)
            ctc_amd.fill(A, 0xC7C05)
            ctc_amd.fill(B, 0xC7C06)
            ms = time_ms(lambda: ctc_amd.grouped_contract(dP, A, B, C), warmup=5, iters=50)
            nbytes = (na + nb + nc) * A.element_size()
            # the same problems one launch each (what a per-problem API would cost)
            plans = {}
            t0 = time.perf_counter()
            sub = 256
            for p, (m, n, k) in list(zip(desc, dims))[:sub]:
                key = (m, n, k)
                if key not in plans:
                    plans[key] = ctc_amd.Plan("ak,kb->ab", dict(a=m, b=n, k=k), name)
                plans[key].execute(A[int(p["off_a"]):], B[int(p["off_b"]):], C[int(p["off_c"]):])
            torch.cuda.synchronize()
            per_launch_ms = (time.perf_counter() - t0) / sub * count * 1e3
            print(json.dumps({"workload": "config5 grouped: 4096 x (ab=ak,kb), extents in {8,16,24,32}, %s" % name,
                              "launches": 1, "ms": round(ms, 4), "TFLOP/s": round(flops / ms * 1e-9, 3),
                              "GB/s": round(nbytes / ms * 1e-6, 1), "flops": flops,
                              "one_launch_per_problem_ms_extrapolated_from_%d" % sub: round(per_launch_ms, 2)}))
    if "fastr" in what or "fastr_large" in what:
        import numpy as np
        from ctc_amd import fastr
        from oracle import fastr as ofastr
        # 32 regions x U{1..8} consensuses x U{20..200} reads, read length 100; fastr_large: 256 regions,
        # enough workgroups to fill the chip several times over
        b = ofastr.make_batch(seed=0xFA57, n_regions=256 if "fastr_large" in what else 32)
        d = fastr.DeviceBatch(b, DEV)
        ms = time_ms(lambda: d.run(), warmup=3, iters=20)
        t0 = time.perf_counter()
        want_score, want_index = ofastr.best_offsets(b, threads=min(16, os.cpu_count() or 1))
        cpu_s = time.perf_counter() - t0
        score, index = d.results()
        exact = bool(np.array_equal(score, want_score) and np.array_equal(index, want_index))
        # work: every (pair, offset, read base) byte compare of a full scan
        compares = 0
        cons = 0
        read_off = 0
        for t in range(len(b["n_consensus"])):
            rl = b["read_len"][read_off:read_off + b["alt_reads"][t]].astype(np.int64)
            for _k in range(b["n_consensus"][t]):
                L = int(b["ref_lens"][cons])
                compares += int(((np.maximum(L - rl, 0) + 1) * rl).sum())
                cons += 1
            read_off += b["alt_reads"][t]
        in_bytes = b["ref"].size + b["read_seq"].size + b["quals"].size + 8 * b["read_len"].size + 8 * d.n_pairs
        print(json.dumps({"workload": "config5(ii) FASTR scoring batch: %d regions, %d (read, consensus) pairs" % (len(b["n_consensus"]), d.n_pairs),
                          "ms": round(ms, 4), "pairs/s": round(d.n_pairs / ms * 1e3), "bit_exact_vs_oracle": exact,
                          "full_scan_byte_compares": compares,
                          "full_scan_Gcompare/s": round(compares / ms * 1e-6, 1),
                          "bytes_moved": int(in_bytes), "GB/s": round(in_bytes / ms * 1e-6, 2),
                          "cpu_oracle_s": round(cpu_s, 3), "cpu_threads": min(16, os.cpu_count() or 1)}))
    if "sw" in what:
        import numpy as np
        from ctc_amd import fastr
        from oracle import sw as osw
        # 64 windows x U{20..100} reads of 60..160 bases against windows of 200..1200 bases
        b = osw.make_pairs(seed=0x5117, n_windows=64, reads_per_window=(20, 100))
        d = fastr.DeviceAlignmentBatch(b, DEV)
        ms = time_ms(lambda: d.run(), warmup=2, iters=10)
        threads = min(16, os.cpu_count() or 1)
        t0 = time.perf_counter()
        w_off, w_n, w_ty, w_ln = osw.align_batch(b, threads=threads)
        cpu_s = time.perf_counter() - t0
        off, n_ops, ty, ln = d.results()
        exact = bool(np.array_equal(off, w_off) and np.array_equal(n_ops, w_n) and
                     osw.cigar_strings(n_ops, ty, ln) == osw.cigar_strings(w_n, w_ty, w_ln))
        cells = int((b["ref_len"].astype(np.int64) * b["read_len"].astype(np.int64)).sum())
        print(json.dumps({"workload": "pairwise alignment batch: %d (window, read) pairs" % d.n_pairs,
                          "ms": round(ms, 4), "pairs/s": round(d.n_pairs / ms * 1e3), "cells": cells,
                          "Gcell/s": round(cells / ms * 1e-6, 2), "bit_exact_vs_oracle": exact,
                          "backtrack_store_MB": round(d.workspace_bytes / 1e6, 1),
                          "cpu_oracle_s": round(cpu_s, 3), "cpu_Gcell/s": round(cells / cpu_s * 1e-9, 3),
                          "cpu_threads": threads}))
    if "ttgt" in what:
        # odd extents: no 16-byte vectors are legal, the fused kernel loads element by element
        for n in (63, 75):
            e = dict(a=n, b=n, i=n, j=n, e=n, m=n)
            contraction_case("aeim,ebmj->abij", e, "f64", "ccsd fp64 extent %d (odd: element loads)" % n, ttgt=True)
            contraction_case("aeim,ebmj->abij", e, "f64", "ccsd fp64 extent %d (odd: element loads)" % n)
        # operands whose unit-stride mode is on the "wrong" side: A's fastest mode is free, B's is contracted
        e = dict(a=64, b=64, i=64, j=64, e=64, m=64)
        contraction_case("eaim,jbem->abij", e, "f64", "ccsd fp64 extent 64, A outer-major / B K-major", ttgt=True)
        contraction_case("eaim,jbem->abij", e, "f64", "ccsd fp64 extent 64, A outer-major / B K-major")
        n = 64
        contraction_case("aeim,ebmj->abij", dict(a=n, b=n, i=n, j=n, e=n, m=n), "f64", "ccsd fp64 extent 64", ttgt=True)
        contraction_case("aeim,ebmj->abij", dict(a=n, b=n, i=n, j=n, e=n, m=n), "f64", "ccsd fp64 extent 64")


if __name__ == "__main__":
    main()
