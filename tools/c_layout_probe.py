import sys, json
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "benchmarks"))
import run_configs as rc
rc.contraction_case("ka,kb->ab", dict(a=8000, b=64000, k=64), "f64", "plain GEMM 8000x64000x64, C rows contiguous")
rc.contraction_case("adij,bcdk->abcijk", dict(a=5, b=40, c=40, i=40, j=40, k=40, d=64), "f64", "config4 slab")
rc.contraction_case("adij,bcdk->abcijk", dict(a=5, b=40, c=40, i=40, j=40, k=48, d=64), "f64", "config4-like, k=48 (runs of 384 B = 3 lines)")
rc.contraction_case("adij,bcdk->abcijk", dict(a=5, b=40, c=40, i=40, j=40, k=32, d=64), "f64", "config4-like, k=32 (runs of 256 B)")
rc.contraction_case("adij,bcdk->abcijk", dict(a=5, b=40, c=40, i=40, j=40, k=64, d=64), "f64", "config4-like, k=64 (runs of 512 B)")
