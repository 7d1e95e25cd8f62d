import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "benchmarks"))
import run_configs as rc
for n in (1024, 2048, 4096):
    rc.contraction_case("ak,bk->ab", dict(a=n, b=n, k=n), "bf16", "plain GEMM bf16 %d^3 K-major" % n)
