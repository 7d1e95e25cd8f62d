"""Averages the counters of a rocprofv3 --pmc run (rocpd sqlite output) per GEMM kernel."""
import glob
import sqlite3
import sys
from collections import defaultdict

for d in sys.argv[1:]:
    acc = defaultdict(lambda: [0.0, 0, 0.0])
    for path in glob.glob(d + "/**/*.db", recursive=True):
        c = sqlite3.connect(path)
        for name, counter, value, dur in c.execute(
                "select kernel_name, counter_name, value, duration from counters_collection"):
            if "gemm" not in name:
                continue
            k = "stream" if "gemm_stream" in name else ("fixup" if "fixup" in name else ("lds_dma" if "gemm_dma" in name else "classic"))
            a = acc[(k, counter)]
            a[0] += float(value)
            a[1] += 1
            a[2] += float(dur)
    print(d)
    for (k, cn), (v, n, dur) in sorted(acc.items()):
        print("  %-8s %-36s %14.5g  (n=%d, %.3f ms)" % (k, cn, v / n, n, dur / n * 1e-6))
