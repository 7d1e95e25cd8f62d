"""One-off stress of the two FASTR kernels against their oracles: several seeds and shapes each
(read lengths 20..250, windows 60..3000, batches of 4..64 regions); prints one line per case and
exits non-zero on the first mismatch.   python tools/fastr_stress.py [cases]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctc_amd import fastr  # noqa: E402
from oracle import fastr as ofastr  # noqa: E402
from oracle import sw as osw  # noqa: E402

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
threads = min(16, os.cpu_count() or 1)
rs = np.random.RandomState(7)
for i in range(cases):
    read_len = int(rs.choice([20, 36, 75, 100, 151, 250]))
    lo = int(rs.choice([60, 200, 900])) + read_len
    hi = lo + int(rs.choice([10, 400, 2000]))
    b = ofastr.make_batch(seed=100 + i, n_regions=int(rs.choice([4, 16, 64])), max_consensus=int(rs.choice([1, 4, 12])),
                          reads=(1, int(rs.choice([3, 40, 150]))), read_len=read_len, ref_len=(lo, hi),
                          mismatch_pct=int(rs.choice([0, 2, 15])), read_stride=read_len + int(rs.choice([0, 0, 28])))
    d = fastr.DeviceBatch(b, dev)
    d.run()
    score, index = d.results()
    ws, wi = ofastr.best_offsets(b, threads=threads)
    ok = bool(np.array_equal(score, ws) and np.array_equal(index, wi))
    print(json.dumps({"kernel": "best_offsets", "case": i, "pairs": int(d.n_pairs), "read_len": read_len, "ref_len": [lo, hi], "exact": ok}), flush=True)
    if not ok:
        sys.exit(1)
for i in range(cases):
    rl = (int(rs.choice([20, 60, 120])), int(rs.choice([130, 160, 250])))
    wl0 = int(rs.choice([64, 200, 700]))
    b = osw.make_pairs(seed=300 + i, n_windows=int(rs.choice([4, 16])), reads_per_window=(1, int(rs.choice([4, 30]))),
                       ref_len=(wl0, wl0 + int(rs.choice([1, 300, 1500]))), read_len=rl, mismatch=float(rs.choice([0.0, 0.02, 0.2])))
    d = fastr.DeviceAlignmentBatch(b, dev, max_ops=64)
    d.run()
    off, n_ops, ty, ln = d.results()
    w_off, w_n, w_ty, w_ln = osw.align_batch(b, threads=threads)
    # a pair whose operations exceed max_ops reports their number and writes none: compare the rest
    fits = n_ops <= 64
    ok = bool(np.array_equal(off, w_off) and np.array_equal(n_ops, w_n) and
              osw.cigar_strings(n_ops[fits], ty[fits], ln[fits]) == osw.cigar_strings(w_n[fits], w_ty[fits], w_ln[fits]))
    print(json.dumps({"kernel": "sw_align", "case": i, "pairs": int(d.n_pairs), "read_len": rl, "window": wl0, "over_max_ops": int((~fits).sum()), "exact": ok}), flush=True)
    if not ok:
        sys.exit(1)
print("all exact")
