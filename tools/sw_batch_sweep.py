import os, sys, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
from ctc_amd import fastr
from oracle import sw as osw
DEV = torch.device("cuda:0")
for nw in (64, 256, 1024):
    b = osw.make_pairs(seed=0x5117, n_windows=nw, reads_per_window=(20, 100))
    d = fastr.DeviceAlignmentBatch(b, DEV)
    for _ in range(2): d.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): d.run()
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 5
    cells = int((b["ref_len"].astype(np.int64) * b["read_len"].astype(np.int64)).sum())
    print(json.dumps({"windows": nw, "pairs": d.n_pairs, "ms": round(ms, 3), "Gcell/s": round(cells / ms * 1e-6, 1), "workspace_GB": round(d.workspace_bytes / 1e9, 2)}), flush=True)
    del d
