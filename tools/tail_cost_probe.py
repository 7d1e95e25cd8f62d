"""What the last, partly filled round of 128 x 128 tiles costs, cut or not: plain products with a chosen number of
tiles (full rounds of 512 + a tail), by cost (default), never cut (CTC_TC_NO_SPLITK=1 in the environment), equal
chunks only (stream_k 1), stream-K shares whenever possible (stream_k 2).  One JSON line per case."""
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


dev = torch.device("cuda:0")
which = sys.argv[1:] or ["f64", "f32"]
# tile grids: rows x columns of 128 x 128 tiles
GRIDS = ((16, 32), (18, 32), (20, 32), (24, 32), (26, 32), (28, 32), (30, 32), (32, 32), (34, 32), (36, 32), (40, 32), (48, 32), (16, 36), (16, 40), (16, 48), (16, 56))
KS = (512, 1024, 4096)
if os.environ.get("TAIL_PROBE_HALF"):
    # tails of half the slots (and 3/8) behind one to five full rounds
    GRIDS = ((24, 32), (40, 32), (56, 32), (72, 32), (88, 32), (22, 32), (38, 32), (54, 32))
    KS = (512, 1024, 2048, 4096, 8192)
EDGE = 128
if os.environ.get("TAIL_PROBE_SMALL"):
    # the 64 x 64 tile (run under CTC_TC_FORCE_TILE=64): tile counts from a quarter of a CU each to two rounds
    EDGE = 64
    GRIDS = ((8, 8), (8, 16), (12, 16), (16, 16), (16, 20), (16, 24), (16, 32), (20, 32), (24, 32), (28, 32), (32, 32), (32, 36), (32, 40), (32, 48), (32, 64))
    KS = (576, 1024, 4096)
if os.environ.get("TAIL_PROBE_ODDK"):
    # an odd K: A by element loads, the register-staged kernel on the large tile
    GRIDS = ((4, 16), (8, 16), (12, 16), (16, 16), (16, 20), (16, 24), (16, 32), (20, 32), (24, 32), (32, 32), (32, 40))
    KS = (1025, 4097)
if os.environ.get("TAIL_PROBE_BF16"):
    # the 128 x 128 bf16 tile (the planner takes it below 192 tiles of 256 x 256)
    GRIDS = ((8, 16), (8, 32), (12, 32), (14, 32), (16, 32), (18, 32), (20, 32), (22, 32))
    KS = (1024, 4096, 16384)
    which = ["bf16"]
for dt in which:
    tdt = {"f64": torch.float64, "f32": torch.float32, "bf16": torch.bfloat16}[dt]
    for k in KS:
        for tr, tc in GRIDS:
            m, n = tr * EDGE, tc * EDGE
            A = torch.empty(m * k, dtype=tdt, device=dev)
            B = torch.empty(k * n, dtype=tdt, device=dev)
            C = torch.empty(m * n, dtype=tdt, device=dev)
            if dt == "bf16":
                A.copy_(torch.rand(m * k, dtype=torch.float32, device=dev) - 0.5)
                B.copy_(torch.rand(k * n, dtype=torch.float32, device=dev) - 0.5)
            else:
                ctc_amd.fill(A, 1)
                ctc_amd.fill(B, 2)
            plan = ctc_amd.Plan("ak,kb->ab", dict(a=m, b=n, k=k), dt)
            plan.prepare()
            out = {"dtype": dt, "tiles": tr * tc, "full_rounds": tr * tc // 512, "tail": tr * tc % 512, "k": k, "tile_edge": EDGE, "no_splitk_env": os.environ.get("CTC_TC_NO_SPLITK")}
            for mode, label in ((0, "by_cost"), (1, "chunks_only"), (2, "shares")) if dt != "bf16" else ((0, "by_cost"),):
                ctc_amd.set_option("stream_k", mode)
                ctc_amd.set_option("stream_mode", 1)
                sch = plan.schedule()
                ms = timed(lambda: plan.execute(A, B, C))
                out["tile"] = list(plan.summary()["tile"])[:2]
                out[label] = {"ms": round(ms, 4), "tflops": round(2.0 * m * n * k / ms * 1e-9, 1), "cut": sch["tail_tiles"], "split": sch["split"], "shares": sch["stream_k_shares"]}
                if os.environ.get("CTC_TC_NO_SPLITK"):
                    break
            ctc_amd.set_option("stream_k", 0)
            ctc_amd.set_option("stream_mode", 0)
            print(json.dumps(out), flush=True)
            del A, B, C
            torch.cuda.empty_cache()
