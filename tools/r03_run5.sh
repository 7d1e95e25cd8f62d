#!/bin/bash
# stream-K shares formed over groups of g tiles (g workgroups in K lock-step): parity, then g = 1, 8, 16, 64
set -o pipefail
OUT=$PWD/gpurun_out/r03e
mkdir -p $OUT
timeout -k 10 600 python -u -m pytest tests/test_dma_gemm_gpu.py -x -q --timeout 120 --timeout-method thread -k "stream_k or schedule" > $OUT/test_sk.log 2>&1 || { tail -40 $OUT/test_sk.log; exit 1; }
tail -2 $OUT/test_sk.log
for g in 1 8 16 64; do
  echo "== group $g"
  CTC_TC_STREAMK_GROUP=$g timeout -k 10 300 python -u tools/extent_sweep.py f64 48 56 72 80 88 96 > $OUT/sweep_g$g.jsonl 2> $OUT/sweep_g$g.err || { tail -20 $OUT/sweep_g$g.err; exit 1; }
  python - <<PY
import json
for l in open("$OUT/sweep_g$g.jsonl"):
    r = json.loads(l)
    print(r["extent"], "equal", r["equal_chunks"]["TFLOP/s"], "forced", r["shares_forced"]["TFLOP/s"], "by_cost", r["by_cost"]["TFLOP/s"], r["by_cost"]["shares"])
PY
  CTC_TC_STREAMK=2 CTC_TC_STREAMK_GROUP=$g timeout -k 10 300 python -u tools/slab_rehearsal.py headline > $OUT/slab_g$g.jsonl 2> $OUT/slab_g$g.err || { tail -20 $OUT/slab_g$g.err; exit 1; }
  python - <<PY
import json
for l in open("$OUT/slab_g$g.jsonl"):
    r = json.loads(l)
    print("slab", r["n_ranks"], r["tflops_per_gpu"], r["speedup_ceiling"])
PY
done
