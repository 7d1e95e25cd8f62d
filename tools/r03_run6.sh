#!/bin/bash
# the whole GPU suite, then sweep / rehearsal / bench on the schedule chosen by cost
set -o pipefail
OUT=$PWD/gpurun_out/r03f
mkdir -p $OUT
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $OUT/pytest_gpu.log 2>&1 || { tail -40 $OUT/pytest_gpu.log; exit 1; }
tail -3 $OUT/pytest_gpu.log
timeout -k 10 300 python -u tools/slab_rehearsal.py > $OUT/slab_rehearsal.jsonl 2> $OUT/slab_rehearsal.err || { tail -20 $OUT/slab_rehearsal.err; exit 1; }
cut -c1-330 $OUT/slab_rehearsal.jsonl
timeout -k 10 600 python -u tools/extent_sweep.py > $OUT/extent_sweep.jsonl 2> $OUT/extent_sweep.err || { tail -20 $OUT/extent_sweep.err; exit 1; }
python - <<PY
import json
for l in open("$OUT/extent_sweep.jsonl"):
    r = json.loads(l)
    print(r["extent"], "equal", r["equal_chunks"]["TFLOP/s"], "forced", r["shares_forced"]["TFLOP/s"], "by_cost", r["by_cost"]["TFLOP/s"], r["by_cost"]["shares"], r["by_cost"]["split"])
PY
timeout -k 10 300 python -u bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
cut -c1-700 $OUT/bench.json
