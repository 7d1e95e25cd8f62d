#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03tile
mkdir -p $OUT
cd $R
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $OUT/pytest_gpu.log 2>&1 || { tail -40 $OUT/pytest_gpu.log; exit 1; }
tail -3 $OUT/pytest_gpu.log
export PROBE_CASES="1,2048,1536,1100;1,1792,2048,1280;1,1700,1900,1500;4,700,900,1100;2,1024,1536,1100;3,1000,1000,1200;1,2048,1536,1536;1,1792,2048,2048;4,700,900,1600;1,2000,1500,1030;6,512,512,1400;1,1536,1280,1100;1,1280,1280,1400;1,1536,1536,1600;1,1536,1536,2048;1,1600,1600,1600;1,1408,1408,1800"
timeout -k 10 300 python tools/batched_tail_probe.py > $OUT/ab_planner2.jsonl 2> $OUT/ab.err || { tail -5 $OUT/ab.err; exit 1; }
for t in 64 128; do
CTC_TC_FORCE_TILE=$t timeout -k 10 300 python tools/batched_tail_probe.py > $OUT/ab2_$t.jsonl 2> $OUT/ab.err || { tail -5 $OUT/ab.err; exit 1; }
done
python3 - <<'PY'
import json
rd = lambda f: [json.loads(l) for l in open("gpurun_out/r03tile/%s.jsonl" % f)]
for a, b, c in zip(rd("ab2_64"), rd("ab2_128"), rd("ab_planner2")):
    print(a["dtype"], a["batch"], a["m"], a["n"], a["k"], "tile 64: %.1f | tile 128: %.1f (tiles %d cut %d split %d shares %d) | planner %s %.1f" % (
        a["tflops"], b["tflops"], b["tiles"], b["cut"], b["split"], b["shares"], c["tile"], c["tflops"]))
PY
