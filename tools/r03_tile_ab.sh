#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03tile
mkdir -p $OUT
cd $R
export PROBE_CASES="1,2048,1536,1100;1,1792,2048,1280;1,1700,1900,1500;4,700,900,1100;2,1024,1536,1100;3,1000,1000,1200;1,2048,1536,1536;1,1792,2048,2048;4,700,900,1600;1,2000,1500,1030;6,512,512,1400;1,1536,1280,1100;1,1280,1280,1400"
for t in 64 128; do
CTC_TC_FORCE_TILE=$t timeout -k 10 300 python tools/batched_tail_probe.py > $OUT/ab_$t.jsonl 2> $OUT/ab.err || { tail -5 $OUT/ab.err; exit 1; }
done
timeout -k 10 300 python tools/batched_tail_probe.py > $OUT/ab_planner.jsonl 2> $OUT/ab.err || { tail -5 $OUT/ab.err; exit 1; }
python3 - <<'PY'
import json
rd = lambda f: [json.loads(l) for l in open("gpurun_out/r03tile/ab_%s.jsonl" % f)]
for a, b, c in zip(rd("64"), rd("128"), rd("planner")):
    print(a["dtype"], a["batch"], a["m"], a["n"], a["k"], "tile 64: %.1f (tiles %d cut %d split %d) | tile 128: %.1f (tiles %d cut %d split %d shares %d) | planner %s %.1f" % (
        a["tflops"], a["tiles"], a["cut"], a["split"], b["tflops"], b["tiles"], b["cut"], b["split"], b["shares"], c["tile"], c["tflops"]))
PY
