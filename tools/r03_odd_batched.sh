#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03tile
mkdir -p $OUT
cd $R
export PROBE_CASES="4,700,900,1101;8,500,500,2001;3,1000,1000,3001;16,257,255,4097;2,1501,1503,1025;6,801,803,805;12,333,777,1111;1,1501,1503,4097;1,901,903,8193;40,129,131,2049"
timeout -k 10 300 python tools/batched_tail_probe.py > $OUT/odd_planner.jsonl 2> $OUT/ab.err || { tail -5 $OUT/ab.err; exit 1; }
for t in 64 128; do
CTC_TC_FORCE_TILE=$t CTC_TC_NO_PAD_TRANSPOSES=1 timeout -k 10 300 python tools/batched_tail_probe.py > $OUT/odd_$t.jsonl 2> $OUT/ab.err || { tail -5 $OUT/ab.err; exit 1; }
done
CTC_TC_NO_FLAT_BATCH=1 CTC_TC_NO_BATCHED_CHUNKS=1 CTC_TC_SPLIT_BY_SLOTS=1 timeout -k 10 300 python tools/batched_tail_probe.py > $OUT/odd_before.jsonl 2> $OUT/ab.err || { tail -5 $OUT/ab.err; exit 1; }
python3 - <<'PY'
import json
rd = lambda f: [json.loads(l) for l in open("gpurun_out/r03tile/odd_%s.jsonl" % f)]
for a, b, c, d in zip(rd("64"), rd("128"), rd("planner"), rd("before")):
    print(a["dtype"], a["batch"], a["m"], a["n"], a["k"], "tile 64: %.1f (cut %d split %d) | tile 128: %.1f (tiles %d cut %d split %d) | planner %s %.1f | before today %s %.1f  x%.2f" % (
        a["tflops"], a["cut"], a["split"], b["tflops"], b["tiles"], b["cut"], b["split"], c["tile"], c["tflops"], d["tile"], d["tflops"], c["tflops"] / d["tflops"]))
PY
