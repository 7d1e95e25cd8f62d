#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03plain
mkdir -p $OUT
cd $R
export PROBE_CASES="4096,64,64,64;16384,64,64,64;1024,128,128,128;4096,128,128,128;2048,96,96,96;4096,64,64,512;1024,128,128,1024;8192,48,48,48;512,128,128,4096"
timeout -k 10 300 python tools/batched_tail_probe.py > $OUT/new.jsonl 2> $OUT/new.err || { tail -5 $OUT/new.err; exit 1; }
CTC_TC_NO_FLAT_BATCH=1 timeout -k 10 300 python tools/batched_tail_probe.py > $OUT/old.jsonl 2> $OUT/old.err || { tail -5 $OUT/old.err; exit 1; }
python3 - <<'PY'
import json
rd = lambda f: [json.loads(l) for l in open("gpurun_out/r03plain/%s.jsonl" % f)]
for n, o in zip(rd("new"), rd("old")):
    print(n["dtype"], n["batch"], n["m"], n["n"], n["k"], "grid.y %s %.1f | one list %s %.1f TF/s  x%.3f" % (o["tile"], o["tflops"], n["tile"], n["tflops"], n["tflops"] / o["tflops"]))
PY
