#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03pad
mkdir -p $OUT
cd $R
timeout -k 10 400 python tools/pad_small_probe.py > $OUT/default.jsonl 2> $OUT/err || { tail -5 $OUT/err; exit 1; }
CTC_TC_PAD_MIN_MN=512 timeout -k 10 400 python tools/pad_small_probe.py > $OUT/pad512.jsonl 2> $OUT/err || { tail -5 $OUT/err; exit 1; }
python3 - <<'PY'
import json
rd = lambda f: [json.loads(l) for l in open("gpurun_out/r03pad/%s.jsonl" % f)]
for a, b in zip(rd("default"), rd("pad512")):
    print(a["dtype"], a["spec"], a["m"], a["n"], a["k"], a["layouts"], a["tile"], a["transposes"], a["tflops"], "| min 512:", b["tile"], b["transposes"], b["tflops"], "x%.2f" % (b["tflops"] / a["tflops"]))
PY
