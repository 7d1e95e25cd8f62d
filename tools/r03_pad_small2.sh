#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03pad
mkdir -p $OUT
cd $R
timeout -k 10 400 python tools/pad_small_probe.py > $OUT/default2.jsonl 2> $OUT/err || { tail -5 $OUT/err; exit 1; }
CTC_TC_PAD_MIN_MN=512 CTC_TC_OLD_TILE_RULE=1 timeout -k 10 400 python tools/pad_small_probe.py > $OUT/pad512_old.jsonl 2> $OUT/err || { tail -5 $OUT/err; exit 1; }
python3 - <<'PY'
import json
rd = lambda f: [json.loads(l) for l in open("gpurun_out/r03pad/%s.jsonl" % f)]
for a, b in zip(rd("default2"), rd("pad512_old")):
    if a["tile"] != b["tile"] or a["transposes"] != b["transposes"]:
        print(a["dtype"], a["spec"], a["m"], a["n"], a["k"], a["layouts"], a["tile"], a["transposes"], a["tflops"], "| min 512 + old tile rule:", b["tile"], b["transposes"], b["tflops"], "x%.2f" % (b["tflops"] / a["tflops"]))
PY
