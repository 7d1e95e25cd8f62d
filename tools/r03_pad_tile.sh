#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03tile
mkdir -p $OUT
cd $R
timeout -k 10 300 python tools/pad_tile_probe.py > $OUT/pad_planner.jsonl 2> $OUT/pad.err || { tail -5 $OUT/pad.err; exit 1; }
for t in 64 128; do
CTC_TC_FORCE_TILE=$t timeout -k 10 300 python tools/pad_tile_probe.py > $OUT/pad_$t.jsonl 2> $OUT/pad.err || { tail -5 $OUT/pad.err; exit 1; }
done
python3 - <<'PY'
import json
rd = lambda f: [json.loads(l) for l in open("gpurun_out/r03tile/pad_%s.jsonl" % f)]
for a, b, c in zip(rd("64"), rd("128"), rd("planner")):
    print(a["dtype"], a["spec"], a["M"], a["N"], a["K"], "tile 64 (%s): %.1f | tile 128 (%s): %.1f | planner %s %s %.1f" % (a["transposes"], a["tflops"], b["transposes"], b["tflops"], c["tile"], c["transposes"], c["tflops"]))
PY
