#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03tile
mkdir -p $OUT
cd $R
export PROBE_CASES="16,896,512,96;32,384,1152,288;64,256,256,80;12,512,896,64;16,858,546,89;32,400,1120,288;64,352,208,80;12,558,884,65;1,3584,2048,96;1,3432,2184,89;1,4096,4096,128;1,4000,4000,128;8,1024,1024,160;8,1000,1000,160;4,2048,2048,256;4,2000,2000,256;64,580,404,217;64,640,384,224;32,464,1152,112;32,512,1152,112"
for t in 64 128; do
CTC_TC_FORCE_TILE=$t timeout -k 10 300 python tools/batched_tail_probe.py > $OUT/shortk_$t.jsonl 2> $OUT/ab.err || { tail -5 $OUT/ab.err; exit 1; }
done
timeout -k 10 300 python tools/batched_tail_probe.py > $OUT/shortk_planner.jsonl 2> $OUT/ab.err || { tail -5 $OUT/ab.err; exit 1; }
python3 - <<'PY'
import json, math
rd = lambda f: [json.loads(l) for l in open("gpurun_out/r03tile/shortk_%s.jsonl" % f)]
for a, b, c in zip(rd("64"), rd("128"), rd("planner")):
    w = lambda t: (math.ceil(a["m"] / t) * t) * (math.ceil(a["n"] / t) * t) / (a["m"] * a["n"])
    print(a["dtype"], a["batch"], a["m"], a["n"], a["k"], "padding ratio %.2f | tile 64: %.1f | tile 128: %.1f (x%.2f) | planner %s %.1f" % (w(128) / w(64), a["tflops"], b["tflops"], b["tflops"] / a["tflops"], c["tile"], c["tflops"]))
PY
