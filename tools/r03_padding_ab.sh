#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03tile
mkdir -p $OUT
cd $R
export PROBE_CASES="16,144,672,3872;8,272,1040,4096;4,400,1300,2048;1,528,2112,4096;1,1040,1040,4096;1,1296,1296,2048;1,2096,2096,2048;1,144,4096,4096;1,272,8192,2048;2,656,656,8192;1,1168,1168,1168;1,1424,1424,1424;1,1680,1680,1680;1,400,400,16384;6,784,784,784;1,2320,2320,1024;1,4112,4112,512;1,8208,400,1024"
timeout -k 10 300 python tools/batched_tail_probe.py > $OUT/padw_planner.jsonl 2> $OUT/ab.err || { tail -5 $OUT/ab.err; exit 1; }
for t in 64 128; do
CTC_TC_FORCE_TILE=$t timeout -k 10 300 python tools/batched_tail_probe.py > $OUT/padw_$t.jsonl 2> $OUT/ab.err || { tail -5 $OUT/ab.err; exit 1; }
done
python3 - <<'PY'
import json, math
rd = lambda f: [json.loads(l) for l in open("gpurun_out/r03tile/padw_%s.jsonl" % f)]
for a, b, c in zip(rd("64"), rd("128"), rd("planner")):
    w = lambda t: (math.ceil(a["m"] / t) * t) * (math.ceil(a["n"] / t) * t) / (a["m"] * a["n"])
    print(a["dtype"], a["batch"], a["m"], a["n"], a["k"], "waste 64: %.2f 128: %.2f ratio %.2f | tile 64: %.1f | tile 128: %.1f | planner %s %.1f  x%.2f of better" % (
        w(64), w(128), w(128) / w(64), a["tflops"], b["tflops"], c["tile"], c["tflops"], c["tflops"] / max(a["tflops"], b["tflops"])))
PY
