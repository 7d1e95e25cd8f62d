#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03tail
mkdir -p $OUT
cd $R
export TAIL_PROBE_HALF=1
timeout -k 10 500 python tools/tail_cost_probe.py > $OUT/half_modes.jsonl 2> $OUT/modes.err || { tail -5 $OUT/modes.err; exit 1; }
CTC_TC_NO_SPLITK=1 timeout -k 10 300 python tools/tail_cost_probe.py > $OUT/half_nocut.jsonl 2> $OUT/nocut.err || { tail -5 $OUT/nocut.err; exit 1; }
CTC_TC_NO_SPLITK=1 timeout -k 10 300 python tools/tail_cost_probe.py > $OUT/half_nocut2.jsonl 2> $OUT/nocut.err || { tail -5 $OUT/nocut.err; exit 1; }
python3 - <<'PY'
import json
rd = lambda f: [json.loads(l) for l in open("gpurun_out/r03tail/%s.jsonl" % f)]
for a, b, c in zip(rd("half_modes"), rd("half_nocut"), rd("half_nocut2")):
    f = lambda d: "%.1f (cut %d split %d shares %d)" % (d["tflops"], d["cut"], d["split"], d["shares"])
    print(a["dtype"], "k", a["k"], "tiles", a["tiles"], "=", a["full_rounds"], "x512 +", a["tail"], "| uncut %.1f %.1f | by cost %s | chunks %s | shares %s" % (b["by_cost"]["tflops"], c["by_cost"]["tflops"], f(a["by_cost"]), f(a["chunks_only"]), f(a["shares"])))
PY
