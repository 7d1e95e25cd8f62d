#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03tail
mkdir -p $OUT
cd $R
export TAIL_PROBE_HALF=1
CTC_TC_NO_SPLITK=1 CTC_TC_DMA_PERSIST=1 timeout -k 10 300 python tools/tail_cost_probe.py f64 > $OUT/half_nocut_walk.jsonl 2> $OUT/nocut.err || { tail -5 $OUT/nocut.err; exit 1; }
CTC_TC_NO_SPLITK=1 CTC_TC_DMA_PERSIST=1 timeout -k 10 300 python tools/tail_cost_probe.py f32 > $OUT/half_nocut_walk_f32.jsonl 2> $OUT/nocut.err || { tail -5 $OUT/nocut.err; exit 1; }
CTC_TC_NO_SPLITK=1 CTC_TC_DMA_PERSIST=1 timeout -k 10 300 python tools/batched_tail_probe.py > $OUT/batched_nocut_walk.jsonl 2> $OUT/new.err || { tail -5 $OUT/new.err; exit 1; }
python3 - <<'PY'
import json
rd = lambda f: [json.loads(l) for l in open("gpurun_out/r03tail/%s.jsonl" % f)]
for a in rd("half_nocut_walk") + rd("half_nocut_walk_f32"):
    print(a["dtype"], "k", a["k"], "tiles", a["tiles"], "=", a["full_rounds"], "x512 +", a["tail"], "uncut, one workgroup per slot: %.1f" % a["by_cost"]["tflops"])
for n in rd("batched_nocut_walk"):
    print(n["dtype"], n["batch"], n["m"], n["n"], n["k"], n["tile"], n["tflops"], "tiles", n["tiles"])
PY
