#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03tail
mkdir -p $OUT
cd $R
export TAIL_PROBE_SMALL=1 CTC_TC_FORCE_TILE=64
CTC_TC_NO_SPLITK=1 timeout -k 10 300 python tools/tail_cost_probe.py > $OUT/small_nocut.jsonl 2> $OUT/sweep.err || { tail -5 $OUT/sweep.err; exit 1; }
timeout -k 10 300 python tools/tail_cost_probe.py > $OUT/small_modes.jsonl 2> $OUT/sweep.err || { tail -5 $OUT/sweep.err; exit 1; }
python3 -c "
import json
a = [json.loads(l) for l in open('$OUT/small_nocut.jsonl')]; b = [json.loads(l) for l in open('$OUT/small_modes.jsonl')]
for x, y in zip(a, b):
    d = y['by_cost']
    print(x['dtype'], x['tile'], 'k', x['k'], 'tiles', x['tiles'], 'never cut %.4f ms %.1f TF/s | by cost %.4f ms %.1f (cut %s split %s)' % (x['by_cost']['ms'], x['by_cost']['tflops'], d['ms'], d['tflops'], d['cut'], d['split']))
"
