#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03tail
mkdir -p $OUT
cd $R
export TAIL_PROBE_ODDK=1 CTC_TC_FORCE_TILE=128 CTC_TC_NO_PAD_TRANSPOSES=1
CTC_TC_NO_SPLITK=1 timeout -k 10 300 python tools/tail_cost_probe.py > $OUT/oddk_nocut.jsonl 2> $OUT/sweep.err || { tail -5 $OUT/sweep.err; exit 1; }
timeout -k 10 300 python tools/tail_cost_probe.py > $OUT/oddk_modes.jsonl 2> $OUT/sweep.err || { tail -5 $OUT/sweep.err; exit 1; }
CTC_TC_SPLIT_BY_SLOTS=1 timeout -k 10 300 python tools/tail_cost_probe.py > $OUT/oddk_slots.jsonl 2> $OUT/sweep.err || { tail -5 $OUT/sweep.err; exit 1; }
python3 -c "
import json
rd = lambda f: [json.loads(l) for l in open('$OUT/oddk_%s.jsonl' % f)]
for x, y, z in zip(rd('nocut'), rd('modes'), rd('slots')):
    d, e = y['by_cost'], z['by_cost']
    print(x['dtype'], x['tile'], 'k', x['k'], 'tiles', x['tiles'], 'never cut %.4f ms %.1f TF/s | by cost %.1f (cut %s split %s) | by slots %.1f (cut %s split %s)' % (x['by_cost']['ms'], x['by_cost']['tflops'], d['tflops'], d['cut'], d['split'], e['tflops'], e['cut'], e['split']))
"
