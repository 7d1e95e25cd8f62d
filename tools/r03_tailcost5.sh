#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03tail
mkdir -p $OUT
cd $R
for rep in 1 2; do
for dt in f64 f32; do
timeout -k 10 300 python tools/extent_sweep.py $dt 56 72 76 84 88 96 > $OUT/ab_light_${dt}_$rep.jsonl 2> $OUT/sweep.err || { tail -5 $OUT/sweep.err; exit 1; }
CTC_TC_NO_LIGHT_ROUNDS=1 timeout -k 10 300 python tools/extent_sweep.py $dt 56 72 76 84 88 96 > $OUT/ab_nolight_${dt}_$rep.jsonl 2> $OUT/sweep.err || { tail -5 $OUT/sweep.err; exit 1; }
python3 -c "
import json
a = [json.loads(l) for l in open('$OUT/ab_light_${dt}_$rep.jsonl')]; b = [json.loads(l) for l in open('$OUT/ab_nolight_${dt}_$rep.jsonl')]
for x, y in zip(a, b):
    f = lambda d: '%.2f (cut %s split %s shares %s)' % (d['TFLOP/s'], d['tail_tiles'], d['split'], d['shares'])
    print('$dt', x['extent'], 'light', f(x['by_cost']), '| chunks', x['equal_chunks']['TFLOP/s'], '|| no light', f(y['by_cost']), '| chunks', y['equal_chunks']['TFLOP/s'])
"
done
done
