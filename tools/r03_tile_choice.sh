#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03tile
mkdir -p $OUT
cd $R
for dt in f64 f32; do
for t in 64 128 0; do
if [ $t = 0 ]; then unset CTC_TC_FORCE_TILE; else export CTC_TC_FORCE_TILE=$t; fi
timeout -k 10 300 python tools/extent_sweep.py $dt 24 28 32 36 40 44 48 52 56 > $OUT/sweep_${dt}_$t.jsonl 2> $OUT/sweep.err || { tail -5 $OUT/sweep.err; exit 1; }
done
python3 -c "
import json
rd = lambda t: [json.loads(l) for l in open('$OUT/sweep_${dt}_%s.jsonl' % t)]
for a, b, c in zip(rd(64), rd(128), rd(0)):
    f = lambda r: '%.2f (cut %s split %s shares %s)' % (r['by_cost']['TFLOP/s'], r['by_cost']['tail_tiles'], r['by_cost']['split'], r['by_cost']['shares'])
    print('$dt', a['extent'], 'tile 64:', f(a), '| tile 128:', f(b), '| planner', c['tile'][:2], f(c))
"
done
