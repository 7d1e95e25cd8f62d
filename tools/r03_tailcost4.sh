#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03tail
mkdir -p $OUT
cd $R
timeout -k 10 300 python tools/batched_tail_probe.py > $OUT/batched_new4.jsonl 2> $OUT/new.err || { tail -5 $OUT/new.err; exit 1; }
cut -c1-230 $OUT/batched_new4.jsonl
for dt in f64 f32; do
timeout -k 10 500 python tools/extent_sweep.py $dt 36 40 44 48 52 56 60 64 68 72 76 80 84 88 92 96 > $OUT/extent_sweep4_$dt.jsonl 2> $OUT/sweep.err || { tail -5 $OUT/sweep.err; exit 1; }
python3 -c "
import json
for l in open('$OUT/extent_sweep4_$dt.jsonl'):
    r = json.loads(l); f = lambda d: '%.1f (whole %s cut %s split %s shares %s)' % (d['TFLOP/s'], d['whole_tiles'], d['tail_tiles'], d['split'], d['shares'])
    print('$dt', r['extent'], r['tile'][:2], 'chunks', f(r['equal_chunks']), '| shares', f(r['shares_forced']), '| by cost', f(r['by_cost']))
"
done
