#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03tail
mkdir -p $OUT
cd $R
export TAIL_PROBE_BF16=1
timeout -k 10 300 python tools/tail_cost_probe.py > $OUT/bf16_cut.jsonl 2> $OUT/sweep.err || { tail -5 $OUT/sweep.err; exit 1; }
CTC_TC_NO_SPLITK=1 timeout -k 10 300 python tools/tail_cost_probe.py > $OUT/bf16_nocut.jsonl 2> $OUT/sweep.err || { tail -5 $OUT/sweep.err; exit 1; }
python3 -c "
import json
rd = lambda t: [json.loads(l) for l in open('$OUT/bf16_%s.jsonl' % t)]
for a, b in zip(rd('cut'), rd('nocut')):
    d = a['by_cost']
    print('bf16', a['tile'], 'k', a['k'], 'tiles', a['tiles'], 'by cost %.4f ms %.0f TF/s (cut %s split %s) | never cut %.4f ms %.0f' % (d['ms'], d['tflops'], d['cut'], d['split'], b['by_cost']['ms'], b['by_cost']['tflops']))
"
