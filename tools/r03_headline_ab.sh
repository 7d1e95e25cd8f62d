#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03hab
mkdir -p $OUT
cd $R
for rep in 1 2 3; do
timeout -k 10 200 python bench.py --no-cpu --no-variants --steps 20 --warmup 5 > $OUT/light_$rep.json 2> $OUT/err || { tail -5 $OUT/err; exit 1; }
CTC_TC_NO_LIGHT_ROUNDS=1 timeout -k 10 200 python bench.py --no-cpu --no-variants --steps 20 --warmup 5 > $OUT/nolight_$rep.json 2> $OUT/err || { tail -5 $OUT/err; exit 1; }
python3 -c "
import json
a = json.load(open('$OUT/light_$rep.json')); b = json.load(open('$OUT/nolight_$rep.json'))
print('rep $rep: light', a['value'], a['roofline']['frac'], '| no light', b['value'], b['roofline']['frac'])
"
done
