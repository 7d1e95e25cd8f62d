#!/bin/bash
# what the driver runs at the end of a round: the GPU suite, smoke(), bench.py with its defaults
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03z
mkdir -p $OUT
cd $R
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $OUT/pytest_gpu.log 2>&1 || { tail -40 $OUT/pytest_gpu.log; exit 1; }
tail -3 $OUT/pytest_gpu.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { tail -20 $OUT/smoke.log; exit 1; }
tail -1 $OUT/smoke.log
timeout -k 10 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
python3 -c "import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['verified'], d['cpu_baseline']['value'], list(d['config']['variants']))"
