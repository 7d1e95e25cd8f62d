#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03rand
mkdir -p $OUT
cd $R
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $OUT/pytest_gpu.log 2>&1 || { tail -40 $OUT/pytest_gpu.log; exit 1; }
tail -3 $OUT/pytest_gpu.log
timeout -k 10 500 python tools/random_shapes_ab.py 500 9001 > $OUT/pad110_now.jsonl 2> $OUT/now.err || { tail -5 $OUT/now.err; exit 1; }
