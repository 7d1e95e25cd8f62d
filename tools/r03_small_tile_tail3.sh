#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03tail
mkdir -p $OUT
cd $R
TAIL_PROBE_SMALL=1 CTC_TC_FORCE_TILE=64 timeout -k 10 300 python tools/tail_cost_probe.py f32 > $OUT/small_modes_new_f32.jsonl 2> $OUT/sweep.err || { tail -5 $OUT/sweep.err; exit 1; }
for dt in f64 f32; do
timeout -k 10 300 python tools/extent_sweep.py $dt 16 20 24 28 32 36 40 44 > $OUT/small_cut_new_$dt.jsonl 2> $OUT/sweep.err || { tail -5 $OUT/sweep.err; exit 1; }
done
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $OUT/pytest_gpu.log 2>&1 || { tail -40 $OUT/pytest_gpu.log; exit 1; }
tail -3 $OUT/pytest_gpu.log
