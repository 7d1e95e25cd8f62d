#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03tail
mkdir -p $OUT
cd $R
timeout -k 10 900 python -u -m pytest tests/test_contract_gpu.py tests/test_zz_no_alloc.py tests/test_zz_no_alloc_graph.py tests/test_threads_gpu.py tests/test_contract_fuzz_gpu.py tests/test_dma_gemm_gpu.py -x -q --timeout 120 --timeout-method thread > $OUT/pytest2.log 2>&1 || { tail -40 $OUT/pytest2.log; exit 1; }
tail -3 $OUT/pytest2.log
TAIL_PROBE_SMALL=1 CTC_TC_FORCE_TILE=64 timeout -k 10 300 python tools/tail_cost_probe.py > $OUT/small_modes_new.jsonl 2> $OUT/sweep.err || { tail -5 $OUT/sweep.err; exit 1; }
for dt in f64 f32; do
timeout -k 10 300 python tools/extent_sweep.py $dt 16 20 24 28 32 36 40 44 > $OUT/small_cut_new_$dt.jsonl 2> $OUT/sweep.err || { tail -5 $OUT/sweep.err; exit 1; }
done
