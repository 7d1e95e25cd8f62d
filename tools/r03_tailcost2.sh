#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03tail
mkdir -p $OUT
cd $R
timeout -k 10 600 python -u -m pytest tests/test_dma_gemm_gpu.py tests/test_contract_gpu.py tests/test_zz_no_alloc.py -x -q --timeout 120 --timeout-method thread > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
timeout -k 10 300 python tools/batched_tail_probe.py > $OUT/batched_new3.jsonl 2> $OUT/new.err || { tail -5 $OUT/new.err; exit 1; }
timeout -k 10 500 python tools/tail_cost_probe.py > $OUT/modes3.jsonl 2> $OUT/modes.err || { tail -5 $OUT/modes.err; exit 1; }
TAIL_PROBE_HALF=1 timeout -k 10 500 python tools/tail_cost_probe.py > $OUT/half_modes3.jsonl 2> $OUT/modes.err || { tail -5 $OUT/modes.err; exit 1; }
