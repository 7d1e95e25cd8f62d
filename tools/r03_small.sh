#!/bin/bash
# batches of small products on the grouped kernel: parity tests, the probes before/after
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03small
mkdir -p $OUT
cd $R
timeout -k 10 300 python -u -m pytest tests/test_small_batched_gpu.py tests/test_grouped_gpu.py tests/test_contract_gpu.py -x -q --timeout 120 --timeout-method thread > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
timeout -k 10 200 python tools/small_batched_probe.py > $OUT/small_batched_probe.jsonl 2> $OUT/small_batched_probe.err || { tail -5 $OUT/small_batched_probe.err; exit 1; }
cat $OUT/small_batched_probe.jsonl
timeout -k 10 200 python tools/batch_probe.py > $OUT/batch_probe.jsonl 2> $OUT/batch_probe.err || { tail -5 $OUT/batch_probe.err; exit 1; }
cat $OUT/batch_probe.jsonl
