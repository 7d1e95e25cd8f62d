#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03pad
mkdir -p $OUT
cd $R
timeout -k 10 600 python -u -m pytest tests/test_padded_copies_gpu.py tests/test_contract_gpu.py tests/test_contract_fuzz_gpu.py -x -q --timeout 120 --timeout-method thread > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
timeout -k 10 400 python tools/pad_small_probe.py > $OUT/rule.jsonl 2> $OUT/err || { tail -5 $OUT/err; exit 1; }
timeout -k 10 500 python tools/random_shapes_ab.py 400 4242 > $OUT/rand_rule.jsonl 2> $OUT/err || { tail -5 $OUT/err; exit 1; }
CTC_TC_PAD_MIN_MN=1024 timeout -k 10 500 python tools/random_shapes_ab.py 400 4242 > $OUT/rand_1024.jsonl 2> $OUT/err || { tail -5 $OUT/err; exit 1; }
