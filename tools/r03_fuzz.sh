#!/bin/bash
# one-off campaign: the seeded random tests with other seeds and more cases than the suite runs
set -o pipefail
OUT=$PWD/gpurun_out/r03z
mkdir -p $OUT
for seed in ${SEEDS:-11 12}; do
  CTC_FUZZ_SEED=$seed CTC_FUZZ_CASES=${CASES:-1500} CTC_SK_FUZZ_SEED=$seed CTC_SK_FUZZ_CASES=${SK_CASES:-160} \
    timeout -k 10 1000 python -u -m pytest tests/test_contract_fuzz_gpu.py tests/test_dma_gemm_gpu.py -m gpu -x -q --timeout 900 --timeout-method thread \
    -k "random" > $OUT/fuzz_seed$seed.log 2>&1 || { tail -40 $OUT/fuzz_seed$seed.log; exit 1; }
  tail -2 $OUT/fuzz_seed$seed.log
done
