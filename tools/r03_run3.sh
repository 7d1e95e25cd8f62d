#!/bin/bash
# counters of the config-4 slab on the persistent kernel, regular build and the diagnostic build whose
# A tile is fetched once per workgroup (wrong results; two thirds of the L2 -> LDS bytes gone)
set -o pipefail
OUT=$PWD/gpurun_out/r03c
mkdir -p $OUT
bash tools/pmc_case.sh six > $OUT/pmc_six_regular.txt 2>&1 || { tail -20 $OUT/pmc_six_regular.txt; exit 1; }
CTC_TC_LIB=$PWD/ctc_amd/lib/variants/libctc_tc_a_once.so bash tools/pmc_case.sh six > $OUT/pmc_six_a_once.txt 2>&1 || { tail -20 $OUT/pmc_six_a_once.txt; exit 1; }
cat $OUT/pmc_six_regular.txt $OUT/pmc_six_a_once.txt
