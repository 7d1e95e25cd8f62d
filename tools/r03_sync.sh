#!/bin/bash
# item-walking workgroups of an XCD meet after every round (bounded wait) - time and HBM-side traffic
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03k
mkdir -p $OUT
cd $R
V=$R/ctc_amd/lib/variants/libctc_tc_round_sync.so
CTC_TC_LIB=$V timeout -k 10 300 python -u -m pytest tests/test_contract_gpu.py -x -q -k "extent_96" --timeout 200 --timeout-method thread > $OUT/test.log 2>&1 || { tail -20 $OUT/test.log; exit 1; }
tail -1 $OUT/test.log
for lib in regular sync; do
  if [ $lib = sync ]; then export CTC_TC_LIB=$V; else unset CTC_TC_LIB; fi
  timeout -k 10 200 python -u bench.py --no-cpu --no-variants --steps 20 > $OUT/bench_$lib.json 2> $OUT/bench_$lib.err || { tail $OUT/bench_$lib.err; exit 1; }
  python3 -c "import json; d=json.load(open('$OUT/bench_$lib.json')); print('$lib', d['value'], d['roofline']['kernel_ms'], d['roofline']['kernel_ms_median'])"
done
export CTC_TC_LIB=$V
bash $R/tools/pmc_traffic.sh 96 > $OUT/pmc_traffic.log 2>&1 || { tail -20 $OUT/pmc_traffic.log; exit 1; }
python3 $R/tools/traffic_summary.py $R/gpurun_out/traffic_FETCH_SIZE $R/gpurun_out/traffic_WRITE_SIZE $OUT/traffic_sync.json 96
bash $R/tools/pmc_case.sh ccsd96 > $OUT/pmc_ccsd96_sync.txt 2>&1 || { tail -20 $OUT/pmc_ccsd96_sync.txt; exit 1; }
grep -E "GRBM_GUI_ACTIVE|MFMA_BUSY|TCC_MISS|TCC_HIT" $OUT/pmc_ccsd96_sync.txt
