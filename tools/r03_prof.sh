#!/bin/bash
# HBM-side traffic of the headline kernel (two counter passes), kernel-trace statistics of
# bench.py (headline alone, then with its variants), instruction / busy counters of the headline kernel
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03g
mkdir -p $OUT
bash $R/tools/pmc_traffic.sh 96 > $OUT/pmc_traffic.log 2>&1 || { tail -20 $OUT/pmc_traffic.log; exit 1; }
python3 $R/tools/traffic_summary.py $R/gpurun_out/traffic_FETCH_SIZE $R/gpurun_out/traffic_WRITE_SIZE $OUT/r03_bench_traffic.json 96 || exit 1
find $R/gpurun_out/prof_bench -name "*kernel_stats.csv" -exec cp {} $OUT/bench_headline_kernel_stats.csv \;
tail -1 $R/gpurun_out/prof_bench.log > $OUT/bench_headline_line.json
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_bench_all
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench_all -o bench -- python3 $R/bench.py --no-cpu > $OUT/bench_all_line.json 2> $OUT/bench_all.err || { tail -5 $OUT/bench_all.err; exit 1; }
find $R/gpurun_out/prof_bench_all -name "*kernel_stats.csv" -exec cp {} $OUT/bench_all_kernel_stats.csv \;
cd $R
bash $R/tools/pmc_case.sh ccsd96 > $OUT/pmc_ccsd96.txt 2>&1 || { tail -20 $OUT/pmc_ccsd96.txt; exit 1; }
head -5 $OUT/bench_headline_kernel_stats.csv
cat $OUT/pmc_ccsd96.txt
