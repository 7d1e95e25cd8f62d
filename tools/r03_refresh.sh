#!/bin/bash
# refresh of the evidence the scheduler change touches: rank slabs on one GPU, the headline under the kernel trace
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03refresh
mkdir -p $OUT
cd $R
timeout -k 10 600 python tools/slab_rehearsal.py > $OUT/slab_rehearsal.jsonl 2> $OUT/slab.err || { tail -5 $OUT/slab.err; exit 1; }
cut -c1-400 $OUT/slab_rehearsal.jsonl
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_bench_headline
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench_headline -o bench -- python3 $R/bench.py --no-cpu --no-variants > $OUT/bench_headline_line.json 2> $OUT/bench_headline.err || { tail -5 $OUT/bench_headline.err; exit 1; }
find $R/gpurun_out/prof_bench_headline -name "*kernel_stats.csv" -exec cp {} $OUT/bench_headline_kernel_stats.csv \;
head -4 $OUT/bench_headline_kernel_stats.csv; cut -c1-300 $OUT/bench_headline_line.json
