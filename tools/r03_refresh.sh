#!/bin/bash
# refresh of the evidence the scheduler / tile-choice changes touch, on the final build
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03refresh
mkdir -p $OUT
cd $R
timeout -k 10 600 python tools/slab_rehearsal.py > $OUT/slab_rehearsal.jsonl 2> $OUT/slab.err || { tail -5 $OUT/slab.err; exit 1; }
echo slab ok
timeout -k 10 300 python tools/extent_sweep.py f64 > $OUT/extent_sweep.jsonl 2> $OUT/sweep.err || { tail -5 $OUT/sweep.err; exit 1; }
timeout -k 10 300 python tools/extent_sweep.py f32 > $OUT/extent_sweep_f32.jsonl 2> $OUT/sweep.err || { tail -5 $OUT/sweep.err; exit 1; }
echo sweeps ok
timeout -k 10 300 python tools/batch_probe.py > $OUT/batch_probe.jsonl 2> $OUT/bp.err || { tail -5 $OUT/bp.err; exit 1; }
timeout -k 10 300 python tools/small_batched_probe.py > $OUT/small_batched.jsonl 2> $OUT/sb.err || { tail -5 $OUT/sb.err; exit 1; }
BATCHED_SMALL=1 timeout -k 10 300 python tools/batched_tail_probe.py > $OUT/batched_small_new.jsonl 2> $OUT/bs.err || { tail -5 $OUT/bs.err; exit 1; }
BATCHED_SMALL=1 CTC_TC_NO_FLAT_BATCH=1 CTC_TC_NO_BATCHED_CHUNKS=1 timeout -k 10 300 python tools/batched_tail_probe.py > $OUT/batched_small_old.jsonl 2> $OUT/bs.err || { tail -5 $OUT/bs.err; exit 1; }
timeout -k 10 300 python tools/batched_tail_probe.py > $OUT/batched_tails_new.jsonl 2> $OUT/bt.err || { tail -5 $OUT/bt.err; exit 1; }
CTC_TC_NO_FLAT_BATCH=1 CTC_TC_NO_BATCHED_CHUNKS=1 timeout -k 10 300 python tools/batched_tail_probe.py > $OUT/batched_tails_old.jsonl 2> $OUT/bt.err || { tail -5 $OUT/bt.err; exit 1; }
echo probes ok
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_bench_headline
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench_headline -o bench -- python3 $R/bench.py --no-cpu --no-variants > $OUT/bench_headline_line.json 2> $OUT/bench_headline.err || { tail -5 $OUT/bench_headline.err; exit 1; }
find $R/gpurun_out/prof_bench_headline -name "*kernel_stats.csv" -exec cp {} $OUT/bench_headline_kernel_stats.csv \;
head -4 $OUT/bench_headline_kernel_stats.csv; cut -c1-300 $OUT/bench_headline_line.json
