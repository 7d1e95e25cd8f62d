# Kernel-trace statistics of the headline bench on the final build (alone, and with its variants), in one GPU call.
# Results under gpurun_out/r02c_bench*; the summaries are copied into profiles/ by hand.
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r02c_bench_headline $R/gpurun_out/r02c_bench
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02c_bench_headline -o bench -- python3 $R/bench.py --no-cpu --no-variants > $R/gpurun_out/r02c_bench_headline.log 2> $R/gpurun_out/r02c_bench_headline.err || { tail -5 $R/gpurun_out/r02c_bench_headline.err; exit 1; }
tail -1 $R/gpurun_out/r02c_bench_headline.log | cut -c1-300
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02c_bench -o bench -- python3 $R/bench.py --no-cpu > $R/gpurun_out/r02c_bench.log 2> $R/gpurun_out/r02c_bench.err || { tail -5 $R/gpurun_out/r02c_bench.err; exit 1; }
find $R/gpurun_out/r02c_bench_headline $R/gpurun_out/r02c_bench -name "*kernel_stats.csv"
