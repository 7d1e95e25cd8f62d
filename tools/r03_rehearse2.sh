#!/bin/bash
# bench.py with two ranks over gloo sharing the one GPU, default CPU sample (what a launcher that sets one OpenMP
# thread per worker gets), then the kernel trace of the headline
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03i
mkdir -p $OUT
cd $R
CTC_BENCH_BACKEND=gloo OMP_NUM_THREADS=1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 2 --steps 5 --warmup 2 > $OUT/bench_gloo2_omp1.json 2> $OUT/bench_gloo2_omp1.err
echo "gloo2 (one OpenMP thread) exit $?"; python3 -c "import json; d=json.load(open('$OUT/bench_gloo2_omp1.json')); print(d['value'], d['n_gpus'], d['cpu_baseline'], list(d['config']['variants']))"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_bench_headline
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench_headline -o bench -- python3 $R/bench.py --no-cpu --no-variants > $OUT/bench_headline_line.json 2> $OUT/bench_headline.err || { tail -5 $OUT/bench_headline.err; exit 1; }
find $R/gpurun_out/prof_bench_headline -name "*kernel_stats.csv" -exec cp {} $OUT/bench_headline_kernel_stats.csv \;
head -4 $OUT/bench_headline_kernel_stats.csv; cut -c1-200 $OUT/bench_headline_line.json
