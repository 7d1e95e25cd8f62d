#!/bin/bash
# bench.py rehearsals on one GPU - two ranks over gloo sharing the card (strong-scaling path, variants),
# the watchdog (time-out on record, non-zero exit), and the kernel trace of the headline alone
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03h
mkdir -p $OUT
cd $R
CTC_BENCH_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --cpu-extent 64 > $OUT/bench_gloo2.json 2> $OUT/bench_gloo2.err
echo "gloo2 exit $?"; cut -c1-1500 $OUT/bench_gloo2.json
timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu --variant-timeout 0.5 > $OUT/bench_watchdog.json 2> $OUT/bench_watchdog.err
echo "watchdog exit $?"; python3 -c "import json,sys; d=json.load(open('$OUT/bench_watchdog.json')); print(d['config']['variants'])"
CTC_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu --variant-timeout 1 > $OUT/bench_gloo2_watchdog.json 2> $OUT/bench_gloo2_watchdog.err
echo "gloo2 watchdog exit $?"; python3 -c "import json,sys; d=json.load(open('$OUT/bench_gloo2_watchdog.json')); print(d['n_gpus'], d['config']['variants'])"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_bench_headline
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench_headline -o bench -- python3 $R/bench.py --no-cpu --no-variants > $OUT/bench_headline_line.json 2> $OUT/bench_headline.err || { tail -5 $OUT/bench_headline.err; exit 1; }
find $R/gpurun_out/prof_bench_headline -name "*kernel_stats.csv" -exec cp {} $OUT/bench_headline_kernel_stats.csv \;
head -4 $OUT/bench_headline_kernel_stats.csv; cut -c1-200 $OUT/bench_headline_line.json
