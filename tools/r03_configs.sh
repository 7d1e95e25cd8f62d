#!/bin/bash
# every secondary measurement on the final build, unprofiled (event-timed lines) and under the kernel
# trace (per-kernel durations), plus the fp32 extent sweep
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03i
mkdir -p $OUT
cd $R
timeout -k 10 500 python -u benchmarks/run_configs.py permute adversarial f32 bf16 six small fastr sw > $OUT/configs_results_unprofiled.jsonl 2> $OUT/configs_unprofiled.err || { tail -20 $OUT/configs_unprofiled.err; exit 1; }
timeout -k 10 300 python -u tools/extent_sweep.py f32 > $OUT/extent_sweep_f32.jsonl 2> $OUT/extent_sweep_f32.err || { tail -20 $OUT/extent_sweep_f32.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r03_configs
timeout -s KILL 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_configs -o configs -- python3 $R/benchmarks/run_configs.py permute adversarial f32 bf16 six small fastr sw > $OUT/configs_results_traced.jsonl 2> $OUT/configs_traced.err || { tail -5 $OUT/configs_traced.err; exit 1; }
find $R/gpurun_out/r03_configs -name "*kernel_stats.csv" -exec cp {} $OUT/configs_kernel_stats.csv \;
cut -c1-260 $OUT/configs_results_unprofiled.jsonl
