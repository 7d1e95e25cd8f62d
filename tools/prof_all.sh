# Kernel-trace statistics (rocprofv3 --kernel-trace --stats) of the headline bench and of the
# secondary measurements; results under gpurun_out/prof_r01c_*, to be copied into profiles/.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_r01c_bench $R/gpurun_out/prof_r01c_configs
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01c_bench -o bench -- python3 $R/bench.py --no-cpu > $R/gpurun_out/prof_r01c_bench.log 2>&1 || { tail -5 $R/gpurun_out/prof_r01c_bench.log; exit 1; }
tail -1 $R/gpurun_out/prof_r01c_bench.log | cut -c1-160
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01c_configs -o configs -- python3 $R/benchmarks/run_configs.py permute adversarial f32 bf16 six small fastr sw > $R/gpurun_out/prof_r01c_configs.log 2>&1 || { tail -5 $R/gpurun_out/prof_r01c_configs.log; exit 1; }
find $R/gpurun_out/prof_r01c_bench $R/gpurun_out/prof_r01c_configs -name "*kernel_stats.csv"
