# in one GPU call: the whole -m gpu suite, then kernel-trace statistics of the headline
# bench and of the secondary measurements, then counter passes over the short-K slab (persistent kernel).
# Results under gpurun_out/r02_*; the summaries worth keeping are copied into profiles/ by hand.
set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/r02_pytest.log 2>&1 || { tail -30 gpurun_out/r02_pytest.log; exit 1; }
tail -2 gpurun_out/r02_pytest.log
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r02_bench $R/gpurun_out/r02_configs
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_bench -o bench -- python3 $R/bench.py --no-cpu > $R/gpurun_out/r02_bench.log 2> $R/gpurun_out/r02_bench.err || { tail -5 $R/gpurun_out/r02_bench.err; exit 1; }
tail -1 $R/gpurun_out/r02_bench.log | cut -c1-400
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_configs -o configs -- python3 $R/benchmarks/run_configs.py permute adversarial f32 bf16 six small fastr sw > $R/gpurun_out/r02_configs.log 2>&1 || { tail -5 $R/gpurun_out/r02_configs.log; exit 1; }
find $R/gpurun_out/r02_bench $R/gpurun_out/r02_configs -name "*kernel_stats.csv"
bash $R/tools/pmc_case.sh six > $R/gpurun_out/r02_pmc_six.txt 2>&1 || { tail -5 $R/gpurun_out/r02_pmc_six.txt; exit 1; }
tail -30 $R/gpurun_out/r02_pmc_six.txt
