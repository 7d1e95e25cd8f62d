# after the LDS-DMA GEMM kernel and the two-workgroup persistent geometry, in one GPU call:
# kernel-trace statistics of the headline bench (alone, and with its variants), of the secondary
# measurements, the HBM-side traffic passes and the SQ / GRBM counter passes of both fp64 kernels.
# Results under gpurun_out/r02b_*; the summaries worth keeping are copied into profiles/ by hand.
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r02b_bench_headline $R/gpurun_out/r02b_bench $R/gpurun_out/r02b_configs
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02b_bench_headline -o bench -- python3 $R/bench.py --no-cpu --no-variants > $R/gpurun_out/r02b_bench_headline.log 2> $R/gpurun_out/r02b_bench_headline.err || { tail -5 $R/gpurun_out/r02b_bench_headline.err; exit 1; }
tail -1 $R/gpurun_out/r02b_bench_headline.log | cut -c1-300
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02b_bench -o bench -- python3 $R/bench.py --no-cpu > $R/gpurun_out/r02b_bench.log 2> $R/gpurun_out/r02b_bench.err || { tail -5 $R/gpurun_out/r02b_bench.err; exit 1; }
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02b_configs -o configs -- python3 $R/benchmarks/run_configs.py permute adversarial f32 bf16 six small fastr sw > $R/gpurun_out/r02b_configs.log 2>&1 || { tail -5 $R/gpurun_out/r02b_configs.log; exit 1; }
find $R/gpurun_out/r02b_bench_headline $R/gpurun_out/r02b_bench $R/gpurun_out/r02b_configs -name "*kernel_stats.csv"
bash $R/tools/pmc_traffic.sh 96 > $R/gpurun_out/r02b_traffic.log 2>&1 || { tail -5 $R/gpurun_out/r02b_traffic.log; exit 1; }
python3 $R/tools/traffic_summary.py $R/gpurun_out/traffic_FETCH_SIZE $R/gpurun_out/traffic_WRITE_SIZE $R/gpurun_out/r02b_bench_traffic.json
bash $R/tools/pmc_case.sh ccsd96 > $R/gpurun_out/r02b_pmc_ccsd96.txt 2>&1 || { tail -5 $R/gpurun_out/r02b_pmc_ccsd96.txt; exit 1; }
bash $R/tools/pmc_case.sh six > $R/gpurun_out/r02b_pmc_six.txt 2>&1 || { tail -5 $R/gpurun_out/r02b_pmc_six.txt; exit 1; }
tail -12 $R/gpurun_out/r02b_pmc_ccsd96.txt
cd $R
timeout -k 10 300 python tools/extent_sweep.py f64 > gpurun_out/r02b_extent_sweep_f64.jsonl 2>/dev/null
timeout -k 10 300 python tools/extent_sweep.py f32 > gpurun_out/r02b_extent_sweep_f32.jsonl 2>/dev/null
timeout -k 10 300 python tools/stream_cmp.py all > gpurun_out/r02b_stream_cmp.jsonl 2>/dev/null
