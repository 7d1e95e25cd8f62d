# kernel-trace statistics of the headline bench alone (no variants), counters of the bf16 kernels
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r02_bench_headline
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_bench_headline -o bench -- python3 $R/bench.py --no-cpu --no-variants > $R/gpurun_out/r02_bench_headline.log 2> $R/gpurun_out/r02_bench_headline.err || { tail -5 $R/gpurun_out/r02_bench_headline.err; exit 1; }
tail -1 $R/gpurun_out/r02_bench_headline.log | cut -c1-200
head -4 $R/gpurun_out/r02_bench_headline/bench_kernel_stats.csv
bash $R/tools/pmc_case.sh nt8k_bf16 > $R/gpurun_out/r02_pmc_bf16_nt8k.txt 2>&1 || { tail -5 $R/gpurun_out/r02_pmc_bf16_nt8k.txt; exit 1; }
bash $R/tools/pmc_case.sh ccsd64_bf16 > $R/gpurun_out/r02_pmc_bf16_ccsd64.txt 2>&1 || { tail -5 $R/gpurun_out/r02_pmc_bf16_ccsd64.txt; exit 1; }
grep -h "GRBM_GUI\|MFMA_BUSY\|INSTS_MFMA\|LDS_IDX\|BANK_CONFLICT\|TCC_" $R/gpurun_out/r02_pmc_bf16_nt8k.txt $R/gpurun_out/r02_pmc_bf16_ccsd64.txt
