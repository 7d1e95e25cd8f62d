# Evidence after the bf16 ring kernel took outer-major operands, in one GPU call: kernel-trace statistics of the
# secondary measurements, the ring / two-buffer comparison, the LDS read probe and the SQ / GRBM / TCC counter
# passes of three bf16 cases.  Results under gpurun_out/r02c_*; the summaries are copied into profiles/ by hand.
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r02c_configs
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02c_configs -o configs -- python3 $R/benchmarks/run_configs.py permute adversarial f32 bf16 six small fastr sw > $R/gpurun_out/r02c_configs.log 2>&1 || { tail -5 $R/gpurun_out/r02c_configs.log; exit 1; }
find $R/gpurun_out/r02c_configs -name "*kernel_stats.csv"
for c in nt8k_bf16 nn8k_bf16 ccsd96_bf16; do
  bash $R/tools/pmc_case.sh $c > $R/gpurun_out/r02c_pmc_$c.txt 2>&1 || { tail -5 $R/gpurun_out/r02c_pmc_$c.txt; exit 1; }
done
cd $R
timeout -k 10 200 python tools/bf16_ring_cmp.py > gpurun_out/r02c_bf16_ring.jsonl 2>/dev/null
CTC_TC_NO_BF16_RING_OUTER=1 timeout -k 10 200 python tools/bf16_ring_cmp.py > gpurun_out/r02c_bf16_two_buffer.jsonl 2>/dev/null
timeout -k 10 60 ./tools/lds_read_probe > gpurun_out/r02c_lds_read_probe.txt
tail -3 gpurun_out/r02c_bf16_ring.jsonl
