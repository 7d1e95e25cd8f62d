# permutation and contraction parity tests, then the permutation sweep and the fp64 probe
set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -u -m pytest tests/test_permute_gpu.py tests/test_contract_gpu.py tests/test_contract_fuzz_gpu.py -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/r02_perm_pytest.log 2>&1 || { tail -30 gpurun_out/r02_perm_pytest.log; exit 1; }
tail -2 gpurun_out/r02_perm_pytest.log
timeout -k 10 300 python tools/permute_sweep.py > gpurun_out/r02_permute_sweep.jsonl
head -14 gpurun_out/r02_permute_sweep.jsonl; grep "128, 128, 128\|4096, 4096" gpurun_out/r02_permute_sweep.jsonl || true
timeout -k 10 300 python benchmarks/run_configs.py permute adversarial > gpurun_out/r02_perm_configs.jsonl 2>&1
cut -c1-170 gpurun_out/r02_perm_configs.jsonl
