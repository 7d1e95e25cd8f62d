set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/test_permute_gpu.py -m gpu -x -q --timeout 180 --timeout-method thread 2>&1 | tail -3
echo "== default"; timeout -k 10 120 python tools/perm_probe.py
for l in 0 1; do for s in 0 1; do echo "== NTL$l NTS$s"; CTC_PERM_NTL=$l CTC_PERM_NTS=$s timeout -k 10 120 python tools/perm_probe.py; done; done
