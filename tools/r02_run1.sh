#!/bin/bash
# full GPU test suite, the no-alloc WIP test, bench, and the one-workgroup-per-CU probe
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/r1_pytest.log 2>&1
rc=$?; tail -5 gpurun_out/r1_pytest.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -u -m pytest tests/wip_no_alloc.py -x -v --timeout 120 --timeout-method thread > gpurun_out/r1_noalloc.log 2>&1
rc=$?; tail -15 gpurun_out/r1_noalloc.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py > gpurun_out/r1_bench.json 2> gpurun_out/r1_bench.err || exit 1
cat gpurun_out/r1_bench.json
# classic kernel, one workgroup per CU (extra dynamic LDS), no split: does ONE wave per SIMD keep the pipe busy?
for e in 0 50000; do
  CTC_TC_NO_SPLITK=1 CTC_TC_DEBUG_EXTRA_LDS=$e timeout -k 10 120 python bench.py --extent 64 --no-cpu --no-verify --steps 20 > gpurun_out/r1_onewg_$e.json 2>&1 || exit 1
  python - <<PY
import json
d = json.loads(open("gpurun_out/r1_onewg_$e.json").read().strip().splitlines()[-1])
print("extra_lds", $e, "TFLOP/s", d["value"], "kernel_ms", d["roofline"]["kernel_ms"])
PY
done
