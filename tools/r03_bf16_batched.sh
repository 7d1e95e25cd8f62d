#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03bf16b
mkdir -p $OUT
cd $R
timeout -k 10 600 python -u -m pytest tests/test_contract_bf16_gpu.py tests/test_zz_no_alloc.py tests/test_padded_copies_gpu.py -x -q --timeout 120 --timeout-method thread > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
export BATCHED_BF16=1
timeout -k 10 300 python tools/batched_tail_probe.py > $OUT/new.jsonl 2> $OUT/new.err || { tail -5 $OUT/new.err; exit 1; }
CTC_TC_NO_FLAT_BATCH=1 timeout -k 10 300 python tools/batched_tail_probe.py > $OUT/old.jsonl 2> $OUT/old.err || { tail -5 $OUT/old.err; exit 1; }
python3 - <<'PY'
import json
rd = lambda f: [json.loads(l) for l in open("gpurun_out/r03bf16b/%s.jsonl" % f)]
for n, o in zip(rd("new"), rd("old")):
    print(n["dtype"], n["batch"], n["m"], n["n"], n["k"], "old %s %.0f | new %s %.0f TF/s (tiles %d whole %d cut %d split %d)  x%.3f" % (
        o["tile"], o["tflops"], n["tile"], n["tflops"], n["tiles"], n["whole"], n["cut"], n["split"], n["tflops"] / o["tflops"]))
PY
timeout -k 10 300 python tools/vendor_gemm_cmp.py bf16 > $OUT/vendor_bf16.jsonl 2> $OUT/vendor.err || { tail -5 $OUT/vendor.err; exit 1; }
grep libctc_tc_over_torch $OUT/vendor_bf16.jsonl
