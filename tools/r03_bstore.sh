#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03bstore
mkdir -p $OUT
cd $R
export BSTORE_SWEEP=1
CTC_TC_C_STREAM=0 timeout -k 10 300 python tools/batched_store_probe.py > $OUT/sweep_plain.jsonl 2> $OUT/plain.err || { tail -5 $OUT/plain.err; exit 1; }
CTC_TC_C_STREAM=1 timeout -k 10 300 python tools/batched_store_probe.py > $OUT/sweep_hint.jsonl 2> $OUT/hint.err || { tail -5 $OUT/hint.err; exit 1; }
CTC_TC_C_STREAM=0 timeout -k 10 300 python tools/batched_store_probe.py > $OUT/sweep_plain2.jsonl 2> $OUT/plain.err || { tail -5 $OUT/plain.err; exit 1; }
python3 - <<'PY'
import json
rd = lambda f: [json.loads(l) for l in open("gpurun_out/r03bstore/%s.jsonl" % f)]
for p, h, p2 in zip(rd("sweep_plain"), rd("sweep_hint"), rd("sweep_plain2")):
    tot = p["C_MiB"] * (p["m"] * p["k"] + p["k"] * p["n"] + p["m"] * p["n"]) // (p["m"] * p["n"])
    print(p["dtype"], p["batch"], p["m"], p["n"], p["k"], "C MiB", p["C_MiB"], "total", tot, "plain %.4f (again %.4f) hint %.4f ms  hint/plain time %.3f" % (p["ms"], p2["ms"], h["ms"], h["ms"] / min(p["ms"], p2["ms"])))
PY
