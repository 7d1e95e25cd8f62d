#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03tile
mkdir -p $OUT
cd $R
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $OUT/pytest_gpu.log 2>&1 || { tail -40 $OUT/pytest_gpu.log; exit 1; }
tail -3 $OUT/pytest_gpu.log
export PROBE_CASES="1,2048,1536,1100;1,1792,2048,1280;1,1700,1900,1500;4,700,900,1100;2,1024,1536,1100;3,1000,1000,1200;1,2048,1536,1536;1,1792,2048,2048;4,700,900,1600;1,2000,1500,1030;6,512,512,1400;1,1536,1280,1100;1,1280,1280,1400;1,1536,1536,1600;1,1536,1536,2048;1,1600,1600,1600;1,1408,1408,1800;1,1664,1664,1664;1,1280,1664,1920;1,2304,1408,1280"
timeout -k 10 300 python tools/batched_tail_probe.py > $OUT/ab_planner4.jsonl 2> $OUT/ab.err || { tail -5 $OUT/ab.err; exit 1; }
for t in 64 128; do
CTC_TC_FORCE_TILE=$t timeout -k 10 300 python tools/batched_tail_probe.py > $OUT/ab4_$t.jsonl 2> $OUT/ab.err || { tail -5 $OUT/ab.err; exit 1; }
done
timeout -k 10 300 python tools/pad_tile_probe.py > $OUT/pad_planner4.jsonl 2> $OUT/pad.err || { tail -5 $OUT/pad.err; exit 1; }
for dt in f64 f32; do timeout -k 10 300 python tools/extent_sweep.py $dt 36 40 44 48 > $OUT/sweep4_$dt.jsonl 2> $OUT/sweep.err || { tail -5 $OUT/sweep.err; exit 1; }; done
python3 - <<'PY'
import json
rd = lambda f: [json.loads(l) for l in open("gpurun_out/r03tile/%s.jsonl" % f)]
for a, b, c in zip(rd("ab4_64"), rd("ab4_128"), rd("ab_planner4")):
    print(a["dtype"], a["batch"], a["m"], a["n"], a["k"], "tile 64: %.1f | tile 128: %.1f (tiles %d cut %d split %d shares %d) | planner %s %.1f  x%.2f of better" % (
        a["tflops"], b["tflops"], b["tiles"], b["cut"], b["split"], b["shares"], c["tile"], c["tflops"], c["tflops"] / max(a["tflops"], b["tflops"])))
for dt in ("f64", "f32"):
    print(dt, [(r["extent"], r["tile"][:2], r["TFLOP/s"]) for r in rd("sweep4_%s" % dt)])
PY
