#!/bin/bash
# routing tests, slab rehearsal, routing sweep, bench line
set -o pipefail
OUT=gpurun_out/r03a
mkdir -p $OUT
timeout -k 10 300 python -u -m pytest tests/test_stream_gpu.py -x -q --timeout 120 --timeout-method thread > $OUT/test_stream.log 2>&1 || { tail -30 $OUT/test_stream.log; exit 1; }
tail -2 $OUT/test_stream.log
timeout -k 10 240 python -u tools/slab_rehearsal.py > $OUT/slab_rehearsal.jsonl 2> $OUT/slab_rehearsal.err || { tail -20 $OUT/slab_rehearsal.err; exit 1; }
cat $OUT/slab_rehearsal.jsonl
timeout -k 10 240 python -u tools/stream_cmp.py routing > $OUT/stream_routing.jsonl 2> $OUT/stream_routing.err || { tail -20 $OUT/stream_routing.err; exit 1; }
cat $OUT/stream_routing.jsonl
timeout -k 10 300 python -u bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
cat $OUT/bench.json
