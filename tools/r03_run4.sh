#!/bin/bash
# stream-K shares - parity first, then the extent sweep, the slab rehearsal and the bench line
set -o pipefail
OUT=$PWD/gpurun_out/r03d
mkdir -p $OUT
timeout -k 10 600 python -u -m pytest tests/test_dma_gemm_gpu.py tests/test_zz_no_alloc.py -x -q --timeout 120 --timeout-method thread > $OUT/test_dma.log 2>&1 || { tail -40 $OUT/test_dma.log; exit 1; }
tail -3 $OUT/test_dma.log
timeout -k 10 300 python -u tools/slab_rehearsal.py headline contracted > $OUT/slab_rehearsal.jsonl 2> $OUT/slab_rehearsal.err || { tail -20 $OUT/slab_rehearsal.err; exit 1; }
cat $OUT/slab_rehearsal.jsonl
timeout -k 10 600 python -u tools/extent_sweep.py > $OUT/extent_sweep.jsonl 2> $OUT/extent_sweep.err || { tail -20 $OUT/extent_sweep.err; exit 1; }
cat $OUT/extent_sweep.jsonl
timeout -k 10 300 python -u bench.py --no-variants > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
cat $OUT/bench.json
