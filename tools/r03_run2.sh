#!/bin/bash
# what the config-4 slab gains when the A tile stops travelling L2 -> LDS per item (diagnostic build, wrong results)
set -o pipefail
OUT=gpurun_out/r03b
mkdir -p $OUT
timeout -k 10 120 python -u tools/stream_cmp.py slab > $OUT/slab_regular.jsonl 2> $OUT/slab_regular.err || { tail -20 $OUT/slab_regular.err; exit 1; }
CTC_TC_LIB=$PWD/ctc_amd/lib/variants/libctc_tc_a_once.so timeout -k 10 120 python -u tools/stream_cmp.py slab > $OUT/slab_a_once.jsonl 2> $OUT/slab_a_once.err || { tail -20 $OUT/slab_a_once.err; exit 1; }
cat $OUT/slab_regular.jsonl $OUT/slab_a_once.jsonl
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $OUT/pytest_gpu.log 2>&1 || { tail -40 $OUT/pytest_gpu.log; exit 1; }
tail -3 $OUT/pytest_gpu.log
