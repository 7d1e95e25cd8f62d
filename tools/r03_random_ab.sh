#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03rand
mkdir -p $OUT
cd $R
timeout -k 10 500 python tools/random_shapes_ab.py 260 > $OUT/now.jsonl 2> $OUT/now.err || { tail -5 $OUT/now.err; exit 1; }
CTC_TC_NO_FLAT_BATCH=1 CTC_TC_NO_BATCHED_CHUNKS=1 CTC_TC_SPLIT_BY_SLOTS=1 CTC_TC_NO_LIGHT_ROUNDS=1 CTC_TC_OLD_TILE_RULE=1 timeout -k 10 500 python tools/random_shapes_ab.py 260 > $OUT/before.jsonl 2> $OUT/before.err || { tail -5 $OUT/before.err; exit 1; }
timeout -k 10 500 python tools/random_shapes_ab.py 260 > $OUT/now2.jsonl 2> $OUT/now.err || { tail -5 $OUT/now.err; exit 1; }
wc -l $OUT/*.jsonl
