#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03tail
mkdir -p $OUT
cd $R
export TAIL_PROBE_ODDK=1 CTC_TC_FORCE_TILE=128 CTC_TC_NO_PAD_TRANSPOSES=1
timeout -k 10 300 python tools/tail_cost_probe.py > $OUT/oddk_modes2.jsonl 2> $OUT/sweep.err || { tail -5 $OUT/sweep.err; exit 1; }
