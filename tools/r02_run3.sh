#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_stream_gpu.py -x -q --timeout 180 --timeout-method thread > gpurun_out/r3_stream_pytest.log 2>&1
rc=$?; tail -25 gpurun_out/r3_stream_pytest.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -u tools/stream_cmp.py all > gpurun_out/r3_stream_cmp.jsonl 2> gpurun_out/r3_stream_cmp.err
rc=$?; cat gpurun_out/r3_stream_cmp.jsonl; tail -5 gpurun_out/r3_stream_cmp.err; exit $rc
