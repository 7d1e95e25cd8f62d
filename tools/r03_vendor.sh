#!/bin/bash
# this library beside torch.matmul / torch.einsum on the same box
set -o pipefail
OUT=$PWD/gpurun_out/r03v
mkdir -p $OUT
timeout -k 10 900 python -u tools/vendor_gemm_cmp.py > $OUT/vendor_gemm_cmp.jsonl 2> $OUT/vendor_gemm_cmp.err || { tail -20 $OUT/vendor_gemm_cmp.err; cut -c1-300 $OUT/vendor_gemm_cmp.jsonl; exit 1; }
cut -c1-300 $OUT/vendor_gemm_cmp.jsonl
