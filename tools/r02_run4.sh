#!/bin/bash
set -o pipefail
for v in dx_base dx_nostore dx_nodma dx_nobarrier dx_nowaitv dx_nodma_nostore; do
  CTC_TC_LIB=$PWD/ctc_amd/lib/variants/libctc_tc_$v.so timeout -k 10 120 python -u tools/stream_variants.py 2>&1 | grep -v amdgpu.ids || exit 1
done
