#!/bin/bash
set -o pipefail
for v in dx_stamps; do
  echo "== $v"
  CTC_TC_LIB=$PWD/ctc_amd/lib/variants/libctc_tc_$v.so timeout -k 10 120 python -u tools/stream_variants.py 2>&1 | grep -v amdgpu.ids | sort | uniq -c | sort -rn | head -8 || exit 1
done
