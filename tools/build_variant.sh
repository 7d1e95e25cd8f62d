#!/bin/bash
# Diagnostic builds of libctc_tc.so: recompiles one kernel source (gemm_kernels by default) with extra
# flags and links it with the objects of the regular build.   tools/build_variant.sh <name> "<flags>" [source stem]
# Result: ctc_amd/lib/variants/libctc_tc_<name>.so  (use with CTC_TC_LIB=...)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/ctc_amd/csrc"
mkdir -p ../lib/variants /tmp/ctc_variants
SRC=${3:-gemm_kernels}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$ROOT/include" -I. $2 -c $SRC.hip -o /tmp/ctc_variants/${SRC}_$1.o
OBJS=$(ls ../build/*.o | grep -v -E "/${SRC}(_p[0-9]+)?\.o$")  # also the per-instantiation parts of a source built in parts
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/libctc_tc_$1.so $OBJS /tmp/ctc_variants/${SRC}_$1.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo built ../lib/variants/libctc_tc_$1.so
