#!/bin/bash
# Builds the host side of libctc_tc (api.cpp, sharded.cpp, fastr_api.cpp, planner.cpp, permute_plan.cpp) with
# AddressSanitizer + UndefinedBehaviorSanitizer on the HOST code only, links it with the kernel objects of the
# ordinary build (make in ctc_amd/csrc first) and the driver tools/api_sanitize.cpp, and runs it.  No GPU needed.
#   tools/api_sanitize.sh [cases] [seed]
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/ctc_api_sanitize
mkdir -p "$OUT"
HIPCC=/opt/rocm/bin/hipcc
SAN="-Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-sanitize-recover=all -Xarch_host -fno-omit-frame-pointer"
FLAGS="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -I$ROOT/include -I$ROOT/ctc_amd/csrc"
cd "$ROOT/ctc_amd/csrc"
for f in api sharded fastr_api planner permute_plan; do
  if [ ! -f "$OUT/$f.o" ] || [ "$f.cpp" -nt "$OUT/$f.o" ] || [ api_internal.hpp -nt "$OUT/$f.o" ] || [ kernels.hpp -nt "$OUT/$f.o" ]; then
    $HIPCC $FLAGS $SAN -x hip -c $f.cpp -o "$OUT/$f.o" &
  fi
done
wait
$HIPCC $FLAGS $SAN -x hip -c "$ROOT/tools/api_sanitize.cpp" -o "$OUT/driver.o"
KERNELS=$(ls "$ROOT"/ctc_amd/build/*.o | grep -v -E '/(api|sharded|fastr_api|planner|permute_plan)\.o$')
$HIPCC --offload-arch=gfx950 -fsanitize=address,undefined -fno-gpu-sanitize -o "$OUT/api_sanitize" "$OUT"/driver.o "$OUT"/api.o "$OUT"/sharded.o \
  "$OUT"/fastr_api.o "$OUT"/planner.o "$OUT"/permute_plan.o $KERNELS -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
# leaks are checked for the library's own allocations; the HIP runtime keeps process-lifetime state of its own
cat > "$OUT/lsan.supp" <<SUP
leak:libamdhip64
leak:libhsa-runtime64
leak:librccl
leak:libamd_comgr
SUP
ASAN_OPTIONS=detect_leaks=1:abort_on_error=0 LSAN_OPTIONS=suppressions=$OUT/lsan.supp:print_suppressions=0 UBSAN_OPTIONS=print_stacktrace=1 \
  "$OUT/api_sanitize" "${1:-1000}" "${2:-12345}"
