#!/bin/bash
# Builds the REFERENCE's own host programs from the sources where they lie under
# /root/reference (never copied into this repo); outputs go to oracle/_ref/ only.
#
#   execute_engine, query_engine : the unchanged CTC engine, used as the boundary host
#       in tests (it expands a plugin's command template and system()s it).
#       engine/src/ctc_engine.cpp:230 redeclares `it` inside the scope of the for at
#       :228, which g++ >= 5 rejects; the inner iterator is renamed IN FLIGHT (sed into
#       the compiler's stdin, lines 230-276 only).  Nothing else is touched.
#
# The reference's own build system (autotools) is not run.
set -euo pipefail
REF=${CTC_REFERENCE:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
[ -d "$REF/engine/src" ] || { echo "no reference tree at $REF: keeping prebuilt files in $OUT"; exit 0; }
mkdir -p "$OUT/obj"
CXX="g++ -std=gnu++98 -O1 -w -fpermissive -I$REF/engine/include"
for f in ctc_catalog ctc_logger ctc_plugin ctc_sysconfig ctc_util ctc_workflow; do
    $CXX -c "$REF/engine/src/$f.cpp" -o "$OUT/obj/$f.o"
done
sed '230,276s/\bit\b/it_inner/g' "$REF/engine/src/ctc_engine.cpp" | $CXX -x c++ -c - -o "$OUT/obj/ctc_engine.o"
for prog in execute_engine query_engine; do
    $CXX "$REF/engine/src/$prog.cpp" "$OUT"/obj/ctc_*.o -o "$OUT/$prog"
done
rm -rf "$OUT/obj"
echo "built $OUT/execute_engine $OUT/query_engine"
