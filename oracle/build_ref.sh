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
#   import_script, create_workflow : the reference's script importer (engine/src/import_script.cpp:50-97:
#       inline `name := { commands }` definitions become plugins chained into one workflow) and its
#       interactive workflow editor, compiled unmodified.
#   align_workflow : engine/examples/c++/align_workflow.cpp, compiled unmodified (it builds a four-step
#       workflow through the C++ API and dry-runs it, :64-73).
for prog in execute_engine query_engine import_script create_workflow; do
    $CXX "$REF/engine/src/$prog.cpp" "$OUT"/obj/ctc_*.o -o "$OUT/$prog"
done
$CXX "$REF/engine/examples/c++/align_workflow.cpp" "$OUT"/obj/ctc_*.o -o "$OUT/align_workflow"
rm -rf "$OUT/obj"
echo "built $OUT/execute_engine $OUT/query_engine $OUT/import_script $OUT/create_workflow $OUT/align_workflow"

#   libfastr_ref.so : the reference's CPU scoring path of apps/fastr (findBestOffset and
#       mismatchQualitySumIgnoreCigar), compiled from the reference sources with the vendored
#       bamtools API, wrapped by oracle/fastr_ref_harness.cpp.  Needs zlib, pthread and an
#       OpenCL loader to link (the OpenCL code is never executed).
F=$REF/apps/fastr
if [ -d "$F" ]; then
    mkdir -p "$OUT/fobj"
    FXX="g++ -std=gnu++98 -O1 -w -fpermissive -fPIC -I$F -I$F/bamtools/src"
    objs=""
    for f in AlignedRead BamRegionData Consensus ConsensusHashTable Fasta GenomeLoc HelperFunctions ReadBin \
             SWPairwiseAlignment SortedBamWriter disorder oclHelper split; do
        $FXX -c "$F/$f.cpp" -o "$OUT/fobj/$f.o" &
        objs="$objs $OUT/fobj/$f.o"
    done
    for f in "$F"/bamtools/src/api/*.cpp "$F"/bamtools/src/api/internal/*.cpp; do
        o="$OUT/fobj/bt_$(basename "$f" .cpp).o"
        $FXX -c "$f" -o "$o" &
        objs="$objs $o"
    done
    $FXX -c "$HERE/fastr_ref_harness.cpp" -o "$OUT/fobj/harness.o" &
    wait
    g++ -shared -o "$OUT/libfastr_ref.so" "$OUT/fobj/harness.o" $objs -lz -lpthread -lOpenCL
    rm -rf "$OUT/fobj"
    echo "built $OUT/libfastr_ref.so"
fi
