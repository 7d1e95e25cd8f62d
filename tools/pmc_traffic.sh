# HBM traffic of the headline contraction: FETCH_SIZE and WRITE_SIZE in separate counter passes.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
N=${1:-96}
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/traffic_$C
  timeout -s KILL 240 rocprofv3 --pmc $C -d $R/gpurun_out/traffic_$C -o t -- python3 $R/tools/traffic_probe.py $N 3 > $R/gpurun_out/traffic_$C.log 2>&1 || { tail -5 $R/gpurun_out/traffic_$C.log; exit 1; }
done
rm -rf $R/gpurun_out/prof_bench
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- python3 $R/bench.py --no-cpu --extent $N > $R/gpurun_out/prof_bench.log 2>&1 || { tail -5 $R/gpurun_out/prof_bench.log; exit 1; }
tail -1 $R/gpurun_out/prof_bench.log | cut -c1-200
ls $R/gpurun_out/prof_bench/*
