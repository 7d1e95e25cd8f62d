# HBM traffic of the permutation and skinny-contraction kernels: FETCH_SIZE and WRITE_SIZE in separate passes.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/ptraffic_$C
  timeout -s KILL 240 rocprofv3 --pmc $C -d $R/gpurun_out/ptraffic_$C -o t -- python3 $R/tools/permute_traffic_probe.py > $R/gpurun_out/ptraffic_$C.log 2>&1 || { tail -5 $R/gpurun_out/ptraffic_$C.log; exit 1; }
done
python3 - <<'PY'
import glob, json, os, sqlite3
R = os.environ["GRAFT_REPO_ROOT"]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for path in glob.glob(R + "/gpurun_out/ptraffic_%s/**/*.db" % c, recursive=True):
        db = sqlite3.connect(path)
        for name, cn, value in db.execute("select kernel_name, counter_name, value from counters_collection"):
            if cn != c:
                continue
            key = "copy" if "copyBuffer" in name or "elementwise" in name else "permute_vec" if "permute_vec" in name else "skinny" if "skinny" in name else None
            if key:
                out.setdefault(key, {}).setdefault(c + "_KB", []).append(float(value))
json.dump(out, open(R + "/gpurun_out/permute_traffic_raw.json", "w"), indent=1)
print(json.dumps(out)[:600])
PY
