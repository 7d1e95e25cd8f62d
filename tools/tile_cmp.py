"""The 64 x 64 and the 128 x 128 tile (CTC_TC_FORCE_TILE) on mid-sized fp64 products, with the split each gets:
which tile the planner should take where the large one does not fill the chip by itself.   python tools/tile_cmp.py"""
import json, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import ctc_amd
dev = torch.device("cuda:0")
shapes = [(1024,4096,1024),(1024,4096,4096),(1024,4096,16384),(256,16384,4096),(1024,1024,4096),(1024,1024,16384),(256,4096,16384),(2048,2048,2048),(2048,2048,8192),(1536,1536,4096),(2304,2304,2304),(1600,1600,1600),(512,8192,2048),(1024,2048,4096)]
for (m,n,k) in shapes:
    A = torch.empty(m*k, dtype=torch.float64, device=dev); B = torch.empty(k*n, dtype=torch.float64, device=dev); C = torch.empty(m*n, dtype=torch.float64, device=dev)
    ctc_amd.fill(A,1); ctc_amd.fill(B,2)
    row = {"shape": [m,n,k]}
    for force in ("64","128"):
        os.environ["CTC_TC_FORCE_TILE"] = force
        plan = ctc_amd.Plan("ak,kb->ab", dict(a=m,b=n,k=k), "f64")
        for _ in range(10): plan.execute(A,B,C)
        ms = plan.time(A,B,C,warmup=2,iters=10)
        row["t"+force] = round(2.0*m*n*k/ms*1e-9,1)
        row["split"+force] = plan.schedule()["split"]
    del os.environ["CTC_TC_FORCE_TILE"]
    row["default_tile"] = ctc_amd.Plan("ak,kb->ab", dict(a=m,b=n,k=k), "f64").summary()["tile"][0]
    print(json.dumps(row), flush=True)
