import json, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import ctc_amd
dev = torch.device("cuda:0")
for spec in ("ak,kb->ab", "dkca,bk->abcd"):
  for m in (4, 8, 16):
    for k in (4, 8, 16, 32, 64, 128):
        if spec == "ak,kb->ab": ext = dict(a=m, b=1<<20, k=k)
        else: ext = dict(a=128, b=m, c=256, d=128, k=k)
        ins, outm = spec.split("->"); la, lb = ins.split(",")
        n = lambda s: int(torch.tensor([ext[c] for c in s]).prod())
        A = torch.empty(n(la), dtype=torch.float32, device=dev); B = torch.empty(n(lb), dtype=torch.float32, device=dev); C = torch.empty(n(outm), dtype=torch.float32, device=dev)
        ctc_amd.fill(A,1); ctc_amd.fill(B,2)
        row = {"spec": spec, "m": m, "k": k}
        for name, env in (("skinny", None), ("gemm", "1")):
            if env: os.environ["CTC_TC_NO_SKINNY"] = env
            else: os.environ.pop("CTC_TC_NO_SKINNY", None)
            plan = ctc_amd.Plan(spec, ext, "f32")
            for _ in range(3): plan.execute(A,B,C)
            ms = plan.time(A,B,C,warmup=2,iters=5)
            row[name] = plan.summary()["kernel"]; row[name+"_GB/s"] = round((A.numel()+B.numel()+C.numel())*4/ms*1e-6,1)
        print(json.dumps(row), flush=True)
        del A,B,C
