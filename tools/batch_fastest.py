import json, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import ctc_amd
dev = torch.device("cuda:0")
for spec, ext in [("akz,kbz->abz", dict(z=64,a=256,b=256,k=256)), ("akz,kbz->abz", dict(z=8,a=1024,b=1024,k=1024)), ("azk,kzb->abz", dict(z=32,a=512,b=512,k=512)),
                  ("kza,kzb->zab", dict(z=32,a=512,b=512,k=512))]:
    ins, outm = spec.split("->"); la, lb = ins.split(",")
    n = lambda s: int(torch.tensor([ext[c] for c in s]).prod())
    A = torch.empty(n(la), dtype=torch.float64, device=dev); B = torch.empty(n(lb), dtype=torch.float64, device=dev); C = torch.empty(n(outm), dtype=torch.float64, device=dev)
    ctc_amd.fill(A,1); ctc_amd.fill(B,2)
    plan = ctc_amd.Plan(spec, ext, "f64")
    s = plan.summary()
    for _ in range(3): plan.execute(A,B,C)
    ms = plan.time(A,B,C,warmup=2,iters=5)
    ws = torch.empty(plan.ttgt_workspace_bytes(), dtype=torch.uint8, device=dev)
    for _ in range(3): plan.execute_ttgt(A,B,C,ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): plan.execute_ttgt(A,B,C,ws)
    e1.record(); e1.synchronize()
    ms2 = e0.elapsed_time(e1)/5
    print(json.dumps({"spec": spec, "ext": ext, "layouts": [s["layout_a"], s["layout_b"]], "vec_c": s["vec_c"], "fused_TFLOP/s": round(plan.flops/ms*1e-9,1), "ttgt_TFLOP/s": round(plan.flops/ms2*1e-9,1)}), flush=True)
