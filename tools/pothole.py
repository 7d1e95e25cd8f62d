import json, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import ctc_amd
dev = torch.device("cuda:0")
def run(name, spec, ext, dt="f64"):
    tdt = torch.float64
    ins, outm = spec.split("->"); la, lb = ins.split(",")
    n = lambda s: int(torch.tensor([ext[c] for c in s]).prod())
    A = torch.empty(n(la), dtype=tdt, device=dev); B = torch.empty(n(lb), dtype=tdt, device=dev); C = torch.empty(n(outm), dtype=tdt, device=dev)
    ctc_amd.fill(A, 1); ctc_amd.fill(B, 2)
    plan = ctc_amd.Plan(spec, ext, dt)
    for _ in range(5): plan.execute(A, B, C)
    ms = plan.time(A, B, C, warmup=3, iters=10)
    s = plan.summary()
    print(json.dumps({"case": name, "ms": round(ms, 4), "TFLOP/s": round(plan.flops / ms * 1e-9, 2), "layouts": [s["layout_a"], s["layout_b"]], "tile": s["tile"]}), flush=True)
for n in (3072, 3136, 3200):
    run("plain nn %d" % n, "ak,kb->ab", dict(a=n, b=n, k=n))
for n in (48, 56, 64):
    run("ccsd %d" % n, "aeim,ebmj->abij", dict(a=n, b=n, i=n, j=n, e=n, m=n))
