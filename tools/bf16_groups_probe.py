import sys, os, json, time, statistics, torch
sys.path.insert(0, "/root/repo")
import ctc_amd
def timed(fn, iters=8, seconds=0.1):
    t0=time.perf_counter()
    while time.perf_counter()-t0<seconds:
        fn(); torch.cuda.synchronize()
    each=[]
    for _ in range(iters):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); each.append(e0.elapsed_time(e1))
    return statistics.median(each)
dev=torch.device("cuda:0")
for n in (56, 72, 88, 104):
    ext=dict(a=n,b=n,i=n,j=n,e=n,m=n)
    A=torch.zeros(n**4,dtype=torch.float32,device=dev); B=torch.zeros(n**4,dtype=torch.float32,device=dev)
    ctc_amd.fill(A,1); ctc_amd.fill(B,2); A=A.to(torch.bfloat16); B=B.to(torch.bfloat16)
    C1=torch.empty(n**4,dtype=torch.bfloat16,device=dev); C2=torch.empty(n**4,dtype=torch.bfloat16,device=dev)
    plan=ctc_amd.Plan("aeim,ebmj->abij",ext,"bf16"); plan.prepare()
    s=plan.summary()
    ctc_amd.set_option("pad_transposes",0); t0=timed(lambda: plan.execute(A,B,C1))
    ctc_amd.set_option("pad_transposes",1); t1=timed(lambda: plan.execute(A,B,C2))
    fl=2.0*n**6
    print(json.dumps({"extent":n,"automatic":s["transposes"],"fused_tflops":round(fl/t0*1e-9,1),"auto_tflops":round(fl/t1*1e-9,1),"ratio":round(t0/t1,3),
      "diff":float((C1.float()-C2.float()).abs().max()/C1.float().abs().max())}),flush=True)
