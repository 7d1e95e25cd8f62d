import sys, json, time, statistics, torch
sys.path.insert(0, "/root/repo")
import ctc_amd
def timed(fn, iters=5, seconds=0.1):
    t0=time.perf_counter()
    while time.perf_counter()-t0<seconds:
        fn(); torch.cuda.synchronize()
    each=[]
    for _ in range(iters):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); each.append(e0.elapsed_time(e1))
    return statistics.median(each)
dev=torch.device("cuda:0")
for dt,tdt in (("f64",torch.float64),("f32",torch.float32)):
  for n,d in ((41,64),(39,64),(41,128),(42,64)):
    ext=dict(a=5,b=n,c=n,i=n,j=n,k=n,d=d)
    A=torch.zeros(5*d*n*n,dtype=tdt,device=dev); B=torch.zeros(n*n*d*n,dtype=tdt,device=dev)
    ctc_amd.fill(A,1); ctc_amd.fill(B,2)
    C1=torch.empty(5*n**5,dtype=tdt,device=dev); C2=torch.empty(5*n**5,dtype=tdt,device=dev)
    plan=ctc_amd.Plan("adij,bcdk->abcijk",ext,dt); s=plan.summary()
    names=("launches_classic","launches_dma_gemm","launches_stream","launches_stream_f64_dma")
    ctc_amd.set_option("pad_transposes",0); t0=timed(lambda: plan.execute(A,B,C1))
    ctc_amd.set_option("pad_transposes",1)
    b=[ctc_amd.get_option(x) for x in names]
    t1=timed(lambda: plan.execute(A,B,C2))
    took=[x[9:] for x,y in zip(names,b) if ctc_amd.get_option(x)>y]
    fl=plan.flops
    print(json.dumps({"dtype":dt,"n":n,"d":d,"auto":s["transposes"],"layouts":[s["layout_a"],s["layout_b"]],"fused_tflops":round(fl/t0*1e-9,1),"auto_tflops":round(fl/t1*1e-9,1),
        "took":took,"ratio":round(t0/t1,3),"diff":float((C1-C2).abs().max()/C1.abs().max())}),flush=True)
    del A,B,C1,C2; torch.cuda.empty_cache()
