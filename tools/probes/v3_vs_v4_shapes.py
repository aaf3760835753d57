import json, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from aria_amd import ops
bf16=torch.bfloat16; dev="cuda"
def timeit(fns, iters=12, warm=3):
    for i in range(warm): fns[i%len(fns)]()
    torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters): fns[i%len(fns)]()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/iters*1e-3
res={}
os.environ["ARIA_GEMM_FORCE"]="3"
VARS=[v for v in os.environ.get("PROBE_VARIANTS","0,1").split(",")]
for M,N,K in ((8192,8192,8192),(16384,2560,2560),(16384,7680,2560),(16384,6656,2560),(78400,4304,1152),(78400,1152,4304)):
    xs=[torch.randn(M,K,device=dev).to(bf16) for _ in range(2)]; ws=[(torch.randn(N,K,device=dev)*0.02).to(bf16) for _ in range(2)]
    for v in VARS:
        os.environ["ARIA_GEMM_V4"]=v
        t=timeit([lambda x=x,w=w: ops.gemm(x,w) for x,w in zip(xs,ws)])
        res[f"rcrc_{M}x{N}x{K}_v4={v}"]=round(2*M*N*K/t/1e12,1)
    del xs, ws
# rc,oc dense (dgrad form) and oc,oc (wgrad form)
M,N,K=16384,2560,7680
x=torch.randn(M,K,device=dev).to(bf16); w=(torch.randn(K,N,device=dev)*0.02).to(bf16)
for v in VARS:
    os.environ["ARIA_GEMM_V4"]=v
    res[f"rcoc_{M}x{N}x{K}_v4={v}"]=round(2*M*N*K/timeit([lambda: ops.gemm(x,w,b_oc=True)])/1e12,1)
dy=torch.randn(16384,7680,device=dev).to(bf16); xx=torch.randn(16384,2560,device=dev).to(bf16)
for v in VARS:
    os.environ["ARIA_GEMM_V4"]=v
    res[f"ococ_7680x2560x16384_v4={v}"]=round(2*16384*7680*2560/timeit([lambda: ops.gemm(dy,xx,a_oc=True,b_oc=True)])/1e12,1)
del x,w,dy,xx
E,T,topk=64,16384,6
g=torch.Generator().manual_seed(1)
counts=torch.bincount(torch.randint(0,E,(T*topk,),generator=g),minlength=E)
off=torch.zeros(E+1,dtype=torch.int32); off[1:]=torch.cumsum(counts,0); Mr=int(off[-1]); offd=off.to(dev)
for name,K,N in (("fc1",2560,3328),("fc2",1664,2560)):
    a=[torch.randn(Mr,K,device=dev).to(bf16) for _ in range(2)]
    w=[(torch.randn(E,K,N,device=dev)*0.02).to(bf16) for _ in range(3)]
    dy=[torch.randn(Mr,N,device=dev).to(bf16) for _ in range(2)]
    out=torch.empty(Mr,N,dtype=bf16,device=dev); din=torch.empty(Mr,K,dtype=bf16,device=dev); gw=torch.empty(E,K,N,dtype=bf16,device=dev)
    fl=2*Mr*K*N
    for v in VARS:
        os.environ["ARIA_GEMM_V4"]=v
        res[f"{name}_fwd_v4={v}"]=round(fl/timeit([lambda i=i: ops.grouped_gemm(a[i%2],w[i],offd,out=out) for i in range(3)])/1e12,1)
        res[f"{name}_dgrad_v4={v}"]=round(fl/timeit([lambda i=i: ops.grouped_gemm(dy[i%2],w[i],offd,w_is_kn=False,out=din) for i in range(3)])/1e12,1)
        res[f"{name}_wgrad_v4={v}"]=round(fl/timeit([lambda i=i: ops.grouped_gemm_wgrad(a[i%2],dy[i%2],offd,E,out=gw) for i in range(2)])/1e12,1)
    del a,w,dy,out,din,gw
print(json.dumps(res))
