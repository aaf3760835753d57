import json, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from aria_amd import ops
bf16=torch.bfloat16; dev="cuda"
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/iters*1e-3
res={}
os.environ["ARIA_GEMM_FORCE"]="3"; os.environ["ARIA_GEMM_WIDE_STORE"]="2"
for M,N,K in ((16384,8192,2560),(65536,8192,2560),(16384,7680,2560),(78400,4304,1152),(100352,2560,16384)):
    x=torch.randn(M,K,device=dev).to(bf16); w=(torch.randn(N,K,device=dev)*0.02).to(bf16); out=torch.empty(M,N,dtype=bf16,device=dev)
    for stg in ("0","4","8","16","32"):
        os.environ["ARIA_GEMM_STAGGER"]=stg
        t=timeit(lambda: ops.gemm(x,w,out=out))
        res[f"{M}x{N}x{K}_stagger{stg}"]=[round(2*M*N*K/t/1e12,1), round(t*1e6,1)]
print(json.dumps(res))
