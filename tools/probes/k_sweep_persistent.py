import json, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from aria_amd import ops
bf16=torch.bfloat16; dev="cuda"
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/iters*1e-3
res={}
os.environ["ARIA_GEMM_FORCE"]="3"
M,N=16384,2560
for K in (1024,2048,2560,3072,3584,4096,4608,5120,6144):
    x=torch.randn(M,K,device=dev).to(bf16); w=(torch.randn(N,K,device=dev)*0.02).to(bf16)
    for mode in ("0","1"):
        os.environ["ARIA_GEMM_PERSIST"]=mode
        t=timeit(lambda: ops.gemm(x,w))
        res[f"K{K}_p{mode}"]=[round(2*M*N*K/t/1e12,1), round(t*1e6,1)]
K=2560
x=torch.randn(M,K,device=dev).to(bf16); w=(torch.randn(N,K,device=dev)*0.02).to(bf16)
os.environ["ARIA_GEMM_PERSIST"]="1"
for grid in (64,128,192,256,320,512):
    os.environ["ARIA_GEMM_PERSIST_GRID"]=str(grid)
    t=timeit(lambda: ops.gemm(x,w))
    res[f"K2560_grid{grid}"]=[round(2*M*N*K/t/1e12,1), round(t*1e6,1)]
print(json.dumps(res,indent=0))
