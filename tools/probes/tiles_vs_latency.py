import json, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from aria_amd import ops
bf16=torch.bfloat16; dev="cuda"
def timeit(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/iters*1e-3
res={}
os.environ["ARIA_GEMM_FORCE"]="3"; os.environ["ARIA_GEMM_WIDE_STORE"]="2"
K=64
for tiles_m,tiles_n in ((1,1),(2,4),(4,8),(8,8),(8,16),(16,16),(32,16),(64,32)):
    M,N=256*tiles_m,256*tiles_n
    x=torch.randn(M,K,device=dev).to(bf16); w=(torch.randn(N,K,device=dev)*0.02).to(bf16); out=torch.empty(M,N,dtype=bf16,device=dev)
    t=timeit(lambda: ops.gemm(x,w,out=out))
    res[f"{tiles_m*tiles_n}_tiles_us"]=round(t*1e6,2)
print(json.dumps(res))
