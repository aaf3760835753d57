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
M,N=16384,8192   # 2048 tiles = 8 full rounds
for K in (640,1280,2560,5120,10240):
    x=torch.randn(M,K,device=dev).to(bf16); w=(torch.randn(N,K,device=dev)*0.02).to(bf16)
    for v in ("0","1"):
        os.environ["ARIA_GEMM_V4"]=v
        t=timeit(lambda: ops.gemm(x,w))
        res[f"K{K}_v4={v}"]=[round(2*M*N*K/t/1e12,1), round(t*1e6/8,2)]   # us per round of tiles
print(json.dumps(res))
