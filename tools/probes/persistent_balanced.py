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
os.environ["ARIA_GEMM_FORCE"]="3"; os.environ["ARIA_GEMM_V4"]="0"
# exact multiples of 256 tiles so that the static round-robin is balanced
for M,N,K in ((16384,8192,2560),(16384,8192,1280),(16384,8192,640),(32768,8192,2560)):
    x=torch.randn(M,K,device=dev).to(bf16); w=(torch.randn(N,K,device=dev)*0.02).to(bf16); out=torch.empty(M,N,dtype=bf16,device=dev)
    for mode in ("0","1"):
        os.environ["ARIA_GEMM_PERSIST"]=mode
        t=timeit(lambda: ops.gemm(x,w,out=out))
        res[f"{M}x{N}x{K}_p{mode}"]=[round(2*M*N*K/t/1e12,1), round(t*1e6/(M//256*N//256/256),2)]
print(json.dumps(res))
