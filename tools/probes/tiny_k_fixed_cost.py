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
os.environ["ARIA_GEMM_FORCE"]="3"; os.environ["ARIA_GEMM_V4"]="0"
N=8192
for M in (8192, 16384, 65536):
    rounds=M//256*N//256/256
    for K in (64,128,256,640,2560):
        x=torch.randn(M,K,device=dev).to(bf16); w=(torch.randn(N,K,device=dev)*0.02).to(bf16); out=torch.empty(M,N,dtype=bf16,device=dev)
        t=timeit(lambda: ops.gemm(x,w,out=out))
        res[f"M{M}_K{K}"]=round(t*1e6/rounds,2)
print(json.dumps(res))
