import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from aria_amd import ops
bf16 = torch.bfloat16; dev = "cuda"
os.environ["ARIA_GEMM_FORCE"] = "2"
M, N, K = 16384, 16384, 2560
a = torch.randn(M, K, device=dev).to(bf16); b = torch.randn(N, K, device=dev).to(bf16)
for order in (0, 1, 2, 4, 8):
    os.environ["ARIA_GEMM_ORDER"] = str(order)
    for _ in range(2):
        ops.gemm(a, b)
    torch.cuda.synchronize()
