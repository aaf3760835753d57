import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from aria_amd import ops
bf16 = torch.bfloat16; dev = "cuda"
T, D, I, E, k = 16384, 2560, 1664, 64, 6
x = torch.randn(T, D, device=dev).to(bf16)
logits = torch.randn(T, E, device=dev).to(bf16)
scores, idx, counts = ops.moe_route(logits, k)
off, sorted_src, inv = ops.moe_sort(idx, counts)
fc1 = (torch.randn(E, D, 2 * I, device=dev) * 0.02).to(bf16)
perm = ops.moe_permute(x, sorted_src, k)
for _ in range(3):  # the launch bench.py times: fc1 + SwiGLU epilogue, h kept for the backward
    h, act = ops.grouped_gemm_swiglu(perm, fc1, off, True)
torch.cuda.synchronize()
