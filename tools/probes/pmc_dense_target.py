"""PMC target: three launches each of a dense rc,rc GEMM (8192^3 and 16384 x 7680 x 2560) and the grouped fc1 -- for tools/gpu_pmc_l2.sh."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from aria_amd import ops
bf16, dev = torch.bfloat16, "cuda"
for M, N, K in ((8192, 8192, 8192), (16384, 7680, 2560)):
    x = torch.randn(M, K, device=dev).to(bf16); w = (torch.randn(N, K, device=dev) * 0.02).to(bf16)
    for _ in range(3):
        ops.gemm(x, w)
    torch.cuda.synchronize()
T, D, I, E, k = 16384, 2560, 1664, 64, 6
g = torch.Generator().manual_seed(1)
counts = torch.bincount(torch.randint(0, E, (T * k,), generator=g), minlength=E)
off = torch.zeros(E + 1, dtype=torch.int32); off[1:] = torch.cumsum(counts, 0)
a = torch.randn(T * k, D, device=dev).to(bf16)
w1 = (torch.randn(E, D, 2 * I, device=dev) * 0.02).to(bf16)
for _ in range(3):
    ops.grouped_gemm(a, w1, off.to(dev))
torch.cuda.synchronize()
