"""PMC target: the grouped fc1 GEMM (plain epilogue) with the routed counts of a real batch, and with every expert at exactly 1536 rows
(no ragged row tiles) -- does the L2 hit rate of the grouped launch recover when all tiles take the same time?"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from aria_amd import ops
bf16, dev = torch.bfloat16, "cuda"
T, D, I, E, k = 16384, 2560, 1664, 64, 6
g = torch.Generator().manual_seed(1)
counts = torch.bincount(torch.randint(0, E, (T * k,), generator=g), minlength=E)
off = torch.zeros(E + 1, dtype=torch.int32); off[1:] = torch.cumsum(counts, 0)
off_al = (torch.arange(E + 1, dtype=torch.int32) * 1536)
a = torch.randn(T * k, D, device=dev).to(bf16)
w1 = (torch.randn(E, D, 2 * I, device=dev) * 0.02).to(bf16)
out = torch.empty(T * k, 2 * I, dtype=bf16, device=dev)
for order, o in (("4", off), ("4", off_al), ("516", off)):   # real counts; aligned; real counts with the ragged-last order
    os.environ["ARIA_GEMM_ORDER"] = order
    od = o.to(dev)
    for _ in range(3):
        ops.grouped_gemm(a, w1, od, out=out)
    torch.cuda.synchronize()
