import sys
import torch
sys.path.insert(0, ".")
from aria_amd import ops
bf16 = torch.bfloat16
dev = "cuda"

def run(counts, K, N, what):
    E = len(counts); M = sum(counts)
    a = torch.randn(M, K).to(bf16).to(dev); w = (torch.randn(E, K, N) * 0.3).to(bf16).to(dev)
    off = torch.zeros(E + 1, dtype=torch.int32); off[1:] = torch.cumsum(torch.tensor(counts), 0); off = off.to(dev)
    dy = torch.randn(M, N).to(bf16).to(dev)
    print("case", counts, K, N, what, flush=True)
    if what == "fwd":
        o = ops.grouped_gemm(a, w, off)
        torch.cuda.synchronize()
        ref = torch.cat([a[off[e]:off[e + 1]].float() @ w[e].float() for e in range(E)])
        print("  ok, max err", (o.float() - ref).abs().max().item(), flush=True)
    elif what == "dgrad":
        o = ops.grouped_gemm(dy, w, off, w_is_kn=False); torch.cuda.synchronize(); print("  ok", flush=True)
    else:
        o = ops.grouped_gemm_wgrad(a, dy, off, E, out_dtype=torch.float32); torch.cuda.synchronize(); print("  ok", flush=True)

for what in ("fwd", "dgrad", "wgrad"):
    run([128], 64, 128, what)
    run([256], 64, 128, what)
    run([128, 128], 64, 128, what)
    run([3, 5], 64, 128, what)
    run([3, 0, 130, 5, 0, 0, 70, 1], 72, 136, what)
