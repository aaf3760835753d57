"""target for an LDS bank-conflict PMC pass: v3 GEMM in all operand forms, attention fwd (hd 72/128) and bwd3"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from aria_amd import ops
bf16 = torch.bfloat16; dev = "cuda"
os.environ["ARIA_GEMM_FORCE"] = "3"
ops.GEMM_SPLIT_K = False
M = N = K = 4096
A = torch.randn(M, K, device=dev).to(bf16); B = torch.randn(K, N, device=dev).to(bf16)
for a_oc, b_oc in ((0, 0), (0, 1), (1, 1)):
    aa = A.t().contiguous() if a_oc else A
    bb = B if b_oc else B.t().contiguous()
    for _ in range(2):
        ops.gemm(aa, bb, a_oc=bool(a_oc), b_oc=bool(b_oc))
os.environ["ARIA_GEMM_FORCE"] = "2"
for _ in range(2):
    ops.gemm(A, B, b_oc=True)
for (B_, S, H, hd, causal) in ((4, 2048, 20, 128, True), (4, 4900, 16, 72, False)):
    D = H * hd
    qkv = torch.randn(B_ * S, 3 * D, device=dev).to(bf16)
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    o, lse = ops.attention_fwd(q, k, v, B_, S, H, hd, hd ** -0.5, causal)
    if hd == 128:
        do = torch.randn_like(o)
        ops.attention_bwd(q, k, v, o, do, lse, B_, S, H, hd, hd ** -0.5, causal)
torch.cuda.synchronize()
