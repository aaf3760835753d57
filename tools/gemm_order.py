import os, sys, json, torch
sys.path.insert(0, ".")
from aria_amd import ops
from tools.microbench import timeit
bf16 = torch.bfloat16; dev = "cuda"
T, D, I, E, k = 16384, 2560, 1664, 64, 6
x = torch.randn(T, D, device=dev).to(bf16)
logits = torch.randn(T, E, device=dev).to(bf16)
scores, idx, counts = ops.moe_route(logits, k)
off, sorted_src, inv = ops.moe_sort(idx, counts)
M = T * k
fc1 = (torch.randn(E, D, 2 * I, device=dev) * 0.02).to(bf16)
perm = ops.moe_permute(x, sorted_src, k)
w = (torch.randn(D, D, device=dev) * 0.02).to(bf16)
h = ops.grouped_gemm(perm, fc1, off)
res = {}
for order in ("0", "1", "2", "4", "8", "16"):
    os.environ["ARIA_GEMM_ORDER"] = order
    t = timeit(lambda: ops.grouped_gemm(perm, fc1, off), 5, 2)
    t2 = timeit(lambda: ops.gemm(x, w), 10, 3)
    t3 = timeit(lambda: ops.grouped_gemm_wgrad(perm, h, off, E), 5, 2)
    res[order] = dict(fc1_tf=round(2 * M * D * 2 * I / t / 1e12), dense_tf=round(2 * T * D * D / t2 / 1e12), wgrad_tf=round(2 * M * D * 2 * I / t3 / 1e12))
print(json.dumps(res))
