"""Dense GEMM, same shape, B operand as [N, K] (rc) vs [K, N] (oc) -- and A as [K, M] -- TF/s with the library in the tree."""
import json, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from aria_amd import ops  # noqa: E402
bf16, dev, res = torch.bfloat16, "cuda", {}


def timeit(fns, iters=12, warm=3):
    for i in range(warm):
        fns[i % len(fns)]()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters):
        fns[i % len(fns)]()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


for M, N, K in ((16384, 7680, 2560), (8192, 8192, 8192), (16384, 8192, 2304), (16384, 8192, 2560), (16384, 8192, 2688)):
    f = 2 * M * N * K
    x = [torch.randn(M, K, device=dev).to(bf16) for _ in range(2)]
    xt = [torch.randn(K, M, device=dev).to(bf16) for _ in range(2)]
    w = [(torch.randn(N, K, device=dev) * 0.02).to(bf16) for _ in range(2)]
    wt = [(torch.randn(K, N, device=dev) * 0.02).to(bf16) for _ in range(2)]
    res[f"{M}x{N}x{K} rc,rc"] = round(f / timeit([lambda i=i: ops.gemm(x[i], w[i]) for i in range(2)]) / 1e12, 1)
    res[f"{M}x{N}x{K} rc,oc"] = round(f / timeit([lambda i=i: ops.gemm(x[i], wt[i], b_oc=True) for i in range(2)]) / 1e12, 1)
    res[f"{M}x{N}x{K} oc,oc"] = round(f / timeit([lambda i=i: ops.gemm(xt[i], wt[i], a_oc=True, b_oc=True) for i in range(2)]) / 1e12, 1)
    del x, xt, w, wt
print(json.dumps(res))
