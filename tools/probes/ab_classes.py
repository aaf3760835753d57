"""TF/s of the GEMM classes of the config #3 step with the library that is in the tree, default knobs -- run once per library build on
the SAME box (tools/gpu_ab_lib.sh swaps the file) to compare two builds; operands rotate so nothing stays in the infinity cache."""
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


for M, N, K in ((16384, 2560, 2560), (16384, 7680, 2560), (78400, 1152, 1152), (78400, 1152, 4304)):
    xs = [torch.randn(M, K, device=dev).to(bf16) for _ in range(3)]
    ws = [(torch.randn(N, K, device=dev) * 0.02).to(bf16) for _ in range(3)]
    res[f"dense {M}x{N}x{K}"] = round(2 * M * N * K / timeit([lambda x=x, w=w: ops.gemm(x, w) for x, w in zip(xs, ws)]) / 1e12, 1)
    del xs, ws
M, N, K = 78400, 4304, 1152
xs = [torch.randn(M, K, device=dev).to(bf16) for _ in range(2)]
ws = [(torch.randn(N, K, device=dev) * 0.02).to(bf16) for _ in range(2)]
b = torch.randn(N, device=dev).to(bf16)
res["vit fc1 + bias + gelu"] = round(2 * M * N * K / timeit([lambda x=x, w=w: ops.gemm(x, w, bias=b, act="gelu_tanh") for x, w in zip(xs, ws)]) / 1e12, 1)
del xs, ws
E, T, topk = 64, 16384, 6
g = torch.Generator().manual_seed(1)
counts = torch.bincount(torch.randint(0, E, (T * topk,), generator=g), minlength=E)
off = torch.zeros(E + 1, dtype=torch.int32)
off[1:] = torch.cumsum(counts, 0)
M = int(off[-1])
offd = off.to(dev)
D, I = 2560, 1664
a = [torch.randn(M, D, device=dev).to(bf16) for _ in range(2)]
w1 = [(torch.randn(E, D, 2 * I, device=dev) * 0.02).to(bf16) for _ in range(3)]
w2 = [(torch.randn(E, I, D, device=dev) * 0.02).to(bf16) for _ in range(3)]
h = [torch.randn(M, I, device=dev).to(bf16) for _ in range(2)]
dy1 = [torch.randn(M, 2 * I, device=dev).to(bf16) for _ in range(2)]
f1, f2 = 2 * M * D * 2 * I, 2 * M * I * D
res["fc1 + swiglu fused (h kept)"] = round(f1 / timeit([lambda i=i: ops.grouped_gemm_swiglu(a[i % 2], w1[i], offd, True) for i in range(3)]) / 1e12, 1)
res["fc1 plain"] = round(f1 / timeit([lambda i=i: ops.grouped_gemm(a[i % 2], w1[i], offd) for i in range(3)]) / 1e12, 1)
res["fc2 fwd"] = round(f2 / timeit([lambda i=i: ops.grouped_gemm(h[i % 2], w2[i], offd) for i in range(3)]) / 1e12, 1)
din = torch.empty(M, D, dtype=bf16, device=dev)
res["fc1 dgrad"] = round(f1 / timeit([lambda i=i: ops.grouped_gemm(dy1[i % 2], w1[i], offd, w_is_kn=False, out=din) for i in range(3)]) / 1e12, 1)
gw = torch.empty(E, D, 2 * I, dtype=bf16, device=dev)
res["fc1 wgrad"] = round(f1 / timeit([lambda i=i: ops.grouped_gemm_wgrad(a[i % 2], dy1[i % 2], offd, E, out=gw) for i in range(2)]) / 1e12, 1)
print(json.dumps(res))
