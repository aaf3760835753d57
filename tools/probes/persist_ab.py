"""GEMM classes with the one-tile-per-workgroup kernels and the persistent form (ARIA_GEMM_PERSIST=1), one process, interleaved."""
import json, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from aria_amd import ops  # noqa: E402
from tools.microbench import timeit  # noqa: E402
A, B = (sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else ("0", "1")
bf16, dev, res = torch.bfloat16, "cuda", {}
E, T, topk, D, I = 64, 16384, 6, 2560, 1664
g = torch.Generator().manual_seed(1)
counts = torch.bincount(torch.randint(0, E, (T * topk,), generator=g), minlength=E)
off = torch.zeros(E + 1, dtype=torch.int32); off[1:] = torch.cumsum(counts, 0)
M = int(off[-1]); offd = off.to(dev)
a = [torch.randn(M, D, device=dev).to(bf16) for _ in range(2)]
w1 = [(torch.randn(E, D, 2 * I, device=dev) * 0.02).to(bf16) for _ in range(2)]
dy1 = [torch.randn(M, 2 * I, device=dev).to(bf16) for _ in range(2)]
gw = torch.empty(E, D, 2 * I, dtype=bf16, device=dev)
x = [torch.randn(16384, 2560, device=dev).to(bf16) for _ in range(2)]
wq = [(torch.randn(7680, 2560, device=dev) * 0.02).to(bf16) for _ in range(2)]
x8 = torch.randn(8192, 8192, device=dev).to(bf16); w8 = (torch.randn(8192, 8192, device=dev) * 0.02).to(bf16)
f1 = 2 * M * D * 2 * I
cases = {
    "dense 8192^3": (2 * 8192 ** 3, lambda i: ops.gemm(x8, w8)),
    "dense 16384x7680x2560": (2 * 16384 * 7680 * 2560, lambda i: ops.gemm(x[i % 2], wq[i % 2])),
    "fc1 + swiglu fused": (f1, lambda i: ops.grouped_gemm_swiglu(a[i % 2], w1[i % 2], offd, True)),
    "fc1 plain": (f1, lambda i: ops.grouped_gemm(a[i % 2], w1[i % 2], offd)),
    "fc1 dgrad": (f1, lambda i: ops.grouped_gemm(dy1[i % 2], w1[i % 2], offd, w_is_kn=False)),
    "fc1 wgrad": (f1, lambda i: ops.grouped_gemm_wgrad(a[i % 2], dy1[i % 2], offd, E, out=gw)),
}
for rep in range(2):
    for order in (A, B):
        os.environ["ARIA_GEMM_PERSIST"] = order
        for name, (fl, fn) in cases.items():
            it = [0]
            def call():
                fn(it[0]); it[0] += 1
            res.setdefault(f"{name} | persist {order}", []).append(round(fl / timeit(call, 10, 3) / 1e12, 1))
print(json.dumps(res))
