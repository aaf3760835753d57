"""Grouped GEMM classes with the default tile order and with the ragged-last order (ARIA_GEMM_ORDER=516), same process, interleaved."""
import json, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from aria_amd import ops  # noqa: E402
from tools.microbench import timeit  # noqa: E402
bf16, dev, res = torch.bfloat16, "cuda", {}
E, T, topk, D, I = 64, 16384, 6, 2560, 1664
g = torch.Generator().manual_seed(1)
counts = torch.bincount(torch.randint(0, E, (T * topk,), generator=g), minlength=E)
off = torch.zeros(E + 1, dtype=torch.int32); off[1:] = torch.cumsum(counts, 0)
M = int(off[-1]); offd = off.to(dev)
a = [torch.randn(M, D, device=dev).to(bf16) for _ in range(2)]
w1 = [(torch.randn(E, D, 2 * I, device=dev) * 0.02).to(bf16) for _ in range(3)]
w2 = [(torch.randn(E, I, D, device=dev) * 0.02).to(bf16) for _ in range(3)]
h = [torch.randn(M, I, device=dev).to(bf16) for _ in range(2)]
dy1 = [torch.randn(M, 2 * I, device=dev).to(bf16) for _ in range(2)]
din = torch.empty(M, D, dtype=bf16, device=dev)
f1, f2 = 2 * M * D * 2 * I, 2 * M * I * D
cases = {
    "fc1 + swiglu fused": (f1, lambda i: ops.grouped_gemm_swiglu(a[i % 2], w1[i % 3], offd, True)),
    "fc1 plain": (f1, lambda i: ops.grouped_gemm(a[i % 2], w1[i % 3], offd)),
    "fc2 fwd": (f2, lambda i: ops.grouped_gemm(h[i % 2], w2[i % 3], offd)),
    "fc1 dgrad": (f1, lambda i: ops.grouped_gemm(dy1[i % 2], w1[i % 3], offd, w_is_kn=False, out=din)),
}
for rep in range(2):
    for order in ("4", "516", "2564"):   # default; ragged-last (bit 9); ragged-first (bits 9 + 11)
        os.environ["ARIA_GEMM_ORDER"] = order
        for name, (fl, fn) in cases.items():
            it = [0]
            def call():
                fn(it[0]); it[0] += 1
            t = timeit(call, 12, 3)
            res.setdefault(f"{name} order {order}", []).append(round(fl / t / 1e12, 1))
print(json.dumps(res))
