import json, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from aria_amd import ops
from tools.microbench import timeit
bf16, dev, res = torch.bfloat16, "cuda", {}
x = [torch.randn(16384, 2560, device=dev).to(bf16) for _ in range(2)]
wq = [(torch.randn(7680, 2560, device=dev) * 0.02).to(bf16) for _ in range(2)]
w8k = [(torch.randn(8192, 2560, device=dev) * 0.02).to(bf16) for _ in range(2)]
x8 = torch.randn(8192, 8192, device=dev).to(bf16); w8 = (torch.randn(8192, 8192, device=dev) * 0.02).to(bf16)
xv = torch.randn(78400, 1152, device=dev).to(bf16); wv = (torch.randn(4304, 1152, device=dev) * 0.02).to(bf16)
cases = {"dense 8192^3": (2 * 8192 ** 3, lambda i: ops.gemm(x8, w8)),
         "dense 16384x7680x2560": (2 * 16384 * 7680 * 2560, lambda i: ops.gemm(x[i % 2], wq[i % 2])),
         "dense 16384x8192x2560": (2 * 16384 * 8192 * 2560, lambda i: ops.gemm(x[i % 2], w8k[i % 2])),
         "vit 78400x4304x1152": (2 * 78400 * 4304 * 1152, lambda i: ops.gemm(xv, wv))}
ref = {}
for rep in range(2):
    for v4, order in (("0", "4"), ("1", "4"), ("1", "1028")):
        os.environ["ARIA_GEMM_V4"] = v4; os.environ["ARIA_GEMM_ORDER"] = order
        for name, (fl, fn) in cases.items():
            it = [0]
            def call():
                fn(it[0]); it[0] += 1
            res.setdefault(f"{name} | v4={v4} order={order}", []).append(round(fl / timeit(call, 10, 3) / 1e12, 1))
            out = fn(0).float()
            key = name
            if key not in ref: ref[key] = out
            else: res[f"{name} | v4={v4} order={order} maxdiff"] = float((out - ref[key]).abs().max())
print(json.dumps(res))
