"""The same dense bf16 GEMMs through this library (gemm3) and through torch.matmul (hipBLASLt / rocBLAS, whatever PyTorch-ROCm picks) on the
same box, interleaved -- a reference point under the same power limit, not a product path."""
import json, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from aria_amd import ops  # noqa: E402
from tools.microbench import timeit  # noqa: E402
bf16, dev, res = torch.bfloat16, "cuda", {}
for M, N, K in ((8192, 8192, 8192), (16384, 8192, 2560), (16384, 7680, 2560), (16384, 2560, 2560), (78400, 4304, 1152), (78400, 1152, 4304)):
    xs = [torch.randn(M, K, device=dev).to(bf16) for _ in range(2)]
    ws = [(torch.randn(N, K, device=dev) * 0.02).to(bf16) for _ in range(2)]
    out = torch.empty(M, N, dtype=bf16, device=dev)
    f = 2 * M * N * K
    r = {}
    for rep in range(2):
        i = [0]
        def ours():
            ops.gemm(xs[i[0] % 2], ws[i[0] % 2], out=out); i[0] += 1
        def vendor():
            torch.matmul(xs[i[0] % 2], ws[i[0] % 2].t(), out=out); i[0] += 1
        r.setdefault("aria gemm3", []).append(round(f / timeit(ours, 10, 3) / 1e12, 1))
        r.setdefault("torch.matmul", []).append(round(f / timeit(vendor, 10, 3) / 1e12, 1))
    res[f"{M}x{N}x{K}"] = r
    del xs, ws, out
print(json.dumps(res))
