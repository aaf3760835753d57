import os, sys, json, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from aria_amd import ops, hip
from tools.microbench import timeit
bf16 = torch.bfloat16; dev = "cuda"
E, T, topk = 64, 16384, 6
g = torch.Generator().manual_seed(1)
counts = torch.bincount(torch.randint(0, E, (T * topk,), generator=g), minlength=E)
off = torch.zeros(E + 1, dtype=torch.int32); off[1:] = torch.cumsum(counts, 0); M = int(off[-1]); offd = off.to(dev)
res = {}
for name, K, N in (("fc1", 2560, 3328), ("fc2", 1664, 2560)):
    a = torch.randn(M, K, device=dev).to(bf16)
    w = (torch.randn(E, K, N, device=dev) * 0.02).to(bf16)
    for force in ("2", "3"):
        os.environ["ARIA_GEMM_FORCE"] = force
        for order in ("2", "4", "8"):
            os.environ["ARIA_GEMM_ORDER"] = order
            t = timeit(lambda: ops.grouped_gemm(a, w, offd), 10, 3)
            res[f"{name}_v{force}_o{order}"] = [round(2 * M * K * N / t / 1e12), round(t * 1e6)]
    # uniform counts, tile aligned: isolates the ragged-edge cost
    off2 = torch.arange(E + 1, dtype=torch.int32) * 1536
    os.environ["ARIA_GEMM_ORDER"] = "4"
    for force in ("2", "3"):
        os.environ["ARIA_GEMM_FORCE"] = force
        t = timeit(lambda: ops.grouped_gemm(a, w, off2.to(dev)), 10, 3)
        res[f"{name}_v{force}_aligned"] = [round(2 * M * K * N / t / 1e12), round(t * 1e6)]
print(json.dumps(res))
