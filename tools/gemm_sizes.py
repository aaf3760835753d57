import os, sys, json, torch
sys.path.insert(0, ".")
from aria_amd import ops
from tools.microbench import timeit
bf16 = torch.bfloat16; dev = "cuda"
os.environ["ARIA_GEMM_FORCE"] = "2"
res = {}
for (M, N, K) in ((2048, 2048, 2048), (2048, 2048, 8192), (4096, 4096, 4096), (8192, 8192, 2560), (16384, 2560, 2560), (16384, 16384, 2560)):
    a = torch.randn(M, K, device=dev).to(bf16); b = torch.randn(N, K, device=dev).to(bf16)
    t = timeit(lambda: ops.gemm(a, b), 10, 3)
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    ksteps = K // 64
    waves = -(-tiles // 256)
    res[f"{M}x{N}x{K}"] = dict(tf=round(2 * M * N * K / t / 1e12), us=round(t * 1e6, 1), tiles=tiles, cyc_per_kstep_at_2GHz=round(t * 2.0e9 / (waves * ksteps)))
print(json.dumps(res, indent=0))
