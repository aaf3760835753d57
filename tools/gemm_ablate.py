"""NOTE: the skip flags this tool drives (bits 8..11 of ARIA_GEMM_ORDER in gemm2.hip) were compiled in only for the measurement
recorded in profiles/r01_gemm_tuning.md and have been removed from the product kernel again; re-apply them to re-run.
"""
import os, sys, json, torch
sys.path.insert(0, ".")
from aria_amd import ops
from tools.microbench import timeit
bf16 = torch.bfloat16; dev = "cuda"
os.environ["ARIA_GEMM_FORCE"] = "2"
M, N, K = 16384, 16384, 2560
a = torch.randn(M, K, device=dev).to(bf16); b = torch.randn(N, K, device=dev).to(bf16)
res = {}
for name, abl in (("full", 0), ("no_lds_store", 1), ("no_global_load", 2), ("no_store_no_load", 3), ("no_barrier", 4), ("no_compute", 8),
                  ("only_compute", 7), ("only_compute_with_barrier", 3)):
    os.environ["ARIA_GEMM_ORDER"] = str(2 + (abl << 8))
    t = timeit(lambda: ops.gemm(a, b), 5, 2)
    res[name] = dict(us=round(t * 1e6), cyc_per_kstep=round(t * 2.0e9 / (16 * 40)))
print(json.dumps(res))
