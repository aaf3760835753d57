"""v3 GEMM ablation (flags in bits 12..15 of ARIA_GEMM_ORDER; results are wrong by construction, only the time matters)
NOTE: the skip flags this tool drives (bits 12..15 of ARIA_GEMM_ORDER in gemm3.hip) were compiled in only for the measurement
recorded in profiles/r01_gemm_tuning.md / r01_gemm3_ablate.json and are not in the product kernel; re-apply them to re-run.
"""
import os, sys, json, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from aria_amd import ops
from tools.microbench import timeit
bf16 = torch.bfloat16; dev = "cuda"
os.environ["ARIA_GEMM_FORCE"] = "3"
ops.GEMM_SPLIT_K = False
M = N = 8192; K = 8192
a = torch.randn(M, K, device=dev).to(bf16); b = torch.randn(N, K, device=dev).to(bf16); bt = b.t().contiguous()
res = {}
for layout, (bb, boc) in (("rc_rc", (b, False)), ("rc_oc", (bt, True))):
    for late in (0, 1):
        for name, abl in (("full", 0), ("no_mfma", 1), ("no_frag_reads", 2), ("no_dma", 4), ("no_barrier", 8), ("mfma_only", 2 | 4), ("mfma_only_nobar", 2 | 4 | 8),
                          ("reads_only", 1 | 4), ("dma_only", 1 | 2), ("barriers_only", 1 | 2 | 4)):
            os.environ["ARIA_GEMM_ORDER"] = str(4 + 256 * late + 4096 * abl)
            t = timeit(lambda: ops.gemm(a, bb, b_oc=boc), 5, 2)
            rounds = -(-((M // 256) * (N // 256)) // 256)
            res[f"{layout}_late{late}_{name}"] = round(t * 2.1e9 / (rounds * (K // 64)))   # cycles per K-tile per CU at 2.1 GHz
print(json.dumps(res, indent=0))
os.makedirs("gpurun_out", exist_ok=True); json.dump(res, open("gpurun_out/gemm3_ablate.json", "w"), indent=1)
