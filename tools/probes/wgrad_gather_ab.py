"""experts.fc1's weight gradient at the config #3 shape (98 304 routed rows, 64 experts, K 2560, N 3328): the gathered launch (token rows through
the dispatcher's index, gemm3_kernel<true,true,11>) vs permute + the plain launch, and the fused fc1 + SwiGLU forward on gathered vs permuted rows --
interleaved in one process, HIP events."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aria_amd import ops  # noqa: E402

dev = torch.device("cuda")
g = torch.Generator(device="cuda").manual_seed(0)
T, D, I, E, k = 16384, 2560, 1664, 64, 6
bf16 = torch.bfloat16


def rn(*shape, scale=1.0):
    return (torch.randn(shape, generator=g, device=dev) * scale).to(bf16)


x, logits = rn(T, D), rn(T, E)
scores, idx, counts = ops.moe_route(logits, k)
off, sorted_src, inv = ops.moe_sort(idx, counts)
rows = ops.permuted_token_rows(sorted_src, k)
perm = ops.moe_permute(x, sorted_src, k)
dh = rn(T * k, 2 * I)
fc1 = rn(E, D, 2 * I, scale=0.02)


def timed(fn, n=10):
    for _ in range(2):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


res = {}
for rep in range(2):
    res[f"wgrad_plain_us_{rep}"] = round(timed(lambda: ops.grouped_gemm_wgrad(perm, dh, off, E)), 1)
    res[f"wgrad_gather_us_{rep}"] = round(timed(lambda: ops.grouped_gemm_wgrad_gather(x, rows, dh, off, E)), 1)
    res[f"fc1_swiglu_plain_us_{rep}"] = round(timed(lambda: ops.grouped_gemm_swiglu(perm, fc1, off, True)), 1)
    res[f"fc1_swiglu_gather_us_{rep}"] = round(timed(lambda: ops.grouped_gemm_swiglu_gather(x, rows, fc1, off, True)), 1)
    res[f"permute_us_{rep}"] = round(timed(lambda: ops.moe_permute(x, sorted_src, k)), 1)
flops = 2.0 * T * k * D * 2 * I
res["TFs"] = {key: round(flops / (v * 1e-6) / 1e12, 1) for key, v in res.items() if "permute" not in key}
print(json.dumps(res))
