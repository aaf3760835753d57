"""Permute / unpermute / unpermute-backward at Aria's width (T = 16 384 and 65 536 tokens, top-6, D = 2560): the compile-time-row-width kernels
against the generic ones (ARIA_MOE_GENERIC_DISPATCH=1), same process, interleaved.  One JSON line: microseconds and TB/s of algorithmic bytes."""
import json
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from aria_amd import ops  # noqa: E402
from tools.microbench import timeit  # noqa: E402

bf16, dev, res = torch.bfloat16, "cuda", {}
D, E, k = 2560, 64, 6
for T in (16384, 65536):
    g = torch.Generator(device=dev).manual_seed(T)
    x = torch.randn((T, D), generator=g, device=dev).to(bf16)
    logits = torch.randn((T, E), generator=g, device=dev).to(bf16)
    scores, idx, counts = ops.moe_route(logits, k)
    off, sorted_src, inv = ops.moe_sort(idx, counts)
    eo = torch.randn((T * k, D), generator=g, device=dev).to(bf16)
    shared = torch.randn((T, D), generator=g, device=dev).to(bf16)
    cases = {"permute": (lambda: ops.moe_permute(x, sorted_src, k), (T * D + T * k * D) * 2),
             "unpermute+shared": (lambda: ops.moe_unpermute(eo, inv, scores, k, add=shared), (T * k * D + 2 * T * D) * 2),
             "unpermute_bwd": (lambda: ops.moe_unpermute_bwd(x, eo, inv, scores, k), (T * D + 2 * T * k * D) * 2)}
    for name, (fn, nbytes) in cases.items():
        r = {}
        for rep in range(2):
            for tag, env in (("fixed_width", None), ("generic", "1")):
                if env:
                    os.environ["ARIA_MOE_GENERIC_DISPATCH"] = env
                else:
                    os.environ.pop("ARIA_MOE_GENERIC_DISPATCH", None)
                t = timeit(fn, 20, 3)
                r.setdefault(tag, []).append([round(t * 1e6, 1), round(nbytes / t / 1e12, 2)])
        res[f"T{T} {name}"] = r
    os.environ.pop("ARIA_MOE_GENERIC_DISPATCH", None)
    del x, eo, shared
print(json.dumps(res))
