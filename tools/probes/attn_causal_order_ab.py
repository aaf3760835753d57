"""Causal attention forward / backward at the step's shapes (8 x 2048, 1 x 16 384, 1 x 65 536; 20 heads x 128) with the library in the tree -- run under
tools/gpu_ab_lib.sh for a same-box A/B of the r05 XCD-grouped causal block order (HEAD) against rounds 1-4's (build/old: -DARIA_ATTN_CAUSAL_ORDER=0)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aria_amd import ops  # noqa: E402

bf16, dev = torch.bfloat16, "cuda"
res = {}


def timed(fn, n):
    for _ in range(2):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


for B, S, n in ((8, 2048, 20), (1, 16384, 6), (1, 65536, 3)):
    H, hd = 20, 128
    D = H * hd
    qkv = torch.randn(B * S, 3 * D, device=dev).to(bf16)
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    o, lse = ops.attention_fwd(q, k, v, B, S, H, hd, hd ** -0.5, True)
    do = torch.randn_like(o)
    res[f"{B}x{S}_fwd_ms"] = round(timed(lambda: ops.attention_fwd(q, k, v, B, S, H, hd, hd ** -0.5, True), n), 4)
    res[f"{B}x{S}_bwd_ms"] = round(timed(lambda: ops.attention_bwd(q, k, v, o, do, lse, B, S, H, hd, hd ** -0.5, True), n), 4)
    del qkv, o, do, lse
print(json.dumps(res))
