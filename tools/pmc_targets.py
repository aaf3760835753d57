"""Targets of the round-4 PMC passes (tools/gpu_pmc.sh: rocprofv3 --pmc in separate runs, --kernel-trace only): each launches ONE kernel class
of the step a few times at the step's shape.

    python tools/pmc_targets.py fc1        # experts.fc1 + SwiGLU epilogue, config #3 shape (the bench's roofline kernel)
    python tools/pmc_targets.py attn_bwd   # decoder attention backward (delta + dK/dV + dQ), ONE 65 536-token causal sequence, 20 x 128
    python tools/pmc_targets.py vit_fwd    # ViT attention forward, 16 images x 4900 patches, 16 x 72 (attn_fwd2_kernel<72, 12>)
"""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from aria_amd import ops  # noqa: E402

bf16 = torch.bfloat16
dev = "cuda"
what = sys.argv[1]
g = torch.Generator(device=dev).manual_seed(0)


def rn(*shape, scale=1.0):
    return (torch.randn(shape, generator=g, device=dev) * scale).to(bf16)


if what == "fc1":
    T, D, I, E, k = 16384, 2560, 1664, 64, 6
    x, logits = rn(T, D), rn(T, E)
    scores, idx, counts = ops.moe_route(logits, k)
    off, sorted_src, inv = ops.moe_sort(idx, counts)
    fc1 = rn(E, D, 2 * I, scale=0.02)
    rows = ops.permuted_token_rows(sorted_src, k)   # (r05: the training step's fc1 launch takes the un-permuted tokens + the dispatcher's index)
    for _ in range(3):
        h, act = ops.grouped_gemm_swiglu_gather(x, rows, fc1, off, True)
elif what == "attn_bwd":
    B, S, H, hd = 1, 65536, 20, 128
    D = H * hd
    qkv, do = rn(B * S, 3 * D), rn(B * S, D)
    q, k_, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    o, lse = ops.attention_fwd(q, k_, v, B, S, H, hd, hd ** -0.5, True, None)
    for _ in range(2):
        ops.attention_bwd(q, k_, v, o, do, lse, B, S, H, hd, hd ** -0.5, True, None)
elif what == "vit_fwd":
    B, S, H, hd = 16, 4900, 16, 72
    D = H * hd
    qkv = rn(B * S, 3 * D)
    km = torch.ones(B, S, dtype=torch.uint8, device=dev)
    km[0, 3675:] = 0      # one image with its bottom quarter padded, as in the benchmark batch
    for _ in range(3):
        ops.attention_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B, S, H, hd, hd ** -0.5, False, key_mask=km)
else:
    raise SystemExit(f"unknown target {what}")
torch.cuda.synchronize()
