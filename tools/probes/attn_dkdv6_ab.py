"""attn_bwd6_dkdv_kernel (ARIA_ATTN_DKDV=6: role A one query tile ahead, one barrier per tile) against the default dK/dV kernel, one process,
interleaved: whole aria_attn_bwd (delta + dK/dV + dQ) in ms, dK / dV / dQ compared bit for bit."""
import json
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from aria_amd import ops  # noqa: E402

bf16 = torch.bfloat16
dev = "cuda"
out = {}
for name, B, S, H in (("llm_8x2048_h20_d128", 8, 2048, 20), ("llm_1x16384_h20_d128", 1, 16384, 20), ("llm_1x65536_h20_d128", 1, 65536, 20)):
    hd, D = 128, H * 128
    g = torch.Generator(device=dev).manual_seed(S)
    qkv = torch.randn(B * S, 3 * D, generator=g, device=dev).to(bf16)
    do = (torch.randn(B * S, D, generator=g, device=dev) * 0.5).to(bf16)
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    o, lse = ops.attention_fwd(q, k, v, B, S, H, hd, hd ** -0.5, True, None)
    res, grads = {"default": [], "v6": []}, {}
    for rep in range(4):
        for ver in ("default", "v6"):
            if ver == "v6":
                os.environ["ARIA_ATTN_DKDV"] = "6"
            else:
                os.environ.pop("ARIA_ATTN_DKDV", None)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            dq, dk, dv = ops.attention_bwd(q, k, v, o, do, lse, B, S, H, hd, hd ** -0.5, True, None)
            e.record()
            torch.cuda.synchronize()
            if rep:
                res[ver].append(round(s.elapsed_time(e), 4))
            grads[ver] = (dq, dk, dv)
    os.environ.pop("ARIA_ATTN_DKDV", None)
    out[name] = {"default_ms": res["default"], "v6_ms": res["v6"],
                 "bit_identical": all(bool(torch.equal(a, b)) for a, b in zip(grads["default"], grads["v6"])),
                 "finite": bool(all(torch.isfinite(t.float()).all() for t in grads["v6"]))}
    del qkv, do, o, lse, grads
print(json.dumps(out))
