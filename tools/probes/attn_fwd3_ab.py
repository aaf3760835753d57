"""attn_fwd3 (software-pipelined score tiles, K / V by LDS-DMA) against attn_fwd2 on one box, interleaved in one process (ARIA_ATTN_FWD is
read per call): ms per launch, TF/s of the algorithmic flops, and whether the two are bit-identical.  One JSON line."""
import json
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from aria_amd import hip, ops  # noqa: E402

bf16, dev = torch.bfloat16, "cuda"
g = torch.Generator(device=dev).manual_seed(0)
SHAPES = {  # B, Sq, Skv, H, hd, causal, masked, iters
    "vit_16x4900_h16_d72_masked": (16, 4900, 4900, 16, 72, False, True, 8),
    "projector_16x256q_4900k_h16_d72_masked": (16, 256, 4900, 16, 72, False, True, 20),
    "llm_8x2048_h20_d128_causal": (8, 2048, 2048, 20, 128, True, False, 20),
    "llm_1x16384_h20_d128_causal": (1, 16384, 16384, 20, 128, True, False, 6),
    "llm_1x53248_h20_d128_causal": (1, 53248, 53248, 20, 128, True, False, 3),
    "llm_1x65536_h20_d128_causal": (1, 65536, 65536, 20, 128, True, False, 3),
}
res = {}
for name, (B, Sq, Skv, H, hd, causal, masked, iters) in SHAPES.items():
    D = H * hd
    q = torch.randn((B * Sq, D), generator=g, device=dev).to(bf16)
    kv = torch.randn((B * Skv, 2 * D), generator=g, device=dev).to(bf16)
    km = None
    if masked:
        km = torch.ones(B, Skv, dtype=torch.uint8, device=dev)
        km[0, Skv * 3 // 4:] = 0
    fl = 4.0 * B * H * Sq * Skv * hd * (0.5 if causal else 1.0)
    vers = ("2", "3", "3p") if hd == 72 else ("2", "3")   # hd 72: "3" = 12 waves, LDS-DMA, scores in step; "3p" = 8 waves, pipelined
    out, times = {}, {v: [] for v in vers}
    for rep in range(2):
        for ver in vers:
            os.environ["ARIA_ATTN_FWD"] = ver
            f = lambda: ops.attention_fwd(q, kv[:, :D], kv[:, D:], B, Sq, H, hd, hd ** -0.5, causal, key_mask=km, Skv=Skv)
            o, lse = f()
            assert hip.get_lib().cdll.aria_last_attn_fwd_variant() == int(ver[0])
            out[ver] = (o, lse)
            f()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(iters):
                f()
            e.record()
            torch.cuda.synchronize()
            times[ver].append(round(s.elapsed_time(e) / iters, 4))
    same = all(bool(torch.equal(out["2"][0], out[v][0]) and torch.equal(out["2"][1], out[v][1])) for v in vers)
    res[name] = {**{f"v{v}_ms": times[v] for v in vers}, **{f"v{v}_TF_s": round(fl / min(times[v]) / 1e9, 1) for v in vers},
                 "bit_identical": same, "finite": bool(torch.isfinite(out["3"][0].float()).all())}
    del q, kv, out
os.environ.pop("ARIA_ATTN_FWD", None)
print(json.dumps(res))
