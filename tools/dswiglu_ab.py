"""Same-process A/B of the fused input-gradient + SwiGLU-backward launches (gemm3_kernel<.., .., 5>) against the two-step chains they
replace, at the shapes of one Aria decoder layer of the config-#3 micro-batch (16 384 tokens):
  routed:  d_eo [98304, 2560] x fc2.weight [64, 1664, 2560]^T, h1 [98304, 3328]   (routed counts from a random top-6 routing)
  shared:  dout [16384, 2560] x down_proj.weight [2560, 3328],  gu [16384, 6656]
HIP events around interleaved repetitions; also checks that the fused results equal the chain's bit for bit on hardware."""
import json
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from aria_amd import ops  # noqa: E402

bf16 = torch.bfloat16
dev = torch.device("cuda")
g = torch.Generator(device="cuda").manual_seed(0)


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, generator=g, device=dev) * scale).to(bf16)


def timed(fn, reps):
    ev = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        ev.append((s, e))
    torch.cuda.synchronize()
    t = sorted(s.elapsed_time(e) for s, e in ev)
    return round(t[len(t) // 2] * 1e3, 1)  # median, us


T, D, E, k, I, I2 = 16384, 2560, 64, 6, 1664, 3328
res = {}
# routed
logits = torch.randn(T, E, generator=g, device=dev).to(bf16)
scores, idx, counts = ops.moe_route(logits, k)
offsets, sorted_src, inv = ops.moe_sort(idx, counts)
M = T * k
dy, w, h = rnd(M, D, scale=0.1), rnd(E, I, D, scale=0.02), rnd(M, 2 * I, scale=1.0)


def chain_routed():
    return ops.swiglu_bwd(h, ops.grouped_gemm(dy, w, offsets, w_is_kn=False))


def fused_routed():
    return ops.grouped_gemm_dswiglu(dy, w, offsets, h)


res["routed_equal"] = bool(torch.equal(chain_routed(), fused_routed()))
for _ in range(2):
    chain_routed(), fused_routed()
a, b = [], []
for _ in range(3):
    a.append(timed(chain_routed, 5))
    b.append(timed(fused_routed, 5))
res["routed_chain_us"], res["routed_fused_us"] = a, b
res["routed_dgrad_alone_us"] = timed(lambda: ops.grouped_gemm(dy, w, offsets, w_is_kn=False), 5)
# shared
dy2, w2, h2 = rnd(T, D, scale=0.1), rnd(D, I2, scale=0.02), rnd(T, 2 * I2, scale=1.0)


def chain_shared():
    return ops.swiglu_bwd(h2, ops.gemm(dy2, w2, b_oc=True))


def fused_shared():
    return ops.gemm_dswiglu(dy2, w2, h2, b_oc=True)


ops.GEMM_SPLIT_K = False  # bit-for-bit comparison against the chain WITHOUT the remainder split-K (another fp32 summation order)
res["shared_equal_no_splitk"] = bool(torch.equal(chain_shared(), fused_shared()))
ops.GEMM_SPLIT_K = True
d = (chain_shared().float() - fused_shared().float()).abs().max()
res["shared_max_abs_diff_vs_splitk_chain"] = float(d)
for _ in range(2):
    chain_shared(), fused_shared()
a, b = [], []
for _ in range(3):
    a.append(timed(chain_shared, 5))
    b.append(timed(fused_shared, 5))
res["shared_chain_us"], res["shared_fused_us"] = a, b
res["per_step_gain_ms_28_layers"] = round(28 * ((min(res["routed_chain_us"]) - min(res["routed_fused_us"])) +
                                                (min(res["shared_chain_us"]) - min(res["shared_fused_us"]))) * 1e-3, 2)
print(json.dumps(res))
