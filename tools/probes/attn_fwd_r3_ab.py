"""Attention forward: the in-step kernel of round 2 (ARIA_ATTN_FWD=2) against the phase-staggered one (fwd3: the second wave of every SIMD
runs its P V product one tile late), same box, interleaved: equality of results and HIP-event timing.  -> gpurun_out/attn_fwd_r3_ab.json"""
import json, os, sys
import torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from aria_amd import ops  # noqa: E402
bf16, dev, res = torch.bfloat16, "cuda", {}


def timeit(f, it, warm=2):
    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it


def run(name, B, S, H, hd, causal, masked=False, it=10):
    D = H * hd
    torch.manual_seed(0)
    qkv = torch.randn(B * S, 3 * D, device=dev).to(bf16)
    km = None
    if masked:
        km = torch.ones(B, S, dtype=torch.uint8, device=dev)
        km[0, S * 3 // 4:] = 0
    fl = 4 * B * H * S * S * hd / (2 if causal else 1)
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    f = lambda: ops.attention_fwd(q, k, v, B, S, H, hd, hd ** -0.5, causal, key_mask=km)
    r = {}
    for rep in range(2):
        os.environ["ARIA_ATTN_FWD"] = "2"
        ref = f()
        r.setdefault("in_step_ms", []).append(round(timeit(f, it), 4))
        os.environ.pop("ARIA_ATTN_FWD")
        got = f()
        r.setdefault("staggered_ms", []).append(round(timeit(f, it), 4))
    r["bit_identical"] = bool(torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]))
    r["TF_s_in_step"] = round(fl / min(r["in_step_ms"]) / 1e9, 1)
    r["TF_s_staggered"] = round(fl / min(r["staggered_ms"]) / 1e9, 1)
    res[name] = r
    print(json.dumps({name: r}), flush=True)


run("vit_16x4900_h16_d72_masked", 16, 4900, 16, 72, False, True)
os.environ["ARIA_ATTN_HD72_WAVES"] = "8"
run("vit_16x4900_h16_d72_masked_8waves", 16, 4900, 16, 72, False, True)
os.environ.pop("ARIA_ATTN_HD72_WAVES")
run("llm_8x2048_h20_d128_causal", 8, 2048, 20, 128, True)
run("llm_1x16384_h20_d128_causal", 1, 16384, 20, 128, True)
run("llm_1x65536_h20_d128_causal", 1, 65536, 20, 128, True, it=3)
run("noncausal_4x2048_h20_d128", 4, 2048, 20, 128, False)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "attn_fwd_r3_ab.json"), "w"), indent=1)
