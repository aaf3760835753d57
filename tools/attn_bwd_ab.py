"""attention backward: generation 3 (role-split, 2 waves/SIMD) vs generation 2 on hardware -- equality of results and timing"""
import os, sys, json, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from aria_amd import ops
from tools.microbench import timeit
bf16 = torch.bfloat16; dev = "cuda"
res = {}
def run(name, B, S, H, hd, causal, masked=False):
    D = H * hd
    torch.manual_seed(0)
    qkv = torch.randn(B * S, 3 * D, device=dev).to(bf16)
    km = None
    if masked:
        km = torch.ones(B, S, dtype=torch.uint8, device=dev); km[0, S * 3 // 4:] = 0
    fl = 4 * B * H * S * S * hd / (2 if causal else 1)
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    o, lse = ops.attention_fwd(q, k, v, B, S, H, hd, hd ** -0.5, causal, key_mask=km)
    do = torch.randn_like(o)
    f = lambda: ops.attention_bwd(q, k, v, o, do, lse, B, S, H, hd, hd ** -0.5, causal, key_mask=km)
    os.environ["ARIA_ATTN_BWD"] = "2"
    ref = [x.clone() for x in f()]
    t2 = timeit(f, 5, 2)
    os.environ.pop("ARIA_ATTN_BWD")
    bad = 0
    for rep in range(3):
        got = f()
        bad += sum(int(not torch.equal(a, b)) for a, b in zip(got, ref))
    maxdiff = max(float((a.float() - b.float()).abs().max()) for a, b in zip(got, ref))
    t3 = timeit(f, 5, 2)
    res[name] = dict(ms_v2=round(t2 * 1e3, 3), ms_v3=round(t3 * 1e3, 3), tflops_v3=round(2.5 * fl / t3 / 1e12, 1), mismatching_tensors=bad, maxdiff=maxdiff)
    print(json.dumps({name: res[name]}), flush=True)
run("llm_8x2048_h20_d128_causal", 8, 2048, 20, 128, True)
run("noncausal_4x2048_h20_d128_masked", 4, 2048, 20, 128, False, True)
run("odd_3x1225_h2_d128_causal", 3, 1225, 2, 128, True)
run("long_1x16384_h20_d128_causal", 1, 16384, 20, 128, True)
os.makedirs("gpurun_out", exist_ok=True); json.dump(res, open("gpurun_out/attn_bwd_ab.json", "w"), indent=1)
