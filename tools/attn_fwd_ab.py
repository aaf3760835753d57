"""attention forward: generation 3 (phase-staggered wave groups) vs generation 2 -- equality and timing"""
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
    f = lambda: ops.attention_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B, S, H, hd, hd ** -0.5, causal, key_mask=km)
    os.environ["ARIA_ATTN_FWD"] = "2"
    ref = [x.clone() for x in f()]
    t2 = timeit(f, 5, 2)
    os.environ.pop("ARIA_ATTN_FWD")
    bad = 0
    for rep in range(3):
        got = f()
        bad += sum(int(not torch.equal(a, b)) for a, b in zip(got, ref))
    t3 = timeit(f, 5, 2)
    res[name] = dict(ms_v2=round(t2 * 1e3, 3), ms_v3=round(t3 * 1e3, 3), tflops_v3=round(fl / t3 / 1e12, 1), mismatching_tensors=bad)
    print(json.dumps({name: res[name]}), flush=True)
run("llm_8x2048_h20_d128_causal", 8, 2048, 20, 128, True)
run("vit_16x4900_h16_d72", 16, 4900, 16, 72, False)
run("vit_16x4900_h16_d72_masked", 16, 4900, 16, 72, False, True)
run("vit_16x4900_h16_d128", 16, 4900, 16, 128, False)
run("long_1x16384_h20_d128_causal", 1, 16384, 20, 128, True)
run("odd_3x1225_h2_d64", 3, 1225, 2, 64, False, True)
os.makedirs("gpurun_out", exist_ok=True); json.dump(res, open("gpurun_out/attn_fwd_ab.json", "w"), indent=1)
