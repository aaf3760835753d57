import sys, json, torch
sys.path.insert(0, ".")
from aria_amd import ops
from tools.microbench import timeit
bf16 = torch.bfloat16; dev = "cuda"
res = {}
def run(name, B, S, H, hd, causal, masked=False):
    D = H * hd
    qkv = torch.randn(B * S, 3 * D, device=dev).to(bf16)
    km = None
    if masked:
        km = torch.ones(B, S, dtype=torch.uint8, device=dev); km[0, S * 3 // 4:] = 0
    fl = 4 * B * H * S * S * hd / (2 if causal else 1)
    t = timeit(lambda: ops.attention_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B, S, H, hd, hd ** -0.5, causal, key_mask=km), 5, 2)
    res[name + "_fwd"] = dict(ms=round(t * 1e3, 3), tflops=round(fl / t / 1e12, 1))
    if hd in (64, 128) and "--bwd" in sys.argv:
        o, lse = ops.attention_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B, S, H, hd, hd ** -0.5, causal, key_mask=km)
        do = torch.randn_like(o)
        t = timeit(lambda: ops.attention_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], o, do, lse, B, S, H, hd, hd ** -0.5, causal, key_mask=km), 5, 2)
        res[name + "_bwd"] = dict(ms=round(t * 1e3, 3), tflops=round(2.5 * fl / t / 1e12, 1))
run("llm_8x2048_h20_d128_causal", 8, 2048, 20, 128, True)
run("vit_16x4900_h16_d72", 16, 4900, 16, 72, False)
run("vit_16x4900_h16_d72_masked", 16, 4900, 16, 72, False, True)
run("vit_16x4900_h16_d128", 16, 4900, 16, 128, False)
run("long_1x16384_h20_d128_causal", 1, 16384, 20, 128, True)
print(json.dumps(res, indent=1))
