import os, sys, json, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from aria_amd import ops
from tools.microbench import timeit
bf16 = torch.bfloat16; dev = "cuda"
B, S, H, hd = 16, 4900, 16, 72
D = H * hd
torch.manual_seed(0)
qkv = torch.randn(B * S, 3 * D, device=dev).to(bf16)
km = torch.ones(B, S, dtype=torch.uint8, device=dev); km[0, S * 3 // 4:] = 0
fl = 4 * B * H * S * S * hd
res = {}
for name, mask in (("plain", None), ("masked", km)):
    f = lambda: ops.attention_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B, S, H, hd, hd ** -0.5, False, key_mask=mask)
    os.environ["ARIA_ATTN_HD72_WAVES"] = "8"
    ref = [x.clone() for x in f()]; t8 = timeit(f, 5, 2)
    os.environ.pop("ARIA_ATTN_HD72_WAVES")
    got = f(); bad = sum(int(not torch.equal(a, b)) for a, b in zip(got, ref)); t12 = timeit(f, 5, 2)
    res[name] = dict(ms_8w=round(t8 * 1e3, 3), ms_12w=round(t12 * 1e3, 3), tflops_12w=round(fl / t12 / 1e12, 1), mismatching=bad)
print(json.dumps(res))
