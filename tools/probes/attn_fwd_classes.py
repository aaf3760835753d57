"""ms / TF/s of the attention forward classes of the config #3 step with the library in the tree (run under tools/gpu_ab_lib.sh)."""
import json, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from aria_amd import ops  # noqa: E402
from tools.microbench import timeit  # noqa: E402
bf16, dev, res = torch.bfloat16, "cuda", {}
torch.manual_seed(0)
for name, (B, S, H, hd, causal) in {"vit 16 x 4900 x 16 heads x 72": (16, 4900, 16, 72, False), "decoder 8 x 2048 x 20 heads x 128 causal": (8, 2048, 20, 128, True),
                                    "long 1 x 65536 x 20 x 128 causal": (1, 65536, 20, 128, True)}.items():
    D = H * hd
    qkv = torch.randn(B * S, 3 * D, device=dev).to(bf16)
    f = lambda: ops.attention_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B, S, H, hd, hd ** -0.5, causal)
    t = timeit(f, 5, 2)
    fl = 4 * B * H * S * S * hd * (0.5 if causal else 1.0)
    res[name] = [round(t * 1e3, 3), round(fl / t / 1e12, 1)]
    if hd == 128:
        o, lse = f()
        do = torch.randn_like(o)
        g = lambda: ops.attention_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], o, do, lse, B, S, H, hd, hd ** -0.5, causal)
        tb = timeit(g, 4, 2)
        res[name + " bwd"] = [round(tb * 1e3, 3), round(2.5 * fl / tb / 1e12, 1)]
print(json.dumps(res))
