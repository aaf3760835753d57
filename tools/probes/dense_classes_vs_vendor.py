"""Every plain dense GEMM class of the config #3 step (no fused epilogue) through this library and through torch.matmul (the vendor library
PyTorch-ROCm picks: hipBLASLt / rocBLAS), same box, interleaved: where -- if anywhere -- a plain library GEMM is the better tool
(the task allows hipBLASLt for plain library GEMMs; fused and grouped launches have no library counterpart).  One JSON line: TF/s."""
import json
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from aria_amd import ops  # noqa: E402
from tools.microbench import timeit  # noqa: E402

bf16, dev, res = torch.bfloat16, "cuda", {}
T, D, Is2, V, TL = 16384, 2560, 6656, 100352, 4096       # tokens, hidden, shared gate|up width, vocabulary, labelled rows
P, Dv, Iv = 78400, 1152, 4304                             # ViT: patches of 16 images, width, MLP width
# name: (M, N, K, a_oc, b_oc)   ours: gemm(a, b, a_oc, b_oc);  rc,rc = x @ W^T;  rc,oc = dy @ W;  oc,oc = dy^T @ x
CLASSES = {
    "qkv fwd [T,7680]x2560 rc,rc": (T, 3 * D, D, False, False), "o_proj fwd [T,2560]x2560 rc,rc": (T, D, D, False, False),
    "shared down fwd [T,2560]x3328 rc,rc": (T, D, Is2 // 2, False, False), "lm_head fwd [4096,100352]x2560 rc,rc": (TL, V, D, False, False),
    "qkv dgrad [T,2560]x7680 rc,oc": (T, D, 3 * D, False, True), "o_proj dgrad [T,2560]x2560 rc,oc": (T, D, D, False, True),
    "shared gate|up dgrad [T,2560]x6656 rc,oc": (T, D, Is2, False, True), "lm_head dgrad [4096,2560]x100352 rc,oc": (TL, D, V, False, True),
    "qkv wgrad [7680,2560]xT oc,oc": (3 * D, D, T, True, True), "o_proj wgrad [2560,2560]xT oc,oc": (D, D, T, True, True),
    "shared gate|up wgrad [6656,2560]xT oc,oc": (Is2, D, T, True, True), "shared down wgrad [2560,3328]xT oc,oc": (D, Is2 // 2, T, True, True),
    "lm_head wgrad [100352,2560]x4096 oc,oc": (V, D, TL, True, True),
    "vit qkv fwd [78400,3456]x1152 rc,rc": (P, 3 * Dv, Dv, False, False), "vit o_proj fwd [78400,1152]x1152 rc,rc": (P, Dv, Dv, False, False),
    "vit fc2 fwd [78400,1152]x4304 rc,rc": (P, Dv, Iv, False, False),
}
for name, (M, N, K, a_oc, b_oc) in CLASSES.items():
    a = [(torch.randn((K, M) if a_oc else (M, K), device=dev)).to(bf16) for _ in range(2)]
    b = [(torch.randn((K, N) if b_oc else (N, K), device=dev) * 0.02).to(bf16) for _ in range(2)]
    out = torch.empty(M, N, dtype=bf16, device=dev)
    f = 2.0 * M * N * K
    r = {"ours": [], "vendor": []}
    for rep in range(2):
        i = [0]

        def ours():
            ops.gemm(a[i[0] % 2], b[i[0] % 2], a_oc=a_oc, b_oc=b_oc, out=out)
            i[0] += 1

        def vendor():
            x, w = a[i[0] % 2], b[i[0] % 2]
            torch.matmul(x.t() if a_oc else x, w if b_oc else w.t(), out=out)
            i[0] += 1

        r["ours"].append(round(f / timeit(ours, 8, 2) / 1e12, 1))
        r["vendor"].append(round(f / timeit(vendor, 8, 2) / 1e12, 1))
    r["ms_ours"] = round(f / max(r["ours"]) / 1e9, 4)
    r["ms_vendor"] = round(f / max(r["vendor"]) / 1e9, 4)
    res[name] = r
    del a, b, out
print(json.dumps(res))
