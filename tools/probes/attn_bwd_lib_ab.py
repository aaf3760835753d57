"""r06: attention backward (hd 128, causal) at the training shape (8 x 2048, 20 heads), at 16 K and at 64 K tokens: the tree's library vs another
build of it (default build/abl/libaria_prev.so), SAME process, interleaved; dq / dk / dv of the two builds compared bit for bit.  One JSON line."""
import json, os, sys, ctypes
import torch
root = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, root)
from aria_amd import hip, ops  # noqa: E402

other = hip.HipLibrary(os.path.join(root, sys.argv[1] if len(sys.argv) > 1 else "build/abl/libaria_prev.so"))
tree = hip.get_lib()
bf16, dev = torch.bfloat16, "cuda"
res = {}
for name, B, S, H in (("train_8x2048", 8, 2048, 20), ("16k", 1, 16384, 20), ("64k", 1, 65536, 20)):
    hd = 128
    D = H * hd
    g = torch.Generator(device=dev).manual_seed(S)
    qkv = torch.randn(B * S, 3 * D, device=dev, generator=g).to(bf16)
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    o, lse = ops.attention_fwd(q, k, v, B, S, H, hd, hd ** -0.5, True)
    do = torch.randn(o.shape, device=dev, generator=g).to(bf16)
    outs, times = {}, {"tree": [], "other": []}
    for rep in range(3):
        for arm, lib in (("other", other), ("tree", tree)):
            hip._LIB = lib   # (ops.* go through hip.get_lib())
            for _ in range(2):
                r = ops.attention_bwd(q, k, v, o, do, lse, B, S, H, hd, hd ** -0.5, True)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 10 if S <= 16384 else 3
            a.record()
            for _ in range(n):
                r = ops.attention_bwd(q, k, v, o, do, lse, B, S, H, hd, hd ** -0.5, True)
            b.record()
            torch.cuda.synchronize()
            times[arm].append(round(a.elapsed_time(b) / n, 4))
            outs[arm] = [t.clone() for t in r]
    hip._LIB = tree
    fl = 2.5 * 4 * B * H * S * S * hd / 2
    res[name] = {"ms": times, "tflops_tree_best": round(fl / min(times["tree"]) / 1e9, 1), "tflops_other_best": round(fl / min(times["other"]) / 1e9, 1),
                 "bit_identical": all(torch.equal(x, y) for x, y in zip(outs["tree"], outs["other"]))}
print(json.dumps(res))
