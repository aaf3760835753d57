"""r06: the six grouped launches of a MoE layer at the benchmark shape under two values of the tile-order word (ARIA_GEMM_ORDER; `default` =
unset), product library, interleaved repetitions, outputs compared bit for bit.  usage: order_bits_ab.py <orderA|default> <orderB> [<orderB for the
fused fc1 launch>]   One JSON line."""
import json, os, sys
import torch
root = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, root)
from aria_amd import ops  # noqa: E402
dev, bf16 = "cuda", torch.bfloat16
T, D, I, E, k = 16384, 2560, 1664, 64, 6
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn((T, D), generator=g, device=dev).to(bf16)
logits = torch.randn((T, E), generator=g, device=dev).to(bf16)
scores, idx, counts = ops.moe_route(logits, k)
off, sorted_src, inv = ops.moe_sort(idx, counts)
rows = ops.permuted_token_rows(sorted_src, k)
M = rows.numel()
w1 = (torch.randn((E, D, 2 * I), generator=g, device=dev) * 0.02).to(bf16)
w2 = (torch.randn((E, I, D), generator=g, device=dev) * 0.02).to(bf16)
h, act = ops.grouped_gemm_swiglu_gather(x, rows, w1, off, want_h=True)
dy = torch.randn((M, D), generator=g, device=dev).to(bf16)
dh = ops.grouped_gemm_dswiglu(dy, w2, off, h)
launches = {
    "fc1 + SwiGLU (gathered)": (lambda: ops.grouped_gemm_swiglu_gather(x, rows, w1, off, want_h=True)[1], 2.0 * M * D * 2 * I, True),
    "fc2 forward": (lambda: ops.grouped_gemm(act, w2, off), 2.0 * M * I * D, False),
    "fc2 dgrad + dSwiGLU": (lambda: ops.grouped_gemm_dswiglu(dy, w2, off, h), 2.0 * M * I * D, False),
    "fc1 dgrad": (lambda: ops.grouped_gemm(dh, w1, off, w_is_kn=False), 2.0 * M * D * 2 * I, False),
    "fc1 wgrad (gathered)": (lambda: ops.grouped_gemm_wgrad_gather(x, rows, dh, off, E), 2.0 * M * D * 2 * I, False),
    "fc2 wgrad": (lambda: ops.grouped_gemm_wgrad(act, dy, off, E), 2.0 * M * I * D, False),
}
A, B = sys.argv[1], sys.argv[2]
Bglu = sys.argv[3] if len(sys.argv) > 3 else B


def setord(v):
    if v == "default":
        os.environ.pop("ARIA_GEMM_ORDER", None)
    else:
        os.environ["ARIA_GEMM_ORDER"] = v


res = {"orders": {"A": A, "B": B, "B_fused_fc1": Bglu}}
for name, (fn, flops, glu) in launches.items():
    r = {"A_us": [], "B_us": []}
    outs = {}
    for rep in range(4):
        for arm, val in (("A", A), ("B", Bglu if glu else B)):
            setord(val)
            for _ in range(2):
                o = fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                o = fn()
            b.record()
            torch.cuda.synchronize()
            r[arm + "_us"].append(round(a.elapsed_time(b) * 100.0, 1))
            outs[arm] = o.clone()
    r["bit_identical"] = bool(torch.equal(outs["A"], outs["B"]))
    r["A_TFs"], r["B_TFs"] = round(flops / sorted(r["A_us"])[1] / 1e6, 1), round(flops / sorted(r["B_us"])[1] / 1e6, 1)
    res[name] = r
setord("default")
print(json.dumps(res))
