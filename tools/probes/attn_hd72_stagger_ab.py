"""ViT attention forward (hd 72) at the config #3 step's shape (16 images x 4900 patches x 16 heads) and the projector's cross shape, the three
selectable forms interleaved in ONE process (ARIA_ATTN_HD72_STAGGER is read per launch): 0 = attn_fwd2_kernel<72, 12> (one barrier per key tile),
1 = attn_fwd2s_kernel (wave groups one barrier interval apart: QK / softmax / PV of three tiles overlap on every SIMD), 2 = 1 + wave priority in the
matrix phases.  Prints per-form medians and whether the outputs are bit-equal.
Needs tools/probes/src/attn_fwd2s_stagger.patch applied (git apply; make): the staggered form measured 10 % slower (profiles/r05_attn_hd72_stagger_ab.json) and
is not in the library."""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aria_amd import ops  # noqa: E402

bf16, dev = torch.bfloat16, "cuda"
res = {}


def timed(fn, n):
    fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


for name, B, Sq, Skv, H, masked, n in (("vit_16x4900", 16, 4900, 4900, 16, False, 10), ("vit_16x4900_masked", 16, 4900, 4900, 16, True, 10),
                                       ("projector_16x256x4900", 16, 256, 4900, 16, False, 20), ("vit_4x1225", 4, 1225, 1225, 16, False, 20)):
    hd = 72
    D = H * hd
    q = torch.randn(B * Sq, D, device=dev).to(bf16)
    kv = torch.randn(B * Skv, 2 * D, device=dev).to(bf16)
    km = None
    if masked:
        km = torch.ones(B, Skv, dtype=torch.uint8, device=dev)
        km[::2, 4000:] = 0     # every other image padded: 14 whole key tiles + a partial one masked
    outs, times = {}, {m: [] for m in "012"}
    for rnd in range(4):
        for m in "012":
            os.environ["ARIA_ATTN_HD72_STAGGER"] = m
            f = lambda: ops.attention_fwd(q, kv[:, :D], kv[:, D:], B, Sq, H, hd, hd ** -0.5, False, key_mask=km, Skv=Skv)
            times[m].append(timed(f, n))
            if rnd == 0:
                o, lse = f()
                outs[m] = (o.clone(), lse.clone())
    flops = 4.0 * B * H * Sq * Skv * hd
    res[name] = {f"mode{m}_ms": round(statistics.median(times[m]), 4) for m in "012"}
    res[name].update({f"mode{m}_tflops": round(flops / statistics.median(times[m]) / 1e9, 1) for m in "012"})
    res[name]["bit_equal"] = all(torch.equal(outs["0"][0], outs[m][0]) and torch.equal(outs["0"][1], outs[m][1]) for m in "12")
    res[name]["runs_ms"] = {m: [round(x, 4) for x in times[m]] for m in "012"}
os.environ.pop("ARIA_ATTN_HD72_STAGGER", None)
best = min("012", key=lambda m: res["vit_16x4900"][f"mode{m}_ms"])
res["best_mode_vit"] = best
print(json.dumps(res))
