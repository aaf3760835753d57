"""r06: do two INDEPENDENT grouped launches of the MoE backward finish sooner when they are enqueued on two streams (the second one's
workgroups fill the first one's partly idle last round: the r06 timeline puts 55-110 us per CU of idle behind a launch's last tile) than
back to back on one?  Pairs: {fc1 dgrad, fc1 wgrad (gathered)} and {fc2 dgrad + dSwiGLU, fc2 wgrad}, at the benchmark shape, product
library, interleaved repetitions.  One JSON line."""
import json
import os
import sys

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
main = torch.cuda.current_stream()
side = torch.cuda.Stream()
pairs = {
    "fc1 dgrad | fc1 wgrad (gathered)": (lambda: ops.grouped_gemm(dh, w1, off, w_is_kn=False), lambda: ops.grouped_gemm_wgrad_gather(x, rows, dh, off, E)),
    "fc2 dgrad + dSwiGLU | fc2 wgrad": (lambda: ops.grouped_gemm_dswiglu(dy, w2, off, h), lambda: ops.grouped_gemm_wgrad(act, dy, off, E)),
}
res = {}


def timed(fn, n=10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000.0 / n


for name, (f1, f2) in pairs.items():
    def serial():
        f1()
        f2()

    def overlapped(first_on_side=False):
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            (f1 if first_on_side else f2)()
        (f2 if first_on_side else f1)()
        main.wait_stream(side)

    for _ in range(3):
        serial(), overlapped()
    r = {"serial_us": [], "two_streams_us": [], "two_streams_first_on_side_us": []}
    for rep in range(4):
        r["serial_us"].append(round(timed(serial), 1))
        r["two_streams_us"].append(round(timed(overlapped), 1))
        r["two_streams_first_on_side_us"].append(round(timed(lambda: overlapped(True)), 1))
    r["alone_us"] = [round(timed(f1), 1), round(timed(f2), 1)]
    res[name] = r
print(json.dumps(res))
