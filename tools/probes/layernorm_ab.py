"""LayerNorm forward at the ViT's shape (78 400 rows x 1152): the one-row-at-a-time kernel (ARIA_LAYERNORM_V1=1) against two rows in flight per wave
(default), interleaved in one process; us per call and GB/s of x read + y written."""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aria_amd import ops  # noqa: E402

T, D = 78400, 1152
x = torch.randn(T, D, device="cuda").to(torch.bfloat16)
w, b = torch.randn(D, device="cuda").to(torch.bfloat16), torch.randn(D, device="cuda").to(torch.bfloat16)


def timed(n=40):
    ops.layernorm(x, w, b, 1e-6, want_stats=False)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        ops.layernorm(x, w, b, 1e-6, want_stats=False)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


t = {"v1": [], "v2": []}
for _ in range(5):
    os.environ["ARIA_LAYERNORM_V1"] = "1"
    t["v1"].append(timed())
    os.environ.pop("ARIA_LAYERNORM_V1")
    t["v2"].append(timed())
res = {k: round(statistics.median(v), 1) for k, v in t.items()}
res.update({k + "_GBps": round(2 * T * D * 2 / statistics.median(v) / 1e3, 0) for k, v in t.items()})
print(json.dumps(res))
