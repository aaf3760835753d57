"""Wall-clock marks inside gemm3_kernel (library built with -DARIA_ABL=512) for the fused fc1 + SwiGLU launch and the plain fc1 launch at the
benchmark's shape (98 304 routed rows, 64 experts, K 2560, N 3328): where a tile's time goes -- entry -> first operands, K loop, epilogue
(pack / activation / park / store issue), store acknowledgement.  One JSON line, medians and p10 / p90 in microseconds."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

root = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, root)
from aria_amd import hip, ops  # noqa: E402

lib = hip.HipLibrary(os.path.join(root, "build", "abl", "libgemm_abl512.so"))
lib.cdll.aria_abl_ts.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev, bf16 = "cuda", torch.bfloat16
T, D, I, E, k = 16384, 2560, 1664, 64, 6
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn((T, D), generator=g, device=dev).to(bf16)
logits = torch.randn((T, E), generator=g, device=dev).to(bf16)
scores, idx, counts = ops.moe_route(logits, k)
off, sorted_src, inv = ops.moe_sort(idx, counts)
perm = ops.moe_permute(x, sorted_src, k)
w = (torch.randn((E, D, 2 * I), generator=g, device=dev) * 0.02).to(bf16)
M = perm.shape[0]
h = torch.empty((M, 2 * I), dtype=bf16, device=dev)
act = torch.empty((M, I), dtype=bf16, device=dev)
st = torch.cuda.current_stream().cuda_stream
res = {}


def timeline(name, launch):
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    ts = np.zeros(4096 * 8, dtype=np.uint64)
    lib.cdll.aria_abl_ts(ts.ctypes.data, 4096 * 8)
    full = ts.reshape(4096, 8).astype(np.int64)
    full = full[(full[:, 0] > 0) & (full[:, 5] > full[:, 0])]          # workgroups that ran a tile (ids < 4096)
    rel = (full[:, :6] - full[:, 0].min()) * 0.01
    seg = {"entry->first operands": rel[:, 1] - rel[:, 0], "K loop": rel[:, 2] - rel[:, 1], "epilogue (K loop end -> stores issued)": rel[:, 4] - rel[:, 2],
           "store ack": rel[:, 5] - rel[:, 4], "tile total": rel[:, 5] - rel[:, 0]}
    clk = (full[:, 7] - full[:, 6]) / np.maximum((full[:, 2] - full[:, 1]) * 10.0, 1)
    r = {k2: [round(float(np.percentile(v, q)), 2) for q in (10, 50, 90)] for k2, v in seg.items()}
    r["workgroups sampled"] = int(full.shape[0])
    r["shader clock GHz inside the K loop p50"] = round(float(np.percentile(clk, 50)), 3)
    res[name] = r


fused = lambda: lib.call("aria_grouped_gemm_swiglu_bf16", perm.data_ptr(), w.data_ptr(), h.data_ptr(), act.data_ptr(), off.data_ptr(),
                         E, M, 2 * I, D, D, 2 * I, D * 2 * I, 2 * I, I, st)
fused_act = lambda: lib.call("aria_grouped_gemm_swiglu_bf16", perm.data_ptr(), w.data_ptr(), None, act.data_ptr(), off.data_ptr(),
                             E, M, 2 * I, D, D, 2 * I, D * 2 * I, 2 * I, I, st)
plain = lambda: lib.call("aria_grouped_gemm_bf16", perm.data_ptr(), w.data_ptr(), h.data_ptr(), off.data_ptr(), E, M, 2 * I, D, 1, D, 2 * I,
                         D * 2 * I, 2 * I, st)
os.environ["ARIA_GEMM_FORCE"] = "3"
# interleaved, plain first this time (the clock inside the K loop differed between the launches of the first run: order effect or the launch?)
for rep in range(2):
    timeline(f"plain fc1 #{rep}", plain)
    timeline(f"fused fc1 + SwiGLU (h kept) #{rep}", fused)
    timeline(f"fused fc1 + SwiGLU (act only) #{rep}", fused_act)
os.environ["ARIA_GEMM_ORDER"] = "4"
timeline("fused (h kept), expert-major order", fused)
os.environ["ARIA_GEMM_ORDER"] = "516"
timeline("plain fc1, ragged-last order", plain)
print(json.dumps(res))
