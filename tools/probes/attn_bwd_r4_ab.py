"""Attention backward of this tree against the library built from the previous commit's attn.hip (build/abl/libaria_attn_old.so), same box,
interleaved: ms per aria_attn_bwd call (delta + dK/dV + dQ) and bit-identity of dq / dk / dv.  One JSON line."""
import json
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from aria_amd import hip  # noqa: E402

bf16, dev = torch.bfloat16, "cuda"
libs = {"new": hip.HipLibrary(hip.LIB_PATH), "old": hip.HipLibrary("build/abl/libaria_attn_old.so")}
g = torch.Generator(device=dev).manual_seed(0)
res = {}
for name, (B, S, H, hd, iters) in {"llm_8x2048_h20_d128": (8, 2048, 20, 128, 10), "llm_1x16384_h20_d128": (1, 16384, 20, 128, 4),
                                    "llm_1x65536_h20_d128": (1, 65536, 20, 128, 2)}.items():
    D = H * hd
    qkv = torch.randn((B * S, 3 * D), generator=g, device=dev).to(bf16)
    do = torch.randn((B * S, D), generator=g, device=dev).to(bf16)
    o = torch.empty((B * S, D), dtype=bf16, device=dev)
    lse = torch.empty((B, H, S), dtype=torch.float32, device=dev)
    delta = torch.empty_like(lse)
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    stream = torch.cuda.current_stream().cuda_stream
    libs["new"].call("aria_attn_fwd", q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), None, None, B, S, S, H, hd, 3 * D, 3 * D, 3 * D, D,
                     float(hd ** -0.5), 1, stream)
    outs, times = {}, {"new": [], "old": []}
    for rep in range(2):
        for tag, lib in libs.items():
            dq, dk, dv = (torch.empty((B * S, D), dtype=bf16, device=dev) for _ in range(3))

            def f():
                lib.call("aria_attn_bwd", q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(), delta.data_ptr(),
                         dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), None, None, B, S, S, H, hd, 3 * D, 3 * D, 3 * D, D, D, D, D, float(hd ** -0.5), 1, stream)
            f()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(iters):
                f()
            e.record()
            torch.cuda.synchronize()
            times[tag].append(round(s.elapsed_time(e) / iters, 4))
            outs[tag] = (dq, dk, dv)
    res[name] = {"new_ms": times["new"], "old_ms": times["old"], "bit_identical": all(bool(torch.equal(a, b)) for a, b in zip(outs["new"], outs["old"]))}
    del qkv, do, o, outs
print(json.dumps(res))
