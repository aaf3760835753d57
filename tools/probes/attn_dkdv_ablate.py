"""Where does a query tile's time go in attn_bwd3_dkdv_kernel?  Times aria_attn_bwd of every build/abl/libaria_dkdv_<bits>.so
(tools/probes/build_attn_abl.sh ARIA_DKDV_ABL dkdv ...; bits in attn.hip: 1 no first GEMM, 2 no exponentials, 4 no second GEMM, 8 no P exchange,
16 no mid-tile barrier, 32 no staging, 64 no tile barrier, 128 no dS arithmetic) at 1 x 65 536 and 8 x 2048, 20 x 128, causal; the dQ kernel and
the delta kernel run unchanged in every variant, so differences between variants are the dK/dV kernel's.  One JSON line: ms per call."""
import glob
import json
import os
import re
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from aria_amd import hip  # noqa: E402

bf16, dev = torch.bfloat16, "cuda"
libs = {int(re.search(r"dkdv_(\d+)\.so", p).group(1)): hip.HipLibrary(p) for p in sorted(glob.glob("build/abl/libaria_dkdv_*.so"))}
g = torch.Generator(device=dev).manual_seed(0)
res = {}
for name, (B, S, H, hd, iters) in {"llm_1x65536_h20_d128": (1, 65536, 20, 128, 2), "llm_8x2048_h20_d128": (8, 2048, 20, 128, 10)}.items():
    D = H * hd
    qkv = torch.randn((B * S, 3 * D), generator=g, device=dev).to(bf16)
    do = torch.randn((B * S, D), generator=g, device=dev).to(bf16)
    o = torch.empty((B * S, D), dtype=bf16, device=dev)
    lse = torch.empty((B, H, S), dtype=torch.float32, device=dev)
    delta = torch.empty_like(lse)
    dq, dk, dv = (torch.empty((B * S, D), dtype=bf16, device=dev) for _ in range(3))
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    stream = torch.cuda.current_stream().cuda_stream
    libs[0].call("aria_attn_fwd", q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), None, None, B, S, S, H, hd, 3 * D, 3 * D, 3 * D, D,
                 float(hd ** -0.5), 1, stream)
    out = {}
    for rep in range(2):
        for bits, lib in libs.items():
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
            out.setdefault(str(bits), []).append(round(s.elapsed_time(e) / iters, 4))
    res[name] = out
    del qkv, do, o, dq, dk, dv
print(json.dumps(res))
