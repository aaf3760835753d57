"""Where does a key tile's time go in attn_fwd2_kernel?  Times aria_attn_fwd of every build/abl/libaria_attn_<bits>.so (tools/probes/build_attn_abl.sh;
bits documented in attn.hip: 1 no QK MFMAs, 2 no exponentials, 4 no PV MFMAs, 8 no V reads, 16 no K reads, 32 no staging of the next tile,
64 no barrier, 128 no running maximum) at the ViT shape (16 x 4900, 16 x 72, one padded image) and the decoder's (8 x 2048 and 1 x 65 536, 20 x 128,
causal).  One JSON line: ms per launch per variant."""
import glob
import json
import os
import re
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from aria_amd import hip  # noqa: E402

bf16, dev = torch.bfloat16, "cuda"
libs = {int(re.search(r"attn_(\d+)\.so", p).group(1)): hip.HipLibrary(p) for p in sorted(glob.glob("build/abl/libaria_attn_*.so"))}
g = torch.Generator(device=dev).manual_seed(0)
SHAPES = {"vit_16x4900_h16_d72": (16, 4900, 16, 72, False, True, 6), "llm_8x2048_h20_d128": (8, 2048, 20, 128, True, False, 10),
          "llm_1x65536_h20_d128": (1, 65536, 20, 128, True, False, 3)}
res = {}
for name, (B, S, H, hd, causal, masked, iters) in SHAPES.items():
    D = H * hd
    qkv = (torch.randn((B * S, 3 * D), generator=g, device=dev)).to(bf16)
    o = torch.empty((B * S, D), dtype=bf16, device=dev)
    lse = torch.empty((B, H, S), dtype=torch.float32, device=dev)
    km = None
    if masked:
        km = torch.ones(B, S, dtype=torch.uint8, device=dev)
        km[0, S * 3 // 4:] = 0
    stream = torch.cuda.current_stream().cuda_stream
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    out = {}
    for rep in range(2):
        for bits, lib in libs.items():
            def f():
                lib.call("aria_attn_fwd", q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), None,
                         km.data_ptr() if km is not None else None, B, S, S, H, hd, 3 * D, 3 * D, 3 * D, D, float(hd ** -0.5), int(causal), stream)
            for _ in range(2):
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
    del qkv, o, lse
print(json.dumps(res))
