"""attn_fwd2 of this tree (K fragments of the next k-step requested before the current MFMAs) against the library built from the previous
commit's attn.hip (build/abl/libaria_attn_old.so), same box, interleaved, ARIA_ATTN_FWD=2 for both (so hd 128 also runs v2): ms per launch
and bit-identity.  One JSON line."""
import json
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from aria_amd import hip  # noqa: E402

os.environ["ARIA_ATTN_FWD"] = "2"
bf16, dev = torch.bfloat16, "cuda"
libs = {"new": hip.HipLibrary(hip.LIB_PATH), "old": hip.HipLibrary("build/abl/libaria_attn_old.so")}
g = torch.Generator(device=dev).manual_seed(0)
res = {}
for name, (B, Sq, Skv, H, hd, causal, masked, iters) in {"vit_16x4900_h16_d72_masked": (16, 4900, 4900, 16, 72, False, True, 8),
                                                         "projector_16x256q_4900k_h16_d72": (16, 256, 4900, 16, 72, False, True, 20),
                                                         "llm_8x2048_h20_d128_causal_v2": (8, 2048, 2048, 20, 128, True, False, 20),
                                                         "llm_1x65536_h20_d128_causal_v2": (1, 65536, 65536, 20, 128, True, False, 3)}.items():
    D = H * hd
    q = torch.randn((B * Sq, D), generator=g, device=dev).to(bf16)
    kv = torch.randn((B * Skv, 2 * D), generator=g, device=dev).to(bf16)
    km = None
    if masked:
        km = torch.ones(B, Skv, dtype=torch.uint8, device=dev)
        km[0, Skv * 3 // 4:] = 0
    lse = torch.empty((B, H, Sq), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    k, v = kv[:, :D], kv[:, D:]
    outs, times = {}, {"new": [], "old": []}
    for rep in range(3):
        for tag, lib in libs.items():
            o = torch.empty((B * Sq, D), dtype=bf16, device=dev)

            def f():
                lib.call("aria_attn_fwd", q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), None,
                         km.data_ptr() if km is not None else None, B, Sq, Skv, H, hd, D, 2 * D, 2 * D, D, float(hd ** -0.5), int(causal), stream)
            f()
            f()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(iters):
                f()
            e.record()
            torch.cuda.synchronize()
            times[tag].append(round(s.elapsed_time(e) / iters, 4))
            outs[tag] = o
    res[name] = {"new_ms": times["new"], "old_ms": times["old"], "bit_identical": bool(torch.equal(outs["new"], outs["old"]))}
    del q, kv, outs
print(json.dumps(res))
