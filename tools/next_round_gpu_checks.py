"""Hardware checks for code that so far only ran under the emulator (written when the round's GPU budget was spent).  Run this FIRST in the
next GPU session; flip defaults only for what passes and wins:

    gpurun --timeout 900 -- 'python tools/next_round_gpu_checks.py > gpurun_out/next_round_checks.json'

1. parity: tests/kernel_cases.case_decode_attention (aria_decode_attn, split-KV flash-decoding form) on the device;
2. timing: aria_decode_attn alone, one workgroup per head vs heads x splits, at cache fills 1 K .. 64 K (Aria head shape 20 x 128);
3. timing: full decode step (random-init Aria-25.3B LLM) at long contexts with ARIA_DECODE_SPLIT_KV unset / 1.

Separately (each is one bench line, ~2.5 min):  python bench.py --steps 3   vs   ARIA_LMHEAD_SKIP_MASKED=1 python bench.py --steps 3
(lm_head GEMMs + CE over the labelled 25 % of the positions only; expected ~-15 ms per step, loss identical).
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from aria_amd import gptfast as G  # noqa: E402
from aria_amd import ops  # noqa: E402
from tests import kernel_cases as C  # noqa: E402

bf16 = torch.bfloat16
DRY = "--emu-dry-run" in sys.argv          # exercise this script's own code paths on CPU through the emulator, tiny shapes (no numbers)
if DRY:
    from tests.emu import emu_lib

    emu_lib.install()
dev = torch.device("cpu" if DRY else "cuda")
res = {"parity": {}, "attn_us": {}, "decode_ms": {}}

for H, hd, pos, splits in ([(2, 128, 63, 2)] if DRY else [(2, 128, 0, 4), (2, 128, 63, 2), (3, 128, 64, 2), (20, 128, 2999, 16),
                                                            (20, 128, 20000, 32), (3, 64, 127, 2), (2, 64, 1000, 5)]):
    key = f"H{H}_hd{hd}_pos{pos}_s{splits}"
    try:
        C.case_decode_attention(dev, H, hd, pos, splits)
        res["parity"][key] = "ok"
    except Exception as ex:  # keep going: the report is the point
        res["parity"][key] = f"FAILED {type(ex).__name__}: {ex}"

H, hd = (2, 128) if DRY else (20, 128)
D = H * hd


def timed(fn, n):
    if DRY:
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        return (time.perf_counter() - t0) / n * 1e3
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


for fill in ((96,) if DRY else (1024, 4096, 16384, 65536)):
    S_max = fill + 8
    g = torch.Generator(device=dev).manual_seed(0)
    kc = torch.randn((S_max, D), generator=g, device=dev).to(bf16)
    vc = torch.randn((S_max, D), generator=g, device=dev).to(bf16)
    qkv = torch.randn((3 * D,), generator=g, device=dev).to(bf16)
    fc = torch.rand((S_max, hd // 2, 2), generator=g, device=dev).to(bf16)
    pos = torch.tensor([fill - 1], dtype=torch.int32, device=dev)
    for splits in ((1, 4) if DRY else (1, 4, 8, 16, 32)):
        for _ in range(1 if DRY else 3):
            ops.decode_attention(qkv, fc, pos, kc, vc, H, hd, splits=splits)
        ms = timed(lambda: ops.decode_attention(qkv, fc, pos, kc, vc, H, hd, splits=splits), 1 if DRY else 20)
        res["attn_us"][f"fill{fill}_splits{splits}"] = round(ms * 1e3, 1)
    del kc, vc

torch.set_default_device(dev)
m = G.Transformer(G.ModelArgs(block_size=64, vocab_size=136, n_layer=1, n_head=2, dim=256, intermediate_size=40, n_local_heads=2, head_dim=128,
                              num_experts=8, router_topk=3) if DRY else G.ModelArgs())
torch.set_default_device("cpu")
g = torch.Generator(device=dev).manual_seed(0)
with torch.no_grad():
    for n, p in m.named_parameters():
        if "norm" in n:
            p.fill_(1.0)
        else:
            flat = p.view(-1)
            for o in range(0, flat.numel(), 1 << 28):
                flat[o:o + (1 << 28)].normal_(0.0, 0.02, generator=g)
m.eval()
tok = torch.tensor([[17]], device=dev)
for S_max in ((2100,) if DRY else (4096, 16384, 32768)):
    m.setup_caches(1, S_max)
    for layer in m.layers:  # a filled cache (values irrelevant for timing, finite for the softmax)
        layer.attention.kv_cache.k.normal_(0, 1, generator=g)
        layer.attention.kv_cache.v.normal_(0, 1, generator=g)
    for mode in ("", "1"):
        if mode:
            os.environ["ARIA_DECODE_SPLIT_KV"] = mode
        else:
            os.environ.pop("ARIA_DECODE_SPLIT_KV", None)
        m._engine = None
        with torch.no_grad():
            pos = torch.tensor([S_max - 64], device=dev, dtype=torch.int32)
            reps = 1 if DRY else 30
            for _ in range(1 if DRY else 3):
                m(tok, pos)
            if not DRY:
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(reps):
                m(tok, pos + i)
            if not DRY:
                torch.cuda.synchronize()
        res["decode_ms"][f"Smax{S_max}_split{mode or 0}"] = round((time.perf_counter() - t0) / reps * 1e3, 3)
print(json.dumps(res))
