"""Config #2 (BASELINE.json): Aria-25.3B bf16 single-image generate() on the gptfast surface, 1 x MI355X.
Protocol of gptfast/benchmark.py:10-48: one 980px image (256 image tokens) + a short prompt, max_new_tokens 200, top-k 200,
temperature 0.8, 2 warm-up + 5 timed runs, tok/s = mean(#new tokens) / mean(latency) (whole generate incl. ViT + prefill).
Random-init weights (no checkpoint offline), synthetic image.  Also reports prefill and decode rates separately and the
decode HBM roofline (7.72 GB of weights per token)."""
import argparse
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from aria_amd import gptfast as G  # noqa: E402

bf16 = torch.bfloat16


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=28)
    ap.add_argument("--vit-layers", type=int, default=27)
    ap.add_argument("--new", type=int, default=200)
    ap.add_argument("--runs", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--graph", action="store_true", help="HIP-graph decode (known unstable across prefills in round 1)")
    a = ap.parse_args()
    dev = torch.device("cuda")
    from aria_amd.vision import AriaVisionConfig

    torch.set_default_device(dev)
    model = G.Aria(G.ModelArgs(n_layer=a.layers), AriaVisionConfig(num_hidden_layers=a.vit_layers))
    torch.set_default_device("cpu")
    g = torch.Generator(device="cuda").manual_seed(0)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("norm.weight") or "layer_norm" in n and n.endswith("weight") or "ln_" in n and n.endswith("weight"):
                p.fill_(1.0)
            elif n.endswith("bias"):
                p.zero_()
            else:
                flat = p.view(-1)
                for o in range(0, flat.numel(), 1 << 28):
                    flat[o:o + (1 << 28)].normal_(0.0, 0.02, generator=g)
    model.eval()
    ids = torch.randint(10, 100000, (1, 280), generator=g, device=dev)
    ids[:, 8:264] = 9
    pv = torch.randn((1, 3, 980, 980), generator=g, device=dev).clamp_(-1, 1).to(bf16)
    pm = torch.ones((1, 980, 980), dtype=torch.bool, device=dev)
    model.setup_caches(1, 280 + a.new)
    decoder = None
    lat, ntok = [], []
    for i in range(a.warmup + a.runs):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out, decoder = G.generate(model, ids, a.new, pixel_values=pv, pixel_mask=pm, temperature=0.8, top_k=200, decoder=decoder, use_graph=a.graph)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if i >= a.warmup:
            lat.append(dt)
            ntok.append(out.numel() - ids.numel())
    # split: prefill only / decode only
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        emb = model.prepare_embeddings(ids, pv, pm)
        model(None, torch.arange(280, device=dev), emb, last_only=True)
    torch.cuda.synchronize()
    t_prefill = (time.perf_counter() - t0) / 3
    pos = torch.tensor([280], device=dev, dtype=torch.int32)
    tok = torch.tensor([[11]], device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        decoder(tok, pos)
    torch.cuda.synchronize()
    t_dec = (time.perf_counter() - t0) / 50
    # the model step alone (embedding lookup + engine call, no sampling) and the bare engine call, same box
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        model.llm(tok, pos)
    torch.cuda.synchronize()
    t_step = (time.perf_counter() - t0) / 50
    eng = model.llm._engine
    t_eng = None
    if eng is not None:
        stream = torch.cuda.current_stream().cuda_stream
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            eng._lib.call("aria_decode_token", eng.ptrs, eng._dims_p, eng.eps, stream)
        torch.cuda.synchronize()
        t_eng = (time.perf_counter() - t0) / 50
    weight_bytes = (28 * (6 * 3 * 2560 * 1664 + 3 * 2560 * 3328 + 64 * 2560 + 4 * 2560 * 2560) + 100352 * 2560) * 2
    res = {"metric": "generate tok/s (gptfast protocol, config #2)", "value": round(sum(ntok) / sum(lat), 2), "unit": "tokens/s",
           "published_h100": {"eager": 25.2, "compile": 130.0}, "new_tokens": a.new, "runs": a.runs,
           "prefill_ms_280tok_incl_vit": round(t_prefill * 1e3, 2), "prefill_tok_s": round(280 / t_prefill, 1),
           "decode_ms_per_token": round(t_dec * 1e3, 3), "decode_tok_s": round(1 / t_dec, 1),
           "model_step_ms": round(t_step * 1e3, 3), "engine_call_ms": None if t_eng is None else round(t_eng * 1e3, 3),
           "decode_roofline": {"bound": "hbm", "achieved": round(weight_bytes / t_dec / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                               "frac": round(weight_bytes / t_dec / 8e12, 4)},
           "config": {"layers": a.layers, "vit_layers": a.vit_layers, "hip_graph": bool(a.graph)}}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
