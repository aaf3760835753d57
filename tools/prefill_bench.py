"""Config #4 (BASELINE.json): long-context prefill on 1 x MI355X -- 32 video frames (490px -> 128 tokens each) + text, S = 53 248
(and 65 536 with --seq), Aria-25.3B random-init bf16 through the gptfast surface with a bf16 KV cache (286 720 B/token).
No S x S mask exists anywhere (the reference's default mask would be 4.3 GB at 64K, gptfast/model.py:139-141)."""
import argparse, json, sys, time
import torch
sys.path.insert(0, ".")
from aria_amd import gptfast as G
from aria_amd.vision import AriaVisionConfig

ap = argparse.ArgumentParser()
ap.add_argument("--seq", type=int, default=53248)
ap.add_argument("--frames", type=int, default=32)
ap.add_argument("--layers", type=int, default=28)
ap.add_argument("--vit-layers", type=int, default=27)
ap.add_argument("--runs", type=int, default=2)
a = ap.parse_args()
dev = torch.device("cuda"); bf16 = torch.bfloat16
torch.set_default_device(dev)
model = G.Aria(G.ModelArgs(n_layer=a.layers, block_size=a.seq), AriaVisionConfig(num_hidden_layers=a.vit_layers))
torch.set_default_device("cpu")
g = torch.Generator(device="cuda").manual_seed(0)
with torch.no_grad():
    for n, p in model.named_parameters():
        if "norm" in n and n.endswith("weight") or "ln_" in n and n.endswith("weight"): p.fill_(1.0)
        elif n.endswith("bias"): p.zero_()
        else:
            flat = p.view(-1)
            for o in range(0, flat.numel(), 1 << 28): flat[o:o + (1 << 28)].normal_(0.0, 0.02, generator=g)
model.eval(); model.setup_caches(1, a.seq)
S = a.seq
ids = torch.randint(10, 100000, (1, S), generator=g, device=dev)
ids[:, 16:16 + 128 * a.frames] = 9
pv = torch.randn((a.frames, 3, 490, 490), generator=g, device=dev).clamp_(-1, 1).to(bf16)
pm = torch.ones((a.frames, 490, 490), dtype=torch.bool, device=dev)
ts = []
with torch.no_grad():
    for i in range(a.runs + 1):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        emb = model.prepare_embeddings(ids, pv, pm)
        lg = model(None, torch.arange(S, device=dev), emb, last_only=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if i: ts.append(dt)
t = sum(ts) / len(ts)
flops = S * 7.716e9 + 28 * 4 * 2560 * (S / 2) * S + a.frames * 1.19e12
print(json.dumps({"metric": "prefill tok/s (config #4)", "value": round(S / t, 1), "unit": "tokens/s", "seq_len": S, "frames": a.frames,
                  "seconds": round(t, 3), "algorithmic_tflops": round(flops / 1e12, 1), "achieved_tflops_s": round(flops / t / 1e12, 1),
                  "frac_of_bf16_mfma_peak": round(flops / t / 2.5e15, 4), "kv_cache_GB": round(28 * 2 * S * 2560 * 2 / 1e9, 1),
                  "max_mem_GB": round(torch.cuda.max_memory_allocated() / 1e9, 1), "layers": a.layers}))
