import sys, torch
sys.path.insert(0, ".")
from aria_amd import gptfast as G
from aria_amd.vision import AriaVisionConfig
bf16 = torch.bfloat16; dev = torch.device("cuda")
torch.set_default_device(dev)
model = G.Aria(G.ModelArgs(n_layer=2), AriaVisionConfig(num_hidden_layers=1))
torch.set_default_device("cpu")
with torch.no_grad():
    for p in model.parameters(): p.normal_(0, 0.02)
model.eval()
ids = torch.randint(10, 100000, (1, 280), device=dev); ids[:, 8:264] = 9
pv = torch.randn((1, 3, 980, 980), device=dev).clamp_(-1, 1).to(bf16)
pm = torch.ones((1, 980, 980), dtype=torch.bool, device=dev)
model.setup_caches(1, 300)
with torch.no_grad():
    emb = model.prepare_embeddings(ids, pv, pm); torch.cuda.synchronize(); print("emb ok", flush=True)
    logits = model(None, torch.arange(280, device=dev), emb, last_only=True); torch.cuda.synchronize(); print("prefill ok", flush=True)
    nxt, _ = G.sample(logits, 0.8, 200); torch.cuda.synchronize(); print("sample ok", flush=True)
    dec = G.DecodeGraph(model, 0.8, 200); torch.cuda.synchronize(); print("capture ok", flush=True)
    pos = torch.tensor([280], device=dev, dtype=torch.int32)
    tok = nxt.view(1)
    for i in range(6):
        tok = dec(tok.long(), pos); pos += 1
        torch.cuda.synchronize(); print("decode", i, int(tok), flush=True)
