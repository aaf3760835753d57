import sys, torch
sys.path.insert(0, ".")
from aria_amd import gptfast as G
from aria_amd.vision import AriaVisionConfig
bf16 = torch.bfloat16; dev = torch.device("cuda")
torch.set_default_device(dev)
model = G.Aria(G.ModelArgs(n_layer=2), AriaVisionConfig(num_hidden_layers=1))
torch.set_default_device("cpu")
with torch.no_grad():
    for n, p in model.named_parameters():
        if 'norm' in n and n.endswith('weight') and sys.argv[1] == 'real': p.fill_(1.0)
        else: p.normal_(0, 0.02)
model.eval(); model.setup_caches(1, 296)
ids = torch.randint(10, 100000, (1, 280), device=dev); ids[:, 8:264] = 9
pv = torch.randn((1, 3, 980, 980), device=dev).clamp_(-1, 1).to(bf16)
pm = torch.ones((1, 980, 980), dtype=torch.bool, device=dev)
S = lambda m: (torch.cuda.synchronize(), print(m, flush=True))
with torch.no_grad():
    dec = None
    for rep in range(2):
        emb = model.prepare_embeddings(ids, pv, pm); S(f"{rep} emb")
        lg = model(None, torch.arange(280, device=dev), emb, last_only=True); S(f"{rep} prefill")
        nxt = G.sample(lg, 0.8, 200)[0]; S(f"{rep} sample")
        if dec is None:
            dec = G.DecodeGraph(model, 0.8, 200); S("capture")
        pos = torch.tensor([280], device=dev, dtype=torch.int32); tok = nxt.view(1)
        for i in range(15):
            tok = dec(tok.long(), pos); pos += 1
            S(f"{rep} decode {i} pos {int(pos)}")
