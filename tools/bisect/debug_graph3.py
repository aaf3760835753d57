import sys, torch
sys.path.insert(0, ".")
from aria_amd import gptfast as G
from aria_amd.vision import AriaVisionConfig
bf16 = torch.bfloat16; dev = torch.device("cuda")
mode = sys.argv[1]
torch.set_default_device(dev)
model = G.Aria(G.ModelArgs(n_layer=2), AriaVisionConfig(num_hidden_layers=1))
torch.set_default_device("cpu")
with torch.no_grad():
    for p in model.parameters(): p.normal_(0, 0.02)
model.eval()
model.setup_caches(1, 300)
with torch.no_grad():
    ids = torch.randint(10, 100000, (1, 280), device=dev)
    logits = model(ids, torch.arange(280, device=dev), last_only=True); torch.cuda.synchronize()
    dec = G.DecodeGraph(model, 0.8, 200, use_graph=(mode != "eager")); torch.cuda.synchronize(); print("capture ok", flush=True)
    pos = torch.tensor([280], device=dev, dtype=torch.int32)
    tok = torch.tensor([11], device=dev)
    for i in range(6):
        out = dec(tok.long(), pos)
        if mode in ("pos", "both", "eager"): pos += 1
        if mode in ("tok", "both", "eager"): tok = out
        torch.cuda.synchronize(); print(mode, "decode", i, int(out), flush=True)
