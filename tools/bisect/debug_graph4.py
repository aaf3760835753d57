import sys, torch
sys.path.insert(0, ".")
from aria_amd import gptfast as G
from aria_amd.vision import AriaVisionConfig
bf16 = torch.bfloat16; dev = torch.device("cuda")
nl, with_sample, smax = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
torch.set_default_device(dev)
model = G.Aria(G.ModelArgs(n_layer=nl), AriaVisionConfig(num_hidden_layers=1))
torch.set_default_device("cpu")
with torch.no_grad():
    for p in model.parameters(): p.normal_(0, 0.02)
model.eval(); model.setup_caches(1, smax)
tok = torch.tensor([[11]], device=dev); pos = torch.tensor([5], device=dev, dtype=torch.int32)
out = torch.zeros(1, dtype=torch.int, device=dev)
def step():
    lg = model(tok, pos, last_only=True)
    if with_sample:
        out.copy_(G.sample(lg, 0.8, 200)[0].view(-1))
    return lg
with torch.no_grad():
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): step()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    for i in range(4):
        g.replay(); torch.cuda.synchronize(); print(sys.argv[1:], "replay", i, "ok", flush=True)
