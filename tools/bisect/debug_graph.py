import sys, torch
sys.path.insert(0, ".")
from aria_amd import gptfast as G, ops
from aria_amd.vision import AriaVisionConfig
bf16 = torch.bfloat16; dev = torch.device("cuda")

def graph_run(name, fn, n=3):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    torch.cuda.synchronize()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    print("OK", name, flush=True)
    return out

x = torch.randn(8, 2560, device=dev).to(bf16); w = torch.randn(2560, 2560, device=dev).to(bf16)
graph_run("gemm v1 small", lambda: ops.gemm(x, w))
xb = torch.randn(4096, 2560, device=dev).to(bf16)
graph_run("gemm v2", lambda: ops.gemm(xb, w))
lg = torch.randn(1, 64, device=dev).to(bf16)
def moe_bits():
    sc, idx, cnt = ops.moe_route(lg, 6)
    off, ss, inv = ops.moe_sort(idx, cnt)
    return ops.moe_permute(x[:1], ss, 6)
graph_run("route/sort/permute", moe_bits)
logits = torch.randn(1, 1, 100352, device=dev)
graph_run("sample", lambda: G.sample(logits, 0.8, 200)[0])
torch.set_default_device(dev)
model = G.Aria(G.ModelArgs(n_layer=1), AriaVisionConfig(num_hidden_layers=1))
torch.set_default_device("cpu")
with torch.no_grad():
    for p in model.parameters(): p.normal_(0, 0.02)
model.setup_caches(1, 64)
tok = torch.tensor([[11]], device=dev); pos = torch.tensor([5], device=dev, dtype=torch.int32)
with torch.no_grad():
    layer = model.llm.layers[0]
    x1 = torch.randn(1, 2560, device=dev).to(bf16)
    graph_run("moe ffn", lambda: layer.feed_forward(x1))
    kv_len = pos + 1
    graph_run("attention decode", lambda: layer.attention(x1, 1, 1, model.llm.freqs_cis, pos, kv_len, False))
    graph_run("full model step", lambda: model(tok, pos, last_only=True))
