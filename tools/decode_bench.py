"""decode-step timing on the gptfast Transformer (random-init Aria-25.3B LLM, batch 1): the engine's 6-launch schedule vs the 7-launch one
(ARIA_DECODE_FUSE), graph replay vs plain enqueue, and the tile-GEMM path; model step only (no sampling)."""
import json, os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from aria_amd import gptfast as G
bf16 = torch.bfloat16
dev = torch.device("cuda")
torch.set_default_device(dev)
m = G.Transformer(G.ModelArgs())
torch.set_default_device("cpu")
g = torch.Generator(device="cuda").manual_seed(0)
with torch.no_grad():
    for n, p in m.named_parameters():
        if "norm" in n:
            p.fill_(1.0)
        else:
            flat = p.view(-1)
            for o in range(0, flat.numel(), 1 << 28):
                flat[o:o + (1 << 28)].normal_(0.0, 0.02, generator=g)
m.eval()
m.setup_caches(1, 512)
ids = torch.randint(10, 100000, (1, 280), generator=g, device=dev)
only = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--only=")]
res = {}
with torch.no_grad():
    m(ids, torch.arange(280, device=dev))
    tok = torch.tensor([[17]], device=dev)
    for name, eng, graph, fuse in (("engine_fused6", True, False, "1"), ("engine_unfused7", True, False, "0"), ("engine_fused6_graph", True, True, "1"),
                                   ("engine_fused6_again", True, False, "1"), ("tile_path", False, False, "1")):
        if only and name not in only:
            continue
        os.environ["ARIA_DECODE_FUSE"] = fuse  # read by the library at every aria_decode_token call (csrc/decode.hip)
        m.use_decode_engine, m.decode_graph, m._engine = eng, graph, None
        pos = torch.tensor([280], device=dev, dtype=torch.int32)
        for _ in range(3):
            m(tok, pos)
        torch.cuda.synchronize()
        n = 50
        t0 = time.perf_counter()
        for i in range(n):
            m(tok, pos + i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        res[name] = dict(ms_per_token=round(dt * 1e3, 3), tok_s=round(1 / dt, 1), graph=bool(m._engine is not None and m._engine.graph))
print(json.dumps(res))
