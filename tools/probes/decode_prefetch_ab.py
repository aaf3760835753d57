"""r06: decode step (random-init Aria-25.3B LLM on the gptfast surface, batch 1, 280-position context, graph replay) with the captured graph's
weight-prefetch branch off / on at several widths (ARIA_DECODE_PREFETCH = workgroups of 256 lanes, read when the graph is captured):
ms per token interleaved over the arms, and the logits of a fixed step compared BIT FOR BIT with the branch off.  One JSON line."""
import json, os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from aria_amd import gptfast as G
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
arms = [int(a) for a in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,32,64,128,256,512".split(","))]
res = {"arms_workgroups": arms, "ms_per_token": {str(a): [] for a in arms}, "logits_equal_to_off": {}}
ref = None
with torch.no_grad():
    m(ids, torch.arange(280, device=dev))
    tok = torch.tensor([[17]], device=dev)
    m.use_decode_engine, m.decode_graph = True, True
    for rep in range(3):
        for a in arms:
            os.environ["ARIA_DECODE_PREFETCH"] = str(a)
            m._engine = None   # the graph is captured again under this setting
            pos = torch.tensor([280], device=dev, dtype=torch.int32)
            for _ in range(3):
                out = m(tok, pos)
            if rep == 0:
                lg = out.float().cpu().clone()
                if a == 0:
                    ref = lg
                res["logits_equal_to_off"][str(a)] = bool(torch.equal(lg, ref))
            torch.cuda.synchronize()
            n = 60
            t0 = time.perf_counter()
            for i in range(n):
                m(tok, pos + i)
            torch.cuda.synchronize()
            res["ms_per_token"][str(a)].append(round((time.perf_counter() - t0) / n * 1e3, 4))
            assert m._engine is not None and m._engine.graph
res["median_ms"] = {k: sorted(v)[len(v) // 2] for k, v in res["ms_per_token"].items()}
print(json.dumps(res))
