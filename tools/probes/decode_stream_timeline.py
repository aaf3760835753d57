"""Where a streamed decode token's time goes (a -DARIA_STREAM_ABL=4 build of the library stamps, per layer and stage, when the first workgroup
of the stage became resident and when its last one was done): per-stage completion times along the dependency chain of a few layers, in
microseconds from the token's first workgroup."""
import json
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from aria_amd import gptfast as G  # noqa: E402
from aria_amd import hip  # noqa: E402

for a in sys.argv[1:]:
    if a.startswith("--lib="):
        hip.LIB_PATH = os.path.abspath(a.split("=", 1)[1])
os.environ["ARIA_DECODE_STREAM"] = "1"
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
with torch.no_grad():
    m(ids, torch.arange(280, device=dev))
    m.use_decode_engine, m.decode_graph, m._engine = True, False, None
    pos = torch.tensor([280], device=dev, dtype=torch.int32)
    for i in range(12):
        m(torch.tensor([[17 + i]], device=dev), pos + i)
    torch.cuda.synchronize()
eng = m._engine
L, H = int(eng.dims[0]), int(eng.dims[2])
off = int(eng._lib.cdll.aria_decode_stream_sync_offset(eng._dims_p))
tsw = (4 + L * H + 1) & ~1
allw = eng.scratch[off + 4 * tsw: off + 4 * (tsw + 64 * (L + 3))].view(torch.int64).cpu()
raw = allw[:(L + 1) * 32].view(L + 1, 8, 4)
pacc = allw[(L + 1) * 32:(L + 1) * 32 + 64].view(8, 8)
MASK = (1 << 64) - 1


def first(v):
    v = int(v) & MASK
    return None if v == 0 else (~v) & MASK


names = ["qkv", "attention", "wo", "router+shared up", "routed up", "down+combine"]
t0 = first(raw[0, 0, 0])
out = {"lib": os.path.relpath(hip.LIB_PATH), "streamed": bool(eng.streamed()), "error_word": eng.stream_status(), "unit": "us from the first workgroup",
       "layers": {}}
for layer in (0, 1, 2, 13, 27):
    row = {}
    for s, nme in enumerate(names):
        res, done = first(raw[layer, s, 0]), int(raw[layer, s, 2]) & MASK
        row[nme] = {"first_resident": None if res is None else round((res - t0) / 100.0, 2), "last_done": round((done - t0) / 100.0, 2)}
    out["layers"][str(layer)] = row
lm = first(raw[L, 0, 0])
out["lm_head_first_resident"] = None if lm is None else round((lm - t0) / 100.0, 2)
out["per_layer_us"] = round((int(raw[L - 1, 5, 2]) - int(raw[0, 5, 2])) / 100.0 / (L - 1), 2)
PH = {0: ("qkv", ["start -> rows requested, input complete", "vector in LDS", "rows landed, dots done", "outputs out, counted"]),
      4: ("routed up", ["start -> vector ready, router complete", "routed, rows requested", "rows landed, dots done", "outputs out, counted"]),
      5: ("down+combine", ["start -> router complete", "up-projection complete, images requested", "first rows + images landed",
                           "second rows landed, dots done", "outputs out, counted"])}
out["mean_workgroup_phases_us"] = {}
for st, (nme, labels) in PH.items():
    n = max(1, int(pacc[st, 0]))
    out["mean_workgroup_phases_us"][nme] = {"workgroups": int(pacc[st, 0]), **{lab: round(int(pacc[st, 1 + i]) / n / 100.0, 2) for i, lab in enumerate(labels)}}
print(json.dumps(out, indent=1))
