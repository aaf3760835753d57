"""The streamed decode schedule (ARIA_DECODE_STREAM=1: one launch per token, csrc/decode.hip decode_stream_kernel) against the 6-launch
schedule on the random-init Aria-25.3B LLM, batch 1, 280-token prompt, model step only (no sampling): ms per token of each, interleaved
twice in one process (the shader clock ramps over the first seconds), the logits of every timed token compared bit for bit, the
streamed schedule's error word.  --graph adds the captured form of both."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from aria_amd import gptfast as G  # noqa: E402
from aria_amd import hip  # noqa: E402

for a in sys.argv[1:]:
    if a.startswith("--lib="):  # a variant build of the library (tools/probes/build_decode_variant.sh)
        hip.LIB_PATH = os.path.abspath(a.split("=", 1)[1])

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
toks = torch.randint(10, 100000, (64,), generator=g, device=dev)
res = {"lib": os.path.relpath(hip.LIB_PATH), "runs": {}}
N = 50
ref_logits = None
with torch.no_grad():
    m(ids, torch.arange(280, device=dev))
    order = [("launch6 #0", "0", False), ("streamed #0", "1", False), ("launch6 #1", "0", False), ("streamed #1", "1", False)]
    if "--graph" in sys.argv:
        order += [("launch6 graph", "0", True), ("streamed graph", "1", True)]
    for name, stream, graph in order:
        os.environ["ARIA_DECODE_STREAM"] = stream  # read by the library at every aria_decode_token call
        m.use_decode_engine, m.decode_graph, m._engine = True, graph, None
        pos = torch.tensor([280], device=dev, dtype=torch.int32)
        for i in range(3):
            m(toks[i].view(1, 1), pos)
        torch.cuda.synchronize()
        logs = []
        t0 = time.perf_counter()
        for i in range(N):
            lg = m(toks[i].view(1, 1), pos + i)
            if i % 10 == 0:
                logs.append(lg.clone())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / N
        logs = torch.stack([x.view(-1) for x in logs]).float().cpu()
        if ref_logits is None:
            ref_logits = logs
        eng = m._engine
        res["runs"][name] = dict(ms_per_token=round(dt * 1e3, 4), tok_s=round(1 / dt, 1), streamed=bool(eng.streamed()),
                                 error_word=int(eng.stream_status()), graph=bool(eng.graph),
                                 logits_equal_first_run=bool(torch.equal(logs, ref_logits)), finite=bool(torch.isfinite(logs).all()),
                                 max_abs_diff_vs_first_run=float((logs - ref_logits).abs().max()))
weights_gb = 7.716
for r in res["runs"].values():
    r["hbm_frac"] = round(weights_gb / (r["ms_per_token"] * 1e-3) / 8000.0, 4)
print(json.dumps(res))
