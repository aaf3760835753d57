"""Where does the grouped / short-K GEMM time go?  (round 2)  TF/s of the GEMM classes of the config #3 step, one-tile-per-workgroup
kernels vs the persistent form (ARIA_GEMM_PERSIST), with the decompositions that separate ragged edges, weight streaming and pipeline fill:

    python tools/gemm_probe.py > gpurun_out/gemm_probe.json

Operands rotate through several copies so that nothing stays resident in the 256 MB infinity cache between launches (in the training
step every layer brings its own weights)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from aria_amd import hip, ops  # noqa: E402

bf16 = torch.bfloat16
dev = "cuda"
res = {}


def timeit(fns, iters=12, warm=3):
    """fns: list of callables rotated through (different operand copies)"""
    for i in range(warm):
        fns[i % len(fns)]()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters):
        fns[i % len(fns)]()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def env(**kw):
    for k, v in kw.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)


MODES = {"one_tile": dict(ARIA_GEMM_PERSIST="0"), "persistent": dict(ARIA_GEMM_PERSIST="1")}

# ---- dense, K sweep (pipeline fill / drain per tile)
for M, N, K in ((16384, 2560, 2560), (16384, 2560, 5120), (16384, 2560, 10240), (16384, 7680, 2560), (78400, 4304, 1152)):
    xs = [torch.randn(M, K, device=dev).to(bf16) for _ in range(3)]
    ws = [(torch.randn(N, K, device=dev) * 0.02).to(bf16) for _ in range(3)]
    for name, e in MODES.items():
        env(ARIA_GEMM_FORCE="3", **e)
        t = timeit([lambda x=x, w=w: ops.gemm(x, w) for x, w in zip(xs, ws)])
        res[f"dense_rcrc_{M}x{N}x{K}_{name}"] = round(2 * M * N * K / t / 1e12, 1)
    del xs, ws

# ---- grouped (98304 routed rows, 64 experts)
E, T, topk = 64, 16384, 6
g = torch.Generator().manual_seed(1)
counts = torch.bincount(torch.randint(0, E, (T * topk,), generator=g), minlength=E)
off = torch.zeros(E + 1, dtype=torch.int32)
off[1:] = torch.cumsum(counts, 0)
M = int(off[-1])
offd = off.to(dev)
off_al = (torch.arange(E + 1, dtype=torch.int32) * 1536).to(dev)
lib = hip.get_lib()


def raw_grouped(a, w, out, offs, N, K, b_oc, strideB):
    lib.call("aria_grouped_gemm_bf16", a.data_ptr(), w.data_ptr(), out.data_ptr(), offs.data_ptr(), E, M, N, K, int(b_oc), a.stride(0),
             w.shape[2], strideB, out.stride(0), torch.cuda.current_stream().cuda_stream)


for name, K, N in (("fc1", 2560, 3328), ("fc2", 1664, 2560)):
    a = [torch.randn(M, K, device=dev).to(bf16) for _ in range(2)]
    w = [(torch.randn(E, K, N, device=dev) * 0.02).to(bf16) for _ in range(3)]
    out = torch.empty(M, N, dtype=bf16, device=dev)
    flops = 2 * M * K * N
    for vname, ve in (("v2", dict(ARIA_GEMM_FORCE="2", ARIA_GEMM_PERSIST="0")), ("v3", dict(ARIA_GEMM_FORCE="3", ARIA_GEMM_PERSIST="0")),
                      ("v3p", dict(ARIA_GEMM_FORCE="3", ARIA_GEMM_PERSIST="1"))):
        env(**ve)
        res[f"{name}_fwd_{vname}"] = round(flops / timeit([lambda i=i: raw_grouped(a[i % 2], w[i], out, offd, N, K, 1, K * N) for i in range(3)]) / 1e12, 1)
        res[f"{name}_fwd_{vname}_aligned_counts"] = round(flops / timeit([lambda i=i: raw_grouped(a[i % 2], w[i], out, off_al, N, K, 1, K * N) for i in range(3)]) / 1e12, 1)
        res[f"{name}_fwd_{vname}_shared_weight"] = round(flops / timeit([lambda i=i: raw_grouped(a[i % 2], w[i], out, offd, N, K, 1, 0) for i in range(3)]) / 1e12, 1)
    # dgrad through the same storage: d_in [M, K] = d_out [M, N] @ W_e^T
    dy = [torch.randn(M, N, device=dev).to(bf16) for _ in range(2)]
    din = torch.empty(M, K, dtype=bf16, device=dev)
    for vname, ve in (("v3", dict(ARIA_GEMM_FORCE="3", ARIA_GEMM_PERSIST="0")), ("v3p", dict(ARIA_GEMM_FORCE="3", ARIA_GEMM_PERSIST="1"))):
        env(**ve)
        res[f"{name}_dgrad_{vname}"] = round(flops / timeit([lambda i=i: ops.grouped_gemm(dy[i % 2], w[i], offd, w_is_kn=False, out=din) for i in range(3)]) / 1e12, 1)
    # weight gradient
    gw = torch.empty(E, K, N, dtype=bf16, device=dev)
    for vname, ve in (("v2", dict(ARIA_GEMM_FORCE="2", ARIA_GEMM_PERSIST="0")), ("v3", dict(ARIA_GEMM_FORCE="3", ARIA_GEMM_PERSIST="0")),
                      ("v3p", dict(ARIA_GEMM_FORCE="3", ARIA_GEMM_PERSIST="1"))):
        env(**ve)
        res[f"{name}_wgrad_{vname}"] = round(flops / timeit([lambda i=i: ops.grouped_gemm_wgrad(a[i % 2], dy[i % 2], offd, E, out=gw) for i in range(2)]) / 1e12, 1)
    del a, w, out, dy, din, gw
print(json.dumps(res, indent=1))
