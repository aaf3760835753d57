"""Attention backward on one box: the round-2 library (build/old/libaria_hip.so: bwd3 dK/dV + role-split dQ, register-staged tiles) against
this tree's (bwd3 dK/dV with LDS-DMA staging and early statistics + dQ v5 without role split).  At the commit that produced
profiles/r03_attn_bwd_ab.json the tree also held the role-split dQ kernel (ARIA_ATTN_BWD=3) and the single-pass form with fp32 dQ adds
(aria_attn_bwd_ws; now tools/probes/src/attn_bwd4_single_pass.hip): those arms run only if the library still exports them.  Loads each library by path through ctypes (no package import: the old build lacks the new symbols).  HIP-event timing of
the whole aria_attn_bwd call (delta + kernels).  Writes gpurun_out/attn_bwd_r3_ab.json."""
import ctypes, json, os, sys
import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
P, I64, F32, I32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_int
bf16, dev = torch.bfloat16, "cuda"


def load(path):
    lib = ctypes.CDLL(path)
    lib.aria_attn_fwd.argtypes = [P] * 7 + [I64] * 9 + [F32, I32, P]
    lib.aria_attn_bwd.argtypes = [P] * 12 + [I64] * 12 + [F32, I32, P]
    if hasattr(lib, "aria_attn_bwd_ws"):
        lib.aria_attn_bwd_ws.argtypes = [P] * 12 + [I64] * 12 + [F32, I32, P, I64, P]
        lib.aria_attn_bwd_workspace_bytes.argtypes = [I64] * 4
        lib.aria_attn_bwd_workspace_bytes.restype = I64
    return lib


def bench(lib, B, S, H, hd, causal, single_pass=False, iters=5, warm=2):
    D = H * hd
    torch.manual_seed(0)
    qkv = torch.randn(B * S, 3 * D, device=dev).to(bf16)
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    o = torch.empty(B * S, D, dtype=bf16, device=dev)
    lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    sc = hd ** -0.5
    assert lib.aria_attn_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), None, None, B, S, S, H, hd, 3 * D, 3 * D, 3 * D, D, sc,
                             int(causal), st) == 0
    do = torch.randn_like(o)
    dq, dk, dv = (torch.empty_like(o) for _ in range(3))
    delta = torch.empty_like(lse)
    ws = None
    if single_pass:
        n = lib.aria_attn_bwd_workspace_bytes(B, S, H, hd)
        ws = torch.empty(n, dtype=torch.uint8, device=dev)

    def f():
        args = [q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(), delta.data_ptr(), dq.data_ptr(), dk.data_ptr(),
                dv.data_ptr(), None, None, B, S, S, H, hd, 3 * D, 3 * D, 3 * D, D, D, D, D, sc, int(causal)]
        rc = lib.aria_attn_bwd_ws(*args, ws.data_ptr(), ws.numel(), st) if single_pass else lib.aria_attn_bwd(*args, st)
        assert rc == 0, rc

    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        f()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    fl = 2.5 * 4 * B * H * S * S * hd / (2 if causal else 1)
    return {"ms": round(ms, 3), "algorithmic_TF_s": round(fl / ms / 1e9, 1)}, [t.clone() for t in (dq, dk, dv)]


new, old = load(os.path.join(ROOT, "aria_amd", "libaria_hip.so")), load(os.path.join(ROOT, "build", "old", "libaria_hip.so"))
res = {}
for name, (B, S, H) in {"config3_8x2048_h20": (8, 2048, 20), "long_1x16384_h20": (1, 16384, 20), "long_1x65536_h20": (1, 65536, 20)}.items():
    it = 3 if S > 20000 else 6
    r = {}
    r["r02"], ref = bench(old, B, S, H, 128, True, iters=it)
    r["r03_two_kernels"], got = bench(new, B, S, H, 128, True, iters=it)
    r["bit_identical_to_r02"] = all(torch.equal(a, b) for a, b in zip(got, ref))
    if hasattr(new, "aria_attn_bwd_ws"):
        r["r03_single_pass_atomics"], sp = bench(new, B, S, H, 128, True, single_pass=True, iters=max(2, it // 2))
        r["single_pass_max_abs_diff_dq"] = float((sp[0].float() - ref[0].float()).abs().max())
        r["single_pass_dk_dv_identical"] = bool(torch.equal(sp[1], ref[1]) and torch.equal(sp[2], ref[2]))
    r["r02_again"], _ = bench(old, B, S, H, 128, True, iters=it)
    res[name] = r
    print(json.dumps({name: r}), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "attn_bwd_r3_ab.json"), "w"), indent=1)
