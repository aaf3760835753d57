"""Grouped-K (per-expert wgrad) and small-output dense wgrad timings on hardware."""
import os, sys, json, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from aria_amd import ops, hip
from tools.microbench import timeit
bf16 = torch.bfloat16; dev = "cuda"
E, T, topk = 64, 16384, 6
g = torch.Generator().manual_seed(1)
counts = torch.bincount(torch.randint(0, E, (T * topk,), generator=g), minlength=E)
off = torch.zeros(E + 1, dtype=torch.int32); off[1:] = torch.cumsum(counts, 0); M = int(off[-1]); offd = off.to(dev)
res = {}
def tf(flops, fn):
    t = timeit(fn, 10, 3)
    return [round(flops / t / 1e12), round(t * 1e6), hip.get_lib().cdll.aria_last_gemm_variant()]
for name, K, N in (("fc1_wgrad", 2560, 3328), ("fc2_wgrad", 1664, 2560)):
    a = torch.randn(M, K, device=dev).to(bf16); dy = torch.randn(M, N, device=dev).to(bf16)
    res[name] = tf(2 * M * K * N, lambda: ops.grouped_gemm_wgrad(a, dy, offd, E))
    res[name + "_f32acc"] = tf(2 * M * K * N, lambda: ops.grouped_gemm_wgrad(a, dy, offd, E, out_dtype=torch.float32))
    w = (torch.randn(E, K, N, device=dev) * 0.02).to(bf16)
    res[name.replace("wgrad", "fwd")] = tf(2 * M * K * N, lambda: ops.grouped_gemm(a, w, offd))
    res[name.replace("wgrad", "dgrad")] = tf(2 * M * K * N, lambda: ops.grouped_gemm(dy, w, offd, w_is_kn=False))
    del a, dy, w
# dense wgrads: dW[N_out, K_in] = dY^T X, reduction over T tokens
for name, Nout, Kin in (("o_proj_wgrad", 2560, 2560), ("qkv_wgrad", 7680, 2560), ("shared_down_wgrad", 2560, 3328), ("shared_gateup_wgrad", 6656, 2560),
                        ("lm_head_wgrad", 100352, 2560)):
    x = torch.randn(T, Kin, device=dev).to(bf16); dy = torch.randn(T, Nout, device=dev).to(bf16)
    res[name] = tf(2 * T * Nout * Kin, lambda: ops.gemm(dy, x, a_oc=True, b_oc=True))
    ops.GEMM_SPLIT_K = False
    ref = ops.gemm(dy, x, a_oc=True, b_oc=True)
    res[name + "_nosplit"] = tf(2 * T * Nout * Kin, lambda: ops.gemm(dy, x, a_oc=True, b_oc=True))
    ops.GEMM_SPLIT_K = True
    got = ops.gemm(dy, x, a_oc=True, b_oc=True)
    res[name + "_maxrel"] = float(((got.float() - ref.float()).abs().max() / ref.float().abs().max()))
    del x, dy, ref, got
# dense forward / dgrad shapes (K = 2560 or 3328: short reductions)
for name, M_, N_, K_ in (("o_proj_fwd", T, 2560, 2560), ("qkv_fwd", T, 7680, 2560), ("shared_down_fwd", T, 2560, 3328), ("shared_gateup_fwd", T, 6656, 2560)):
    x = torch.randn(M_, K_, device=dev).to(bf16); w = torch.randn(N_, K_, device=dev).to(bf16)
    res[name] = tf(2 * M_ * N_ * K_, lambda: ops.gemm(x, w))
    del x, w
print(json.dumps(res))
os.makedirs("gpurun_out", exist_ok=True); json.dump(res, open("gpurun_out/gemm_wgrad.json", "w"), indent=1)
