"""Kernel micro-benchmarks on one MI355X (run via gpurun).  Prints achieved TFLOP/s or GB/s per kernel with
HIP-event timing on the stream the kernels are launched on (torch's current stream)."""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from aria_amd import ops  # noqa: E402

bf16 = torch.bfloat16
dev = "cuda"


def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    torch.manual_seed(0)
    res = {}
    T, D, I, E, k, V = 16384, 2560, 1664, 64, 6, 100352
    x = torch.randn(T, D, device=dev).to(bf16)
    w = (torch.randn(D, D, device=dev) * 0.02).to(bf16)
    dy = torch.randn(T, D, device=dev).to(bf16)
    for name, fn, fl in (
        ("linear_fwd_rc_rc_16384x2560x2560", lambda: ops.gemm(x, w), 2 * T * D * D),
        ("linear_dgrad_rc_oc", lambda: ops.gemm(dy, w, b_oc=True), 2 * T * D * D),
        ("linear_wgrad_oc_oc", lambda: ops.gemm(dy, x, a_oc=True, b_oc=True), 2 * T * D * D),
    ):
        t = timeit(fn)
        res[name] = dict(ms=t * 1e3, tflops=fl / t / 1e12)
    # routed experts at config #3 size
    logits = torch.randn(T, E, device=dev).to(bf16)
    scores, idx, counts = ops.moe_route(logits, k)
    off, sorted_src, inv = ops.moe_sort(idx, counts)
    M = T * k
    fc1 = (torch.randn(E, D, 2 * I, device=dev) * 0.02).to(bf16)
    fc2 = (torch.randn(E, I, D, device=dev) * 0.02).to(bf16)
    perm = ops.moe_permute(x, sorted_src, k)
    h = ops.grouped_gemm(perm, fc1, off)
    act = ops.swiglu(h)
    eo = ops.grouped_gemm(act, fc2, off)
    for name, fn, fl in (
        ("grouped_fc1_fwd", lambda: ops.grouped_gemm(perm, fc1, off), 2 * M * D * 2 * I),
        ("grouped_fc2_fwd", lambda: ops.grouped_gemm(act, fc2, off), 2 * M * I * D),
        ("grouped_fc1_dgrad", lambda: ops.grouped_gemm(h, fc1, off, w_is_kn=False), 2 * M * D * 2 * I),
        ("grouped_fc1_wgrad", lambda: ops.grouped_gemm_wgrad(perm, h, off, E), 2 * M * D * 2 * I),
    ):
        t = timeit(fn, iters=5, warmup=2)
        res[name] = dict(ms=t * 1e3, tflops=fl / t / 1e12)
    for name, fn, by in (
        ("route", lambda: ops.moe_route(logits, k), T * E * 2),
        ("sort", lambda: ops.moe_sort(idx, counts), M * 12),
        ("permute", lambda: ops.moe_permute(x, sorted_src, k), T * D * 2 + M * D * 2),
        ("unpermute", lambda: ops.moe_unpermute(eo, inv, scores, k, add=x), M * D * 2 + 2 * T * D * 2),
        ("swiglu", lambda: ops.swiglu(h), M * 3 * I * 2),
        ("rmsnorm", lambda: ops.rmsnorm(x, w[0], 1e-6), 2 * T * D * 2),
    ):
        t = timeit(fn)
        res[name] = dict(us=t * 1e6, gbps=by / t / 1e9)
    if hasattr(ops, "attention_fwd") and "--attn" in sys.argv:
        B, S, H, hd = 8, 2048, 20, 128
        qkv = torch.randn(B * S, 3 * D, device=dev).to(bf16)
        fl = 4 * B * H * S * S * hd / 2
        t = timeit(lambda: ops.attention_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B, S, H, hd, hd ** -0.5, True), 5, 2)
        res["attn_fwd_causal_8x2048"] = dict(ms=t * 1e3, tflops=fl / t / 1e12)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
