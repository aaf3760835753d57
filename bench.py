#!/usr/bin/env python
"""bench.py -- tokens/s (fwd+bwd) of the Aria-25.3B hot path on N MI355X (one process per GPU, RCCL over xGMI).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = one forward + backward of the hot path over one synthetic micro-batch per GPU at BASELINE.json config #3's
per-GPU shape (per_device_train_batch_size 8 x max_seq_length 2048, recipes/config_full.yaml:12,30): full-width, full-depth
Aria-25.3B MoE decoder (28 layers, 64 experts top-6, vocab 100352, random-init bf16 weights N(0,0.02)), shifted masked
cross-entropy, full backward incl. router aux-loss gradients; data-parallel gradient all-reduce overlapped with backward
when N > 1 (weak scaling: per-GPU work fixed).  Inputs are resident in HBM before the timed region.
Prints ONE JSON line (rank 0) with the driver's contract fields + `roofline` (dominant kernel: the fc1 grouped expert GEMM with its SwiGLU epilogue, gemm3.hip,
timed live with HIP events on the launch stream) + `cpu_baseline` (the CPU oracle timed on the host cores, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL across processes on this driver); before the HIP runtime starts

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

bf16 = torch.bfloat16


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--time-grouped", action="store_true", help="diagnostics: HIP-event time of every grouped-M GEMM class -> stderr")
    ap.add_argument("--batch", type=int, default=8, help="per-GPU micro-batch (config_full.yaml: 8)")
    ap.add_argument("--seq", type=int, default=2048, help="max_seq_length (config_full.yaml: 2048)")
    ap.add_argument("--layers", type=int, default=28, help="debug only: fewer layers makes the number INVALID")
    ap.add_argument("--recompute", action="store_true", help="gradient checkpointing per layer (reference recipe); "
                    "off by default: 288 GB HBM holds all activations")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--images", type=int, default=2, help="980px images per sample (NLVR2-style: 2); 0 = text only")
    ap.add_argument("--vit-layers", type=int, default=27, help="debug only")
    ap.add_argument("--no-overlap", action="store_true", help="exchange gradients after backward instead of under it")
    ap.add_argument("--allreduce", action="store_true", help="all-reduce every gradient (each replica keeps the full average) instead of the "
                    "ZeRO-2 reduce-scatter onto the owner of the optimizer shard (recipes/accelerate_configs/zero2.yaml)")
    ap.add_argument("--long", action="store_true", help="north_star's target shape instead of config #3's: ONE 65 536-token image+text sequence "
                    "per GPU (8 x 980px images = 2048 image tokens + text), gradient checkpointing on (activations of 28 layers at 64K tokens "
                    "exceed 288 GB otherwise); same metric, reported as a second documented line (profiles/)")
    ap.add_argument("--no-long64k", action="store_true", help="skip the `long64k` sub-record (N = 1 only: three timed steps of the north_star "
                    "target shape -- one 65 536-token image+text sequence, gradient checkpointing on -- after the main measurement)")
    ap.add_argument("--long64k-seq", type=int, default=65536, help="debug only (CPU dry run of the sub-record's code path)")
    ap.add_argument("--long64k-images", type=int, default=8, help="debug only")
    ap.add_argument("--no-inference-records", action="store_true", help="skip the `generate_config2` / `prefill_config4` sub-records (N = 1 only: "
                    "BASELINE configs #2 and #4 on the gptfast surface of the SAME weights, after the training measurement)")
    ap.add_argument("--gen-new", type=int, default=200, help="debug only (config #2 protocol: 200 new tokens)")
    ap.add_argument("--gen-image", type=int, default=1, help="debug only (config #2: one 980px image)")
    ap.add_argument("--prefill-seq", type=int, default=53248, help="debug only (config #4: 32 x 128 frame tokens + 49 152 text tokens)")
    ap.add_argument("--prefill-frames", type=int, default=32, help="debug only (config #4: 32 frames at 490px)")
    ap.add_argument("--sub-record-repeats", type=int, default=0, help="debug only (the CPU dry run): > 0 = that many timed repetitions in the "
                    "generate / LoRA / fusions-A/B sub-records instead of their protocol's 5 / 3 / 4")
    ap.add_argument("--no-fusions-ab", action="store_true", help="skip the `step_fusions_ab` sub-record (N = 1 only: 4 + 4 extra steps, the round-5 "
                    "launch fusions switched off / on alternately inside this process -- box variance cancels)")
    ap.add_argument("--no-launch-classes", action="store_true", help="skip the `launch_classes` pass (N = 1 only: ONE extra step after the timed "
                    "region with HIP events around every large launch class -> `roofline.worst_large_launch`)")
    ap.add_argument("--no-lora-record", action="store_true", help="skip the `lora_config` sub-record (N = 1 only: recipes/config_lora.yaml's adapter set "
                    "on the same model and micro-batch, three timed steps + the frozen-base forward + input-gradient reference)")
    ap.add_argument("--ep", action="store_true", help="BASELINE config #5 instead of #3: routed experts sharded over the N ranks (all-to-all "
                                                      "dispatch over xGMI), everything else data-parallel; not what the driver runs")
    args = ap.parse_args()
    if args.long:
        args.batch, args.seq, args.images, args.recompute = 1, 65536, 8, True
    return args


def init_params(model, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("norm.weight") or n.endswith("layernorm.weight"):
                p.fill_(1.0)
            else:
                # N(0, 0.02): router / expert weights are torch.empty in the reference -> explicit init
                chunk = 1 << 28
                flat = p.view(-1)
                for o in range(0, flat.numel(), chunk):
                    flat[o:o + chunk].normal_(0.0, 0.02, generator=g)


PMC_FILE = "r06_pmc_fc1.json"
KERNEL_REV = "r06"    # bumped with every change of the v3 K loop / epilogues: a PMC pass of an older build does not describe this one
PMC_KERNEL_TAG = "gemm3_kernel<rc,oc,8>+gather+wide_store+ragged_last+swiglu@r06"   # the fc1 launch the committed PMC pass profiled


def fc1_kernel_tag(variant, gather=None):
    """What the default path launched for experts.fc1 in THIS run, spelled like profiles/r06_pmc_fc1.json's kernel_tag."""
    if variant != 3:
        return f"gemm{variant}_kernel<rc,oc>"
    if gather is None:
        gather = os.environ.get("ARIA_FUSE_GATHER", "1") != "0" and os.environ.get("ARIA_FUSE_WGRAD_GATHER", "1") != "0" and \
            os.environ.get("ARIA_FUSE_SWIGLU", "1") != "0"
    wide = os.environ.get("ARIA_GEMM_WIDE_STORE", "1") != "0"
    fused = os.environ.get("ARIA_FUSE_SWIGLU", "1") != "0"
    order = os.environ.get("ARIA_GEMM_ORDER")
    ragged_last = fused if order is None else bool(int(order) & 512)   # the fused launch's default tile order since r04b
    return ("gemm3_kernel<rc,oc," + ("8>+gather" if gather and fused else "3>") + ("+wide_store" if wide else "") +
            ("+ragged_last" if ragged_last else "+expert_major") + ("+swiglu" if fused else "") + "@" + KERNEL_REV)


def pmc_traffic(variant=3, gather=None):
    """Bytes beyond the L2s per fc1 launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs of
    tools/gemm_pmc_target.py = the same shape and the same fused launch; gfx950 FETCH_SIZE x2 correction applied) -- counters cannot be read inside the timed run.
    None unless the pass profiled exactly the kernel / epilogue / tile order this run launched."""
    try:
        with open(os.path.join(ROOT, "profiles", PMC_FILE)) as f:
            d = json.load(f)
        return round(d["hbm_bytes_per_launch"]) if d["kernel_tag"] == fc1_kernel_tag(variant, gather) else None
    except Exception:
        return None


def cpu_baseline(cfg_kwargs, seconds_budget=40.0):
    """The CPU oracle (oracle/aria_oracle.py, 'port' of the reference algorithm: the reference itself cannot travel to the GPU box) timed on
    this host at the benchmark's sequence shape: one full-width decoder layer + lm_head/CE, fwd+bwd, fp32, B=1 S=2048 (one of the 8
    samples of the micro-batch); extrapolated to 28 layers."""
    from oracle import aria_oracle as O

    torch.manual_seed(0)
    nthreads = torch.get_num_threads()
    cfg = O.LMConfig(**{k: v for k, v in cfg_kwargs.items() if k in O.LMConfig.__dataclass_fields__})
    cfg.num_hidden_layers = 1
    D, E, I, V = cfg.hidden_size, cfg.moe_num_experts, cfg.moe_intermediate_size, cfg.vocab_size
    I2 = I * cfg.moe_num_shared_experts
    shapes = {
        "model.layers.0.self_attn.q_proj.weight": (D, D), "model.layers.0.self_attn.k_proj.weight": (D, D),
        "model.layers.0.self_attn.v_proj.weight": (D, D), "model.layers.0.self_attn.o_proj.weight": (D, D),
        "model.layers.0.mlp.router.weight": (E, D), "model.layers.0.mlp.experts.fc1.weight": (E, D, 2 * I),
        "model.layers.0.mlp.experts.fc2.weight": (E, I, D), "model.layers.0.mlp.shared_experts.gate_proj.weight": (I2, D),
        "model.layers.0.mlp.shared_experts.up_proj.weight": (I2, D), "model.layers.0.mlp.shared_experts.down_proj.weight": (D, I2),
    }
    w = {k: (torch.randn(*s) * 0.02).requires_grad_(True) for k, s in shapes.items()}
    w["model.layers.0.input_layernorm.weight"] = torch.ones(D, requires_grad=True)
    w["model.layers.0.post_attention_layernorm.weight"] = torch.ones(D, requires_grad=True)
    S = 2048
    pos = torch.arange(S)[None]

    def layer_step(s):
        x = torch.randn(1, s, D, requires_grad=True)
        y = O.decoder_layer(x, w, "model.layers.0.", cfg, pos[:, :s], training=True)
        y.sum().backward()

    layer_step(S)                      # warm-up at the timed shape (allocator, thread pool, oneDNN primitive caches)
    t_layers = []
    for _ in range(3):                 # 1 warm-up + 3 timed, median
        t0 = time.perf_counter()
        layer_step(S)
        t_layers.append(time.perf_counter() - t0)
    t_layer = sorted(t_layers)[1]
    lm_w = (torch.randn(V, D) * 0.02).requires_grad_(True)
    h = torch.randn(S, D, requires_grad=True)
    labels = torch.randint(0, V, (S,))
    t_heads = []
    for i in range(3):                 # first pass = warm-up, median of the other two is the smaller-or-equal middle: keep all three, take the median
        t0 = time.perf_counter()
        loss = torch.nn.functional.cross_entropy(torch.nn.functional.linear(h, lm_w), labels)
        loss.backward()
        t_heads.append(time.perf_counter() - t0)
    t_head = sorted(t_heads)[1]
    value = S / (28 * t_layer + t_head)
    ref = {}
    try:   # the LIVE reference timed in the build container (it cannot travel): committed figures, carried beside the port's
        with open(os.path.join(ROOT, "profiles", "r05_cpu_reference_config1.json")) as f:
            d = json.load(f)
        ref = {"reference_tokens_per_s": d["port_shape"]["reference_tokens_per_s_28_layers"],
               "reference_cores": d["host"]["torch.get_num_threads"],
               "reference_sample": "LIVE reference (aria/model, sequential_gemm fallback, eager attention) at THIS sample's shape -- one full-width decoder "
                                   "layer fwd+bwd B=1 S=2048, best of 5, x 28 layers -- timed in the build container "
                                   "(profiles/r05_cpu_reference_config1.json); the same file's config #1 forward: reference "
                                   f"{d['config1']['extrapolated_28_layers']['reference_tokens_per_s']} tok/s vs port "
                                   f"{d['config1']['extrapolated_28_layers']['port_tokens_per_s']}",
               "port_tokens_per_s_same_host_as_reference": d["port_shape"]["port_tokens_per_s_28_layers"]}
    except Exception:  # noqa: BLE001
        pass
    return {"value": round(value, 3), "unit": "tokens/s", "cores": nthreads, "kind": "port", **ref,
            "sample": f"oracle fp32, 1 of 28 full-width decoder layers fwd+bwd (1 warm-up + 3 timed at B=1,S={S}: "
                      f"{', '.join(f'{t:.2f}' for t in t_layers)} s, median {t_layer:.2f}) + lm_head/CE fwd+bwd (median of 3: {t_head:.2f} s); "
                      f"value = S/(28*t_layer+t_head); os.cpu_count()={os.cpu_count()}; the LIVE reference (cannot travel to this box) timed in the build "
                      f"container, 8 cores (profiles/r05_cpu_reference_config1.json): config #1 forward 39.2 tok/s vs this port 36.7; at THIS shape "
                      f"31.0 s per layer vs the port's 3.1 (its sequential_gemm backward builds full-size zero gradients) -- the port overstates the "
                      f"reference's CPU rate ~10x here"}


def long64k_record(model, cfg, make_inputs, ops, steps=3, warmup=1, S=65536, n_img=8):
    """north_star's target shape next to the config #3 line, in the SAME run: ONE 65 536-token image+text sequence (8 x 980px images =
    2048 image tokens + text) per step, fwd+bwd with the recipe's gradient checkpointing (activations of 28 layers at 64K tokens exceed
    288 GB otherwise); same metric.  `roofline` here = the attention backward (half of this step): algorithmic flops per call =
    2.5 x the causal forward's 4 * (S^2 / 2) * hd * H (five GEMM units of S x S x hd per head, recompute of S inside the kernel not
    counted), duration from HIP events around every attention-backward call of the timed steps."""
    H, hd = cfg.num_attention_heads, cfg.hidden_size // cfg.num_attention_heads
    model.zero_grad(set_to_none=True)
    torch.cuda.empty_cache()
    before = cfg.gradient_checkpointing
    cfg.gradient_checkpointing = True
    batch = make_inputs(1, S, n_img, 4321)
    events = []
    orig_bwd = ops.attention_bwd
    timing = {"on": False}

    def timed_attention_bwd(q, k, v, o, do, lse, B, S_, H_, hd_, *a, **kw):
        if not (timing["on"] and S_ == S and hd_ == hd):
            return orig_bwd(q, k, v, o, do, lse, B, S_, H_, hd_, *a, **kw)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = orig_bwd(q, k, v, o, do, lse, B, S_, H_, hd_, *a, **kw)
        e.record()
        events.append((s, e))
        return r

    ops.attention_bwd = timed_attention_bwd
    try:
        def step():
            model.zero_grad(set_to_none=True)
            out = model(**batch, return_logits=False)
            out.loss.backward()
            return out.loss

        for _ in range(warmup):
            loss = step()
        torch.cuda.synchronize()
        timing["on"] = True
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        ops.attention_bwd = orig_bwd
        cfg.gradient_checkpointing = before
    durs = [s.elapsed_time(e) * 1e-3 for s, e in events]
    avg = sum(durs) / max(1, len(durs))
    flops = 2.5 * 4.0 * (S * S / 2.0) * hd * H
    return {"workload": f"north_star target shape: Aria-25.3B random-init, ONE {S}-token sequence ({n_img} x 980px images + text), frozen ViT fwd -> "
                        "projector -> 28-layer MoE decoder fwd+bwd, gradient checkpointing on (level in `recompute_level`: 'moe' keeps a layer's "
                        "token-sized tensors and rebuilds the expert-row tensors, 'layer' keeps layer inputs + flash (o, lse))",
            "recompute_level": getattr(model.language_model.model, "last_recompute_level", None),
            "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 1), "value": round(S * steps / dt, 1), "unit": "tokens/s",
            "loss": round(float(loss), 4), "max_mem_GB": round(torch.cuda.max_memory_allocated() / 1e9, 1),
            "roofline": {"kernel": f"attention backward (aria_attn_bwd: delta + dK/dV/dQ), causal S={S}, {H} x {hd}", "bound": "mfma",
                         "achieved": round(flops / avg / 1e12, 1) if durs else None, "peak": 2500.0, "unit": "TFLOP/s",
                         "frac": round(flops / avg / 2.5e15, 4) if durs else None, "launches_timed": len(durs),
                         "avg_call_ms": round(avg * 1e3, 3), "algorithmic_flops_per_call": flops}}


def decode_weight_bytes(t) -> int:
    """Bytes of bf16 weights one decoded token streams (7.716 GB at Aria's shape): per layer the k routed experts' three
    matrices, the shared expert, the router and the four attention projections; + lm_head."""
    D, I, Is = t.hidden_size, t.moe_intermediate_size, t.moe_intermediate_size * t.moe_num_shared_experts
    per_layer = t.moe_topk * 3 * D * I + 3 * D * Is + t.moe_num_experts * D + 4 * D * D
    return (t.num_hidden_layers * per_layer + t.vocab_size * D) * 2


def generate_config2_record(twin, tcfg, new_tokens=200, runs=5, warmup=2, n_img=1, img_px=980, qtok=256, img_token=9):
    """BASELINE config #2 by the reference's protocol (gptfast/benchmark.py:10-48): one 980px image (256 image tokens) + a short prompt =
    280 positions, max_new_tokens 200, top-k 200, temperature 0.8, 2 warm-up + 5 timed whole generates (ViT + prefill + decode + sampling),
    tok/s = mean(#new tokens) / mean(latency); then 50 decode steps alone for the per-token figure its HBM roofline is quoted on."""
    from aria_amd import gptfast as G

    dev = twin.llm.output.weight.device
    g = torch.Generator(device="cuda").manual_seed(2)
    T = 24 + n_img * qtok
    ids = torch.randint(10, tcfg.vocab_size, (1, T), generator=g, device=dev)
    pv = pm = None
    if n_img:
        ids[:, 8:8 + n_img * qtok] = img_token
        pv = torch.randn((n_img, 3, img_px, img_px), generator=g, device=dev).clamp_(-1, 1).to(bf16)
        pm = torch.ones((n_img, img_px, img_px), dtype=torch.bool, device=dev)
    twin.setup_caches(1, T + new_tokens)
    decoder, lat, ntok = None, [], []
    with torch.no_grad():
        for i in range(warmup + runs):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out, decoder = G.generate(twin, ids, new_tokens, pixel_values=pv, pixel_mask=pm, temperature=0.8, top_k=200, decoder=decoder)
            torch.cuda.synchronize()
            if i >= warmup:
                lat.append(time.perf_counter() - t0)
                ntok.append(out.numel() - ids.numel())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        emb = twin.prepare_embeddings(ids, pv, pm)
        twin(None, torch.arange(T, device=dev), emb, last_only=True)
        torch.cuda.synchronize()
        t_prefill = time.perf_counter() - t0
        pos = torch.tensor([T], device=dev, dtype=torch.int32)
        tok = torch.tensor([[11]], device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            decoder(tok, pos)
        torch.cuda.synchronize()
        t_dec = (time.perf_counter() - t0) / 50
    wbytes = decode_weight_bytes(tcfg)
    eng = twin.llm._engine
    return {"workload": f"config#2 (gptfast/benchmark.py protocol): {n_img} x {img_px}px image + prompt = {T} positions, {new_tokens} new tokens, top-k 200, "
                        f"T 0.8, {warmup} warm-up + {runs} timed whole generates (ViT + prefill + decode + sampling); same random-init weights",
            "value": round(sum(ntok) / sum(lat), 2), "unit": "tokens/s", "runs": runs, "warmup": warmup, "new_tokens": new_tokens,
            "mean_latency_s": round(sum(lat) / len(lat), 4), "published_h100": {"eager": 25.2, "compile": 130.0},
            "prefill_ms_incl_vit": round(t_prefill * 1e3, 2), "decode_ms_per_token": round(t_dec * 1e3, 3),
            "decode_engine": bool(eng is not None),
            "decode_schedule": "6 launches per layer",
            "roofline": {"kernel": "decode step incl. sampling (aria_decode_token + aria_sample_topk)", "bound": "hbm",
                         "achieved": round(wbytes / t_dec / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(wbytes / t_dec / 8e12, 4),
                         "algorithmic_bytes_per_token": wbytes, "traffic": None}}


def prefill_config4_record(twin, tcfg, S=53248, frames=32, runs=2, img_px=490, qtok=128, img_token=9, vit_tf_per_frame=1.19e12):
    """BASELINE config #4: ONE long-context prefill (32 frames at 490px = 4096 image tokens + 49 152 text tokens) through the gptfast
    surface with a bf16 KV cache, last-position logits; 1 warm-up + `runs` timed.  Roofline: bf16 MFMA on the algorithmic flops
    (7.716 GF/token of GEMMs, 28 * 4 * D * S/2 per token of causal attention, 1.19 TF per 490px frame of ViT)."""
    dev = twin.llm.output.weight.device
    g = torch.Generator(device="cuda").manual_seed(4)
    ids = torch.randint(10, tcfg.vocab_size, (1, S), generator=g, device=dev)
    pv = pm = None
    if frames:
        ids[:, 16:16 + qtok * frames] = img_token
        pv = torch.randn((frames, 3, img_px, img_px), generator=g, device=dev).clamp_(-1, 1).to(bf16)
        pm = torch.ones((frames, img_px, img_px), dtype=torch.bool, device=dev)
    twin.setup_caches(1, S)
    torch.cuda.reset_peak_memory_stats()
    ts = []
    with torch.no_grad():
        for i in range(runs + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            emb = twin.prepare_embeddings(ids, pv, pm)
            lg = twin(None, torch.arange(S, device=dev), emb, last_only=True)
            torch.cuda.synchronize()
            if i:
                ts.append(time.perf_counter() - t0)
    t = sum(ts) / len(ts)
    D, L = tcfg.hidden_size, tcfg.num_hidden_layers
    gemm_per_token = decode_weight_bytes(tcfg)          # 2 flops per weight of the token's path = bytes of bf16 weights
    head = 2.0 * tcfg.vocab_size * D                    # lm_head per position: last_only=True computes ONE position's logits, so only one is
    flops = S * (gemm_per_token - head) + head + L * 4.0 * D * (S / 2.0) * S + frames * vit_tf_per_frame   # counted (executed work)
    return {"workload": f"config#4: ONE {S}-position prefill ({frames} x {img_px}px frames = {frames * qtok} image tokens + text) on the gptfast surface, "
                        f"bf16 KV cache, last-position logits (lm_head counted for that ONE position; gptfast/model.py:232-233 computes all S: +{S * 2.0 * tcfg.vocab_size * D / 1e12:.0f} TF there); "
                        f"1 warm-up + {runs} timed; same random-init weights",
            "value": round(S / t, 1), "unit": "tokens/s", "seconds": round(t, 4), "runs": runs, "finite_logits": bool(torch.isfinite(lg.float()).all()),
            "kv_cache_GB": round(L * 2 * S * D * 2 / 1e9, 1), "max_mem_GB": round(torch.cuda.max_memory_allocated() / 1e9, 1),
            "roofline": {"kernel": "whole prefill (GEMMs + causal attention + ViT)", "bound": "mfma", "achieved": round(flops / t / 1e12, 1),
                         "peak": 2500.0, "unit": "TFLOP/s", "frac": round(flops / t / 2.5e15, 4), "algorithmic_flops": flops, "traffic": None}}


FUSION_SWITCHES = ("ARIA_FUSE_WGRAD_GATHER", "ARIA_FUSE_ROUTER", "ARIA_FUSE_QKV_ROPE")


def executed_flops_per_step(cfg, B, S, labelled_rows, n_images, vit_layers=27, P=4900, Dv=1152, Iv=4304, Hv=16, hdv=72, qtok=256):
    """The flops the config #3 step EXECUTES (not the algorithmic 3 x forward of SURVEY 8d: the ViT is frozen = forward only, lm_head + CE run
    on the labelled rows only, the attention backward is counted at the algorithm's 2.5 x forward).  Decoder GEMMs per token and layer =
    SURVEY 8d's 2 * (4 D^2 + D E + k 3 D I + 3 D Is); x 3 for forward + input gradient + weight gradient.  ViT per image and layer =
    2 P (4 Dv^2 + 2 Dv Iv) + 4 P^2 hd H.  Projector (trainable) per image ~ 0.1 TF forward, x 3."""
    D, E, k, I = cfg.hidden_size, cfg.moe_num_experts, cfg.moe_topk, cfg.moe_intermediate_size
    Is = I * cfg.moe_num_shared_experts
    T = B * S
    H, hd = cfg.num_attention_heads, cfg.hidden_size // cfg.num_attention_heads
    layer_gemm = 2.0 * (4 * D * D + D * E + k * 3 * D * I + 3 * D * Is)
    llm = 3.0 * cfg.num_hidden_layers * layer_gemm * T
    attn = 3.5 * cfg.num_hidden_layers * 4.0 * B * H * (S * S / 2.0) * hd
    head = 3.0 * 2.0 * D * cfg.vocab_size * labelled_rows
    vit = n_images * vit_layers * (2.0 * P * (4 * Dv * Dv + 2 * Dv * Iv) + 4.0 * P * P * hdv * Hv) + n_images * 2.0 * P * 588 * Dv
    proj = 3.0 * n_images * (2.0 * P * 2 * Dv * Dv * 2 + 4.0 * qtok * P * Dv + 2.0 * qtok * (Dv * D + D * D))
    return {"decoder_gemms": llm, "decoder_attention": attn, "lm_head": head, "vit_forward": vit, "projector": proj,
            "total": llm + attn + head + vit + proj}


def launch_classes_record(step, ops, peak_tf=2500.0, min_ms=4.0):
    """ONE extra step after the timed region with a HIP-event pair around every large launch class (GEMM families by operand form and
    shape, attention by head dim and direction): per class launches, ms per step, TF/s on the launch's own algorithmic flops, fraction of
    the dense bf16 MFMA peak.  `worst_large_launch` in the driver line = the lowest fraction among the classes that take >= min_ms of the
    step (the headline `roofline` is the BEST grouped launch; a reader also needs the worst and the whole step)."""
    ev = {}

    def wrap(name, key_flops):
        orig = getattr(ops, name)

        def timed(*a, **kw):
            key, fl = key_flops(*a, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = orig(*a, **kw)
            e.record()
            rec = ev.setdefault(key, [0.0, []])
            rec[0] += fl
            rec[1].append((s, e))
            return r

        setattr(ops, name, timed)
        return orig

    def kf_gemm(a, b, **kw):
        a_oc, b_oc = bool(kw.get("a_oc")), bool(kw.get("b_oc"))
        M, K = (a.shape[1], a.shape[0]) if a_oc else (a.shape[0], a.shape[1])
        N = b.shape[1] if b_oc else b.shape[0]
        return f"gemm {'oc' if a_oc else 'rc'},{'oc' if b_oc else 'rc'} M{M} N{N} K{K}", 2.0 * M * N * K

    table = {
        "gemm": kf_gemm,
        "gemm_swiglu": lambda x, w, **kw: (f"gemm+swiglu M{x.shape[0]} N{w.shape[0]} K{x.shape[1]}", 2.0 * x.shape[0] * w.shape[0] * x.shape[1]),
        "gemm_dswiglu": lambda dy, w, h, **kw: (f"gemm+dswiglu M{dy.shape[0]} I{h.shape[1] // 2} K{dy.shape[1]}", 2.0 * dy.shape[0] * (h.shape[1] // 2) * dy.shape[1]),
        "gemm_qkv_rope": lambda x, wqkv, *a, **kw: (f"gemm+rope (q|k|v) M{x.shape[0]} N{wqkv.shape[0]} K{x.shape[1]}", 2.0 * x.shape[0] * wqkv.shape[0] * x.shape[1]),
        "grouped_gemm": lambda a, w, off, **kw: (f"grouped {'fwd' if kw.get('w_is_kn', True) else 'dgrad'} rows{a.shape[0]} w{tuple(w.shape[1:])}",
                                                 2.0 * a.shape[0] * w.shape[1] * w.shape[2]),
        "grouped_gemm_swiglu": lambda a, w, off, **kw: (f"grouped fc1+swiglu rows{a.shape[0]} w{tuple(w.shape[1:])}", 2.0 * a.shape[0] * w.shape[1] * w.shape[2]),
        "grouped_gemm_swiglu_gather": lambda x, rows, w, off, **kw: (f"grouped fc1+swiglu (gathered rows) rows{rows.numel()} w{tuple(w.shape[1:])}",
                                                                    2.0 * rows.numel() * w.shape[1] * w.shape[2]),
        "grouped_gemm_dswiglu": lambda dy, w, off, h, **kw: (f"grouped fc2 dgrad+dswiglu rows{dy.shape[0]} w{tuple(w.shape[1:])}",
                                                            2.0 * dy.shape[0] * w.shape[1] * w.shape[2]),
        "grouped_gemm_wgrad": lambda a, dy, off, E, **kw: (f"grouped wgrad rows{a.shape[0]} [{a.shape[1]} x {dy.shape[1]}]", 2.0 * a.shape[0] * a.shape[1] * dy.shape[1]),
        "grouped_gemm_wgrad_gather": lambda x, rows, dy, off, E, **kw: (f"grouped wgrad (gathered rows) rows{dy.shape[0]} [{x.shape[1]} x {dy.shape[1]}]",
                                                                       2.0 * dy.shape[0] * x.shape[1] * dy.shape[1]),
        "attention_fwd": lambda q, k, v, B, S, H, hd, scale, causal, *a, **kw: (f"attention fwd hd{hd} S{S}{' causal' if causal else ''}",
                                                                                4.0 * B * H * S * S * hd * (0.5 if causal else 1.0)),
        "attention_bwd": lambda q, k, v, o, do, lse, B, S, H, hd, scale, causal, *a, **kw: (f"attention bwd hd{hd} S{S}{' causal' if causal else ''}",
                                                                                           2.5 * 4.0 * B * H * S * S * hd * (0.5 if causal else 1.0)),
    }
    saved = {name: wrap(name, kf) for name, kf in table.items() if hasattr(ops, name)}
    try:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        for name, orig in saved.items():
            setattr(ops, name, orig)
    rows = []
    for key, (fl, pairs) in ev.items():
        ms = sum(s.elapsed_time(e) for s, e in pairs)
        if ms <= 0:
            continue
        tf = fl / (ms * 1e-3) / 1e12
        rows.append({"class": key, "launches": len(pairs), "ms_per_step": round(ms, 2), "tflops": round(tf, 1), "frac": round(tf / peak_tf, 4)})
    rows.sort(key=lambda r: -r["ms_per_step"])
    large = [r for r in rows if r["ms_per_step"] >= min_ms]
    worst = min(large, key=lambda r: r["frac"]) if large else None
    return {"step_ms_with_events": round(dt * 1e3, 1), "classes_ge_%gms" % min_ms: large, "timed_ms_total": round(sum(r["ms_per_step"] for r in rows), 1),
            "worst": worst}


def step_fusions_ab(step, pairs=4):
    """The config #3 step with round 5's launch fusions OFF and ON, alternating inside this process (same box, same clocks, same allocator
    state): OFF = the tokens permuted into a [6T, D] copy for fc1 and its weight gradient, gating GEMM + routing as two launches, RoPE and
    its inverse as passes of their own (round 4's step); ON = the default.  The switches are read per call.  Medians of `pairs` steps."""
    import statistics

    t = {"off": [], "on": []}
    prev = {k: os.environ.get(k) for k in FUSION_SWITCHES}
    try:
        for i in range(pairs + 1):                     # (first pair = warm-up of both paths)
            for arm in ("off", "on"):
                for k in FUSION_SWITCHES:
                    if arm == "off":
                        os.environ[k] = "0"
                    else:
                        os.environ.pop(k, None)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                step()
                torch.cuda.synchronize()
                if i:
                    t[arm].append((time.perf_counter() - t0) * 1e3)
    finally:
        for k, v in prev.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    off, on = statistics.median(t["off"]), statistics.median(t["on"])
    return {"off_ms_per_step": round(off, 1), "on_ms_per_step": round(on, 1), "gain_ms": round(off - on, 1), "pairs": pairs,
            "off_runs_ms": [round(v, 1) for v in t["off"]], "on_runs_ms": [round(v, 1) for v in t["on"]],
            "switches_off_arm": {k: "0" for k in FUSION_SWITCHES},
            "note": "interleaved in ONE process: off = K2 in the training step, K1 and both RoPE fusions switched off (round 4's launches), on = "
                    "the default path; not switchable and therefore in BOTH arms: packed q|k|v / gate|up weights, the row-level image merge, "
                    "the device-scalar loss scaling"}


def lora_config_record(model, cfg, step, ops, B, S, steps=3):
    """recipes/config_lora.yaml on the SAME model and micro-batch (the reference's cheapest recipe: 1 GPU): LoRA r = 8, alpha = 32, dropout
    0.05 on fc1 / fc2 / q,k,v,o_proj / gate,up,down_proj / lm_head of the language model (aria/train.py:100-112, aria/lora/layers.py:129-139),
    ViT and projector frozen, every base weight frozen.  Timed: 1 warm-up + `steps` steps of forward + backward (a) with the adapters through
    the fused node (their second projection inside the base launches), (b) the frozen-base reference -- the un-adapted model with only the
    embedding trainable, i.e. forward + every input gradient and NO weight gradient: the floor a LoRA step cannot go below -- and (c) the
    adapters with the recipe's gradient checkpointing.  The adapters are unwrapped afterwards (B factors are drawn N(0, 0.02) for the timing --
    peft's zero init would put exact zeros through the extension tile -- so nothing is merged)."""
    from aria_amd.lora import GroupedGemmLoraLayer, LinearLoraLayer, apply_lora_from_config

    lm = model.language_model
    req = {n: p.requires_grad for n, p in model.named_parameters()}

    def timed(n=steps):
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    # (b) frozen base: forward + input gradients only
    for n, p in model.named_parameters():
        p.requires_grad_(n.endswith("embed_tokens.weight"))
    t_floor = timed()
    # (a) the recipe's adapters
    recipe = dict(use_peft=True, lora_r=8, lora_alpha=32, lora_dropout=0.05, freeze_vit=True, freeze_projector=True,
                  lora_target_modules=["fc1", "fc2", "q_proj", "k_proj", "v_proj", "linear", "o_proj", "up_proj", "down_proj", "out_proj", "gate_proj",
                                       "lm_head"])
    apply_lora_from_config(model, recipe)
    g = torch.Generator(device="cuda").manual_seed(5)
    adapters = [(n, m) for n, m in model.named_modules() if isinstance(m, (GroupedGemmLoraLayer, LinearLoraLayer))]
    with torch.no_grad():
        for _, m in adapters:
            m.lora_B.weight.normal_(0.0, 0.02, generator=g)
    ev = []
    orig = ops.grouped_gemm_swiglu_lora

    def timed_fc1(*a, **k):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = orig(*a, **k)
        e.record()
        ev.append((s, e))
        return r

    ops.grouped_gemm_swiglu_lora = timed_fc1
    try:
        t_lora = timed()
        durs = [s.elapsed_time(e) * 1e-3 for s, e in ev[len(ev) // (steps + 1):]]   # (the warm-up step's launches dropped)
        ops.grouped_gemm_swiglu_lora = orig
        cfg.gradient_checkpointing = True
        t_ckpt = timed()
    finally:
        ops.grouped_gemm_swiglu_lora = orig
        cfg.gradient_checkpointing = False
        for name, layer in adapters:   # unwrap (nothing merged: the base weights were never touched)
            parent = model.get_submodule(name.rsplit(".", 1)[0]) if "." in name else model
            setattr(parent, name.rsplit(".", 1)[-1], layer.base_layer)
        for n, p in model.named_parameters():
            p.requires_grad_(req.get(n, True))
        model.zero_grad(set_to_none=True)
    M = B * S * cfg.moe_topk
    flops = 2.0 * M * (cfg.hidden_size + 8) * (2 * cfg.moe_intermediate_size)
    avg = sum(durs) / max(1, len(durs))
    n_ad = len(adapters)
    del adapters
    return {"workload": f"recipes/config_lora.yaml on the config #3 micro-batch ({B} x {S}, 2 x 980px images per sample), 1 GPU: LoRA r=8 alpha=32 dropout=0.05 on "
                        f"{n_ad} modules of the {cfg.num_hidden_layers}-layer language model (fc1, fc2, q/k/v/o_proj, gate/up/down_proj, lm_head), ViT + projector + every base "
                        "weight frozen; forward + backward; gradient checkpointing OFF in `ms_per_step` (`recipe_grad_checkpointing_ms` = the recipe as written)",
            "ms_per_step": round(t_lora * 1e3, 1), "value": round(B * S / t_lora, 1), "unit": "tokens/s", "steps": steps, "warmup": 1,
            "frozen_base_fwd_dgrad_ms": round(t_floor * 1e3, 1), "over_frozen_base": round(t_lora / t_floor, 4),
            "adapter_path": "module-by-module (ARIA_LORA_FUSED=0)" if os.environ.get("ARIA_LORA_FUSED", "1") == "0" else
                            "fused node: adapters' second projection inside the base launches (K-extension)",
            "recipe_grad_checkpointing_ms": round(t_ckpt * 1e3, 1),
            "roofline": {"kernel": "gemm3_kernel<rc,oc,3>+swiglu+K-extension grouped-M (experts.fc1 + LoRA second projection + SwiGLU, one launch)",
                         "bound": "mfma", "achieved": round(flops / avg / 1e12, 1) if durs else None, "peak": 2500.0, "unit": "TFLOP/s",
                         "frac": round(flops / avg / 2.5e15, 4) if durs else None, "launches_timed": len(durs), "avg_launch_ms": round(avg * 1e3, 4),
                         "algorithmic_flops_per_launch": flops, "traffic": None}}


def self_launch(args):
    """`python bench.py --gpus N` (no torchrun around it): re-exec this script under torch.distributed.run with N ranks on
    127.0.0.1, one per GPU, so that the plain form measures N GPUs instead of silently measuring one."""
    import socket

    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"bench.py: --gpus {args.gpus} without a launcher -> re-exec under torch.distributed.run ({args.gpus} ranks, port {port})",
          file=sys.stderr, flush=True)
    os.execv(sys.executable, cmd)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)  # does not return
    # stdout carries the contract line and NOTHING else: RCCL (version banner, flushed from libc's buffer when the process exits, i.e.
    # AFTER the JSON line) and gloo ("Rank 0 is connected ...") write to fd 1 themselves.  The JSON goes to a private duplicate of the
    # original stdout; fd 1 itself points at stderr from here on.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or args.ep:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:  # --ep on one GPU: a one-rank RCCL group, so that the expert-parallel code path (all-to-all dispatch / combine on RCCL,
            os.environ.setdefault("MASTER_PORT", "29531")   # _ep_local shards) runs on hardware even without a second GPU
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        else:
            dist.init_process_group("nccl", device_id=dev)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the line would not measure what it names")
    if world > 1:
        n_seen = dist.get_world_size()
        if rank == 0:
            print(f"bench.py: RCCL process group up, {n_seen} ranks, device {torch.cuda.get_device_name(local)}", file=sys.stderr, flush=True)
        if n_seen != args.gpus:
            raise SystemExit(f"bench.py: process group has {n_seen} ranks, --gpus {args.gpus}")

    from aria_amd import hip, ops
    from aria_amd.modeling_aria import AriaConfig, AriaForConditionalGeneration
    from aria_amd.moe_lm import AriaMoELMConfig
    from aria_amd.parallel import GradSync
    from aria_amd.vision import AriaVisionConfig

    cfg_kwargs = dict(hidden_size=2560, num_hidden_layers=args.layers, num_attention_heads=20, vocab_size=100352,
                      moe_intermediate_size=1664, moe_num_experts=64, moe_topk=6, moe_num_shared_experts=2,
                      rms_norm_eps=1e-6, rope_theta=5_000_000.0, moe_z_loss_coeff=1e-5, moe_aux_loss_coeff=1e-3)  # aria/model/moe_lm.py:57-58 defaults
    cfg = AriaMoELMConfig(**cfg_kwargs, gradient_checkpointing=args.recompute)
    IMG_TOKEN, QTOK = 9, 256                       # gptfast/model.py:54; 980px image -> 4900 patches -> 256 query tokens
    acfg = AriaConfig(vision_config=AriaVisionConfig(num_hidden_layers=args.vit_layers), text_config=cfg,
                      projector_patch_to_query_dict={1225: 128, 4900: 256}, image_token_index=IMG_TOKEN)
    torch.set_default_device(dev)
    model = AriaForConditionalGeneration(acfg)
    torch.set_default_device("cpu")
    init_params(model, seed=0)  # same weights on every rank (DP replicas)
    model.train()
    model.freeze_vit()          # recipes/config_full.yaml:39-42: ViT frozen, projector + LLM trainable
    if args.ep:
        model.enable_expert_parallel()  # every rank built the same 64 experts (same seed) and keeps its 64 / N
    sync = GradSync(model, overlap=not args.no_overlap, mode="all_reduce" if args.allreduce else "reduce_scatter") if world > 1 else None

    B, S, V = args.batch, args.seq, cfg.vocab_size
    n_img = args.images

    def make_inputs(B, S, n_img, seed):
        g = torch.Generator(device="cuda").manual_seed(seed)
        ids = torch.randint(10, V, (B, S), generator=g, device=dev)
        pixel_values = pixel_mask = None
        if n_img > 0:
            for j in range(n_img):                      # contiguous runs of 256 image placeholders per image
                ids[:, 16 + j * (QTOK + 28): 16 + j * (QTOK + 28) + QTOK] = IMG_TOKEN
            pixel_values = torch.randn((B * n_img, 3, 980, 980), generator=g, device=dev).clamp_(-1, 1).to(bf16)
            pixel_mask = torch.ones((B * n_img, 980, 980), dtype=torch.bool, device=dev)
            pixel_mask[0, 735:, :] = False              # one image with its bottom 25 % rows padded (mask path exercised)
        labels = ids.clone()
        labels[:, : int(0.75 * S)] = -100  # prompt masked like an SFT sample
        return dict(input_ids=ids, pixel_values=pixel_values, pixel_mask=pixel_mask, labels=labels)

    batch = make_inputs(B, S, n_img, 1234 + rank)

    # live timing of the dominant kernel: fc1 grouped expert GEMM forward launches (HIP events on the launch stream)
    fc1_events = []
    orig_gg = ops.grouped_gemm

    other_events = {}  # --time-grouped: every other grouped-M launch class (diagnostics only, printed to stderr)

    def timed_grouped_gemm(a, w, offsets, *, w_is_kn=True, out=None):
        if args.time_grouped and timed_grouped_gemm.on and not (w_is_kn and w.shape[1] == cfg.hidden_size):
            key = ("fwd" if w_is_kn else "dgrad") + f"_K{a.shape[1]}"
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = orig_gg(a, w, offsets, w_is_kn=w_is_kn, out=out)
            e.record()
            other_events.setdefault(key, []).append((s, e))
            return r
        if w_is_kn and w.shape[1] == cfg.hidden_size and fc1_events is not None and timed_grouped_gemm.on:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = orig_gg(a, w, offsets, w_is_kn=w_is_kn, out=out)
            e.record()
            fc1_events.append((s, e))
            timed_grouped_gemm.variant = hip.get_lib().cdll.aria_last_gemm_variant()  # which kernel family the library dispatched to
            return r
        return orig_gg(a, w, offsets, w_is_kn=w_is_kn, out=out)

    timed_grouped_gemm.on = False
    timed_grouped_gemm.variant = 0
    ops.grouped_gemm = timed_grouped_gemm

    # the default path launches experts.fc1 fused with its SwiGLU epilogue (one launch: same GEMM, epilogue included in its time)
    orig_ggs = ops.grouped_gemm_swiglu

    def timed_grouped_gemm_swiglu(a, w, offsets, want_h=True):
        if not timed_grouped_gemm.on or fc1_events is None:
            return orig_ggs(a, w, offsets, want_h=want_h)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = orig_ggs(a, w, offsets, want_h=want_h)
        e.record()
        fc1_events.append((s, e))
        timed_grouped_gemm.variant = hip.get_lib().cdll.aria_last_gemm_variant()
        timed_grouped_gemm.fused = True
        return r

    timed_grouped_gemm.fused = False
    timed_grouped_gemm.gather = False
    ops.grouped_gemm_swiglu = timed_grouped_gemm_swiglu

    # ... since r05 on the UN-permuted tokens through the dispatcher's index (K2 in the training step too): the same launch, gathered A rows
    orig_ggsg = ops.grouped_gemm_swiglu_gather

    def timed_grouped_gemm_swiglu_gather(x, rows, w, offsets, want_h=False):
        if not timed_grouped_gemm.on or fc1_events is None:
            return orig_ggsg(x, rows, w, offsets, want_h=want_h)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = orig_ggsg(x, rows, w, offsets, want_h=want_h)
        e.record()
        fc1_events.append((s, e))
        timed_grouped_gemm.variant = hip.get_lib().cdll.aria_last_gemm_variant()
        timed_grouped_gemm.fused = timed_grouped_gemm.gather = True
        return r

    ops.grouped_gemm_swiglu_gather = timed_grouped_gemm_swiglu_gather

    dense_events = {}  # --time-grouped also classifies the dense GEMMs by operand form and size class (diagnostics)
    orig_gemm = ops.gemm

    def timed_gemm(a, b, **kw):
        if not (args.time_grouped and timed_grouped_gemm.on):
            return orig_gemm(a, b, **kw)
        a_oc, b_oc = bool(kw.get("a_oc")), bool(kw.get("b_oc"))
        M, K = (a.shape[1], a.shape[0]) if a_oc else (a.shape[0], a.shape[1])
        N = b.shape[1] if b_oc else b.shape[0]
        key = f"{'oc' if a_oc else 'rc'},{'oc' if b_oc else 'rc'} M{M} N{N} K{K}"
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = orig_gemm(a, b, **kw)
        e.record()
        dense_events.setdefault(key, []).append((s, e))
        return r

    ops.gemm = timed_gemm

    def step():
        model.zero_grad(set_to_none=True)
        out = model(**batch, return_logits=False)
        out.loss.backward()
        if sync is not None:
            sync.finish()
        return out.loss

    def barrier():
        if world > 1:
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = step()
    barrier()
    timed_grouped_gemm.on = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    timed_grouped_gemm.on = False
    if world > 1:
        import torch.distributed as dist

        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    tokens = world * B * S * args.steps
    if rank == 0:
        M = B * S * cfg.moe_topk
        flops_launch = 2.0 * M * cfg.hidden_size * (2 * cfg.moe_intermediate_size)
        durs = [s.elapsed_time(e) * 1e-3 for s, e in fc1_events]
        if args.time_grouped:
            import sys as _sys

            diag = {k: round(sum(s.elapsed_time(e) for s, e in v) / len(v), 4) for k, v in other_events.items()}
            diag["fwd_fc1"] = round(sum(durs) / max(1, len(durs)) * 1e3, 4)
            print("grouped-M avg launch ms: " + json.dumps(diag), file=_sys.stderr, flush=True)
            dd = {k: [len(v) // args.steps, round(sum(s.elapsed_time(e) for s, e in v) / args.steps, 3)] for k, v in dense_events.items()}
            dd = dict(sorted(dd.items(), key=lambda kv: -kv[1][1])[:14])
            print("dense GEMM classes [launches/step, ms/step]: " + json.dumps(dd), file=_sys.stderr, flush=True)
        avg = sum(durs) / max(1, len(durs))
        achieved = flops_launch / avg / 1e12 if durs else None
        peak = 2500.0
        res = {
            "metric": "tokens/sec (fwd+bwd) Aria-25.3B bf16", "value": round(tokens / dt, 2), "unit": "tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": ("north_star target shape: Aria-25.3B random-init, per GPU ONE 65 536-token sequence " if args.long else
                                    "config#3 per-GPU shape (recipes/config_full.yaml): Aria-25.3B random-init, per GPU 8 samples x ") +
                                   f"({n_img} x 980px images + text) padded to S={S}; frozen 27-layer ViT fwd (4900 patches/img) -> "
                                   "trainable projector (256 tok/img) -> 28-layer MoE decoder (64 experts top-6, D=2560, V=100352) "
                                   "fwd+bwd incl. lm_head+CE (labelled rows) and router aux-loss grads; image-token count check of modeling_aria.py:265-271 ON; "
                                   "recipe gradient checkpointing: " +
                                   ("ON" if args.recompute else "OFF (288 GB holds every activation; the recipe-as-written number is `recipe_grad_checkpointing`)"),
                       "layers": args.layers, "vit_layers": args.vit_layers, "images_per_sample": n_img, "global_batch": world * B, "seq_len": S,
                       "parallelism": (f"dp{world}+ep{world}" if args.ep else f"dp{world}") if world > 1 else ("single+ep1" if args.ep else "single"),
                       "grad_exchange": None if world == 1 else ("all_reduce" if args.allreduce else "reduce_scatter (ZeRO-2)"), "grad_checkpointing": bool(args.recompute),
                       "optimizer_in_step": False,  # metric = fwd+bwd; AdamW state (299 GB fp32) only exists sharded over >= 2 GPUs
                       "loss": round(float(loss), 4)},
            "roofline": {"kernel": fc1_kernel_tag(timed_grouped_gemm.variant, timed_grouped_gemm.gather) + " grouped-M (experts.fc1 forward" +
                                   (" + SwiGLU epilogue)" if timed_grouped_gemm.fused else ")"), "bound": "mfma",
                         "achieved": None if achieved is None else round(achieved, 1), "peak": peak, "unit": "TFLOP/s",
                         "frac": None if achieved is None else round(achieved / peak, 4),
                         "traffic": pmc_traffic(timed_grouped_gemm.variant, timed_grouped_gemm.gather),
                         "launches_timed": len(durs), "avg_launch_ms": round(avg * 1e3, 4),
                         "algorithmic_flops_per_launch": flops_launch},
        }
        # whole-step fraction: EXECUTED flops (frozen ViT forward only, lm_head + CE on the labelled rows only) / ms_per_step / peak
        try:
            labelled = int((batch["labels"][:, 1:] != -100).sum())
            fl = executed_flops_per_step(cfg, B, S, labelled, B * n_img, vit_layers=args.vit_layers)
            res["roofline"]["step_executed_tflop"] = round(fl["total"] / 1e12, 1)
            res["roofline"]["step_frac"] = round(fl["total"] / (dt / args.steps) / (peak * 1e12), 4)
            res["roofline"]["step_flops_breakdown_tflop"] = {k: round(v / 1e12, 1) for k, v in fl.items() if k != "total"}
        except Exception as ex:  # noqa: BLE001
            res["roofline"]["step_frac_error"] = f"{type(ex).__name__}: {ex}"[:200]
        try:
            if world == 1 and not args.no_launch_classes:
                ops.gemm, ops.grouped_gemm, ops.grouped_gemm_swiglu, ops.grouped_gemm_swiglu_gather = orig_gemm, orig_gg, orig_ggs, orig_ggsg
                lc = launch_classes_record(step, ops, peak)
                res["roofline"]["worst_large_launch"] = lc.pop("worst")
                res["launch_classes"] = lc
        except Exception as ex:  # noqa: BLE001
            res["launch_classes"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
        if args.layers != 28 or args.vit_layers != 27 or (n_img != 2 and not args.long):
            res["config"]["INVALID"] = "reduced depth / no images (debug run)"
        full_depth = args.layers == 28 and args.vit_layers == 27
        # sub-records (N = 1 only).  Whatever happens inside them, the contract line above is printed: a failure is recorded, not raised.
        try:
            if world == 1 and not args.long and not args.recompute and not args.no_long64k:
                # the recipe's own setting (recipes/config_full.yaml:17 gradient_checkpointing: true) on the SAME micro-batch, three timed steps:
                # the headline keeps every activation (288 GB holds them); this is the number for a reader who wants the recipe as written
                cfg.gradient_checkpointing = True
                step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    step()
                torch.cuda.synchronize()
                dtr = (time.perf_counter() - t0) / 3
                cfg.gradient_checkpointing = False
                res["recipe_grad_checkpointing"] = {"ms_per_step": round(dtr * 1e3, 1), "value": round(B * S / dtr, 1), "unit": "tokens/s", "steps": 3,
                                                    "warmup": 1, "recompute_level": getattr(model.language_model.model, "last_recompute_level", None),
                                                    "note": "config #3 step with recipes/config_full.yaml:17 gradient_checkpointing on (per decoder layer; "
                                                            "level 'moe': token-sized tensors kept, the four expert-row tensors rebuilt in the backward; "
                                                            "level 'layer': layer inputs + the flash kernel's (o, lse) kept, the layer re-run)"}
            if world == 1 and not args.long and not args.no_long64k and (full_depth or args.long64k_seq != 65536):
                res["long64k"] = long64k_record(model, cfg, make_inputs, ops, steps=3, warmup=1, S=args.long64k_seq, n_img=args.long64k_images)
                if not full_depth or args.long64k_seq != 65536:
                    res["long64k"]["INVALID"] = "debug run (reduced depth / sequence)"
        except Exception as ex:  # noqa: BLE001
            res["sub_records_error"] = f"{type(ex).__name__}: {ex}"[:400]
            cfg.gradient_checkpointing = bool(args.recompute)
        try:
            if world == 1 and not args.long and not args.ep and not args.recompute and not args.no_fusions_ab:
                res["step_fusions_ab"] = step_fusions_ab(step, pairs=args.sub_record_repeats or 4)
                if not full_depth or args.sub_record_repeats:
                    res["step_fusions_ab"]["INVALID"] = "debug run (reduced depth / repetitions)"
        except Exception as ex:  # noqa: BLE001
            res["step_fusions_ab"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
        try:
            if world == 1 and not args.long and not args.ep and not args.recompute and not args.no_lora_record:
                res["lora_config"] = lora_config_record(model, cfg, step, ops, B, S, steps=args.sub_record_repeats or 3)
                if not full_depth or args.sub_record_repeats:
                    res["lora_config"]["INVALID"] = "debug run (reduced depth)"
        except Exception as ex:  # noqa: BLE001
            res["lora_record_error"] = f"{type(ex).__name__}: {ex}"[:400]
        try:
            if world == 1 and not args.long and not args.ep and not args.no_inference_records:
                # BASELINE configs #2 (generate) and #4 (long prefill) on the gptfast surface of the SAME weights (the reference's own
                # checkpoint conversion, on the device), so that every published inference number is timed by whoever runs this file
                model.zero_grad(set_to_none=True)
                batch = None
                torch.cuda.empty_cache()
                model.eval()
                t0 = time.perf_counter()
                twin = model.to_gptfast()
                torch.cuda.synchronize()
                t_conv = time.perf_counter() - t0
                debug = not full_depth or args.gen_new != 200 or args.gen_image != 1 or args.prefill_seq != 53248 or args.prefill_frames != 32 or args.sub_record_repeats
                img_px = model.config.vision_config.image_size
                rep = dict(runs=args.sub_record_repeats, warmup=1) if args.sub_record_repeats else {}
                res["generate_config2"] = generate_config2_record(twin, cfg, new_tokens=args.gen_new, n_img=args.gen_image, img_px=img_px,
                                                                  qtok=acfg.projector_patch_to_query_dict.get((img_px // 14) ** 2, QTOK), img_token=IMG_TOKEN,
                                                                  **rep)
                res["generate_config2"]["hf_to_gptfast_s"] = round(t_conv, 2)
                res["prefill_config4"] = prefill_config4_record(twin, cfg, S=args.prefill_seq, frames=args.prefill_frames, img_token=IMG_TOKEN)
                if debug:
                    res["generate_config2"]["INVALID"] = res["prefill_config4"]["INVALID"] = "debug run (reduced depth / lengths)"
                del twin
        except Exception as ex:  # noqa: BLE001
            res["inference_records_error"] = f"{type(ex).__name__}: {ex}"[:400]
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(cfg_kwargs)
            except Exception as ex:  # keep the bench line even if the host is too small for the sample
                res["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port",
                                       "sample": f"failed: {type(ex).__name__}: {ex}"}
        print(json.dumps(res), file=json_out, flush=True)
    json_out.close()
    if world > 1 or args.ep:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
