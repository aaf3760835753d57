"""The LIVE reference (aria/model from /root/reference through oracle/ref_shims.py) timed on this host's cores, next to the oracle port
(VERDICT r4 missing #5 / next #9).  Build-container only: /root/reference does not exist on the GPU box, so bench.py's `cpu_baseline`
times the port there; this file is what ties the port's number to the reference's.

  (1) BASELINE config #1 -- "Aria-Base-8K random-init, 1 image (490px) + 128 text tokens, CPU float32 forward via reference aria/model":
      AriaForConditionalGeneration at full widths, 27-layer ViT, L = 1 and L = 5 decoder layers (28 layers of fp32 weights are 99.6 GB;
      this container has 62 GB: SURVEY F10), forward, 1 warm-up + 5 timed, best; T(28) = T(1) + 27 (T(5) - T(1)) / 4, labelled as an
      extrapolation.  The port (O.aria_forward) on the same weights beside it.
  (2) the port's own shape (bench.py cpu_baseline): ONE full-width decoder layer inside a 1-layer LM, fwd + bwd, B = 1, S = 2048, reference
      vs port, same weights.

    python tools/cpu_reference_timing.py > profiles/r05_cpu_reference_config1.json"""
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import aria_oracle as O  # noqa: E402
from oracle.ref_shims import load_reference  # noqa: E402

TEXT = dict(hidden_size=2560, num_attention_heads=20, num_key_value_heads=20, vocab_size=100352, intermediate_size=1664 * 2,
            moe_intermediate_size=1664, moe_num_experts=64, moe_topk=6, moe_num_shared_experts=2, rms_norm_eps=1e-6, rope_theta=5_000_000.0,
            max_position_embeddings=8192, pad_token_id=0)
VISION = dict(hidden_size=1152, num_attention_heads=16, num_hidden_layers=27, intermediate_size=4304, patch_size=14, image_size=490,
              num_channels=3, layer_norm_eps=1e-6, hidden_act="gelu_pytorch_tanh")
IMG, Q = 9, 128


def init(model, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("norm.weight") or ("layer_norm" in n or "ln_" in n) and n.endswith("weight"):
                p.fill_(1.0)
            else:
                p.normal_(0.0, 0.02, generator=g)


def timed(fn, runs=5):
    fn()
    ts = []
    for _ in range(runs):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts), ts   # best of the timed runs: the first ones page freshly initialised fp32 weights in (up to 10x slower)


def config1(ns, L):
    acfg = ns.cfg.AriaConfig(vision_config={**VISION, "model_type": "aria_vision_model"},
                             text_config={**TEXT, "num_hidden_layers": L, "model_type": "aria_moe_lm"},
                             projector_patch_to_query_dict={1225: Q, 4900: 256}, image_token_index=IMG, attn_implementation="eager", pad_token_id=0)
    model = ns.mdl.AriaForConditionalGeneration(acfg).eval()
    init(model, 0)
    g = torch.Generator().manual_seed(1)
    S = 14 + Q + 128
    ids = torch.randint(10, TEXT["vocab_size"], (1, S), generator=g)
    ids[0, 7:7 + Q] = IMG
    pv = torch.randn(1, 3, 490, 490, generator=g).clamp_(-1, 1)
    pm = torch.ones(1, 490, 490, dtype=torch.bool)
    am = torch.ones(1, S, dtype=torch.long)

    def ref():
        with torch.no_grad():
            return model(input_ids=ids, pixel_values=pv, pixel_mask=pm, attention_mask=am).logits

    t_ref, all_ref = timed(ref)
    w = {k: v.detach() for k, v in model.state_dict().items()}
    tc = O.LMConfig(**{k: v for k, v in {**TEXT, "num_hidden_layers": L}.items() if k in O.LMConfig.__dataclass_fields__})
    vc = O.VisionConfig(**{k: v for k, v in VISION.items() if k in O.VisionConfig.__dataclass_fields__})
    ocfg = O.AriaOracleConfig(text=tc, vision=vc, patch_to_query={1225: Q, 4900: 256}, projector_heads=16, image_token_index=IMG)

    def port():
        with torch.no_grad():
            return O.aria_forward(ids, pv, pm, am, None, w, ocfg)[0]

    t_port, all_port = timed(port)
    err = float((port() - ref()).abs().max())
    return S, t_ref, all_ref, t_port, all_port, err


def layer_shape(ns, S=2048, V=512):
    text = {**TEXT, "num_hidden_layers": 1, "vocab_size": V, "moe_z_loss_coeff": 1e-5, "moe_aux_loss_coeff": 1e-3}
    cfg = ns.moe.AriaMoELMConfig(**text, attn_implementation="eager")
    lm = ns.moe.AriaMoELMForCausalLM(cfg).train()
    init(lm, 2)
    ids = torch.randint(0, V, (1, S), generator=torch.Generator().manual_seed(3))

    def ref():
        lm.zero_grad(set_to_none=True)
        lg = lm(input_ids=ids).logits
        torch.nn.functional.cross_entropy(lg[:, :-1].reshape(-1, V), ids[:, 1:].reshape(-1)).backward()

    t_ref, all_ref = timed(ref)
    w = {k: v.detach().clone().requires_grad_(True) for k, v in lm.state_dict().items()}
    ocfg = O.LMConfig(**{k: v for k, v in text.items() if k in O.LMConfig.__dataclass_fields__})

    def port():
        for v in w.values():
            v.grad = None
        lg = O.lm_forward(w["model.embed_tokens.weight"][ids], w, ocfg, training=True)
        torch.nn.functional.cross_entropy(lg[:, :-1].reshape(-1, V), ids[:, 1:].reshape(-1)).backward()

    t_port, all_port = timed(port)
    return t_ref, all_ref, t_port, all_port


def main():
    ns = load_reference()
    torch.manual_seed(0)
    out = {"what": "LIVE reference (aria/model of /root/reference, 5 import shims: oracle/ref_shims.py; grouped_gemm absent -> the reference's own "
                   "sequential_gemm fallback, eager attention) on this container's host cores, fp32, 1 warm-up + 5 timed, best of 5; "
                   "the oracle port (oracle/aria_oracle.py) on the same weights and inputs beside it",
           "host": {"os.cpu_count": os.cpu_count(), "torch.get_num_threads": torch.get_num_threads(), "torch": torch.__version__}}
    S, r1, ar1, p1, ap1, e1 = config1(ns, 1)
    _, r2, ar2, p2, ap2, e2 = config1(ns, 5)
    ext_ref, ext_port = r1 + 27 * (r2 - r1) / 4, p1 + 27 * (p2 - p1) / 4
    out["config1"] = {"workload": f"BASELINE config #1: 1 x 490px image (1225 patches -> {Q} tokens) + 128 text tokens + 14 template = {S} positions, "
                                  "forward, 27-layer ViT + projector + L decoder layers at full width + lm_head (V = 100352)",
                      "reference_s": {"L=1": round(r1, 3), "L=5": round(r2, 3), "runs_L1": [round(t, 3) for t in ar1], "runs_L5": [round(t, 3) for t in ar2]},
                      "port_s": {"L=1": round(p1, 3), "L=5": round(p2, 3), "runs_L1": [round(t, 3) for t in ap1], "runs_L5": [round(t, 3) for t in ap2]},
                      "max_abs_logit_difference_port_vs_reference": {"L=1": e1, "L=5": e2},
                      "extrapolated_28_layers": {"note": "T(28) = T(1) + 27 (T(5) - T(1)) / 4: 28 layers of fp32 weights (99.6 GB) do not fit this host (SURVEY F10)",
                                                 "reference_s": round(ext_ref, 2), "reference_tokens_per_s": round(S / ext_ref, 3),
                                                 "port_s": round(ext_port, 2), "port_tokens_per_s": round(S / ext_port, 3)},
                      "port_over_reference_time": round(ext_port / ext_ref, 3)}
    lr, alr, lp, alp = layer_shape(ns)
    out["port_shape"] = {"workload": "bench.py cpu_baseline's sample: ONE full-width decoder layer inside a 1-layer LM (V = 512), fwd + bwd with the "
                                     "router's aux losses, B = 1, S = 2048",
                         "reference_s": round(lr, 3), "reference_runs": [round(t, 3) for t in alr], "port_s": round(lp, 3),
                         "port_runs": [round(t, 3) for t in alp], "port_over_reference_time": round(lp / lr, 3)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
