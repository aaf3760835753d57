"""A few full-width decoder layers with recipes/config_lora.yaml's adapters, three steps -- the target of a rocprofv3 kernel trace
(tools/gpu_session.sh ... "py=..." or rocprofv3 --kernel-trace --stats -- python tools/lora_profile.py): which kernels the LoRA step adds on
top of the frozen-base forward + input gradients.  Prints ms per step with the adapters and for the frozen base."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aria_amd.lora import apply_lora_from_config  # noqa: E402
from aria_amd.moe_lm import AriaMoELMConfig, AriaMoELMForCausalLM  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B, S, V = 8, 2048, 100352
dev = torch.device("cuda")
cfg = AriaMoELMConfig(hidden_size=2560, num_hidden_layers=L, num_attention_heads=20, vocab_size=V, moe_intermediate_size=1664, moe_num_experts=64,
                      moe_topk=6, moe_num_shared_experts=2)
with torch.device(dev):
    lm = AriaMoELMForCausalLM(cfg)
g = torch.Generator(device="cuda").manual_seed(0)
with torch.no_grad():
    for n, p in lm.named_parameters():
        p.fill_(1.0) if "norm" in n else p.normal_(0.0, 0.02, generator=g)
ids = torch.randint(10, V, (B, S), generator=g, device=dev)
labels = ids.clone()
labels[:, : int(0.75 * S)] = -100


def step():
    lm.zero_grad(set_to_none=True)
    lm(input_ids=ids, labels=labels, return_logits=False).loss.backward()


def timed(n=3):
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for n, p in lm.named_parameters():
    p.requires_grad_(n.endswith("embed_tokens.weight"))
floor = timed()
apply_lora_from_config(lm, dict(lora_r=8, lora_alpha=32, lora_dropout=0.05,
                                lora_target_modules=["fc1", "fc2", "q_proj", "k_proj", "v_proj", "o_proj", "up_proj", "down_proj", "gate_proj", "lm_head"]))
with torch.no_grad():
    for n, p in lm.named_parameters():
        if "lora_B" in n:
            p.normal_(0.0, 0.02, generator=g)
lm.train()
print(json.dumps({"layers": L, "frozen_base_ms": round(floor, 2), "lora_ms": round(timed(), 2)}))
