import os, sys, torch
sys.path.insert(0, os.getcwd())
from tests import model_cases as M
from aria_amd.lora import apply_lora_from_config
from aria_amd.moe_lm import AriaMoELMForCausalLM
bf16 = torch.bfloat16
dev = "cuda"
d = dict(hidden_size=128, num_attention_heads=2, num_key_value_heads=2, num_hidden_layers=2, vocab_size=96, moe_intermediate_size=128,
         moe_num_experts=8, moe_topk=2, moe_num_shared_experts=2)
def run(ckpt):
    torch.manual_seed(3)
    lm = AriaMoELMForCausalLM(M.make_cfg(d, gradient_checkpointing=ckpt))
    with torch.no_grad():
        for n, p in lm.named_parameters():
            p.copy_(torch.ones(p.shape) if "norm" in n else (torch.randn(p.shape) * 0.08).to(bf16))
    apply_lora_from_config(lm, dict(lora_r=8, lora_alpha=32, lora_dropout=0.0,
                                    lora_target_modules=["fc1", "fc2", "q_proj", "k_proj", "v_proj", "o_proj", "up_proj", "down_proj", "gate_proj"]))
    with torch.no_grad():
        for n, p in lm.named_parameters():
            if "lora_B" in n:
                p.copy_((torch.randn(p.shape) * 0.05).to(bf16))
    lm = lm.to(dev).train()
    ids = torch.randint(1, 96, (2, 33), generator=torch.Generator().manual_seed(1))
    out = lm(input_ids=ids.to(dev), labels=ids.to(dev))
    out.loss.backward()
    return float(out.loss.detach()), {n: p.grad.detach().float().cpu() for n, p in lm.named_parameters() if p.grad is not None}
a = run(False); b = run(False); c = run(True); e = run(True)
print("loss", a[0], b[0], c[0], e[0])
for name, (x, y) in (("stored vs stored", (a, b)), ("stored vs ckpt", (a, c)), ("ckpt vs ckpt", (c, e))):
    bad = [(n, float((x[1][n] - y[1][n]).abs().max()), float(x[1][n].abs().max())) for n in x[1] if not torch.equal(x[1][n], y[1][n])]
    print(name, len(bad), bad[:6])
