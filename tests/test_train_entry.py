"""aria_amd.train entry (mirror of aria/train.py) on CPU through the emulator: a few optimizer steps on a toy model reduce the loss,
the recipe YAML is parsed, and the fused AdamW kernel matches torch.optim.AdamW on fp32 master weights."""
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def emu():
    from tests.emu import emu_lib

    emu_lib.install()
    yield
    emu_lib.uninstall()


def test_adamw_kernel_matches_torch():
    from aria_amd import ops

    torch.manual_seed(0)
    n = 1000
    w0 = torch.randn(n)
    p = w0.to(torch.bfloat16)
    master, m, v = p.float().clone(), torch.zeros(n), torch.zeros(n)
    ref = torch.nn.Parameter(p.float().clone())
    opt = torch.optim.AdamW([ref], lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1)
    for step in range(1, 4):
        g = torch.randn(n).to(torch.bfloat16)
        ops.adamw_step_(p, g, master, m, v, lr=1e-2, beta1=0.9, beta2=0.95, eps=1e-8, weight_decay=0.1, step=step)
        ref.grad = g.float()
        opt.step()
        assert torch.allclose(master, ref.detach(), atol=1e-5, rtol=1e-5)
        assert torch.equal(p, master.to(torch.bfloat16))


def test_recipe_yaml_is_honoured():
    from aria_amd.train import load_config

    ref = "/root/reference/recipes/config_full.yaml"
    if not os.path.exists(ref):
        pytest.skip("reference recipes not present")
    cfg = load_config(["--config", ref, "max_steps=3", "--learning_rate", "1e-4"])
    assert cfg["per_device_train_batch_size"] == 8 and cfg["gradient_accumulation_steps"] == 2 and cfg["max_seq_length"] == 2048
    assert cfg["freeze_vit"] is True and cfg["freeze_projector"] is False and cfg["gradient_checkpointing"] is True
    assert cfg["max_steps"] == 3 and cfg["learning_rate"] == 1e-4 and cfg["adam_beta2"] == 0.95


def test_toy_finetune_reduces_loss():
    from aria_amd.train import main

    hist = main(["--tiny", "per_device_train_batch_size=2", "gradient_accumulation_steps=1", "max_seq_length=24", "max_steps=6",
                 "learning_rate=1e-2", "weight_decay=0.0", "warmup_ratio=0.0", "images_per_sample=1", "logging_steps=100",
                 "synthetic_fixed=true"])
    assert len(hist) == 6 and hist[-1] < hist[0] - 0.3, hist


def test_lora_recipe_trains_only_the_expert_adapters():
    """recipes/config_lora.yaml keys (use_peft, lora_r, lora_alpha, lora_target_modules): adapters on experts.fc1 / fc2, everything
    else frozen, and a few steps still reduce the loss of a fixed batch."""
    from aria_amd.train import main

    hist = main(["--tiny", "per_device_train_batch_size=2", "gradient_accumulation_steps=1", "max_seq_length=24", "max_steps=4",
                 "learning_rate=3e-2", "weight_decay=0.0", "warmup_ratio=0.0", "images_per_sample=1", "logging_steps=100",
                 "synthetic_fixed=true", "use_peft=true", "lora_r=8", "lora_alpha=32", "freeze_projector=true",
                 'lora_target_modules=["fc1","fc2","q_proj"]'])
    assert len(hist) == 4 and hist[-1] < hist[0], hist
