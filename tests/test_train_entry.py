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


def test_toy_finetune_reduces_loss(tmp_path):
    from aria_amd.train import main

    hist = main(["--tiny", "per_device_train_batch_size=2", "gradient_accumulation_steps=1", "max_seq_length=24", "max_steps=4",
                 "learning_rate=1e-2", "weight_decay=0.0", "warmup_ratio=0.0", "images_per_sample=1", "logging_steps=100",
                 "synthetic_fixed=true", "save_final=true", f"output_dir={tmp_path}"])
    assert len(hist) == 4 and hist[-1] < hist[0] - 0.2, hist
    # trainer.save_model(output_dir) (aria/train.py:247-249): an HF checkpoint directory the model class loads back
    from aria_amd.modeling_aria import AriaForConditionalGeneration

    assert (tmp_path / "config.json").exists() and (tmp_path / "model.safetensors.index.json").exists()
    again = AriaForConditionalGeneration.from_pretrained(str(tmp_path))
    assert sum(p.numel() for p in again.parameters()) > 0


def test_lora_recipe_trains_only_the_adapters(tmp_path):
    """recipes/config_lora.yaml keys (use_peft, lora_r, lora_alpha, lora_dropout, the recipe's full lora_target_modules list): adapters on
    experts.fc1 / fc2 (grouped) and on the LM's Linear projections + lm_head, everything else frozen, and a few steps still reduce the
    loss of a fixed batch."""
    from aria_amd.train import main

    hist = main(["--tiny", "per_device_train_batch_size=2", "gradient_accumulation_steps=1", "max_seq_length=24", "max_steps=3",
                 "learning_rate=1e-2", "weight_decay=0.0", "warmup_ratio=0.0", "images_per_sample=1", "logging_steps=100",
                 "synthetic_fixed=true", "use_peft=true", "lora_r=8", "lora_alpha=32", "lora_dropout=0.05", "freeze_projector=true",
                 'lora_target_modules=["fc1","fc2","q_proj","k_proj","v_proj","linear","o_proj","up_proj","down_proj","out_proj",'
                 '"gate_proj","lm_head"]', "save_final=true", f"output_dir={tmp_path}"])
    assert len(hist) == 3 and hist[-1] < hist[0], hist
    import json

    from safetensors.torch import load_file

    adapter = load_file(str(tmp_path / "adapter_model.safetensors"))
    acfg = json.load(open(tmp_path / "adapter_config.json"))
    assert acfg["r"] == 8 and acfg["lora_alpha"] == 32 and "lm_head" in acfg["target_modules"]
    assert adapter and all(".lora_A." in k or ".lora_B." in k for k in adapter)
    assert any(k.endswith("lm_head.lora_B.weight") for k in adapter) and any("experts.fc1.lora_A" in k for k in adapter)
    assert float(max(v.float().abs().max() for k, v in adapter.items() if ".lora_B." in k)) > 0.0  # B left its zero init: it trained
    # the adapter goes back onto a fresh base model
    from aria_amd.lora import load_lora_adapter, lora_state_dict
    from aria_amd.train import build_model, load_config

    fresh, _ = build_model(load_config(["--tiny"]), torch.device("cpu"))
    names = load_lora_adapter(fresh, str(tmp_path))
    assert any(n.endswith("lm_head") for n in names) and len(names) == len(adapter) // 2
    got = lora_state_dict(fresh)
    assert all(torch.equal(got[k].cpu(), adapter[k]) for k in adapter)


def test_checkpoint_resume_continues_the_same_run(tmp_path):
    """save_strategy / resume_from_checkpoint (recipes/config_full.yaml:20-21): weights + sharded optimizer state + step + data position --
    an interrupted run resumed from its checkpoint reproduces the uninterrupted loss history exactly."""
    import json

    from aria_amd.train import latest_checkpoint, main

    common = ["--tiny", "per_device_train_batch_size=1", "gradient_accumulation_steps=2", "max_seq_length=24", "learning_rate=1e-2",
              "weight_decay=0.1", "warmup_ratio=0.25", "images_per_sample=1", "logging_steps=100"]
    straight = main(common + ["max_steps=3", "save_strategy=steps", "save_steps=2", f"output_dir={tmp_path / 'b'}"])
    ck = latest_checkpoint(str(tmp_path / "b"))
    assert ck.endswith("checkpoint-2") and json.load(open(f"{ck}/trainer_state.json"))["global_step"] == 2
    assert {"config.json", "model.safetensors.index.json", "optimizer_rank0.pt", "trainer_state.json"} <= set(__import__("os").listdir(ck))
    resumed = main(common + ["max_steps=3", "resume_from_checkpoint=true", f"output_dir={tmp_path / 'b'}"])   # runs step 3 only
    assert len(straight) == 3 and resumed == straight, (resumed, straight)
