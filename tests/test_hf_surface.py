"""Seam B3 as Hugging Face sees it (aria_amd/hf.py): PreTrainedModel / GenerationMixin subclasses registered with the Auto classes under
the reference's model types -- from_pretrained / save_pretrained through HF's own machinery, the reference's checkpoint key names, HF
generate() (cache-free) agreeing with the native decode engine, and two optimizer steps under transformers.Trainer (the reference's
fine-tune recipe runs trl.SFTTrainer = a Trainer subclass; aria/train.py:231-246).  CPU: kernels through the SIMT emulator."""
import os
import tempfile

import pytest
import torch

from tests.emu import emu_lib

bf16 = torch.bfloat16


@pytest.fixture(autouse=True)
def _emu():
    emu_lib.install()
    yield
    emu_lib.uninstall()


def tiny():
    from aria_amd import hf

    cfg = hf.AriaHFConfig(vision_config=dict(hidden_size=64, num_hidden_layers=1, num_attention_heads=1, intermediate_size=128, image_size=56),
                          text_config=dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, vocab_size=512, moe_intermediate_size=64,
                                           moe_num_experts=8, moe_topk=2),
                          projector_patch_to_query_dict={16: 4}, image_token_index=9)
    torch.manual_seed(0)
    return hf, hf.AriaForConditionalGeneration(cfg)


def batch():
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(10, 512, (1, 12), generator=g)
    ids[0, 2:6] = 9
    return ids, torch.randn((1, 3, 56, 56), generator=g).to(bf16)


def test_pretrained_model_surface_and_auto_round_trip():
    from transformers import AutoConfig, AutoModelForCausalLM, GenerationMixin, PreTrainedModel

    hf, m = tiny()
    assert isinstance(m, PreTrainedModel) and isinstance(m, GenerationMixin)
    assert set(m._no_split_modules) == {"MoEDecoderLayer", "VisionEncoderLayer"} and m.supports_gradient_checkpointing
    assert m.get_input_embeddings() is m.language_model.model.embed_tokens and m.get_output_embeddings() is m.language_model.lm_head
    keys = set(m.state_dict())
    assert {"language_model.model.layers.0.mlp.experts.fc1.weight", "language_model.model.layers.1.mlp.router.weight",
            "language_model.lm_head.weight", "multi_modal_projector.query", "vision_tower.vision_model.embeddings.patch_embedding.weight"} <= keys
    ids, pv = batch()
    out = m(input_ids=ids, pixel_values=pv, labels=ids)
    assert out.loss.ndim == 0 and torch.isfinite(out.loss)
    out.loss.backward()
    assert m.language_model.model.layers[0].mlp.experts.fc1.weight.grad is not None
    m.freeze_vit()
    assert not any(p.requires_grad for p in m.vision_tower.parameters())
    m.gradient_checkpointing_enable()
    assert m.is_gradient_checkpointing
    m.gradient_checkpointing_disable()
    with tempfile.TemporaryDirectory() as d:
        m.save_pretrained(d)
        assert {"config.json", "model.safetensors"} <= set(os.listdir(d))
        cfg = AutoConfig.from_pretrained(d)
        assert type(cfg) is hf.AriaHFConfig and cfg.text_config.moe_num_experts == 8 and cfg.projector_patch_to_query_dict == {16: 4}
        m2 = AutoModelForCausalLM.from_pretrained(d)          # resolves "aria" to the native-kernel class
        assert type(m2) is hf.AriaForConditionalGeneration and m2.dtype == bf16
        sd1, sd2 = m.state_dict(), m2.state_dict()
        assert set(sd1) == set(sd2) and all(torch.equal(sd1[k], sd2[k]) for k in sd1)
        from aria_amd.modeling_aria import AriaForConditionalGeneration as Native

        n = Native.from_pretrained(d)                          # ... and the non-HF class reads the HF-written directory
        assert all(torch.equal(sd1[k], n.state_dict()[k]) for k in sd1)
        m2.eval(), n.eval()
        with torch.no_grad():
            a = m2(input_ids=ids, pixel_values=pv).logits
            b = n(input_ids=ids, pixel_values=pv).logits
        assert torch.equal(a, b)


def test_hf_generate_matches_native_greedy_decode():
    _, m = tiny()
    m.eval()
    ids, pv = batch()
    got = m.generate(input_ids=ids, pixel_values=pv, max_new_tokens=5, do_sample=False)       # GenerationMixin loop, no cache
    want = m.generate_fast(input_ids=ids, pixel_values=pv, max_new_tokens=5, do_sample=False)  # gptfast twin + decode engine
    assert got.shape == (1, 17) and torch.equal(got[:, :12], ids)
    assert torch.equal(got.cpu(), want.cpu())


def test_two_steps_under_transformers_trainer():
    transformers = pytest.importorskip("transformers")
    pytest.importorskip("accelerate")
    from transformers import Trainer, TrainingArguments

    _, m = tiny()
    m.freeze_vit()
    ids, pv = batch()

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return 4

        def __getitem__(self, i):
            return {"input_ids": ids[0], "labels": ids[0], "pixel_values": pv[0]}

    def collate(rows):
        return {"input_ids": torch.stack([r["input_ids"] for r in rows]), "labels": torch.stack([r["labels"] for r in rows]),
                "pixel_values": torch.stack([r["pixel_values"] for r in rows])}

    before = m.language_model.lm_head.weight.detach().clone()
    with tempfile.TemporaryDirectory() as d:
        args = TrainingArguments(output_dir=d, per_device_train_batch_size=1, max_steps=2, learning_rate=1e-2, report_to=[], use_cpu=True,
                                 save_strategy="no", logging_steps=1, remove_unused_columns=False, bf16=False, dataloader_pin_memory=False)
        tr = Trainer(model=m, args=args, train_dataset=DS(), data_collator=collate)
        res = tr.train()
    assert res.global_step == 2 and torch.isfinite(torch.tensor(res.training_loss))
    assert not torch.equal(before, m.language_model.lm_head.weight.detach())
