"""aria_amd.inference (mirror of aria/inference.py) on CPU through the emulator: a tiny checkpoint directory + a LoRA adapter directory +
an image file -> load_model (adapter folded in) -> chat template / processor -> sampled tokens -> decoded text."""
import json

import numpy as np
import pytest
import torch

from oracle.ref_processing import StubTokenizer


class Tok(StubTokenizer):
    def decode(self, ids, skip_special_tokens=True):
        out, buf = "", bytearray()
        for i in ids:
            if i < len(self.SPECIAL):
                out += buf.decode("utf-8", "replace") + self.SPECIAL[i]
                buf = bytearray()
            elif 16 <= i < 272:
                buf.append(i - 16)
        return out + buf.decode("utf-8", "replace")


@pytest.fixture(scope="module", autouse=True)
def emu():
    from tests.emu import emu_lib

    emu_lib.install()
    yield
    emu_lib.uninstall()


def test_inference_script_pieces(tmp_path):
    from PIL import Image

    from aria_amd import inference as I
    from aria_amd import processing as P
    from aria_amd.lora import apply_lora_from_config, lora_state_dict
    from aria_amd.modeling_aria import AriaConfig, AriaForConditionalGeneration
    from aria_amd.train import save_output

    args = I.parse_arguments(["--base_model_path", "b", "--tokenizer_path", "t", "--image_path", "i.png", "--prompt", "hi", "--split_image"])
    assert args.max_image_size == 980 and args.split_image and args.peft_model_path is None   # aria/inference.py:30-52 flags / defaults
    tok = Tok()
    img_id, end_id = Tok.SPECIAL.index("<|img|>"), Tok.SPECIAL.index("<|im_end|>")
    cfg = AriaConfig(vision_config=dict(hidden_size=64, num_hidden_layers=1, num_attention_heads=1, intermediate_size=64, image_size=490),
                     text_config=dict(hidden_size=64, num_hidden_layers=2, num_attention_heads=1, vocab_size=512, moe_intermediate_size=16,
                                      moe_num_experts=8, moe_topk=2, max_position_embeddings=512),
                     projector_patch_to_query_dict={1225: 128}, image_token_index=img_id)
    torch.manual_seed(0)
    base = AriaForConditionalGeneration(cfg)
    with torch.no_grad():
        for n, p in base.named_parameters():
            p.copy_((torch.ones(p.shape) if ("norm" in n or "ln_" in n) and n.endswith("weight") else torch.randn(p.shape) * 0.05).to(p.dtype))
    base_dir, peft_dir = tmp_path / "base", tmp_path / "adapter"
    base.save_pretrained(str(base_dir))
    lcfg = dict(use_peft=True, lora_r=8, lora_alpha=32, lora_dropout=0.0, freeze_vit=True, freeze_projector=True,
                lora_target_modules=["fc1", "fc2", "q_proj", "lm_head"], output_dir=str(peft_dir), model_name_or_path=str(base_dir))
    apply_lora_from_config(base, lcfg)
    with torch.no_grad():
        for n, p in base.named_parameters():
            if "lora_B" in n:
                p.copy_((torch.randn(p.shape) * 0.3).to(p.dtype))
    factors = {k: v.clone() for k, v in lora_state_dict(base).items()}
    save_output(base, lcfg)
    assert json.load(open(peft_dir / "adapter_config.json"))["base_model_name_or_path"] == str(base_dir)

    plain = I.load_model(str(base_dir), device="cpu")
    tuned = I.load_model(str(base_dir), str(peft_dir), device="cpu")
    assert not any(".lora_" in k for k in tuned.state_dict())                  # folded in: plain modules, reference key names
    w0 = plain.state_dict()["language_model.lm_head.weight"].float()
    w1 = tuned.state_dict()["language_model.lm_head.weight"].float()
    delta = factors["language_model.lm_head.lora_B.weight"].float() @ factors["language_model.lm_head.lora_A.weight"].float() * 4.0
    assert float((w1 - w0 - delta).abs().max()) <= 2e-2 * float(delta.abs().max()) + 1e-3

    rng = np.random.default_rng(1)
    Image.fromarray(rng.integers(0, 255, (60, 90, 3), dtype=np.uint8)).save(tmp_path / "img.png")
    proc = P.AriaProcessor(tokenizer=tok, image_processor=P.AriaVisionProcessor(max_image_size=490), image_token="<|img|>")
    inputs = I.prepare_input(str(tmp_path / "img.png"), "what is this?", proc, 490, False)
    assert int((inputs["input_ids"] == img_id).sum()) == 128 and inputs["pixel_values"].shape[-1] == 490
    torch.manual_seed(3)
    text = I.inference(str(tmp_path / "img.png"), "what is this?", tuned, proc, max_image_size=490, max_new_tokens=5)
    assert isinstance(text, str) and "<|im_end|>" not in text


def test_gptfast_generator_chat_and_benchmark(tmp_path):
    """gptfast/generate.py Generator / chat.py AriaChat / benchmark.py run_benchmark surface on a tiny model.pth (emulator)."""
    from PIL import Image

    from aria_amd import gptfast as G
    from aria_amd import gptfast_generate as GG
    from aria_amd import processing as P
    from aria_amd.vision import AriaVisionConfig

    tok = Tok()
    img_id, end_id = Tok.SPECIAL.index("<|img|>"), Tok.SPECIAL.index("<|im_end|>")
    args = G.ModelArgs(block_size=1024, vocab_size=512, n_layer=2, n_head=1, dim=64, intermediate_size=16, n_local_heads=1, head_dim=64,
                       rope_base=10000.0, norm_eps=1e-5, num_experts=8, router_topk=2, num_shared_experts=2, image_token_index=img_id)
    vc = AriaVisionConfig(hidden_size=64, num_hidden_layers=1, num_attention_heads=1, intermediate_size=64, image_size=490)
    torch.manual_seed(0)
    model = G.Aria(args, vc, {1225: 128})
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.copy_((torch.ones(p.shape) if ("norm" in n or "ln_" in n) and n.endswith("weight") else torch.randn(p.shape) * 0.05).to(p.dtype))
    torch.save(model.state_dict(), tmp_path / "model.pth")

    class Proc490(P.AriaProcessor):  # the Generator calls the processor with its default image size (980 -> 4900 patches): too slow for the emulator
        def __call__(self, *a, **k):
            k.setdefault("max_image_size", 490)
            return super().__call__(*a, **k)

    proc = Proc490(tokenizer=tok, image_processor=P.AriaVisionProcessor(max_image_size=490), image_token="<|img|>")
    # loading path of generate.py:187-222 (tiny ModelArgs instead of the 25 B defaults)
    torch.manual_seed(1)
    twin = G.Aria(args, vc, {1225: 128})
    G.load_model_pth(twin, torch.load(tmp_path / "model.pth"), strict=True)
    assert all(torch.equal(a, b) for a, b in zip(model.state_dict().values(), twin.state_dict().values()))

    mc = GG.ModelConfig(checkpoint_path=tmp_path / "model.pth", device="cpu", compile=True)      # compile flags: accepted, ignored
    gc = GG.GenerationConfig(max_new_tokens=2, top_k=5, temperature=0.8, cache_size=600, stop_strings=["<|im_end|>", "\n\n"])
    gen = GG.Generator(mc, gc, model=twin.eval(), processor=proc)
    assert gen._stops() == ([end_id], ["\n\n"])
    rng = np.random.default_rng(2)
    path = tmp_path / "cat.png"
    Image.fromarray(rng.integers(0, 255, (50, 70, 3), dtype=np.uint8)).save(path)
    messages = [{"role": "user", "content": [{"text": None, "type": "image"}, {"text": "describe the image", "type": "text"}]}]
    torch.manual_seed(5)
    text_only = [{"role": "user", "content": [{"text": "count to three", "type": "text"}]}]
    new = gen.generate(text_only, None, detokenize=False)               # (the image path runs once, in the chat turn below: 7 s of emulated ViT)
    assert new.dim() == 1 and 1 <= new.numel() <= 2                      # generated part only, like generate.py:174
    assert twin.llm.max_seq_length >= 600                               # cache_size pre-sized the static KV cache
    with pytest.raises(ValueError):
        GG.Generator(mc, GG.GenerationConfig(max_new_tokens=600, cache_size=100), model=twin, processor=proc).generate(messages, Image.open(path))

    chat = GG.AriaChat(mc, gc, generator=gen)
    first = chat.chat("hello")                                          # text-only turn (no ViT), then a turn that brings an image
    second = chat.chat("what is in the picture?", str(path))
    assert [m.role for m in chat.history] == ["user", "assistant", "user", "assistant"] and chat.history[2].image_path == str(path)
    assert isinstance(first, str) and isinstance(second, str) and "<|im_end|>" not in first + second
    msgs, imgs = chat.format_prompt()
    assert len(imgs) == 1 and msgs[2]["content"][0] == {"text": None, "type": "image"} and msgs[0]["content"] == [{"text": "hello", "type": "text"}]
    chat.reset()
    assert chat.history == []

    res = GG.run_benchmark(gen, text_only, None, num_runs=1, warmup=0)
    assert set(res) == {"mean_latency", "std_latency", "mean_tokens", "std_tokens", "tokens_per_second"} and res["tokens_per_second"] > 0
