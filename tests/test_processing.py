"""Data-side callers of the hot path, CPU only, bit-exact.

* AriaVisionProcessor: against tests/golden/processing.json (written by oracle/ref_processing.py from the reference's own class) and,
  when /root/reference is present, tensor-for-tensor against the reference itself.
* AriaProcessor: the strings the reference's own tests assert (tests/test_aria_processor.py:41-187), with a stub tokenizer in place of
  the unreachable hub tokenizer.
* apply_chat_template_and_tokenize / collate_fn: against the reference's functions on the same stub tokenizer."""
import copy
import json
import os

import pytest
import torch

from aria_amd import processing as P
from oracle import ref_shims
from oracle.ref_processing import StubTokenizer, digest, test_images as make_images

HERE = os.path.dirname(os.path.abspath(__file__))
needs_reference = pytest.mark.skipif(not ref_shims.reference_available(), reason="/root/reference not present")

MESSAGES = [{"role": "user", "content": [{"text": None, "type": "image"}, {"text": "describe the image", "type": "text"}]}]
PROMPT = "<|im_start|>user\n<fim_prefix><|img|><fim_suffix>describe the image<|im_end|>\n<|im_start|>assistant\n"


@pytest.fixture
def processor():
    return P.AriaProcessor(tokenizer=StubTokenizer(), image_processor=P.AriaVisionProcessor(max_image_size=490), image_token="<|img|>")


@pytest.fixture
def sample_image():
    return make_images()[0]  # 768 x 768, like the reference's tests


@pytest.mark.parametrize("size,split", [(490, False), (490, True), (980, False)])
def test_vision_processor_matches_the_reference_fixture(size, split):
    want = json.load(open(os.path.join(HERE, "golden", "processing.json")))[f"{size}_{int(split)}"]
    r = P.AriaVisionProcessor(max_image_size=490)(make_images(), max_image_size=size, split_image=split)
    pv, pm = r["pixel_values"], r["pixel_mask"]
    assert list(pv.shape) == want["pixel_values_shape"] and pv.dtype == torch.float32 and pm.dtype == torch.bool
    assert r["num_crops"].tolist() == want["num_crops"]
    assert int(pm.sum()) == want["mask_true"]
    assert [float(pv[i % pv.shape[0], i % 3, (37 * i) % size, (91 * i) % size]) for i in range(8)] == want["sample"]
    assert digest(pv) == want["pixel_values_sha256"] and digest(pm) == want["pixel_mask_sha256"]


@needs_reference
def test_vision_processor_equals_the_reference_tensor_for_tensor():
    from oracle.ref_processing import load_reference_processing

    ref = load_reference_processing().vp.AriaVisionProcessor(max_image_size=490)
    mine = P.AriaVisionProcessor(max_image_size=490)
    for kw in (dict(max_image_size=490, split_image=True), dict(max_image_size=None, min_image_size=None)):
        a, b = ref(make_images(), **kw), mine(make_images(), **kw)
        for k in ("pixel_values", "pixel_mask", "num_crops"):
            assert torch.equal(a[k], b[k]), (k, kw)
    for fn in ("_select_best_resolution",):
        for w, h in ((768, 768), (1000, 300), (300, 1000), (10, 10), (5000, 700)):
            assert getattr(load_reference_processing().vp, fn)(w, h, P.DEFAULT_SPLIT_RATIO, 490) == P.select_best_resolution(
                w, h, P.DEFAULT_SPLIT_RATIO, 490)


@needs_reference
@pytest.mark.parametrize("w,h", [(1, 1), (17, 335), (337, 336), (491, 50), (981, 600), (3000, 3)])
def test_vision_processor_equals_the_reference_on_odd_sizes(w, h):
    """degenerate and off-by-one sizes around the 336 / 490 / 980 thresholds (a wider sweep, 461 size x mode combinations up to 3000 px,
    was run once offline: no mismatch)"""
    import numpy as np
    from PIL import Image

    from oracle.ref_processing import load_reference_processing

    img = Image.fromarray(np.random.default_rng(w * 7 + h).integers(0, 255, (h, w, 3), dtype=np.uint8))
    ref_cls = load_reference_processing().vp.AriaVisionProcessor
    for size, split in ((490, False), (490, True), (980, False)):
        a = ref_cls(max_image_size=size)([img], max_image_size=size, split_image=split)
        b = P.AriaVisionProcessor(max_image_size=size)([img], max_image_size=size, split_image=split)
        for k in ("pixel_values", "pixel_mask", "num_crops"):
            assert torch.equal(a[k], b[k]), (k, size, split)


def test_invalid_max_image_size_raises(processor, sample_image):
    with pytest.raises(ValueError):
        processor(text=PROMPT, images=[sample_image], return_tensors="pt", max_image_size=1000)


# ---- the reference's own string KATs (tests/test_aria_processor.py)
def test_apply_chat_template(processor):
    assert processor.apply_chat_template(MESSAGES, add_generation_prompt=True) == PROMPT
    assert processor.apply_chat_template(MESSAGES, add_generation_prompt=False) == PROMPT[: -len("<|im_start|>assistant\n")]


def test_chat_template_with_multiple_messages(processor):
    messages = [
        {"role": "user", "content": [{"text": None, "type": "image"}, {"text": "What's in this image?", "type": "text"}]},
        {"role": "assistant", "content": "This is a beautiful landscape."},
        {"role": "user", "content": [{"text": "Can you describe it in more detail?", "type": "text"}]},
    ]
    assert processor.apply_chat_template(messages, add_generation_prompt=True) == (
        "<|im_start|>user\n<fim_prefix><|img|><fim_suffix>What's in this image?<|im_end|>\n<|im_start|>assistant\nThis is a beautiful "
        "landscape.<|im_end|>\n<|im_start|>user\nCan you describe it in more detail?<|im_end|>\n<|im_start|>assistant\n")


@pytest.mark.parametrize("size,n_tok", [(980, 256), (490, 128)])
def test_end_to_end_processing(processor, sample_image, size, n_tok):
    text = processor.apply_chat_template(MESSAGES, add_generation_prompt=True)
    inputs, prompts = processor(text=text, images=[sample_image], return_tensors="pt", max_image_size=size, return_final_prompts=True)
    assert {"input_ids", "attention_mask", "pixel_values", "pixel_mask"} <= set(inputs)
    assert inputs["input_ids"].dim() == 2 and inputs["attention_mask"].dim() == 2 and inputs["pixel_values"].dim() == 4
    assert inputs["input_ids"].device.type == "cpu" and inputs["pixel_values"].dtype == torch.float32
    assert prompts[0] == PROMPT.replace("<|img|>", "<|img|>" * n_tok)
    assert int((inputs["input_ids"] == StubTokenizer.SPECIAL.index("<|img|>")).sum()) == n_tok  # what the model's scatter expects


def test_multiple_images_in_conversation(processor, sample_image):
    messages = [{"role": "user", "content": [{"text": None, "type": "image"}, {"text": None, "type": "image"},
                                              {"text": "Compare the two images.", "type": "text"}]}]
    text = processor.apply_chat_template(messages, add_generation_prompt=True)
    inputs, prompts = processor(text=text, images=[sample_image, sample_image], return_tensors="pt", max_image_size=980,
                                return_final_prompts=True)
    assert inputs["pixel_values"].shape[0] == 2
    want = ("<|im_start|>user\n<fim_prefix><|img|><fim_suffix><fim_prefix><|img|><fim_suffix>Compare the two images.<|im_end|>\n"
            "<|im_start|>assistant\n")
    assert prompts[0] == want.replace("<|img|>", "<|img|>" * 256)


def test_split_image(processor, sample_image):
    text = processor.apply_chat_template(MESSAGES, add_generation_prompt=True)
    inputs, prompts = processor(text=text, images=[sample_image], return_tensors="pt", max_image_size=490, split_image=True,
                                return_final_prompts=True)
    assert inputs["pixel_values"].shape == (5, 3, 490, 490) and inputs["pixel_mask"].shape == (5, 490, 490)
    want = "<|im_start|>user\n<fim_prefix><|img|><|img|><|img|><|img|><|img|><fim_suffix>describe the image<|im_end|>\n<|im_start|>assistant\n"
    assert prompts[0] == want.replace("<|img|>", "<|img|>" * 128)


# ---- ChatML ids + label masking, collate
CONVERSATIONS = [
    [{"role": "user", "content": [{"text": None, "type": "image"}, {"text": "What is this?", "type": "text"}]},
     {"role": "assistant", "content": [{"text": "A cat.", "type": "text"}]},
     {"role": "user", "content": [{"text": "Sure?", "type": "text"}]},
     {"role": "assistant", "content": [{"text": "Yes, a cat on a mat.", "type": "text"}]}],
    [{"role": "user", "content": [{"text": "hi", "type": "text"}]}, {"role": "assistant", "content": [{"text": "hello", "type": "text"}]}],
]


def test_label_masking_properties():
    tok = StubTokenizer()
    tok.pad_token = tok.unk_token
    out = P.apply_chat_template_and_tokenize(copy.deepcopy(CONVERSATIONS), tok, iter([torch.tensor(2)]), max_length=4096, max_image_size=490)
    ids, labels, mask = out["input_ids"], out["labels"], out["attention_mask"]
    assert ids.shape == labels.shape == mask.shape and ids.dtype == torch.long
    assert int((ids[0] == 2).sum()) == 2 * 128                       # 2 crops x 128 image tokens
    assert bool(((labels == -100) | (labels == ids)).all())          # a label is either ignored or the token itself
    assert bool((labels[~mask] == -100).all())                       # padding is never a target
    seen = tok.encode("A cat.") + tok.encode("<|im_end|>") + tok.encode("\n")
    row = labels[0][labels[0] != -100].tolist()
    assert row[: len(seen)] == seen                                   # the first supervised span is the first assistant answer + <|im_end|>\n
    assert 2 not in row and tok.encode("What is this?")[0] not in row[:1]
    cut = P.apply_chat_template_and_tokenize(copy.deepcopy(CONVERSATIONS), tok, iter([torch.tensor(2)]), max_length=50, max_image_size=490)
    assert cut["input_ids"].shape[1] == 50 and torch.equal(cut["input_ids"], ids[:, :50])
    with pytest.raises(ValueError):
        P.apply_chat_template_and_tokenize(copy.deepcopy(CONVERSATIONS), tok, iter([torch.tensor(1)]), max_image_size=500)


@needs_reference
def test_chatml_and_collate_equal_the_reference():
    from oracle.ref_processing import load_reference_processing

    ns = load_reference_processing()
    tok = StubTokenizer()
    tok.pad_token = tok.unk_token
    for size, length in ((490, 4096), (980, 4096), (490, 300)):
        a = ns.data.apply_chat_template_and_tokenize(copy.deepcopy(CONVERSATIONS), tok, iter([torch.tensor(3)]), max_length=length,
                                                     max_image_size=size)
        b = P.apply_chat_template_and_tokenize(copy.deepcopy(CONVERSATIONS), tok, iter([torch.tensor(3)]), max_length=length,
                                               max_image_size=size)
        assert set(a) == set(b) and all(torch.equal(a[k], b[k]) for k in a), (size, length)
    # collate: same examples through the reference's image processor + its ChatML function vs the product's collate_fn
    imgs = make_images()[:2]
    examples = [{"images": [imgs[0]], "messages": CONVERSATIONS[0], "video": None}, {"images": None, "messages": CONVERSATIONS[1], "video": None}]
    refp = ns.vp.AriaVisionProcessor(max_image_size=490)
    image_inputs = refp([imgs[0]], split_image=True)
    want = ns.data.apply_chat_template_and_tokenize(copy.deepcopy([e["messages"] for e in examples]), tok,
                                                    iter(image_inputs.pop("num_crops")), max_length=100000, max_image_size=490)
    want.update(image_inputs)
    want["pixel_values"] = want["pixel_values"].to(torch.bfloat16)
    got = P.collate_fn(copy.deepcopy(examples), tok, P.AriaVisionProcessor(max_image_size=490), split_image=True, max_seq_length=100000)
    assert set(got) == set(want)
    for k in want:
        assert torch.equal(got[k], want[k]), k


def test_processor_output_feeds_the_model_end_to_end():
    """AriaProcessor -> AriaForConditionalGeneration (tiny dims, kernels through the SIMT emulator): a 490-px image becomes 1225 patches,
    the projector maps them to 128 image tokens and the prompt carries exactly 128 image placeholders -- the model's own token/feature
    count check (modeling_aria.py:265-271) passes and the logits are finite."""
    from tests.emu import emu_lib

    from aria_amd.modeling_aria import AriaConfig, AriaForConditionalGeneration

    emu_lib.install()
    try:
        tok = StubTokenizer()
        proc = P.AriaProcessor(tokenizer=tok, image_processor=P.AriaVisionProcessor(max_image_size=490), image_token="<|img|>")
        text = proc.apply_chat_template(MESSAGES, add_generation_prompt=True)
        inputs = proc(text=text, images=[make_images()[3]], return_tensors="pt", max_image_size=490)  # the 40 x 30 image: mostly padding
        img_id = StubTokenizer.SPECIAL.index("<|img|>")
        cfg = AriaConfig(vision_config=dict(hidden_size=64, num_hidden_layers=1, num_attention_heads=1, intermediate_size=64, image_size=490),
                         text_config=dict(hidden_size=64, num_hidden_layers=1, num_attention_heads=1, vocab_size=512, moe_intermediate_size=16,
                                          moe_num_experts=8, moe_topk=2, max_position_embeddings=512),
                         projector_patch_to_query_dict={1225: 128}, image_token_index=img_id)
        model = AriaForConditionalGeneration(cfg).eval()
        with torch.no_grad():
            for n, p in model.named_parameters():
                p.copy_((torch.ones(p.shape) if ("norm" in n or "ln_" in n) and n.endswith("weight") else torch.randn(p.shape) * 0.05).to(p.dtype))
            out = model(input_ids=inputs["input_ids"], pixel_values=inputs["pixel_values"].to(torch.bfloat16), pixel_mask=inputs["pixel_mask"],
                        attention_mask=inputs["attention_mask"], return_logits=True)
        assert int((inputs["input_ids"] == img_id).sum()) == 128
        assert out.logits.shape == (1, inputs["input_ids"].shape[1], 512) and bool(torch.isfinite(out.logits.float()).all())
    finally:
        emu_lib.uninstall()


def test_processor_save_and_load_round_trip(tmp_path):
    """processing_aria.py:216-275: preprocessor_config.json next to the tokenizer files; a directory without either gives the default
    image processor and (like the reference) no tokenizer."""
    vp = P.AriaVisionProcessor(max_image_size=490, min_image_size=300, image_mean=(0.4, 0.5, 0.6), image_std=(0.2, 0.3, 0.4))
    saved = []

    class Tk(StubTokenizer):
        def save_pretrained(self, d):
            saved.append(d)

    P.AriaProcessor(image_processor=vp, tokenizer=Tk()).save_pretrained(str(tmp_path))
    assert saved == [str(tmp_path)] and (tmp_path / "preprocessor_config.json").exists()
    back = P.AriaVisionProcessor.from_pretrained(str(tmp_path))
    assert back.to_dict() == vp.to_dict()
    img = make_images()[3]
    a, b = vp([img]), back([img], max_image_size=None)
    assert torch.equal(a["pixel_values"], vp([img], max_image_size=980)["pixel_values"]) and b["pixel_values"].shape[-1] == 490
    with pytest.warns(UserWarning):
        proc = P.AriaProcessor.from_pretrained(str(tmp_path))          # no tokenizer files there
    assert proc.tokenizer is None and proc.image_processor.max_image_size == 490
    assert P.AriaVisionProcessor.from_pretrained(str(tmp_path / "empty")).to_dict() == P.AriaVisionProcessor().to_dict()
