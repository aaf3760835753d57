"""The immediate CALLERS of the hot path: what turns images + chat messages into the tensors
``AriaForConditionalGeneration.forward`` consumes.  Torch + PIL only (the reference needs torchvision, which this image lacks).

Mirrors, with the same names, argument meaning and error behaviour:
  * ``AriaVisionProcessor``                aria/model/vision_processor.py:29-283  -> pixel_values [N,3,S,S] f32, pixel_mask [N,S,S] bool,
                                                                                    num_crops [n_images]
  * ``AriaProcessor``                      aria/model/processing_aria.py:40-205   -> image-token expansion + tokenizer call
  * ``apply_chat_template_and_tokenize``   aria/data.py:29-120                    -> ChatML ids + labels (user turns masked)
  * ``collate_fn``                         aria/train.py:117-209
The chat template itself lives in the hub tokenizer's config (not in the reference repo); ``AriaProcessor.apply_chat_template`` restates
it so that the strings of the reference's own tests (tests/test_aria_processor.py:41-82) come out character for character.

This is CPU data plumbing: nothing here touches the GPU library, and parity is bit-exact (integer / string work, and PIL does the
resampling in both implementations).
"""
from __future__ import annotations

import re
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np
import torch
from PIL import Image, ImageOps

DEFAULT_SPLIT_RATIO = [[1, 2], [1, 3], [1, 4], [1, 5], [1, 6], [1, 7], [1, 8], [2, 4], [2, 3], [2, 2], [2, 1], [3, 1], [3, 2],
                       [4, 1], [4, 2], [5, 1], [6, 1], [7, 1], [8, 1]]
IGNORE_TOKEN_ID = -100


def select_best_resolution(img_width: int, img_height: int, target_ratios: Sequence[Sequence[int]], patch_size: int):
    """vision_processor.py:29-61: the (w, h) tiling whose aspect ratio is closest; on ties the later one wins if the image has more
    than half the tiling's pixels."""
    aspect_ratio = img_width / img_height
    best_diff = float("inf")
    best_w, best_h = 1, 1
    area = int(img_width) * int(img_height)
    for rw, rh in target_ratios:
        diff = abs(aspect_ratio - rw / rh)
        if diff < best_diff:
            best_diff, best_w, best_h = diff, rw, rh
        elif diff == best_diff and area > 0.5 * patch_size * patch_size * rw * rh:
            best_w, best_h = rw, rh
    return best_w, best_h


def split_image(image: Image.Image, split: bool, split_ratio, patch_size: int) -> List[Image.Image]:
    """vision_processor.py:64-105: resize to the best tiling, cut row-major patch_size tiles, and put the whole image first (unless
    the tiling is 1 x 1)."""
    if not split:
        return [image]
    rw, rh = select_best_resolution(image.width, image.height, split_ratio, patch_size)
    resized = image.resize((patch_size * rw, patch_size * rh))
    tiles = []
    for i in range(rw * rh):
        x, y = (i % rw) * patch_size, (i // rw) * patch_size
        tiles.append(resized.crop((x, y, x + patch_size, y + patch_size)))
    if len(tiles) != 1:
        tiles.insert(0, image)
    return tiles


def keep_ratio_resize_and_pixel_mask(img: Image.Image, max_size: int, min_size: int = 336, padding_value: int = 0):
    """vision_processor.py:108-151: long side -> max_size (bicubic), short side at least min_size, pad right/bottom; the mask is True
    on real pixels."""
    img = img.convert("RGB")
    scale = max_size / max(img.size)
    w, h = img.size
    new_size = (max_size, max(int(h * scale), min_size)) if w >= h else (max(int(w * scale), min_size), max_size)
    resized = img.resize(new_size, resample=Image.Resampling.BICUBIC)
    padded = ImageOps.expand(resized, (0, 0, max_size - new_size[0], max_size - new_size[1]), fill=padding_value)
    mask = torch.zeros(max_size, max_size)
    mask[: new_size[1], : new_size[0]] = 1
    return padded, mask.bool()


class AriaVisionProcessor:
    """Callable with the reference's signature; returns a dict (the reference's BatchFeature is a dict subclass)."""

    def __init__(self, max_image_size=980, min_image_size=336, image_mean=(0.5, 0.5, 0.5), image_std=(0.5, 0.5, 0.5)):
        self.max_image_size = max_image_size
        self.min_image_size = min_image_size
        self.image_mean = list(image_mean)
        self.image_std = list(image_std)
        self.model_input_names = ["pixel_values"]

    # ---- preprocessor_config.json (BaseImageProcessor.save_pretrained / from_pretrained layout: a flat JSON of the constructor fields)
    def to_dict(self) -> Dict:
        return {"image_processor_type": "AriaVisionProcessor", "max_image_size": self.max_image_size, "min_image_size": self.min_image_size,
                "image_mean": list(self.image_mean), "image_std": list(self.image_std)}

    def save_pretrained(self, save_directory: str) -> str:
        import json
        import os

        os.makedirs(save_directory, exist_ok=True)
        path = os.path.join(save_directory, "preprocessor_config.json")
        with open(path, "w") as f:
            json.dump(self.to_dict(), f, indent=2)
        return path

    @classmethod
    def from_pretrained(cls, path: str) -> "AriaVisionProcessor":
        """``path``: a checkpoint directory (or the JSON file itself); a directory without the file gives the defaults, like the hub
        checkpoint whose config only names the class."""
        import json
        import os

        file = os.path.join(path, "preprocessor_config.json") if os.path.isdir(path) else path
        raw = {}
        if os.path.exists(file):
            with open(file) as f:
                raw = json.load(f)
        keys = ("max_image_size", "min_image_size", "image_mean", "image_std")
        return cls(**{k: raw[k] for k in keys if k in raw})

    def _to_normalized_tensor(self, img: Image.Image) -> torch.Tensor:
        # torchvision ToTensor (uint8 HWC -> float CHW / 255) followed by Normalize ((x - mean) / std), both in fp32
        x = torch.from_numpy(np.array(img, dtype=np.uint8, copy=True)).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
        mean = torch.tensor(self.image_mean, dtype=torch.float32).view(-1, 1, 1)
        std = torch.tensor(self.image_std, dtype=torch.float32).view(-1, 1, 1)
        return x.sub_(mean).div_(std)

    def __call__(self, images, max_image_size: Optional[int] = 980, min_image_size: Optional[int] = 336, return_tensors="pt",
                 split_image: bool = False, split_ratio=DEFAULT_SPLIT_RATIO) -> Dict[str, torch.Tensor]:
        max_size = self.max_image_size if max_image_size is None else max_image_size
        min_size = self.min_image_size if min_image_size is None else min_image_size
        if max_size not in (490, 980):
            raise ValueError("max_image_size must be either 490 or 980")
        if isinstance(images, Image.Image):
            images = [images]
        pixel_values, pixel_masks, num_crops = [], [], []
        for image in images:
            crops = split_image_fn(image, split_image, split_ratio, max_size)
            num_crops.append(torch.tensor(len(crops)))
            for crop in crops:
                padded, mask = keep_ratio_resize_and_pixel_mask(crop, max_size, min_size)
                pixel_values.append(self._to_normalized_tensor(padded))
                pixel_masks.append(mask)
        return {"pixel_values": torch.stack(pixel_values), "pixel_mask": torch.stack(pixel_masks), "num_crops": torch.stack(num_crops)}

    preprocess = __call__


split_image_fn = split_image  # (the processor's keyword argument shadows the function name inside __call__)


def _num_image_tokens(max_image_size: int) -> int:
    if max_image_size == 490:
        return 128
    if max_image_size == 980:
        return 256
    raise ValueError(f"max_image_size must be either 490 or 980, got {max_image_size}")


class AriaProcessor:
    """processing_aria.py:40-205.  ``tokenizer`` is any HF-style tokenizer (callable on str / list of str, ``pad_token``,
    ``unk_token``)."""

    def __init__(self, image_processor: Optional[AriaVisionProcessor] = None, tokenizer=None, patch_size: int = 490,
                 chat_template: Optional[str] = None, image_token: str = "<|img|>"):
        self.image_processor = image_processor if image_processor is not None else AriaVisionProcessor(max_image_size=patch_size)
        self.tokenizer = tokenizer
        if self.tokenizer is not None and getattr(self.tokenizer, "pad_token", None) is None:
            self.tokenizer.pad_token = self.tokenizer.unk_token
        self.chat_template = chat_template
        self.image_token = image_token

    def apply_chat_template(self, messages: List[Dict], add_generation_prompt: bool = False) -> str:
        """ChatML as the hub template renders it: ``<|im_start|>{role}\\n{content}<|im_end|>\\n`` per message, an image item is
        ``<fim_prefix><|img|><fim_suffix>``, string content is taken verbatim (tests/test_aria_processor.py:41-82)."""
        out = []
        for message in messages:
            content = message["content"]
            if isinstance(content, str):
                text = content
            else:
                parts = []
                for item in content:
                    if item["type"] == "text":
                        parts.append(item["text"])
                    elif item["type"] == "image":
                        parts.append("<fim_prefix>" + self.image_token + "<fim_suffix>")
                    else:
                        raise ValueError(f"Unknown content type {item['type']} in message")
                text = "".join(parts)
            out.append(f"<|im_start|>{message['role']}\n{text}<|im_end|>\n")
        if add_generation_prompt:
            out.append("<|im_start|>assistant\n")
        return "".join(out)

    def __call__(self, text, images=None, padding=False, truncation=None, max_length: Optional[int] = None,
                 max_image_size: Optional[int] = 980, split_image: bool = False, return_tensors="pt",
                 return_final_prompts: bool = False):
        if isinstance(text, str):
            text = [text]
        elif not isinstance(text, list) and not isinstance(text[0], str):
            raise ValueError("Invalid input text. Please provide a string, or a list of strings")
        if images is not None:
            image_inputs = self.image_processor(images, return_tensors=return_tensors, max_image_size=max_image_size,
                                                split_image=split_image)
            crop_iter = iter(image_inputs.pop("num_crops"))
            # one image token per crop of that image, then every token becomes 128 / 256 copies (the projector's query count)
            prompts = [re.sub(re.escape(self.image_token), lambda _: int(next(crop_iter)) * self.image_token, p) for p in text]
            size = max_image_size if max_image_size is not None else self.image_processor.max_image_size
            n_tok = _num_image_tokens(size)
            prompts = [p.replace(self.image_token, self.image_token * n_tok) for p in prompts]
        else:
            image_inputs, prompts = {}, text
        text_inputs = self.tokenizer(prompts, return_tensors=return_tensors, padding=padding, truncation=truncation,
                                     max_length=max_length)
        batch = {**text_inputs, **image_inputs}
        return (batch, prompts) if return_final_prompts else batch

    def save_pretrained(self, save_directory: str, **kwargs) -> None:
        """processing_aria.py:216-229: the image processor's config and the tokenizer files side by side."""
        if self.image_processor is not None:
            self.image_processor.save_pretrained(save_directory)
        if self.tokenizer is not None and hasattr(self.tokenizer, "save_pretrained"):
            self.tokenizer.save_pretrained(save_directory)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, tokenizer_path: Optional[str] = None,
                        image_processor_path: Optional[str] = None, **kwargs) -> "AriaProcessor":
        """processing_aria.py:231-275 for LOCAL directories (no hub access): image processor from ``preprocessor_config.json``, tokenizer
        through ``AutoTokenizer`` (slow tokenizer, like the reference); a tokenizer that fails to load leaves ``tokenizer = None``."""
        image_processor = AriaVisionProcessor.from_pretrained(image_processor_path or pretrained_model_name_or_path)
        tokenizer = chat_template = None
        try:
            from transformers import AutoTokenizer

            tokenizer = AutoTokenizer.from_pretrained(tokenizer_path or pretrained_model_name_or_path, use_fast=False)
            chat_template = getattr(tokenizer, "chat_template", None)
        except Exception as e:  # same behaviour as the reference: warn and go on without a tokenizer
            import warnings

            warnings.warn(f"Failed to load tokenizer from {tokenizer_path or pretrained_model_name_or_path}: {e}")
        return cls(image_processor=image_processor, tokenizer=tokenizer, chat_template=chat_template)

    def batch_decode(self, *args, **kwargs):
        if self.tokenizer is None:
            raise ValueError("Tokenizer is not initialized. Please provide a valid tokenizer.")
        return self.tokenizer.batch_decode(*args, **kwargs)

    def decode(self, *args, **kwargs):
        if self.tokenizer is None:
            raise ValueError("Tokenizer is not initialized. Please provide a valid tokenizer.")
        return self.tokenizer.decode(*args, **kwargs)

    @property
    def model_input_names(self):
        return list(dict.fromkeys(list(self.tokenizer.model_input_names) + list(self.image_processor.model_input_names)))


def apply_chat_template_and_tokenize(messages_batch: List[List[Dict]], tokenizer, num_image_crop: Iterable = iter([]),
                                     max_length: int = 1024, max_image_size: int = 980) -> Dict[str, torch.Tensor]:
    """aria/data.py:29-120: ChatML ids built piecewise (so that the role prefix length is known), labels = ids on assistant turns
    after the ``<|im_start|>assistant\\n`` prefix and -100 everywhere else, right padding / truncation to min(longest, max_length),
    attention_mask = ids != pad."""
    ids = lambda s: list(tokenizer(s).input_ids)  # noqa: E731
    im_start, user, assistant, im_end, nl = ids("<|im_start|>"), ids("user"), ids("assistant"), ids("<|im_end|>"), ids("\n")
    n_tok = _num_image_tokens(max_image_size)
    num_image_crop = iter(num_image_crop)

    def render(item):
        if item["type"] == "text":
            return item["text"]
        if item["type"] == "image":
            return "<fim_prefix>" + "<|img|>" * int(next(num_image_crop)) + "<fim_suffix>"
        raise ValueError(f"Unknown content type {item['type']} in message")

    input_ids, targets = [], []
    for messages in messages_batch:
        row, tgt = [], []
        for message in messages:
            role = message["role"]
            text = "".join(render(item) for item in message["content"]).replace("<|img|>", "<|img|>" * n_tok)
            piece = im_start + (user if role == "user" else assistant) + nl + ids(text) + im_end + nl
            if role == "user":
                tgt.extend([IGNORE_TOKEN_ID] * len(piece))
            elif role == "assistant":
                prefix = len(im_start) + len(assistant) + len(nl)
                tgt.extend([IGNORE_TOKEN_ID] * prefix + piece[prefix:])
            else:
                raise ValueError(f"Unknown role: {role}")
            row.extend(piece)
        input_ids.append(row)
        targets.append(tgt)
    width = min(max(len(r) for r in input_ids), max_length)
    pad = tokenizer.pad_token_id
    for i in range(len(input_ids)):
        short = width - len(input_ids[i])
        if short > 0:
            input_ids[i] = input_ids[i] + [pad] * short
            targets[i] = targets[i] + [IGNORE_TOKEN_ID] * short
        else:
            input_ids[i] = input_ids[i][:width]
            targets[i] = targets[i][:width]
    ids_t = torch.tensor(input_ids, dtype=torch.long)
    return {"input_ids": ids_t, "labels": torch.tensor(targets, dtype=torch.long), "attention_mask": ids_t.ne(pad)}


def collate_fn(examples, tokenizer, processor, split_image: bool = False, max_seq_length: int = 1024):
    """aria/train.py:117-209.  ``processor`` is the image processor (it is called on the images alone).  Video examples must carry
    their frames already decoded (``example["video"]["frames"]``): frame extraction (decord, aria/load_video.py) is out of scope."""
    images, messages = [], []
    for example in examples:
        if example.get("video"):
            frames = example["video"].get("frames")
            if frames is None:
                raise NotImplementedError("video decoding is out of scope: pass pre-extracted frames in example['video']['frames']")
            images.extend(frames)
            for message in example["messages"]:
                for idx, item in enumerate(message["content"]):
                    if item["type"] == "video":
                        del message["content"][idx]
                        for j in range(len(frames)):
                            message["content"].insert(idx + j, {"text": None, "type": "image"})
            messages.append(example["messages"])
        else:
            if example.get("images"):
                images.extend(example["images"])
            messages.append(example["messages"])
    if images:
        images = [Image.open(im).convert("RGB") if isinstance(im, str) else im for im in images]
        image_inputs = processor(images, split_image=split_image)
        batch = apply_chat_template_and_tokenize(messages, tokenizer, iter(image_inputs.pop("num_crops")), max_length=max_seq_length,
                                                 max_image_size=processor.max_image_size)
        batch.update(image_inputs)
        batch["pixel_values"] = batch["pixel_values"].to(torch.bfloat16)
    else:
        batch = apply_chat_template_and_tokenize(messages, tokenizer, max_length=max_seq_length)
    return batch
