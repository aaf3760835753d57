"""Host classes of the gptfast surface above ``generate()`` (gptfast/generate.py:187-346, gptfast/chat.py:8-66, gptfast/benchmark.py:10-48):
``ModelConfig`` / ``GenerationConfig`` / ``Generator`` / ``AriaChat`` / ``run_benchmark`` with the reference's constructor arguments and
method names, on ``aria_amd.gptfast`` (tile kernels for the prefill, the one-call-per-token engine for decode).  The ``compile*`` switches
are accepted and ignored: there is no tracing compiler on this path -- the decode step is already one C call."""
from __future__ import annotations

import time
from pathlib import Path
from statistics import mean, stdev
from typing import List, Optional

import torch

from . import gptfast as G


class GenerationConfig:
    def __init__(self, max_new_tokens: int = 100, top_k: int = 200, temperature: float = 0.8, cache_size: Optional[int] = None,
                 linear_causal_mask: bool = False, stop_strings: Optional[List[str]] = None):
        self.max_new_tokens, self.top_k, self.temperature = max_new_tokens, top_k, temperature
        self.cache_size, self.linear_causal_mask, self.stop_strings = cache_size, linear_causal_mask, stop_strings


class ModelConfig:
    def __init__(self, checkpoint_path, device: str = "cuda", precision: torch.dtype = torch.bfloat16, compile: bool = False,
                 compile_prefill: bool = False, apply_regional_compilation: bool = False):
        self.checkpoint_path, self.device, self.precision = Path(checkpoint_path), device, precision
        self.compile, self.compile_prefill, self.apply_regional_compilation = compile, compile_prefill, apply_regional_compilation


def load_model_and_processor(checkpoint_path: Path, device, precision=torch.bfloat16, model_args: Optional[G.ModelArgs] = None, processor=None):
    """``model.pth`` (convert_hf_checkpoint.py / aria_amd.checkpoint.convert_hf_checkpoint) -> ``Aria`` on ``device``; the processor comes
    from the tokenizer files next to it (generate.py:218-220) unless one is handed in."""
    if precision != torch.bfloat16:
        raise ValueError("the MI355X path computes in bf16")
    prev = torch.get_default_device() if hasattr(torch, "get_default_device") else None
    torch.set_default_device(device)
    try:
        model = G.Aria(model_args or G.ModelArgs())
    finally:
        torch.set_default_device(prev if prev is not None else "cpu")
    G.load_model_pth(model, torch.load(str(checkpoint_path), map_location="cpu", mmap=True, weights_only=True), strict=True)
    if processor is None:
        from transformers import AutoTokenizer

        from .processing import AriaProcessor, AriaVisionProcessor

        processor = AriaProcessor(tokenizer=AutoTokenizer.from_pretrained(str(Path(checkpoint_path).parent), use_fast=False),
                                  image_processor=AriaVisionProcessor())
    return model.eval(), processor


class Generator:
    """generate.py:281-346.  ``Generator(model_config, generation_config).generate(messages, image, detokenize=True)``."""

    def __init__(self, model_config: ModelConfig, generation_config: GenerationConfig, model: Optional[G.Aria] = None, processor=None):
        self.model_config, self.generation_config = model_config, generation_config
        self.model, self.processor = model, processor
        self._decoder = None
        if self.model is None:
            self.model, self.processor = load_model_and_processor(model_config.checkpoint_path, model_config.device, model_config.precision,
                                                                  processor=processor)

    def _stops(self):
        """single-token stop strings become a device-side comparison; anything longer keeps the reference's decode-and-compare callback"""
        tok, single, multi = self.processor.tokenizer, [], []
        for s in self.generation_config.stop_strings or []:
            ids = tok.encode(s)
            (single if len(ids) == 1 else multi).append(ids[0] if len(ids) == 1 else s)
        return single, multi

    def generate(self, messages: List[dict], image=None, detokenize: bool = True):
        gc, dev = self.generation_config, self.model_config.device
        text = self.processor.apply_chat_template(messages, add_generation_prompt=True)
        inputs = self.processor(text=text, images=image, return_tensors="pt")
        ids = inputs["input_ids"].to(dev)
        pv = inputs["pixel_values"].to(dev).to(torch.bfloat16) if inputs.get("pixel_values") is not None else None
        pm = inputs["pixel_mask"].to(dev) if inputs.get("pixel_mask") is not None else None
        single, multi = self._stops()
        tok = self.processor.tokenizer

        def early_stop(tokens):
            last = int(tokens[-1])
            if last in single[1:]:
                return True
            if multi:
                decoded = tok.decode(torch.cat(tokens).tolist())
                return any(decoded.endswith(s) for s in multi)
            return False

        out, self._decoder = G.generate(self.model, ids, gc.max_new_tokens, pixel_values=pv, pixel_mask=pm, temperature=gc.temperature,
                                        top_k=gc.top_k, decoder=self._decoder, stop_token=single[0] if single else None,
                                        callback=early_stop if (multi or len(single) > 1) else None, cache_size=gc.cache_size)
        new = out[ids.shape[1]:]  # the reference returns the generated part only (generate.py:174)
        return tok.decode(new.tolist()) if detokenize else new


class ChatMessage:
    def __init__(self, role: str, content: str, image_path: Optional[str] = None):
        self.role, self.content, self.image_path = role, content, image_path


class AriaChat:
    """chat.py:15-66: running history, every turn re-sends the whole conversation (images included)."""

    def __init__(self, model_config: ModelConfig, generation_config: GenerationConfig, generator: Optional[Generator] = None):
        self.generator = generator or Generator(model_config, generation_config)
        self.history: List[ChatMessage] = []

    def add_message(self, role: str, content: str, image_path: Optional[str] = None):
        self.history.append(ChatMessage(role, content, image_path))

    def format_prompt(self):
        from PIL import Image

        messages, images = [], []
        for m in self.history:
            parts = []
            if m.image_path:
                parts.append({"text": None, "type": "image"})
                img = m.image_path
                if isinstance(img, str):
                    if img.startswith(("http://", "https://")):
                        raise RuntimeError("no network on this box: pass a local image path")
                    img = Image.open(img)
                images.append(img.convert("RGB"))
            parts.append({"text": m.content, "type": "text"})
            messages.append({"role": m.role, "content": parts})
        return messages, images

    def chat(self, message: str, image_path: Optional[str] = None) -> str:
        self.add_message("user", message, image_path)
        messages, images = self.format_prompt()
        reply = self.generator.generate(messages, images or None).split("<|assistant|>")[-1].strip()
        for s in self.generator.generation_config.stop_strings or []:
            reply = reply.replace(s, "").strip()
        self.add_message("assistant", reply)
        return reply

    def reset(self):
        self.history = []


def run_benchmark(generator: Generator, messages: List[dict], image, num_runs: int = 5, warmup: int = 2) -> dict:
    """benchmark.py:10-48 (the protocol behind the reference's published tokens/s): warm-up generations, then whole-generate latencies."""
    for _ in range(warmup):
        generator.generate(messages, image)
    lat, count = [], []
    dev = generator.model_config.device
    for _ in range(num_runs):
        t0 = time.perf_counter()
        out = generator.generate(messages, image, detokenize=False)
        if str(dev).startswith("cuda"):
            torch.cuda.synchronize()
        lat.append(time.perf_counter() - t0)
        count.append(len(out))
    return {"mean_latency": mean(lat), "std_latency": stdev(lat) if len(lat) > 1 else 0, "mean_tokens": mean(count),
            "std_tokens": stdev(count) if len(count) > 1 else 0, "tokens_per_second": mean(count) / mean(lat)}
