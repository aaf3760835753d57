"""TEST INFRASTRUCTURE ONLY -- loads the reference's own data-side modules (``aria/model/vision_processor.py``,
``aria/model/processing_aria.py``, ``aria/data.py``) from ``/root/reference`` so that tests can pin ``aria_amd/processing.py`` against
them, and ``make_golden_processing()`` can write the small fixture the GPU box uses.

One extra shim on top of oracle/ref_shims.py: the reference imports ``torchvision.transforms`` (absent from this image) for exactly
two things, ``ToTensor`` and ``Normalize``.  A stand-in module provides them with torchvision's documented arithmetic (uint8 HWC ->
float32 CHW / 255; ``(x - mean) / std`` in place, fp32).  Resizing / padding / cropping are PIL calls made by the reference itself.

Nothing under ``aria_amd/`` may import this module.
"""
from __future__ import annotations

import hashlib
import importlib
import json
import os
import sys
import types

import numpy as np
import torch

from oracle import ref_shims


def _install_torchvision_stand_in():
    if "torchvision" in sys.modules:
        return
    import importlib.machinery

    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")
    tv.__spec__ = importlib.machinery.ModuleSpec("torchvision", None)
    tr.__spec__ = importlib.machinery.ModuleSpec("torchvision.transforms", None)
    tv.__version__ = "0.0.0-stand-in"

    class ToTensor:
        def __call__(self, img):
            return torch.from_numpy(np.array(img, dtype=np.uint8, copy=True)).permute(2, 0, 1).contiguous().to(torch.float32).div(255)

    class Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = mean, std

        def __call__(self, x):
            mean = torch.as_tensor(self.mean, dtype=x.dtype).view(-1, 1, 1)
            std = torch.as_tensor(self.std, dtype=x.dtype).view(-1, 1, 1)
            return x.clone().sub_(mean).div_(std)

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    tr.ToTensor, tr.Normalize, tr.Compose = ToTensor, Normalize, Compose
    tv.transforms = tr
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tr


_loaded = None


def load_reference_processing():
    """namespace with vp (vision_processor), pa (processing_aria, or None if its transformers imports no longer resolve), data"""
    global _loaded
    if _loaded is not None:
        return _loaded
    ref_shims.load_reference()  # installs the aria / aria.model namespace stubs
    # resolve everything the reference takes from transformers BEFORE the stand-in exists: transformers probes torchvision with
    # importlib.util.find_spec and caches "absent"
    from transformers import AutoTokenizer, BaseImageProcessor, BatchFeature, TensorType  # noqa: F401
    from transformers.utils import import_utils

    for probe in ("is_torchvision_available", "is_torchvision_v2_available"):
        if hasattr(import_utils, probe):
            getattr(import_utils, probe)()
    _install_torchvision_stand_in()
    if not hasattr(BaseImageProcessor, "_set_processor_class"):  # removed after transformers 4.46; the reference calls it in __init__
        BaseImageProcessor._set_processor_class = lambda self, name: None
    ns = types.SimpleNamespace()
    ns.vp = importlib.import_module("aria.model.vision_processor")
    try:
        ns.pa = importlib.import_module("aria.model.processing_aria")
    except Exception as ex:  # transformers moved tokenization_utils names between 4.46 and 5.x
        ns.pa, ns.pa_error = None, ex
    ns.data = importlib.import_module("aria.data")
    _loaded = ns
    return ns


class StubTokenizer:
    """Deterministic HF-style tokenizer for the data-side tests (the hub tokenizer is unreachable): special strings are single ids,
    everything else is UTF-8 bytes + 16."""
    SPECIAL = ["<|im_start|>", "<|im_end|>", "<|img|>", "<fim_prefix>", "<fim_suffix>", "<unk>"]
    model_input_names = ["input_ids", "attention_mask"]

    def __init__(self):
        self.unk_token = "<unk>"
        self.pad_token = None
        self._split = __import__("re").compile("(" + "|".join(__import__("re").escape(s) for s in self.SPECIAL) + ")")

    @property
    def pad_token_id(self):
        return self.SPECIAL.index(self.pad_token if self.pad_token is not None else self.unk_token)

    def encode(self, text):
        out = []
        for piece in self._split.split(text):
            if piece in self.SPECIAL:
                out.append(self.SPECIAL.index(piece))
            else:
                out.extend(16 + b for b in piece.encode("utf-8"))
        return out

    def __call__(self, text, return_tensors=None, padding=False, truncation=None, max_length=None):
        if isinstance(text, str):
            return types.SimpleNamespace(input_ids=self.encode(text))
        rows = [self.encode(t) for t in text]
        if truncation and max_length:
            rows = [r[:max_length] for r in rows]
        width = max(len(r) for r in rows)
        if padding == "max_length" and max_length:
            width = max_length
        ids = [r + [self.pad_token_id] * (width - len(r)) for r in rows] if padding or len(rows) == 1 else rows
        mask = [[1] * len(r) + [0] * (len(i) - len(r)) for r, i in zip(rows, ids)]
        if return_tensors in ("pt", "PYTORCH") or str(return_tensors).lower().endswith("pytorch"):
            return {"input_ids": torch.tensor(ids), "attention_mask": torch.tensor(mask)}
        return {"input_ids": ids, "attention_mask": mask}


def test_images():
    """seeded images covering: square, wide with the min-size clamp, tall, tiny"""
    from PIL import Image

    rng = np.random.default_rng(7)
    sizes = [(768, 768), (1000, 300), (211, 977), (40, 30)]
    return [Image.fromarray(rng.integers(0, 255, (h, w, 3), dtype=np.uint8)) for (w, h) in sizes]


def digest(t: torch.Tensor) -> str:
    return hashlib.sha256(t.contiguous().numpy().tobytes()).hexdigest()


def make_golden_processing(path: str):
    """shape / sha256 / a few samples of what the REFERENCE's AriaVisionProcessor returns for test_images(), both sizes, with and
    without splitting -- small enough to commit (tests/golden/processing.json)."""
    ns = load_reference_processing()
    proc = ns.vp.AriaVisionProcessor(max_image_size=490)
    out = {}
    for size, split in ((490, False), (490, True), (980, False)):
        if True:
            r = proc(test_images(), max_image_size=size, split_image=split)
            pv, pm, nc = r["pixel_values"], r["pixel_mask"], r["num_crops"]
            out[f"{size}_{int(split)}"] = dict(pixel_values_shape=list(pv.shape), pixel_values_sha256=digest(pv), pixel_mask_sha256=digest(pm),
                                              num_crops=nc.tolist(), sample=[float(pv[i % pv.shape[0], i % 3, (37 * i) % size, (91 * i) % size]) for i in range(8)],
                                              mask_true=int(pm.sum()))
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    return out


if __name__ == "__main__":
    print(json.dumps({k: v["num_crops"] for k, v in make_golden_processing(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "processing.json")).items()}))
