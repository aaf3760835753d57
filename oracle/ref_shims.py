"""TEST INFRASTRUCTURE ONLY -- loader for the *real* reference (``/root/reference``).

Imports the reference's own ``aria/model`` and ``gptfast/model.py`` Python modules
through the five shims SURVEY.md section 0 (F5) / section 8c list, so that
``oracle/make_golden.py`` can generate golden fixtures from the reference itself
and ``tests/test_oracle_vs_reference.py`` can pin the CPU restatement
(``oracle/aria_oracle.py``) against it.  ``/root/reference`` exists only in the
build container: everything that runs on the GPU box uses the committed
fixtures under ``tests/golden/`` instead.

Nothing under ``aria_amd/`` may import this module.
"""
from __future__ import annotations

import importlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("ARIA_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "aria", "model"))


_loaded = None


def load_reference():
    """Return a namespace with the reference modules (moe, vis, proj, cfg, mdl, gptfast)."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not reference_available():
        raise RuntimeError(f"reference checkout not found at {REFERENCE_ROOT}")
    import torch
    import transformers.models.llama.modeling_llama as ml

    # shim 1: the dict the reference imports (moe_lm.py:31) was removed in transformers>=4.48
    if not hasattr(ml, "LLAMA_ATTENTION_CLASSES"):
        ml.LLAMA_ATTENTION_CLASSES = {
            k: ml.LlamaAttention for k in ("eager", "sdpa", "flash_attention_2")
        }
    # shim 2: namespace stubs so the package __init__ (which needs torchvision) never runs
    for name, path in (
        ("aria", os.path.join(REFERENCE_ROOT, "aria")),
        ("aria.model", os.path.join(REFERENCE_ROOT, "aria", "model")),
    ):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [path]
            sys.modules[name] = m
    # shim 3: torch.histc on CPU Long tensors (moe_lm.py:264-269 is broken on CPU as written)
    if not getattr(torch.histc, "_aria_shim", False):
        _h = torch.histc

        def histc(x, bins=100, min=0, max=0):
            if not x.is_floating_point() and x.device.type == "cpu":
                return _h(x.float(), bins=bins, min=min, max=max).to(torch.long)
            return _h(x, bins=bins, min=min, max=max)

        histc._aria_shim = True
        torch.histc = histc
    # shim 4: GroupedGEMM.forward calls torch.cuda.set_device(input.device) unconditionally
    if not getattr(torch.cuda.set_device, "_aria_shim", False):
        _sd = torch.cuda.set_device

        def set_device(d):
            if isinstance(d, torch.device) and d.type == "cpu":
                return None
            return _sd(d)

        set_device._aria_shim = True
        torch.cuda.set_device = set_device

    ns = types.SimpleNamespace()
    ns.moe = importlib.import_module("aria.model.moe_lm")
    ns.vis = importlib.import_module("aria.model.vision_encoder")
    ns.proj = importlib.import_module("aria.model.projector")
    ns.cfg = importlib.import_module("aria.model.configuration_aria")
    ns.mdl = importlib.import_module("aria.model.modeling_aria")
    gpath = os.path.join(REFERENCE_ROOT, "gptfast")
    spec = importlib.util.spec_from_file_location("aria_ref_gptfast_model", os.path.join(gpath, "model.py"))
    gm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gm)
    ns.gptfast = gm
    _loaded = ns
    return ns
