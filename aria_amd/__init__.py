"""MI355X-native hot path of rhymes-ai/Aria.  The names ``aria.model`` exports (aria/model/__init__.py:20-26) resolve here too, lazily:

    from aria_amd import AriaForConditionalGeneration, AriaProcessor, GroupedGEMM
"""
_EXPORTS = {
    "AriaConfig": "modeling_aria", "AriaForConditionalGeneration": "modeling_aria",
    "AriaMoELMForCausalLM": "moe_lm", "GroupedGEMM": "moe_lm", "MoEAuxLossAutoScaler": "moe_lm", "AriaMoELMConfig": "moe_lm",
    "AriaProcessor": "processing", "AriaVisionProcessor": "processing",
    "AriaProjector": "vision", "AriaVisionModel": "vision", "AriaVisionConfig": "vision",
}
__all__ = sorted(_EXPORTS)


def __getattr__(name):
    if name in _EXPORTS:
        import importlib

        return getattr(importlib.import_module(f"{__name__}.{_EXPORTS[name]}"), name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
