"""aria_amd.lora.get_lora_target_modules against the reference's own known-answer tests for that function
(/root/reference/tests/test_get_target_modules.py:9-173 -- its module-name fixture and the three expected selections, restated here as data
because ``aria.lora.utils`` imports trl, which is not in this image) plus the ``freeze_llm_layers`` rule (aria/lora/utils.py:44-53)."""
import pytest

from aria_amd.lora import get_lora_target_modules

VIT = "vision_tower.vision_model."
PROJ = "multi_modal_projector."
LM = "language_model."

# the reference fixture (test_get_target_modules.py:9-46), grouped by tower; order matters: the function keeps the model's module order
NAMED = ([VIT + n for n in ("embeddings.patch_embedding", "embeddings.position_embedding", "encoder.layers.0.self_attn.k_proj",
                            "encoder.layers.0.self_attn.v_proj", "encoder.layers.0.self_attn.q_proj", "encoder.layers.0.self_attn.out_proj",
                            "encoder.layers.0.layer_norm1", "encoder.layers.0.mlp.fc1", "encoder.layers.0.mlp.fc2",
                            "encoder.layers.0.layer_norm2")]
         + [PROJ + n for n in ("query", "cross_attn.q_proj", "cross_attn.k_proj", "cross_attn.v_proj", "cross_attn.multihead_attn.in_proj_weight",
                               "cross_attn.multihead_attn.out_proj", "cross_attn.linear", "cross_attn.layer_norm", "cross_attn.ln_kv", "ln_ffn",
                               "ffn.linear_in", "ffn.linear_out")]
         + [LM + n for n in ("model.embed_tokens", "model.layers.0.self_attn.q_proj", "model.layers.0.self_attn.k_proj",
                             "model.layers.0.self_attn.v_proj", "model.layers.0.self_attn.o_proj", "model.layers.0.mlp.gate_proj",
                             "model.layers.0.mlp.up_proj", "model.layers.0.mlp.down_proj", "model.layers.0.input_layernorm",
                             "model.layers.0.post_attention_layernorm", "model.norm", "lm_head")])
TARGETS = ["fc2", "linear_out", "lm_head", "q_proj", "linear_in", "linear", "o_proj", "up_proj", "fc1", "k_proj", "down_proj", "v_proj",
           "out_proj", "gate_proj"]

# the expected selections of the reference tests (:70-86, :112-127, :153-167)
WANT_VIT = [VIT + "encoder.layers.0." + n for n in ("self_attn.k_proj", "self_attn.v_proj", "self_attn.q_proj", "self_attn.out_proj", "mlp.fc1",
                                                     "mlp.fc2")]
WANT_PROJ = [PROJ + n for n in ("cross_attn.q_proj", "cross_attn.k_proj", "cross_attn.v_proj", "cross_attn.multihead_attn.out_proj",
                                "cross_attn.linear", "ffn.linear_in", "ffn.linear_out")]
WANT_LM = [LM + n for n in ("model.layers.0.self_attn.q_proj", "model.layers.0.self_attn.k_proj", "model.layers.0.self_attn.v_proj",
                            "model.layers.0.self_attn.o_proj", "model.layers.0.mlp.gate_proj", "model.layers.0.mlp.up_proj",
                            "model.layers.0.mlp.down_proj", "lm_head")]


@pytest.mark.parametrize("frozen,want", [("freeze_vit", WANT_PROJ + WANT_LM), ("freeze_projector", WANT_VIT + WANT_LM),
                                         ("freeze_llm", WANT_VIT + WANT_PROJ)])
def test_reference_known_answers(frozen, want):
    cfg = dict(freeze_vit=False, freeze_projector=False, freeze_llm=False, lora_target_modules=TARGETS)
    cfg[frozen] = True
    got = get_lora_target_modules(NAMED, cfg)
    assert got == want
    assert not any({"freeze_vit": "vision_tower", "freeze_projector": "multi_modal_projector", "freeze_llm": "language_model"}[frozen] in n
                   for n in got)


def test_frozen_llm_layers_are_skipped_and_names_appear_once():
    names = [LM + f"model.layers.{i}.self_attn.q_proj" for i in (0, 1, 10, 11)] + [LM + "model.layers.1.mlp.experts.fc1", LM + "lm_head"]
    cfg = dict(freeze_vit=True, freeze_projector=True, freeze_llm=False, freeze_llm_layers=[1, 11],
               lora_target_modules=["q_proj", "fc1", "proj", "lm_head"])  # "q_proj" and "proj" both match: one entry per module (utils.py:55-61)
    assert get_lora_target_modules(names, cfg) == [LM + "model.layers.0.self_attn.q_proj", LM + "model.layers.10.self_attn.q_proj", LM + "lm_head"]
