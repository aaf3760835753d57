"""Module-level parity at BASELINE.json's config widths on the MI355X: the 256x256 GEMM kernels and the
expert-major XCD tile list are the kernels under test here (asserted through aria_last_gemm_variant).  Measured per-tensor metrics go
to gpurun_out/fullwidth_parity.json."""
import os

import pytest

from tests import fullwidth_cases as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARIA_TEXT = dict(hidden_size=2560, num_attention_heads=20, num_key_value_heads=20, moe_intermediate_size=1664, moe_num_experts=64, moe_topk=6)
ARIA_VIT = dict(hidden_size=1152, num_attention_heads=16, intermediate_size=4304)


@pytest.fixture(autouse=True)
def _report():
    yield
    F.dump_report(os.path.join(ROOT, "gpurun_out", "fullwidth_parity.json"))


def test_decoder_layer_aria_width_T4096():
    """ONE decoder layer fwd+bwd at D 2560 / 20 x 128 / E 64 top-6 / I 1664, T = 2 x 2048 = 4096 (24 576 expert rows: 1664 tiles of the
    256 x 256 grouped GEMM) inside a 1-layer LM with a small vocabulary: logits, loss, all 15 gradients."""
    F.case_lm(DEV, "decoder_layer_T4096", hidden=2560, heads=20, experts=64, topk=6, inter=1664, vocab=4096, layers=1, B=2, S=2048,
              expect_big_gemm=True)


def test_two_decoder_layers_aria_width_ragged():
    """Two layers, T = 3 x 1000 (ragged row tiles, S not a multiple of any tile)."""
    F.case_lm(DEV, "decoder_2layers_T3000", hidden=2560, heads=20, experts=64, topk=6, inter=1664, vocab=2048, layers=2, B=3, S=1000,
              expect_big_gemm=True, seed=6)


def test_vit_layer_980px_padded_and_projector_256():
    """ONE ViT layer at 1152 / 16 x 72 / 4304 on 980-px images (4900 patches) with a padded pixel_mask + the projector with 256 queries."""
    F.case_vit_projector(DEV, "vit_980_layer", hidden=1152, heads=16, inter=4304, image=980, layers=1, queries=256, out_dim=2560, n_images=2,
                         valid_rows=735)


def test_config1_end_to_end():
    """BASELINE config #1: 490-px image (1225 patches -> 128 tokens) + 128 text tokens, full widths, full vocabulary, 2 + 2 layers."""
    F.case_aria_config1(DEV, "config1_490px_128text", text=dict(ARIA_TEXT, num_hidden_layers=2, vocab_size=100352),
                        vision=dict(ARIA_VIT, num_hidden_layers=2, image_size=490), queries=128, n_text=128)


def test_causal_attention_S16384_fwd_bwd():
    F.case_long_attention(DEV, "attention_causal_S16384", S=16384, H=2, hd=128)


# ---------------------------------------------------------------- the north_star's target shape and config #4's (VERDICT r2 missing #1)
def test_causal_attention_S65536_fwd_bwd():
    """One head of a 65 536-token causal sequence: flash forward + backward against the block-wise fp32 oracle (pinned on the eager form)."""
    F.case_long_attention(DEV, "attention_causal_S65536", S=65536, H=1, hd=128, stream_block=4096)


def test_vit_attention_hd72_bwd_S4900_masked():
    """hd-72 backward at the 980-px patch count (the r02 cases stopped at S = 300)."""
    F.case_vit_attention_bwd(DEV, "vit_attention_hd72_S4900", S=4900, H=2)


def test_decoder_layer_aria_width_T16384_recompute():
    """The always-on form of the long-sequence layer case: ONE full-width decoder layer at T = 16 384 (98 304 expert rows) with the recipe's
    gradient checkpointing (selective: flash (o, lse) kept), block-wise oracle attention; loss and all 15 gradients."""
    F.case_lm(DEV, "decoder_layer_T16384_recompute", hidden=2560, heads=20, experts=64, topk=6, inter=1664, vocab=2048, layers=1, B=1,
              S=16384, expect_big_gemm=True, seed=8, recompute=True, eval_pass=False, stream_block=4096)


def test_prefill_gptfast_16384_two_layers():
    """The always-on form of the config #4 case: 16 384-token prefill through the gptfast surface, 2 full-width layers."""
    F.case_prefill_gptfast(DEV, "prefill_gptfast_S16384", hidden=2560, heads=20, experts=64, topk=6, inter=1664, vocab=4096, layers=2,
                           S=16384)


# The two cases below ARE the north_star's target shape and config #4.  Their fp32 oracle is 17 + 5 minutes on the GPU box's host cores, so
# (VERDICT r3 next #1a) the SAME oracle code is evaluated by torch's fp32 kernels on the device -- after ``F.oracle_device_pin`` has shown,
# in the same process, that the device's fp32 evaluation equals the host's on a full-width layer at T = 4096 (logits, loss, every gradient
# to 2e-4): always on, no environment switch.
def test_oracle_on_device_equals_oracle_on_host_T4096():
    F.oracle_device_pin(DEV)
    assert F.REPORT["oracle_device_pin_cuda"]["logits"]["rel_l2"] <= 2e-4


@pytest.mark.parametrize("level", ["moe", "layer"])
def test_decoder_layer_aria_width_T65536_recompute(level):
    """ONE decoder layer at T = 65 536 (393 216 expert rows; byte offsets beyond 2^31 in the grouped GEMMs) with the recipe's gradient
    checkpointing at both recompute levels ("moe": what the 64K benchmark line runs -- the expert-row tensors rebuilt by
    ``functional.moe_rematerialize``; "layer": the reference recipe's form): loss and all 15 gradients of a 1-layer LM."""
    F.case_lm(DEV, f"decoder_layer_T65536_recompute_{level}", hidden=2560, heads=20, experts=64, topk=6, inter=1664, vocab=2048, layers=1, B=1,
              S=65536, expect_big_gemm=True, seed=7, recompute=level, eval_pass=False, stream_block=4096, oracle_device=DEV)


def test_config4_prefill_53248_two_layers():
    """BASELINE config #4: 53 248-token prefill through the gptfast surface, 2 full-width layers, last-position logits."""
    F.case_prefill_gptfast(DEV, "config4_prefill_S53248", hidden=2560, heads=20, experts=64, topk=6, inter=1664, vocab=4096, layers=2,
                           S=53248, oracle_device=DEV)


def test_grouped_gemm_operands_beyond_2g():
    """``gemm3_kernel<.., .., 3 / 5 / 6>`` on 430 080 expert rows: every operand crosses 2^31 bytes."""
    F.case_grouped_gemm_beyond_2g(DEV, "grouped_gemm_beyond_2g")


# ---------------------------------------------------------------- FULL DEPTH: what every published number is measured on
def test_lm_28_layers_aria_width_S2048():
    """28-layer AriaMoELMForCausalLM at Aria's widths and vocabulary, B = 1, S = 2048: eval logits, the hidden state after every layer (the
    growth curve goes into the report), training loss, gradients of layers 0 / 13 / 27 + embedding + final norm + lm_head; fp32 oracle on
    the device (pinned on the host oracle in this process), a bf16 run of the oracle code as the third arm."""
    F.case_lm_full_depth(DEV, "lm_28layers_S2048", oracle_device=DEV)


def test_vit_27_layers_980px_and_projector():
    """27-layer ViT + the 256-query projector on two 980-px images, one padded in rows and columns."""
    F.case_vit_full_depth(DEV, "vit_27layers_980px", oracle_device=DEV)


def test_config2_28_layers_prefill_and_16_decode_engine_steps():
    """Config #2 at full depth: a 280-position prefill, then 16 greedy decode-engine steps; every step's logits vs the oracle over the
    growing sequence, the token stream checked wherever the oracle's margin is resolvable, the engine's per-layer routing forced."""
    F.case_decode_full_depth(DEV, "config2_28layers_prefill280_decode16", oracle_device=DEV)


def test_lora_recipe_adapters_aria_width_T4096():
    """recipes/config_lora.yaml's adapter set on ONE full-width decoder layer + lm_head, T = 2 x 2048: the fused node (LoRA's second projection
    inside the base launches) vs the fp32 oracle on merged weights: loss and the gradients of all 20 LoRA factors pairs."""
    F.case_lm_lora(DEV, "lora_layer_T4096", hidden=2560, heads=20, experts=64, topk=6, inter=1664, vocab=4096, layers=1, B=2, S=2048)
