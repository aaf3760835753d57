"""Module-level parity on hardware (-m gpu): aria_amd modules (HIP kernels) vs the oracle on the golden fixtures."""
import pytest

from tests import model_cases as M

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_moe_layer_golden(golden):
    M.case_moe_layer_golden(DEV, golden)


def test_moe_layer_train_golden(golden):
    M.case_moe_layer_train_golden(DEV, golden)


def test_lm_golden(golden):
    M.case_lm_golden(DEV, golden)


def test_lm_golden_with_recompute(golden):
    M.case_lm_golden(DEV, golden, recompute=True)


def test_vit_projector_golden(golden):
    M.case_vit_projector_golden(DEV, golden)


def test_aria_full_golden(golden):
    M.case_aria_full_golden(DEV, golden)


def test_gptfast_golden(golden):
    M.case_gptfast_golden(DEV, golden)

def test_lora_grouped_gemm():
    M.case_lora_grouped_gemm(DEV)


@pytest.mark.parametrize("head_dim,max_seq", [(64, 24), (128, 24), (128, 16400)])
def test_decode_engine(head_dim, max_seq):
    M.case_decode_engine(DEV, head_dim, max_seq)


def test_hf_to_gptfast_bridge(golden):
    M.case_hf_to_gptfast_bridge(DEV, golden)


def test_decode_engine_reference_golden(golden):
    M.case_decode_engine_reference_golden(DEV, golden)


def test_decode_engine_aria_width():  # the decode kernels' Aria-width instantiations against the fp32 oracle (passes through the emulator too)
    M.case_decode_engine_aria_width(DEV, n_tokens=6)


@pytest.mark.parametrize("aria_width", [False, True])
def test_decode_engine_fused_schedule(aria_width):  # 6-launch decode schedule == 7-launch schedule, bit for bit (toy widths and Aria's)
    M.case_decode_engine_fused_schedule(DEV, aria_width=aria_width, n_tokens=6 if aria_width else 4)


def test_training_step_without_permuted_copy(monkeypatch):   # K2 in the training step: gathered fc1 forward + gathered weight gradient, bit for bit
    monkeypatch.setenv("ARIA_GEMM_FORCE", "3")
    M.case_training_step_without_permuted_copy(DEV)


def test_lora_fused_sites_with_dropout():   # adapters inside the base launches, dropout masks applied in forward and backward (site level)
    M.case_lora_fused_sites(DEV)


def test_lora_fused_node_matches_modular():
    M.case_lora_fused_vs_modular(DEV)


def test_lora_linear_lm():  # last on purpose: newest composition of already-covered kernels
    M.case_lora_linear_lm(DEV)


def test_projection_weights_are_packed_and_the_fused_operand_is_a_view():
    M.case_packed_projection_weights(DEV)
