"""Module-level parity through the SIMT emulator (CPU): aria_amd modules vs the oracle on the golden fixtures."""
import pytest
import torch

from tests import model_cases as M

DEV = "cpu"


@pytest.fixture(scope="module", autouse=True)
def emu():
    from tests.emu import emu_lib

    emu_lib.install()
    yield
    emu_lib.uninstall()


def test_moe_layer_golden(golden):
    M.case_moe_layer_golden(DEV, golden)


def test_moe_layer_train_golden(golden):
    M.case_moe_layer_train_golden(DEV, golden)


def test_lm_golden(golden):
    M.case_lm_golden(DEV, golden)


def test_vit_projector_golden(golden):
    M.case_vit_projector_golden(DEV, golden)


def test_aria_full_golden(golden):
    M.case_aria_full_golden(DEV, golden)


def test_gptfast_golden(golden):
    M.case_gptfast_golden(DEV, golden)

def test_lora_grouped_gemm():
    M.case_lora_grouped_gemm(DEV)


def test_lora_linear_lm():
    M.case_lora_linear_lm(DEV)


@pytest.mark.parametrize("head_dim,max_seq", [(128, 16400)])  # (64, 24), (128, 24) run on hardware; hd 64 is covered by the split-KV case below
def test_decode_engine(head_dim, max_seq):
    M.case_decode_engine(DEV, head_dim, max_seq)


def test_decode_engine_split_kv(monkeypatch):
    """opt-in flash-decoding form of the engine's attention (ARIA_DECODE_SPLIT_KV): 4 key ranges for S_max = 4000, all but the first empty
    at these positions; tests/test_emu_kernels.py::test_decode_attention_split_kv covers populated ranges."""
    monkeypatch.setenv("ARIA_DECODE_SPLIT_KV", "1")
    M.case_decode_engine(DEV, 64, 4000)


def test_hf_to_gptfast_bridge(golden):
    M.case_hf_to_gptfast_bridge(DEV, golden)


def test_decode_engine_reference_golden(golden):
    M.case_decode_engine_reference_golden(DEV, golden)


def test_seam_attention_interface_native_hd72_without_grad():
    """aria_amd.seams.attention_interface as the (frozen) ViT calls it: non-causal, head_dim 72, a key mask that is not a prefix, no grad ->
    the native hd-72 forward kernel (no padding to 128); against fp32 torch on the same bf16 operands."""
    import types

    from aria_amd import seams

    B, H, S, hd = 2, 2, 70, 72
    g = torch.Generator().manual_seed(0)
    q, k, v = (torch.randn(B, H, S, hd, generator=g).to(torch.bfloat16) for _ in range(3))
    mask = torch.ones(B, S, dtype=torch.bool)
    mask[1, 5:9] = False
    mask[1, 60:] = False
    with torch.no_grad():
        out, w = seams.attention_interface(types.SimpleNamespace(is_causal=False), q, k, v, mask, scaling=hd ** -0.5)
    assert w is None and out.shape == (B, S, H, hd)
    sc = torch.einsum("bhqd,bhkd->bhqk", q.float(), k.float()) * hd ** -0.5
    sc = sc.masked_fill(~mask[:, None, None, :], float("-inf"))
    want = torch.einsum("bhqk,bhkd->bqhd", torch.softmax(sc, -1), v.float())
    M.rel_close(out, want, 2e-2, "hd 72 attention through the seam")
