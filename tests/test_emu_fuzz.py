"""Property-based sweeps (hypothesis, bounded and derandomized: the same examples every run) of the index-heavy kernels through the SIMT
emulator: shapes the fixed parametrisations do not hit -- T = 1, k = E, experts without rows, ragged widths -- for the router (bit-exact
indices under the tie protocol), the dispatcher (stable order, permute / unpermute bit-exact) and the grouped GEMM family."""
import pytest
import torch
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from tests import kernel_cases as C

DEV = "cpu"
COMMON = dict(deadline=None, derandomize=True, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])


@pytest.fixture(scope="module", autouse=True)
def emu():
    from tests.emu import emu_lib

    emu_lib.install()
    yield
    emu_lib.uninstall()


@settings(max_examples=25, **COMMON)
@given(T=st.integers(1, 300), E=st.sampled_from([2, 4, 8, 16, 64, 96, 200]), kfrac=st.floats(0.0, 1.0), f32=st.booleans())
def test_route_any_shape(T, E, kfrac, f32):
    k = max(1, min(8, E, int(round(kfrac * min(8, E)))))
    C.case_route(DEV, T, max(E, 4), k, torch.float32 if f32 else torch.bfloat16, exact=True)   # (the case forces ties through columns 1 and 3)


@settings(max_examples=20, **COMMON)
@given(T=st.integers(1, 260), E=st.sampled_from([2, 4, 8, 64]), k=st.integers(1, 6), D8=st.integers(1, 20))
def test_dispatch_any_shape(T, E, k, D8):
    C.case_dispatch(DEV, T, E, min(k, E), exact=True, D=8 * D8)


@settings(max_examples=12, **COMMON)
@given(counts=st.lists(st.one_of(st.just(0), st.integers(0, 70), st.integers(120, 300)), min_size=1, max_size=9),
       K8=st.integers(1, 12), N8=st.integers(1, 20))
def test_grouped_gemm_any_counts(counts, K8, N8):
    C.case_grouped_gemm(DEV, counts, K=8 * K8, N=8 * N8)


@settings(max_examples=10, **COMMON)
@given(counts=st.lists(st.one_of(st.just(0), st.integers(1, 70), st.integers(120, 520)), min_size=1, max_size=6).filter(lambda c: sum(c) > 0),
       K64=st.integers(1, 3), I128=st.integers(1, 3), T=st.integers(1, 300))
def test_fused_glu_launches_any_counts(counts, K64, I128, T, monkeypatch):
    """round 3: the SwiGLU-backward epilogue (gemm3_kernel<.., .., 5>) and the split gate / up launch (<false, false, 6>) on ragged / empty
    experts, partial column tiles and single-row dense problems: equal to their two-step chains bit for bit."""
    monkeypatch.setenv("ARIA_GEMM_FORCE", "3")
    C.case_gemm_dswiglu_fused(DEV, counts, 64 * K64, 128 * I128, T)
    C.case_gemm_swiglu_split(DEV, counts, 64 * K64, 128 * I128, T)


@settings(max_examples=12, **COMMON)
@given(V=st.integers(1, 3000), kfrac=st.floats(0.0, 1.2), temp=st.sampled_from([1e-6, 0.5, 0.8, 1.0, 2.0]), seed=st.integers(0, 5))
def test_sample_topk_any_vocabulary(V, kfrac, temp, seed):
    C.case_sample_topk(DEV, V, max(1, int(kfrac * V)) if kfrac <= 1.0 else None, temp, seed)


@settings(max_examples=8, **COMMON)
@given(M=st.integers(1, 600), N8=st.integers(1, 70), K64=st.integers(1, 2), b_oc=st.booleans())
def test_gemm_accumulate_any_shape(M, N8, K64, b_oc, monkeypatch):
    """round 5: `C += A B + bias` on a bf16 C through the complete-row epilogue (old tile staged through the LDS; ragged row AND column tiles,
    single rows, a strided C): one rounding, == the fp32-output launch + an fp32 add."""
    monkeypatch.setenv("ARIA_GEMM_FORCE", "3")
    C.case_gemm_accumulate_exact(DEV, M, 8 * N8, 64 * K64, b_oc)


@settings(max_examples=6, **COMMON)
@given(B=st.integers(1, 2), S=st.integers(1, 200), H=st.integers(1, 2), causal=st.booleans(), use_len=st.booleans(), extra=st.integers(0, 40))
def test_attention_backward_rope_any_shape(B, S, H, causal, use_len, extra):
    """round 5: the inverse RoPE inside the attention backward == the backward + the in-place inverse pass, any length (ragged last blocks,
    single tokens), tables longer than the sequence."""
    C.case_attention_bwd_rope(DEV, B, S, H, causal, use_len, s_rope=S + extra)


@settings(max_examples=10, **COMMON)
@given(T=st.integers(1, 200), kD=st.sampled_from([(2, 512), (3, 128), (1, 64), (6, 2560)]))
def test_unpermute_with_residual_any_rows(T, kD):
    k, D = kD
    if D == 2560 and T > 40:
        T = 40   # (the emulator's patience)
    C.case_unpermute_with_residual(DEV, T, D, k, E=8 if k < 6 else 64)


@settings(max_examples=14, **COMMON)
@given(T=st.integers(1, 150), D=st.sampled_from([256, 512, 1024, 2560]), E=st.sampled_from([32, 64]), k=st.integers(1, 8))
def test_router_fused_any_shape(T, D, E, k):
    """round 5: the one-launch router with its operands staged through the LDS == gemm + route, bit for bit, on ragged token blocks and every
    supported width / top-k."""
    C.case_router_fused(DEV, T if D < 2560 else min(T, 40), D, E, k)


@settings(max_examples=12, **COMMON)
@given(T=st.integers(1, 700), D8=st.integers(1, 320))
def test_layernorm_any_shape(T, D8):
    C.case_layernorm_two_rows_in_flight(DEV, T, 8 * D8)
