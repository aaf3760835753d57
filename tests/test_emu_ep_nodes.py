"""The fused autograd nodes the expert-parallel layer is built from (ExpertsGluFn: fc1 + SwiGLU in one launch, SharedGluFn: gate||up +
SwiGLU) against their unfused chains (ExpertsGemmFn -> SwiGLUFn, two linears -> SwiGLUFn) through the emulator: forward bit-identical,
input / weight gradients equal (aria/model/moe_lm.py:368-395, 505-525)."""
import pytest
import torch

from tests.emu import emu_lib

bf16 = torch.bfloat16


@pytest.fixture(autouse=True)
def _emu():
    emu_lib.install()
    yield
    emu_lib.uninstall()


def _rnd(*shape, seed, scale=0.3):
    return (torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale).to(bf16)


def test_experts_glu_node_equals_unfused_chain():
    from aria_amd import autograd as AG

    E, D, I = 4, 64, 128
    counts = torch.tensor([9, 0, 17, 5], dtype=torch.int32)
    off = torch.zeros(E + 1, dtype=torch.int32)
    off[1:] = torch.cumsum(counts, 0)
    M = int(off[-1])
    x0, w0, gy = _rnd(M, D, seed=1), _rnd(E, D, 2 * I, seed=2), _rnd(M, I, seed=3)
    outs = []
    for fused in (True, False):
        x, w = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
        act = AG.ExpertsGluFn.apply(x, w, off) if fused else AG.SwiGLUFn.apply(AG.ExpertsGemmFn.apply(x, w, off))
        act.backward(gy)
        outs.append((act.detach(), x.grad, w.grad))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_shared_glu_node_equals_two_linears():
    from aria_amd import autograd as AG

    T, D, I2 = 37, 64, 128
    x0, g0, u0, gy = _rnd(T, D, seed=4), _rnd(I2, D, seed=5), _rnd(I2, D, seed=6), _rnd(T, I2, seed=7)
    outs = []
    for fused in (True, False):
        x, g, u = (t.clone().requires_grad_(True) for t in (x0, g0, u0))
        act = AG.SharedGluFn.apply(x, g, u) if fused else AG.SwiGLUFn.apply(torch.cat([AG.linear(x, g), AG.linear(x, u)], dim=-1))
        act.backward(gy)
        outs.append((act.detach(), x.grad, g.grad, u.grad))
    assert torch.equal(outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1:], outs[1][1:]):
        err = (a.float() - b.float()).abs().max()
        assert float(err) <= 2e-2 * float(b.float().abs().max()) + 1e-6, float(err)  # dx: one GEMM with K = 2 I2 vs an accumulate pass
