"""Expert-parallel path ON THE GPU (VERDICT r2 missing #5): until now only gloo + the emulator had ever run it.  A one-rank RCCL process
group on cuda:0: the all-to-all dispatch / combine (``all_to_all_single`` on RCCL), the device-built reorder permutation, the fused
fc1 + SwiGLU / gate||up nodes of the EP layer and the ``_ep_local`` bookkeeping, against the plain (all experts local) fused layer on the
same weights and tokens -- at Aria's layer width.  Multi-rank semantics stay covered by tests/test_ep_gloo.py (2 and 4 gloo ranks)."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu
bf16 = torch.bfloat16


@pytest.fixture(scope="module")
def rccl_one_rank():
    import torch.distributed as dist

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


def _close(a, b, what, tol=2e-2):
    a, b = a.float(), b.float()
    err = (a - b).norm() / b.norm().clamp(min=1e-30)
    assert float(err) <= tol, (what, float(err))


def test_ep_layer_matches_local_layer_at_aria_width(rccl_one_rank):
    from aria_amd import autograd as AG
    from aria_amd.expert_parallel import ep_moe_forward
    from aria_amd.functional import MoEConfig

    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(3)
    D, E, I, k, T = 2560, 64, 1664, 6, 3000

    def w(*shape):
        return (torch.randn(shape, generator=g, device=dev) * 0.02).to(bf16)

    base = [w(E, D), w(E, D, 2 * I), w(E, I, D), w(2 * I, D), w(2 * I, D), w(D, 2 * I)]
    x0 = torch.randn((T, D), generator=g, device=dev).to(bf16)
    gy = torch.randn((T, D), generator=g, device=dev).to(bf16)
    cfg = MoEConfig(topk=k, num_experts=E, z_loss_coeff=1e-3, aux_loss_coeff=1e-2, aux_scale=1.0)
    res = []
    for fn in (lambda x, ps: ep_moe_forward(x, *ps, cfg), lambda x, ps: AG.MoELayerFn.apply(x, *ps, cfg)):
        ps = [t.clone().requires_grad_(True) for t in base]
        x = x0.clone().requires_grad_(True)
        out = fn(x, ps)
        out.backward(gy)
        torch.cuda.synchronize()
        res.append((out.detach(), x.grad, [p.grad for p in ps]))
    (o1, dx1, g1), (o2, dx2, g2) = res
    # same kernels on the same rows in the same order (world 1: the all-to-all is the identity) -> forward bit-identical
    assert torch.equal(o1, o2)
    _close(dx1, dx2, "dx", 1e-2)
    for i, (a, b) in enumerate(zip(g1, g2)):
        _close(a, b, f"grad {i}", 1e-2)


def test_model_with_expert_parallel_enabled_trains_like_the_local_one(rccl_one_rank):
    """MoELayer.enable_expert_parallel on a 2-layer LM (module-by-module decoder path, _ep_local shards) == the fused local path."""
    from aria_amd.moe_lm import AriaMoELMConfig, AriaMoELMForCausalLM

    dev = torch.device("cuda", 0)
    cfg = AriaMoELMConfig(hidden_size=512, num_hidden_layers=2, num_attention_heads=4, vocab_size=1024, moe_intermediate_size=256,
                          moe_num_experts=16, moe_topk=4, moe_num_shared_experts=2)
    torch.manual_seed(0)
    a = AriaMoELMForCausalLM(cfg)
    with torch.no_grad():
        for n, p in a.named_parameters():
            if "norm" not in n:
                p.normal_(0, 0.02)
    b = AriaMoELMForCausalLM(cfg)
    b.load_state_dict(a.state_dict())
    a, b = a.to(dev).train(), b.to(dev).train()
    for layer in b.model.layers:
        layer.mlp.enable_expert_parallel()
    assert all(getattr(l.mlp.experts.fc1.weight, "_ep_local", False) for l in b.model.layers)
    ids = torch.randint(0, 1024, (2, 300), generator=torch.Generator().manual_seed(1)).to(dev)
    la = a(input_ids=ids, labels=ids, return_logits=False).loss
    lb = b(input_ids=ids, labels=ids, return_logits=False).loss
    la.backward()
    lb.backward()
    assert abs(float(la) - float(lb)) <= 2e-3 * abs(float(la)), (float(la), float(lb))
    ga = dict(a.named_parameters())
    for n, p in b.named_parameters():
        _close(p.grad, ga[n].grad, n, 3e-2)


def test_rccl_collectives_the_dp_path_relies_on(rccl_one_rank):
    """The exact torch.distributed calls of aria_amd.parallel on RCCL (one rank: the values are trivial, the point is that this ROCm build
    accepts them): in-place reduce_scatter_tensor onto the rank's slice with AVG and SUM on bf16, all_reduce AVG on bf16 and fp32,
    all_gather_into_tensor of the updated slice, all_to_all_single with split sizes; then GradSync + ShardedAdamW + global_grad_norm on a
    small module with the process group present."""
    dist = rccl_one_rank
    dev = torch.device("cuda", 0)
    g = torch.arange(4096, device=dev, dtype=torch.float32).to(bf16)
    ref = g.clone()
    for op in (dist.ReduceOp.AVG, dist.ReduceOp.SUM):
        flat = g.clone()
        dist.reduce_scatter_tensor(flat[0:4096], flat, op=op)
        assert torch.equal(flat, ref)
    for dt in (bf16, torch.float32):
        t = ref.to(dt).clone()
        dist.all_reduce(t, op=dist.ReduceOp.AVG)
        assert torch.equal(t, ref.to(dt))
    out = torch.empty_like(ref)
    dist.all_gather_into_tensor(out, ref.clone())
    assert torch.equal(out, ref)
    rows = torch.randn(10, 64, device=dev).to(bf16)
    got = torch.empty_like(rows)
    dist.all_to_all_single(got, rows, [10], [10])
    assert torch.equal(got, rows)
    from aria_amd.parallel import GradSync, ShardedAdamW, clip_scale, global_grad_norm

    net = torch.nn.Sequential(torch.nn.Linear(512, 512, bias=False), torch.nn.Linear(512, 300)).to(dev).to(bf16)
    sync = GradSync(net, mode="reduce_scatter")          # world 1: no hooks, finish() is a no-op -- the constructor's AVG probe still runs
    opt = ShardedAdamW(net.named_parameters(), lr=1e-3)
    x = torch.randn(8, 512, device=dev).to(bf16)
    net(x).float().square().mean().backward()
    sync.finish()
    norm = global_grad_norm(opt.params, sync)
    want = float(torch.sqrt(sum(p.grad.float().square().sum() for p in net.parameters())))
    assert abs(norm - want) <= 1e-3 * want, (norm, want)
    before = net[0].weight.detach().clone()
    opt.step(grad_scale=clip_scale(norm, 1.0))
    assert not torch.equal(before, net[0].weight.detach())
