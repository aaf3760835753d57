"""Expert parallelism (config #5) on CPU: two gloo ranks, each with its own tokens and half of the experts, must reproduce the
single-process MoE layer (all experts local) on the same tokens -- outputs, input gradients, replicated-parameter gradients per
rank, and expert gradients summed over the ranks' contributions.  Kernels run through the SIMT emulator."""
import os
import socket
import tempfile

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "moe_layer.pt")
bf16 = torch.bfloat16


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs(rank, D):
    g = torch.Generator().manual_seed(7 + rank)
    return torch.randn(21 + 4 * rank, D, generator=g).to(bf16), torch.randn(21 + 4 * rank, D, generator=g).to(bf16)


def _params(g):
    w = {k: v.to(bf16) for k, v in g["weights"].items()}
    return [w["router.weight"], w["experts.fc1.weight"], w["experts.fc2.weight"], w["shared_experts.gate_proj.weight"],
            w["shared_experts.up_proj.weight"], w["shared_experts.down_proj.weight"]]


def _cfg(g):
    from aria_amd.functional import MoEConfig

    return MoEConfig(topk=g["cfg"]["moe_topk"], num_experts=g["cfg"]["moe_num_experts"], z_loss_coeff=1e-3, aux_loss_coeff=1e-2,
                     aux_scale=1.0)


def _worker(rank, world, port, outdir):
    import sys

    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.emu import emu_lib

    emu_lib.install()
    from aria_amd.expert_parallel import ep_moe_forward, shard_expert_weights

    g = torch.load(GOLDEN, map_location="cpu", weights_only=False)
    router, fc1, fc2, gate, up, down = _params(g)
    f1, f2 = shard_expert_weights(fc1, fc2, rank, world)
    ps = [t.clone().requires_grad_(True) for t in (router, f1, f2, gate, up, down)]
    x, gy = _inputs(rank, router.shape[1])
    x = x.requires_grad_(True)
    out = ep_moe_forward(x, *ps, _cfg(g))
    out.backward(gy)
    torch.save(dict(out=out.detach().float(), dx=x.grad.float(), grads=[p.grad.float() for p in ps]), os.path.join(outdir, f"ep{rank}.pt"))
    dist.destroy_process_group()


def test_expert_parallel_two_ranks_match_single_process():
    world = 2
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), d), nprocs=world, join=True)
        got = [torch.load(os.path.join(d, f"ep{r}.pt")) for r in range(world)]
    from tests.emu import emu_lib

    emu_lib.install()
    try:
        from aria_amd import autograd as AG

        g = torch.load(GOLDEN, map_location="cpu", weights_only=False)
        ref = []
        for r in range(world):
            ps = [t.clone().requires_grad_(True) for t in _params(g)]
            x, gy = _inputs(r, ps[0].shape[1])
            x = x.requires_grad_(True)
            out = AG.MoELayerFn.apply(x, *ps, _cfg(g))
            out.backward(gy)
            ref.append(dict(out=out.detach().float(), dx=x.grad.float(), grads=[p.grad.float() for p in ps]))
    finally:
        emu_lib.uninstall()

    def close(a, b, what, tol=2e-2):
        err = (a - b).abs().max()
        assert err <= tol * b.abs().max().clamp(min=1e-6) + 1e-6, (what, float(err), float(b.abs().max()))

    E = ref[0]["grads"][1].shape[0]
    per = E // world
    for r in range(world):
        close(got[r]["out"], ref[r]["out"], f"out rank {r}")
        close(got[r]["dx"], ref[r]["dx"], f"dx rank {r}")
        for i in (0, 3, 4, 5):  # replicated parameters: each rank holds the gradient of ITS tokens (DP-reduced separately)
            close(got[r]["grads"][i], ref[r]["grads"][i], f"replicated grad {i} rank {r}")
        for i in (1, 2):        # expert shards: contributions of BOTH ranks' tokens
            want = sum(ref[s]["grads"][i][r * per:(r + 1) * per] for s in range(world))
            close(got[r]["grads"][i], want, f"expert grad {i} rank {r}", 3e-2)
