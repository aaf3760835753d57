"""N > 1 path on CPU: two gloo ranks run the tiny golden LM (kernels through the SIMT emulator) on different batches with
aria_amd.parallel.GradSync (overlapped per-parameter all-reduce + packed small tensors); the synchronised gradients must
equal the average of the two single-process gradients."""
import os
import socket
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "lm.pt")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(g):
    from aria_amd.moe_lm import AriaMoELMConfig, AriaMoELMForCausalLM, load_reference_state_dict

    lm = AriaMoELMForCausalLM(AriaMoELMConfig(**g["cfg"]))
    load_reference_state_dict(lm, g["weights"])
    return lm.train()


def _ids(rank, g):
    gen = torch.Generator().manual_seed(100 + rank)
    return torch.randint(1, g["cfg"]["vocab_size"], (2, 12), generator=gen)


def _worker(rank, world, port, outdir, overlap):
    import sys

    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.emu import emu_lib

    emu_lib.install()
    from aria_amd.parallel import GradSync

    g = torch.load(GOLDEN, map_location="cpu", weights_only=False)
    lm = _build(g)
    sync = GradSync(lm, overlap=overlap)
    ids = _ids(rank, g)
    out = lm(input_ids=ids, labels=ids, return_logits=False)
    out.loss.backward()
    sync.finish()
    torch.save({n: p.grad.float() for n, p in lm.named_parameters() if p.grad is not None}, os.path.join(outdir, f"g{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [True, False])
def test_two_rank_gradient_exchange_matches_average(overlap):
    world = 2
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), d, overlap), nprocs=world, join=True)
        got = [torch.load(os.path.join(d, f"g{r}.pt")) for r in range(world)]
    from tests.emu import emu_lib

    emu_lib.install()
    try:
        g = torch.load(GOLDEN, map_location="cpu", weights_only=False)
        singles = []
        for r in range(world):
            lm = _build(g)
            ids = _ids(r, g)
            lm(input_ids=ids, labels=ids, return_logits=False).loss.backward()
            singles.append({n: p.grad.float() for n, p in lm.named_parameters() if p.grad is not None})
    finally:
        emu_lib.uninstall()
    assert set(got[0]) == set(singles[0])
    for n in singles[0]:
        want = sum(s_[n] for s_ in singles) / world
        for r in range(world):
            err = (got[r][n] - want).abs().max()
            assert err <= 1e-2 * want.abs().max().clamp(min=1e-6) + 1e-6, (n, r, float(err))
        assert all(torch.equal(got[0][n], got[r_][n]) for r_ in range(1, world)), n  # replicas hold identical gradients after the exchange


# ---------------------------------------------------------------- gradient accumulation and the ZeRO-2 (reduce-scatter) exchange
def _accum_worker(rank, world, port, outdir, mode, accum):
    import sys

    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import contextlib

    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aria_amd.parallel import GradSync, shard_bounds

    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(300, 300), torch.nn.Linear(300, 301, bias=True))  # 90 000 / 90 300 elements: both "large"
    sync = GradSync(net, mode=mode)
    for micro in range(accum):
        x = torch.randn(5, 300, generator=torch.Generator().manual_seed(10 * rank + micro))
        with (sync.no_sync() if micro < accum - 1 else contextlib.nullcontext()):
            (net(x).square().mean() / accum).backward()
    sync.finish()
    out = {}
    for n, p in net.named_parameters():
        lo, hi, _ = shard_bounds(p.numel(), world, rank)
        out[n] = p.grad.reshape(-1)[lo:hi].clone() if mode == "reduce_scatter" and p.numel() > (1 << 16) else p.grad.clone()
    torch.save({"grads": out, "bytes": sync.bytes_exchanged}, os.path.join(outdir, f"a{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,accum", [("all_reduce", 2), ("reduce_scatter", 1), ("reduce_scatter", 3)])
def test_accumulation_exchanges_once_and_averages(mode, accum):
    """with gradient accumulation the hooks used to fire -- and, on the SUM path, divide -- once per micro-step.
    Now only the last micro-step exchanges; the result is the rank-average of the accumulated gradients, and the payload handed
    to collectives does not depend on the accumulation depth."""
    from aria_amd.parallel import shard_bounds

    world = 2
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_accum_worker, args=(world, _free_port(), d, mode, accum), nprocs=world, join=True)
        got = [torch.load(os.path.join(d, f"a{r}.pt")) for r in range(world)]
    singles = []
    for r in range(world):
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(300, 300), torch.nn.Linear(300, 301, bias=True))
        for micro in range(accum):
            x = torch.randn(5, 300, generator=torch.Generator().manual_seed(10 * r + micro))
            (net(x).square().mean() / accum).backward()
        singles.append({n: p.grad.clone() for n, p in net.named_parameters()})
    n_large = sum(p.numel() for p in net.parameters() if p.numel() > (1 << 16))
    n_small = sum(p.numel() for p in net.parameters() if p.numel() <= (1 << 16))
    for r in range(world):
        assert got[r]["bytes"] == 4 * (n_large + n_small)                      # every element handed to a collective exactly once
        for n, g in got[r]["grads"].items():
            want = sum(s_[n] for s_ in singles) / world
            if mode == "reduce_scatter" and want.numel() > (1 << 16):
                lo, hi, _ = shard_bounds(want.numel(), world, r)
                want = want.reshape(-1)[lo:hi]
            torch.testing.assert_close(g, want, rtol=1e-5, atol=1e-7)


def test_sharded_adamw_odd_numel_and_decay_groups():
    """the last element of an odd-sized tensor used to be skipped; weight decay applied to norms and biases.  One
    rank, emulated kernels: every element equals a plain fp32 AdamW with HF Trainer's decay rule."""
    from tests.emu import emu_lib

    emu_lib.install()
    try:
        from aria_amd.parallel import ShardedAdamW

        torch.manual_seed(1)
        params = {"layer.weight": torch.randn(7, 3).bfloat16(), "layer.bias": torch.randn(7).bfloat16(), "norm.weight": torch.randn(5).bfloat16()}
        params = {k: torch.nn.Parameter(v) for k, v in params.items()}
        ref = {k: v.detach().float().clone() for k, v in params.items()}
        m = {k: torch.zeros_like(v) for k, v in ref.items()}
        vv = {k: torch.zeros_like(v) for k, v in ref.items()}
        opt = ShardedAdamW(list(params.items()), lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1)
        assert opt.decay == [0.1, 0.0, 0.0]
        for step in range(1, 4):
            for k, p in params.items():
                p.grad = torch.randn(p.shape, generator=torch.Generator().manual_seed(step * 7 + len(k))).bfloat16()
            opt.step()
            for k, p in params.items():
                g = p.grad.float()
                m[k] = 0.9 * m[k] + 0.1 * g
                vv[k] = 0.95 * vv[k] + 0.05 * g * g
                wd = 0.1 if k == "layer.weight" else 0.0
                ref[k] = ref[k] - 1e-2 * ((m[k] / (1 - 0.9 ** step)) / ((vv[k] / (1 - 0.95 ** step)).sqrt() + 1e-8) + wd * ref[k])
        for k, p in params.items():
            torch.testing.assert_close(p.detach().float(), ref[k].bfloat16().float(), rtol=0, atol=2e-2 * ref[k].abs().max().item())
            torch.testing.assert_close(opt.state[list(params).index(k)]["master"], ref[k].reshape(-1), rtol=2e-5, atol=1e-6)
    finally:
        emu_lib.uninstall()


# ---------------------------------------------------------------- gradient clipping norm over owned slices
def _norm_worker(rank, world, port, outdir, mode):
    import sys

    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.emu import emu_lib

    emu_lib.install()
    from aria_amd.parallel import GradSync, global_grad_norm

    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(300, 300), torch.nn.Linear(300, 301, bias=True)).bfloat16()
    sync = GradSync(net, mode=mode)
    x = torch.randn(5, 300, generator=torch.Generator().manual_seed(10 * rank)).bfloat16()
    net(x).float().square().mean().backward()
    sync.finish()
    norm = global_grad_norm(list(net.parameters()), sync)
    torch.save({"norm": norm}, os.path.join(outdir, f"n{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["all_reduce", "reduce_scatter"])
def test_global_grad_norm_counts_every_element_once(mode):
    """HF max_grad_norm / DeepSpeed gradient_clipping (zero2.yaml:5): after a ZeRO-2 reduce-scatter only the owned slice of a large
    gradient is valid, so the norm is assembled from owned slices + one scalar all-reduce -- equal on every rank to the norm of the
    rank-averaged full gradient."""
    world = 2
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_norm_worker, args=(world, _free_port(), d, mode), nprocs=world, join=True)
        got = [torch.load(os.path.join(d, f"n{r}.pt"))["norm"] for r in range(world)]
    singles = []
    for r in range(world):
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(300, 300), torch.nn.Linear(300, 301, bias=True)).bfloat16()
        x = torch.randn(5, 300, generator=torch.Generator().manual_seed(10 * r)).bfloat16()
        net(x).float().square().mean().backward()
        singles.append([p.grad.float() for p in net.parameters()])
    want = float(torch.sqrt(sum(((a + b) / 2).square().sum() for a, b in zip(*singles))))
    assert got[0] == got[1]
    assert abs(got[0] - want) <= 1e-2 * want, (got, want)


def test_clip_scale_and_single_process_norm():
    from tests.emu import emu_lib

    emu_lib.install()
    try:
        from aria_amd.parallel import clip_scale, global_grad_norm

        ps = [torch.nn.Parameter(torch.zeros(n).bfloat16()) for n in (7, 1027, 16)]
        for i, p in enumerate(ps):
            g = torch.randn(p.numel() + 3, generator=torch.Generator().manual_seed(i)).bfloat16()
            p.grad = g[3:] if i == 1 else g[:-3].clone()  # i == 1: an element-aligned (not 16-byte aligned) view, like a shard of a flattened gradient
        want = float(torch.sqrt(sum(p.grad.float().square().sum() for p in ps)))
        got = global_grad_norm(ps)
        assert abs(got - want) <= 1e-5 * want, (got, want)
        assert clip_scale(got, None) == 1.0 and clip_scale(0.5, 1.0) == 1.0
        assert abs(clip_scale(4.0, 1.0) - 1.0 / (4.0 + 1e-6)) < 1e-12
    finally:
        emu_lib.uninstall()
