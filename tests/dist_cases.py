"""Multi-rank cases of the N > 1 path, written once for both transports: two (or more) gloo ranks on the CPU through the SIMT emulator
(tests/test_dist_cases_gloo.py, runs everywhere) and the SAME workers over RCCL with one rank per GPU (tests/test_gpu_rccl_multi.py:
runs whenever >= 2 GPUs are visible, skipped otherwise -- VERDICT r3 next #8: the first multi-GPU lease should test, not debug).

  * ``dp_lm``: the golden 2-layer LM on a different batch per rank under ``aria_amd.parallel.GradSync`` in both exchange modes (per-tensor
    all-reduce; ZeRO-2 in-place ``reduce_scatter_tensor`` onto the owner's ``shard_bounds`` slice, launched from post-accumulate hooks
    under the backward), then ``global_grad_norm`` over owned slices, one ``ShardedAdamW`` step and its all-gather of the updated bf16
    parameters.  Checked against the average of single-process gradients and a plain fp32 AdamW.
  * ``ep_layer``: the MoE layer with the experts sharded over the ranks (``expert_parallel.ep_moe_forward``: all-to-all(v) dispatch and
    combine) on different tokens per rank against the local fused layer -- outputs, input gradients, replicated and expert gradients.

Reference: the reference shards nothing itself (DeepSpeed ZeRO-2 through accelerate, recipes/accelerate_configs/zero2.yaml)."""
import os
import socket
import tempfile

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN_LM = os.path.join(ROOT, "tests", "golden", "lm.pt")
GOLDEN_MOE = os.path.join(ROOT, "tests", "golden", "moe_layer.pt")
bf16 = torch.bfloat16


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup(rank, world, port, backend):
    """-> device of this rank.  gloo: CPU tensors, kernels through the emulator; nccl (= RCCL): cuda:rank, the real library."""
    import sys

    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist

    if backend == "nccl":
        torch.cuda.set_device(rank)
        dev = torch.device("cuda", rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        from tests.emu import emu_lib

        emu_lib.install()
        dev = torch.device("cpu")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    return dev


def _local_device(backend):
    if backend == "nccl":
        return torch.device("cuda", 0)
    from tests.emu import emu_lib

    emu_lib.install()
    return torch.device("cpu")


# ------------------------------------------------------------------------------------------------------------------ DP
def _build_lm(dev):
    from aria_amd.moe_lm import AriaMoELMConfig, AriaMoELMForCausalLM, load_reference_state_dict

    g = torch.load(GOLDEN_LM, map_location="cpu", weights_only=False)
    lm = AriaMoELMForCausalLM(AriaMoELMConfig(**g["cfg"]))
    load_reference_state_dict(lm, g["weights"])
    return lm.to(dev).train(), g["cfg"]["vocab_size"]


def _lm_ids(rank, vocab):
    return torch.randint(1, vocab, (2, 12), generator=torch.Generator().manual_seed(100 + rank))


LR, BETAS, EPS, WD = 1e-2, (0.9, 0.95), 1e-8, 0.1


def _dp_worker(rank, world, port, outdir, backend, mode, overlap):
    dev = _setup(rank, world, port, backend)
    import torch.distributed as dist

    from aria_amd.parallel import GradSync, ShardedAdamW, global_grad_norm, shard_bounds

    lm, vocab = _build_lm(dev)
    sync = GradSync(lm, overlap=overlap, mode=mode)
    ids = _lm_ids(rank, vocab).to(dev)
    lm(input_ids=ids, labels=ids, return_logits=False).loss.backward()
    sync.finish()
    owned = {}
    for n, p in lm.named_parameters():
        lo, hi, _ = shard_bounds(p.numel(), world, rank)
        assert sync.owned_slice(p).data_ptr() == p.grad.reshape(-1)[lo:hi].data_ptr()
        owned[n] = sync.owned_slice(p).float().cpu().clone()
    full = {n: p.grad.float().cpu().clone() for n, p in lm.named_parameters()} if mode == "all_reduce" else None
    norm = global_grad_norm([p for p in lm.parameters()], sync)
    named = [(n, p) for n, p in lm.named_parameters() if p.requires_grad]
    opt = ShardedAdamW(named, lr=LR, betas=BETAS, eps=EPS, weight_decay=WD)
    opt.step()
    params = {n: p.detach().float().cpu().clone() for n, p in lm.named_parameters()}
    torch.save({"owned": owned, "full": full, "norm": norm, "params": params, "bytes": sync.bytes_exchanged}, os.path.join(outdir, f"dp{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def run_dp_lm(backend: str, world: int, mode: str, overlap: bool = True):
    from aria_amd.parallel import ShardedAdamW, shard_bounds

    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_dp_worker, args=(world, free_port(), d, backend, mode, overlap), nprocs=world, join=True)
        got = [torch.load(os.path.join(d, f"dp{r}.pt")) for r in range(world)]
    dev = _local_device(backend)
    singles, start = [], None
    for r in range(world):
        lm, vocab = _build_lm(dev)
        if start is None:
            start = {n: p.detach().float().cpu().clone() for n, p in lm.named_parameters()}
        ids = _lm_ids(r, vocab).to(dev)
        lm(input_ids=ids, labels=ids, return_logits=False).loss.backward()
        singles.append({n: p.grad.float().cpu() for n, p in lm.named_parameters()})
    want = {n: sum(s[n] for s in singles) / world for n in singles[0]}
    tol = 1.5e-2   # the exchange averages bf16 gradients in bf16 (RCCL AVG) or fp32-packed (small tensors); the reference average is fp32
    for r in range(world):
        for n, w in want.items():
            lo, hi, _ = shard_bounds(w.numel(), world, r)
            ws = w.reshape(-1)[lo:hi]
            err = (got[r]["owned"][n] - ws).abs().max() if ws.numel() else torch.tensor(0.0)
            assert err <= tol * w.abs().max().clamp(min=1e-6) + 1e-6, (mode, n, r, float(err))
            if mode == "all_reduce":
                err = (got[r]["full"][n] - w).abs().max()
                assert err <= tol * w.abs().max().clamp(min=1e-6) + 1e-6, (n, r, float(err))
    if mode == "all_reduce":
        for n in want:
            assert all(torch.equal(got[0]["full"][n], got[r]["full"][n]) for r in range(1, world)), n   # replicas agree bit for bit
    wn = float(torch.sqrt(sum(w.double().square().sum() for w in want.values())))
    for r in range(world):
        assert got[r]["norm"] == got[0]["norm"] and abs(got[r]["norm"] - wn) <= 2e-2 * wn, (got[r]["norm"], wn)
    # one AdamW step from the averaged gradient (HF Trainer's decay rule), then every rank holds the SAME updated bf16 parameters
    for n, w in want.items():
        g = w.to(bf16).float()
        m, v = (1 - BETAS[0]) * g, (1 - BETAS[1]) * g * g
        upd = (m / (1 - BETAS[0])) / ((v / (1 - BETAS[1])).sqrt() + EPS)
        wd = WD if ShardedAdamW.decays(n) else 0.0
        ref = start[n] - LR * (upd + wd * start[n])
        for r in range(world):
            assert torch.equal(got[r]["params"][n], got[0]["params"][n]), (n, r)          # the all-gather left identical replicas
        # sign(g) * lr dominates: an element whose averaged gradient is ~0 may step either way -> compare where |g| is resolvable
        ok = g.abs() > 1e-2 * g.abs().max().clamp(min=1e-12)
        diff = (got[0]["params"][n] - ref).abs()[ok]
        assert diff.numel() == 0 or float(diff.max()) <= 0.25 * LR + 2e-2 * float(start[n].abs().max()), (n, float(diff.max()))
    return got


# ------------------------------------------------------------------------------------------------------------------ EP
def _ep_inputs(rank, D):
    g = torch.Generator().manual_seed(7 + rank)
    return torch.randn(21 + 4 * rank, D, generator=g).to(bf16), torch.randn(21 + 4 * rank, D, generator=g).to(bf16)


def _ep_params(dev, width=None):
    """The golden MoE layer (D 64, I 32: the re-ordering form of the EP layer), or -- ``width`` = (D, I, E, k) -- random weights at a width
    the segment launches take (K % 64 == 0, I % 128 == 0: the exchange's output consumed in arrival order)."""
    from aria_amd.functional import MoEConfig

    if width is not None:
        D, I, E, k = width
        gen = torch.Generator().manual_seed(4242)

        def r(*shape):
            return (torch.randn(shape, generator=gen) * 0.08).to(bf16).to(dev)

        cfg = MoEConfig(topk=k, num_experts=E, z_loss_coeff=1e-3, aux_loss_coeff=1e-2, aux_scale=1.0)
        return [r(E, D), r(E, D, 2 * I), r(E, I, D), r(2 * I, D), r(2 * I, D), r(D, 2 * I)], cfg
    g = torch.load(GOLDEN_MOE, map_location="cpu", weights_only=False)
    w = {k: v.to(bf16).to(dev) for k, v in g["weights"].items()}
    cfg = MoEConfig(topk=g["cfg"]["moe_topk"], num_experts=g["cfg"]["moe_num_experts"], z_loss_coeff=1e-3, aux_loss_coeff=1e-2, aux_scale=1.0)
    return [w["router.weight"], w["experts.fc1.weight"], w["experts.fc2.weight"], w["shared_experts.gate_proj.weight"],
            w["shared_experts.up_proj.weight"], w["shared_experts.down_proj.weight"]], cfg


def _ep_worker(rank, world, port, outdir, backend, width):
    dev = _setup(rank, world, port, backend)
    import torch.distributed as dist

    from aria_amd import ops
    from aria_amd.expert_parallel import ep_moe_forward, shard_expert_weights

    (router, fc1, fc2, gate, up, down), cfg = _ep_params(dev, width)
    seg_calls = []
    if width is not None:   # the segment launches must be what runs at this width
        orig = ops.grouped_gemm_swiglu_seg
        ops.grouped_gemm_swiglu_seg = lambda *a, **k: (seg_calls.append(1), orig(*a, **k))[1]
    f1, f2 = shard_expert_weights(fc1, fc2, rank, world)
    ps = [t.clone().requires_grad_(True) for t in (router, f1, f2, gate, up, down)]
    x, gy = (t.to(dev) for t in _ep_inputs(rank, router.shape[1]))
    x = x.requires_grad_(True)
    for _ in range(2):   # twice: the second call re-uses the side stream / pinned split buffers of the first
        for p in ps:
            p.grad = None
        x.grad = None
        out = ep_moe_forward(x, *ps, cfg)
        out.backward(gy)
    if dev.type == "cuda":
        torch.cuda.synchronize()
    chunks = int(os.environ.get("ARIA_EP_CHUNKS", "1")) if int(os.environ.get("ARIA_EP_CHUNK_MIN", "1024")) <= 8 else 1
    assert width is None or len(seg_calls) == 2 * chunks, (seg_calls, chunks)
    torch.save(dict(out=out.detach().float().cpu(), dx=x.grad.float().cpu(), grads=[p.grad.float().cpu() for p in ps]),
               os.path.join(outdir, f"ep{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def run_ep_layer(backend: str, world: int, width=None):
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_ep_worker, args=(world, free_port(), d, backend, width), nprocs=world, join=True)
        got = [torch.load(os.path.join(d, f"ep{r}.pt")) for r in range(world)]
    dev = _local_device(backend)
    from aria_amd import autograd as AG

    ref = []
    for r in range(world):
        params, cfg = _ep_params(dev, width)
        ps = [t.clone().requires_grad_(True) for t in params]
        x, gy = (t.to(dev) for t in _ep_inputs(r, ps[0].shape[1]))
        x = x.requires_grad_(True)
        out = AG.MoELayerFn.apply(x, *ps, cfg)
        out.backward(gy)
        ref.append(dict(out=out.detach().float().cpu(), dx=x.grad.float().cpu(), grads=[p.grad.float().cpu() for p in ps]))

    def close(a, b, what, tol=2e-2):
        err = (a - b).abs().max()
        assert err <= tol * b.abs().max().clamp(min=1e-6) + 1e-6, (what, float(err), float(b.abs().max()))

    E = ref[0]["grads"][1].shape[0]
    per = E // world
    for r in range(world):
        close(got[r]["out"], ref[r]["out"], f"out rank {r}")
        close(got[r]["dx"], ref[r]["dx"], f"dx rank {r}")
        for i in (0, 3, 4, 5):  # replicated parameters: each rank holds the gradient of ITS tokens (reduced by the DP exchange)
            close(got[r]["grads"][i], ref[r]["grads"][i], f"replicated grad {i} rank {r}")
        for i in (1, 2):        # expert shards: contributions of EVERY rank's tokens
            want = sum(ref[s]["grads"][i][r * per:(r + 1) * per] for s in range(world))
            close(got[r]["grads"][i], want, f"expert grad {i} rank {r}", 3e-2)
    return got
