"""Data-parallel gradient exchange for one-process-per-GPU fine-tuning (RCCL over xGMI through torch.distributed).

The reference has no in-tree collective: DP comes from DeepSpeed ZeRO-2 via accelerate
(recipes/accelerate_configs/zero2.yaml:3-17: optimizer state + gradients partitioned, parameters replicated).  Here the
exchange is designed for MI355X/xGMI instead of translated:

* ZeRO-2 shape (``mode="reduce_scatter"``, the default): a gradient is REDUCE-SCATTERED in place, so rank r ends up with the
  rank-average of exactly the 1/W slice ``[r*per, (r+1)*per)`` of each flattened tensor that its ``ShardedAdamW`` state owns
  (the rest of ``.grad`` is scratch afterwards); the optimizer updates that slice and all-gathers the bf16 parameters.  Two
  passes of the 49.8 GB of trainable bf16 over xGMI per optimizer step (reduce-scatter + all-gather) instead of the three an
  all-reduce + all-gather costs.  ``mode="all_reduce"`` keeps every replica's full averaged gradient (tests, gradient checks).
* Aria's parameters are few and huge (per layer: fc1 1.09 GB, fc2 0.55 GB, 7 dense matrices of 13-17 MB), so there is
  no bucketing-by-copy: every gradient is exchanged IN PLACE, as its own collective, the moment autograd has accumulated
  it (``register_post_accumulate_grad_hook``), on RCCL's stream -- i.e. overlapped with the rest of backward.
  Gradients are produced last-layer-first, so the exchange of layer i runs under the compute of layers < i.
* Gradient accumulation (``gradient_accumulation_steps`` of the recipe): only the LAST micro-step exchanges.  Wrap the
  earlier micro-steps in ``with sync.no_sync():`` -- their hooks do nothing, the gradients just accumulate locally, and
  nothing touches a ``.grad`` between the launch of its collective and ``finish()``.
* Tiny tensors (norm weights, 5 KB each) would waste a collective launch each; they are packed into one flat buffer
  and all-reduced once at the end (every rank keeps them whole: their optimizer shards are slices of the same values).
* xGMI is point-to-point (7 links/GPU): large messages let RCCL spread each collective over all links.

``finish()`` waits for the outstanding collectives and averages (pre-division happens via ``op=AVG`` when the backend
has it, otherwise by ONE scale per distinct tensor after SUM).
"""
from __future__ import annotations

import contextlib
from typing import List, Optional

import torch
import torch.distributed as dist

SMALL_NUMEL = 1 << 16


def shard_bounds(n: int, world: int, rank: int):
    """The slice of a flattened n-element tensor whose optimizer state (and, under ZeRO-2, reduced gradient) rank ``rank``
    owns: equal shards of ``per`` = ceil(n / world) rounded up to an even count (the AdamW kernel works on bf16 pairs), the
    last one(s) cut at n.  -> (lo, hi, per)"""
    per = (n + world - 1) // world
    per += per & 1
    return min(n, rank * per), min(n, (rank + 1) * per), per


class GradSync:
    def __init__(self, module: torch.nn.Module, process_group: Optional[dist.ProcessGroup] = None, overlap: bool = True,
                 mode: str = "all_reduce", ep_dp_group: Optional[dist.ProcessGroup] = None):
        """``ep_dp_group`` (DP x EP grid, expert-parallel degree < world): the ranks that hold the SAME expert shard -- one per data-parallel
        replica of the expert-parallel group.  Their shard gradients each cover the tokens of one replica and are summed over this group."""
        if mode not in ("all_reduce", "reduce_scatter"):
            raise ValueError(f"GradSync mode {mode!r}")
        self.module = module
        self.pg = process_group
        self.ep_dp_group = ep_dp_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.overlap = overlap
        self.mode = mode
        self.enabled = True                                 # False inside no_sync(): accumulate locally, exchange nothing
        self.handles: List = []
        self.small: List[torch.nn.Parameter] = []
        self.ep_local: List[torch.nn.Parameter] = []   # expert shards of an expert-parallel model: complete on their owner, never all-reduced
        self.large_pending: List[torch.nn.Parameter] = []
        self._launched = set()                              # ids of the parameters whose collective is in flight (one per step)
        self._hooks = []
        self.bytes_exchanged = 0                            # payload handed to collectives since construction (diagnostics / tests)
        backend = dist.get_backend(process_group) if dist.is_initialized() else "none"
        self._avg = dist.ReduceOp.AVG if backend == "nccl" else None
        if self._avg is not None and self.world > 1:
            # probe once (every rank alike): a collective library without an averaging reduction for bf16 falls back to SUM + scale
            try:
                probe = torch.ones(2, dtype=torch.bfloat16, device=next(module.parameters()).device)
                dist.all_reduce(probe, op=self._avg, group=process_group)
                if abs(float(probe[0]) - 1.0) > 1e-3:
                    raise RuntimeError("averaging all-reduce returned a wrong value")
            except Exception:
                self._avg = None
        if self.world > 1:
            for p in module.parameters():
                if not p.requires_grad:
                    continue
                if getattr(p, "_ep_local", False):
                    self.ep_local.append(p)
                    continue
                if p.numel() <= SMALL_NUMEL:
                    self.small.append(p)
                else:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    @contextlib.contextmanager
    def no_sync(self):
        """Micro-steps before the last one of a gradient-accumulation window: gradients accumulate locally, nothing is exchanged."""
        before = self.enabled
        self.enabled = False
        try:
            yield
        finally:
            self.enabled = before

    def _op(self):
        return self._avg if self._avg is not None else dist.ReduceOp.SUM

    def _reduce(self, t: torch.Tensor, async_op: bool):
        self.bytes_exchanged += t.numel() * t.element_size()
        return dist.all_reduce(t, op=self._op(), group=self.pg, async_op=async_op)

    def _exchange(self, p: torch.nn.Parameter):
        """Launch the collective of one large gradient; returns (handle, tensor to scale after a SUM)."""
        g = p.grad
        if self.mode == "reduce_scatter" and g.is_contiguous():
            flat = g.view(-1)
            lo, hi, per = shard_bounds(flat.numel(), self.world, self.rank)
            if per * self.world == flat.numel():  # equal shards (every large Aria tensor): in place, output = this rank's slice of the input
                self.bytes_exchanged += flat.numel() * flat.element_size()
                h = dist.reduce_scatter_tensor(flat[lo:hi], flat, op=self._op(), group=self.pg, async_op=True)
                return h, flat[lo:hi]
        return self._reduce(g, True), g

    def _on_grad(self, p: torch.nn.Parameter):
        if not self.enabled or id(p) in self._launched:
            return
        if self.overlap:
            self._launched.add(id(p))
            self.handles.append(self._exchange(p))
        elif all(q is not p for q in self.large_pending):
            self.large_pending.append(p)

    def finish(self):
        """Call after the LAST backward() of the step: completes every exchange.  Afterwards ``.grad`` holds the rank-average
        (mode all_reduce) or, for large tensors under reduce_scatter, the rank-average in this rank's ``shard_bounds`` slice."""
        if self.world <= 1:
            return
        ep_handles = []
        if self.ep_dp_group is not None and dist.get_world_size(self.ep_dp_group) > 1:
            for p in self.ep_local:  # DP x EP: every replica of the shard saw its own expert-parallel group's tokens -> sum over the replicas
                if p.grad is not None:
                    self.bytes_exchanged += p.grad.numel() * p.grad.element_size()
                    ep_handles.append(dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=self.ep_dp_group, async_op=True))
        for h in ep_handles:
            h.wait()
        for p in self.ep_local:  # the owner(s) accumulated the contributions of EVERY rank's tokens (sum): same 1/W as the averaged replicas
            if p.grad is not None:
                p.grad.div_(self.world)
        for p in self.large_pending:
            self.handles.append(self._exchange(p))
        self.large_pending = []
        flat = None
        smalls = [p for p in self.small if p.grad is not None]
        if smalls:
            flat = torch.cat([p.grad.reshape(-1).float() for p in smalls])
            self.handles.append((self._reduce(flat, True), flat))
        for h, _ in self.handles:
            h.wait()
        if self._avg is None:
            for _, t in self.handles:   # one entry per distinct tensor per step (``_launched``): divided exactly once
                t.div_(self.world)
        if flat is not None:
            o = 0
            for p in smalls:
                n = p.numel()
                p.grad.copy_(flat[o:o + n].view_as(p.grad))
                o += n
        self.handles = []
        self._launched = set()

    def owned_slice(self, p: torch.nn.Parameter) -> torch.Tensor:
        """The part of ``p.grad`` (flattened) that holds the rank-average after ``finish()`` on EVERY path: this rank's ``shard_bounds``
        slice.  Under ``mode="reduce_scatter"`` the rest of a large gradient is scratch (tensors that took the all-reduce fallback and the
        small ones are valid everywhere -- their owned slice is still this one); an expert-parallel shard is owned whole."""
        flat = p.grad.reshape(-1)
        if getattr(p, "_ep_local", False):
            return flat
        lo, hi, _ = shard_bounds(flat.numel(), self.world, self.rank)
        return flat[lo:hi]

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def shard_experts_rank_slices(num_experts: int, world: int, rank: int) -> range:
    """Expert-parallel ownership used by config #5 (experts [rank*E/world, (rank+1)*E/world))."""
    per = num_experts // world
    return range(rank * per, (rank + 1) * per)


class ShardedAdamW:
    """AdamW with fp32 master weights and optimizer state sharded over the data-parallel ranks (the role DeepSpeed ZeRO-2 plays in
    the reference recipe, recipes/accelerate_configs/zero2.yaml): every rank keeps master/m/v (12 B/param) only for its 1/W slice
    of each parameter's flattened storage, updates that slice with the fused HIP AdamW kernel after the gradient all-reduce and
    all-gathers the updated bf16 slices in place.  Aria-25.3B: 299 GB of optimizer state -> 37 GB per GPU at W = 8.
    Only the rank's own slice of each gradient is read, so it works behind either GradSync mode (ZeRO-2's reduce-scatter leaves
    exactly that slice reduced).

    ``params``: tensors, or (name, tensor) pairs as from ``named_parameters()``.  With names, weight decay follows HF Trainer's
    ``get_decay_parameter_names`` (what the reference recipe runs under): none for normalisation weights and biases."""

    @staticmethod
    def decays(name: str) -> bool:
        low = name.lower()
        return not (low.endswith("bias") or "norm" in low or ".ln_" in low or low.startswith("ln_"))

    def __init__(self, params, lr=5e-6, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1, process_group=None):
        items = list(params)
        named = bool(items) and isinstance(items[0], (tuple, list))
        pairs = [(n, p) for n, p in items] if named else [(None, p) for p in items]
        pairs = [(n, p) for n, p in pairs if p.requires_grad]
        self.params = [p for _, p in pairs]
        self.decay = [weight_decay if (n is None or self.decays(n)) else 0.0 for n, _ in pairs]
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.step_count = 0
        self.state = []
        for p in self.params:
            n = p.numel()
            local = bool(getattr(p, "_ep_local", False))      # an expert-parallel shard: every rank holds DIFFERENT experts -> whole state here
            lo, hi, per = (0, n, n + (n & 1)) if local else shard_bounds(n, self.world, self.rank)
            flat = p.detach().view(-1)
            self.state.append(dict(lo=lo, hi=hi, per=per, local=local, master=flat[lo:hi].float().clone(),
                                   m=torch.zeros(hi - lo, dtype=torch.float32, device=p.device),
                                   v=torch.zeros(hi - lo, dtype=torch.float32, device=p.device)))

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def state_dict(self) -> dict:
        """This rank's shard of the optimizer state (fp32 master / m / v slices + the shard bounds they belong to) and the step counter:
        what a resumed run needs next to the bf16 weights (one file per rank, like ZeRO's per-rank optimizer shards)."""
        return {"step_count": self.step_count, "world": self.world, "rank": self.rank,
                "shards": [{"lo": st["lo"], "hi": st["hi"], "master": st["master"].detach().cpu(), "m": st["m"].detach().cpu(),
                            "v": st["v"].detach().cpu()} for st in self.state]}

    def load_state_dict(self, sd: dict) -> None:
        if sd["world"] != self.world or sd["rank"] != self.rank or len(sd["shards"]) != len(self.state):
            raise ValueError(f"optimizer shard of rank {sd['rank']}/{sd['world']} with {len(sd['shards'])} tensors does not fit "
                             f"rank {self.rank}/{self.world} with {len(self.state)}")
        self.step_count = int(sd["step_count"])
        for st, src in zip(self.state, sd["shards"]):
            if (st["lo"], st["hi"]) != (src["lo"], src["hi"]):
                raise ValueError("optimizer shard bounds changed (different parameter set or world size)")
            for k in ("master", "m", "v"):
                st[k].copy_(src[k].to(st[k].device))

    @torch.no_grad()
    def step(self, lr: Optional[float] = None, grad_scale: float = 1.0):
        from . import ops

        self.step_count += 1
        lr = self.lr if lr is None else lr
        b1, b2 = self.betas
        for p, st, wd in zip(self.params, self.state, self.decay):
            if p.grad is None:
                continue
            lo, hi = st["lo"], st["hi"]
            if hi > lo:
                n = hi - lo
                pf, gf = p.view(-1)[lo:hi], p.grad.reshape(-1)[lo:hi]
                if n & 1:  # odd tail (the last shard of an odd-sized tensor): the kernel's arithmetic on ONE element, in torch
                    n -= 1
                    g = gf[n:].float() * grad_scale
                    st["m"][n:].mul_(b1).add_(g, alpha=1.0 - b1)
                    st["v"][n:].mul_(b2).addcmul_(g, g, value=1.0 - b2)
                    bc1, bc2 = 1.0 - b1 ** self.step_count, 1.0 - b2 ** self.step_count
                    w = st["master"][n:]
                    w.sub_(lr * ((st["m"][n:] / bc1) / ((st["v"][n:] / bc2).sqrt() + self.eps) + wd * w))
                    pf[n:].copy_(w.to(pf.dtype))
                if n:
                    ops.adamw_step_(pf[:n], gf[:n].contiguous(), st["master"][:n], st["m"][:n], st["v"][:n], lr=lr, beta1=b1,
                                    beta2=b2, eps=self.eps, weight_decay=wd, step=self.step_count, grad_scale=grad_scale)
            if self.world > 1 and not st["local"]:
                flat = p.view(-1)
                n_all, per = flat.numel(), st["per"]
                if n_all == per * self.world:
                    dist.all_gather_into_tensor(flat, flat[lo:hi].clone(), group=self.pg)
                else:  # ragged last shard: gather padded
                    buf = torch.zeros(per * self.world, dtype=flat.dtype, device=flat.device)
                    mine = torch.zeros(per, dtype=flat.dtype, device=flat.device)
                    mine[: hi - lo] = flat[lo:hi]
                    dist.all_gather_into_tensor(buf, mine, group=self.pg)
                    flat.copy_(buf[:n_all])


def global_grad_norm(params, sync: Optional[GradSync] = None, process_group=None) -> float:
    """L2 norm of the complete (rank-averaged) gradient, as ``torch.nn.utils.clip_grad_norm_`` / DeepSpeed's ``gradient_clipping`` see it
    (recipes/accelerate_configs/zero2.yaml:5 ``gradient_clipping: auto`` = HF ``max_grad_norm`` 1.0).  Call after ``sync.finish()``.

    Every rank adds the squares of the slices it OWNS (``GradSync.owned_slice``: valid under both exchange modes; the slices of the ranks
    partition every replicated tensor) and one scalar is all-reduced.  Expert-parallel shards are different tensors on every rank of an
    expert-parallel group and the same tensor on the ranks of ``ep_dp_group``: counted once per shard (by the group's first rank).
    The squares are summed by the HIP library (``aria_sumsq_bf16``, deterministic); one host read of the scalar per optimizer step."""
    from . import ops

    params = [p for p in params if p.grad is not None]
    if not params:
        return 0.0
    dev = params[0].grad.device
    acc = torch.zeros(1, dtype=torch.float32, device=dev)
    ws = torch.empty(1024, dtype=torch.float32, device=dev)
    world = sync.world if sync is not None else 1
    count_ep = True
    if sync is not None and sync.ep_dp_group is not None and dist.get_world_size(sync.ep_dp_group) > 1:
        count_ep = dist.get_rank(sync.ep_dp_group) == 0
    for p in params:
        local = bool(getattr(p, "_ep_local", False))
        if local and not count_ep:
            continue
        g = sync.owned_slice(p) if (sync is not None and world > 1) else p.grad.reshape(-1)
        if g.numel() == 0:
            continue
        if g.dtype != torch.bfloat16:
            g = g.to(torch.bfloat16)
        ops.sumsq_(g.contiguous(), acc, ws, accumulate=True)
    if world > 1:
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=process_group if process_group is not None else sync.pg)
    return float(acc.sqrt())


def clip_scale(norm: float, max_grad_norm: Optional[float]) -> float:
    """``clip_grad_norm_``'s coefficient: min(1, max_norm / (norm + 1e-6)); 1.0 when clipping is off (None / <= 0)."""
    if not max_grad_norm or max_grad_norm <= 0:
        return 1.0
    return min(1.0, float(max_grad_norm) / (norm + 1e-6))


def cosine_lr(step: int, total: int, base_lr: float, warmup_ratio: float = 0.01) -> float:
    """lr_scheduler_type: cosine with warmup_ratio (recipes/config_full.yaml:27-28) as HF Trainer runs it: the learning rate of the
    step-th optimizer step (1-based) is ``lr_lambda(step - 1)`` of get_cosine_schedule_with_warmup (the scheduler advances AFTER
    the optimizer), with ``num_warmup_steps = ceil(total * warmup_ratio)`` (TrainingArguments.get_warmup_steps) -- so the first
    step of a run with warm-up uses lr 0."""
    import math

    warm = math.ceil(total * warmup_ratio)
    cur = step - 1
    if cur < warm:
        return base_lr * cur / max(1, warm)
    prog = (cur - warm) / max(1, total - warm)
    return base_lr * 0.5 * (1.0 + math.cos(math.pi * min(1.0, prog)))
