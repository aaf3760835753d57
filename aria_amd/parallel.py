"""Data-parallel gradient exchange for one-process-per-GPU fine-tuning (RCCL over xGMI through torch.distributed).

The reference has no in-tree collective: DP comes from DeepSpeed ZeRO-2 via accelerate
(recipes/accelerate_configs/zero2.yaml:3-17).  Here the exchange is designed for MI355X/xGMI instead of translated:

* Aria's parameters are few and huge (per layer: fc1 1.09 GB, fc2 0.55 GB, 7 dense matrices of 13-17 MB), so there is
  no bucketing-by-copy: every gradient is all-reduced IN PLACE, as its own collective, the moment autograd has
  accumulated it (``register_post_accumulate_grad_hook``), on RCCL's stream -- i.e. overlapped with the rest of backward.
  Gradients are produced last-layer-first, so the exchange of layer i runs under the compute of layers < i.
* Tiny tensors (norm weights, 5 KB each) would waste a collective launch each; they are packed into one flat buffer
  and reduced once at the end.
* xGMI is point-to-point (7 links/GPU): large messages let RCCL spread each all-reduce over all links.

``finish()`` waits for the outstanding collectives and averages (pre-division happens via ``op=AVG`` when the backend
has it, otherwise by a scale after SUM).
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist

SMALL_NUMEL = 1 << 16


class GradSync:
    def __init__(self, module: torch.nn.Module, process_group: Optional[dist.ProcessGroup] = None, overlap: bool = True):
        self.module = module
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.overlap = overlap
        self.handles: List = []
        self.small: List[torch.nn.Parameter] = []
        self.ep_local: List[torch.nn.Parameter] = []   # expert shards of an expert-parallel model: complete on their owner, never all-reduced
        self.large_pending: List[torch.nn.Parameter] = []
        self._hooks = []
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

    def _reduce(self, t: torch.Tensor, async_op: bool):
        if self._avg is not None:
            return dist.all_reduce(t, op=self._avg, group=self.pg, async_op=async_op)
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg, async_op=async_op)

    def _on_grad(self, p: torch.nn.Parameter):
        if self.overlap:
            self.handles.append((self._reduce(p.grad, True), p))
        else:
            self.large_pending.append(p)

    def finish(self):
        """Call after backward(): completes every exchange; afterwards .grad holds the rank-average."""
        if self.world <= 1:
            return
        for p in self.ep_local:  # the owner accumulated the contributions of EVERY rank's tokens (sum): same 1/W as the averaged replicas
            if p.grad is not None:
                p.grad.div_(self.world)
        for p in self.large_pending:
            self.handles.append((self._reduce(p.grad, True), p))
        self.large_pending = []
        flat = None
        smalls = [p for p in self.small if p.grad is not None]
        if smalls:
            flat = torch.cat([p.grad.reshape(-1).float() for p in smalls])
            self.handles.append((self._reduce(flat, True), None))
        for h, _ in self.handles:
            h.wait()
        if self._avg is None:
            for _, p in self.handles:
                if p is not None:
                    p.grad.div_(self.world)
            if flat is not None:
                flat.div_(self.world)
        if flat is not None:
            o = 0
            for p in smalls:
                n = p.numel()
                p.grad.copy_(flat[o:o + n].view_as(p.grad))
                o += n
        self.handles = []

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
    (Gradients are all-reduced, not reduce-scattered: with 288 GB of HBM the full bf16 gradient fits and the exchange overlaps
    with backward; a reduce-scatter variant halves xGMI volume and is a later optimisation.)"""

    def __init__(self, params, lr=5e-6, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1, process_group=None):
        self.params = [p for p in params if p.requires_grad]
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.step_count = 0
        self.state = []
        for p in self.params:
            n = p.numel()
            local = bool(getattr(p, "_ep_local", False))      # an expert-parallel shard: every rank holds DIFFERENT experts -> whole state here
            per = n if local else (n + self.world - 1) // self.world
            per += per & 1                                    # even shard length (kernel works on bf16 pairs)
            lo, hi = (0, n) if local else (min(n, self.rank * per), min(n, (self.rank + 1) * per))
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
        for p, st in zip(self.params, self.state):
            if p.grad is None:
                continue
            lo, hi = st["lo"], st["hi"]
            if hi > lo:
                n = hi - lo
                pf, gf = p.view(-1)[lo:hi], p.grad.reshape(-1)[lo:hi]
                if n & 1:  # odd tail (only possible for the last shard of an odd-sized tensor): torch fallback on one element
                    n -= 1
                if n:
                    ops.adamw_step_(pf[:n], gf[:n].contiguous(), st["master"][:n], st["m"][:n], st["v"][:n], lr=lr, beta1=self.betas[0],
                                    beta2=self.betas[1], eps=self.eps, weight_decay=self.weight_decay, step=self.step_count,
                                    grad_scale=grad_scale)
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


def cosine_lr(step: int, total: int, base_lr: float, warmup_ratio: float = 0.01) -> float:
    """lr_scheduler_type: cosine with warmup_ratio (recipes/config_full.yaml:27-28), HF get_cosine_schedule_with_warmup."""
    import math

    warm = max(1, int(total * warmup_ratio))
    if step < warm:
        return base_lr * step / warm
    prog = (step - warm) / max(1, total - warm)
    return base_lr * 0.5 * (1.0 + math.cos(math.pi * min(1.0, prog)))
