"""Expert parallelism for the Aria MoE block (BASELINE.json config #5; an ADDITION: the reference's TokenDispatcher is the
Megatron "allgather-free" LOCAL dispatcher with the all-to-all stripped, aria/model/moe_lm.py:296-365).

64 experts are sharded over the W ranks of a process group (rank g owns experts [g*E/W, (g+1)*E/W)); every rank routes its own
tokens, rows travel to the owning rank with an all-to-all(v) (RCCL over xGMI: 7/8 of the rows leave the GPU, 30 720 B per token
per layer each way), the local grouped GEMM runs over E/W experts, and a second all-to-all brings the expert outputs back to be
combined with the local scores and the (replicated) shared expert.  Expert weight gradients are complete on the owning rank
(no DP reduction when EP = world); replicated parameters (router, shared expert, attention, norms) are reduced by
``aria_amd.parallel.GradSync`` as usual.

Everything is built from small autograd Functions over the C-ABI kernels, so the same pieces are the differentiable form of
the reference's dispatcher seam.  The exchange uses ``all_to_all_single`` on RCCL and falls back to pairwise isend/irecv on
backends without all-to-all (gloo: used by the CPU tests).
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist

from . import autograd as AG
from . import functional as Fn
from . import ops

bf16 = torch.bfloat16


# --------------------------------------------------------------------------------------------- differentiable dispatcher pieces
class RouteFn(torch.autograd.Function):
    """logits -> scores (differentiable, incl. the training-only z / load-balancing loss gradients); idx, counts are integers."""

    @staticmethod
    def forward(ctx, logits, cfg: Fn.MoEConfig):
        scores, idx, counts = ops.moe_route(logits, cfg.topk)
        ctx.save_for_backward(logits, scores, idx, counts)
        ctx.cfg = cfg
        ctx.mark_non_differentiable(idx, counts)
        return scores, idx, counts

    @staticmethod
    def backward(ctx, dscores, _di, _dc):
        logits, scores, idx, counts = ctx.saved_tensors
        c = ctx.cfg
        return ops.moe_route_bwd(logits, idx, scores, AG._c(dscores), counts, c.z_loss_coeff, c.aux_loss_coeff, c.aux_scale), None


class PermuteFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, sorted_src, inv, k):
        ctx.save_for_backward(inv)
        ctx.k = k
        return ops.moe_permute(x, sorted_src, k)

    @staticmethod
    def backward(ctx, dperm):
        (inv,) = ctx.saved_tensors
        return ops.moe_unpermute(AG._c(dperm), inv, None, ctx.k), None, None, None


class UnpermuteFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eo, inv, scores, add, k):
        ctx.save_for_backward(eo, inv, scores)
        ctx.k = k
        return ops.moe_unpermute(eo, inv, scores, k, add=add)

    @staticmethod
    def backward(ctx, dout):
        eo, inv, scores = ctx.saved_tensors
        dout = AG._c(dout)
        d_eo, dscores = ops.moe_unpermute_bwd(dout, eo, inv, scores, ctx.k)
        return d_eo, None, dscores, dout, None


class GatherRowsFn(torch.autograd.Function):
    """out[i] = x[index[i]] for a PERMUTATION index (backward = gather with the inverse permutation)."""

    @staticmethod
    def forward(ctx, x, index, inverse):
        ctx.save_for_backward(index, inverse)
        return ops.moe_permute(x, index, 1)

    @staticmethod
    def backward(ctx, dy):
        index, inverse = ctx.saved_tensors
        return ops.moe_permute(AG._c(dy), inverse, 1), None, None


def _all_to_all_rows(rows: torch.Tensor, send_splits: List[int], recv_splits: List[int], group) -> torch.Tensor:
    D = rows.shape[1]
    out = torch.empty((sum(recv_splits), D), dtype=rows.dtype, device=rows.device)
    if dist.get_backend(group) == "nccl":
        dist.all_to_all_single(out, rows, recv_splits, send_splits, group=group)
        return out
    # pairwise exchange (gloo has no all-to-all)
    rank, W = dist.get_rank(group), dist.get_world_size(group)
    so = [0]
    ro = [0]
    for a, b in zip(send_splits, recv_splits):
        so.append(so[-1] + a)
        ro.append(ro[-1] + b)
    out[ro[rank]:ro[rank + 1]] = rows[so[rank]:so[rank + 1]]
    reqs, keep = [], []
    for peer in range(W):
        if peer == rank:
            continue
        if send_splits[peer]:
            chunk = rows[so[peer]:so[peer + 1]].contiguous()
            keep.append(chunk)
            reqs.append(dist.isend(chunk, dist.get_global_rank(group, peer) if group else peer, group=group))
        if recv_splits[peer]:
            buf = torch.empty((recv_splits[peer], D), dtype=rows.dtype, device=rows.device)
            reqs.append((dist.irecv(buf, dist.get_global_rank(group, peer) if group else peer, group=group), buf, peer))
    for r in reqs:
        if isinstance(r, tuple):
            r[0].wait()
            out[ro[r[2]]:ro[r[2] + 1]] = r[1]
        else:
            r.wait()
    return out


class AllToAllRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rows, send_splits, recv_splits, group):
        ctx.meta = (send_splits, recv_splits, group)
        return _all_to_all_rows(AG._c(rows), send_splits, recv_splits, group)

    @staticmethod
    def backward(ctx, dy):
        send_splits, recv_splits, group = ctx.meta
        return _all_to_all_rows(AG._c(dy), recv_splits, send_splits, group), None, None, None


# --------------------------------------------------------------------------------------------- the EP MoE block
def shard_expert_weights(fc1: torch.Tensor, fc2: torch.Tensor, rank: int, world: int):
    """Views of the local experts' weights ([E/W, D, 2I], [E/W, I, D])."""
    per = fc1.shape[0] // world
    return fc1[rank * per:(rank + 1) * per], fc2[rank * per:(rank + 1) * per]


def ep_chunks(T: int) -> int:
    """Token chunks the exchange is cut into (ARIA_EP_CHUNKS; default 1 = the whole micro-batch as one exchange): chunk i + 1's dispatch
    all-to-all runs under chunk i's grouped GEMMs, chunk i's combine under chunk i + 1's.  Never chunks of fewer than 1024 tokens.
    OPT-IN: measured on ONE MI355X (bench.py --ep at world 1, where the "exchange" is a device copy and there is no link time to hide;
    profiles/r05_bench_ep_world1.json) 2 chunks cost +5.8 % and 4 chunks +22.6 % of the step (more launches, smaller grouped GEMMs, the copies
    competing with the GEMMs for HBM) -- whether 2 chunks pay at 8 GPUs, where 7 / 8 of the rows cross xGMI, is for the first multi-GPU lease
    to measure."""
    import os

    c = max(1, int(os.environ.get("ARIA_EP_CHUNKS", "1")))
    least = max(8, int(os.environ.get("ARIA_EP_CHUNK_MIN", "1024")))   # (the tests lower it to chunk their 21-token batches)
    return max(1, min(c, T // least))


class _Streams:
    """Main / communication stream pair of one forward (CPU tensors: every method is a no-op, the code path is the same)."""

    def __init__(self, ref: torch.Tensor):
        self.on = ref.is_cuda
        if self.on:
            self.main = torch.cuda.current_stream(ref.device)
            self.comm = _side_stream(ref.device, 1)

    def on_comm(self):
        import contextlib

        return torch.cuda.stream(self.comm) if self.on else contextlib.nullcontext()

    def comm_after_main(self, *tensors):   # what follows on the comm stream sees everything enqueued on main so far
        if self.on:
            self.comm.wait_stream(self.main)
            for t in tensors:
                t.record_stream(self.comm)

    def main_after_comm(self, *tensors):
        if self.on:
            self.main.wait_stream(self.comm)
            for t in tensors:
                t.record_stream(self.main)



def _local_experts(rows: torch.Tensor, recv_counts: torch.Tensor, fc1_local, fc2_local, W: int, El: int) -> torch.Tensor:
    """fc1 + glu + fc2 of this rank's experts over the rows one exchange delivered, ordered (source rank, local expert); recv_counts
    [W, El] on the device.  -> expert outputs in the SAME row order.  A rank-local choice between two forms (no collective inside, so ranks
    may differ): the SEGMENT launches take the rows as they arrived (r04) -- v3-only, 32-bit per-lane DMA offsets, i.e.
    2 * rows * leading dimension < 2^32 for fc1's input and fc2's -- and the REORDER form (local-expert-major copy, grouped launches that
    fall back to the v2 kernels) takes anything.  The row count is data dependent (a routing skew towards this rank's experts grows it):
    the test is made here, on the count the host knows (ADVICE r4 for the whole exchange, ADVICE r5 for every chunk of the chunked one)."""
    K, K2 = fc1_local.shape[1], fc2_local.shape[1]
    R = rows.shape[0]
    seg_rows_ok = 2 * R * K < (1 << 32) and 2 * R * K2 < (1 << 32)
    if seg_rows_ok and ops.segments_supported(K) and ops.segments_supported(K2) and ops.glu_fusable(K, fc1_local.shape[2]):
        seg_off = torch.zeros(W * El + 1, dtype=torch.int32, device=rows.device)
        seg_off[1:] = torch.cumsum(recv_counts.reshape(-1), 0).to(torch.int32)
        act = AG.ExpertsGluSegFn.apply(rows, fc1_local, seg_off)
        return AG.ExpertsGemmSegFn.apply(act, fc2_local, seg_off)
    # reorder to local-expert-major -- destination segment (e, s) takes source segment (s, e)
    rcd = recv_counts.long()                                                       # [source rank, local expert] on the device
    src_start = (torch.cumsum(rcd.reshape(-1), 0) - rcd.reshape(-1)).view(W, El)   # where segment (s, e) starts in `rows`
    lens_em, src_em = rcd.t().reshape(-1), src_start.t().reshape(-1)               # expert-major order of the segments
    dst_start = torch.cumsum(lens_em, 0) - lens_em
    order = (torch.repeat_interleave(src_em - dst_start, lens_em, output_size=R) + torch.arange(R, device=rows.device)).to(torch.int32)
    inverse = torch.empty_like(order)
    inverse[order.long()] = torch.arange(R, dtype=torch.int32, device=rows.device)
    local_in = GatherRowsFn.apply(rows, order, inverse)
    local_off = torch.zeros(El + 1, dtype=torch.int32, device=rows.device)
    local_off[1:] = torch.cumsum(rcd.sum(0), 0).to(torch.int32)
    if ops.glu_fusable(K, fc1_local.shape[2]):                                     # fc1 + glu in one launch, as on the local path
        act = AG.ExpertsGluFn.apply(local_in, fc1_local, local_off)
    else:
        act = AG.SwiGLUFn.apply(AG.ExpertsGemmFn.apply(local_in, fc1_local, local_off))
    eo_local = AG.ExpertsGemmFn.apply(act, fc2_local, local_off)
    return GatherRowsFn.apply(eo_local, inverse, order)                            # (source rank, local expert) order again


def _ep_moe_forward_chunked(x, router_w, fc1_local, fc2_local, gate_w, up_w, down_w, cfg: Fn.MoEConfig, group, C: int) -> torch.Tensor:
    """The segment form of ``ep_moe_forward`` with the exchange cut into C token chunks (the Megatron dispatcher the
    reference's is derived from, moe_lm.py:296-365, exchanges the whole micro-batch at once and waits for it):

      main stream:  route / sort / permute of every chunk | GEMMs(0) | GEMMs(1) | ... | un-permute(0..C-1)
      comm stream:                                         a2a_d(0) a2a_d(1) ...  a2a_c(0)  a2a_c(1) ...
      side stream:  shared expert (whole micro-batch)

    a2a_d(i + 1) runs under GEMMs(i) and a2a_c(i) under GEMMs(i + 1); the host reads the split sizes of ALL chunks with one copy (one host
    wait per layer, hidden behind the shared expert's enqueue as before).  Autograd runs every node's backward on the stream its forward ran
    on, so the backward overlaps the same way in reverse.  Results per row are those of the unchunked form (a row's products do not depend on
    what else is in the launch); weight gradients are summed over the chunks."""
    W, rank = dist.get_world_size(group), dist.get_rank(group)
    E, k = cfg.num_experts, cfg.topk
    El = E // W
    T = x.shape[0]
    bounds = [(T * c // C) // 8 * 8 for c in range(C)] + [T]
    st = _Streams(x)
    parts = []
    for c in range(C):
        xc = x[bounds[c]:bounds[c + 1]]
        logits = AG.linear(xc, router_w)
        scores, idx, counts = RouteFn.apply(logits, cfg)
        offsets, sorted_src, inv = ops.moe_sort(idx, counts)
        parts.append(dict(scores=scores, inv=inv, counts=counts, perm=PermuteFn.apply(xc, sorted_src, inv, k)))
    send_counts = torch.stack([p["counts"].view(W, El) for p in parts], dim=1).contiguous()          # [dest rank, chunk, local expert]
    recv_counts = torch.empty_like(send_counts)                                                       # [source rank, chunk, local expert]
    if dist.get_backend(group) == "nccl":
        dist.all_to_all_single(recv_counts, send_counts, group=group)
    else:
        gathered = [torch.empty_like(send_counts) for _ in range(W)]
        dist.all_gather(gathered, send_counts, group=group)
        recv_counts = torch.stack([g[rank] for g in gathered])
    both_dev = torch.stack([send_counts, recv_counts])
    sh, side = None, None
    if x.is_cuda:
        both = torch.empty(both_dev.shape, dtype=both_dev.dtype, pin_memory=True)
        both.copy_(both_dev, non_blocking=True)
        copied = torch.cuda.Event()
        copied.record()
        side = _side_stream(x.device)
        side.wait_stream(st.main)
        with torch.cuda.stream(side):
            sh = _shared_expert(x, gate_w, up_w, down_w)
        copied.synchronize()
    else:
        both = both_dev.cpu()
    send_sp = [both[0][:, c].sum(1).tolist() for c in range(C)]
    recv_sp = [both[1][:, c].sum(1).tolist() for c in range(C)]
    # dispatch all-to-alls, in chunk order, on the comm stream (they wait for the permutes only)
    st.comm_after_main(*[p["perm"] for p in parts])
    ev_d = []
    for c, p in enumerate(parts):
        with st.on_comm():
            p["rows"] = AllToAllRowsFn.apply(p["perm"], send_sp[c], recv_sp[c], group)
            if st.on:
                e = torch.cuda.Event()
                e.record(st.comm)
                ev_d.append(e)
    ev_g = []
    for c, p in enumerate(parts):
        if st.on:
            st.main.wait_event(ev_d[c])
            p["rows"].record_stream(st.main)
        # (per chunk, on the row count this rank actually received: segment launches, or the reorder form when a skew pushed the chunk
        # past their 32-bit row offsets -- ADVICE r5: the fixed "4 x its share" estimate in ep_moe_forward is not a bound)
        p["eo_local"] = _local_experts(p["rows"], recv_counts[:, c], fc1_local, fc2_local, W, El)
        if st.on:
            e = torch.cuda.Event()
            e.record(st.main)
            ev_g.append(e)
            st.comm.wait_event(e)
            p["eo_local"].record_stream(st.comm)
        with st.on_comm():
            p["eo"] = AllToAllRowsFn.apply(p["eo_local"], recv_sp[c], send_sp[c], group)
    st.main_after_comm(*[p["eo"] for p in parts])
    if sh is None:
        sh = _shared_expert(x, gate_w, up_w, down_w)
    else:
        st.main.wait_stream(side)
        sh.record_stream(st.main)
        x.record_stream(side)
    outs = [UnpermuteFn.apply(p["eo"], p["inv"], p["scores"], sh[bounds[c]:bounds[c + 1]], k) for c, p in enumerate(parts)]
    return torch.cat(outs, dim=0)


def ep_moe_forward(x: torch.Tensor, router_w, fc1_local, fc2_local, gate_w, up_w, down_w, cfg: Fn.MoEConfig,
                   group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """MoELayer.forward with the routed experts sharded over `group`.  x [T, D] (this rank's tokens) -> [T, D].
    One small device -> host copy per call for the row counts of the all-to-all (as in Megatron's alltoall dispatcher; the reference's
    LOCAL path had two syncs per layer: ``tokens_per_expert.cpu()`` per grouped GEMM, moe_lm.py:478), hidden under the shared expert, which
    runs on a side stream concurrently with the dispatch / combine all-to-alls.  The backward re-uses the forward's split sizes."""
    W, rank = dist.get_world_size(group), dist.get_rank(group)
    E, k = cfg.num_experts, cfg.topk
    El = E // W
    assert fc1_local.shape[0] == El
    C = ep_chunks(x.shape[0])
    if (C > 1 and ops.segments_supported(fc1_local.shape[1]) and ops.segments_supported(fc2_local.shape[1])
            and ops.glu_fusable(fc1_local.shape[1], fc1_local.shape[2])):
        return _ep_moe_forward_chunked(x, router_w, fc1_local, fc2_local, gate_w, up_w, down_w, cfg, group, C)
    logits = AG.linear(x, router_w)
    scores, idx, counts = RouteFn.apply(logits, cfg)
    offsets, sorted_src, inv = ops.moe_sort(idx, counts)
    perm = PermuteFn.apply(x, sorted_src, inv, k)                                  # expert-major == destination-rank-major
    # exchange per-(rank, local expert) row counts
    send_counts = counts.view(W, El).contiguous()
    recv_counts = torch.empty_like(send_counts)                                    # [source rank, local expert]
    if dist.get_backend(group) == "nccl":
        dist.all_to_all_single(recv_counts, send_counts, group=group)
    else:
        gathered = [torch.empty_like(send_counts) for _ in range(W)]
        dist.all_gather(gathered, send_counts, group=group)
        recv_counts = torch.stack([g[rank] for g in gathered])
    # The all-to-all's split sizes are host integers in torch.distributed: ONE device -> host copy per layer forward carries both count
    # matrices (the backward re-uses them); the reorder permutation and the local offsets are built on the device.  On the GPU the copy
    # is asynchronous and the (replicated) shared expert is enqueued on a SIDE STREAM before the host waits for it: the GPU is never idle
    # behind the sync, and the shared expert's GEMMs then run UNDER the dispatch all-to-all (which is issued from the main stream and
    # waits only for `perm`) instead of after the combine.
    both_dev = torch.stack([send_counts, recv_counts])
    sh, side = None, None
    if x.is_cuda:
        both = torch.empty(both_dev.shape, dtype=both_dev.dtype, pin_memory=True)
        both.copy_(both_dev, non_blocking=True)
        copied = torch.cuda.Event()
        copied.record()
        main = torch.cuda.current_stream(x.device)
        side = _side_stream(x.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            sh = _shared_expert(x, gate_w, up_w, down_w)
        copied.synchronize()
    else:
        both = both_dev.cpu()
    send_splits, recv_splits = both[0].sum(1).tolist(), both[1].sum(1).tolist()
    rows = AllToAllRowsFn.apply(perm, send_splits, recv_splits, group)             # ordered (source rank, local expert)
    # r04: where the widths and the received row count allow it the grouped launches take the exchange's output AS IT ARRIVED -- W * El
    # segments ordered (source rank, local expert), segment g multiplying with local expert g % El -- so no row passes over [6T, D] in front
    # of fc1 or behind fc2, forward or backward; else the rows are reordered to local-expert-major and back (_local_experts)
    back = _local_experts(rows, recv_counts, fc1_local, fc2_local, W, El)
    eo = AllToAllRowsFn.apply(back, recv_splits, send_splits, group)               # my rows, original expert-major order
    # shared expert (replicated) and weighted combine
    if sh is None:
        sh = _shared_expert(x, gate_w, up_w, down_w)
    else:  # computed on the side stream under the all-to-alls: the combine waits for it (and the allocator is told who uses what)
        torch.cuda.current_stream(x.device).wait_stream(side)
        sh.record_stream(torch.cuda.current_stream(x.device))
        x.record_stream(side)
    return UnpermuteFn.apply(eo, inv, scores, sh, k)


_SIDE_STREAMS = {}


def _side_stream(device, which: int = 0):
    """Extra streams per device -- 0: the shared expert, 1: the all-to-alls of the chunked exchange (created once: stream creation is not free)."""
    key = (torch.device(device).index, which)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


def _shared_expert(x, gate_w, up_w, down_w):
    """SharedExpertMLP (moe_lm.py:368-395): gate || up as one GEMM with the SwiGLU epilogue where the width allows it."""
    if ops.glu_fusable(x.shape[1], 2 * gate_w.shape[0]):
        sact = AG.SharedGluFn.apply(x, gate_w, up_w)
    else:
        sact = AG.SwiGLUFn.apply(torch.cat([AG.linear(x, gate_w), AG.linear(x, up_w)], dim=-1))
    return AG.linear(sact, down_w)
