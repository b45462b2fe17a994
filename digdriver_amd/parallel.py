"""Multi-GPU plan: one process per GPU, bins/elements sharded, tiny per-cohort exchanges over RCCL.

The reference has no distributed path (its only multi-GPU use is single-process nn.DataParallel for the CNN,
kfold_mutations_main.py:143).  The burden-test path shards naturally:

  * bins are split into contiguous genome-ordered ranges, one per rank; an element belongs to the rank that owns
    its first overlapped bin; each rank additionally loads the few foreign bins its boundary elements touch
    (a halo), so per-element accumulation never needs remote data;
  * the only cross-rank data are per-cohort sufficient statistics (scale-factor sums, sequence-model counts):
    a few kB.  They are ALL-GATHERED and summed in rank order on every rank, so the result is bit-identical on
    all ranks and independent of the collective's internal reduction order;
  * results stay sharded (or are gathered to rank 0 for the TSV).

`torch.distributed` is used with backend "nccl" (= RCCL over xGMI) on GPUs and "gloo" in the CPU tests.
"""
import numpy as np


def bin_ranges(n_bins, world):
    """Contiguous, near-equal bin ranges [lo, hi) per rank."""
    edges = [(n_bins * r) // world for r in range(world + 1)]
    return [(edges[r], edges[r + 1]) for r in range(world)]


def plan_shards(ov_ptr, ov_idx, n_bins, world):
    """Partition elements by the owner of their first overlapped bin and build each rank's local CSR.

    Returns a list (one entry per rank) of dicts:
        elements   global element ids owned by the rank (ascending)
        bin_rows   global bin rows the rank must hold: its own range plus the halo (ascending)
        n_halo     how many of those lie outside the rank's own range
        ov_ptr / ov_idx   CSR of the owned elements, re-indexed into `bin_rows`
    Elements without any overlapped bin go to rank 0."""
    ov_ptr = np.asarray(ov_ptr, np.int64)
    ov_idx = np.asarray(ov_idx, np.int64)
    E = len(ov_ptr) - 1
    ranges = bin_ranges(n_bins, world)
    his = np.array([hi for _, hi in ranges])
    nov = np.diff(ov_ptr)
    first = np.where(nov > 0, ov_idx[np.minimum(ov_ptr[:-1], max(len(ov_idx) - 1, 0))] if len(ov_idx) else 0, 0)
    owner = np.searchsorted(his, first, side="right")
    owner = np.where(nov > 0, owner, 0)
    plans = []
    for r, (lo, hi) in enumerate(ranges):
        elts = np.flatnonzero(owner == r)
        cnt = nov[elts]
        take = (np.concatenate([np.arange(ov_ptr[e], ov_ptr[e + 1]) for e in elts]) if len(elts) and cnt.sum()
                else np.zeros(0, np.int64))
        touched = ov_idx[take]
        halo = np.unique(touched[(touched < lo) | (touched >= hi)])
        rows = np.concatenate([halo[halo < lo], np.arange(lo, hi), halo[halo >= hi]])
        local = np.searchsorted(rows, touched)
        plans.append(dict(rank=r, elements=elts, bin_rows=rows, n_halo=int(len(halo)),
                          ov_ptr=np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64), ov_idx=local.astype(np.int32)))
    return plans


def rank_ordered_sum(t, group=None):
    """All-gather `t` from every rank and add the pieces in rank order: identical bits everywhere."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return t
    pieces = [torch.empty_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(pieces, t.contiguous(), group=group)
    out = pieces[0].clone()
    for p in pieces[1:]:
        out += p
    return out


def scale_factors(local_exp_sum, local_n_snv, local_n_indel, group=None):
    """Cohort scale factors from sharded sufficient statistics (transfer_tools.py:148-156):
    cj = sum_r N_SNV_OBS_r / sum_r sum(Y_PRED[~FLAG])_r, cj_indel likewise.  One exchange of a [3, C] tensor."""
    import torch
    part = torch.stack([local_exp_sum, local_n_snv.to(local_exp_sum.dtype), local_n_indel.to(local_exp_sum.dtype)])
    return scale_factors_from_part(part, group)


def scale_factors_from_part(part, group=None, out=None):
    """Same, from the shard's statistics already laid out as one [3, C] f64 tensor (row 0 = sum(Y_PRED[~FLAG]),
    rows 1-2 = observed SNV / indel totals): all-gather, then -- on the device -- one kernel that adds the shards in
    rank order and divides (dig_scale_factors); host tensors (the gloo tests) use the same arithmetic in torch."""
    import torch
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        world = dist.get_world_size(group)
        parts = torch.empty((world,) + tuple(part.shape), dtype=part.dtype, device=part.device)
        if part.is_cuda:
            dist.all_gather_into_tensor(parts, part.contiguous(), group=group)
        else:
            dist.all_gather(list(parts.unbind(0)), part.contiguous(), group=group)
    else:
        parts = part.unsqueeze(0)
    if parts.is_cuda:
        from . import engine
        return engine.scale_factors_from_parts(parts, out=out)
    tot = parts[0].clone()
    for r in range(1, parts.shape[0]):      # rank order
        tot += parts[r]
    return tot[1] / tot[0], tot[2] / tot[0]


def gather_to_rank0(t, group=None):
    """Variable-length row gather (result frames) to rank 0; returns the concatenation on rank 0, None elsewhere."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return t
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    mx = int(max(int(s.item()) for s in sizes))
    pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    if rank != 0:
        return None
    return torch.cat([b[: int(s.item())] for b, s in zip(bufs, sizes)], dim=0)


def average_gradients(params, group=None):
    """Data-parallel gradient exchange for the CNN (replaces nn.DataParallel, kfold_mutations_main.py:143): all
    gradients are packed into ONE flat buffer, summed with a single all-reduce (RCCL ring over xGMI: one large
    message instead of ~60 small ones) and divided by the world size.  No-op without a process group."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat /= dist.get_world_size(group)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


def strided_positions(n_rows, bs, rank, world):
    """Positions, in the global visiting order of an epoch, of the rows rank `rank` processes when every batch of `bs`
    rows is dealt out as batch[rank::world] (NNTrainer.train)."""
    pos = [np.arange(j + rank, min(j + bs, n_rows), world) for j in range(0, n_rows, bs)]
    return np.concatenate(pos) if pos else np.zeros(0, np.int64)


def gather_visiting_order(t, n_rows, bs, group=None):
    """Re-assemble per-row tensors of a data-parallel epoch in the GLOBAL visiting order on every rank: each rank holds
    the rows of strided_positions(n_rows, bs, rank, world) (dim 0 of `t`); one padded all-gather, then a scatter by
    position.  The GP of the region model must see the activations of all training rows, as nn.DataParallel hands
    them to the reference (kfold_mutations_main.py:143,177), not a 1/world sample."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return t
    world = dist.get_world_size(group)
    pos = [strided_positions(n_rows, bs, r, world) for r in range(world)]
    mx = max(len(p) for p in pos)
    pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    out = torch.empty((n_rows,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    for r in range(world):
        out[torch.as_tensor(pos[r], device=t.device)] = bufs[r][: len(pos[r])]
    return out


def broadcast_module_buffers(module, src=0, group=None):
    """BatchNorm running statistics (and every other buffer) of rank `src` to all ranks: nn.DataParallel keeps the
    running statistics of replica 0 only (the other replicas' updates are discarded), so this is the reference's
    semantics -- and it keeps eval-mode outputs, and therefore every epoch-selection decision, identical on all ranks."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    for b in module.buffers():
        dist.broadcast(b, src=src, group=group)


def broadcast_flag(value, device, src=0, group=None):
    """A yes/no decision taken on rank `src`, made known to every rank."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return bool(value)
    f = torch.tensor([1 if value else 0], dtype=torch.int32, device=device)
    dist.broadcast(f, src=src, group=group)
    return bool(f.item())
