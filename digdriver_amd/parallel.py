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
import os

import numpy as np

# DIG_FORCE_COLLECTIVES=1 (or parallel.FORCE_COLLECTIVES = True): every exchange step below goes through the process
# group's backend even when the group has ONE rank -- a world of 1 normally skips them.  With backend "nccl" on one GPU
# this is the check that the RCCL calls of the multi-GPU paths (arguments, device placement, stream ordering) work
# (tests/test_gpu_rccl_world1.py); results must have the bits of the no-group path.
FORCE_COLLECTIVES = os.environ.get("DIG_FORCE_COLLECTIVES") == "1"


def collectives_on(group=None):
    """Do the exchange steps run?  A process group exists and has more than one rank, or FORCE_COLLECTIVES."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or FORCE_COLLECTIVES


def comm_device(device, group=None):
    """Where tensors handed to a collective must live: `device` (a GPU) under the "nccl" (= RCCL) backend, the host
    under "gloo"."""
    import torch
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_backend(group) == "nccl":
        return torch.device(device)
    return torch.device("cpu")


def bin_ranges(n_bins, world):
    """Contiguous, near-equal bin ranges [lo, hi) per rank."""
    edges = [(n_bins * r) // world for r in range(world + 1)]
    return [(edges[r], edges[r + 1]) for r in range(world)]


N_CHUNKS = 64      # canonical chunks of the bin grid: every shard boundary of 1, 2, 4, ..., 64 ranks is a chunk boundary


def canonical_chunks(n_bins, k=N_CHUNKS):
    """Boundaries floor(n_bins j / k), j = 0..k, of the canonical chunks the sufficient statistics are summed in
    (dig_scale_suffstats_chunked): the chunk sums, added first to last, give the same bits however the bins are sharded."""
    return np.array([(int(n_bins) * j) // k for j in range(k + 1)], dtype=np.int64)


def shard_inputs(w, plan, world, n_chunks=N_CHUNKS):
    """This rank's slice of a global problem `w` (host arrays in the layouts of include/dig_hip.h, e.g.
    bench.make_workload): bin-table rows (own range + halo), its elements with the CSR re-indexed into those rows, its
    canonical chunks as row offsets into the local table, and its share of the cohort totals (integer counts)."""
    assert n_chunks % world == 0, "the number of ranks must divide the %d canonical chunks" % n_chunks
    rows, elts, r = plan["bin_rows"], plan["elements"], plan["rank"]
    n_bins = w["bin_mu"].shape[0]
    lo, hi = bin_ranges(n_bins, world)[r]
    halo_lo = int(np.searchsorted(rows, lo))                    # halo rows in front of the rank's own range
    edges = canonical_chunks(n_bins, n_chunks)
    per = n_chunks // world
    own = edges[r * per:(r + 1) * per + 1]
    assert own[0] == lo and own[-1] == hi
    out = {k: np.ascontiguousarray(w[k][rows]) for k in ("bin_mu", "bin_std", "bin_y", "bin_flag", "bin_ctx")}
    for k in ("L", "strand_minus", "obs_snv", "obs_samples", "obs_indel"):
        out[k] = np.ascontiguousarray(w[k][elts])
    out["ov_ptr"], out["ov_idx"], out["d_pr"] = plan["ov_ptr"], plan["ov_idx"], w["d_pr"]
    out["chunk_rows"] = (own - lo + halo_lo).astype(np.int64)
    share = lambda tot: np.floor(np.asarray(tot, np.float64) * (r + 1) / world) - np.floor(np.asarray(tot, np.float64) * r / world)
    out["n_snv_obs"], out["n_ind_obs"] = share(w["n_snv_obs"]), share(w["n_ind_obs"])
    out["elements"] = elts
    return out


class ShardedPipeline:
    """One rank of the bin-sharded burden-test path (BASELINE configs[3]): the shard's tables resident on the
    device, scale factors through the chunked all-gather, dig_element_pipeline on the shard's elements.  The reference
    has no counterpart (one cohort per process, single GPU)."""

    def __init__(self, shard, device, group=None, n_chunks=N_CHUNKS, world=None):
        import torch
        from . import engine
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=device)
        self.td = {k: t(v) for k, v in shard.items() if isinstance(v, np.ndarray) and k not in ("chunk_rows", "elements")}
        self.elements = shard["elements"]
        E, C = shard["L"].shape[0], shard["d_pr"].shape[0]
        self.E, self.C = E, C
        self.out_acc = engine.alloc_accumulate_outputs(E, C, 1, device)
        self.out_stats = torch.empty((len(engine.ES_PLANES), E, C), dtype=torch.float64, device=device)
        td = self.td
        self.pipe = engine.PipelinePlan(td["bin_mu"], td["bin_std"], td["bin_y"], td["bin_flag"], td["bin_ctx"], td["ov_ptr"],
                                        td["ov_idx"], td["L"], td["strand_minus"], td["d_pr"], td["obs_snv"], td["obs_samples"],
                                        td["obs_indel"], out_acc=self.out_acc, out_stats=self.out_stats) if E else None
        self.scale = engine.ChunkedScaleFactorPlan(td["bin_mu"], td["bin_flag"], td["n_snv_obs"], td["n_ind_obs"],
                                                   shard["chunk_rows"], n_chunks, group=group, world=world)
        self.cj = torch.empty(C, dtype=torch.float64, device=device)
        self.cj_indel = torch.empty(C, dtype=torch.float64, device=device)

    def step(self, stream=None):
        """Scale factors (own chunk sums -> all-gather -> first-to-last sum) then the element pipeline of the shard."""
        self.scale.run(self.cj, self.cj_indel, stream=stream)
        if self.pipe is not None:
            self.pipe.run(self.cj, self.cj_indel, stages=7, stream=stream)
        return self.out_acc, self.out_stats


def csr_take(ov_ptr, ov_idx, rows):
    """The concatenated CSR entries of the rows `rows` (one vectorised gather, no per-row Python loop)."""
    ov_ptr = np.asarray(ov_ptr, np.int64)
    rows = np.asarray(rows, np.int64)
    cnt = ov_ptr[rows + 1] - ov_ptr[rows]
    out_ptr = np.concatenate([[0], np.cumsum(cnt)])
    take = np.repeat(ov_ptr[rows] - out_ptr[:-1], cnt) + np.arange(int(out_ptr[-1]))
    return np.asarray(ov_idx)[take]


def plan_one_shard(rank, elts, cnt, touched, lo, hi):
    """One rank's plan from its elements (global ids, ascending), their bin counts and the concatenation of their
    overlapped GLOBAL bin rows: own bin range [lo, hi) plus the halo, CSR re-indexed into those rows."""
    touched = np.asarray(touched, np.int64)
    halo = np.unique(touched[(touched < lo) | (touched >= hi)])
    rows = np.concatenate([halo[halo < lo], np.arange(lo, hi), halo[halo >= hi]])
    local = np.searchsorted(rows, touched)
    return dict(rank=rank, elements=np.asarray(elts, np.int64), bin_rows=rows, n_halo=int(len(halo)),
                ov_ptr=np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64), ov_idx=local.astype(np.int32))


def plan_shards(ov_ptr, ov_idx, n_bins, world, only_rank=None):
    """Partition elements by the owner of their first overlapped bin and build each rank's local CSR.

    Returns a list (one entry per rank; with `only_rank` = r a list holding rank r's plan alone) of dicts:
        elements   global element ids owned by the rank (ascending)
        bin_rows   global bin rows the rank must hold: its own range plus the halo (ascending)
        n_halo     how many of those lie outside the rank's own range
        ov_ptr / ov_idx   CSR of the owned elements, re-indexed into `bin_rows`
    Elements without any overlapped bin go to rank 0."""
    ov_ptr = np.asarray(ov_ptr, np.int64)
    ov_idx = np.asarray(ov_idx, np.int64)
    ranges = bin_ranges(n_bins, world)
    his = np.array([hi for _, hi in ranges])
    nov = np.diff(ov_ptr)
    first = np.where(nov > 0, ov_idx[np.minimum(ov_ptr[:-1], max(len(ov_idx) - 1, 0))] if len(ov_idx) else 0, 0)
    owner = np.searchsorted(his, first, side="right")
    owner = np.where(nov > 0, owner, 0)
    plans = []
    for r, (lo, hi) in enumerate(ranges):
        if only_rank is not None and r != only_rank:
            continue
        elts = np.flatnonzero(owner == r)
        plans.append(plan_one_shard(r, elts, nov[elts], csr_take(ov_ptr, ov_idx, elts), lo, hi))
    return plans


def rank_ordered_sum(t, group=None):
    """All-gather `t` from every rank and add the pieces in rank order: identical bits everywhere."""
    import torch
    import torch.distributed as dist
    if not collectives_on(group):
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
    if collectives_on(group):
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
    if not collectives_on(group):
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
    if not collectives_on(group):
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
    if not collectives_on(group):
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
    if not collectives_on(group):
        return
    for b in module.buffers():
        dist.broadcast(b, src=src, group=group)


def broadcast_flag(value, device, src=0, group=None):
    """A yes/no decision taken on rank `src`, made known to every rank."""
    import torch
    import torch.distributed as dist
    if not collectives_on(group):
        return bool(value)
    f = torch.tensor([1 if value else 0], dtype=torch.int32, device=device)
    dist.broadcast(f, src=src, group=group)
    return bool(f.item())


def chunked_scale_factors_reference(part, n_own, group=None):
    """The exchange of engine.ChunkedScaleFactorPlan on host tensors (gloo tests, small tools): `part` [n_own + 2, C] =
    this rank's chunk sums followed by its observed SNV / indel counts -> all-gather -> chunk sums added first to last in
    chunk (= rank) order, counts added in rank order -> (cj, cj_indel).  The device path does the same in
    dig_scale_factors_chunked."""
    import torch
    import torch.distributed as dist
    if collectives_on(group):
        world = dist.get_world_size(group)
        parts = [torch.empty_like(part) for _ in range(world)]
        dist.all_gather(parts, part.contiguous(), group=group)
    else:
        parts = [part]
    e = torch.zeros_like(part[0])
    for p in parts:                       # chunk order: rank r owns chunks r * n_own .. (r + 1) * n_own - 1
        for j in range(n_own):
            e = e + p[j]
    s = sum(p[n_own] for p in parts)
    d = sum(p[n_own + 1] for p in parts)
    return s / e, d / e


# ---------------------------------------------------------------------------------------------
# per-base / tiled route and CNN + GP prediction, sharded by bins (BASELINE configs[4], SURVEY 8e: "bins are independent
# for a1-a7"; nb_model.py:188-234 is a loop over bins)
# ---------------------------------------------------------------------------------------------
def shard_rows(n, rank, world):
    """Positions [lo, hi) of a list of n bins that rank `rank` of `world` takes: contiguous, genome order kept, so that the
    concatenation of the ranks' results in rank order IS the single-process result."""
    return (n * rank) // world, (n * (rank + 1)) // world


def all_gather_rows(t, group=None):
    """Variable-length row all-gather: the concatenation, in rank order, of every rank's `t` (dim 0), on EVERY rank."""
    import torch
    import torch.distributed as dist
    if not collectives_on(group):
        return t
    world = dist.get_world_size(group)
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    pad = torch.zeros((max(sizes),) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad.contiguous(), group=group)
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)], dim=0)


class ShardedTiles:
    """One rank of the per-base / tiled NB test sharded by bins (nb_model.py:188-234: one independent iteration per bin).

    Rank r takes the bins shard_rows(R, r, world) of the genome-ordered bin list with their mu / sigma columns, the slab
    of the packed genome those bins need (PackedGenome.slab: coordinates shifted per chromosome, same positions, contexts
    and chromosome-end clipping as on the whole genome) and the mutations that start inside the slab.  The three launches
    (dig_base_tile_probs -> dig_tile_mut_counts -> dig_tiled_nb_test) need nothing from another rank: results have the
    single-device bits.  The one exchange of the route is the Benjamini-Hochberg pass (get_q_vals, nb_model.py:340-342),
    which ranks ALL p-values of a cohort: q_values() all-gathers the cohort's valid-tile p-values in rank order (= the
    single-process frame order) and every rank keeps the q-values of its own tiles."""

    def __init__(self, genome, chroms, starts, ends, s_prob, mu, sigma, mut_chrom, mut_start, mut_end, mut_cohort, binsize,
                 device, rank, world, group=None):
        self.rank, self.world, self.group, self.device, self.binsize = int(rank), int(world), group, device, int(binsize)
        chroms = np.asarray([str(c) for c in chroms])
        starts, ends = np.asarray(starts, np.int64), np.asarray(ends, np.int64)
        R = len(chroms)
        self.R_total = R
        self.lo, self.hi = shard_rows(R, rank, world)
        # tiles per bin: the same on every rank (what the longest bin of the WHOLE list needs)
        self.n_tiles = int(max(1, -(-int((ends - starts).max() if R else 1) // self.binsize)))
        sl = slice(self.lo, self.hi)
        self.chroms = chroms[sl]
        self.genome, self.shift = genome.slab(self.chroms, starts[sl], ends[sl]) if self.hi > self.lo else (genome, None)
        if self.hi > self.lo:
            ci = genome.chrom_index(self.chroms)
            self.reg_shift = self.shift[ci]
            self.starts, self.ends = starts[sl] - self.reg_shift, ends[sl] - self.reg_shift
        else:
            self.reg_shift = np.zeros(0, np.int64)
            self.starts, self.ends = starts[sl], ends[sl]
        self.s_prob = np.asarray(s_prob, np.float64)
        self.C = self.s_prob.shape[0]
        assert self.s_prob.ndim == 2 and self.s_prob.shape[1] in (64, 1024), "s_prob must be [C, 64] or [C, 1024]"
        from . import engine
        engine.check_tile_regions(starts[sl], ends[sl], 1 if self.s_prob.shape[1] == 64 else 2)   # TRUE coordinates
        self.mu = np.ascontiguousarray(np.asarray(mu, np.float64).reshape(self.C, R)[:, sl])
        self.sigma = np.ascontiguousarray(np.asarray(sigma, np.float64).reshape(self.C, R)[:, sl])
        # mutations that start inside the slab (a row counts in the tile that holds its START, nb_model.py:160-163)
        mc = np.asarray(mut_chrom).astype(str)
        ms, me, co = np.asarray(mut_start, np.int64), np.asarray(mut_end, np.int64), np.asarray(mut_cohort, np.int32)
        if self.hi > self.lo and len(mc):
            known = {n.replace("chr", ""): i for i, n in enumerate(genome.names)}
            mi = np.array([known.get(c.replace("chr", ""), -1) for c in mc])
            ok = mi >= 0
            sh = np.where(ok, self.shift[np.maximum(mi, 0)], 0)
            ln = np.where(ok, self.genome.lengths[np.maximum(mi, 0)], 0)
            keep = ok & (ms - sh >= 0) & (ms - sh < ln)
            self.mut = (mc[keep], (ms - sh)[keep], (me - sh)[keep], co[keep])
            self.mut_chrom_index = mi[keep].astype(np.int64)
        else:
            self.mut = (mc[:0], ms[:0], me[:0], co[:0])
            self.mut_chrom_index = np.zeros(0, np.int64)
        self.result = self._mut_dev = None

    def run(self):
        """The rank's tiles: dict of device tensors pval, exp, pt [C, R_r, n_tiles], k i32, first_pos [R_r] (TRUE
        coordinates), n_valid [R_r].  The tensors belong to the object and are rewritten by the next run()."""
        import torch
        from . import engine
        dev, C, Rr = self.device, self.C, self.hi - self.lo
        if Rr == 0:
            z = lambda dt: torch.empty((C, 0, self.n_tiles), dtype=dt, device=dev)
            self.result = dict(pval=z(torch.float64), exp=z(torch.float64), pt=z(torch.float64), k=z(torch.int32),
                               first_pos=torch.empty(0, dtype=torch.int64, device=dev), n_valid=torch.empty(0, dtype=torch.int32, device=dev))
            return self.result
        from . import _lib
        from .data_tools import tabulate_gpu
        if self._mut_dev is None:
            # everything a step needs goes to the device ONCE: the slab, the region table (as join blocks too), the rank's
            # mutations, the model; a step is then five C-ABI calls with device pointers
            ci = self.genome.chrom_index(self.chroms)
            words, off, ln = self.genome.on_device(dev)
            t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
            _, ms, me, co = self.mut
            self._mut_dev = dict(words=words, off=off, ln=ln, rc=t(ci), rs=t(self.starts), re=t(self.ends), S=t(self.s_prob),
                                 mu=t(self.mu), sg=t(self.sigma), mc=t(self.mut_chrom_index), ms=t(ms), me=t(me), co=t(co),
                                 blocks=tabulate_gpu.ElementBlocks(ci, self.starts, self.ends, np.arange(Rr), Rr, dev),
                                 shift=t(self.reg_shift))
        d = self._mut_dev
        p = _lib.dev_ptr
        T = self.n_tiles
        if "pt" not in d:                            # the outputs live as long as the object: a step allocates nothing
            d["pt"] = torch.empty((C, Rr, T), dtype=torch.float64, device=dev)
            d["first"] = torch.empty(Rr, dtype=torch.int64, device=dev)
            d["nval"] = torch.empty(Rr, dtype=torch.int32, device=dev)
            d["k"] = torch.empty((C, Rr, T), dtype=torch.int32, device=dev)
            d["pval"] = torch.empty((C, Rr, T), dtype=torch.float64, device=dev)
            d["exp"] = torch.empty((C, Rr, T), dtype=torch.float64, device=dev)
        pt, first, nval, k, pval, ex = d["pt"], d["first"], d["nval"], d["k"], d["pval"], d["exp"]
        with torch.cuda.device(dev):
            n_up = 1 if d["S"].shape[1] == 64 else 2          # [C, 64] trinucleotide or [C, 1024] penta-nucleotide tables
            _lib.call("dig_base_tile_probs_ctx", p(d["words"]), d["words"].numel(), p(d["off"]), p(d["ln"]), len(self.genome.names),
                      p(d["rc"]), p(d["rs"]), p(d["re"]), Rr, p(d["S"]), C, n_up, self.binsize, T, p(pt), p(first), p(nval),
                      _lib.stream_ptr())
            pm, pb = tabulate_gpu.overlap_pairs(d["blocks"], d["mc"], d["ms"], d["me"])
            pr = d["blocks"].elt[pb.long()].to(torch.int32).contiguous()
            _lib.call("dig_tile_mut_counts", p(pm), p(pr), pm.numel(), p(d["ms"]), p(d["co"]), p(first), p(nval), self.binsize, T, Rr, C,
                      p(k), _lib.stream_ptr())
            _lib.call("dig_tiled_nb_test", p(pt), 1, p(k), p(d["mu"]), p(d["sg"]), p(pval), p(ex), C, Rr, T, _lib.stream_ptr())
        self.result = dict(pval=pval, exp=ex, pt=pt, k=k, first_pos=first + d["shift"], n_valid=nval)
        return self.result

    def valid_pvalues(self, cohort):
        """The cohort's p-values of the rank's existing tiles, in (bin, tile) order: its rows of the reference's frame."""
        import torch
        r = self.result
        t = torch.arange(self.n_tiles, device=r["pval"].device)[None, :]
        mask = t < r["n_valid"][:, None]
        return r["pval"][cohort][mask], mask

    def q_values(self, cohort, gathered=None, offset=None):
        """Benjamini-Hochberg q-values (nb_model.get_q_vals) of the rank's tiles of one cohort among the tiles of ALL ranks:
        [R_r, n_tiles], NaN where a bin has no such tile.  Default: the exchange is the sample sort of sample_sort_q_values over the
        process group (no rank ever holds another rank's whole list).  `gathered` + `offset`: the caller did the exchange
        (one-device rank walks) -- the concatenation of valid_pvalues() of all ranks in rank order, and the position of THIS
        rank's first p-value in it."""
        import torch
        from .sequence_model import nb_model
        mine, mask = self.valid_pvalues(cohort)
        out = torch.full(mask.shape, float("nan"), dtype=torch.float64, device=mine.device)
        if gathered is None:
            out[mask] = sample_sort_q_values(mine.reshape(1, -1).contiguous(), self.group)[0]
            return out
        if offset is None:
            raise ValueError("q_values(gathered=...) needs offset= (where this rank's p-values start in `gathered`)")
        everything, before = gathered, int(offset)
        q_all = nb_model.get_q_vals(everything) if everything.is_cuda else torch.as_tensor(nb_model.get_q_vals(everything.numpy()))
        out[mask] = q_all[before:before + mine.numel()]
        return out


def _sorted_rows(p):
    """Every row of a [C, n] float64 tensor ascending (NaN last) and the order: the library's radix sort on the device
    (dig_sort_rows), numpy's stable sort for host tensors (the gloo tests of the exchange)."""
    import torch
    C, n = p.shape
    if p.is_cuda:
        from . import _lib
        p = p.contiguous()
        ps = torch.empty_like(p)
        order = torch.empty((C, n), dtype=torch.int32, device=p.device)
        rp = np.arange(C + 1, dtype=np.int64) * n
        wsb = int(_lib.load().dig_bh_ragged_workspace(_lib.host_ptr(rp), C))
        ws = torch.empty(wsb, dtype=torch.uint8, device=p.device)
        with torch.cuda.device(p.device):
            _lib.call("dig_sort_rows", _lib.dev_ptr(p), _lib.host_ptr(rp), C, _lib.dev_ptr(ps), _lib.dev_ptr(order), _lib.dev_ptr(ws), wsb,
                      _lib.stream_ptr())
        return ps, order.long()
    a = p.numpy()
    order = np.argsort(a, axis=1, kind="stable")
    return torch.from_numpy(np.take_along_axis(a, order, 1)), torch.from_numpy(order)


def _bh_ranges(rows, row_ptr, n_global, rank0):
    """(q, row_min) of ragged rows that are the ranks rank0 + 1 .. of lists of n_global values, without the ranks behind them:
    dig_bh_qvalues_ragged on the device, the same operations in numpy for host tensors."""
    import torch
    from .sequence_model import nb_model
    if rows.is_cuda:
        return nb_model.bh_ragged(rows, row_ptr, n_global=n_global, rank0=rank0, want_row_min=True)
    a = rows.numpy()
    q = np.empty_like(a)
    rmin = np.full(len(row_ptr) - 1, np.inf)
    for r in range(len(row_ptr) - 1):
        x = a[row_ptr[r]:row_ptr[r + 1]]
        if x.size == 0:
            continue
        order = np.argsort(x, kind="stable")
        v = x[order] / ((rank0[r] + np.arange(1, x.size + 1)) / float(n_global[r]))
        v = np.minimum.accumulate(v[::-1])[::-1]
        rmin[r] = v[0]
        out = np.empty_like(v)
        out[order] = np.minimum(v, 1.0)
        q[row_ptr[r]:row_ptr[r + 1]] = out
    return torch.from_numpy(q), torch.from_numpy(rmin)


def sample_sort_q_values(p, group=None, samples=64):
    """Benjamini-Hochberg q-values (nb_model.get_q_vals, nb_model.py:340-342) of this rank's p-values among the p-values of ALL
    ranks, for every cohort at once: p [C, n_local] float64 (device tensor under RCCL, host tensor under gloo) -> q [C, n_local].

    A sample sort: every rank sorts its own lists (dig_sort_rows), `samples` evenly spaced values per cohort and rank make the
    world - 1 splitters of a cohort (all-gather of world x C x samples doubles), one all-to-all sends every value to the rank that
    owns its range of the global order, that rank ranks its range (dig_bh_qvalues_ragged with rank0 = the number of smaller
    values and n_global = the length of the whole list) and reports the minimum of p / (rank / n) over it; the ranks' minima
    are all-gathered (world x C doubles), a range is finished with the minimum of the ranges behind it, and the q-values travel
    back the way the p-values came.  Per link (world - 1) / world of a rank's OWN values cross twice -- round 5 all-gathered every
    p-value of every rank to every rank (17 GB per genome x 37 at 8 ranks) and repeated the whole sort on each.  The result has
    the bits of the single-process form: q depends on the value of p and its global rank only (equal p-values have equal q)."""
    import torch
    import torch.distributed as dist
    from .sequence_model import nb_model
    p = p.to(torch.float64)
    C, n = p.shape
    if not collectives_on(group):
        if p.is_cuda:
            return nb_model.get_q_vals_rows(p)
        return torch.from_numpy(np.stack([nb_model.get_q_vals(row) for row in p.numpy()])) if C else p.clone()
    world, rank, dev, inf = dist.get_world_size(group), dist.get_rank(group), p.device, float("inf")
    ps, order = _sorted_rows(p)
    # ---- splitters ----
    if n > 0:
        pos = torch.clamp(torch.div(torch.arange(1, samples + 1, device=dev) * n, samples + 1, rounding_mode="floor"), max=n - 1)
        smp = torch.nan_to_num(ps[:, pos], nan=inf, posinf=inf)
    else:
        smp = torch.full((C, samples), inf, dtype=torch.float64, device=dev)
    bufs = [torch.empty_like(smp) for _ in range(world)]
    dist.all_gather(bufs, smp.contiguous(), group=group)
    pool = torch.sort(torch.cat(bufs, dim=1), dim=1).values                    # [C, world * samples]: the same on every rank
    split = pool[:, samples * torch.arange(1, world, device=dev)] if world > 1 else pool[:, :0]
    search = torch.nan_to_num(ps, nan=inf, posinf=inf)                          # (a NaN counts as +inf: behind every splitter or with the +inf)
    cut = torch.searchsorted(search, split.contiguous(), right=True) if n > 0 else torch.zeros((C, world - 1), dtype=torch.int64, device=dev)
    bounds = torch.cat([torch.zeros((C, 1), dtype=torch.int64, device=dev), cut, torch.full((C, 1), n, dtype=torch.int64, device=dev)], 1)
    counts = (bounds[:, 1:] - bounds[:, :-1]).contiguous()                      # [C, world]: what goes to every rank
    cbufs = [torch.empty_like(counts) for _ in range(world)]
    dist.all_gather(cbufs, counts, group=group)
    counts_all = torch.stack(cbufs).cpu().numpy()                               # [source, C, destination]
    bnd = bounds.cpu().numpy()
    # ---- the values to their ranges: destination-major, cohort-minor ----
    send = torch.cat([ps[c, bnd[c, k]:bnd[c, k + 1]] for k in range(world) for c in range(C)]) if C else ps.reshape(-1)
    send_counts = [int(counts_all[rank, :, k].sum()) for k in range(world)]
    recv_counts = [int(counts_all[s, :, rank].sum()) for s in range(world)]
    recv = torch.empty(sum(recv_counts), dtype=torch.float64, device=dev)
    dist.all_to_all_single(recv, send.contiguous(), recv_counts, send_counts, group=group)
    # received: source-major, cohort-minor -> rows: cohort-major, source-minor
    seg = counts_all[:, :, rank]                                                # [source, C]
    src_off = np.concatenate([[0], np.cumsum(seg.reshape(-1))])                 # offsets in `recv`, (s, c) order
    pieces = [(c, s, int(src_off[s * C + c]), int(seg[s, c])) for c in range(C) for s in range(world)]
    rows = torch.cat([recv[o:o + m] for (_, _, o, m) in pieces]) if pieces else recv
    row_ptr = np.concatenate([[0], np.cumsum(seg.sum(0))]).astype(np.int64)
    n_global = counts_all.sum((0, 2)).astype(np.float64)
    rank0 = counts_all[:, :, :rank].sum((0, 2)).astype(np.int64)
    q_rows, row_min = _bh_ranges(rows.contiguous(), row_ptr, n_global, rank0)
    # ---- the ranges behind: their minima, last rank first (np.minimum: a NaN makes everything in front of it NaN) ----
    mbufs = [torch.empty(C, dtype=torch.float64, device=dev) for _ in range(world)]
    dist.all_gather(mbufs, row_min.to(dev).contiguous(), group=group)
    mins = torch.stack(mbufs).cpu().numpy()
    carry = np.full(C, inf)
    for k in range(world - 1, rank, -1):
        carry = np.minimum(mins[k], carry)
    carry_t = torch.repeat_interleave(torch.as_tensor(carry, device=dev), torch.as_tensor(np.diff(row_ptr), device=dev))
    q_rows = torch.minimum(q_rows, carry_t)
    # ---- and back: source-major again, then this rank's sorted lists, then the places the p-values came from ----
    row_off = row_ptr[:-1][:, None] + np.concatenate([np.zeros((C, 1), np.int64), np.cumsum(seg.T, 1)[:, :-1]], 1) if C else np.zeros((0, world), np.int64)
    back = torch.cat([q_rows[int(row_off[c, s]):int(row_off[c, s]) + int(seg[s, c])] for s in range(world) for c in range(C)]) if C else q_rows
    got = torch.empty(sum(send_counts), dtype=torch.float64, device=dev)
    dist.all_to_all_single(got, back.contiguous(), send_counts, recv_counts, group=group)
    dst_off = np.concatenate([[0], np.cumsum(counts_all[rank].T.reshape(-1))])  # offsets in `got`, (k, c) order
    q_sorted = torch.cat([got[int(dst_off[k * C + c]):int(dst_off[k * C + c + 1])] for c in range(C) for k in range(world)]).reshape(C, n) if C else got.reshape(C, n)
    out = torch.empty_like(q_sorted)
    out.scatter_(1, order, q_sorted)
    return out


def _sharded_tiles_q_values_all(self, max_elements=None):
    """q_values() for ALL cohorts at once: [C, R_r, n_tiles] (NaN where a bin has no such tile): one sample sort over the process
    group for the rank's valid-tile p-values of every cohort (sample_sort_q_values) -- the same bits as q_values(c) for every c
    and as the single-process form.  (`max_elements` is accepted for callers of round 5 and ignored: nothing is gathered.)"""
    import torch
    r = self.result
    t = torch.arange(self.n_tiles, device=r["pval"].device)[None, :]
    mask = t < r["n_valid"][:, None]
    C = r["pval"].shape[0]
    flat = r["pval"].reshape(C, -1)
    if int(mask.sum()) == mask.numel():                         # every bin is whole: the lists are the planes as they lie
        return sample_sort_q_values(flat, self.group).reshape(r["pval"].shape)
    idx = mask.reshape(-1).nonzero().squeeze(1)                 # (a boolean index per cohort plane would look for these 37 times)
    out = torch.full(r["pval"].shape, float("nan"), dtype=torch.float64, device=flat.device)
    out.reshape(C, -1).index_copy_(1, idx, sample_sort_q_values(flat.index_select(1, idx), self.group))
    return out


ShardedTiles.q_values_all = _sharded_tiles_q_values_all


def standardisation_stats(X, y, group=None, comm=None):
    """Feature means / population standard deviations and label mean / std over the rows of ALL ranks -- what sklearn's
    StandardScaler and y.mean() / y.std() give the reference on the whole training set (gp_trainer.py:107-120) -- from
    this rank's rows X [n_r, d], y [n_r]: two passes (sums, then squared deviations from the global means), each ONE
    rank-ordered sum of d + 2 numbers (all-gather + first-to-last add: the same bits on every rank).  Zero-variance
    columns keep scale 1.  `comm`: device the two small vectors are exchanged on (comm_device(): the GPU under RCCL); the
    arithmetic itself stays on the host in float64.  Returns (mean [d], std [d], y_mean, y_std, n)."""
    import torch
    X = torch.as_tensor(np.ascontiguousarray(X), dtype=torch.float64)
    y = torch.as_tensor(np.ascontiguousarray(y), dtype=torch.float64).reshape(-1)
    comm = torch.device("cpu") if comm is None else comm
    total = lambda v: rank_ordered_sum(v.to(comm), group).cpu()
    s1 = total(torch.cat([torch.tensor([float(X.shape[0])], dtype=torch.float64), X.sum(0), y.sum().reshape(1)]))
    n = float(s1[0])
    mean, y_mean = s1[1:-1] / n, s1[-1] / n
    s2 = total(torch.cat([((X - mean) ** 2).sum(0), ((y - y_mean) ** 2).sum().reshape(1)]))
    std = torch.sqrt(s2[:-1] / n)
    std = torch.where(std == 0, torch.ones_like(std), std)
    return mean.numpy(), std.numpy(), float(y_mean), float(torch.sqrt(s2[-1] / n)), int(n)


def broadcast_state(state, idx_feat, device, comm_device, src=0, group=None, dtype=None):
    """A fitted GP predictor (dict of tensors, SparseGP.predictor_state) and the kept feature columns from rank `src` to every
    rank: shapes first (the receivers do not know m or d), then the payload as one flat buffer."""
    import torch
    import torch.distributed as dist
    if not collectives_on(group):
        return state, idx_feat
    keys = ("Z", "L", "LB", "c", "scalars")
    rank = dist.get_rank(group)
    dtype = dtype or torch.float64
    hdr = torch.zeros(3, dtype=torch.int64, device=comm_device)
    if rank == src:
        hdr = torch.tensor([state["Z"].shape[0], state["Z"].shape[1], len(idx_feat)], dtype=torch.int64, device=comm_device)
    dist.broadcast(hdr, src=src, group=group)
    m, d, nf = (int(v) for v in hdr.tolist())
    shapes = {"Z": (m, d), "L": (m, m), "LB": (m, m), "c": (m,), "scalars": (4,)}
    total = sum(int(np.prod(shapes[k])) for k in keys) + nf
    if rank == src:
        flat = torch.cat([state[k].reshape(-1).to(torch.float64) for k in keys] +
                         [torch.as_tensor(np.asarray(idx_feat), dtype=torch.float64, device=state["Z"].device)]).to(comm_device)
    else:
        flat = torch.empty(total, dtype=torch.float64, device=comm_device)
    dist.broadcast(flat, src=src, group=group)
    out, off = {}, 0
    for k in keys:
        cnt = int(np.prod(shapes[k]))
        out[k] = flat[off:off + cnt].reshape(shapes[k]).to(device=device, dtype=dtype)
        off += cnt
    return out, flat[off:off + nf].cpu().numpy().astype(np.int64)
