#!/usr/bin/env python
"""bench.py -- genomic elements tested per second on MI355X (BASELINE.json metric).

One "step" = one pass of the burden-test hot path over one batch of synthetic input that is
already resident in HBM:

    per-cohort sufficient statistics -> (RCCL all-gather when N > 1) -> scale factors cj
    dig_element_pipeline = dig_accumulate_elements (genic_driver_tools.py:300-431 for all elements x cohorts)
                         + dig_element_stats (transfer_tools.py:272-302,343-344,473-482,594-615,731-747,1086-1087)
                           as one operation (rate sums fused into the statistics kernel; every output of both written)

Workload at N=1: BASELINE.json configs[2] ("whole genome, 37 cohorts batched, 1 MI355X"; the
metric is quoted on whole-genome x 37 cohorts and this fits one GPU): 288 000 10-kb bins,
37 cohorts, 120 091 elements (20 091 gene-like + 100 000 noncoding), K = 192 substitution types,
synthetic data per SURVEY 8d.  One "element tested" = one (element, cohort) pair.

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by torch.distributed.run (one rank per GPU, RCCL).  --mode (BASELINE configs[3]: "whole genome x 37
cohorts bin-sharded across 8 GPUs, RCCL all-gather of sufficient stats"):
    sharded   (default) ONE genome of 288 000 bins cut into contiguous bin ranges, one per rank (+ the halo of foreign bins
              its boundary elements touch); the element set grows with N (120 091 per GPU, each element on the rank that
              owns its first bin), so per-GPU work is fixed ("weak").  Every step all-gathers the chunk sums of the
              per-cohort sufficient statistics and forms the scale factors from all of them (first-to-last sum over 64
              canonical chunks: identical bits for every N, tests/test_gpu_sharded.py).  N = 1 is configs[2] exactly.
    strong    the same sharding of ONE configs[2] problem (120 091 elements in all): strong scaling, extra curve.
    replicas  every rank its own whole-genome problem, no exchange inside the step.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# HIP multiplexes streams onto a few hardware queues (4 by default); with RCCL's own streams in the process the side
# stream of the step would share the main stream's queue and serialise behind it.  Must be set before HIP starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12   # B/s, MI355X_MICROARCH.md chip table


# --------------------------------------------------------------------------------------
# synthetic workload: seeded, no reference code or data needed
# --------------------------------------------------------------------------------------
ELEMENT_BLOCK = 8192     # the per-element arrays are drawn block by block, every block from its own seeded stream


def _workload_tables(n_bins, n_elements, n_cohorts, seed, window, max_blocks):
    """The parts of the synthetic problem every rank needs whole and that are cheap to draw (about a second for the whole
    genome x 37 cohorts): the bin tables [N, C], the 1-D element / block arrays, the cohort parameters."""
    rng = np.random.default_rng(seed)
    N, E, C = n_bins, n_elements, n_cohorts
    # bins: 22 "chromosomes" of equal size, genome ordered
    per_chrom = (N + 21) // 22
    bin_chrom = (np.arange(N) // per_chrom + 1).astype(np.int32)
    bin_start = ((np.arange(N) % per_chrom) * window).astype(np.int64)
    bin_mu = rng.gamma(9.0, 3.0, (N, C))
    bin_std = rng.gamma(4.0, 1.0, (N, C))
    bin_y = rng.poisson(bin_mu).astype(np.int32)
    bin_flag = (rng.uniform(size=(N, 1)) < 0.1).repeat(C, axis=1).astype(np.uint8)
    ctx_p = rng.dirichlet(np.ones(64))
    bin_ctx = rng.multinomial(window, ctx_p, size=N).astype(np.int32)
    # elements: 1..max_blocks blocks of 200-3000 bp, sorted by genome position
    first_bin = np.sort(rng.integers(0, N, E))
    nblk = rng.integers(1, max_blocks + 1, E)
    blk_ptr = np.concatenate([[0], np.cumsum(nblk)]).astype(np.int64)
    nb_tot = int(blk_ptr[-1])
    owner = np.repeat(np.arange(E), nblk)
    blen = rng.integers(200, 3000, nb_tot)
    gap = rng.integers(0, 4000, nb_tot)
    step = blen + gap
    cs = np.cumsum(step) - step                         # running offset inside each element
    off = cs - cs[blk_ptr[owner]]
    base = bin_start[first_bin][owner] + rng.integers(0, window, E)[owner]
    blk_start = base + off
    blk_end = blk_start + blen
    # keep every block inside its chromosome's bin table
    chrom_end = ((np.bincount(bin_chrom, minlength=24)[bin_chrom[first_bin]]) * window)[owner]
    over = np.maximum(blk_end - chrom_end, 0)
    shift = np.zeros(E, np.int64)
    np.maximum.at(shift, owner, over)
    blk_start = np.maximum(blk_start - shift[owner], 0)
    blk_end = blk_start + blen
    elt_len = np.zeros(E, np.int64)
    np.add.at(elt_len, owner, blen)
    strand_minus = (rng.uniform(size=E) < 0.5).astype(np.uint8)
    d_pr = rng.dirichlet(np.ones(192), size=C) * 1e-6 * 192
    cj = rng.uniform(0.2, 3.0, C)
    cj_indel = rng.uniform(0.02, 0.3, C)
    # per-cohort totals used by the genome-mode scale factor (transfer_tools.py:148-156)
    exp_unflagged = (bin_mu * (bin_flag == 0)).sum(axis=0)
    return dict(bin_mu=bin_mu, bin_std=bin_std, bin_y=bin_y, bin_flag=bin_flag, bin_ctx=bin_ctx, bin_chrom=bin_chrom,
                bin_start=bin_start, ctx_p=ctx_p, first_bin=first_bin, blk_ptr=blk_ptr, blk_start=blk_start, blk_end=blk_end,
                elt_chrom=bin_chrom[first_bin], elt_len=elt_len, strand_minus=strand_minus, d_pr=d_pr, cj=cj, cj_indel=cj_indel,
                n_snv_obs=np.rint(exp_unflagged * cj),      # so that N_SNV_OBS / sum(Y_PRED[~FLAG]) ~= cj
                n_ind_obs=np.rint(exp_unflagged * cj_indel), window=window, seed=seed)


def _element_blocks(t, blocks):
    """Everything per element for the elements of the blocks `blocks` (ascending; block b = elements b * ELEMENT_BLOCK ...):
    the CSR of overlapped GLOBAL bin rows, the element context counts L and the observed counts.  A block is always drawn
    whole and from its own stream (seeded by (seed, block)), with parameters that depend on the global tables only: an
    element has the same values whoever draws its block -- the whole problem on one rank, or a shard on one of eight.
    Observed counts: OBS ~ NB(alpha, p) under the model itself (Gamma-Poisson with the element's accumulated
    alpha = MU^2 / SIGMA^2 and theta = SIGMA^2 / MU * cj), 1 % planted 5x drivers.
    Returns (element ids, ov_ptr, ov_idx, L [n, 1, 192], obs_snv, obs_samples, obs_indel [n, C])."""
    from digdriver_amd import engine
    E, C = len(t["first_bin"]), t["bin_mu"].shape[1]
    blocks = np.asarray(blocks, np.int64)
    elts = np.concatenate([np.arange(b * ELEMENT_BLOCK, min((b + 1) * ELEMENT_BLOCK, E)) for b in blocks]) if len(blocks) \
        else np.zeros(0, np.int64)
    bp = t["blk_ptr"]
    cnt = bp[elts + 1] - bp[elts]
    ptr = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    take = np.repeat(bp[elts] - ptr[:-1], cnt) + np.arange(int(ptr[-1]))
    ov_ptr, ov_idx = engine.ideal_overlaps(t["elt_chrom"][elts], ptr, t["blk_start"][take], t["blk_end"][take], t["window"],
                                           t["bin_chrom"], t["bin_start"])
    n = len(elts)
    L64 = np.empty((n, 64), np.int32)
    obs = [np.empty((n, C), np.int32) for _ in range(3)]
    row0 = 0
    for b in blocks:
        lo, hi = int(b) * ELEMENT_BLOCK, min((int(b) + 1) * ELEMENT_BLOCK, E)
        m = hi - lo
        rng = np.random.default_rng([int(t["seed"]), 7919, int(b)])
        p0, p1 = int(ov_ptr[row0]), int(ov_ptr[row0 + m])
        idx, starts = ov_idx[p0:p1], (ov_ptr[row0:row0 + m] - p0)
        seg_mu = np.add.reduceat(t["bin_mu"][idx], starts, axis=0)
        seg_var = np.add.reduceat(t["bin_std"][idx] ** 2, starts, axis=0)
        nbin = np.diff(ov_ptr[row0:row0 + m + 1])[:, None]
        frac = (t["elt_len"][lo:hi][:, None] / (nbin * float(t["window"])))          # ~ P_SUM
        alpha = seg_mu ** 2 / seg_var
        theta = seg_var / seg_mu
        out = slice(row0, row0 + m)
        L64[out] = rng.poisson(np.outer(t["elt_len"][lo:hi], t["ctx_p"]))
        driver = np.where(rng.uniform(size=(m, 1)) < 0.01, 5.0, 1.0)
        obs[0][out] = rng.poisson(rng.gamma(alpha, theta * t["cj"][None, :] * frac) * driver)
        obs[1][out] = rng.binomial(obs[0][out], 0.93)
        obs[2][out] = rng.poisson(rng.gamma(alpha, theta * t["cj_indel"][None, :] * frac) * driver)
        row0 += m
    L = np.repeat(L64, 3, axis=1)[:, None, :].astype(np.int32)
    return elts, ov_ptr, ov_idx, L, obs[0], obs[1], obs[2]


_SHARED_KEYS = ("bin_mu", "bin_std", "bin_y", "bin_flag", "bin_ctx", "d_pr", "cj", "cj_indel", "n_snv_obs", "n_ind_obs", "window")


def make_workload(n_bins=288_000, n_elements=120_091, n_cohorts=37, seed=3, window=10_000, max_blocks=3):
    """The whole synthetic problem: a dict of host arrays in the HBM layouts of include/dig_hip.h."""
    t = _workload_tables(n_bins, n_elements, n_cohorts, seed, window, max_blocks)
    n_blocks = (n_elements + ELEMENT_BLOCK - 1) // ELEMENT_BLOCK
    _, ov_ptr, ov_idx, L, obs_snv, obs_samples, obs_indel = _element_blocks(t, np.arange(n_blocks))
    w = {k: t[k] for k in _SHARED_KEYS}
    w.update(ov_ptr=ov_ptr, ov_idx=ov_idx, L=L, strand_minus=t["strand_minus"], obs_snv=obs_snv, obs_samples=obs_samples,
             obs_indel=obs_indel)
    return w


def make_shard_workload(rank, world, n_bins=288_000, n_elements=120_091, n_cohorts=37, seed=3, window=10_000, max_blocks=3):
    """Rank `rank`'s shard of make_workload(n_bins, n_elements, ...) without building the whole problem: exactly what
    parallel.shard_inputs(make_workload(...), parallel.plan_shards(...)[rank], world) returns (tests/test_distributed_gloo.py),
    from the cheap global tables plus the element blocks that hold the rank's elements.  An element belongs to the rank
    that owns its first overlapped bin; that bin lies at most a few bins in front of the element's anchor bin, so the
    rank's elements are among those anchored in its own bin range or just behind it."""
    from digdriver_amd import parallel
    t = _workload_tables(n_bins, n_elements, n_cohorts, seed, window, max_blocks)
    lo, hi = parallel.bin_ranges(n_bins, world)[rank]
    fb = t["first_bin"]
    margin = 64                                           # bins; an element's blocks are moved left by less than its own span
    a, b = int(np.searchsorted(fb, lo, side="left")), int(np.searchsorted(fb, hi + margin, side="left"))
    blocks = np.arange(a // ELEMENT_BLOCK, (max(b, a + 1) - 1) // ELEMENT_BLOCK + 1) if b > a else np.zeros(0, np.int64)
    elts, ov_ptr, ov_idx, L, obs_snv, obs_samples, obs_indel = _element_blocks(t, blocks)
    nov = np.diff(ov_ptr)
    assert (nov > 0).all(), "an element without bins belongs to rank 0 wherever it lies: build the whole problem instead"
    first = ov_idx[ov_ptr[:-1]]
    mine = np.flatnonzero((first >= lo) & (first < hi))
    plan = parallel.plan_one_shard(rank, elts[mine], nov[mine], parallel.csr_take(ov_ptr, ov_idx, mine), lo, hi)
    w = {k: t[k] for k in _SHARED_KEYS}
    w.update(L=L, strand_minus=t["strand_minus"], obs_snv=obs_snv, obs_samples=obs_samples, obs_indel=obs_indel)
    # shard_inputs indexes the per-element arrays with GLOBAL element ids: hand it views that accept them
    local = {k: _Rebased(w[k], elts) for k in ("L", "obs_snv", "obs_samples", "obs_indel")}
    w.update(local)
    out = parallel.shard_inputs(w, plan, world)
    out["cj"], out["cj_indel"] = t["cj"], t["cj_indel"]
    return out, plan


class _Rebased:
    """array[global element ids] for an array that holds the rows of the ascending id list `ids` only."""

    def __init__(self, rows, ids):
        self.rows, self.ids = rows, ids

    def __getitem__(self, want):
        pos = np.searchsorted(self.ids, want)
        assert np.array_equal(self.ids[pos], want)
        return self.rows[pos]


def committed_traffic(kernel_prefixes, exclude=()):
    """HBM bytes per launch of the dominant operation from the committed rocprofv3 PMC summary of this same
    command (profiles/rNN_traffic.json, made by tools/make_profile_summary.py: separate --pmc FETCH_SIZE /
    WRITE_SIZE passes, KiB units, FETCH_SIZE x2 on gfx950).  None when no summary is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    tot = sum(v["hbm_bytes"] for k, v in d["kernels"].items() if any(p in k for p in kernel_prefixes) and not any(x in k for x in exclude))
    return (tot or None), os.path.basename(files[-1])


def committed_mfma_busy(kernel_substr):
    """Share of a kernel's cycles in which its matrix pipes were busy, from the committed PMC summary (profiles/rNN_valu.json:
    SQ_VALU_MFMA_BUSY_CYCLES over SQ_BUSY_CYCLES-equivalent GRBM_GUI_ACTIVE).  None when the summary has no such counter."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_valu.json")))
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    for k, v in d["kernels"].items():
        if kernel_substr in k and v.get("SQ_VALU_MFMA_BUSY_CYCLES") and v.get("GRBM_GUI_ACTIVE"):
            # busy cycles summed over the 1024 SIMDs against the kernel's cycles (GRBM_GUI_ACTIVE is summed over the 8 XCDs)
            return (v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (v["GRBM_GUI_ACTIVE"] / 8.0), os.path.basename(files[-1])
    return None, None


def committed_valu_frac(kernel_substr):
    """FP64-VALU utilisation of a kernel from the committed PMC summary (profiles/rNN_valu.json): cycles its SIMDs spent
    issuing VALU instructions (SQ_ACTIVE_INST_VALU, quad-cycles, / 1024 SIMDs) over the kernel's cycles (GRBM_GUI_ACTIVE,
    summed over the 8 XCDs).  None when no summary is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_valu.json")))
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    for k, v in d["kernels"].items():
        if kernel_substr in k and v.get("SQ_ACTIVE_INST_VALU") and v.get("GRBM_GUI_ACTIVE"):
            return (v["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0) / (v["GRBM_GUI_ACTIVE"] / 8.0), os.path.basename(files[-1])
    return None, os.path.basename(files[-1])


def algorithmic_bytes(E, C, nbar_ov):
    """SURVEY 8d definitions (unfused)."""
    acc = E * (784 + 260 * nbar_ov) + E * C * (21 * nbar_ov + 32)
    stats = 100.0 * E * C
    return acc, stats


# --------------------------------------------------------------------------------------
def cpu_baseline(w, sample_elements, want_seconds=15.0):
    """Time the oracle (numpy + the same scipy ufuncs the reference calls) on a bounded sample of
    the same workload, single thread.  Test infrastructure: the oracle is the checker/baseline,
    never the product."""
    from oracle import dig_oracle as O
    E = w["L"].shape[0]
    C = w["d_pr"].shape[0]
    n = min(sample_elements, E)
    ptr = w["ov_ptr"][: n + 1]
    idx = w["ov_idx"][: ptr[-1]]

    def one_pass():
        t0 = time.perf_counter()
        acc = O.accumulate_elements_fast(w["bin_mu"], w["bin_std"], w["bin_y"], w["bin_flag"], w["bin_ctx"], ptr, idx,
                                         w["L"][:n], w["strand_minus"][:n].astype(bool), w["d_pr"])
        O.element_stats(acc["MU"], acc["SIGMA"], acc["P"][:, 0, :], acc["P_INDEL"][:, None], w["obs_snv"][:n],
                        w["obs_samples"][:n], w["obs_indel"][:n], w["cj"][None, :], w["cj_indel"][None, :])
        return time.perf_counter() - t0

    dt = one_pass()
    reps = 1
    total = dt
    while total < want_seconds and reps < 8:
        total += one_pass()
        reps += 1
    value = n * C * reps / total
    out = {"value": value, "unit": "element-cohort tests/s", "cores": 1, "kind": "port",
           "sample": "first %d of %d elements x %d cohorts, %d pass(es), oracle/dig_oracle.py "
                     "(vectorised numpy accumulate + scipy betainc/nbinom.pmf/chi2.sf), 1 thread" % (n, E, C, reps)}
    return out


_POOL_W = None   # workload shared with forked workers (copy-on-write)


def _oracle_slice(bounds):
    """One oracle pass over elements [a, b) of the shared workload (worker of cpu_baseline_all_cores)."""
    from oracle import dig_oracle as O
    a, b = bounds
    w = _POOL_W
    if b <= a:
        return 0
    q0, q1 = int(w["ov_ptr"][a]), int(w["ov_ptr"][b])
    ptr = w["ov_ptr"][a: b + 1] - q0
    idx = w["ov_idx"][q0:q1]
    acc = O.accumulate_elements_fast(w["bin_mu"], w["bin_std"], w["bin_y"], w["bin_flag"], w["bin_ctx"], ptr, idx,
                                     w["L"][a:b], w["strand_minus"][a:b].astype(bool), w["d_pr"])
    O.element_stats(acc["MU"], acc["SIGMA"], acc["P"][:, 0, :], acc["P_INDEL"][:, None], w["obs_snv"][a:b],
                    w["obs_samples"][a:b], w["obs_indel"][a:b], w["cj"][None, :], w["cj_indel"][None, :])
    return b - a


def _usable_cores():
    """Processes for the multi-process baseline: the reference's own default P = min(max(1, ncpu - 2), 20)
    (auxilaries/utils.py:3-8), with ncpu taken from the affinity mask and capped by a cgroup CPU quota if there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    if os.environ.get("BENCH_CPU_PROCS"):
        return max(1, int(os.environ["BENCH_CPU_PROCS"]))
    return min(max(1, n - 2), 20)


def cpu_baseline_all_cores(w, want_seconds=8.0):
    """The same oracle in P processes over contiguous element slices -- how the reference itself parallelises
    (multiprocessing.Pool over chunks, genic_driver_tools.py:447-457; P = its default, see _usable_cores).  Must run before the
    process touches the GPU (workers are forked).  Test infrastructure, never the product."""
    global _POOL_W
    import multiprocessing as mp
    cores = _usable_cores()
    E, C = w["L"].shape[0], w["d_pr"].shape[0]
    _POOL_W = w
    edges = np.linspace(0, E, cores + 1).astype(int)
    chunks = list(zip(edges[:-1], edges[1:]))
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    with mp.get_context("fork").Pool(cores) as pool:
        pool.map(_oracle_slice, [(0, min(64, E))] * cores)            # start the workers, import the oracle
        total, reps = 0.0, 0
        while total < want_seconds and reps < 20:
            t0 = time.perf_counter()
            done = sum(pool.map(_oracle_slice, chunks, chunksize=1))
            total += time.perf_counter() - t0
            reps += 1
            assert done == E
    _POOL_W = None
    return {"value": E * C * reps / total, "unit": "element-cohort tests/s", "cores": cores, "kind": "port",
            "sample": "all %d elements x %d cohorts, %d pass(es), oracle/dig_oracle.py in %d forked processes "
                      "(one contiguous element slice each)" % (E, C, reps, cores)}



# --------------------------------------------------------------------------------------
# the other rows of SURVEY 8d (gather, CNN forward, per-base tiles, context counting): short legs run AFTER the timed
# region, each timed per launch with HIP events on the stream it is launched on; reported as `aux_rooflines`
# --------------------------------------------------------------------------------------
def project_strong_scaling(dev, n_bins, n_elements, n_cohorts, seed, max_ranks=8, steps=200, warm=20):
    """A one-GPU PROJECTION of the strong split of BASELINE configs[3] (no multi-GPU box exists for this
    repository): for N = 1, 2, 4, ... max_ranks every rank's shard of the ONE configs[2] problem (make_shard_workload: its bins
    + halo, its elements) is built and its step -- parallel.ShardedPipeline.step: own chunk sums, scale factors, dot kernel,
    statistics kernel, on one stream -- is timed ON THIS GPU with HIP events; a rank's step keeps its own launch, staging and tail
    costs (the dot kernel alone carries ~18 us of them).  projected step(N) = the slowest rank's step + the cost of one RCCL
    all-gather call of the (64 / N + 2) x C doubles a rank exchanges, measured here at world 1 (launch + local copy: the ring's
    hops over xGMI are NOT in it).  Not a measurement of N GPUs."""
    import torch
    from digdriver_amd import parallel
    stream = torch.cuda.Stream(device=dev)
    out = {"what": "PROJECTION from one GPU, not a multi-GPU measurement: every rank's shard of the strong configs[3] split timed on this "
                   "GPU (parallel.ShardedPipeline.step on one stream, %d steps behind %d warm-up steps, HIP events), slowest rank + "
                   "the world-1 cost of the all-gather call" % (steps, warm), "ranks": {}}
    # the all-gather call, world 1, on a side stream of its own
    ag_us = None
    try:
        import torch.distributed as dist
        import socket
        own = not dist.is_initialized()
        if own:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
        ag_us = {}
        for N in (2, 4, 8):
            x = torch.zeros((parallel.N_CHUNKS // N + 2) * n_cohorts, dtype=torch.float64, device=dev)
            bufs = [torch.empty_like(x)]
            with torch.cuda.stream(stream):
                for _ in range(20):
                    dist.all_gather(bufs, x)
                ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ea.record(stream)
                for _ in range(200):
                    dist.all_gather(bufs, x)
                eb.record(stream)
            torch.cuda.synchronize()
            ag_us[N] = ea.elapsed_time(eb) / 200 * 1e3
        if own:
            dist.destroy_process_group()
    except Exception as exc:                                  # (no RCCL here: the projection goes without the exchange, and says so)
        out["all_gather_error"] = repr(exc)
    out["all_gather_call_us_world1"] = ag_us
    base = None
    N = 1
    while N <= max_ranks:
        per_rank = []
        for r in range(N):
            shard, _ = make_shard_workload(r, N, n_bins, n_elements, n_cohorts, seed)
            sp = parallel.ShardedPipeline(shard, dev, group=None, world=N)
            # a rank's step without its peers: its own chunk sums, the first-to-last sum over a stacked [N, 64 / N + 2, C] array
            # (its own part N times: the arithmetic of the real thing, dummy factors out) and the pipeline on the workload's own
            # scale factors, so that the statistics kernel does its usual work
            stacked = sp.scale.part.unsqueeze(0).repeat(N, 1, 1).contiguous()
            true_cj = torch.as_tensor(shard["cj"], device=dev)
            true_cji = torch.as_tensor(shard["cj_indel"], device=dev)

            def one_step():
                part = sp.scale.enqueue_part(stream)
                stacked[r].copy_(part)
                sp.scale.finish(stacked, sp.cj, sp.cj_indel, stream=stream)
                if sp.pipe is not None:
                    sp.pipe.run(true_cj, true_cji, stages=7, stream=stream)
            with torch.cuda.stream(stream):
                for _ in range(warm):
                    one_step()
                ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ea.record(stream)
                for _ in range(steps):
                    one_step()
                eb.record(stream)
            torch.cuda.synchronize()
            per_rank.append({"rank": r, "elements": int(sp.E), "bins_with_halo": int(sp.td["bin_mu"].shape[0]),
                             "step_us": ea.elapsed_time(eb) / steps * 1e3})
            del sp, shard
            torch.cuda.empty_cache()
        slowest = max(p["step_us"] for p in per_rank)
        step_us = slowest + ((ag_us or {}).get(N, 0.0) if N > 1 else 0.0)
        if N == 1:
            base = step_us
        out["ranks"][str(N)] = {"projected_step_us": step_us, "slowest_rank_step_us": slowest,
                                "all_gather_call_us": (ag_us or {}).get(N) if N > 1 else 0.0,
                                "projected_speedup": base / step_us, "projected_efficiency": base / step_us / N, "per_rank": per_rank}
        N *= 2
    return out


def aux_rooflines(dev):
    import torch
    from digdriver_amd import _lib, engine
    from digdriver_amd.data_tools.genome import PackedGenome
    from digdriver_amd.region_model.data_aux.dataset_generator import BinTrackStore
    from digdriver_amd.region_model.nets.cnn_predictors import SimpleMultiTaskResNet, flops_per_bin

    def timeit(fn, n=10, warm=2):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n * 1e-3

    def hbm(kernel, by, dt, what, **extra):
        return dict({"kernel": kernel, "bound": "hbm", "achieved": by / dt / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": by / dt / HBM_PEAK, "algorithmic_bytes_per_launch": by, "avg_launch_ms": dt * 1e3, "workload": what}, **extra)

    out = []
    g = torch.Generator(device=dev).manual_seed(2)
    # a1: per-bin track gather from the HBM-resident int16 matrix (B bins x 100 positions x 735 tracks per launch)
    N, L, T, B = 16384, 100, 735, 4096
    x16 = torch.randint(0, 101, (N, L, T), dtype=torch.int16, device=dev, generator=g)
    rows = torch.randint(0, N, (B,), dtype=torch.int64, device=dev, generator=g)
    dt = timeit(lambda: engine.gather_bins(x16, rows, None, out_dtype="f32", transpose=False))
    out.append(hbm("dig_gather_bins (all tracks, i16 -> f32, row-major: what the CNN's GEMM path consumes)", B * L * T * 6.0, dt,
                   "%d bins x %d x %d per launch; L T 2 B read + L T 4 B written per bin" % (B, L, T), bins_per_s=B / dt))
    sel = torch.sort(torch.randperm(T, device=dev, generator=g)[:512]).values.to(torch.int32)
    dt = timeit(lambda: engine.gather_bins(x16, rows, sel, out_dtype="f32", transpose=False))
    out.append(hbm("dig_gather_bins (512 of 735 tracks: gather_rows_subset_wide_kernel)", B * L * (T * 2.0 + 512 * 4.0), dt,
                   "%d bins per launch; whole source rows read, 512 tracks written" % B, bins_per_s=B / dt))
    # a3: CNN forward, fp32, T = 735, 37 heads (PyTorch-ROCm GEMMs on the MFMA units; BatchNorm folded)
    C_heads, Bc = 37, 2048
    torch.manual_seed(0)
    net = SimpleMultiTaskResNet((Bc, L, T), C_heads).eval().to(dev).fold_batchnorm()
    store = BinTrackStore(x16)
    crow = rows[:Bc].cpu().numpy()
    from digdriver_amd.region_model import predict as _predict
    _predict.tune_gemms(dev)                          # (as predict() and NNTrainer do: the GEMM shapes are timed once, in the warm-up calls)
    with torch.no_grad():
        dt = timeit(lambda: net.forward_gemm(store.batch(crow, channels_first=False)), n=5, warm=2)
    fl = float(flops_per_bin(T, C_heads)) * Bc
    out.append({"kernel": "gather + SimpleMultiTaskResNet forward (fp32, BN folded, tap-accumulated GEMMs, tuned per shape)", "bound": "mfma",
                "achieved": fl / dt / 1e12, "peak": 157.3, "unit": "TFLOP/s", "frac": fl / dt / 157.3e12,
                "algorithmic_flops_per_launch": fl, "avg_launch_ms": dt * 1e3, "workload": "%d bins, T = 735, 37 heads" % Bc,
                "bins_per_s": Bc / dt})
    del net
    # f4: one NNTrainer training step -- gather-fed, T = 735, 37 heads, batch 128: the defaults of kfold_mutations_main.py:52-76 --
    # train-mode forward, summed per-task MSE, backward, Adam (nn_trainer.py:40-91), by the REAL trainer object over 64 batches
    from digdriver_amd.region_model.trainers.nn_trainer import NNTrainer, adam_for
    import contextlib
    import io
    bs_t, n_tr = 128, 64 * 128
    torch.manual_seed(0)
    net_t = SimpleMultiTaskResNet((bs_t, L, T), C_heads).to(dev)
    opt = adam_for(net_t, dev)                        # what mutations_main / kfold_mutations_main build (fused, capturable Adam)
    lab = [np.random.default_rng(9 + c).gamma(9.0, 3.0, N) for c in range(C_heads)]
    tr = NNTrainer(net_t, opt, torch.nn.MSELoss(), bs_t, list(range(C_heads)), store, np.arange(n_tr), np.arange(n_tr, n_tr + 256), lab, dev, seed=1)
    with contextlib.redirect_stdout(io.StringIO()):
        tr.train(0)                                   # warm-up epoch (kernel selection, allocator)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.train(1)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (n_tr // bs_t)
    fl = 3.0 * float(flops_per_bin(T, C_heads)) * bs_t
    out.append({"kernel": "NNTrainer.train step: dig_gather_bins + SimpleMultiTaskResNet train-mode forward + 37 MSE losses + backward + Adam "
                          "(fp32, batch 128: the reference's defaults)", "bound": "mfma", "achieved": fl / dt / 1e12, "peak": 157.3,
                "unit": "TFLOP/s", "frac": fl / dt / 157.3e12, "algorithmic_flops_per_launch": fl, "avg_launch_ms": dt * 1e3,
                "flops_note": "3 x the forward's algorithmic flops (forward, input-gradient and weight-gradient products)",
                "workload": "one epoch of 64 batches of 128 bins, T = 735, 37 heads, host clock over the whole epoch (the trainer's "
                            "per-epoch bookkeeping included); whole batches are replayed as one captured graph (NNTrainer._graph_step)",
                "captured_graph": getattr(tr, "_graph", None) is not None, "bins_per_s": bs_t / dt, "epoch_288000_bins_s": 288_000 / (bs_t / dt)})
    del net_t, opt, tr, store, x16
    torch.cuda.empty_cache()
    # a18 back half: per-tile exact NB test, 37 cohorts x 8 000 bins x 200 tiles
    Cc, nb, nt = 37, 8000, 200
    rng = np.random.default_rng(5)
    mu = torch.as_tensor(rng.gamma(9.0, 3.0, (Cc, nb)), device=dev)
    sg = torch.as_tensor(rng.gamma(4.0, 1.0, (Cc, nb)), device=dev)
    pt = torch.as_tensor(rng.dirichlet(np.ones(nt), size=nb), device=dev)
    k = torch.poisson(mu[:, :, None] * pt[None, :, :]).to(torch.int32)
    dt = timeit(lambda: engine.tiled_nb_test(pt, k, mu, sg), n=5, warm=1)
    units = Cc * nb * nt
    out.append(hbm("dig_tiled_nb_test", 28.0 * units + 16.0 * Cc * nb, dt, "37 cohorts x 8 000 bins x 200 tiles; 28 B per (tile, cohort)",
                   tests_per_s=units / dt))
    del pt, k, mu, sg
    torch.cuda.empty_cache()
    # f3 + a18 front half on a genome-sized packed sequence (2.88 G bases)
    nwin, window = 288_000, 10_000
    nbases = nwin * window
    words = torch.randint(0, 2 ** 31 - 1, (nbases // 8 + 2,), dtype=torch.int32, device=dev, generator=g) & 0x33333333
    words[0] = 0x44444444
    words[-1] = 0x44444444
    off = torch.zeros(1, dtype=torch.int64, device=dev)
    ln = torch.full((1,), nbases, dtype=torch.int64, device=dev)
    rc = torch.zeros(nwin, dtype=torch.int32, device=dev)
    rs = torch.arange(nwin, dtype=torch.int64, device=dev) * window
    re_ = rs + window
    rm = torch.zeros(nwin, dtype=torch.uint8, device=dev)
    res = torch.empty((nwin, 64), dtype=torch.int32, device=dev)
    p = _lib.dev_ptr
    dt = timeit(lambda: _lib.call("dig_count_contexts", p(words), words.numel(), p(off), p(ln), 1, p(rc), p(rs), p(re_), p(rm), nwin,
                                  p(res), _lib.stream_ptr()), n=5, warm=1)
    out.append(hbm("dig_count_contexts (4-bit genome: the round-3 form)", nbases * 0.5 + nwin * 256.0, dt,
                   "all 288 000 10-kb windows of a 2.88 Gb packed genome; 0.5 B per base + 256 B per window", bases_per_s=nbases / dt))
    # ... and the form everything uses since round 4: 2 bits per base + the list of non-ACGT runs (here: 300 runs, as in hg19)
    words2 = torch.randint(-2 ** 31, 2 ** 31 - 1, (4 + nbases // 16 + 24,), dtype=torch.int32, device=dev, generator=g)
    n_int = 300
    ns_h = np.sort(np.random.default_rng(6).choice(nbases - 200_000, n_int, replace=False)).astype(np.int64) + 64
    ne_h = ns_h + np.random.default_rng(7).integers(1, 50_000, n_int)
    keep = np.concatenate([[True], ns_h[1:] > ne_h[:-1]])
    ns_h, ne_h = ns_h[keep], ne_h[keep]
    bk_h = np.searchsorted(ne_h, np.arange(((nbases + 64) >> 12) + 2, dtype=np.int64) << 12, side="right").astype(np.int32)
    ns_d, ne_d, bk_d = (torch.as_tensor(a, device=dev) for a in (ns_h, ne_h, bk_h))
    dt = timeit(lambda: _lib.call("dig_count_contexts2", p(words2), words2.numel(), p(ns_d), p(ne_d), len(ns_h), p(bk_d), len(bk_h), p(off),
                                  p(ln), 1, p(rc), p(rs), p(re_), p(rm), nwin, p(res), _lib.stream_ptr()), n=5, warm=1)
    out.append(hbm("dig_count_contexts2 (2-bit genome, 4-mers at even bases)", nbases * 0.25 + nwin * 256.0, dt,
                   "all 288 000 10-kb windows of a 2.88 Gb genome at 2 bits per base; 0.25 B per base + 256 B per window",
                   bases_per_s=nbases / dt))
    del res, words2
    S = torch.rand((37, 64), device=dev, generator=g, dtype=torch.float64) * 1e-2
    chunk = 36_000
    ptile = torch.empty((37, chunk, 200), dtype=torch.float64, device=dev)
    first = torch.empty(chunk, dtype=torch.int64, device=dev)
    nval = torch.empty(chunk, dtype=torch.int32, device=dev)
    dt = timeit(lambda: _lib.call("dig_base_tile_probs", p(words), words.numel(), p(off), p(ln), 1, p(rc), p(rs), p(re_), chunk, p(S), 37,
                                  50, 200, p(ptile), p(first), p(nval), _lib.stream_ptr()), n=3, warm=2)
    tiles = chunk * 200
    fl = 2.0 * 64 * 37 * tiles
    out.append({"kernel": "dig_base_tile_probs (base_tile_probs_roles_kernel: walker + multiplier waves, v_mfma_f64_16x16x4 + 4x4x4 quads)", "bound": "mfma", "achieved": fl / dt / 1e12,
                "peak": 78.6, "unit": "TFLOP/s", "frac": fl / dt / 78.6e12, "algorithmic_flops_per_launch": fl,
                "algorithmic_bytes_per_launch": chunk * window * 0.5 + tiles * 37 * 8.0,
                "hbm_frac": (chunk * window * 0.5 + tiles * 37 * 8.0) / dt / HBM_PEAK, "avg_launch_ms": dt * 1e3,
                "workload": "36 000 bins x 200 tiles x 37 cohorts per launch (an eighth of BASELINE configs[4])",
                "whole_genome_x37_ms": dt * 1e3 * nwin / chunk})
    dt3 = dt
    # the reference's DEFAULT contexts (n_up = n_down = 2, 1 024-entry tables): a gather-sum through the LDS
    S5 = torch.rand((37, 1024), device=dev, generator=g, dtype=torch.float64) * 1e-2
    dt = timeit(lambda: _lib.call("dig_base_tile_probs_ctx", p(words), words.numel(), p(off), p(ln), 1, p(rc), p(rs), p(re_), chunk, p(S5), 37,
                                  2, 50, 200, p(ptile), p(first), p(nval), _lib.stream_ptr()), n=3, warm=2)
    reads = float(chunk) * window * 37
    out.append({"kernel": "dig_base_tile_probs_ctx, n_up = 2 (penta-nucleotide contexts, the reference's default: base_tile_probs_rows_kernel, passes of 16 + 16 + 5 cohorts)",
                "bound": "lds", "achieved": reads * 8.0 / dt / 1e12, "peak": 150.0, "unit": "TB/s of LDS reads (8 B per position and cohort)",
                "frac": reads * 8.0 / dt / 150e12, "algorithmic_bytes_per_launch": chunk * window * 0.5 + tiles * 37 * 8.0,
                "hbm_frac": (chunk * window * 0.5 + tiles * 37 * 8.0) / dt / HBM_PEAK, "avg_launch_ms": dt * 1e3,
                "times_the_trinucleotide_kernel": dt / dt3, "table_reads_per_s": reads / dt,
                "workload": "36 000 bins x 200 tiles x 37 cohorts per launch; 13.3 G table reads",
                "whole_genome_x37_ms": dt * 1e3 * nwin / chunk})
    del words, ptile, first, nval, S, S5
    torch.cuda.empty_cache()
    # BASELINE configs[4] as a ROUTE (nb_model.py:126-234,340-342): tile probabilities -> interval join + dig_tile_mut_counts ->
    # dig_tiled_nb_test -> Benjamini-Hochberg q-values of every cohort, for an eighth of the genome (what one of 8 GPUs holds),
    # in tiles per second; trinucleotide tables and the reference's DEFAULT penta-nucleotide tables (n_up = n_down = 2)
    from digdriver_amd import parallel
    Rr, Cr, Wr, Br = 36_000, 37, 10_000, 50
    n_chrom = 3
    per = Rr // n_chrom
    rng = np.random.default_rng(4)
    wh = rng.integers(0, 2 ** 32, (per * Wr * n_chrom) // 8 + 2, dtype=np.uint64).astype(np.uint32) & np.uint32(0x33333333)
    wh[0] = wh[-1] = 0x44444444
    names = ["chr%d" % (i + 1) for i in range(n_chrom)]
    genome = PackedGenome(names, np.arange(n_chrom, dtype=np.int64) * per * Wr, np.full(n_chrom, per * Wr, np.int64), wh)
    chroms = np.repeat(names, per)
    starts = np.tile(np.arange(per, dtype=np.int64) * Wr, n_chrom)
    mu_r, sg_r = rng.uniform(5, 45, (Cr, Rr)), rng.uniform(1, 7, (Cr, Rr))
    M = 500_000                                                   # (4 M mutations per genome x 37 cohorts: an eighth)
    mci, msr, cor = rng.integers(0, n_chrom, M), rng.integers(0, per * Wr, M).astype(np.int64), rng.integers(0, Cr, M).astype(np.int32)
    for n_up, width in ((1, 64), (2, 1024)):
        Sr = rng.uniform(0, 1e-2, (Cr, width))
        sh = parallel.ShardedTiles(genome, chroms, starts, starts + Wr, Sr, mu_r, sg_r, np.array(names)[mci], msr, msr + 1, cor, Br, dev, 0, 1)
        sh.run()
        sh.q_values_all()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sh.run()
        torch.cuda.synchronize()
        t_run = time.perf_counter() - t0
        t0 = time.perf_counter()
        sh.q_values_all()
        torch.cuda.synchronize()
        t_q = time.perf_counter() - t0
        tiles_r = float(Rr) * (Wr // Br) * Cr
        out.append({"kernel": "per-base route (BASELINE configs[4], an eighth of the genome): dig_base_tile_probs_ctx(n_up = %d) + interval "
                              "join + dig_tile_mut_counts + dig_tiled_nb_test + Benjamini-Hochberg q-values of all 37 cohorts" % n_up,
                    "bound": "hbm", "achieved": 28.0 * tiles_r / (t_run + t_q) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                    "frac": 28.0 * tiles_r / (t_run + t_q) / HBM_PEAK, "algorithmic_bytes_per_launch": 28.0 * tiles_r,
                    "avg_launch_ms": (t_run + t_q) * 1e3, "kernels_ms": t_run * 1e3, "q_values_ms": t_q * 1e3,
                    "kernels_frac": 28.0 * tiles_r / t_run / HBM_PEAK,
                    "q_values_how": "dig_bh_qvalues_ragged: the library's radix sort of the KEYS of all 37 lists (four passes over the upper 36 bits "
                                    "+ a fix-up), the records of the reverse running minimum as a small table per list, every tile's q-value "
                                    "looked up by its own p-value",
                    "tile_cohort_tests_per_s": tiles_r / (t_run + t_q),
                    "contexts": "trinucleotide (64-entry tables)" if n_up == 1 else "penta-nucleotide (1 024-entry tables: the reference's default)",
                    "workload": "36 000 10-kb bins x 200 tiles of 50 positions x 37 cohorts, 500 000 mutations; host clock, device drained "
                                "after the kernels and after the q-values; 28 B per (tile, cohort) by SURVEY 8d",
                    "whole_genome_x37_ms_on_one_gpu": (t_run + t_q) * 1e3 * 8})
        del sh
        torch.cuda.empty_cache()
    return out

# --------------------------------------------------------------------------------------
def _flush_c_stdout():
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--bins", type=int, default=288_000)
    ap.add_argument("--elements", type=int, default=120_091)
    ap.add_argument("--cohorts", type=int, default=37)
    ap.add_argument("--cpu-sample", type=int, default=100_000, help="elements in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--mode", choices=["both", "sharded", "strong", "replicas"], default="both",
                    help="how N > 1 ranks divide the work (see the module docstring); all are configs[2] at N = 1.  both (default): "
                         "`value` on the strong problem (BASELINE configs[3]: ONE 120 091-element problem cut over the N GPUs), then "
                         "the weak curve (the element set grows with N) under `weak_scaling` in the same line")
    ap.add_argument("--contexts-on", choices=["main", "side"], default="main",
                    help="stream of the pipeline's context stage (it depends on the step's inputs only, like the scale factors)")
    ap.add_argument("--form", choices=["auto", "general"], default="auto",
                    help="auto: the plan checks L once (plan time) for the three-fold context repetition of sequence_tools.py:560-564 "
                         "and runs the 64-context form of the accumulation when it holds; general: the 192-substitution form")
    ap.add_argument("--pack-bins", type=int, default=1,
                    help="1 (default): the plans gather from the packed bin records built at plan time (dig_bin_records_pack); "
                         "0: from the four bin tables as handed in (A/B)")
    ap.add_argument("--outputs", choices=["auto", "planes", "records"], default="records",
                    help="layout of the statistics stage's ten outputs per pair.  records (default; DIG_PIPE_RECORDS: one aligned 5 120-byte "
                         "run per 64-pair tile) = what the product runs: driver_model.cohort_batch consumes them through "
                         "dig_element_records_unpack, which writes every plane cohort-major -- what its result frames read -- in one kernel.  "
                         "planes: eleven [E, C] arrays, the form of dig_element_stats (cohort_batch(output_form='planes') transposes every plane "
                         "on the device behind the pass).  auto: engine.records_form_is_faster times whole passes of both forms on THIS card "
                         "and keeps the faster (developer A/B; the pool's cards differ in which one wins).  No re-layout is part of the timed "
                         "step in any form; same bits")
    ap.add_argument("--aux", type=int, default=1,
                    help="1: after the timed region (N = 1 only) run short legs of the other SURVEY 8d kernels -- track gather, CNN "
                         "forward, per-base tiles, context counting -- and report them as aux_rooflines; 0: skip")
    ap.add_argument("--e2e", type=int, default=1,
                    help="1: after the timed region (N = 1 only) write configs[2] as FILES (37 HDF5 maps, element data, 37 mutation "
                         "files) and time the drop-in from files to 37 results.txt, stage by stage (tools/e2e_bench.py; ~40 s, 1.7 GB "
                         "under --e2e-dir); reported as e2e; 0: skip")
    ap.add_argument("--e2e-dir", default=None, help="scratch directory of the e2e leg (default: a fresh directory under the system's temp)")
    ap.add_argument("--project-ranks", type=int, default=0,
                    help="N > 1 (one GPU only): behind the timed region, build every rank's shard of the strong configs[3] split for 2, 4, ... N "
                         "ranks, time each rank's step on THIS GPU and report the slowest + the world-1 cost of the all-gather call as "
                         "projected_strong_scaling -- a projection, never part of `value` (about a minute of host time per 8 shards)")
    ap.add_argument("--side-lead", type=int, default=0,
                    help="the side stream starts the scale factors of step t when the main stream has finished step "
                         "t - SIDE_LEAD (0: free-running, the default; see DESIGN.md section 4)")
    ap.add_argument("--settle-passes", type=int, default=100,
                    help="untimed passes in the loop's two-stream form at the end of the settle phase")
    ap.add_argument("--settle-ms", type=float, default=400.0,
                    help="untimed: the sequential evaluation the loop is checked against is repeated for this long before "
                         "the W warm-up steps (brings the GPU out of its idle power state; 0 = evaluate once)")
    return ap.parse_args(argv)


def gpu_identity():
    """{PCI address: serial number} of the GPUs in sysfs (what `rocm-smi --showserial` prints) -- no child process: under
    rocprofv3 --pmc the profiler's library has initialised the GPU before this program starts, and a process in that state
    must not start another program.  The pool holds two kinds of MI355X that differ by 5-10 % on the statistics kernel
    (DESIGN.md section 8): the line says which card a number came from."""
    import glob
    out = {}
    for c in glob.glob("/sys/class/drm/card[0-9]*/device"):
        pci = os.path.basename(os.path.realpath(c)).lower()           # 0000:c5:00.0
        for name in ("serial_number", "unique_id"):
            try:
                v = open(os.path.join(c, name)).read().strip()
            except OSError:
                continue
            if v:
                out[pci] = v
                break
    return out


def gpu_serial_of(dev, serials):
    """Serial number of torch device `dev`: sysfs entry with the device's PCI address (one entry: that one)."""
    import torch
    if len(serials) == 1:
        return next(iter(serials.values()))
    try:
        p = torch.cuda.get_device_properties(dev)
        pci = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        return serials.get(pci)
    except Exception:
        return None


# the statistics kernel's own time splits the pool's cards into two groups (round 5's kernel: see DESIGN.md section 8)
STATS_KERNEL_KIND_SPLIT_US = 136.0


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): start the N ranks ourselves.  This parent has not
    imported torch and never touches a GPU -- a process that has initialised HIP must not be replaced or forked into ranks --
    it starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>` as a CHILD, relays
    rank 0's one JSON line (the last line of the child's stdout that parses as JSON) and exits with the child's code."""
    import socket
    import subprocess
    assert "torch" not in sys.modules, "the launching process must not have imported torch"
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, BENCH_LAUNCHED_BY="bench.py", BENCH_PARENT_IMPORTED_TORCH="0")
    # (the pool's host driver supports dmabuf IPC only: with the legacy mode RCCL's buffer exchange between the ranks' processes fails
    #  with `hipIpcGetMemHandle: invalid argument` -- a documented property of this image, where the variable is already exported; it is
    #  kept for a launch from an environment that lost it.  Never observed by this repository on more than one device: no such box.)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for cand in reversed(p.stdout.splitlines()):
        try:
            json.loads(cand)
            line = cand
            break
        except ValueError:
            continue
    if p.returncode != 0 or line is None:
        sys.stderr.write(p.stdout[-4000:])
        raise SystemExit(p.returncode or 1)
    print(line, flush=True)
    return 0


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("BENCH_LAUNCH_PROBE") == "fail":
        # launcher self-check: rank 1 dies before anything touches a device (the parent must relay a non-zero code and no line)
        if rank == 1:
            raise SystemExit(3)
        return
    if os.environ.get("BENCH_LAUNCH_PROBE") == "1":
        # launcher self-check (tests/test_bench_launcher.py, CPU): what a rank sees, before anything touches a device
        if rank == 0:
            peers = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                    "BENCH_LAUNCHED_BY", "BENCH_PARENT_IMPORTED_TORCH")}
            print(json.dumps({"probe": peers, "gpus": args.gpus, "torch_imported_before_main": "torch" in sys.modules}), flush=True)
        return
    saved_stdout = None
    if rank != 0:
        # only rank 0 reports: whatever the other ranks' libraries write to stdout (RCCL's banner) must not land after
        # rank 0's JSON line
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    elif world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1":
        # rank 0: RCCL prints its version banner to the C stdout; send everything written to descriptor 1 while the
        # job runs to stderr, and give the descriptor back for the one JSON line at the end
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch %d ranks (or leave WORLD_SIZE unset: bench.py starts them itself)"
                         % (args.gpus, world, args.gpus))
    ctx_gpu = gpu_identity() if rank == 0 else {}
    # (BENCH_FORCE_BOTH=1: a one-GPU check of the N > 1 control flow -- two workloads timed one after the other in one process)
    modes = [args.mode] if args.mode != "both" else (["strong", "sharded"] if world > 1 or os.environ.get("BENCH_FORCE_BOTH") == "1"
                                                     else ["sharded"])
    ctx = {"rank": rank, "local_rank": local_rank, "world": world, "gpu_serials": ctx_gpu}
    res = None
    for i, mode in enumerate(modes):
        r = run_workload(args, mode, ctx, primary=(i == 0))
        if rank == 0:
            if i == 0:
                res = r
            else:
                res["weak_scaling" if mode == "sharded" else mode] = {
                    k: r[k] for k in ("value", "unit", "ms_per_step", "scaling", "config", "roofline", "roofline_step",
                                      "matches_sequential_evaluation", "finite_pvalues")}
    # RCCL writes a version banner to the C stdout of the process, which is block-buffered when stdout is a pipe and would
    # come out AFTER the result at exit: tear the process group down and flush the C streams first, so that the JSON
    # line is the last line this process prints.
    if ctx.get("use_dist"):
        ctx["dist"].destroy_process_group()
    import ctypes
    ctypes.CDLL(None).fflush(None)
    if saved_stdout is not None:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    if rank == 0:
        print(json.dumps(res), flush=True)


def run_workload(args, mode, ctx, primary=True):
    """One workload (`mode`) on this rank: build the rank's shard, W warm-up steps, K timed steps between barriers, max over
    ranks; rank 0 returns the result dictionary.  The device and the process group are set up by the first call (ctx)."""
    import torch
    from digdriver_amd import _lib, engine, parallel
    rank, local_rank, world = ctx["rank"], ctx["local_rank"], ctx["world"]
    # The global problem and this rank's shard of it (at N = 1 the shard is the whole problem)
    sharded = mode != "replicas" and world > 1
    n_elements_global = args.elements * (world if mode == "sharded" else 1)
    if sharded:
        # only this rank's shard is built: the cheap global tables + the element blocks that hold its elements
        w, plan = make_shard_workload(rank, world, args.bins, n_elements_global, args.cohorts, seed=args.seed)
    else:
        w_global = make_workload(args.bins, args.elements, args.cohorts, seed=args.seed + rank)
        plan = parallel.plan_shards(w_global["ov_ptr"], w_global["ov_idx"], args.bins, 1, only_rank=0)[0]
        w = parallel.shard_inputs(w_global, plan, 1)
        w["cj"], w["cj_indel"] = w_global["cj"], w_global["cj_indel"]
        del w_global
    E_total = n_elements_global if sharded else args.elements * world
    # CPU baselines first: the all-core one forks workers, which must happen before this process touches the GPU
    cpu_res = (None, None)
    if world == 1 and args.cpu_sample > 0 and primary:
        cpu_res = (cpu_baseline(w, args.cpu_sample), cpu_baseline_all_cores(w))
    if "dev" not in ctx:
        _lib.require_device()
        torch.cuda.set_device(local_rank)
        ctx["dev"] = torch.device("cuda", local_rank)
        ctx["dist"] = None
        # BENCH_FORCE_DIST=1 takes the N > 1 code path (process group, all-gather, barrier, max-reduce) with whatever world
        # size the environment gives, also 1: a one-GPU check that the RCCL calls of the multi-GPU path work
        ctx["use_dist"] = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"
        if ctx["use_dist"]:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            dist.init_process_group("nccl", device_id=ctx["dev"])
            ctx["dist"] = dist
            if os.environ.get("BENCH_FORCE_DIST") == "1":
                parallel.FORCE_COLLECTIVES = True       # a world of one sends its all-gather through RCCL too
    dev, dist, use_dist = ctx["dev"], ctx["dist"], ctx["use_dist"]

    E, C = w["L"].shape[0], w["d_pr"].shape[0]
    N = w["bin_mu"].shape[0]                          # rows this rank holds (own range + halo)
    N_own = int(w["chunk_rows"][-1] - w["chunk_rows"][0])
    nbar = float(len(w["ov_idx"])) / E
    # what ONE evaluation costs a process that has nothing on the device yet (a real elementDriver call runs one step per map):
    # upload, plan, first run -- timed here, before anything else has run (`one_shot` in the line)
    torch.zeros(1, device=dev)
    torch.cuda.synchronize()
    t_one = time.perf_counter()
    td = {k: torch.as_tensor(v, device=dev) for k, v in w.items() if isinstance(v, np.ndarray) and k not in ("chunk_rows", "elements")}
    torch.cuda.synchronize()
    one_shot = {"h2d_ms": (time.perf_counter() - t_one) * 1e3, "h2d_bytes": int(sum(v.numel() * v.element_size() for v in td.values()))}
    out_stats = torch.empty((len(engine.ES_PLANES), E, C), dtype=torch.float64, device=dev)
    out_acc = engine.alloc_accumulate_outputs(E, C, 1, dev)
    # the shard's statistics for the scale factors as one [3, C] tensor: row 0 is filled by dig_scale_suffstats each
    # step, rows 1-2 hold the observed SNV / indel totals of the cohorts (inputs).  The scale factors of a step depend on
    # that step's inputs only, so they are formed on a side stream that never waits for the main stream: every step has
    # its own (tiny: 5 x C doubles) buffer set and its own event, the main stream waits for the event of its step before
    # the statistics stage.  (With two alternating sets the side stream had to be released by an event recorded on the
    # main stream; that record is a barrier packet and cost a 7 us bubble per step in front of the dot kernel, and the
    # reduction, squeezed beside the dot kernel, finished 5 us after it: rocprofv3 kernel trace, 0.301 -> 0.290 ms.)
    SIDE_LEAD_STEPS = 3                      # the side stream is enqueued this many steps ahead of the main stream (slack for the all-gather at N > 1)
    # a short run (the driver's --steps 20 is 3.6 ms of GPU time) is followed by an UNTIMED-by-the-driver loop of 1 000 more steps of
    # the same kind, reported beside it as ms_per_step_1000 (primary workload only)
    EXTRA_STEPS = 1000 if (args.steps < 100 and primary) else 0
    RING = args.steps + args.warmup + EXTRA_STEPS + SIDE_LEAD_STEPS + 2      # one set of scale-factor buffers per step: the side stream never waits
    cj_outs = [(torch.empty(C, dtype=torch.float64, device=dev), torch.empty(C, dtype=torch.float64, device=dev))
               for _ in range(RING)]
    main_stream = torch.cuda.current_stream(dev)
    side_stream = torch.cuda.Stream(device=dev, priority=-1 if use_dist else 0)     # (-1: a hardware queue of its own even when RCCL holds streams too; alone, the normal priority is 1 - 2 us better)
    side_done = [torch.cuda.Event() for _ in range(RING)]   # scale factors of a step are ready
    main_done = [torch.cuda.Event() for _ in range(RING)]   # the main stream has finished a step (paces the side stream)
    step_no = [0]

    # argument marshalling once, outside the loop (a step is then a handful of ctypes calls: the host stays ahead)
    # --contexts-on side: the context stage of a step (strand-permuted context counts of the elements + the dot kernel's
    # parameter table: inputs only) runs on the side stream behind that step's scale factors, up to 32 steps ahead of the
    # main stream, which is left with the dot kernel and the statistics.  What the stage writes must then exist once per
    # step in flight: a ring of plans, each with its own workspace (30 MB) and its own R_SIZE column; all other outputs
    # are written by the main stream and shared.
    ctx_side = args.contexts_on == "side"
    PLAN_RING = 32 if ctx_side else 1

    use_records = [args.outputs == "records" and bool(args.pack_bins)]

    def make_plan(k, records=None):
        acc_k = out_acc if k == 0 else dict(out_acc, R_SIZE=torch.empty_like(out_acc["R_SIZE"]))
        return engine.PipelinePlan(td["bin_mu"], td["bin_std"], td["bin_y"], td["bin_flag"], td["bin_ctx"], td["ov_ptr"],
                                   td["ov_idx"], td["L"], td["strand_minus"], td["d_pr"], td["obs_snv"], td["obs_samples"],
                                   td["obs_indel"], out_acc=acc_k, out_stats=out_stats, compact=args.form == "auto",
                                   pack_bins=(pipes[0] if pipes else True) if args.pack_bins else False,      # (plan time; shared)
                                   records_out=use_records[0] if records is None else records)
    pipes = []
    torch.cuda.synchronize()
    t_one = time.perf_counter()
    pipes.append(make_plan(0))
    torch.cuda.synchronize()
    one_shot["plan_ms"] = (time.perf_counter() - t_one) * 1e3       # records packed, L checked and compacted, workspace, argument marshalling
    t_one = time.perf_counter()
    pipes[0].run(td["cj"], td["cj_indel"], stages=7, stream=torch.cuda.current_stream(dev))
    torch.cuda.synchronize()
    one_shot["first_run_ms"] = (time.perf_counter() - t_one) * 1e3  # code objects loaded, first launch of every kernel
    one_shot["total_ms"] = one_shot["h2d_ms"] + one_shot["plan_ms"] + one_shot["first_run_ms"]
    one_shot["what"] = ("fresh process, inputs in host memory: upload of every input (h2d), engine.PipelinePlan (plan), one dig_element_pipeline call "
                        "with given scale factors and its synchronisation (first_run); the timed loop below runs on warm plans")
    # --outputs auto: which layout of the statistics stage's outputs is faster on THIS card (untimed, a property of the plan
    # like the compact form: the statistics stage alone, three rounds of 25 launches of each form, HIP events)
    output_form = {"chosen": "records" if use_records[0] else "planes", "how": "--outputs %s" % args.outputs,
                   "consumer": "driver_model.cohort_batch runs the record form (dig_element_records_unpack -> cohort-major planes)"}
    # canonical-chunk form: the same bits for every sharding of the bins (dig_scale_suffstats_chunked); a "replicas" rank
    # exchanges nothing (its plan has world = 1)
    exchange = use_dist and (sharded or world == 1)
    scale_plan = engine.ChunkedScaleFactorPlan(td["bin_mu"], td["bin_flag"], td["n_snv_obs"], td["n_ind_obs"], w["chunk_rows"],
                                               parallel.N_CHUNKS, world=None if exchange else 1)

    if args.outputs == "auto" and args.pack_bins:
        try:
            # developer A/B (engine.records_form_is_faster): whole passes of a plane plan and of a record plan, timed once on this
            # card; a tie keeps the planes
            alt = make_plan(0, records=True)
            chosen, cal = engine.records_form_is_faster(pipes[0], alt, td["cj"], td["cj_indel"])
            torch.cuda.synchronize()
            use_records[0] = bool(chosen)
            output_form = {"chosen": "records" if use_records[0] else "planes",
                           "how": "--outputs auto = engine.records_form_is_faster: whole passes (dot + statistics kernel, given scale factors) "
                                  "of both forms timed once on this card before the run (untimed)",
                           "pass_us": {k: round(v, 1) for k, v in cal.items()}}
            if use_records[0]:
                pipes[0] = alt
            del alt
        except Exception as exc:                            # (the calibration must never cost the bench line)
            output_form = {"chosen": "planes", "how": "--outputs auto: calibration failed (%r)" % (exc,)}
            use_records[0] = False
    for k in range(1, PLAN_RING):
        pipes.append(make_plan(k))
    pipe = pipes[0]
    def run_pipe(plan, cj, cji, stages, stream):
        """One dig_element_pipeline call with the scale factors the side stream formed."""
        return plan.run(cj, cji, stages=stages, stream=stream)

    def enqueue_scale_factors(t):
        """Side stream: (1) chunk sums of the per-cohort sufficient statistics of this rank's bins (transfer_tools.py:148-156)
        -> (2) all-gather of [chunk sums ; observed counts] over RCCL when the bins are sharded ((64 / N + 2) x C doubles per
        rank) -> (3) first-to-last sum of all 64 chunk sums and the divisions: scale factors of step t, into buffer set t."""
        b = t % RING
        cj_out = cj_outs[b]
        with torch.cuda.stream(side_stream):
            if args.side_lead > 0 and t >= args.side_lead:
                side_stream.wait_event(main_done[(t - args.side_lead) % RING])
            if t == 0 or not os.environ.get("BENCH_NO_SIDE"):
                scale_plan.run(cj_out[0], cj_out[1], stream=side_stream)
            else:                                   # developer probe (the line says so): the first step's factors for every step
                cj_outs[b] = cj_outs[0]
            if ctx_side:
                # plan t % 32 was last used by step t - 32: the main stream's throttle event of step t - 16 (recorded
                # in front of that step) says that everything up to step t - 17 has run
                if t % THROTTLE_EVERY == 0 and t >= PLAN_RING:
                    side_stream.wait_event(throttle_events[(t // THROTTLE_EVERY - 1) % len(throttle_events)])
                which = sample_which(t)
                staged("contexts", which, lambda: pipes[t % PLAN_RING].run(cj_out[0], cj_out[1], stages=1, stream=side_stream),
                       side_stream)
            side_done[b].record(side_stream)
            if sampling[0] and t % SLACK_EVERY == SLACK_SLOT and slack_events:
                ev = slack_events.pop()                      # when this step's scale factors were ready (side stream's clock)
                ev.record(side_stream)
                slack_side[t] = ev

    queued = [-1]      # last step whose scale factors have been enqueued
    # Per-stage durations are sampled INSIDE the timed loop: on every 8th step one stage (contexts, dot or statistics) is
    # bracketed by two HIP events on the main stream.  An event record is a barrier packet (~6 us), so each step carries at
    # most one pair and seven steps in eight carry none (< 1 % of the loop time).
    # (short runs -- the driver's 20 steps -- bracket the dominant stage on every 4th step, 5 samples instead of 2, and the
    #  two smaller stages in turn on another: one bracket on every second step, about 1 % of the loop)
    # (a sampled step costs the loop ~30 us -- the timed launches wait for their own begin and end signals: every 8th step sampled was
    #  0.1845 ms per step where every 64th is 0.1815 and none 0.1807, same box -- so long runs sample every 32nd step and a short
    #  run two of its timed steps)
    SHORT = args.steps < 100
    SAMPLE_EVERY = 32 if not SHORT else 10
    if os.environ.get("BENCH_SAMPLE_EVERY"):                 # developer probe: what the stage timers cost the loop
        SAMPLE_EVERY = int(os.environ["BENCH_SAMPLE_EVERY"])
    sample_slot = {1: "contexts", 3: "dot", 5: "statistics"}
    samples = {"contexts": [], "dot": [], "statistics": []}
    sampling = [False]
    warming = [False]

    # every event of the run exists before the loop starts (HIP creates the object behind a torch event at its first
    # record, and a growing pool of them costs a one-off stall of tens of milliseconds at some point of the loop)
    # how far ahead of its consumer the side stream runs: on every 8th step (one that carries no stage bracket) the moment the
    # step's scale factors are ready is recorded on the side stream and the moment the main stream starts waiting for them on
    # the main stream; slack = the second minus the first (negative: the main stream had to wait)
    SLACK_EVERY, SLACK_SLOT = 8, 7
    slack_events = [torch.cuda.Event(enable_timing=True) for _ in range(2 * ((args.steps + args.warmup) // SLACK_EVERY + 4))]
    slack_side, slack_main = {}, {}
    n_sample_events = 2 * (4 * ((args.steps + args.warmup) // SAMPLE_EVERY + 2))
    sample_events = [torch.cuda.Event(enable_timing=True) for _ in range(n_sample_events)]
    # the dot and the statistics kernel are timed by the library's stage timers (dig_stage_timer_*, include/dig_hip.h): the
    # kernel's own begin and end, taken from its dispatch -- no event packets around the stage, the step stays ONE call.
    # (Two events around the statistics stage read 148 - 154 us on the common GPUs of the pool where rocprofv3's kernel
    #  trace of the same run says 137: the packets split the call in three and change what the side stream's kernels run
    #  beside -- profiles/r04b_*.)
    timer_pool = [engine.StageTimer() for _ in range(2 * ((args.steps + args.warmup) // SAMPLE_EVERY + 4) + 1)]
    timer_samples = {"dot": [], "statistics": []}
    for e in side_done + main_done + sample_events + slack_events:
        e.record(main_stream)
    torch.cuda.synchronize()

    THROTTLE_EVERY = 16
    throttle_events = [torch.cuda.Event() for _ in range(8)]
    for e in throttle_events:
        e.record(main_stream)
    torch.cuda.synchronize()

    def staged(name, which, fn, stream=None):
        if which != name:
            return fn()
        stream = stream or main_stream
        a, b_ = sample_events.pop(), sample_events.pop()
        a.record(stream)
        fn()
        b_.record(stream)
        samples[name].append((a, b_))

    def sample_which(t):
        """The stage bracketed on step t (None: no bracket)."""
        if not sampling[0]:
            # short runs bracket the two SMALLER stages on their (untimed) warm-up steps instead of inside the K timed ones: a
            # bracket splits the step's one call into three and adds two event packets, ~10 us of a 200-us step
            if SHORT and warming[0]:
                which = ("dot", "contexts")[t % 2]
                return "dot" if which == "contexts" and pipe.compact else which
            return None
        if SHORT:                   # short runs: statistics (the `roofline` kernel) on the timed steps 1 mod SAMPLE_EVERY
            which = "statistics" if t % SAMPLE_EVERY == 1 else None
        else:
            which = sample_slot.get(t % SAMPLE_EVERY)
        if which == "contexts" and pipe.compact:       # the compact form has no context kernel: contexts + dot are one launch
            which = "dot"
        return which

    def step():
        # One step = wait for this step's scale factors (side stream, normally long done) + ONE dig_element_pipeline
        # call on the main stream: context kernel (unless --contexts-on side put it behind the scale factors), dot kernel,
        # statistics stream pass (which finishes its own slow pairs), back to back.  On
        # the sampled steps the call is split into its stages so that one of them can be bracketed by events.  All
        # outputs of accumulation and statistics are written every step; every step computes its own scale factors
        # from the bin tables.
        t = step_no[0]
        step_no[0] += 1
        b = t % RING
        # The host enqueues a step in 0.05 ms, the GPU takes 0.23: unchecked, the host runs hundreds of steps ahead, the
        # HIP queue fills up, and the runtime then blocks the enqueue until the backlog has drained COMPLETELY -- 32 ms
        # in a 200-step run, with the GPU idle at the end of it (BENCH_TRACE=1 shows the enqueue times).  One event on
        # the main stream every 16 steps (a 5 us packet: 0.1 %) keeps the host between 32 and 48 steps ahead instead.
        if t % THROTTLE_EVERY == 0:
            k = t // THROTTLE_EVERY
            throttle_events[k % len(throttle_events)].record(main_stream)
            if k >= 3:
                throttle_events[(k - 3) % len(throttle_events)].synchronize()
        # this step's (first call only) and the coming steps' scale factors; nothing beyond the last step of the run
        while queued[0] < min(t + SIDE_LEAD_STEPS, args.warmup + args.steps + EXTRA_STEPS - 1):
            queued[0] += 1
            enqueue_scale_factors(queued[0])
        cj, cji = cj_outs[b]
        if t in slack_side and slack_events:
            ev = slack_events.pop()
            ev.record(main_stream)
            slack_main[t] = ev
        main_stream.wait_event(side_done[b])
        which = sample_which(t)
        plan = pipes[t % PLAN_RING]
        if which in ("dot", "statistics") and len(timer_pool) >= 3:      # (one timer stays for the self-test after the loop)
            # a sampled step: both kernels of the call report their own durations; the call itself is what every other step issues
            for name, stage in (("dot", _lib.DIG_PIPE_DOT), ("statistics", _lib.DIG_PIPE_STATISTICS)):
                tm = timer_pool.pop()
                tm.arm(stage)
                timer_samples[name].append(tm)
            which = None
        if ctx_side:
            if which in (None, "contexts"):
                plan.run(cj, cji, stages=2 | 4 | 8, stream=main_stream)
            else:
                staged("dot", which, lambda: plan.run(cj, cji, stages=2, stream=main_stream))
                staged("statistics", which, lambda: plan.run(cj, cji, stages=4 | 8, stream=main_stream))
        elif which is None:
            run_pipe(plan, cj, cji, 7, main_stream)
        else:
            staged("contexts", which, lambda: run_pipe(plan, cj, cji, 1, main_stream))    # context kernel
            staged("dot", which, lambda: run_pipe(plan, cj, cji, 2, main_stream))         # dot kernel
            staged("statistics", which, lambda: run_pipe(plan, cj, cji, 4 | 8, main_stream))  # statistics (header cleared by the stages=1 call)
        if args.side_lead > 0:
            main_done[b].record(main_stream)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # Untimed settle phase: the plain sequential evaluation the overlapped loop is compared with afterwards, repeated for
    # --settle-ms.  A fresh process starts from the GPU's idle power state and a cold TLB / L2: with W = 5 warm-up
    # steps (1.5 ms of GPU time) the first timed steps ran 10-15 % slower than the sustained rate (BENCH_r01: 0.301 ms
    # at --steps 20 against 0.27-0.29 ms at --steps 1000).  This is setup work, not a step: no timed step depends on it.
    # Python's cyclic collector must not run inside the loop: a full collection over the process's objects (torch, numpy,
    # thousands of events and closures) takes 30-40 ms -- 150 steps' worth of GPU time; with K = 200 it used to land in the
    # timed region every time (BENCH_TRACE=1: one 38 ms enqueue), with K = 20 or 1000 it did not.  It is collected and
    # switched off HERE, in front of the settle phase: between the settle phase and the first warm-up step it left the GPU
    # idle for 40 ms, and the first ~30 steps after such a pause run up to 15 % slower (rocprofv3 kernel trace of a
    # W = 40 run: stream pass 137 -> 154 us, dot kernel 55 -> 74 us at steps 8-22 of the loop, back to normal by step 34;
    # the barrier in front of the timed region, 0.2 ms, does not do that).
    import gc
    gc.collect()
    gc.disable()
    seq_cj = torch.empty(C, dtype=torch.float64, device=dev)
    seq_cji = torch.empty(C, dtype=torch.float64, device=dev)
    scale_plan.run(seq_cj, seq_cji)                  # (a collective when the bins are sharded: every rank is here)
    t_settle = time.perf_counter()
    # (a plan of its own -- own outputs, own workspace, the same form of the accumulation as the loop's plans -- run in
    #  stream order on the main stream, nothing overlapped)
    seq_plan = engine.PipelinePlan(td["bin_mu"], td["bin_std"], td["bin_y"], td["bin_flag"], td["bin_ctx"], td["ov_ptr"],
                                   td["ov_idx"], td["L"], td["strand_minus"], td["d_pr"], td["obs_snv"], td["obs_samples"],
                                   td["obs_indel"], compact=args.form == "auto", pack_bins=pipe if args.pack_bins else False)
    assert seq_plan.compact == pipe.compact
    ref_acc, ref_stats = seq_plan.acc, seq_plan.stats
    n_settle = 0
    settle_cj, settle_cji = seq_cj, seq_cji
    while True:
        seq_plan.run(settle_cj, settle_cji, stages=7, stream=main_stream)
        n_settle += 1
        if n_settle % 16 == 0:
            torch.cuda.synchronize()
        if (time.perf_counter() - t_settle) * 1e3 >= args.settle_ms:
            break
    torch.cuda.synchronize()
    # ... and the side stream with it: the sequential evaluation above never touches the side stream's hardware queue, and
    # the first ~15 steps after its first use run 10 % slower (BENCH_TRACE=1: statistics-stage brackets of 196 us at timed
    # steps 5-13 of a W = 5 run against 170-177 before and after; with W = 50 none).  A fixed number of untimed passes in
    # the loop's own form (the same count on every rank: the side stream's all-gather is a collective).
    reh_cj, reh_cji = torch.empty_like(seq_cj), torch.empty_like(seq_cji)
    reh_ev = torch.cuda.Event()
    for _ in range(args.settle_passes):
        with torch.cuda.stream(side_stream):
            scale_plan.run(reh_cj, reh_cji, stream=side_stream)
            reh_ev.record(side_stream)
        main_stream.wait_event(reh_ev)
        pipe.run(seq_cj, seq_cji, stages=7, stream=main_stream)
    torch.cuda.synchronize()
    settle_ms = (time.perf_counter() - t_settle) * 1e3
    warming[0] = True
    for _ in range(args.warmup):
        step()
    warming[0] = False
    barrier()
    # The step's kernels run on two streams and overlap, so the roofline is taken for the step as a whole: two HIP events
    # on the main stream bracket the K timed steps (the main stream waits for the side stream's scale factors inside
    # every step, so its clock covers both); per-kernel durations are in the committed rocprofv3 summary.  Events inside
    # the loop would be barrier packets in the queue (~6 us each) and slow the thing being measured.
    ev_begin, ev_end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev_begin.record(main_stream)
    sampling[0] = True
    trace = [] if os.environ.get("BENCH_TRACE") == "1" else None      # developer switch: host time of every enqueue
    for _ in range(args.steps):
        if trace is not None:
            t1 = time.perf_counter()
        step()
        if trace is not None:
            trace.append(time.perf_counter() - t1)
    sampling[0] = False
    ev_end.record(main_stream)
    if trace is not None and rank == 0:
        top = sorted(range(len(trace)), key=lambda i: -trace[i])[:6]
        print("BENCH_TRACE longest enqueues (step, ms):", [(i, round(trace[i] * 1e3, 3)) for i in top], "median ms",
              round(sorted(trace)[len(trace) // 2] * 1e3, 4), file=sys.stderr)
    host_enqueue_s = time.perf_counter() - t0        # the host's share: enqueueing K steps (it must stay below dt)
    barrier()
    gc.enable()
    dt = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ms_step = ev_begin.elapsed_time(ev_end) / args.steps
    ms_step_1000 = None
    if EXTRA_STEPS:
        gc.disable()
        barrier()
        t1 = time.perf_counter()
        for _ in range(EXTRA_STEPS):
            step()
        barrier()
        d1 = time.perf_counter() - t1
        gc.enable()
        if use_dist:
            tmax = torch.tensor([d1], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            d1 = float(tmax.item())
        ms_step_1000 = d1 / EXTRA_STEPS * 1e3
    slack_us = sorted(slack_side[t].elapsed_time(slack_main[t]) * 1e3 for t in slack_main)
    stage_ms = {k: (sum(a.elapsed_time(b_) for a, b_ in v) / len(v) if v else None) for k, v in samples.items()}
    stage_n = {k: len(v) for k, v in samples.items()}
    stage_ms_raw = {}
    for k, tms in timer_samples.items():
        got = []
        for tm in tms:
            try:
                got.append(tm.read_ms())
            except _lib.DigHipError:         # the armed stage was not launched through the timed path (another kernel form): no sample
                pass
        if got:
            stage_ms_raw[k], stage_n[k] = sum(got) / len(got), len(got)
    timed_by_stage_timers = bool(stage_ms_raw)
    # What such a timer reads for a kernel that does nothing (one wave): readings carry a dispatch share of that order.  Against
    # rocprofv3's kernel trace of the same run (tools/stage_timer_check.sh, profiles/r04_stage_timer_check.txt) the statistics kernel reads
    # 6 - 9 us high, the dot kernel 2 - 3 us: avg_launch_ms is the RAW mean reading (the roofline fraction is if anything
    # understated); the empty-kernel reading is in the line beside it.
    empty_kernel_us = None
    if timer_pool:
        got = []
        for _ in range(12):
            timer_pool[0].selftest(main_stream)
            got.append(timer_pool[0].read_ms() * 1e3)
        empty_kernel_us = sorted(got[2:])[len(got[2:]) // 2]
    for k, raw in stage_ms_raw.items():
        stage_ms[k] = raw
    for tm in timer_pool + timer_samples["dot"] + timer_samples["statistics"]:
        tm.close()
    # what a bracket itself costs: the same two events around a one-element fill (a ~1.5 us kernel), after the timed region
    cal = []
    one = torch.empty(1, dtype=torch.float64, device=dev)
    with torch.cuda.stream(main_stream):
        for _ in range(24):
            ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ea.record(main_stream)
            one.zero_()
            eb.record(main_stream)
            cal.append((ea, eb))
    torch.cuda.synchronize()
    bracket_us = sorted(x.elapsed_time(y) for x, y in cal[4:])[len(cal[4:]) // 2] * 1e3
    if trace is not None and rank == 0:
        print("BENCH_TRACE statistics-stage brackets (us):", [round(a.elapsed_time(b_) * 1e3, 1) for a, b_ in samples["statistics"]][:40],
              file=sys.stderr)
    if pipe.records_out:                                     # the loop's last result: blocks -> planes (dig_element_records_unpack), then checked like the plane form
        pipes[(step_no[0] - 1) % PLAN_RING].unpack()
        torch.cuda.synchronize()
    ok = bool(torch.isfinite(out_stats[1]).all().item())
    # the overlapped loop must have produced what a plain sequential evaluation produces (bit for bit)
    torch.cuda.synchronize()
    last_cj, last_cji = cj_outs[(step_no[0] - 1) % RING]
    if not (torch.equal(last_cj, seq_cj) and torch.equal(last_cji, seq_cji)):
        raise SystemExit("bench: the scale factors of the overlapped loop and of the sequential evaluation disagree")
    same = bool(torch.equal(torch.nan_to_num(ref_stats, nan=-7.0), torch.nan_to_num(out_stats, nan=-7.0))) and \
        bool(torch.equal(ref_acc["MU"], out_acc["MU"])) and bool(torch.equal(ref_acc["P"], out_acc["P"]))
    if not same:
        raise SystemExit("bench: the overlapped step loop and the sequential evaluation disagree")
    # which of the pool's two kinds of MI355X this is: the statistics kernel's own time WITH PLANE OUTPUTS on this workload
    gpu_kind_us = (stage_ms.get("statistics") or 0.0) * 1e3 or None
    if pipe.records_out:
        gpu_kind_us = None
        try:
            pl = make_plan(0, records=False)
            with torch.cuda.stream(main_stream):
                pl.run(td["cj"], td["cj_indel"], stages=7, stream=main_stream)
                ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ea.record(main_stream)
                for _ in range(8):
                    pl.run(td["cj"], td["cj_indel"], stages=4, stream=main_stream)
                eb.record(main_stream)
            torch.cuda.synchronize()
            gpu_kind_us = ea.elapsed_time(eb) / 8 * 1e3
            del pl
        except Exception:                                    # (never at the cost of the bench line)
            gpu_kind_us = None
    ws = getattr(pipe, "ws", None)                            # the plan's own workspace
    slow_frac = None
    if ws is not None:
        off = (_lib.workspace_bytes("accumulate", E, C) + 255) // 256 * 256
        hdr = ws[off:off + 16].view(torch.int32)     # header [2] + [3]: pairs finished from the worklist / from the LDS queues
        slow_frac = float(int(hdr[2].item()) + int(hdr[3].item())) / (E * C)

    if rank == 0:
        units = float(E_total) * C * args.steps
        b_acc, b_stat = algorithmic_bytes(E, C, nbar)
        b_suff = 9.0 * N_own * C                                 # sufficient statistics: Y_PRED f64 + FLAG u8 per (own bin, cohort)
        # SURVEY 8d's algorithmic bytes, split by the stage that moves them (the three parts add up to b_acc + b_stat):
        stage_bytes = {"contexts": E * (260.0 * nbar + 4), "dot": E * 780.0 + 8.0 * E * C,
                       "statistics": E * C * (21.0 * nbar + 24 + 100)}
        if pipe.compact:            # one launch does the work of both accumulation stages (SURVEY's unfused count is kept)
            stage_bytes["dot"] += stage_bytes["contexts"]
        stage_kernels = {"contexts": ["acc_region"], "dot": ["acc_dot"], "statistics": ["element_stats_"]}
        default_shape = (args.bins, args.elements, args.cohorts) == (288_000, 120_091, 37) and world == 1

        def roof(name, by, ms, prefixes, n):
            if not ms:
                return None
            # (the trace of the command holds BOTH forms of the statistics kernel -- the sequential reference evaluation runs the plane
            #  form: only the form the timed steps ran is counted)
            other_form = ("3, false", "0, false") if pipe.records_out else ("3, true",)      # (template arguments follow the flag)
            traffic, src = committed_traffic(prefixes, other_form) if default_shape else (None, None)
            ach = by / (ms * 1e-3) / 1e9
            return {"bound": "hbm", "kernel": name, "achieved": ach, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                    "frac": ach / (HBM_PEAK / 1e9), "traffic": traffic, "traffic_source": src,
                    "algorithmic_bytes_per_launch": by, "avg_launch_ms": ms, "launches_timed": n}

        stage_names = {
            "statistics": "dig_element_pipeline statistics stage: element_stats_stream_fused_kernel (one launch: the "
                          "stream pass and, at the end of every workgroup, the pairs it could not finish in passing); outputs as %s"
                          % ("tile-blocked records (DIG_PIPE_RECORDS: one aligned 5 120-byte run per 64-pair tile)" if pipe.records_out else
                             "eleven planes"),
            "contexts": "dig_element_pipeline contexts stage: acc_region_kernel",
            "dot": ("dig_element_pipeline accumulation (contexts + dot in one launch): acc_dot_ctx_kernel, the 64-context form for "
                    "context-repeated L -- checked and compacted to [E, 64] ONCE at plan time (dig_element_pipeline_prepare), "
                    "not per step; v_mfma_f64_16x16x4_f64 + v_mfma_f64_4x4x4_f64 for the last 5 cohorts") if pipe.compact else
                   "dig_element_pipeline dot stage: acc_dot_mfma_kernel (v_mfma_f64_16x16x4_f64 + v_mfma_f64_4x4x4_f64 for the last 5 cohorts)"}
        stage_roofs = {k: roof(stage_names[k], stage_bytes[k], stage_ms[k], stage_kernels[k], stage_n[k])
                       for k in ("statistics", "contexts", "dot")}
        if stage_roofs["dot"] is not None and default_shape:
            # (HBM-bound by its bytes: 137.7 MB algorithmic; what its matrix pipe does is reported beside it from the committed
            #  counters -- round 3 priced this stage at SURVEY's 768 flops per pair against the matrix peak, three times what the
            #  compact form issues: that read as "at peak" and was not)
            mf, src = committed_mfma_busy("acc_dot")
            stage_roofs["dot"]["mfma_busy_frac"], stage_roofs["dot"]["mfma_busy_source"] = mf, src
        d_bytes = b_acc + b_stat + b_suff
        step_roof = roof("step = dig_scale_factors || dig_element_pipeline (both streams, overlapped)", d_bytes, ms_step,
                         ["acc_region", "acc_dot", "element_stats_", "suffstats", "scale_factors"], args.steps)
        if step_roof is not None:
            # SURVEY 8d's unfused definitions (above) next to what the step's kernels are asked to read and write: L as the
            # plan-time [E, 64] copy instead of [E, 192], the pre-masked rate table without its flag bytes, P written once and
            # read once between the two kernels, the packed bin records (20 B per gathered (bin, cohort) instead of 21)
            nnz = float(len(w["ov_idx"]))
            if pipe.compact:
                actual = (E * (256.0 + 260.0 * nbar + 4 + 12) + 8.0 * E * C) + (E * C * (8.0 + 20.0 * nbar + 24 + 100)) + 8.0 * N_own * C
            else:
                actual = d_bytes
            step_roof["bytes_the_step_moves"] = actual
            step_roof["frac_of_hbm_by_bytes_moved"] = actual / (ms_step * 1e-3) / HBM_PEAK
            step_roof["note"] = ("frac uses SURVEY 8d's unfused algorithmic bytes (L as [E, 192], FLAG bytes of the sufficient statistics); "
                                 "bytes_the_step_moves counts what the plan's kernels actually read and write")
        dominant_roof = stage_roofs["statistics"] or step_roof
        if dominant_roof is stage_roofs["statistics"] and default_shape:
            # SURVEY 8d asks for both figures of this kernel: the HBM fraction above and the FP64-VALU utilisation of its
            # streaming pass (what actually bounds it)
            vf, src = committed_valu_frac("element_stats_stream_fused_kernel<1024, true, 3, %s" % ("true" if pipe.records_out else "false"))
            dominant_roof["valu_frac"], dominant_roof["valu_frac_source"] = vf, src
        if dominant_roof is stage_roofs["statistics"]:
            if timed_by_stage_timers:
                dominant_roof["timing"] = ("avg_launch_ms: the kernel's own begin and end on the sampled steps (HIP events filled by the "
                                           "launch itself: dig_stage_timer_*, hipExtLaunchKernelGGL), i.e. what rocprofv3's kernel trace "
                                           "reports; no event packets around the stage")
                dominant_roof["timer_of_an_empty_kernel_us"] = empty_kernel_us
                dominant_roof["timing"] += ("; a reading carries the dispatch's share -- the same timer reads timer_of_an_empty_kernel_us for a "
                                            "kernel that does nothing -- and is 6 - 9 us above rocprofv3's kernel trace of the same run "
                                            "(profiles/r04_stage_timer_check.txt): frac is understated by that much, not corrected")
                dominant_roof["bracket_of_a_one_element_fill_us"] = bracket_us
                dominant_roof["bracket_note"] = ("for comparison: two events recorded around a one-element fill kernel take the figure above "
                                                 "-- the cost of bracketing a stage with packets, which rounds 1-3 included in avg_launch_ms")
            else:
                dominant_roof["bracket_of_a_one_element_fill_us"] = bracket_us
                dominant_roof["bracket_note"] = ("avg_launch_ms is the raw event-to-event time of a bracket; a bracket around a one-element "
                                                 "fill kernel takes the figure above, so ~4-6 us of avg_launch_ms are the two event packets "
                                                 "and the dispatch, not the kernel")
        res = {
            "metric": "genomic elements tested/sec (whole node), whole-genome x 37 cohorts",
            "value": units / dt, "unit": "element-cohort tests/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if (mode == "strong" and world > 1) else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[2]: whole genome, %d 10-kb bins, %d cohorts batched, %d elements, K=192 "
                                    "substitution types, mean %.2f bins/element" % (args.bins, C, E_total, nbar)) if world == 1 else
                                   {"sharded": "BASELINE configs[3]: ONE genome of %d 10-kb bins cut into %d contiguous bin ranges (one per "
                                               "GPU, + halo), %d cohorts batched, %d elements in all (%d per GPU: the element set grows "
                                               "with N), all-gather of the per-cohort chunk sums every step",
                                    "strong": "BASELINE configs[3], strong form: ONE configs[2] problem (%d bins / %d ranks, %d cohorts, %d "
                                              "elements in all, about %d per GPU), all-gather of the per-cohort chunk sums every step",
                                    "replicas": "%d-bin genome replicated on each of %d GPUs, %d cohorts, %d elements in all (%d per GPU), "
                                                "no exchange"}[mode] % (args.bins, world, C, E_total, E_total // world),
                       "mode": mode, "contexts_on": args.contexts_on,
                       "scale_factors": "DEVELOPER PROBE (BENCH_NO_SIDE): formed ONCE, not per step -- not a valid bench line" if os.environ.get("BENCH_NO_SIDE") else (
                                         "kernels of their own on a side stream, several steps ahead" + (" (chunk sums all-gathered over RCCL)" if use_dist and exchange else "")),
                       "bins": args.bins, "bins_on_rank0": N, "cohorts": C, "elements_total": E_total,
                       "elements_on_rank0": E, "parallelism": "bins sharded x%d" % world if mode != "replicas" else "replicas x%d" % world},
            "ms_per_step_1000": ms_step_1000,
            "ms_per_step_1000_note": ("the same step, 1 000 more times after the timed region (own barriers, host clock, max over "
                                      "ranks); NOT part of `value`: the asked %d steps are %.1f ms of GPU time" % (args.steps, dt * 1e3)
                                      if ms_step_1000 is not None else None),
            "gpu": {"serial": gpu_serial_of(dev, ctx["gpu_serials"]), "cards_in_sysfs": len(ctx["gpu_serials"]),
                    "kind": (None if not gpu_kind_us or not default_shape else "fast" if gpu_kind_us < STATS_KERNEL_KIND_SPLIT_US else "common"),
                    "kind_from": ("the timed loop's own stage timer" if not pipe.records_out else
                                  "eight launches of the plane form of the statistics stage behind the timed region (the loop ran the record form)"),
                    "kind_note": "the pool's MI355X fall into two groups by the statistics kernel's own time WITH PLANE OUTPUTS on this "
                                 "workload (split at %.0f us; DESIGN.md section 8): given when the run used that form; with the record "
                                 "form both groups run alike (output_form.pass_us has this card's two whole-pass times); `serial`: "
                                 "sysfs, the number rocm-smi --showserial prints" % STATS_KERNEL_KIND_SPLIT_US},
            "roofline": dominant_roof,
            "roofline_step": step_roof,
            "roofline_other_stages": [stage_roofs["contexts"], stage_roofs["dot"]],
            "operations": {
                "accumulation_form": ("compact: L repeats every context count three times (sequence_tools.py:560-564), verified on the "
                                      "device and compacted to [E, 64] at plan time; contexts + dot = acc_dot_ctx_kernel (K = 128)"
                                      if pipe.compact else "general: 192 substitution columns, acc_region_kernel + acc_dot_mfma_kernel (K = 256)"),
                "bin_tables": ("packed at plan time (dig_bin_records_pack): {Y_PRED, STD^2} + Y_TRUE | FLAG << 31 per (bin, cohort), "
                               "two gathers per overlapped bin in the statistics stage" if pipe.records is not None else
                               "the four tables as handed in: four gathers per overlapped bin"),
                "main stream": ("one dig_element_pipeline call per step: acc_dot_ctx_kernel (contexts + dot), "
                                "element_stats_stream_fused_kernel" if pipe.compact else
                                "one dig_element_pipeline call per step (stages DOT | STATISTICS): acc_dot_mfma_kernel, "
                                "element_stats_stream_fused_kernel" if ctx_side else
                                "one dig_element_pipeline call per step: acc_region_kernel (contexts + table), "
                                "acc_dot_mfma_kernel, element_stats_stream_fused_kernel"),
                "side stream": "what depends on a step's inputs only, for the coming steps (own buffers and event per step): "
                               "suffstats_chunk_stage1, suffstats_chunk_stage2 (+ all-gather of the chunk sums when N > 1), "
                               "scale_factors_chunked_kernel" +
                               (", then the pipeline's CONTEXTS stage of that step (acc_region_kernel; a ring of 32 "
                                "workspaces, at most 32 steps ahead of the main stream)" if ctx_side else ""),
                "algorithmic_bytes": {"accumulate": b_acc, "element_stats": b_stat, "scale_suffstats": b_suff}},
            "kernel_timing": "HIP events on the stream a stage is launched on (statistics and dot: main, filled by the kernel launch "
                             "itself -- dig_stage_timer_*; contexts: %s, bracketed): "
                             "`roofline` times the statistics kernel on every %d-th timed step, `roofline_other_stages` %s, "
                             "`roofline_step` brackets all %d timed steps on the main stream (side-stream work overlapped); rocprofv3 "
                             "per-kernel averages of the same command: profiles/"
                             % (args.contexts_on, SAMPLE_EVERY, "one of the smaller stages on every %d-th timed step" % SAMPLE_EVERY
                                if not SHORT else "the smaller stages on the warm-up steps (a run this short keeps them out of "
                                                          "its timed steps)", args.steps),
            "output_form": output_form,
            "one_shot": one_shot,
            "finite_pvalues": ok, "matches_sequential_evaluation": same, "slow_pair_fraction": slow_frac,
            "host_enqueue_ms_per_step": host_enqueue_s / args.steps * 1e3,
            "side_stream_slack_us": ({"min": slack_us[0], "median": slack_us[len(slack_us) // 2], "samples": len(slack_us),
                                      "what": "main stream's arrival at the wait for a step's scale factors minus the moment the side "
                                              "stream had them ready (events on both streams, every 8th timed step); negative = the "
                                              "main stream waited"} if slack_us else None),
            "untimed_settle": {"ms": settle_ms, "sequential_evaluations": n_settle, "two_stream_passes": args.settle_passes,
                               "what": "the sequential evaluation the loop is checked against, repeated before the warm-up steps"},
        }
        if args.aux and world == 1 and primary:
            del td, pipes, seq_plan
            torch.cuda.empty_cache()
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)                                 # (the GEMM tuner and the libraries may print: stdout carries the JSON line only)
            try:
                res["aux_rooflines"] = aux_rooflines(dev)
            except Exception as exc:                      # (never at the cost of the bench line)
                res["aux_rooflines"] = [{"kernel": "aux legs", "error": repr(exc)}]
            finally:
                sys.stdout.flush()
                _flush_c_stdout()                        # (C-level prints -- RCCL's banner -- sit in libc's buffer when stdout is a file)
                os.dup2(saved, 1)
                os.close(saved)
        if args.project_ranks > 1 and world == 1 and primary:
            try:
                del td, pipes, seq_plan
            except NameError:
                pass
            torch.cuda.empty_cache()
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)                                 # (RCCL prints its banner to the C stdout: stdout carries the JSON line only)
            try:
                res["projected_strong_scaling"] = project_strong_scaling(dev, args.bins, args.elements, args.cohorts, args.seed,
                                                                         max_ranks=args.project_ranks)
            except Exception as exc:                      # (never at the cost of the bench line)
                res["projected_strong_scaling"] = {"error": repr(exc)}
            finally:
                sys.stdout.flush()
                _flush_c_stdout()                        # (C-level prints -- RCCL's banner -- sit in libc's buffer when stdout is a file)
                os.dup2(saved, 1)
                os.close(saved)
        if args.e2e and world == 1 and primary and (args.bins, args.elements, args.cohorts) == (288_000, 120_091, 37):
            # the number a user of the drop-in sees: files in, results.txt out; never mixed into `value`
            import tempfile
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import e2e_bench
            torch.cuda.empty_cache()
            work = args.e2e_dir or tempfile.mkdtemp(prefix="dig_e2e_")
            saved = os.dup(1)
            os.dup2(2, 1)                                 # the pipeline's progress lines go to stderr: stdout carries the JSON line only
            try:
                res["e2e"] = e2e_bench.run_e2e(workdir=work)
            except Exception as exc:                      # (a full disk must not cost the bench line)
                res["e2e"] = {"error": repr(exc)}
            finally:
                sys.stdout.flush()
                _flush_c_stdout()                        # (C-level prints -- RCCL's banner -- sit in libc's buffer when stdout is a file)
                os.dup2(saved, 1)
                os.close(saved)
        if args.cpu_sample > 0 and world == 1:
            res["cpu_baseline"], res["cpu_baseline_all_cores"] = cpu_res
        else:
            res["cpu_baseline"] = None
    torch.cuda.empty_cache()
    return res if rank == 0 else None


if __name__ == "__main__":
    main()
