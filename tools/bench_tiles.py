#!/usr/bin/env python
"""BASELINE configs[4] -- the per-base tiled NB test over every 10-kb bin of the genome x 37 cohorts -- on N GPUs of one node:
parallel.ShardedTiles, one rank per GPU, bins sharded in contiguous ranges (each rank: its slab of the packed genome, its
mu / sigma columns, the mutations that start inside the slab).  No exchange inside the timed step (bins are independent,
nb_model.py:188-234); the Benjamini-Hochberg pass over one cohort's p-values (all-gather over RCCL) is timed separately.

    python tools/bench_tiles.py                                   # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 \
        tools/bench_tiles.py --gpus 8

Rank 0 prints one JSON line (strong scaling: the genome is fixed, every rank takes R / N bins).  Developer tool / secondary
benchmark: the judged line is bench.py's."""
import argparse
import json
import os
import sys
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--bins", type=int, default=288_000)
    ap.add_argument("--cohorts", type=int, default=37)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mutations", type=int, default=4_000_000)
    args = ap.parse_args()
    import torch
    from digdriver_amd import _lib, parallel
    from digdriver_amd.data_tools.genome import PackedGenome
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs torch.distributed.run with %d ranks" % (args.gpus, args.gpus))
    _lib.require_device()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    R, C, W, B = args.bins, args.cohorts, 10_000, 50
    n_chrom = 24
    per = R // n_chrom
    R = per * n_chrom
    rng = np.random.default_rng(4)                                    # the same genome on every rank
    words = (rng.integers(0, 2 ** 32, (per * W * n_chrom) // 8 + 2, dtype=np.uint64).astype(np.uint32) & np.uint32(0x33333333))
    words[0] = words[-1] = 0x44444444
    genome = PackedGenome(["chr%d" % i for i in range(n_chrom)], np.arange(n_chrom, dtype=np.int64) * per * W,
                          np.full(n_chrom, per * W, np.int64), words)
    chroms = np.repeat(["chr%d" % i for i in range(n_chrom)], per)
    starts = np.tile(np.arange(per, dtype=np.int64) * W, n_chrom)
    ends = starts + W
    S = rng.uniform(0, 1e-2, (C, 64))
    mu, sg = rng.uniform(5, 45, (C, R)), rng.uniform(1, 7, (C, R))
    M = args.mutations
    mci = rng.integers(0, n_chrom, M)
    ms = rng.integers(0, per * W, M).astype(np.int64)
    co = rng.integers(0, C, M).astype(np.int32)
    sh = parallel.ShardedTiles(genome, chroms, starts, ends, S, mu, sg, np.array(["chr%d" % i for i in mci]), ms, ms + 1, co, B, dev,
                               rank, world)
    del words, genome

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        sh.run()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = sh.run()
    barrier()
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    q = sh.q_values(0)
    barrier()
    dt_q = time.perf_counter() - t1
    ok = bool(torch.isfinite(res["pval"]).all().item()) and float(res["pt"].sum(dim=2).sub(1).abs().max()) < 1e-12
    if dist is not None:
        tm = torch.tensor([dt, dt_q], dtype=torch.float64, device=dev)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        dt, dt_q = float(tm[0]), float(tm[1])
        dist.destroy_process_group()
    if rank == 0:
        units = float(R) * 200 * C * args.steps
        print(json.dumps({"metric": "per-base tile-cohort NB tests/s (BASELINE configs[4])", "value": units / dt, "unit": "tile-cohort tests/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                          "higher_is_better": True, "scaling": "strong", "dtype": "f64", "data": "synthetic",
                          "config": {"workload": "%d 10-kb bins x 200 tiles of 50 positions x %d cohorts, %d mutations, bins sharded x%d "
                                                 "(genome slab + mu / sigma + mutations per rank)" % (R, C, M, world),
                                     "bins_on_rank0": sh.hi - sh.lo},
                          "step": "dig_base_tile_probs + interval join + dig_tile_mut_counts + dig_tiled_nb_test over the rank's bins",
                          "bh_q_values_one_cohort_ms": dt_q * 1e3, "finite_and_normalised": ok,
                          "q_finite": bool(torch.isfinite(q[sh.valid_pvalues(0)[1]]).all().item())}), flush=True)


if __name__ == "__main__":
    main()
