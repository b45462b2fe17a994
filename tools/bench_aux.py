#!/usr/bin/env python
"""Roofline measurements for the two remaining HBM-bound rows of the scope table that bench.py's step does not
contain: the per-bin track gather (a1, dig_gather_bins) and the per-base tiled NB test (a18, dig_tiled_nb_test).
Synthetic inputs per SURVEY 8d; prints one JSON object (committed as profiles/rNN_aux.json).  Developer tool."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from digdriver_amd import engine                     # noqa: E402

HBM_PEAK = 8.0e12


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


def main():
    dev = torch.device("cuda:0")
    out = {"gather_bins": [], "tiled_nb_test": []}
    g = torch.Generator(device=dev).manual_seed(2)
    # ---- a1: gather of B bins x 100 positions x 735 tracks from an HBM-resident matrix -------------------------
    N, L, T, B = 20000, 100, 735, 4096
    x16 = torch.randint(0, 10000, (N, L, T), dtype=torch.int16, device=dev, generator=g)   # round(x, 2) * 100 values
    rows = torch.randint(0, N, (B,), dtype=torch.int64, device=dev, generator=g)
    sel = torch.arange(T, dtype=torch.int32, device=dev)
    for name, x in (("i16", x16), ("f32", x16[: N // 2].float())):
        for odt in ("bf16", "f32"):
            for tr in (False, True):
                r = rows % x.shape[0]
                dt = timeit(lambda: engine.gather_bins(x, r, None, out_dtype=odt, transpose=tr))     # all tracks
                by = B * L * T * (x.element_size() + (2 if odt == "bf16" else 4))
                out["gather_bins"].append({"in": name, "out": odt, "channels_first": tr, "bins": B, "L": L, "T": T,
                                           "ms": dt * 1e3, "bins_per_s": B / dt, "algorithmic_GBps": by / dt / 1e9,
                                           "frac_hbm_peak": by / dt / HBM_PEAK})
        del x
    # a track-selection file (dataset_generator.py:57-80): 512 of the 735 tracks, row-major output
    sel512 = torch.sort(torch.randperm(T, device=dev, generator=g)[:512]).values.to(torch.int32)
    for odt in ("bf16", "f32"):
        dt = timeit(lambda: engine.gather_bins(x16, rows, sel512, out_dtype=odt, transpose=False))
        by = B * L * (T * 2 + 512 * (2 if odt == "bf16" else 4))
        out["gather_bins"].append({"in": "i16", "out": odt, "channels_first": False, "tracks": 512, "bins": B, "L": L, "T": T, "ms": dt * 1e3,
                                   "bins_per_s": B / dt, "algorithmic_GBps": by / dt / 1e9, "frac_hbm_peak": by / dt / HBM_PEAK})
    del x16
    torch.cuda.empty_cache()
    # ---- a18: per-base tiles, 50-bp tiles of 10-kb bins (200 tiles per bin), C cohorts ------------------------------
    C, nb, nt = 37, 8000, 200
    rng = np.random.default_rng(5)
    mu = torch.as_tensor(rng.gamma(9.0, 3.0, (C, nb)), device=dev)
    sigma = torch.as_tensor(rng.gamma(4.0, 1.0, (C, nb)), device=dev)
    pt_np = rng.dirichlet(np.ones(nt), size=nb)                       # per-tile share of the bin's rate
    pt = torch.as_tensor(pt_np, device=dev)
    k = torch.poisson(mu[:, :, None] * pt[None, :, :]).to(torch.int32)
    dt = timeit(lambda: engine.tiled_nb_test(pt, k, mu, sigma), n=5, warm=1)
    units = C * nb * nt
    by = 28.0 * units + 16.0 * C * nb
    pv, ex = engine.tiled_nb_test(pt, k, mu, sigma)
    out["tiled_nb_test"].append({"cohorts": C, "bins": nb, "tiles_per_bin": nt, "tile_cohort_tests": units, "ms": dt * 1e3,
                                 "tests_per_s": units / dt, "algorithmic_GBps": by / dt / 1e9,
                                 "frac_hbm_peak": by / dt / HBM_PEAK, "finite": bool(torch.isfinite(pv).all().item()),
                                 "whole_genome_x37_seconds": 288000 * nt * C / (units / dt)})
    # ---- f3: trinucleotide contexts of all 10-kb windows of a genome-sized packed sequence --------------------------
    del pt, k, mu, sigma, pv, ex
    torch.cuda.empty_cache()
    from digdriver_amd.data_tools.genome import PackedGenome
    nwin, window = 288000, 10000
    nbases = nwin * window
    words = (torch.randint(0, 2 ** 31 - 1, (nbases // 8 + 2,), dtype=torch.int32, device=dev, generator=g) & 0x33333333)
    words[0] = 0x44444444
    words[-1] = 0x44444444
    genome = PackedGenome(["chr1"], [0], [nbases], np.zeros(2, np.uint32))          # host copy not needed: device-resident
    genome._dev[(dev.type, dev.index)] = (words, torch.zeros(1, dtype=torch.int64, device=dev),
                                          torch.full((1,), nbases, dtype=torch.int64, device=dev))
    starts = np.arange(nwin, dtype=np.int64) * window
    chroms = ["chr1"] * nwin
    ci = genome.chrom_index(chroms[:1])
    from digdriver_amd import _lib
    rc = torch.zeros(nwin, dtype=torch.int32, device=dev)
    rs, re_ = torch.as_tensor(starts, device=dev), torch.as_tensor(starts + window, device=dev)
    rm = torch.zeros(nwin, dtype=torch.uint8, device=dev)
    res = torch.empty((nwin, 64), dtype=torch.int32, device=dev)
    off, ln = genome._dev[(dev.type, dev.index)][1:]

    def run():
        _lib.call("dig_count_contexts", _lib.dev_ptr(words), words.numel(), _lib.dev_ptr(off), _lib.dev_ptr(ln), 1,
                  _lib.dev_ptr(rc), _lib.dev_ptr(rs), _lib.dev_ptr(re_), _lib.dev_ptr(rm), nwin, _lib.dev_ptr(res),
                  _lib.stream_ptr())
    dt = timeit(run, n=5, warm=1)
    by = nbases * 0.5 + nwin * 256.0
    out["count_contexts"] = [{"windows": nwin, "window_bp": window, "bases": nbases, "ms": dt * 1e3,
                              "bases_per_s": nbases / dt, "algorithmic_GBps": by / dt / 1e9,
                              "frac_hbm_peak": by / dt / HBM_PEAK, "total_counted": int(res.sum(dtype=torch.int64).item())}]
    # ---- a18 front half: tile probabilities of every 10-kb bin of the genome for 37 cohorts (BASELINE configs[4]) -----
    S = torch.rand((37, 64), device=dev, generator=g, dtype=torch.float64) * 1e-2
    chunk = 36_000                                          # bins per call (the outputs of the whole genome are 17 GB)
    # the bare C-ABI calls with device-resident arguments (engine.base_tile_probs prepares them on the host per call:
    # chromosome names -> indices, three small uploads -- that is the caller's cost, not the kernel's)
    wd, off_t, ln_t = genome.on_device(dev)
    ci_t = torch.as_tensor(np.repeat(np.asarray(genome.chrom_index(chroms[:1]), np.int32), nwin), device=dev)
    rs_t, re_t = torch.as_tensor(starts, device=dev), torch.as_tensor(starts + window, device=dev)
    pt = torch.empty((37, chunk, 200), dtype=torch.float64, device=dev)
    first = torch.empty(chunk, dtype=torch.int64, device=dev)
    nval = torch.empty(chunk, dtype=torch.int32, device=dev)

    def run_tiles():
        for s0 in range(0, nwin, chunk):
            nb_ = min(chunk, nwin - s0)
            _lib.call("dig_base_tile_probs", _lib.dev_ptr(wd), wd.numel(), _lib.dev_ptr(off_t), _lib.dev_ptr(ln_t), 1,
                      _lib.dev_ptr(ci_t[s0:]), _lib.dev_ptr(rs_t[s0:]), _lib.dev_ptr(re_t[s0:]), nb_, _lib.dev_ptr(S), 37, 50, 200,
                      _lib.dev_ptr(pt), _lib.dev_ptr(first), _lib.dev_ptr(nval), _lib.stream_ptr())
    dt = timeit(run_tiles, n=2, warm=1)
    tiles = nwin * 200
    out["base_tile_probs"] = [{"bins": nwin, "tiles_per_bin": 200, "cohorts": 37, "tile_cohort_values": tiles * 37, "ms": dt * 1e3,
                               "tile_cohort_values_per_s": tiles * 37 / dt, "flops": 2.0 * 64 * 37 * tiles,
                               "fp64_TFLOPs": 2.0 * 64 * 37 * tiles / dt / 1e12, "frac_fp64_vector_peak": 2.0 * 64 * 37 * tiles / dt / 78.6e12,
                               "algorithmic_bytes": nbases * 0.5 + tiles * 37 * 8.0,
                               "algorithmic_GBps": (nbases * 0.5 + tiles * 37 * 8.0) / dt / 1e9,
                               "frac_hbm_peak": (nbases * 0.5 + tiles * 37 * 8.0) / dt / HBM_PEAK}]
    del pt, first, nval
    torch.cuda.empty_cache()
    # ---- f1: mutation x element-block interval join (dig_overlap_join_count / fill) ----------------------------------
    del words, res
    torch.cuda.empty_cache()
    from digdriver_amd.data_tools import tabulate_gpu
    rng = np.random.default_rng(7)
    n_blk, n_mut = 360_000, 20_000_000                      # 120 k elements x 3 blocks; 37 cohorts x ~5e5 mutations
    per_chrom = 130_000_000
    b_chrom = rng.integers(1, 23, n_blk)
    b_start = rng.integers(0, per_chrom, n_blk)
    b_end = b_start + rng.integers(200, 3000, n_blk)
    blocks = tabulate_gpu.ElementBlocks(b_chrom, b_start, b_end, np.arange(n_blk) // 3, n_blk // 3, dev)
    m_chrom = torch.randint(1, 23, (n_mut,), dtype=torch.int64, device=dev, generator=g)
    m_start = torch.randint(0, per_chrom, (n_mut,), dtype=torch.int64, device=dev, generator=g)
    m_end = m_start + 1
    pairs = [0]

    def run_join():
        pm, pb = tabulate_gpu.overlap_pairs(blocks, m_chrom, m_start, m_end)
        pairs[0] = pm.numel()
    dt = timeit(run_join, n=3, warm=1)
    by = n_mut * 24.0 + pairs[0] * 8.0
    out["overlap_join"] = [{"mutations": n_mut, "blocks": n_blk, "pairs": pairs[0], "ms": dt * 1e3,
                            "mutations_per_s": n_mut / dt, "algorithmic_GBps": by / dt / 1e9,
                            "note": "count + prefix sum + fill; two binary searches over the block keys per mutation "
                                    "(latency-bound, not HBM-bound)"}]
    # the whole observed-count tabulation (join + de-duplication + per-(element, sample) counts + [E, C] planes)
    n_coh, per = 37, n_mut // 37
    cohorts = []
    for c in range(n_coh):
        sl = slice(c * per, (c + 1) * per)
        cohorts.append(dict(chrom=m_chrom[sl], start=m_start[sl], end=m_end[sl],
                            uid=torch.arange(per, dtype=torch.int64, device=dev),
                            sample=torch.randint(0, 500, (per,), dtype=torch.int64, device=dev, generator=g),
                            indel=(torch.rand(per, device=dev, generator=g) < 0.08).long(),
                            cohort=torch.full((per,), c, dtype=torch.int64, device=dev),
                            sample_names=["S%d" % j for j in range(500)]))
    tot = [0]

    def run_tab():
        o1, o2, o3, _ = tabulate_gpu.tabulate_cohorts(blocks, cohorts)
        tot[0] = int(o1.sum().item() + o3.sum().item())
    dt = timeit(run_tab, n=3, warm=1)
    out["tabulate_cohorts"] = [{"cohorts": n_coh, "mutations": n_coh * per, "elements": blocks.n_elements,
                                "counted": tot[0], "ms": dt * 1e3, "mutations_per_s": n_coh * per / dt}]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
