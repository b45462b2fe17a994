#!/usr/bin/env python
"""A/B builds of the tile-probability kernel (developer tool).

    python tools/tile_variant_bench.py lib.so:DIG_TILES_FORM=classic lib.so other.so ...

Each spec runs in its own process (DIG_HIP_LIB + environment).  Workload: 36 000 10-kb bins of a random packed genome
with runs of N, 200 tiles of 50 positions, 37 cohorts.  The first spec's output is the reference of the others."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(ref_path, write_ref):
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    from digdriver_amd import engine
    from digdriver_amd.data_tools.genome import PackedGenome
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(11)
    nwin, window, C = int(os.environ.get("TB_BINS", 36000)), 10000, int(os.environ.get("TB_C", 37))
    nbases = nwin * window
    words = (torch.randint(0, 2 ** 31 - 1, (nbases // 8 + 2,), dtype=torch.int32, device=dev, generator=g) & 0x33333333)
    words[0] = 0x44444444
    words[-1] = 0x44444444
    words[5000:5400] = 0x44444444                            # a run of N across several tiles
    words[min(20001, words.numel() - 2)] = 0x33334333           # single N bases
    words[1250 * 7:1250 * 8] = 0x44444444                    # a whole bin of N (T = 0)
    genome = PackedGenome(["chr1"], [0], [nbases], np.zeros(2, np.uint32))
    genome._dev[(dev.type, dev.index)] = (words, torch.zeros(1, dtype=torch.int64, device=dev),
                                          torch.full((1,), nbases, dtype=torch.int64, device=dev))
    starts = np.arange(nwin, dtype=np.int64) * window
    chroms = ["chr1"] * nwin
    S = torch.rand((C, 64), device=dev, generator=g, dtype=torch.float64) * 1e-2
    n_tiles = int(os.environ.get("TB_TILES", 200))
    binsize = int(os.environ.get("TB_BINSIZE", 50))

    from digdriver_amd import _lib
    ci = genome.chrom_index(chroms)
    wd, off, ln = genome.on_device(dev)
    rc, rs, re_ = torch.as_tensor(ci, device=dev), torch.as_tensor(starts, device=dev), torch.as_tensor(starts + window, device=dev)
    pt = torch.empty((C, nwin, n_tiles), dtype=torch.float64, device=dev)
    first = torch.empty(nwin, dtype=torch.int64, device=dev)
    nval = torch.empty(nwin, dtype=torch.int32, device=dev)

    def run():                                               # the bare C-ABI call: no host preparation inside the timing
        _lib.call("dig_base_tile_probs", _lib.dev_ptr(wd), wd.numel(), _lib.dev_ptr(off), _lib.dev_ptr(ln), 1, _lib.dev_ptr(rc),
                  _lib.dev_ptr(rs), _lib.dev_ptr(re_), nwin, _lib.dev_ptr(S), C, binsize, n_tiles, _lib.dev_ptr(pt),
                  _lib.dev_ptr(first), _lib.dev_ptr(nval), _lib.stream_ptr())
        return pt, first, nval
    chk = engine.base_tile_probs(genome, chroms[:64], starts[:64], starts[:64] + window, S, binsize, n_tiles=n_tiles, device=dev)
    pt, first, nval = run()
    torch.cuda.synchronize()
    times = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        pt, first, nval = run()
        b.record()
        torch.cuda.synchronize()
        times.append(a.elapsed_time(b))
    assert torch.equal(torch.nan_to_num(chk[0]), torch.nan_to_num(pt[:, :64])), "engine wrapper and bare call disagree"
    out = {"ms": min(times), "ms_med": sorted(times)[2], "whole_genome_ms": min(times) * 288000 / nwin}
    try:                                                     # -DDIG_TM_TIMING builds: cycles per phase, wave 0 of every workgroup
        import ctypes
        fn = _lib.load().dig_debug_tile_profile
        buf = (ctypes.c_ulonglong * 8)()
        fn(buf)                                              # (clears the counters of the launches so far)
        run()
        fn(buf)
        tot = float(sum(buf))
        out["phase_cycles_per_region_block"] = [round(v / nwin, 1) for v in buf]
        out["phase_share"] = [round(v / tot, 3) for v in buf]
        out["phases"] = ("one-role kernel: stage+zero | barrier | hist | barrier | Hsum(+tail) | barrier | T+product+stores | barrier; "
                         "two-role kernel: walker loads | zero | walk | sums | words+description | barrier, multiplier product | barrier")
    except AttributeError:
        pass
    sel = np.unique(np.clip(np.r_[0:40, 495:505, 1995:2005, nwin - 20:nwin], 0, nwin - 1))
    got = pt[:, torch.as_tensor(sel, device=dev)].cpu().numpy()
    summ = torch.nan_to_num(pt, nan=0.0).sum(dim=2).cpu().numpy()          # per (cohort, bin): 1 when the bin has any valid base
    if write_ref:
        np.savez(ref_path, got=got, summ=summ, first=first.cpu().numpy(), nval=nval.cpu().numpy())
    else:
        ref = np.load(ref_path)
        m = np.isfinite(ref["got"]) & (ref["got"] != 0)
        out["max_rel"] = float((np.abs(got[m] - ref["got"][m]) / np.abs(ref["got"][m])).max())
        out["nan_mismatch"] = int((np.isnan(got) != np.isnan(ref["got"])).sum())
        out["zero_mismatch"] = int(((got == 0) != (ref["got"] == 0)).sum())
        out["max_abs_sum_diff"] = float(np.abs(summ - ref["summ"]).max())
        out["first_nval_equal"] = bool(np.array_equal(first.cpu().numpy(), ref["first"]) and np.array_equal(nval.cpu().numpy(), ref["nval"]))
    print("RESULT " + json.dumps(out), flush=True)


def main():
    if sys.argv[1] == "--child":
        return child(sys.argv[2], sys.argv[3] == "1")
    ref = "/tmp/tile_variant_ref.npz"
    for i, spec in enumerate(sys.argv[1:]):
        lib, _, envs = spec.partition(":")
        path = lib if os.path.isabs(lib) else os.path.join(ROOT, "tools", "variants", lib)
        env = dict(os.environ, DIG_HIP_LIB=path)
        env.update(dict(kv.split("=") for kv in envs.split(",") if kv))
        p = subprocess.run([sys.executable, __file__, "--child", ref, "1" if i == 0 else "0"], env=env, capture_output=True, text=True)
        line = [l for l in p.stdout.split("\n") if l.startswith("RESULT ")]
        print("%-44s %s" % (spec, line[0][7:] if line else "FAILED\n" + p.stdout[-1500:] + p.stderr[-3000:]), flush=True)


if __name__ == "__main__":
    main()
