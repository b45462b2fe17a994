#!/usr/bin/env python
"""Time the tile-probability kernels: trinucleotide (dig_base_tile_probs) and penta-nucleotide (dig_base_tile_probs_ctx, n_up = 2)
on 36 000 bins x 200 tiles x 37 cohorts (developer tool)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from digdriver_amd import _lib
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(2)
chunk, window, C = int(os.environ.get("BINS", 36000)), 10_000, int(os.environ.get("COHORTS", 37))
nbases = chunk * window
words = torch.randint(0, 2 ** 31 - 1, (nbases // 8 + 2,), dtype=torch.int32, device=dev, generator=g) & 0x33333333
words[0] = 0x44444444; words[-1] = 0x44444444
off = torch.zeros(1, dtype=torch.int64, device=dev); ln = torch.full((1,), nbases, dtype=torch.int64, device=dev)
rc = torch.zeros(chunk, dtype=torch.int32, device=dev); rs = torch.arange(chunk, dtype=torch.int64, device=dev) * window
re_ = rs + window
p = _lib.dev_ptr
pt = torch.empty((C, chunk, 200), dtype=torch.float64, device=dev)
first = torch.empty(chunk, dtype=torch.int64, device=dev); nval = torch.empty(chunk, dtype=torch.int32, device=dev)
def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
S3 = torch.rand((C, 64), device=dev, generator=g, dtype=torch.float64) * 1e-2
S5 = torch.rand((C, 1024), device=dev, generator=g, dtype=torch.float64) * 1e-2
t3 = timeit(lambda: _lib.call("dig_base_tile_probs", p(words), words.numel(), p(off), p(ln), 1, p(rc), p(rs), p(re_), chunk, p(S3), C, 50, 200, p(pt), p(first), p(nval), _lib.stream_ptr()))
t5 = timeit(lambda: _lib.call("dig_base_tile_probs_ctx", p(words), words.numel(), p(off), p(ln), 1, p(rc), p(rs), p(re_), chunk, p(S5), C, 2, 50, 200, p(pt), p(first), p(nval), _lib.stream_ptr()))
print(json.dumps({"ms_trinucleotide": t3, "ms_penta": t5, "ratio": t5 / t3, "bins": chunk, "cohorts": C,
                  "penta_table_reads_per_s": chunk * window * C / (t5 * 1e-3)}))
try:
    import ctypes
    fn = _lib.load().dig_debug_tile_profile
    buf = (ctypes.c_ulonglong * 8)()
    fn(buf)
    _lib.call("dig_base_tile_probs_ctx", p(words), words.numel(), p(off), p(ln), 1, p(rc), p(rs), p(re_), chunk, p(S5), C, 2, 50, 200, p(pt), p(first), p(nval), _lib.stream_ptr())
    fn(buf)
    tot = float(sum(buf)) or 1.0
    print("phase share (output+loop | barrier | words->LDS+request | codes+barrier | walk | barrier | reduce+barrier | -):", [round(v / tot, 3) for v in buf],
          "cycles per region-pass:", round(tot / (chunk * ((C + 7) // 8)), 1))
except AttributeError:
    pass
try:                                    # the row walk (dig_tiles_rows.hip), -DDIG_TM_TIMING build
    import ctypes
    fn = _lib.load().dig_debug_rows_profile
    buf = (ctypes.c_ulonglong * 8)()
    fn(buf)
    _lib.call("dig_base_tile_probs_ctx", p(words), words.numel(), p(off), p(ln), 1, p(rc), p(rs), p(re_), chunk, p(S5), C, 2, 50, 200, p(pt), p(first), p(nval), _lib.stream_ptr())
    fn(buf)
    for name, part in (("first wave", buf[0:4]), ("last wave", buf[4:8])):
        tot = float(sum(part))
        print("row walk,", name, "phase share (walk | barrier | bases -> LDS + output | barrier):", [round(v / tot, 3) for v in part],
              "cycles per region (all passes):", round(tot / chunk, 1))
except AttributeError:
    pass
