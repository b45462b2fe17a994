#!/usr/bin/env python3
"""Profiling target: a few bare dig_base_tile_probs launches (36 000 bins x 200 tiles x 37 cohorts).  Developer tool."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from digdriver_amd import _lib                                       # noqa: E402
from digdriver_amd.data_tools.genome import PackedGenome             # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(11)
nwin, window, C, n_tiles, binsize = 36000, 10000, 37, 200, 50
nbases = nwin * window
words = (torch.randint(0, 2 ** 31 - 1, (nbases // 8 + 2,), dtype=torch.int32, device=dev, generator=g) & 0x33333333)
words[0] = 0x44444444
words[-1] = 0x44444444
genome = PackedGenome(["chr1"], [0], [nbases], np.zeros(2, np.uint32))
genome._dev[(dev.type, dev.index)] = (words, torch.zeros(1, dtype=torch.int64, device=dev),
                                      torch.full((1,), nbases, dtype=torch.int64, device=dev))
starts = np.arange(nwin, dtype=np.int64) * window
S = torch.rand((C, 64), device=dev, generator=g, dtype=torch.float64) * 1e-2
wd, off, ln = genome.on_device(dev)
rc = torch.zeros(nwin, dtype=torch.int32, device=dev)
rs, re_ = torch.as_tensor(starts, device=dev), torch.as_tensor(starts + window, device=dev)
pt = torch.empty((C, nwin, n_tiles), dtype=torch.float64, device=dev)
first = torch.empty(nwin, dtype=torch.int64, device=dev)
nval = torch.empty(nwin, dtype=torch.int32, device=dev)
for _ in range(int(os.environ.get("TP_REPS", 6))):
    _lib.call("dig_base_tile_probs", _lib.dev_ptr(wd), wd.numel(), _lib.dev_ptr(off), _lib.dev_ptr(ln), 1, _lib.dev_ptr(rc),
              _lib.dev_ptr(rs), _lib.dev_ptr(re_), nwin, _lib.dev_ptr(S), C, binsize, n_tiles, _lib.dev_ptr(pt),
              _lib.dev_ptr(first), _lib.dev_ptr(nval), _lib.stream_ptr())
torch.cuda.synchronize()
print("done", float(torch.nan_to_num(pt[0, 0]).sum()))
