#!/usr/bin/env python
"""Time dig_count_contexts (4-bit) and dig_count_contexts2 (2-bit) on 288 000 10-kb windows (developer tool)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from digdriver_amd import _lib
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(2)
nwin, window = 288_000, int(os.environ.get("WINDOW", 10_000))
nbases = nwin * window
p = _lib.dev_ptr
off = torch.zeros(1, dtype=torch.int64, device=dev); ln = torch.full((1,), nbases, dtype=torch.int64, device=dev)
rc = torch.zeros(nwin, dtype=torch.int32, device=dev); rs = torch.arange(nwin, dtype=torch.int64, device=dev) * window
re_ = rs + window; rm = torch.zeros(nwin, dtype=torch.uint8, device=dev)
res = torch.empty((nwin, 64), dtype=torch.int32, device=dev)
def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
words = torch.randint(0, 2 ** 31 - 1, (nbases // 8 + 2,), dtype=torch.int32, device=dev, generator=g) & 0x33333333
words[0] = 0x44444444; words[-1] = 0x44444444
t4 = timeit(lambda: _lib.call("dig_count_contexts", p(words), words.numel(), p(off), p(ln), 1, p(rc), p(rs), p(re_), p(rm), nwin, p(res), _lib.stream_ptr()))
r4 = res.clone()
# the same sequence at 2 bits
w = words[1:-1].to(torch.int64) & 0xffffffff
codes = torch.stack([(w >> (4 * k)) & 3 for k in range(8)], 1).reshape(-1, 16)
w2 = torch.zeros(codes.shape[0], dtype=torch.int64, device=dev)
for k in range(16): w2 |= codes[:, k] << (2 * k)
words2 = torch.zeros(4 + w2.numel() + 24, dtype=torch.int32, device=dev)
words2[4:4 + w2.numel()] = torch.where(w2 >= 2 ** 31, w2 - 2 ** 32, w2).to(torch.int32)
del w, codes, w2
t2 = timeit(lambda: _lib.call("dig_count_contexts2", p(words2), words2.numel(), None, None, 0, None, 0, p(off), p(ln), 1, p(rc), p(rs), p(re_), p(rm), nwin, p(res), _lib.stream_ptr()))
same = bool(torch.equal(r4, res))
print(json.dumps({"ms_4bit": t4, "ms_2bit": t2, "same_counts": same, "hbm_frac_4bit": (nbases * 0.5 + nwin * 256) / (t4 * 1e-3) / 8e12,
                  "hbm_frac_2bit": (nbases * 0.25 + nwin * 256) / (t2 * 1e-3) / 8e12, "window": window}))
