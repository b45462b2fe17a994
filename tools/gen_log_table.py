#!/usr/bin/env python
"""Generate the 128-entry (1/c, log c) table of fast_log_normal (digdriver_amd/csrc/dig_math.hpp) and the
polynomial for log1p(r) on |r| <= 2^-8 * 1.17, with mpmath at 80 digits.  Developer tool; prints C++ to paste.

Reduction: x = 2^k z, z in [0.6875, 1.375) (bit trick with OFF = bits(0.6875)); bin i = top 7 mantissa bits of
bits(x) - OFF: 80 bins of width 2^-8 below 1, 48 bins of width 2^-7 above.  c_i = bin centre, invc = double(1/c_i),
logc = double(-log(invc)) (so the pair is self-consistent: log z = logc + log1p(z * invc - 1) exactly)."""
import struct

import mpmath as mp
import numpy as np

mp.mp.dps = 80
OFF = 0x3fe6000000000000


def as_double(bits):
    return struct.unpack("<d", struct.pack("<Q", bits))[0]


rows, rmax = [], 0.0
for i in range(128):
    lo = as_double(OFF + (i << 45))
    hi = as_double(OFF + ((i + 1) << 45))
    c = (mp.mpf(lo) + mp.mpf(hi)) / 2
    if hi == 1.0:
        # the bin below 1 takes c = 1 instead of its centre: r = z - 1 exactly (|r| <= 2^-8, inside the polynomial's range) and
        # log z = log1p(r) without the cancellation logc + log1p(r) suffers for z -> 1 (absolute error 1e-18, relative
        # 1e-12 at z = 1 - 4e-7: the logarithm of a success probability near 1 is multiplied by alpha ~ 1e6)
        c = mp.mpf(1)
    invc = float(1 / c)
    logc = float(-mp.log(mp.mpf(invc)))
    rows.append((invc, logc))
    rmax = max(rmax, abs(float(mp.mpf(lo) * invc - 1)), abs(float(mp.mpf(hi) * invc - 1)))
print("// max |r| = %.6g" % rmax)
print("static __device__ __constant__ const double kLogTabRom[128][2] = {")
for invc, logc in rows:
    print("    {%s, %s}," % (float(invc).hex(), float(logc).hex()))
print("};")

# minimax-ish (Chebyshev-node least squares in high precision) polynomial P of degree 4 with
#   log1p(r) ~= r + r^2 * P(r),  |r| <= rmax
deg = 4
nodes = [rmax * mp.cos(mp.pi * (2 * j + 1) / (2 * 40)) for j in range(40)]
A = mp.matrix(len(nodes), deg + 1)
b = mp.matrix(len(nodes), 1)
for j, r in enumerate(nodes):
    for d in range(deg + 1):
        A[j, d] = r ** d
    b[j] = (mp.log1p(r) - r) / (r * r)
coef = mp.lu_solve(A.T * A, A.T * b)
coef = [float(c) for c in coef]
print("// log1p(r) = r + r^2 (c0 + c1 r + c2 r^2 + c3 r^3 + c4 r^4):")
print("//", ", ".join("%.17g" % c for c in coef))
worst = 0
for r in np.linspace(-rmax, rmax, 4001):
    if r == 0:
        continue
    P = 0.0
    for c in reversed(coef):
        P = P * r + c
    approx = mp.mpf(r) + mp.mpf(r) ** 2 * mp.mpf(P)
    worst = max(worst, abs(float(approx - mp.log1p(mp.mpf(r)))))
print("// worst absolute truncation error on the interval: %.3g" % worst)
