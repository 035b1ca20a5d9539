"""Statistics of the dropout hash of csrc/common.h (dropout_quad_words: three multiply-folds per quad of elements, 16 random bits
per element) next to the rounds-1/2 hash (two rounds of xorshift-multiply per pair), in numpy on the CPU: drop rate per field,
chi-square of each 16-bit field over 2^22 counters (expected 65 535 +- 362), the largest |z| of the correlation between the
drop events of two fields of the same quad, of any two fields at counter lags 1, 2, 3, 7, one attention row (99 quads), one head
(99 x 393), 2^16 and 2^20, and between two consecutive raw seeds.  |z| values of a few units are what independent bits give.

    python tools/hash_quality.py"""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def fold(x, c):
    p = x.astype(np.uint64) * np.uint64(c)
    return (p & M32) ^ (p >> np.uint64(32))


def quad_new(seed, quad):
    seed_lo, seed_hi = np.uint64(seed & 0xFFFFFFFF), np.uint64(seed >> 32)
    lo, hi = quad & M32, quad >> np.uint64(32)
    x = (lo + seed_lo + (((hi << np.uint64(16)) | (hi >> np.uint64(16))) & M32)) & M32
    y = (fold(x, 0x9E3779B1) ^ seed_hi) & M32
    return fold(y, 0x85EBCA77), fold(y, 0xC2B2AE3D)


def quad_old(seed, quad):
    seed_lo, seed_hi = np.uint64(seed & 0xFFFFFFFF), np.uint64(seed >> 32)

    def h32(v):
        v = v ^ (v >> np.uint64(16)); v = (v * np.uint64(0x7feb352d)) & M32
        v = v ^ (v >> np.uint64(15)); v = (v * np.uint64(0x846ca68b)) & M32
        return v ^ (v >> np.uint64(16))
    lo = quad & M32
    return h32((2 * lo + seed_lo) & M32) ^ seed_hi, h32((2 * lo + np.uint64(1) + seed_lo) & M32) ^ seed_hi


def fields(w0, w1):
    f = np.uint64(0xFFFF)
    return [w0 & f, (w0 >> np.uint64(16)) & f, w1 & f, (w1 >> np.uint64(16)) & f]


def report(name, fn, seed, n=1 << 22, p=0.1):
    quad = np.arange(n, dtype=np.uint64)
    f = fields(*fn(seed, quad))
    keep = [x >= np.uint64(int(p * 65536)) for x in f]
    d = [(~k).astype(np.float64) - p for k in keep]
    sd = p * (1 - p)
    chi = [int((((np.bincount(x.astype(np.int64), minlength=65536) - n / 65536.0) ** 2) / (n / 65536.0)).sum()) for x in f]
    within = max(abs((d[i] * d[j]).sum() / sd / np.sqrt(n)) for i in range(4) for j in range(i + 1, 4))
    lags = {lag: max(abs((d[i][:-lag] * d[j][lag:]).sum() / sd / np.sqrt(n - lag)) for i in range(4) for j in range(4))
            for lag in (1, 2, 3, 7, 99, 99 * 393, 1 << 16, 1 << 20)}
    print("%-4s seed %x: drop rates %s  chi2 %s  within-quad max|z| %.2f  lagged max|z| %s"
          % (name, seed, [round(1 - k.mean(), 5) for k in keep], chi, within, {k: round(v, 1) for k, v in lags.items()}))
    return d


for name, fn in (("new", quad_new), ("old", quad_old)):
    a = report(name, fn, 0x1234567890ABCDEF)
    b = report(name, fn, 0x1234567890ABCDF0)
    print("     consecutive seeds, per field z:", [round((a[i] * b[i]).sum() / 0.09 / np.sqrt(len(a[i])), 2) for i in range(4)])
