"""Two-level attention-dropout hash: r = rng_hash(row) (strong, once per query row), x = level2(r, pair) (cheap, per key pair).
Prints keep-rate, correlations (adjacent pairs, adjacent rows, lo/hi halves, seed+1) and a chi-square of the top byte of both
halves, for the shapes the kernels use (rows x 256 pairs)."""
import numpy as np
M = np.uint64(0xffffffff)
U = np.uint64


def mul24(a, b):
    return ((a & U(0xffffff)) * U(b & 0xffffff)) & M


def rot(x, r):
    return ((x >> U(r)) | (x << U(32 - r))) & M


def rng_hash(idx, s0, s1):       # common.h:rng_hash
    x = (idx + U(s0)) & M
    x ^= x >> U(15)
    x = (mul24(x, 0x9E3779) + U(s1)) & M
    x ^= x >> U(13)
    x = (mul24(x, 0xC2B2AF) + rot(x, 7)) & M
    x ^= x >> U(11)
    x = (mul24(x, 0x85EBCB) + (x >> U(5))) & M
    return x ^ (x >> U(16))


def l2a(r, j):
    x = (r + mul24(j, 0x9E3779)) & M
    x ^= x >> U(15)
    x = mul24(x, 0xC2B2AF)
    return x ^ (x >> U(13))


def l2b(r, j):                    # + rotate-add of the pre-multiply word (keeps the top byte the multiply drops)
    x = (r + mul24(j, 0x9E3779)) & M
    x ^= x >> U(15)
    x = (mul24(x, 0xC2B2AF) + rot(x, 7)) & M
    return x ^ (x >> U(14))


def l2c(r, j):
    x = r ^ mul24(j + U(1), 0x9E3779)
    x = (mul24(x, 0xC2B2AF) + (x >> U(9))) & M
    return x ^ (x >> U(15))


def stats(fn, name, rows=4096, pairs=256, thr=6554):
    for s0, s1 in [(0x12345678, 0x9abcdef0), (1, 2), (0xdeadbeef, 0)]:
        q = np.arange(rows, dtype=np.uint64)[:, None] + U(3 * 512 * 5)
        j = np.arange(pairs, dtype=np.uint64)[None, :]
        r = rng_hash(q, s0, s1)
        v = fn(r, j)
        lo = (v & U(0xffff)).astype(np.int64); hi = (v >> U(16)).astype(np.int64)
        klo = (lo >= thr).astype(np.float64); khi = (hi >= thr).astype(np.float64)
        c = lambda a, b: np.corrcoef(a.ravel(), b.ravel())[0, 1]
        v2 = fn(rng_hash(q, s0 + 1, s1), j); k2 = ((v2 & U(0xffff)).astype(np.int64) >= thr).astype(np.float64)
        n = rows * pairs
        chi = lambda h: (((np.bincount(h.ravel() >> 8, minlength=256) - n / 256) ** 2) / (n / 256)).sum()
        # per-row keep counts: variance vs binomial
        kr = np.concatenate([klo, khi], axis=1).sum(1); p = 1 - thr / 65536
        print(f"{name} keep {klo.mean():.5f} {khi.mean():.5f} c(lo,hi) {c(klo, khi):+.5f} adjpair {c(klo[:, 1:], klo[:, :-1]):+.5f} "
              f"adjrow {c(klo[1:], klo[:-1]):+.5f} pair+2 {c(khi[:, 2:], khi[:, :-2]):+.5f} seed+1 {c(klo, k2):+.5f} chi {chi(lo):.0f} {chi(hi):.0f} "
              f"rowvar/binom {kr.var() / (2 * pairs * p * (1 - p)):.3f}")


if __name__ == "__main__":
    for fn, name in ((l2a, "l2a"), (l2b, "l2b"), (l2c, "l2c")):
        stats(fn, name)


def l2d(r, j):
    x = (r + mul24(j, 0x9E3779)) & M
    x = mul24(x, 0xC2B2AF)
    return x ^ (x >> U(13))


def l2e(r, j):                    # fold, multiply, and take the halves from the product's top 32 of 40 bits
    x = (r + mul24(j, 0x9E3779)) & M
    x ^= x >> U(15)
    p = (x & U(0xffffff)) * U(0xC2B2AF)
    return (p >> U(12)) & M


if __name__ == "__main__":
    print()
    for fn, name in ((l2d, "l2d"), (l2e, "l2e")):
        stats(fn, name)


def l2f(r, j):                    # 16-bit folds (one SDWA xor each)
    x = (r + mul24(j, 0x9E3779)) & M
    x ^= x >> U(16)
    x = mul24(x, 0xC2B2AF)
    return x ^ (x >> U(16))


if __name__ == "__main__":
    print()
    stats(l2f, "l2f")
    stats(l2a, "l2a-big", rows=16384)
