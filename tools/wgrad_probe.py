#!/usr/bin/env python3
"""Round 4: is a 256 x 256 tile + split-K 2 the faster form for one layer's weight gradients (K = 8192)?  One TN problem with the layer's tile
count (108 tiles of 256 x 256 = [6912] x [1024], both operands K-major, fp32 out) through the existing kernels: tile 256 with split-K 2 (216
blocks, one per CU) against the 128-row kernel (432 tiles, the grouped launch's form).  Times include the split-K reduce pass."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from aptai_amd import ops


def bench(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g).to(torch.bfloat16)
    K = 8192
    for M, N in ((6912, 1024), (3072, 768), (2304, 768)):
        a, b = rnd(K, M), rnd(K, N)
        out = torch.empty(M, N, device="cuda")
        fl = 2.0 * M * N * K
        for tile, sk in ((128, 1), (256, 1), (256, 2), (256, 4), (192, 1)):
            try:
                t = bench(lambda: ops.gemm(a, b, M, N, K, a_kmajor=True, b_kmajor=True, out_f32=True, out=out, tile=tile, split_k=sk))
                print(f"TN {M:5d} x {N:5d} x {K}: tile {tile} split-K {sk}: {t:7.1f} us  {fl / t / 1e6:6.0f} TF/s")
            except Exception as e:
                print(f"TN {M} x {N}: tile {tile} split-K {sk}: {e!r}"[:150])


if __name__ == "__main__":
    main()
