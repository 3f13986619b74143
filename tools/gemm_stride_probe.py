#!/usr/bin/env python3
"""Development: does the row stride of the operands (power-of-two-ish multiples of the L2 channel interleave) limit operand delivery?
NT GEMMs of the step with K-contiguous operands at their natural leading dimension and at padded ones (run on the GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from aptai_amd import ops
from tools.gemm_round import bench

M = 8192
g = torch.Generator(device="cuda").manual_seed(0)
for (n, k) in ((768, 3072), (768, 768), (3072, 768), (2304, 768), (768, 2304)):
    for tile in (192, 128, 256):
        if tile == 256 and n % 256:
            continue
        line = f"NT {M}x{n}x{k} tile {tile}:"
        for pad in (0, 8, 32, 64, 136):
            lda = k + pad
            a = torch.randn(M, lda, device="cuda", generator=g).to(torch.bfloat16)
            b = torch.randn(n, lda, device="cuda", generator=g).to(torch.bfloat16)
            t = bench(lambda: ops.gemm(a, b, M, n, k, lda=lda, ldb=lda, tile=tile), iters=30)
            line += f"  pad {pad:3d}: {t:6.1f} us"
        print(line, flush=True)
