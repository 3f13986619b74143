#!/usr/bin/env python3
"""GEMM shapes of the wav2vec2-large train step at 8 x 10 s per GPU (M = 4096) across the tile kernels."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aptai_amd import ops
from tools.gemm_bench import bench
M = 4096
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(*s, device="cuda", generator=g).to(torch.bfloat16)
shapes = {"qkv": (M, 3072, 1024), "out": (M, 1024, 1024), "ffn1": (M, 4096, 1024), "ffn2": (M, 1024, 4096)}
for name, (m, n, k) in shapes.items():
    a, b = rnd(m, k), rnd(n, k)
    row = []
    for tile in (64, 128, 192, 256):
        t = bench(lambda: ops.gemm(a, b, m, n, k, tile=tile), iters=50)
        row.append(f"t{tile} {t:6.1f}us")
    print(f"NT {name:5s} {m}x{n}x{k}: " + " | ".join(row), flush=True)
for name, (m, n, k) in {"d-ffn2": (M, 4096, 1024), "d-ffn1": (M, 1024, 4096), "d-out": (M, 1024, 1024), "d-qkv": (M, 1024, 3072)}.items():
    a, b = rnd(m, k), rnd(k, n)
    row = []
    for tile in (64, 128, 192, 256):
        t = bench(lambda: ops.gemm(a, b, m, n, k, b_kmajor=True, tile=tile), iters=50)
        row.append(f"t{tile} {t:6.1f}us")
    print(f"NN {name:6s} {m}x{n}x{k}: " + " | ".join(row), flush=True)
