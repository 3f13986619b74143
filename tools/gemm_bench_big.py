#!/usr/bin/env python3
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aptai_amd import ops
from tools.gemm_bench import bench
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(*s, device="cuda", generator=g).to(torch.bfloat16)
for (m, n, k) in [(4096, 4096, 4096), (8192, 8192, 8192), (8192, 3072, 768), (8192, 3072, 3072), (8192, 4096, 768), (16384, 4096, 1024)]:
    a, b = rnd(m, k), rnd(n, k)
    for tile in (128, 256):
        t = bench(lambda: ops.gemm(a, b, m, n, k, tile=tile), iters=10)
        print(f"NT {m}x{n}x{k} tile{tile}: {t:8.1f}us {2*m*n*k/t/1e6:7.1f} TF", flush=True)
