#!/usr/bin/env python3
"""Development: run a few aptai_gemm_bf16 shapes repeatedly so that `rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum` sees their L2 behaviour."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from aptai_amd import ops

M = 8192
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(*s, device="cuda", generator=g).to(torch.bfloat16)
for (n, k, tile) in ((768, 768, 192), (768, 3072, 192), (3072, 768, 128), (2048, 768, 256)):
    a, b = rnd(M, k), rnd(n, k)
    other = rnd(M, 3072)                               # evict between launches, like the step does
    for _ in range(6):
        ops.gemm(a, b, M, n, k, tile=tile)
        other.mul_(1.0)
torch.cuda.synchronize()
