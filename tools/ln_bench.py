#!/usr/bin/env python3
"""LayerNorm fwd/bwd timing at the hot-path shape (8192 x 768); APTAI_LN_BWD_BLOCKS=<n> varies the backward grid."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aptai_amd import ops
from tools.gemm_bench import bench
M, H = 8192, 768
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(M, H, device="cuda", generator=g).to(torch.bfloat16)
dy = torch.randn(M, H, device="cuda", generator=g).to(torch.bfloat16)
w = torch.randn(H, device="cuda"); b = torch.randn(H, device="cuda")
y, m, r = ops.layernorm_fwd(x, w, b, 1e-5)
t = bench(lambda: ops.layernorm_fwd(x, w, b, 1e-5))
print(f"ln_fwd {t:.1f}us  ({M*H*4/t/1e6:.2f} TB/s algorithmic)")
for p in (0.0, 0.1):
    t = bench(lambda: ops.layernorm_bwd(dy, x, m, r, w, dropout_p=p, seed=3))
    nb = M * H * (6 if p == 0 else 8)
    print(f"ln_bwd p={p}: {t:.1f}us (both kernels)  ({nb/t/1e6:.2f} TB/s algorithmic)")
