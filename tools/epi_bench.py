#!/usr/bin/env python3
"""Heavy-epilogue GEMMs of the FFN (fwd: bias+GELU+dropout+pre-activation copy; bwd: dropout x gelu'(aux))."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aptai_amd import ops
from tools.gemm_bench import bench
M, H, I = 8192, 768, 3072
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(*s, device="cuda", generator=g).to(torch.bfloat16)
x, w1, w2 = rnd(M, H), rnd(I, H), rnd(H, I)
b1 = torch.randn(I, device="cuda")
u = torch.empty(M, I, device="cuda", dtype=torch.bfloat16)
dh = rnd(M, H)
t0 = bench(lambda: ops.gemm(x, w1, M, I, H))
t1 = bench(lambda: ops.gemm(x, w1, M, I, H, bias=b1, gelu=True, out_pre=u, dropout_p=0.1, seed=3))
t2 = bench(lambda: ops.gemm(dh, w2, M, I, H, b_kmajor=True))
t3 = bench(lambda: ops.gemm(dh, w2, M, I, H, b_kmajor=True, dgelu_aux=u, dropout_p=0.1, seed=3))
print(f"ffn1 plain {t0:.1f}us | +bias+gelu+drop+pre {t1:.1f}us || d-ffn2 plain {t2:.1f}us | +drop*dgelu {t3:.1f}us")
for rep in range(2):
  for name, kw in [("plain", {}), ("all4", dict(bias=b1, gelu=True, out_pre=u, dropout_p=0.1, seed=3)), ("bias", dict(bias=b1)), ("bias+gelu", dict(bias=b1, gelu=True)), ("bias+pre", dict(bias=b1, out_pre=u)),
                   ("bias+gelu+pre", dict(bias=b1, gelu=True, out_pre=u)), ("bias+drop", dict(bias=b1, dropout_p=0.1, seed=3)),
                   ("bias+gelu+drop", dict(bias=b1, gelu=True, dropout_p=0.1, seed=3))]:
    t = bench(lambda: ops.gemm(x, w1, M, I, H, **kw), iters=100)
    print(f"  ffn1 {name:16s} {t:.1f}us")
