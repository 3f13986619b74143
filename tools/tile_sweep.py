#!/usr/bin/env python3
"""Every transformer GEMM of the base train step with its real epilogue, across the tile kernels (auto = the dispatcher's pick)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aptai_amd import ops
from tools.gemm_bench import bench
M, H, I = 8192, 768, 3072
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(*s, device="cuda", generator=g).to(torch.bfloat16)
x, xi = rnd(M, H), rnd(M, I)
res = rnd(M, H)
u = rnd(M, I)
cases = {
    "qkv  NT 8192x2304x768  +bias": (lambda t: ops.gemm(x, wqkv, M, 3 * H, H, bias=b3, tile=t)),
    "out  NT 8192x768x768   +bias+res+drop": (lambda t: ops.gemm(x, wo, M, H, H, bias=b1, residual=res, dropout_p=0.1, seed=3, tile=t)),
    "ffn1 NT 8192x3072x768  +bias+gelu+drop+pre": (lambda t: ops.gemm(x, w1, M, I, H, bias=bI, gelu=True, out_pre=u, pre_dgelu=True, dropout_p=0.1, seed=3, tile=t)),
    "ffn2 NT 8192x768x3072  +bias+res+drop": (lambda t: ops.gemm(xi, w2, M, H, I, bias=b1, residual=res, dropout_p=0.1, seed=3, tile=t)),
    "dffn2 NN 8192x3072x768 *aux": (lambda t: ops.gemm(x, w2, M, I, H, b_kmajor=True, mul_aux=u, tile=t)),
    "dffn1 NN 8192x768x3072 +res": (lambda t: ops.gemm(xi, w1, M, H, I, b_kmajor=True, residual=res, tile=t)),
    "dout NN 8192x768x768": (lambda t: ops.gemm(x, wo, M, H, H, b_kmajor=True, tile=t)),
    "dqkv NN 8192x768x2304  +res": (lambda t: ops.gemm(x3, wqkv, M, H, 3 * H, b_kmajor=True, residual=res, tile=t)),
}
wqkv, wo, w1, w2 = rnd(3 * H, H), rnd(H, H), rnd(I, H), rnd(H, I)
b3, b1, bI = torch.randn(3 * H, device="cuda"), torch.randn(H, device="cuda"), torch.randn(I, device="cuda")
x3 = rnd(M, 3 * H)
for rep in range(2):
    for name, fn in cases.items():
        row = []
        for t in (0, 64, 128, 192, 256):
            row.append(f"{'auto' if t == 0 else 't' + str(t)} {bench(lambda: fn(t), iters=60):6.1f}")
        print(f"{name:44s} " + " | ".join(row), flush=True)
