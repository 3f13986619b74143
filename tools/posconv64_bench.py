import sys, os
sys.path.insert(0, os.getcwd())
import torch
from aptai_amd import ops
from tools.gemm_bench import bench
B, Tp, H, G, Kw = 8, 512, 1024, 16, 128
Cg, pad = H // G, Kw // 2
rows_p, K = Tp + 2 * pad, Kw * Cg
g = torch.Generator(device="cuda").manual_seed(0)
xg = torch.randn(G * B * rows_p * Cg + Cg * 8, device="cuda", generator=g).to(torch.bfloat16)
wf = (torch.randn(G, Cg, K, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
bias = torch.randn(H, device="cuda")
res = torch.randn(B * Tp, H, device="cuda", generator=g).to(torch.bfloat16)
batch = dict(outer=B, inner=G, a=(rows_p * Cg, B * rows_p * Cg), b=(0, Cg * K), c=(Tp * H, Cg), bias=(0, Cg), res=(Tp * H, Cg), aux=(Tp * H, Cg))
for first in (0, 1):
    ref = ops.gemm(xg[first * Cg:], wf, Tp, Cg, K, lda=Cg, ldb=K, out=torch.empty(B * Tp, H, device="cuda", dtype=torch.bfloat16), ldc=H, bias=bias, gelu=True, residual=res, ldr=H, batch=batch, tile=128)
    got = ops.posconv_gemm(xg, wf, torch.empty_like(ref), B, Tp, H, G, Kw, pad, first_row=first, bias=bias, gelu=True, residual=res)
    torch.cuda.synchronize()
    print("first_row", first, "max abs diff", (got.float() - ref.float()).abs().max().item(), "scale", ref.float().abs().max().item())
out = torch.empty_like(ref)
t0 = bench(lambda: ops.gemm(xg, wf, Tp, Cg, K, lda=Cg, ldb=K, out=out, ldc=H, bias=bias, gelu=True, residual=res, ldr=H, batch=batch, tile=128), iters=20)
t1 = bench(lambda: ops.posconv_gemm(xg, wf, out, B, Tp, H, G, Kw, pad, bias=bias, gelu=True, residual=res), iters=20)
fl = 2.0 * B * Tp * H * K
print(f"large posconv: implicit GEMM {t0:.1f} us ({fl/t0/1e6:.0f} TF) | kernel {t1:.1f} us ({fl/t1/1e6:.0f} TF)")
