"""Positional-conv GEMM (grouped Conv1d k=128, 16 groups of 48 channels) as the batched overlapping-row GEMM the model issues."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aptai_amd import ops
from tools.gemm_bench import bench
B, Tp, H, G, Kw = 16, 512, 768, 16, 128
Cg, pad = H // G, Kw // 2
rows_p, K = Tp + 2 * pad, Kw * Cg
g = torch.Generator(device="cuda").manual_seed(0)
xg = torch.randn(G * B * rows_p * Cg + Cg * 8, device="cuda", generator=g).to(torch.bfloat16)
wf = (torch.randn(G, Cg, K, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
out = torch.empty(B * Tp, H, device="cuda", dtype=torch.bfloat16)
bias = torch.randn(H, device="cuda")
batch = dict(outer=B, inner=G, a=(rows_p * Cg, B * rows_p * Cg), b=(0, Cg * K), c=(Tp * H, Cg), bias=(0, Cg), res=(Tp * H, Cg), aux=(Tp * H, Cg))
fl = 2.0 * B * Tp * H * K
for tile in (128,):
    t = bench(lambda: ops.gemm(xg, wf, Tp, Cg, K, lda=Cg, ldb=K, out=out, ldc=H, bias=bias, gelu=True, batch=batch, tile=tile), iters=20)
    print(f"tile={tile}: {t:.1f} us  {fl/t/1e6:.0f} TF useful")
res = torch.randn(B * Tp, H, device="cuda", generator=g).to(torch.bfloat16)
ref = ops.gemm(xg, wf, Tp, Cg, K, lda=Cg, ldb=K, out=torch.empty_like(out), ldc=H, bias=bias, gelu=True, residual=res, ldr=H, batch=batch, tile=128)
got = ops.posconv_gemm(xg, wf, torch.empty_like(out), B, Tp, H, G, Kw, pad, bias=bias, gelu=True, residual=res)
torch.cuda.synchronize()
print("posconv kernel vs implicit GEMM: max abs diff", (got.float() - ref.float()).abs().max().item(), "scale", ref.float().abs().max().item())
t = bench(lambda: ops.posconv_gemm(xg, wf, out, B, Tp, H, G, Kw, pad, bias=bias, gelu=True, residual=res), iters=20)
print(f"tile=posconv kernel: {t:.1f} us  {fl/t/1e6:.0f} TF useful")
# ---- weight gradient: packed dU x packed X over the long frame axis
dug = torch.randn(G * B * rows_p * Cg + Cg * 8, device="cuda", generator=g).to(torch.bfloat16)
kred = B * rows_p - 2 * pad
ref_dw = torch.empty((G, Cg, K), device="cuda", dtype=torch.float32)
ops.gemm(dug[pad * Cg:], xg, Cg, K, kred, a_kmajor=True, b_kmajor=True, out_f32=True, lda=Cg, ldb=Cg, out=ref_dw, ldc=K,
         batch=dict(outer=1, inner=G, a=(0, B * rows_p * Cg), b=(0, B * rows_p * Cg), c=(0, Cg * K)))
got_dw = ops.posconv_wgrad(dug, xg, torch.empty_like(ref_dw), B, Tp, H, G, Kw, pad)
torch.cuda.synchronize()
print("posconv wgrad vs TN GEMM: max abs diff", (got_dw - ref_dw).abs().max().item(), "scale", ref_dw.abs().max().item())
flw = 2.0 * G * Cg * K * kred
t0 = bench(lambda: ops.gemm(dug[pad * Cg:], xg, Cg, K, kred, a_kmajor=True, b_kmajor=True, out_f32=True, lda=Cg, ldb=Cg, out=ref_dw, ldc=K,
                            batch=dict(outer=1, inner=G, a=(0, B * rows_p * Cg), b=(0, B * rows_p * Cg), c=(0, Cg * K))), iters=10)
t1 = bench(lambda: ops.posconv_wgrad(dug, xg, got_dw, B, Tp, H, G, Kw, pad), iters=10)
print(f"tile=wgrad TN GEMM {t0:.1f} us ({flw/t0/1e6:.0f} TF useful) | posconv wgrad kernel {t1:.1f} us ({flw/t1/1e6:.0f} TF useful)")
