"""Exact-integer probes of aptai_gemm_mxfp8: operand layout (unit scales), then block scales one at a time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aptai_amd import ops
g = torch.Generator().manual_seed(0)
M, N, K = 128, 128, 256
ai = torch.randint(-2, 3, (M, K), generator=g).float()
bi = torch.randint(-2, 3, (N, K), generator=g).float()
aq = ai.to(torch.float8_e4m3fn).view(torch.uint8).cuda()
bq = bi.to(torch.float8_e4m3fn).view(torch.uint8).cuda()
one = torch.full((M, K // 32), 127, dtype=torch.uint8).cuda()
ref = ai @ bi.t()
out = ops.gemm_mxfp8(aq, one, bq, one.clone(), M, N, K).float().cpu()
print("unit scales: max err", (out - ref).abs().max().item(), "ref max", ref.abs().max().item())
if (out - ref).abs().max().item() > 0:
    # which k contribute?  one-hot probes
    for k0 in (0, 1, 15, 16, 31, 32, 33, 63, 64, 100, 255):
        a1 = torch.zeros(M, K); a1[:, k0] = 1.0
        b1 = torch.zeros(N, K); b1[:, k0] = torch.arange(N).float() % 7 + 1
        o = ops.gemm_mxfp8(a1.to(torch.float8_e4m3fn).view(torch.uint8).cuda(), one, b1.to(torch.float8_e4m3fn).view(torch.uint8).cuda(), one.clone(), M, N, K).float().cpu()
        print(" one-hot k", k0, "row0[:8]", o[0, :8].tolist(), "expected", (b1[:8, k0]).tolist())
# scale probes
for blk in range(K // 32):
    sa = one.clone(); sa[:, blk] = 128                      # x2 on A block blk
    o = ops.gemm_mxfp8(aq, sa, bq, one.clone(), M, N, K).float().cpu()
    r = ref + ai[:, blk * 32:(blk + 1) * 32] @ bi[:, blk * 32:(blk + 1) * 32].t()
    print("A scale x2 on block", blk, "max err", (o - r).abs().max().item())
    sb = one.clone(); sb[:, blk] = 126
    o = ops.gemm_mxfp8(aq, one.clone(), bq, sb, M, N, K).float().cpu()
    r = ref - 0.5 * ai[:, blk * 32:(blk + 1) * 32] @ bi[:, blk * 32:(blk + 1) * 32].t()
    print("B scale /2 on block", blk, "max err", (o - r).abs().max().item())
