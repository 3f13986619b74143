#!/usr/bin/env python3
"""First conv layer, backward (GroupNorm mode): aptai_conv0_bwd against torch autograd in fp64 on a small case, and its time at 16 x 10 s.
APTAI_CONV0_BWD_MFMA=0 / 1 (one process each) selects the all-vector or the matrix-pipe weight pass (run on the GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from aptai_amd import ops
from tools.gemm_round import bench


def case(B, S, seed):
    g = torch.Generator().manual_seed(seed)
    audio = torch.randn(B, S, generator=g)
    w = torch.randn(512, 1, 10, generator=g) * 0.3
    gamma = 1.0 + 0.1 * torch.randn(512, generator=g)
    beta = 0.1 * torch.randn(512, generator=g)
    T = (S - 10) // 5 + 1
    Ta = (T + 63) // 64 * 64
    dy = (torch.randn(B, Ta, 512, generator=g) * 0.5).to(torch.bfloat16)
    dy[:, T:] = 0
    return audio, w, gamma, beta, dy, T, Ta


def main():
    audio, w, gamma, beta, dy, T, Ta = case(2, 16000, 0)
    dev = "cuda"
    out = torch.empty(2, Ta, 512, device=dev, dtype=torch.bfloat16)
    stats = ops.conv0_fwd(audio.to(dev), w.to(dev), None, gamma.to(dev), beta.to(dev), 0, out, T, Ta, want_stats=True)
    dw, _, dg, db = ops.conv0_bwd(audio.to(dev), w.to(dev), None, gamma.to(dev), beta.to(dev), 0, dy.to(dev), T, Ta, stats)
    # fp64 reference with the kernels' GELU (x * sigmoid(x (a1 + a3 x^2 + a5 x^4)), tools/gelu_fit.py)
    a1, a3, a5 = 1.59499531, 7.40885562e-2, -7.23764583e-4
    wd, gd, bd = w.double().requires_grad_(), gamma.double().requires_grad_(), beta.double().requires_grad_()
    v = torch.nn.functional.conv1d(audio.double()[:, None], wd, stride=5)                       # [B][512][T]
    xh = (v - v.mean(-1, keepdim=True)) / torch.sqrt(v.var(-1, unbiased=False, keepdim=True) + 1e-5)
    z = xh * gd[None, :, None] + bd[None, :, None]
    y = z * torch.sigmoid(z * (a1 + a3 * z * z + a5 * z ** 4))
    (y * dy[:, :T].double().transpose(1, 2)).sum().backward()
    for name, got, ref in (("dweight", dw, wd.grad), ("dgamma", dg, gd.grad), ("dbeta", db, bd.grad)):
        err = (got.double().cpu() - ref).abs().max().item() / ref.abs().max().item()
        print(f"{name}: max err / max |ref| = {err:.2e}")
        assert err < 2e-3, name
    audio, w, gamma, beta, dy, T, Ta = case(16, 160000, 1)
    a_, w_, g_, b_, d_ = audio.to(dev), w.to(dev), gamma.to(dev), beta.to(dev), dy.to(dev)
    out = torch.empty(16, Ta, 512, device=dev, dtype=torch.bfloat16)
    stats = ops.conv0_fwd(a_, w_, None, g_, b_, 0, out, T, Ta, want_stats=True)
    t = bench(lambda: ops.conv0_bwd(a_, w_, None, g_, b_, 0, d_, T, Ta, stats), iters=10)
    print(f"conv0_bwd at 16 x 10 s (APTAI_CONV0_BWD_MFMA={os.environ.get('APTAI_CONV0_BWD_MFMA', '1')}): {t:.0f} us")


if __name__ == "__main__":
    main()
