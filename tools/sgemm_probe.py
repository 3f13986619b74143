#!/usr/bin/env python3
"""The fp32 matrix-core GEMM of the Force_APTAI heads (aptai_sgemm_f32, csrc/force.hip) at the shapes of the BiLSTM's input / recurrent
weights (the four launches that carry half of the heads' 1.06 ms of fp32 GEMM time per step): time and fraction of the fp32 matrix peak
(256 CUs x 256 flop / clock x 2.0 GHz = 131 TFLOP/s).  Round 4 (development patches, not kept): without the LDS staging writes xproj takes 102
instead of 153 us, without the matrix instructions the four launches take 66-79 us - the two phases add up instead of overlapping; a 2-deep LDS
ring with the staging writes placed behind the matrix instructions gave 151 / 139 / 136 / 75 us against 153 / 124 / 128 / 72 (no gain at 3
instead of 5 waves per SIMD), so the kernel is as it was."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from aptai_amd import ops


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e3)
    return statistics.median(ts)


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    M = 8192
    x = torch.randn(M, 256, device="cuda", generator=g)
    wih = torch.randn(2048, 256, device="cuda", generator=g) * 0.05
    bias = torch.randn(2048, device="cuda", generator=g)
    dg = torch.randn(M, 2048, device="cuda", generator=g) * 0.1
    hout = torch.randn(M, 512, device="cuda", generator=g)
    cases = [("xproj   x W_ih^T + b     [8192 x 2048 x 256]", 2 * M * 2048 * 256, lambda: ops.linear_f32(x, wih, bias)),
             ("dx      dgates W_ih      [8192 x 256 x 2048]", 2 * M * 2048 * 256, lambda: ops.sgemm(dg, 2048, 1, wih, 256, 1, M, 256, 2048)),
             ("dW_ih   dgates^T x       [2048 x 256 x 8192]", 2 * M * 2048 * 256, lambda: ops.sgemm(dg, 1, 2048, x, 256, 1, 2048, 256, M)),
             ("dW_hh   dgates^T h_prev  [1024 x 256 x 8191]", 2 * (M - 1) * 1024 * 256, lambda: ops.sgemm(dg[1:], 1, 2048, hout, 512, 1, 1024, 256, M - 1))]
    for name, flops, fn in cases:
        t = timeit(fn)
        print(f"{name}: {t:7.1f} us  {flops / t / 1e6:6.1f} TFLOP/s = {flops / t / 1e6 / 131:.2f} of the fp32 matrix peak")
    print("lib", os.environ.get("APTAI_HIP_LIB", "(product)"))


if __name__ == "__main__":
    main()
