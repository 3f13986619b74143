#!/usr/bin/env python3
"""Exact-index mode: the split-operand GEMM shapes of one transformer layer (K' = 3 K, split-out epilogues) per forced tile - the table
ops.gemm_split's tile rule is fitted to.  Usage: python tools/exact_gemm_tiles.py [pieces]"""
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
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    g = torch.Generator(device="cuda").manual_seed(0)
    M, H, I = 8192, 768, 3072
    shapes = [("q|k|v split-out", M, 3 * H, H, dict(split_out=P, split_bcol=H)), ("out-proj fp32 + residual", M, H, H, dict()),
              ("ffn1 gelu split-out", M, I, H, dict(split_out=P, gelu=True)), ("ffn2 fp32", M, H, I, dict())]
    for name, m, n, k, kw in shapes:
        a = torch.randn(m, P * k, device="cuda", generator=g).to(torch.bfloat16)
        w = torch.randn(n, P * k, device="cuda", generator=g).to(torch.bfloat16)
        bias = torch.randn(n, device="cuda", generator=g)
        line = f"{name:28s} [{m}] x {n} x {P * k}:"
        for tile in (128, 192, 256, 448):
            try:
                out = None
                fn = lambda: ops.gemm(a, w, m, n, P * k, out_f32=True, bias=bias, tile=tile, **kw)
                fn()
                ts = [timeit(fn) for _ in range(3)]
                line += f"  tile {tile}: {statistics.median(ts):7.1f} us"
            except Exception as e:       # noqa: BLE001 - a tile that refuses the shape / epilogue
                line += f"  tile {tile}: refused"
        print(line, flush=True)


if __name__ == "__main__":
    main()
