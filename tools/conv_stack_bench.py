#!/usr/bin/env python3
"""The frozen conv feature encoder's layers 1-6 as the model issues them (overlapping-row GEMMs, lda = stride * 512, GELU epilogue)
at B = 16 x 10 s: forced-tile timings (run on the GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from aptai_amd import ops
from tools.gemm_round import bench


def main():
    B, C = 16, 512
    frames = [31999, 15999, 7999, 3999, 1999, 999, 499]
    kernel = [10, 3, 3, 3, 3, 2, 2]
    stride = [5, 2, 2, 2, 2, 2, 2]
    g = torch.Generator(device="cuda").manual_seed(0)
    total = {}
    for i in range(1, 7):
        k, s = kernel[i], stride[i]
        Mi = B * frames[i]
        rows_in = B * frames[i - 1] + 8
        a = (torch.randn(rows_in + 8, C, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
        w = (torch.randn(C, k * C, device="cuda", generator=g) * 0.03).to(torch.bfloat16)
        out = torch.empty(Mi + 8, C, device="cuda", dtype=torch.bfloat16)
        fl = 2.0 * Mi * C * k * C
        line = f"L{i} M={Mi} K={k * C}:"
        for tile in (0, 256, 257, 192, 128):
            if tile == 192 and C % 192:
                continue
            try:
                t = bench(lambda: ops.gemm(a, w, Mi, C, k * C, lda=s * C, out=out, ldc=C, gelu=True, tile=tile), iters=10)
                t0 = bench(lambda: ops.gemm(a, w, Mi, C, k * C, lda=s * C, out=out, ldc=C, tile=tile), iters=10)
            except Exception as e:  # noqa: BLE001
                line += f" | {tile}: {str(e)[:40]}"
                continue
            total[tile] = total.get(tile, 0.0) + t
            line += f" | {tile}: {t:6.1f} us {fl / t / 1e6:5.0f} TF (plain {t0:6.1f})"
        print(line, flush=True)
    print("sum over layers:", {k: round(v, 1) for k, v in total.items()})


if __name__ == "__main__":
    main()
