#!/usr/bin/env python3
"""What ONE round of 256 x 256 tiles costs on the transformer shapes (stream-K planning): forced-tile timings of
aptai_gemm_bf16 at [8192] x {256 * n} x {768, 3072}, NT and NN, against the default tile rule (run on the GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from aptai_amd import ops


def bench(fn, iters=40):
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


def main():
    M = 8192
    g = torch.Generator(device="cuda").manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g).to(torch.bfloat16)
    print("layout shape tile: plain us (TF/s) | heavy epilogue us")
    shapes = ((3072, 768), (2304, 768), (768, 768), (768, 3072), (768, 2304), (2048, 768), (2048, 3072), (1024, 768))
    if len(sys.argv) > 1 and sys.argv[1] == "short":
        shapes = shapes[:4]
    if len(sys.argv) > 1 and sys.argv[1] == "dgrad":
        shapes = ((768, 768), (768, 2304), (768, 3072), (3072, 768))
    for (n, k) in shapes:
        for km in (False, True):
            a = rnd(M, k)
            b = rnd(k, n) if km else rnd(n, k)
            bias = torch.randn(n, device="cuda")
            res = rnd(M, n)
            pre = torch.empty(M, n, device="cuda", dtype=torch.bfloat16)
            for tile in (0, 257, 256, 128, 192, 64):
                if tile == 192 and n % 192:
                    continue
                try:
                    t = bench(lambda: ops.gemm(a, b, M, n, k, b_kmajor=km, tile=tile))
                    if km:
                        t2 = bench(lambda: ops.gemm(a, b, M, n, k, b_kmajor=km, tile=tile, mul_aux=res))
                    else:
                        t2 = bench(lambda: ops.gemm(a, b, M, n, k, tile=tile, bias=bias, gelu=True, dropout_p=0.1, seed=1, out_pre=pre,
                                                    pre_dgelu=True))
                except Exception as e:  # noqa: BLE001
                    print(f"{'NN' if km else 'NT'} {M}x{n}x{k} tile {tile}: {e}")
                    continue
                fl = 2.0 * M * n * k
                print(f"{'NN' if km else 'NT'} {M}x{n}x{k} tile {tile:3d}: {t:7.1f} us {fl / t / 1e6:7.1f} TF/s | {t2:7.1f} us {fl / t2 / 1e6:7.1f} TF/s",
                      flush=True)


if __name__ == "__main__":
    main()
