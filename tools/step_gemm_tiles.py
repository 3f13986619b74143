#!/usr/bin/env python3
"""Every bf16-output GEMM of one transformer layer of the APTAI train step (B = 16 x 10 s, base) with the epilogue the model gives it,
per forced tile and under the default rule (run on the GPU box): the table the tile rule in aptai_gemm_bf16 is fitted to."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from aptai_amd import ops
from tools.gemm_round import bench


def main():
    M, H, I = 8192, 768, 3072
    g = torch.Generator(device="cuda").manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g).to(torch.bfloat16)
    x, xi = rnd(M, H), rnd(M, I)
    res, aux = rnd(M, H), rnd(M, I)
    pre = torch.empty(M, I, device="cuda", dtype=torch.bfloat16)
    bias = {n: torch.randn(n, device="cuda") for n in (H, 3 * H, I)}
    wqkv, wo, w1, w2 = rnd(3 * H, H), rnd(H, H), rnd(I, H), rnd(H, I)
    dqkv = rnd(M, 3 * H)
    qs = ops.attention_qscale(H, 12)
    cases = [
        ("fwd qkv   bias+colscale", lambda t: ops.gemm(x, wqkv, M, 3 * H, H, bias=bias[3 * H], colscale=(H, qs), tile=t), 2.0 * M * 3 * H * H),
        ("fwd out   bias+res+drop", lambda t: ops.gemm(x, wo, M, H, H, bias=bias[H], residual=res, dropout_p=0.1, seed=1, tile=t), 2.0 * M * H * H),
        ("fwd ffn1  bias+gelu+drop+dgelu", lambda t: ops.gemm(x, w1, M, I, H, bias=bias[I], gelu=True, out_pre=pre, pre_dgelu=True, dropout_p=0.1,
                                                            seed=1, tile=t), 2.0 * M * I * H),
        ("fwd ffn2  bias+res+drop", lambda t: ops.gemm(xi, w2, M, H, I, bias=bias[H], residual=res, dropout_p=0.1, seed=1, tile=t), 2.0 * M * H * I),
        ("bwd ffn2  NN mul_aux", lambda t: ops.gemm(x, w2, M, I, H, b_kmajor=True, mul_aux=aux, tile=t), 2.0 * M * I * H),
        ("bwd ffn1  NN residual", lambda t: ops.gemm(xi, w1, M, H, I, b_kmajor=True, residual=res, tile=t), 2.0 * M * H * I),
        ("bwd ffn1  NN plain", lambda t: ops.gemm(xi, w1, M, H, I, b_kmajor=True, tile=t), 2.0 * M * H * I),
        ("bwd out   NN plain", lambda t: ops.gemm(x, wo, M, H, H, b_kmajor=True, tile=t), 2.0 * M * H * H),
        ("bwd qkv   NN residual", lambda t: ops.gemm(dqkv, wqkv, M, H, 3 * H, b_kmajor=True, residual=res, tile=t), 2.0 * M * H * 3 * H),
        ("bwd qkv   NN plain", lambda t: ops.gemm(dqkv, wqkv, M, H, 3 * H, b_kmajor=True, tile=t), 2.0 * M * H * 3 * H),
    ]
    tiles = (0, 64, 128, 192, 256)
    print(f"{'case':34s}" + "".join(f"{('auto' if t == 0 else t):>9}" for t in tiles) + "   best")
    tot = {t: 0.0 for t in tiles}
    best_sum = 0.0
    for name, fn, fl in cases:
        ts = {}
        for t in tiles:
            try:
                ts[t] = bench(lambda: fn(t), iters=30)
            except Exception:  # noqa: BLE001
                ts[t] = float("nan")
        forced = {t: v for t, v in ts.items() if t and v == v}
        b = min(forced, key=forced.get)
        best_sum += forced[b]
        for t in tiles:
            if ts[t] == ts[t]:
                tot[t] += ts[t]
        print(f"{name:34s}" + "".join(f"{ts[t]:9.1f}" for t in tiles) + f"   {b} ({fl / forced[b] / 1e6:.0f} TF/s)", flush=True)
    print(f"{'sum (us)':34s}" + "".join(f"{tot[t]:9.1f}" for t in tiles) + f"   best-of {best_sum:.1f}")


if __name__ == "__main__":
    main()
