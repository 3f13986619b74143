#!/usr/bin/env python3
"""Round 4: the 256 x 192 tile (tile = 448, csrc/gemm_t4.hip) against the other tiles of aptai_gemm_bf16 and the vendor library
(torch -> hipBLASLt; measurement only) on the bf16-output GEMMs of an encoder layer.  Random operands; the variants of one shape
are timed INTERLEAVED in rounds (guide 5.4 rule 24): median and minimum over the rounds.
    python tools/gemm_t4_bench.py [--rounds 7] [--heavy]"""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from aptai_amd import ops


def time_once(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--heavy", action="store_true", help="with the training epilogues of the step instead of plain GEMMs")
    ap.add_argument("--tiles", default="0,64,128,192,256,448")
    a = ap.parse_args()
    tiles = [int(t) for t in a.tiles.split(",")]
    g = torch.Generator(device="cuda").manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g).to(torch.bfloat16)
    M = 8192
    cases = [("NT qkv", M, 2304, 768, False), ("NT ffn1", M, 3072, 768, False), ("NT out", M, 768, 768, False), ("NT ffn2", M, 768, 3072, False),
             ("NT L-qkv", 4096, 3072, 1024, False), ("NT L-ffn1", 4096, 4096, 1024, False), ("NT L-ffn2", 4096, 1024, 4096, False),
             ("NN d-ffn2", M, 3072, 768, True), ("NN d-ffn1", M, 768, 3072, True), ("NN d-qkv", M, 768, 2304, True),
             ("NN L-d-ffn2", 4096, 4096, 1024, True), ("NN L-d-qkv", 4096, 1024, 3072, True)]
    print(f"{'case':12s} {'M x N x K':>18s} | " + " ".join(f"{('t' + str(t)) if t else 'auto':>13s}" for t in tiles) + f" {'library':>13s}   (median / min us)")
    for name, m, n, k, bkm in cases:
        x = rnd(m, k)
        w = rnd(k, n) if bkm else rnd(n, k)
        kw = dict(b_kmajor=True) if bkm else {}
        if a.heavy:
            if name.endswith("ffn1") and not bkm:
                pre = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
                kw.update(bias=torch.randn(n, device="cuda"), gelu=True, out_pre=pre, pre_dgelu=True, dropout_p=0.1, seed=5)
            elif bkm and "d-ffn2" in name:
                kw.update(mul_aux=rnd(m, n))
            elif bkm:
                kw.update(residual=rnd(m, n))
            elif "qkv" in name:
                kw.update(bias=torch.randn(n, device="cuda"), colscale=(n // 3, 0.125 * 1.4426950408889634))
            else:
                kw.update(bias=torch.randn(n, device="cuda"), residual=rnd(m, n), dropout_p=0.1, seed=5)
        out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
        fns = {}
        for t in tiles:
            def f(t=t):
                ops.gemm(x, w, m, n, k, out=out, tile=t, **kw)
            try:
                f()
                torch.cuda.synchronize()
                fns[t] = f
            except Exception as e:                      # tile not built for this layout / shape
                fns[t] = None
        fns["lib"] = (lambda: torch.matmul(x, w)) if bkm else (lambda: torch.nn.functional.linear(x, w))
        for f in fns.values():
            if f is not None:
                for _ in range(3):
                    f()
        torch.cuda.synchronize()
        res = {kk: [] for kk in fns}
        for _ in range(a.rounds):
            for kk, f in fns.items():
                if f is not None:
                    res[kk].append(time_once(f, a.iters))
        cells = []
        for kk in list(tiles) + ["lib"]:
            cells.append(f"{statistics.median(res[kk]):6.1f}/{min(res[kk]):6.1f}" if res[kk] else f"{'-':>13s}")
        fl = 2.0 * m * n * k
        best = min((min(v), kk) for kk, v in res.items() if v and kk != "lib")
        print(f"{name:12s} {m:6d}x{n:5d}x{k:5d} | " + " ".join(f"{c:>13s}" for c in cells) + f"   best own: {best[1]} {fl / best[0] / 1e6:5.0f} TF/s, lib {fl / min(res['lib']) / 1e6:5.0f}")


if __name__ == "__main__":
    main()
