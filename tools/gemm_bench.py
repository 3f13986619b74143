#!/usr/bin/env python3
"""Per-shape timing of aptai_gemm_bf16 on the shapes of the APTAI train step (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aptai_amd import ops

def bench(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us

def main():
    tile = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    _g = ops.gemm
    ops.gemm = lambda *a, **kw: _g(*a, tile=tile, **kw)
    print(f"tile={tile}")
    M = 8192
    g = torch.Generator(device="cuda").manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g).to(torch.bfloat16)
    rows = []
    for name, (m, n, k) in {"qkv": (M, 2304, 768), "out": (M, 768, 768), "ffn1": (M, 3072, 768), "ffn2": (M, 768, 3072),
                            "proj": (M, 768, 512), "L-qkv": (4096, 3072, 1024), "L-ffn1": (4096, 4096, 1024), "L-ffn2": (4096, 1024, 4096)}.items():
        a, b = rnd(m, k), rnd(n, k)
        bias = torch.randn(n, device="cuda")
        res = rnd(m, n)
        t = bench(lambda: ops.gemm(a, b, m, n, k))
        t2 = bench(lambda: ops.gemm(a, b, m, n, k, bias=bias, residual=res))
        t3 = bench(lambda: ops.gemm(a, b, m, n, k, bias=bias, gelu=True, dropout_p=0.1, seed=1))
        rows.append(f"NT {name:7s} {m}x{n}x{k}: plain {t:7.1f}us {2*m*n*k/t/1e6:7.1f} TF | bias+res {t2:7.1f}us {2*m*n*k/t2/1e6:7.1f} TF | bias+gelu+drop {t3:7.1f}us {2*m*n*k/t3/1e6:7.1f} TF")
    # conv layers (implicit GEMM, overlapping rows)
    for name, (rows_out, kw) in {"conv1": (16 * 16384, 3), "conv4": (16 * 2048, 3), "conv5": (16 * 1024, 2)}.items():
        x = rnd(rows_out * 2 + 8, 512); w = rnd(512, kw * 512)
        t = bench(lambda: ops.gemm(x, w, rows_out, 512, kw * 512, lda=1024, gelu=True), iters=10)
        rows.append(f"NT {name:7s} {rows_out}x512x{kw*512}: gelu {t:7.1f}us {2*rows_out*512*kw*512/t/1e6:7.1f} TF")
    for name, (m, n, k) in {"d-ffn2": (M, 3072, 768), "d-ffn1": (M, 768, 3072), "d-out": (M, 768, 768), "d-qkv": (M, 768, 2304)}.items():
        a, b = rnd(m, k), rnd(k, n)
        t = bench(lambda: ops.gemm(a, b, m, n, k, b_kmajor=True))
        rows.append(f"NN {name:7s} {m}x{n}x{k}: {t:7.1f}us {2*m*n*k/t/1e6:7.1f} TF")
    for name, (m, n, k, s) in {"w-qkv": (2304, 768, M, 4), "w-out": (768, 768, M, 14), "w-ffn1": (3072, 768, M, 3), "w-ffn2": (768, 3072, M, 3),
                               "w-qkv/1": (2304, 768, M, 1), "w-ffn1/1": (3072, 768, M, 1), "w-out/4": (768, 768, M, 4), "w-ffn1/8": (3072, 768, M, 8)}.items():
        a, b = rnd(k, m), rnd(k, n)
        ws = torch.empty(s * m * n * 4 + 16, device="cuda", dtype=torch.uint8)
        out = torch.empty(m, n, device="cuda")
        t = bench(lambda: ops.gemm(a, b, m, n, k, a_kmajor=True, b_kmajor=True, out_f32=True, split_k=s, workspace=ws, out=out))
        rows.append(f"TN {name:8s} {m}x{n}x{k} split{s}: {t:7.1f}us {2*m*n*k/t/1e6:7.1f} TF")
    x = rnd(M, 3072)
    t = bench(lambda: ops.colsum(x, M, 3072)); rows.append(f"colsum 8192x3072: {t:.1f}us  {M*3072*2/t/1e6:.2f} TB/s")
    x = rnd(M, 768)
    t = bench(lambda: ops.colsum(x, M, 768)); rows.append(f"colsum 8192x768: {t:.1f}us")
    print("\n".join(rows))

if __name__ == "__main__":
    main()
