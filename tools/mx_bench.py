"""MXFP8 GEMM (csrc/mxgemm.hip) against the bf16 kernels on the encoder-layer shapes of 16 x 10 s (M = 8192), base and large."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aptai_amd import ops
from tools.attn_sweep import bench
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(*s, device="cuda", generator=g).to(torch.bfloat16)
for name, (M, N, K) in {"qkv": (8192, 2304, 768), "out": (8192, 768, 768), "ffn1": (8192, 3072, 768), "ffn2": (8192, 768, 3072),
                        "large qkv": (8192, 3072, 1024), "large ffn1": (8192, 4096, 1024), "large ffn2": (8192, 1024, 4096)}.items():
    a, w = rnd(M, K), rnd(N, K)
    bias = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    wq, ws = ops.mx_quantize(w)
    aq = torch.empty(M, K, device="cuda", dtype=torch.uint8); a_s = torch.empty(M, K // 32, device="cuda", dtype=torch.uint8)
    t16 = bench(lambda: ops.gemm(a, w, M, N, K, bias=bias, out=out))
    tq = bench(lambda: ops.mx_quantize(a, out=(aq, a_s)))
    t8 = bench(lambda: ops.gemm_mxfp8(aq, a_s, wq, ws, M, N, K, bias=bias, out=out))
    fl = 2.0 * M * N * K
    print(f"{name:11s} bf16 {t16:6.1f} us ({fl / t16 / 1e6:5.0f} TF) | mxfp8 gemm {t8:6.1f} us ({fl / t8 / 1e6:5.0f} TF) + quantise {tq:5.1f} us "
          f"-> {t16 / (t8 + tq):.2f}x", flush=True)
