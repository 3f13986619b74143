import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aptai_amd import ops
from tools.gemm_bench import bench
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(*s, device="cuda", generator=g).to(torch.bfloat16)
for (m, n) in [(8192, 3072), (8192, 768)]:
    for k in (64, 128, 256, 512, 768, 1536, 3072):
        a, b = rnd(m, k), rnd(n, k)
        out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
        r = []
        for tile in (128, 256):
            t = bench(lambda: ops.gemm(a, b, m, n, k, tile=tile, out=out), iters=20)
            r.append(f"tile{tile} {t:7.1f}us {2*m*n*k/t/1e6:7.1f}TF")
        print(f"{m}x{n}x{k:5d}: " + " | ".join(r) + f" | out {m*n*2/1e6:.0f}MB", flush=True)
