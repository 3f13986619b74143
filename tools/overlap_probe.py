"""Do an HBM-bound stream (Adam-like elementwise passes) and an MFMA-bound stream (the step's backward GEMMs) overlap?"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aptai_amd import ops


def main():
    M, H, I = 8192, 768, 3072
    g = torch.Generator(device="cuda").manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g).to(torch.bfloat16)
    x, xi, w1, w2 = rnd(M, H), rnd(M, I), rnd(I, H), rnd(H, I)
    p = torch.randn(7_100_000, device="cuda")           # one layer's parameters
    m, v, gr = torch.zeros_like(p), torch.zeros_like(p), torch.randn_like(p)

    def gemms(n=24):
        for _ in range(n):
            ops.gemm(x, w2, M, I, H, b_kmajor=True)       # dgrad-shaped
            ops.gemm(xi, w1, M, H, I, b_kmajor=True)

    def adam(n=24):
        for _ in range(n):
            torch._foreach_add_([m, v], [gr, gr], alpha=0.1)       # ~ 8 + 8 + 8 B per element per pass
            p.addcdiv_(m, v.abs_().add_(1.0), value=-1e-3)

    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def timed(fn, s):
        with torch.cuda.stream(s):
            fn(2); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s); fn(); e1.record(s)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    tg, ta = timed(gemms, s1), timed(adam, s2)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    torch.cuda.synchronize()
    with torch.cuda.stream(s1):
        e[0].record(s1); gemms(); e[1].record(s1)
    with torch.cuda.stream(s2):
        e[2].record(s2); adam(); e[3].record(s2)
    torch.cuda.synchronize()
    print(f"gemms alone {tg:.3f} ms, elementwise alone {ta:.3f} ms, sum {tg + ta:.3f}")
    print(f"concurrent: gemms {e[0].elapsed_time(e[1]):.3f} ms, elementwise {e[2].elapsed_time(e[3]):.3f} ms, span {max(e[0].elapsed_time(e[1]), e[0].elapsed_time(e[3])):.3f} ms")


if __name__ == "__main__":
    main()
