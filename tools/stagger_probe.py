#!/usr/bin/env python3
"""Round-4 experiment: does STAGGERING the two co-resident blocks of a CU pay on an epilogue-heavy GEMM (FFN1 forward: bias + GELU + saved
derivative + dropout, 60-72 us of which ~28 are vector work while the matrix pipe idles)?  -DAPTAI_EXP_STAGGER build
(tools/ab/stag/lib_stagger.so as APTAI_HIP_LIB): every block of the 128-row kernel draws an arrival number from a per-CU counter
(HW_ID / XCC_ID) and the SECOND arrival on a CU (first round only: the blocks that follow in its slot inherit the offset) sleeps `s` x ~1 us
before it starts, so that one block's epilogue meets the other's main loop.  Results are unchanged (timing only).  Prints the launch time per
delay for the model's M = 8192 (3 rounds of tiles) and for M = 32768 (12 rounds: the steady state, the delay itself amortised)."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from aptai_amd import ops


def run(M):
    g = torch.Generator(device="cuda").manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g).to(torch.bfloat16)
    N, K = 3072, 768
    x, w = rnd(M, K), rnd(N, K)
    bias = torch.randn(N, device="cuda", generator=g)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    pre = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    kw = dict(gelu=True, pre_dgelu=True, dropout_p=0.1, seed=5)
    cnt = torch.zeros(2048, device="cuda", dtype=torch.int32)

    def single(sleep=0):
        if sleep:
            cnt.zero_()
            os.environ["APTAI_EXP_SLEEP"] = str(sleep)
            os.environ["APTAI_EXP_CU_COUNT"] = str(cnt.data_ptr())
        ops.gemm(x, w, M, N, K, out=out, bias=bias, out_pre=pre, tile=128, **kw)
        os.environ["APTAI_EXP_SLEEP"] = "0"

    def timeit(fn, iters=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3

    res = {}
    for rnd_i in range(4):
        for sl in (0, 2, 4, 6, 8, 10, 12, 16):
            res.setdefault(sl, []).append(timeit(lambda: single(sl)))
    print(f"[{M}] x {N} x {K}, FFN1 forward epilogue, 128 x 128 tiles, {M // 128 * N // 128} blocks ({M // 128 * N // 128 / 512:.0f} rounds)")
    for k, v in res.items():
        print(f"  second block of each CU delayed {k:2d} x ~1 us: median {statistics.median(v):6.1f} us  min {min(v):6.1f}"
              + ("   (includes a ~2 us memset of the counters)" if k else ""))
    print("  CUs with a delayed block in the last launch:", int((cnt >= 2).sum()), "of", int((cnt > 0).sum()))


def main():
    run(8192)
    run(32768)


if __name__ == "__main__":
    main()
