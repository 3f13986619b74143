"""Fixed cost vs per-tile cost of the attention kernels: utterance length sweep at the hot-path shape (B=16, heads=12, Tp=512).
The kernels skip key tiles beyond `len`, so time(len) = fixed (prologue + epilogue) + tiles(len) * per-tile cost."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aptai_amd import ops


def bench(fn, reps=20, iters=10):
    """Device time per call: `reps` calls captured into one hipGraph (no host gaps), replayed `iters` times."""
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps):
            fn()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (iters * reps) * 1e3   # us

def main():
    B, Tp, heads = 16, 512, 12
    H = heads * 64
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(B * Tp, 3 * H, device="cuda", generator=g).to(torch.bfloat16)
    dctx = torch.randn(B * Tp, H, device="cuda", generator=g).to(torch.bfloat16)
    for p in (0.0, 0.1):
        for L in (64, 128, 256, 384, 512):
            lens = torch.full((B,), L, dtype=torch.int32, device="cuda")
            t = bench(lambda: ops.attention_fwd(qkv, lens, B, Tp, H, heads, dropout_p=p, seed=1))
            ctx, st = ops.attention_fwd(qkv, lens, B, Tp, H, heads, dropout_p=p, seed=1)
            tb = bench(lambda: ops.attention_bwd(qkv, lens, ctx, dctx, st, B, Tp, H, heads, dropout_p=p, seed=1, dctx_zero_beyond_len=True))
            print(f"p={p} len={L:4d}: fwd {t:6.1f} us | bwd {tb:6.1f} us", flush=True)



if __name__ == "__main__":
    main()
