"""A few launches of each attention kernel at the hot-path shape (B=16, heads=12, Tp=512, len=499) for rocprofv3 --pmc runs."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aptai_amd import ops


def main():
    B, Tp, heads = 16, 512, 12
    H = heads * 64
    p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(B * Tp, 3 * H, device="cuda", generator=g).to(torch.bfloat16)
    dctx = torch.randn(B * Tp, H, device="cuda", generator=g).to(torch.bfloat16)
    lens = torch.full((B,), 499, dtype=torch.int32, device="cuda")
    for _ in range(5):
        ctx, st = ops.attention_fwd(qkv, lens, B, Tp, H, heads, dropout_p=p, seed=1)
        ops.attention_bwd(qkv, lens, ctx, dctx, st, B, Tp, H, heads, dropout_p=p, seed=1, dctx_zero_beyond_len=True)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
