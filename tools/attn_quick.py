"""fwd / bwd attention time at the hot-path shape (B=16, heads=12, Tp=512) for len in {64, 499}; p from argv."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aptai_amd import ops
from tools.attn_sweep import bench


def main():
    B, Tp, heads = 16, 512, 12
    H = heads * 64
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(B * Tp, 3 * H, device="cuda", generator=g).to(torch.bfloat16)
    dctx = torch.randn(B * Tp, H, device="cuda", generator=g).to(torch.bfloat16)
    for p in (0.0, 0.1):
        row = []
        for L in (64, 499):
            lens = torch.full((B,), L, dtype=torch.int32, device="cuda")
            t = bench(lambda: ops.attention_fwd(qkv, lens, B, Tp, H, heads, dropout_p=p, seed=1))
            ctx, st = ops.attention_fwd(qkv, lens, B, Tp, H, heads, dropout_p=p, seed=1)
            tb = bench(lambda: ops.attention_bwd(qkv, lens, ctx, dctx, st, B, Tp, H, heads, dropout_p=p, seed=1, dctx_zero_beyond_len=True))
            tp = bench(lambda: ops.attention_bwd(qkv, lens, ctx, dctx, st, B, Tp, H, heads, dropout_p=p, seed=1, dctx_zero_beyond_len=True,
                                                 q_prescaled=True))
            row.append(f"len {L}: fwd {t:5.1f} bwd {tb:5.1f} bwd(prescaled q) {tp:5.1f}")
        print(f"stagger={os.environ.get('APTAI_ATTN_STAGGER', '0')} p={p}: " + " | ".join(row), flush=True)


if __name__ == "__main__":
    main()
