import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aptai_amd import ops
from tools.gemm_bench import bench
B, Tp, heads = 16, 512, 12
H = heads * 64
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B * Tp, 3 * H, device="cuda", generator=g).to(torch.bfloat16)
dctx = torch.randn(B * Tp, H, device="cuda", generator=g).to(torch.bfloat16)
lens = torch.tensor([499] * 8 + [410, 430, 450, 470, 480, 490, 495, 499], dtype=torch.int32, device="cuda")
fl_f = 4 * B * heads * Tp * Tp * 64
for pre in (False, True):          # the model runs the pre-scaled form (Q columns scaled in the q|k|v GEMM epilogue)
    for p in (0.0, 0.1):
        t = bench(lambda: ops.attention_fwd(qkv, lens, B, Tp, H, heads, dropout_p=p, seed=1, q_prescaled=pre))
        ctx, st = ops.attention_fwd(qkv, lens, B, Tp, H, heads, dropout_p=p, seed=1, q_prescaled=pre)
        tb = bench(lambda: ops.attention_bwd(qkv, lens, ctx, dctx, st, B, Tp, H, heads, dropout_p=p, seed=1, dctx_zero_beyond_len=True,
                                             q_prescaled=pre))
        print(f"prescaled={int(pre)} p={p}: fwd {t:6.1f}us {fl_f/t/1e6:6.0f} TF | bwd {tb:6.1f}us {2.5*fl_f/tb/1e6:6.0f} TF(alg)", flush=True)
