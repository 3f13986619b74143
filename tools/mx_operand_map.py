import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aptai_amd import ops
g = torch.Generator().manual_seed(0)
M, N, K = 128, 128, 256
ai = torch.randint(-2, 3, (M, K), generator=g).float()
bi = torch.randint(-2, 3, (N, K), generator=g).float()
aq = ai.to(torch.float8_e4m3fn).view(torch.uint8).cuda()
bq = bi.to(torch.float8_e4m3fn).view(torch.uint8).cuda()
one = torch.full((M, K // 32), 127, dtype=torch.uint8).cuda()
ref = ai @ bi.t()
for (r0, b0) in ((7, 0), (7, 1), (40, 2), (100, 5)):
    sa = one.clone(); sa[r0, b0] = 128
    o = ops.gemm_mxfp8(aq, sa, bq, one.clone(), M, N, K).float().cpu()
    d = o - ref
    rows = torch.nonzero(d.abs().sum(1)).flatten().tolist()
    want = ai[r0, b0 * 32:(b0 + 1) * 32] @ bi[:, b0 * 32:(b0 + 1) * 32].t()
    print(f"A scale at (row {r0}, blk {b0}): changed rows {rows}; ", end="")
    for r in rows[:3]:
        # which block's contribution matches?
        best = None
        for b in range(K // 32):
            c = ai[r, b * 32:(b + 1) * 32] @ bi[:, b * 32:(b + 1) * 32].t()
            if torch.allclose(d[r], c): best = b
        print(f"row {r}: delta == contribution of block {best}; ", end="")
    print()
sb = one.clone(); sb[9, 3] = 128
o = ops.gemm_mxfp8(aq, one.clone(), bq, sb, M, N, K).float().cpu()
d = o - ref
cols = torch.nonzero(d.abs().sum(0)).flatten().tolist()
print("B scale at (row 9, blk 3): changed cols", cols)
for c_ in cols[:3]:
    for b in range(K // 32):
        c = ai[:, b * 32:(b + 1) * 32] @ bi[c_, b * 32:(b + 1) * 32]
        if torch.allclose(d[:, c_], c): print("  col", c_, "delta == contribution of block", b)
