import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aptai_amd import ops
M, N, K = 128, 128, 256
aq = torch.full((M, K), 0x38, dtype=torch.uint8).cuda()
bq = torch.full((N, K), 0x38, dtype=torch.uint8).cuda()
one = torch.full((M, K // 32), 127, dtype=torch.uint8).cuda()
def run(sa, sb, name):
    o = ops.gemm_mxfp8(aq, sa, bq, sb, M, N, K).float().cpu()
    print(name, "out[0,0]", o[0, 0].item(), "out[5,70]", o[5, 70].item(), "out[100,3]", o[100, 3].item(), "unique", torch.unique(o).tolist()[:6])
run(one, one.clone(), "unit")
sa = torch.full_like(one, 128); run(sa, one.clone(), "A all 128 (expect 512)")
sa = one.clone(); sa[:, 0] = 128; run(sa, one.clone(), "A blk0 128 (expect 288)")
sa = one.clone(); sa[:, 5] = 129; run(sa, one.clone(), "A blk5 129 (expect 352)")
sa = one.clone(); sa[7, :] = 128; run(sa, one.clone(), "A row7 128 (row 7 -> 512)")
sb = one.clone(); sb[:, 2] = 128; run(one.clone(), sb, "B blk2 128 (expect 288)")
print("sa sample", sa[:2].tolist())
