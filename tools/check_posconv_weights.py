import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aptai_amd import ops
g = torch.Generator().manual_seed(0)
H, Cg, Kw = 768, 48, 128
v = (torch.randn(H, Cg, Kw, generator=g) * 0.02).cuda()
gain = (torch.rand(1, 1, Kw, generator=g) + 0.5).cuda()
wf, wd, norm = ops.posconv_weight(v, gain.reshape(-1), 16)
torch.cuda.synchronize()
ref_norm = v.double().pow(2).sum(dim=(0, 1)).sqrt().cpu()
w = (v.double() * gain.double() / ref_norm.cuda().view(1, 1, Kw)).cpu()              # [H][Cg][Kw]
ref_wf = w.view(16, Cg, Cg, Kw).permute(0, 1, 3, 2).reshape(16, Cg, Kw * Cg)          # [grp][n][kk][c]
ref_wd = w.view(16, Cg, Cg, Kw).flip(-1).permute(0, 2, 3, 1).reshape(16, Cg, Kw * Cg)  # [grp][c][kk'][n]
print("norm rel err", ((norm.cpu().double() - ref_norm).abs() / ref_norm).max().item())
print("wf max err", (wf.cpu().double() - ref_wf).abs().max().item(), "scale", ref_wf.abs().max().item())
print("wd max err", (wd.cpu().double() - ref_wd).abs().max().item())
np.save(sys.argv[1], wf.float().cpu().numpy())
