#!/usr/bin/env python3
"""First conv layer (GroupNorm mode) forward: time per call and agreement between the all-vector kernel and the fp32-MFMA kernel
(APTAI_CONV0_MFMA=0 / 1 select them per process: run twice)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aptai_amd import ops

B, S = 16, 160000
T0 = (S - 10) // 5 + 1
Ta = 32768
g = torch.Generator(device="cuda").manual_seed(0)
audio = torch.randn(B, S, device="cuda", generator=g)
w = torch.randn(512, 1, 10, device="cuda", generator=g) * 0.3
gamma = torch.rand(512, device="cuda", generator=g) + 0.5
beta = torch.randn(512, device="cuda", generator=g) * 0.1
out = torch.zeros(B * Ta + 8, 512, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    ops.conv0_fwd(audio, w, None, gamma, beta, 0, out, T0, Ta)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.conv0_fwd(audio, w, None, gamma, beta, 0, out, T0, Ta)
e1.record(); torch.cuda.synchronize()
print(f"APTAI_CONV0_MFMA={os.environ.get('APTAI_CONV0_MFMA', '1')}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call (moments + conv pass), checksum "
      f"{out.float().abs().sum().item():.6e} {out[5 * Ta + 1234].float().sum().item():.6f}")
torch.save(out[:Ta * 2].cpu(), f"/tmp/conv0_{os.environ.get('APTAI_CONV0_MFMA', '1')}.pt")
if os.path.exists("/tmp/conv0_0.pt") and os.path.exists("/tmp/conv0_1.pt"):
    a, b = torch.load("/tmp/conv0_0.pt").float(), torch.load("/tmp/conv0_1.pt").float()
    print("max |vector - mfma| =", (a - b).abs().max().item(), "of", a.abs().max().item(), "; unequal elements:", (a != b).float().mean().item())
