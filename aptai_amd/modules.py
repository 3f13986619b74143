"""Head building blocks with the reference's names and state-dict keys (models/modules.py), computed by the
HIP kernels in libaptai_hip.so.  ``ConvBank`` (models/modules.py:156, unused by every model) is not built."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import hostlogic, ops
from .wav2vec2 import _Holder


class _FirFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, taps):
        B, L, C = y.shape
        ctx.taps = taps
        yc = y.float().contiguous()
        out = torch.empty_like(yc)
        ops.lowpass_fir(yc, C, L, taps, out, C, L, B, L, L, C, C)
        return out

    @staticmethod
    def backward(ctx, g):
        B, L, C = g.shape
        gc = g.float().contiguous()
        out = torch.empty_like(gc)
        ops.lowpass_fir(gc, C, L, ctx.taps, out, C, L, B, L, L, C, C)     # symmetric taps: self-adjoint
        return out, None


class LowPassFilterLayer(nn.Module):
    """models/modules.py:13-61: 51-tap Hann-windowed sinc, fp64, 'same' padding — one HIP kernel, no host hop."""

    def __init__(self, device, cutoff, sampling_rate, out_dim=9):
        super().__init__()
        self.device = device
        self.out_dim = out_dim
        taps = torch.tensor(hostlogic.lowpass_taps(cutoff, sampling_rate), device=device)
        self.N = taps.numel()
        self.filter_weights = taps.view(1, 1, -1)
        self.lowpass = _Holder()
        self.lowpass.weight = nn.Parameter(self.filter_weights.clone(), requires_grad=False)

    def taps(self) -> torch.Tensor:
        return self.lowpass.weight.detach().reshape(-1).contiguous()

    def forward(self, y):
        return _FirFn.apply(y, self.taps())
