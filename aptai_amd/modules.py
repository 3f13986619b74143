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


# ------------------------------------------------------------------------------------- Force_APTAI building blocks
# Parameter holders with the reference's attribute / state-dict names; the arithmetic runs in force_aptai._ForceHeadsFn.
class CrossAttention(nn.Module):
    """models/modules.py:129-153 (q, k, layer_norm)."""

    def __init__(self, frame_dim, phn_dim, att_dim):
        super().__init__()
        self.q = nn.Linear(frame_dim, att_dim)
        self.k = nn.Linear(phn_dim, att_dim)
        self.layer_norm = nn.LayerNorm(att_dim * 2)


class RNN(nn.Module):
    """models/modules.py:190-214 (lstm, linear.0, linear.3)."""

    def __init__(self, hidden_dim, out_dim, drop=0.1):
        super().__init__()
        self.lstm = nn.LSTM(hidden_dim, hidden_dim, bidirectional=True, num_layers=1, batch_first=True)
        self.linear = nn.Sequential(nn.Linear(2 * hidden_dim, hidden_dim), nn.Dropout(drop), nn.Tanh(),
                                    nn.Linear(hidden_dim, out_dim))


class PositionalEncoding(nn.Module):
    """models/modules.py:217-235: sinusoidal table registered as buffer ``pe`` (max_len, 1, d_model)."""

    def __init__(self, d_model: int, dropout: float = 0.1, max_len: int = 60):
        super().__init__()
        import math
        self.dropout = nn.Dropout(p=dropout)
        # fp32 throughout, like the reference, so the buffer is bit-identical: angle[t][i] = t * 10000^(-2i/d); the table
        # interleaves sin (even columns) and cos (odd columns)
        freq = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * (-math.log(10000.0) / d_model))
        angle = torch.arange(max_len, dtype=torch.float32)[:, None] * freq[None, :]
        table = torch.stack((angle.sin(), angle.cos()), dim=-1).reshape(max_len, 1, d_model)
        self.register_buffer('pe', table.contiguous())


class ForwardSumLoss(nn.Module):
    """models/modules.py:65-117: blank log-prob -1 prepended, per-sample log-softmax over N_b+1 classes, CTC with the
    monotonic targets 1..N_b, mean over the batch — one launch of the CTC kernels for the whole batch."""

    def __init__(self, blank_logprob=-1):
        super().__init__()
        self.blank_logprob = blank_logprob
