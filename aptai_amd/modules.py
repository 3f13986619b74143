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
# Each block works on its own (the reference's `forward` signatures and return values, differentiable, fp32 HIP kernels behind
# small autograd Functions).  Force_APTAI.forward itself runs the fused chain in force_aptai._ForceHeadsFn over the same kernels.
def _seed_of(module) -> int:
    """Fresh dropout seed per call of a stand-alone block (forward and backward of one call share it)."""
    module._calls = getattr(module, "_calls", 0) + 1
    return (id(module) * 1000003 + module._calls) & 0x7FFFFFFFFFFFFFFF


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b in fp32 (aptai_sgemm_f32 on the f32 matrix instruction); x (..., K) -> (..., N)."""

    @staticmethod
    def forward(ctx, x, w, b):
        K, N = w.shape[1], w.shape[0]
        x2 = x.reshape(-1, K).float().contiguous()
        y = ops.linear_f32(x2, w.detach().float().contiguous(), None if b is None else b.detach().float().contiguous())
        ctx.save_for_backward(x2, w)
        ctx.shape, ctx.has_b = x.shape, b is not None
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        N, K = w.shape
        M = x2.shape[0]
        dy2 = dy.reshape(-1, N).float().contiguous()
        wc = w.detach().float().contiguous()
        dx = ops.sgemm(dy2, N, 1, wc, K, 1, M, K, N).view(ctx.shape)
        dw = ops.sgemm(dy2, 1, N, x2, K, 1, N, K, M)
        db = ops.colsum_f32(dy2, M, N) if ctx.has_b else None
        return dx, dw, db


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g, b, eps):
        x2 = x.reshape(-1, x.shape[-1]).float().contiguous()
        y, m, r = ops.layernorm_f32_fwd(x2, g.detach().contiguous(), b.detach().contiguous(), eps)
        ctx.save_for_backward(x2, m, r, g)
        ctx.shape = x.shape
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, m, r, g = ctx.saved_tensors
        dx, dg, db = ops.layernorm_f32_bwd(dy.reshape(x2.shape).float().contiguous(), x2, m, r, g.detach().contiguous())
        return dx.view(ctx.shape), dg, db, None


class _XattCoreFn(torch.autograd.Function):
    """energy = q k^T + mask; att = softmax(energy); out = [att k | q]  (models/modules.py:143-151)."""

    @staticmethod
    def forward(ctx, q, k, mask_i32):
        B, T, A = q.shape
        N = k.shape[1]
        qc, kc = q.float().contiguous(), k.float().contiguous()
        raw = ops.sgemm(qc, A, 1, kc, 1, A, T, N, A, batch=B, bsa=T * A, bsb=N * A, bsc=T * N)
        energy, att, att_log, _ = ops.xattn_softmax_fwd(raw, mask_i32, B, T, N)
        cat = torch.empty((B, T, 2 * A), device=q.device, dtype=torch.float32)
        ops.sgemm(att, N, 1, kc, A, 1, T, A, N, out=cat, ldc=2 * A, batch=B, bsa=T * N, bsb=N * A, bsc=T * 2 * A)
        cat[:, :, A:] = qc
        ctx.save_for_backward(qc, kc, att, att_log)
        return cat, energy.view(B, T, N)

    @staticmethod
    def backward(ctx, dcat, denergy):
        qc, kc, att, att_log = ctx.saved_tensors
        B, T, A = qc.shape
        N = kc.shape[1]
        dcat = dcat.float().contiguous()
        d_att = ops.sgemm(dcat, 2 * A, 1, kc, 1, A, T, N, A, batch=B, bsa=T * 2 * A, bsb=N * A, bsc=T * N)
        dk = ops.sgemm(att, 1, N, dcat, 2 * A, 1, N, A, T, batch=B, bsa=T * N, bsb=T * 2 * A, bsc=N * A)
        d_raw = ops.xattn_softmax_bwd(att, att_log, d_att, None)
        if denergy is not None:
            d_raw = d_raw + denergy.reshape(d_raw.shape).float()
        dq = dcat[:, :, A:].contiguous()
        ops.sgemm(d_raw, N, 1, kc, A, 1, T, A, N, out=dq, ldc=A, accumulate=True, batch=B, bsa=T * N, bsb=N * A, bsc=T * A)
        ops.sgemm(d_raw, 1, N, qc, A, 1, N, A, T, out=dk, ldc=A, accumulate=True, batch=B, bsa=T * N, bsb=T * A, bsc=N * A)
        return dq, dk.view(B, N, A), None


class CrossAttention(nn.Module):
    """models/modules.py:129-153 (q, k, layer_norm)."""

    def __init__(self, frame_dim, phn_dim, att_dim):
        super().__init__()
        self.q = nn.Linear(frame_dim, att_dim)
        self.k = nn.Linear(phn_dim, att_dim)
        self.layer_norm = nn.LayerNorm(att_dim * 2)

    def forward(self, frame_hidden, phn_hidden, labels_att_mask):
        """frame_hidden (B,T,F), phn_hidden (B,N,P), labels_att_mask (B,N) 0/1 -> (att_out (B,T,2A), energy (B,T,N))."""
        q_frame = _LinearFn.apply(frame_hidden, self.q.weight, self.q.bias)
        k_phn = _LinearFn.apply(phn_hidden, self.k.weight, self.k.bias)
        mask = (labels_att_mask != 0).to(torch.int32).contiguous()
        cat, energy = _XattCoreFn.apply(q_frame, k_phn, mask)
        att_out = _LayerNormFn.apply(cat, self.layer_norm.weight, self.layer_norm.bias, self.layer_norm.eps)
        return att_out, energy


class _LstmCoreFn(torch.autograd.Function):
    """Bidirectional LSTM recurrence over packed sequences on precomputed input projections (csrc/lstm.hip)."""

    @staticmethod
    def forward(ctx, xproj, whh0, whh1, lens_i32, B, Tp, T):
        whh = torch.stack([whh0.detach(), whh1.detach()]).float().contiguous()
        need = torch.is_grad_enabled()
        hout, gates, cst = ops.lstm_fwd(xproj.contiguous(), whh, lens_i32, B, Tp, T, save=True)
        ctx.save_for_backward(whh, lens_i32, gates, cst, hout)
        ctx.dims = (B, Tp, T)
        return hout

    @staticmethod
    def backward(ctx, dhout):
        whh, lens_i32, gates, cst, hout = ctx.saved_tensors
        B, Tp, T = ctx.dims
        M = B * Tp
        dgates = ops.lstm_bwd(dhout.float().contiguous(), whh, lens_i32, gates, cst, B, Tp, T)
        # dW_hh[dir] = sum_t dgates[t][dir]^T h_prev[t][dir]: the zero rows between utterances make one shifted product exact
        dwhh0 = ops.sgemm(dgates[1:], 1, 2048, hout, 512, 1, 1024, 256, M - 1)
        dwhh1 = ops.sgemm(dgates[:, 1024:], 1, 2048, hout[1:, 256:], 512, 1, 1024, 256, M - 1)
        # dgates is in the kernels' gate-interleaved column order, like xproj (the caller permuted W_ih's rows): the rows of the two
        # products come out in that order and go back to weight_hh's gate-major one
        inv = ops.lstm_gate_perm(dgates.device)[1][:1024]
        return dgates, dwhh0[inv], dwhh1[inv], None, None, None, None


class _TanhDropFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        y = ops.tanh_dropout_fwd(x.float().contiguous(), p, seed)
        ctx.save_for_backward(y)
        ctx.p, ctx.seed = p, seed
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return ops.tanh_dropout_bwd(y, dy.float().contiguous(), ctx.p, ctx.seed), None, None


class RNN(nn.Module):
    """models/modules.py:190-214 (lstm, linear.0, linear.3)."""

    def __init__(self, hidden_dim, out_dim, drop=0.1):
        super().__init__()
        self.lstm = nn.LSTM(hidden_dim, hidden_dim, bidirectional=True, num_layers=1, batch_first=True)
        self.linear = nn.Sequential(nn.Linear(2 * hidden_dim, hidden_dim), nn.Dropout(drop), nn.Tanh(),
                                    nn.Linear(hidden_dim, out_dim))

    def forward(self, embeddings, lens):
        """embeddings (B,T,256), lens per-utterance frame counts -> (out (B,Tmax,out_dim), hidden_tvs (B,Tmax,512)).
        Batch > 1: packed-sequence semantics, outputs cut to the longest utterance; batch 1: the LSTM runs over ALL frames
        (models/modules.py:209-212).  The shipped batch > 1 branch raises NameError (:207); its evident intent is followed."""
        B, T, D = embeddings.shape
        if D != 256 or self.lstm.hidden_size != 256:
            raise ValueError("the LSTM kernels are built for hidden size 256")
        dev = embeddings.device
        lens_l = [int(v) for v in (lens.tolist() if torch.is_tensor(lens) else lens)]
        run_lens = [T] if B == 1 else lens_l
        Tp = T + 1                                          # one zero row between utterances (see _LstmCoreFn.backward)
        xp = torch.zeros((B, Tp, D), device=dev, dtype=torch.float32)
        xp[:, :T] = embeddings
        l = self.lstm
        # rows in the LSTM kernels' gate-interleaved order (csrc/lstm.hip; autograd takes the gradients back through the gather)
        perm = ops.lstm_gate_perm(dev)[0]
        wih = torch.cat([l.weight_ih_l0, l.weight_ih_l0_reverse])[perm]
        bsum = torch.cat([l.bias_ih_l0 + l.bias_hh_l0, l.bias_ih_l0_reverse + l.bias_hh_l0_reverse])[perm]
        xproj = _LinearFn.apply(xp.view(B * Tp, D), wih, bsum)
        lens_i32 = torch.tensor(run_lens, dtype=torch.int32, device=dev)
        hout = _LstmCoreFn.apply(xproj, l.weight_hh_l0, l.weight_hh_l0_reverse, lens_i32, B, Tp, T)
        Tmax = T if B == 1 else max(lens_l)
        hidden = hout.view(B, Tp, 512)[:, :Tmax]
        h1 = _LinearFn.apply(hidden, self.linear[0].weight, self.linear[0].bias)
        h1 = _TanhDropFn.apply(h1, self.linear[1].p if self.training else 0.0, _seed_of(self))
        out = _LinearFn.apply(h1, self.linear[3].weight, self.linear[3].bias)
        return out, hidden


class _PeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pe, p, seed):
        S, B, D = x.shape
        xb = x.permute(1, 0, 2).float().contiguous().view(B * S, D)
        ids = torch.arange(B * S, dtype=torch.int32, device=x.device)
        out = ops.embed_pe_fwd(ids, xb, pe[:S].reshape(S, D).float().contiguous(), S, p, seed)
        ctx.p, ctx.seed = p, seed
        return out.view(B, S, D).permute(1, 0, 2)

    @staticmethod
    def backward(ctx, dy):
        S, B, D = dy.shape
        db = dy.permute(1, 0, 2).float().contiguous().view(B * S, D)
        dx = ops.dropout_f32(db, ctx.p, ctx.seed)
        return dx.view(B, S, D).permute(1, 0, 2), None, None, None


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

    def forward(self, x):
        """x (seq_len, batch, d_model) -> dropout(x + pe[:seq_len])."""
        return _PeFn.apply(x, self.pe, self.dropout.p if self.training else 0.0, _seed_of(self))


class _FwdSumFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, att, text_lens_i32, mel_lens_i32, blank_logprob):
        B, _, T, N = att.shape
        dev = att.device
        rows = torch.zeros((B * T, 64), device=dev, dtype=torch.float32)
        rows[:, 0] = blank_logprob
        rows[:, 1:1 + N] = att.reshape(B * T, N)
        targets = torch.arange(1, N + 1, dtype=torch.int32, device=dev)[None, :].repeat(B, 1).contiguous()
        vs = (text_lens_i32 + 1).contiguous()
        loss, nll, _, ws = ops.ctc_fwd(rows, 64, T, targets, mel_lens_i32, text_lens_i32, B, T, N + 1, blank=0, reduction="mean",
                                       zero_infinity=True, vocab_sizes_i32=vs, want_log_probs=False)
        ctx.saved = (rows, targets, mel_lens_i32, text_lens_i32, vs, ws, nll, (B, T, N))
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        rows, targets, mel, txt, vs, ws, nll, (B, T, N) = ctx.saved
        d = ops.ctc_bwd(rows, 64, T, targets, mel, txt, B, T, N + 1, ws, nll, g.float().reshape(1).contiguous(), blank=0, reduction="mean",
                        zero_infinity=True, vocab_sizes_i32=vs, ldd=64, out_dtype=torch.float32)
        return d[:, 1:1 + N].reshape(B, 1, T, N), None, None, None


class ForwardSumLoss(nn.Module):
    """models/modules.py:65-117: blank log-prob -1 prepended, per-sample log-softmax over N_b+1 classes, CTC with the
    monotonic targets 1..N_b, mean over the batch — one launch of the CTC kernels for the whole batch."""

    def __init__(self, blank_logprob=-1):
        super().__init__()
        self.blank_logprob = blank_logprob

    def forward(self, attn_logprob, text_lens, mel_lens):
        """attn_logprob (B,1,T,N) log-attention, text_lens / mel_lens per-utterance phoneme / frame counts -> scalar loss."""
        dev = attn_logprob.device
        if attn_logprob.shape[-1] > 63:
            raise ValueError("ForwardSumLoss kernels hold at most 63 phoneme slots")
        tl = torch.as_tensor(text_lens, dtype=torch.int32).to(dev).contiguous()
        ml = torch.as_tensor(mel_lens, dtype=torch.int32).to(dev).contiguous()
        return _FwdSumFn.apply(attn_logprob.float(), tl, ml, float(self.blank_logprob))
