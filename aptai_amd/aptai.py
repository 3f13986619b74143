"""``APTAI`` — drop-in for the reference's models/aptai.py:14-182 on MI355X.

Same constructor, ``forward(epoch, **batch)`` signature, returned dict keys, inference helper and state-dict
keys (``wav2vec2.*``, ``tv_head.2.*``, ``phn_head.2.*``, ``tv_lowpass.lowpass.weight``).  Differences, all
parameterised with the reference's value as default: the head width follows ``pretrain_cfg.hidden_size`` (the
reference hard-codes 1024, models/aptai.py:46,54) and the tapped hidden state is the last one (the reference
hard-codes ``hidden_states[24]``, :81 — identical for its 24-layer backbone); ``n_tv`` (9) and ``n_phn`` (46)
are arguments so BASELINE.json's 12-track variant can be benchmarked.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .config import W2V2Config
from .hostlogic import TV_NAMES
from .modules import LowPassFilterLayer
from .wav2vec2 import Wav2Vec2Model, _seed

_PADN = 64      # head Linears run on the MFMA GEMM with N padded to 64


def heads_fwd(h, tv_w, tv_b, ph_w, ph_b, st):
    """tanh/leaky-relu -> Linear(H, n) x2 -> FIR -> masked MSE + masked CE (+ argmax), models/aptai.py:83-106.
    Returns ((loss, mse, ce, tvs, pred, logits), saved)."""
    g, M, H = st.g, st.g.M, h.shape[1]
    dev = h.device
    n_tv, n_phn = tv_w.shape[0], ph_w.shape[0]
    a_tv, a_ph = ops.head_act_fwd(h, st.p_tv, st.p_ph, st.seed)

    def pad_w(w, b):
        wp = torch.zeros((_PADN, H), device=dev, dtype=torch.bfloat16)
        ops.cast_bf16(w.detach(), wp[:w.shape[0]])
        bp = torch.zeros(_PADN, device=dev, dtype=torch.float32)
        bp[:b.shape[0]] = b.detach()
        return wp, bp
    wtv, btv = pad_w(tv_w, tv_b)
    wph, bph = pad_w(ph_w, ph_b)
    tv_raw = ops.gemm(a_tv, wtv, M, _PADN, H, bias=btv, out_f32=True)
    logits = ops.gemm(a_ph, wph, M, _PADN, H, bias=bph, out_f32=True)
    tvs = torch.empty((g.B, g.T, n_tv), device=dev, dtype=torch.float32)
    ops.lowpass_fir(tv_raw, _PADN, g.Tp, st.taps, tvs, n_tv, g.T, g.B, g.T, g.T, n_tv, n_tv)
    scalars, pred = ops.aptai_loss_fwd(tvs, st.tv_tgt, logits, _PADN, g.Tp, st.phn_tgt, g.B, g.T, n_tv, n_phn, st.w_mse, st.w_ce)
    saved = SimpleNamespace(h=h, a_tv=a_tv, a_ph=a_ph, wtv=wtv, wph=wph, tvs=tvs, logits=logits, scalars=scalars, n_tv=n_tv,
                            n_phn=n_phn)
    return (scalars[0], scalars[1], scalars[2], tvs, pred, logits), saved


def heads_bwd(s, st, gl):
    """Returns (dh, dW_tv, db_tv, dW_ph, db_ph); ``gl`` = device scalar gradient of the loss (or None = 1)."""
    g, M, H = st.g, st.g.M, s.h.shape[1]
    # data parallel: normalise by the GLOBAL valid counts / world (dp.GlobalLossNorm) instead of this shard's own counts
    norm = getattr(st, "norm_scalars", None)
    d_tvs, d_logits = ops.aptai_loss_bwd(s.tvs, st.tv_tgt, s.logits, _PADN, g.Tp, st.phn_tgt, g.B, g.T, s.n_tv, s.n_phn, st.w_mse,
                                         st.w_ce, norm() if callable(norm) else (norm if norm is not None else s.scalars), gl,
                                         ldd=_PADN)
    d_tvraw = torch.empty((M, _PADN), device=s.h.device, dtype=torch.bfloat16)
    ops.lowpass_fir(d_tvs, s.n_tv, g.T, st.taps, d_tvraw, _PADN, g.Tp, g.B, g.T, g.Tp, s.n_tv, _PADN)
    da_tv = ops.gemm(d_tvraw, s.wtv, M, H, _PADN, b_kmajor=True)
    da_ph = ops.gemm(d_logits, s.wph, M, H, _PADN, b_kmajor=True)
    sk = max(1, min(16, M // 1024))
    dwtv = ops.gemm(d_tvraw, s.a_tv, _PADN, H, M, a_kmajor=True, b_kmajor=True, out_f32=True, split_k=sk)
    dwph = ops.gemm(d_logits, s.a_ph, _PADN, H, M, a_kmajor=True, b_kmajor=True, out_f32=True, split_k=sk)
    dbtv = ops.colsum(d_tvraw, M, _PADN)
    dbph = ops.colsum(d_logits, M, _PADN)
    dh = ops.head_act_bwd(s.h, da_tv, da_ph, st.p_tv, st.p_ph, st.seed)
    return dh, dwtv[:s.n_tv], dbtv[:s.n_tv], dwph[:s.n_phn], dbph[:s.n_phn]


class _HeadsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, tv_w, tv_b, ph_w, ph_b, st):
        (loss, mse, ce, tvs, pred, logits), ctx.saved = heads_fwd(h, tv_w, tv_b, ph_w, ph_b, st)
        ctx.st = st
        loss, mse, ce = loss.clone(), mse.clone(), ce.clone()
        ctx.mark_non_differentiable(mse, ce, tvs, pred, logits)
        return loss, mse, ce, tvs, pred, logits

    @staticmethod
    def backward(ctx, gloss, *_):
        out = heads_bwd(ctx.saved, ctx.st, gloss.float().reshape(1).contiguous())
        ctx.saved = None
        return out + (None,)


class APTAI(nn.Module):
    def __init__(self, device, vocab, huggingface_model_id, pretrain_cfg, cache_dir, phn_drop=0.1, tv_drop=0.1,
                 freeze_feature_encoder=True, n_tv=9, n_phn=46):
        super().__init__()
        self.device = device
        self.vocab = vocab
        self.huggingface_model_id = huggingface_model_id
        self.pretrain_cfg = pretrain_cfg
        self.n_tv, self.n_phn = n_tv, n_phn
        if n_tv > _PADN or n_phn > _PADN:
            raise ValueError("head widths above 64 are not built")

        self.wav2vec2 = Wav2Vec2Model.from_pretrained(huggingface_model_id, config=pretrain_cfg, cache_dir=cache_dir).to(device)
        self.wav2vec2.gradient_checkpointing_enable()
        if freeze_feature_encoder:
            self.wav2vec2.freeze_feature_encoder()
        H = self.wav2vec2.config.hidden_size

        self.dp_loss_norm = None       # aptai_amd.dp.GlobalLossNorm under data parallelism: masked means over the global batch
        self.tv_head = nn.Sequential(nn.Dropout(tv_drop), nn.Tanh(), nn.Linear(H, n_tv))
        self.tv_lowpass = LowPassFilterLayer(self.device, 10, 49, n_tv)
        self.phn_head = nn.Sequential(nn.Dropout(phn_drop), nn.LeakyReLU(), nn.Linear(H, n_phn))

    def _heads(self, w2v2_out, tv_targets, phn_targets):
        g = w2v2_out._geom
        h = w2v2_out._flat_last                       # hidden_states[-1] == hidden_states[24] for the large backbone
        dev = h.device
        if tv_targets is None:
            tv_targets = torch.full((g.B, g.T, self.n_tv), -100.0, device=dev)
            phn_targets = torch.zeros((g.B, g.T), device=dev, dtype=torch.long)
        tr = self.training
        st = SimpleNamespace(g=g, p_tv=self.tv_head[0].p if tr else 0.0, p_ph=self.phn_head[0].p if tr else 0.0,
                             seed=_seed(self.wav2vec2.base_seed, self.wav2vec2._step, 999), taps=self.tv_lowpass.taps(),
                             tv_tgt=tv_targets.contiguous(), phn_tgt=phn_targets.contiguous(), w_mse=0.5, w_ce=0.5)
        if self.dp_loss_norm is not None and tr:
            self.dp_loss_norm.begin(st.tv_tgt, st.phn_tgt)       # 2-scalar all-reduce under the head GEMMs
            st.norm_scalars = self.dp_loss_norm.scalars           # resolved (waited for) in the backward
        return _HeadsFn.apply(h, self.tv_head[2].weight, self.tv_head[2].bias, self.phn_head[2].weight,
                              self.phn_head[2].bias, st)

    def forward(self, epoch, audio_inputs, audio_lengths, phn_frames_49hz, LA=None, LP=None, JA=None, TTCL=None,
                TTCD=None, TMCL=None, TMCD=None, TBCL=None, TBCD=None, **extra_tvs):
        tracks = [LA, LP, JA, TTCL, TTCD, TMCL, TMCD, TBCL, TBCD]
        tracks = [t for t in tracks if t is not None] + [extra_tvs[k] for k in sorted(extra_tvs)]
        if len(tracks) != self.n_tv:
            raise ValueError(f"expected {self.n_tv} TV tracks, got {len(tracks)}")
        tv_targets = torch.stack(tracks, dim=-1).float()                       # models/aptai.py:67-70
        w2v2_out = self.wav2vec2(audio_inputs, attention_mask=audio_lengths[:, None], return_dict=True,
                                 output_hidden_states=True)
        loss, mse, ce, tvs, pred, _ = self._heads(w2v2_out, tv_targets, phn_frames_49hz)
        return {'loss': loss, 'mse_loss': mse, 'ce_loss': ce, 'tvs_pred': tvs, 'phn_fc_pred': pred}

    def get_config(self):
        return {'device': self.device, 'vocab': self.vocab, 'huggingface_model_id': self.huggingface_model_id,
                'pretrain_cfg': self.pretrain_cfg}

    def get_aptai_output(self, wav):
        """models/aptai.py:125-179."""
        self.eval()
        device = next(self.parameters()).device
        with torch.no_grad():
            if type(wav) is torch.Tensor:
                wav = wav[0]
            wav_input = torch.unsqueeze(torch.Tensor(wav), dim=0).to(device)
            wav_len = torch.unsqueeze(torch.LongTensor([len(wav)]), dim=0).to(device)
            w2v2_out = self.wav2vec2(wav_input, attention_mask=wav_len[:, None], return_dict=True, output_hidden_states=True)
            _, _, _, tvs, pred, logits = self._heads(w2v2_out, None, None)
            g = w2v2_out._geom
            phn_logits = logits.view(g.B, g.Tp, -1)[:, :g.T, :self.n_phn]
            phn_probs = F.softmax(phn_logits, dim=-1)
            tvs_out = tvs.squeeze(dim=0).cpu().numpy()
            names = TV_NAMES if self.n_tv == 9 else tuple(f"TV{i}" for i in range(self.n_tv))
            tvs_pred_dict = {n: [row[i] for row in tvs_out] for i, n in enumerate(names)}
            return {
                'phn_fc_probs': phn_probs.permute(2, 1, 0).squeeze(dim=0).cpu().numpy(),      # the reference's `.T` on (1, T, C): (C, T, 1)
                'phn_fc_logits': phn_logits.squeeze(dim=0).cpu().numpy(),
                'phn_fc_pred': pred.squeeze(dim=0).cpu().numpy(),
                'tvs_pred': tvs_pred_dict,
            }
