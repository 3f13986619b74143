"""MI355X-native ``Wav2Vec2Model``: same module surface and state-dict keys as the HuggingFace class the
reference instantiates (models/aptai.py:33-40, models/w2v2_pr.py:28-33), computed by hand-written HIP
kernels (libaptai_hip.so) through ``aptai_amd.ops``.  No ``transformers`` import, no CPU fallback.

Layout: activations are channels-last bf16 ``[B*Tp][C]`` with ``Tp`` = frames per utterance padded to a
multiple of 128 (>= T+2); frames in [T, Tp) are treated exactly like the reference's padded frames (zeroed
before the encoder, masked as attention keys, never touched by a loss), so they cannot influence valid
frames.  Parameters stay fp32 ``nn.Parameter``s (drop-in for ``torch.optim.Adam`` and ``state_dict``); bf16
compute copies are cached per parameter version.  "HF:n" cites modeling_wav2vec2.py (transformers 5.15.0).
"""
from __future__ import annotations

import json
import math
import os
from types import SimpleNamespace
from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import hostlogic, ops
from .config import W2V2Config


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def _rows_with_zero_slack(rows: int, C: int, dev, slack: int = 8) -> torch.Tensor:
    """[rows][C] bf16 whose last `slack` rows are zero and whose other rows are about to be written whole by a kernel: the
    overlapping-row GEMM of the next conv layer (and the dgrad accumulation of the previous one) reads a few rows past the last
    utterance, and 0 x NaN must not reach a weight gradient.  A full torch.zeros of these buffers (up to 537 MB each, ~20 per
    Wav2Vec2_PR step) cost 0.39 ms per step (rocprofv3, profiles/r03_pr_kernel_stats.csv)."""
    t = torch.empty((rows, C), device=dev, dtype=torch.bfloat16)
    t[rows - slack:].zero_()
    return t


class _Holder(nn.Module):
    """Parameter container (the arithmetic lives in the HIP kernels, not in module forwards)."""


def _linear(out_f: int, in_f: int, std: float = 0.02) -> _Holder:
    m = _Holder()
    m.weight = nn.Parameter(torch.randn(out_f, in_f) * std)
    m.bias = nn.Parameter(torch.zeros(out_f))
    return m


def _norm(c: int) -> _Holder:
    m = _Holder()
    m.weight = nn.Parameter(torch.ones(c))
    m.bias = nn.Parameter(torch.zeros(c))
    return m


class Wav2Vec2BaseModelOutput(dict):
    """Minimal stand-in for HF's ModelOutput: attribute, key and integer access (``outputs[0]``)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __getitem__(self, k):
        if isinstance(k, int):
            return [v for v in self.values() if v is not None][k]
        return dict.__getitem__(self, k)


# =================================================================================== generic autograd glue
class _OpFn(torch.autograd.Function):
    """forward(impl, x, *params): impl.fwd(x, params) -> (outputs tuple, saved); backward -> impl.bwd(saved, grads)."""

    @staticmethod
    def forward(ctx, impl, need, x, *params):
        outs, saved = impl.fwd(x, params, need)          # grad mode is off in here: `need` is decided by the caller
        ctx.impl, ctx.saved = impl, saved
        return outs if len(outs) > 1 else outs[0]

    @staticmethod
    def backward(ctx, *grads):
        dx, pgrads = ctx.impl.bwd(ctx.saved, grads, ctx.needs_input_grad[2])
        ctx.saved = None
        return (None, None, dx) + tuple(pgrads)


def _run(impl, x, *params):
    need = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params if p is not None))
    return _OpFn.apply(impl, need, x, *params)


class _ConvStackFn(torch.autograd.Function):
    """Trainable feature encoder (Wav2Vec2_PR fine-tuning, train/train_phoneme_recognizer.py): forward saves the layer
    outputs / pre-activations, backward = Wav2Vec2Model._conv_backward."""

    @staticmethod
    def forward(ctx, audio, model, g, *params):
        feats, sv = model._conv_forward(audio, g, save=True)
        ctx.model, ctx.g, ctx.sv = model, g, sv
        return feats

    @staticmethod
    def backward(ctx, dfeats):
        grads = ctx.model._conv_backward(ctx.sv, ctx.g, dfeats.contiguous())
        ctx.sv = None
        return (None, None, None) + tuple(grads)


def _seed(base: int, *ids: int) -> int:
    s = base & 0xFFFFFFFFFFFF
    for i in ids:
        s = (s * 1000003 + i + 1) & 0xFFFFFFFFFFFFFFFF
    # splitmix64 finaliser: the kernels use the two 32-bit halves as independent stream keys (csrc/common.h rng_hash)
    s = ((s ^ (s >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    s = ((s ^ (s >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return s ^ (s >> 31)


# =================================================================================== encoder layer
class _LayerImpl:
    """One transformer layer (HF:575-654).  post-LN (base) or pre-LN (do_stable_layer_norm, large)."""

    def __init__(self, cfg: W2V2Config, geom, lens_i32, w, training: bool, seed: int):
        self.cfg, self.g, self.lens, self.w, self.training, self.seed = cfg, geom, lens_i32, w, training, seed

    # params: ln1.w ln1.b ln2.w ln2.b, then the 12 fp32 linear parameters (q,k,v weights; q,k,v biases; out w,b;
    # ffn1 w,b; ffn2 w,b).  The forward reads their cached bf16 copies in `w`; the backward returns their gradients.
    def fwd(self, x, params, need):
        cfg, g, w = self.cfg, self.g, self.w
        M, H, I = g.M, cfg.hidden_size, cfg.intermediate_size
        ln1w, ln1b, ln2w, ln2b = params[0], params[1], params[2], params[3]
        tr = self.training
        p_h = cfg.hidden_dropout if tr else 0.0
        p_a = cfg.activation_dropout if tr else 0.0
        p_att = cfg.attention_dropout if tr else 0.0
        s = SimpleNamespace(x=x)
        pre = cfg.do_stable_layer_norm
        mx = getattr(self, "mx", None)
        if mx is not None and not need:
            return self._fwd_mxfp8(x, params, mx)
        if pre:
            n1, s.m1, s.r1 = ops.layernorm_fwd(x, ln1w, ln1b, cfg.layer_norm_eps, save_stats=need)
            attn_in = n1
            s.n1 = n1
        else:
            attn_in = x
        # the projection's epilogue hands Q over as q * head_dim^-0.5 * log2(e) (rounded once): the attention kernels then see
        # their exp2 arguments directly (q_prescaled)
        s.qkv = ops.gemm(attn_in, w.wqkv, M, 3 * H, H, bias=w.bqkv, colscale=(H, ops.attention_qscale(H, cfg.num_attention_heads)))
        s.ctx, s.lse = ops.attention_fwd(s.qkv, self.lens, g.B, g.Tp, H, cfg.num_attention_heads, q_prescaled=True, dropout_p=p_att,
                                         seed=_seed(self.seed, 1), save_lse=need)
        s.s1 = ops.gemm(s.ctx, w.wo, M, H, H, bias=w.bo, residual=x, dropout_p=p_h, seed=_seed(self.seed, 2))
        if pre:
            ffn_in, s.m2, s.r2 = ops.layernorm_fwd(s.s1, ln2w, ln2b, cfg.layer_norm_eps, save_stats=need)
            s.n2 = ffn_in
        else:
            ffn_in, s.m1, s.r1 = ops.layernorm_fwd(s.s1, ln1w, ln1b, cfg.layer_norm_eps, save_stats=need)
            s.x1 = ffn_in
        s.u = torch.empty((M, I), device=x.device, dtype=torch.bfloat16) if need else None
        # with gradients: s.u holds dropmask/(1-p) * gelu'(pre-activation), the factor the dgrad epilogue multiplies by
        s.hact = ops.gemm(ffn_in, w.w1, M, I, H, bias=w.b1, gelu=True, out_pre=s.u, pre_dgelu=need and _PRE_DGELU, dropout_p=p_a,
                          seed=_seed(self.seed, 3))
        res2 = s.s1 if pre else ffn_in
        s.s2 = ops.gemm(s.hact, w.w2, M, H, I, bias=w.b2, residual=res2, dropout_p=p_h, seed=_seed(self.seed, 4))
        if pre:
            y = s.s2
        else:
            y, s.m2, s.r2 = ops.layernorm_fwd(s.s2, ln2w, ln2b, cfg.layer_norm_eps, save_stats=need)
        s.p = (p_h, p_a, p_att)
        s.ln = (ln1w, ln1b, ln2w, ln2b)
        return (y,), (s if need else None)

    def fwd_f32res(self, x, x32, params):
        """Inference only, opt-in (set_encoder_precision("bf16_f32res")): the residual stream stays in fp32 - out-proj and FFN2
        write fp32 and add the fp32 residual in their epilogues, LayerNorm reads fp32 and hands the next GEMM a bf16 copy and the
        next residual add an fp32 one.  GEMM operands and attention stay bf16.  Returns (y_bf16 | None, y_f32)."""
        cfg, g, w = self.cfg, self.g, self.w
        M, H, I = g.M, cfg.hidden_size, cfg.intermediate_size
        ln1w, ln1b, ln2w, ln2b = params[:4]
        heads = cfg.num_attention_heads
        qs = (H, ops.attention_qscale(H, heads))
        if cfg.do_stable_layer_norm:                       # pre-LN (large): x32 is the un-normalised stream
            n1, _ = ops.layernorm_fwd_f32in(x32, ln1w, ln1b, cfg.layer_norm_eps, want_f32=False)
            qkv = ops.gemm(n1, w.wqkv, M, 3 * H, H, bias=w.bqkv, colscale=qs)
            ctx, _ = ops.attention_fwd(qkv, self.lens, g.B, g.Tp, H, heads, save_lse=False, q_prescaled=True)
            s1 = ops.gemm(ctx, w.wo, M, H, H, bias=w.bo, out_f32=True, residual_f32=x32, tile=128)
            n2, _ = ops.layernorm_fwd_f32in(s1, ln2w, ln2b, cfg.layer_norm_eps, want_f32=False)
            u = ops.gemm(n2, w.w1, M, I, H, bias=w.b1, gelu=True)
            return None, ops.gemm(u, w.w2, M, H, I, bias=w.b2, out_f32=True, residual_f32=s1, tile=128)
        qkv = ops.gemm(x, w.wqkv, M, 3 * H, H, bias=w.bqkv, colscale=qs)
        ctx, _ = ops.attention_fwd(qkv, self.lens, g.B, g.Tp, H, heads, save_lse=False, q_prescaled=True)
        s1 = ops.gemm(ctx, w.wo, M, H, H, bias=w.bo, out_f32=True, residual_f32=x32, tile=128)
        n1, n1_32 = ops.layernorm_fwd_f32in(s1, ln1w, ln1b, cfg.layer_norm_eps)
        u = ops.gemm(n1, w.w1, M, I, H, bias=w.b1, gelu=True)
        s2 = ops.gemm(u, w.w2, M, H, I, bias=w.b2, out_f32=True, residual_f32=n1_32, tile=128)
        return ops.layernorm_fwd_f32in(s2, ln2w, ln2b, cfg.layer_norm_eps)

    def _fwd_mxfp8(self, x, params, mx):
        """Inference-only forward with MX block-scaled FP8 operands in the Linear layers where they pay (BASELINE configs[4];
        csrc/mxgemm.hip): q|k|v, FFN1 and FFN2.  Their inputs are quantised on the fly (E4M3 elements, E8M0 scale per 32 k; FFN1 hands its
        GELU output to FFN2 as MXFP8 straight from the GEMM epilogue), the frozen weights were quantised once.  The attention output
        projection stays bf16: at [rows] x 768 x 768 the quantiser's launch costs more than the fp8 product saves (tools/mx_bench.py).
        LayerNorm, attention, biases, residual stream and accumulation are as in the bf16 path."""
        cfg, g, w = self.cfg, self.g, self.w
        M, H, I = g.M, cfg.hidden_size, cfg.intermediate_size
        ln1w, ln1b, ln2w, ln2b = params[0], params[1], params[2], params[3]
        pre = cfg.do_stable_layer_norm
        if pre:             # the normalised rows feed the q|k|v GEMM only: they leave the LayerNorm kernel as MXFP8
            _, aq, a_s = ops.layernorm_fwd_mx(x, ln1w, ln1b, cfg.layer_norm_eps)
        else:
            aq, a_s = ops.mx_quantize(x)
        qkv = ops.gemm_mxfp8(aq, a_s, mx.wqkv[0], mx.wqkv[1], M, 3 * H, H, bias=w.bqkv)
        ctx, _ = ops.attention_fwd(qkv, self.lens, g.B, g.Tp, H, cfg.num_attention_heads, save_lse=False)
        s1 = ops.gemm(ctx, w.wo, M, H, H, bias=w.bo, residual=x)
        if pre:             # FFN1's input only
            ffn_in, fq, f_s = ops.layernorm_fwd_mx(s1, ln2w, ln2b, cfg.layer_norm_eps)
        else:               # post-LN: also the residual of FFN2
            ffn_in, fq, f_s = ops.layernorm_fwd_mx(s1, ln1w, ln1b, cfg.layer_norm_eps, want_bf16=True)
        hq, h_s = ops.gemm_mxfp8_mxout(fq, f_s, mx.w1[0], mx.w1[1], M, I, H, bias=w.b1, gelu=True)
        s2 = ops.gemm_mxfp8(hq, h_s, mx.w2[0], mx.w2[1], M, H, I, bias=w.b2, residual=s1 if pre else ffn_in)
        y = s2 if pre else ops.layernorm_fwd(s2, ln2w, ln2b, cfg.layer_norm_eps, save_stats=False)[0]
        return (y,), None

    def bwd(self, s, grads, x_needs, defer_wgrad=False):
        """dgrad chain on the caller's stream; the weight/bias gradients are independent of that chain and can go to a side
        stream (APTAI_SIDE_STREAM=1).  Off by default: each GEMM already fills the 256 CUs, the A/B was neutral.
        defer_wgrad (graph runner): return (dx, (LayerNorm grads, None), pending) WITHOUT the layer's grouped weight-gradient
        launch; `bwd_wgrad(pending)` issues it - the runner replays it on a side stream beside the NEXT layer's backward."""
        cfg, g, w = self.cfg, self.g, self.w
        M, H, I = g.M, cfg.hidden_size, cfg.intermediate_size
        heads = cfg.num_attention_heads
        p_h, p_a, p_att = s.p
        ln1w, ln1b, ln2w, ln2b = s.ln
        dy = grads[0].contiguous()
        pre = cfg.do_stable_layer_norm
        sk = w.split_k
        main = torch.cuda.current_stream()
        side = _side_stream(dy.device)

        def on_side(fn):
            if not _USE_SIDE_STREAM:
                return fn()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                return fn()
        # ---- FFN block
        if pre:
            ds2 = dy                                             # y = s1 + D(ffn)
            d_ffn_out = ops.dropout(ds2, p_h, _seed(self.seed, 4)) if p_h > 0 else ds2
            ffn_in = s.n2
        else:
            ds2, d_ffn_out, dg2, db2 = ops.layernorm_bwd(dy, s.s2, s.m2, s.r2, ln2w, dropout_p=p_h, seed=_seed(self.seed, 4))
            if d_ffn_out is None:
                d_ffn_out = ds2
            ffn_in = s.x1
        grouped = _GROUPED_WGRAD
        if not grouped:
            dw2, dbias2 = on_side(lambda: (ops.gemm(d_ffn_out, s.hact, H, I, M, a_kmajor=True, b_kmajor=True, out_f32=True, split_k=sk[3]),
                                           ops.colsum(d_ffn_out, M, H)))
        if _PRE_DGELU:
            du = ops.gemm(d_ffn_out, w.w2, M, I, H, b_kmajor=True, mul_aux=s.u)
        else:                                        # APTAI_PRE_DGELU=0: recompute mask and gelu' in the dgrad epilogue (A/B)
            du = ops.gemm(d_ffn_out, w.w2, M, I, H, b_kmajor=True, dgelu_aux=s.u, dropout_p=p_a, seed=_seed(self.seed, 3))
        if not grouped:
            dw1, dbias1 = on_side(lambda: (ops.gemm(du, ffn_in, I, H, M, a_kmajor=True, b_kmajor=True, out_f32=True, split_k=sk[2]),
                                           ops.colsum(du, M, I)))
        if pre:
            dn2 = ops.gemm(du, w.w1, M, H, I, b_kmajor=True)
            ds1, d_att_out, dg2, db2 = ops.layernorm_bwd(dn2, s.s1, s.m2, s.r2, ln2w, dres=ds2, dropout_p=p_h,
                                                         seed=_seed(self.seed, 2))
        else:
            dx1 = ops.gemm(du, w.w1, M, H, I, b_kmajor=True, residual=ds2)
            ds1, d_att_out, dg1, db1 = ops.layernorm_bwd(dx1, s.s1, s.m1, s.r1, ln1w, dropout_p=p_h, seed=_seed(self.seed, 2))
        if d_att_out is None:
            d_att_out = ds1
        # ---- attention block
        if not grouped:
            dwo, dbo = on_side(lambda: (ops.gemm(d_att_out, s.ctx, H, H, M, a_kmajor=True, b_kmajor=True, out_f32=True, split_k=sk[1]),
                                        ops.colsum(d_att_out, M, H)))
        dctx = ops.gemm(d_att_out, w.wo, M, H, H, b_kmajor=True)
        dqkv = ops.attention_bwd(s.qkv, self.lens, s.ctx, dctx, s.lse, g.B, g.Tp, H, heads, q_prescaled=True, dropout_p=p_att,
                                 seed=_seed(self.seed, 1), dctx_zero_beyond_len=True)
        attn_in = s.n1 if pre else s.x
        if not grouped:
            dwqkv, dbqkv = on_side(lambda: (ops.gemm(dqkv, attn_in, 3 * H, H, M, a_kmajor=True, b_kmajor=True, out_f32=True, split_k=sk[0]),
                                            ops.colsum(dqkv, M, 3 * H)))
        if pre:
            dn1 = ops.gemm(dqkv, w.wqkv, M, H, 3 * H, b_kmajor=True)
            dx, _, dg1, db1 = ops.layernorm_bwd(dn1, s.x, s.m1, s.r1, ln1w, dres=ds1)
        else:
            dx = ops.gemm(dqkv, w.wqkv, M, H, 3 * H, b_kmajor=True, residual=ds1)
        if grouped and defer_wgrad:
            pending = SimpleNamespace(d_ffn_out=d_ffn_out, hact=s.hact, du=du, ffn_in=ffn_in, d_att_out=d_att_out, ctx=s.ctx, dqkv=dqkv,
                                      attn_in=attn_in, M=M, H=H, I=I)
            return dx, (dg1, db1, dg2, db2), pending
        if grouped:
            # all weight and bias gradients of the layer in ONE launch (aptai_gemm_bf16_grouped): 4 x (dY^T X) + 4 x (1^T dY),
            # 432 + 54 full-K tiles for wav2vec2-base = one round of the 512 block slots, no split-K slabs, no reduce kernels
            tn = dict(a_kmajor=True, b_kmajor=True, out_f32=True)
            ones = ops.ones_kmajor(M, dy.device)
            dw2, dw1, dwo, dwqkv, r2, r1, ro, rq = ops.gemm_grouped([
                (d_ffn_out, s.hact, H, I, M, tn), (du, ffn_in, I, H, M, tn), (d_att_out, s.ctx, H, H, M, tn),
                (dqkv, attn_in, 3 * H, H, M, tn),
                (ones, d_ffn_out, 8, H, M, tn), (ones, du, 8, I, M, tn), (ones, d_att_out, 8, H, M, tn),
                (ones, dqkv, 8, 3 * H, M, tn)])
            dbias2, dbias1, dbo, dbqkv = r2[0], r1[0], ro[0], rq[0]
        main.wait_stream(side)
        return dx, (dg1, db1, dg2, db2, dwqkv[0:H], dwqkv[H:2 * H], dwqkv[2 * H:3 * H], dbqkv[0:H], dbqkv[H:2 * H],
                    dbqkv[2 * H:3 * H], dwo, dbo, dw1, dbias1, dw2, dbias2)


    @staticmethod
    def bwd_wgrad(p):
        """The grouped weight / bias gradient launch of one layer from the tensors its dgrad chain left behind (see bwd).  Returns
        the 12 Linear-parameter gradients in the order of bwd's tuple (after the four LayerNorm gradients)."""
        M, H, I = p.M, p.H, p.I
        tn = dict(a_kmajor=True, b_kmajor=True, out_f32=True)
        ones = ops.ones_kmajor(M, p.du.device)
        dw2, dw1, dwo, dwqkv, r2, r1, ro, rq = ops.gemm_grouped([
            (p.d_ffn_out, p.hact, H, I, M, tn), (p.du, p.ffn_in, I, H, M, tn), (p.d_att_out, p.ctx, H, H, M, tn),
            (p.dqkv, p.attn_in, 3 * H, H, M, tn),
            (ones, p.d_ffn_out, 8, H, M, tn), (ones, p.du, 8, I, M, tn), (ones, p.d_att_out, 8, H, M, tn),
            (ones, p.dqkv, 8, 3 * H, M, tn)])
        dbias2, dbias1, dbo, dbqkv = r2[0], r1[0], ro[0], rq[0]
        return (dwqkv[0:H], dwqkv[H:2 * H], dwqkv[2 * H:3 * H], dbqkv[0:H], dbqkv[H:2 * H], dbqkv[2 * H:3 * H], dwo, dbo, dw1, dbias1,
                dw2, dbias2)


_SIDE_STREAMS = {}
_GROUPED_WGRAD = os.environ.get("APTAI_GROUPED_WGRAD", "1") != "0"
_PRE_DGELU = os.environ.get("APTAI_PRE_DGELU", "1") != "0"
_SPEC_ON_DEVICE = os.environ.get("APTAI_SPEC_ON_DEVICE", "1") != "0"      # =0: numpy sampler with HF's RNG order (host round trip)
_USE_SIDE_STREAM = os.environ.get("APTAI_SIDE_STREAM", "0") != "0"     # measured neutral on MI355X (A/B 15.95 vs 15.98 ms/step)


def _side_stream(device):
    st = _SIDE_STREAMS.get(device)
    if st is None:
        st = torch.cuda.Stream(device=device)
        _SIDE_STREAMS[device] = st
    return st


# =================================================================================== front end of the encoder
class _FrontImpl:
    """Feature projection (HF:422-434) -> SpecAugment fill + padded-frame zeroing (HF:1292-1295, 678-681) ->
    h + GELU(grouped positional conv(h)) (HF:326-379, 689-692) -> [LayerNorm for post-LN models] -> dropout."""

    def __init__(self, cfg, geom, lens_i32, spec, training, seed, model):
        self.cfg, self.g, self.lens, self.spec, self.training, self.seed, self.model = cfg, geom, lens_i32, spec, training, seed, model

    def _packed(self, key, dev):
        cfg, g = self.cfg, self.g
        Cg = cfg.hidden_size // cfg.num_conv_pos_embedding_groups
        pad = cfg.num_conv_pos_embeddings // 2
        rows_p = g.Tp + 2 * pad
        n = cfg.num_conv_pos_embedding_groups * g.B * rows_p * Cg + Cg * 8
        buf = self.model._scratch(key, n, dev)
        return buf, rows_p, Cg, pad

    def _posconv_batch(self, Cg, rows_p, K, a_off=0):
        cfg, g = self.cfg, self.g
        G = cfg.num_conv_pos_embedding_groups
        H = cfg.hidden_size
        return dict(outer=g.B, inner=G, a=(rows_p * Cg, g.B * rows_p * Cg), b=(0, Cg * K), c=(g.Tp * H, Cg),
                    bias=(0, Cg), res=(g.Tp * H, Cg), aux=(g.Tp * H, Cg))

    def fwd(self, feats, params, need):
        cfg, g = self.cfg, self.g
        (fplw, fplb, pw, pb, embed, pcg, pcv, pcb, elw, elb) = params
        M, H, C = g.M, cfg.hidden_size, feats.shape[1]
        tr = self.training
        s = SimpleNamespace(feats=feats, params=params)
        s.n0, s.m0, s.r0 = ops.layernorm_fwd(feats, fplw, fplb, cfg.layer_norm_eps, save_stats=need)
        wp = self.model._cached(("proj",), [pw], lambda: ops.cast_bf16(pw))
        s.p_fp = cfg.feat_proj_dropout if tr else 0.0
        h0 = ops.gemm(s.n0, wp, M, H, C, bias=pb, dropout_p=s.p_fp, seed=_seed(self.seed, 11))
        ops.frame_mask_fwd(h0, self.lens, self.spec, embed, g.B, g.Tp, g.T, H)
        s.h0 = h0
        # positional conv as ONE batched implicit GEMM over (utterance, group) on a zero-gapped group-major copy
        G, Kw = cfg.num_conv_pos_embedding_groups, cfg.num_conv_pos_embeddings
        xg, rows_p, Cg, pad = self._packed("pc_x", feats.device)
        ops.posconv_pack(h0, xg, g.B, g.Tp, H, G, pad)
        s.wf, s.wd, s.norm = self.model._cached(("posconv",), [pcg, pcv], lambda: ops.posconv_weight(pcv, pcg.reshape(-1), G))
        K = Kw * Cg
        s.u = torch.empty((M, H), device=feats.device, dtype=torch.bfloat16) if need else None
        s.s = torch.empty((M, H), device=feats.device, dtype=torch.bfloat16)
        if ops.posconv_kernel_fits(H, G, Kw):            # wav2vec2-base: dedicated Toeplitz-window kernel (csrc/posconv.hip)
            ops.posconv_gemm(xg, s.wf, s.s, g.B, g.Tp, H, G, Kw, pad, bias=pcb, gelu=True, residual=h0, out_pre=s.u)
        else:
            ops.gemm(xg, s.wf, g.Tp, Cg, K, lda=Cg, ldb=K, out=s.s, ldc=H, bias=pcb, gelu=True, residual=h0, ldr=H, out_pre=s.u,
                     batch=self._posconv_batch(Cg, rows_p, K))
        s.xg_key = "pc_x"
        self.model._scratch_owner["pc_x"] = s
        if cfg.do_stable_layer_norm:
            y = s.s
        else:
            y, s.m1, s.r1 = ops.layernorm_fwd(s.s, elw, elb, cfg.layer_norm_eps, save_stats=need)
        s.p_h = cfg.hidden_dropout if tr else 0.0
        if s.p_h > 0:
            y = ops.dropout(y, s.p_h, _seed(self.seed, 12))
        return (y,), (s if need else None)

    def bwd(self, s, grads, x_needs):
        cfg, g = self.cfg, self.g
        (fplw, fplb, pw, pb, embed, pcg, pcv, pcb, elw, elb) = s.params
        M, H, C = g.M, cfg.hidden_size, s.feats.shape[1]
        G, Kw = cfg.num_conv_pos_embedding_groups, cfg.num_conv_pos_embeddings
        dy = grads[0].contiguous()
        if s.p_h > 0:
            dy = ops.dropout(dy, s.p_h, _seed(self.seed, 12))
        delw = delb = None
        if cfg.do_stable_layer_norm:
            ds = dy
        else:
            ds, _, delw, delb = ops.layernorm_bwd(dy, s.s, s.m1, s.r1, elw)
        # ---- positional conv backward
        dug, rows_p, Cg, pad = self._packed("pc_du", dy.device)
        K = Kw * Cg
        du_rm = torch.empty((M, H), device=dy.device, dtype=torch.bfloat16)
        ops.posconv_pack(ds, dug, g.B, g.Tp, H, G, pad, u=s.u, rowmajor_out=du_rm)
        dpcb = ops.colsum(du_rm, M, H)
        dh0 = torch.empty((M, H), device=dy.device, dtype=torch.bfloat16)
        if ops.posconv_kernel_fits(H, G, Kw):
            ops.posconv_gemm(dug, s.wd, dh0, g.B, g.Tp, H, G, Kw, pad, first_row=1, residual=ds)
        else:
            ops.gemm(dug[Cg:], s.wd, g.Tp, Cg, K, lda=Cg, ldb=K, out=dh0, ldc=H, residual=ds, ldr=H,
                     batch=self._posconv_batch(Cg, rows_p, K))
        xg, _, _, _ = self._packed(s.xg_key, dy.device)
        if self.model._scratch_owner.get(s.xg_key) is not s:
            ops.posconv_pack(s.h0, xg, g.B, g.Tp, H, G, pad)         # another forward reused the scratch: repack
        kred = g.B * rows_p - 2 * pad
        dwf = torch.empty((G, Cg, K), device=dy.device, dtype=torch.float32)
        if ops.posconv_kernel_fits(H, G, Kw, wgrad=True):
            ops.posconv_wgrad(dug, xg, dwf, g.B, g.Tp, H, G, Kw, pad)
        else:
            ops.gemm(dug[pad * Cg:], xg, Cg, K, kred, a_kmajor=True, b_kmajor=True, out_f32=True, lda=Cg, ldb=Cg, out=dwf, ldc=K,
                     batch=dict(outer=1, inner=G, a=(0, g.B * rows_p * Cg), b=(0, g.B * rows_p * Cg), c=(0, Cg * K)))
        # weight-norm backward (parameter-sized fp32 math): w = g * v / ||v||
        dpcv, dgain = ops.posconv_weight_bwd(dwf, pcv.detach().contiguous(), pcg.detach().reshape(-1).contiguous(), s.norm, G)
        dpcg = dgain.reshape(pcg.shape)
        # ---- mask + projection + LN
        dembed = ops.frame_mask_bwd(dh0, self.lens, self.spec, g.B, g.Tp, g.T, H, want_dembed=embed is not None)
        if s.p_fp > 0:
            dh0 = ops.dropout(dh0, s.p_fp, _seed(self.seed, 11))
        sk = max(1, min(8, M // 2048))
        dpw = ops.gemm(dh0, s.n0, H, C, M, a_kmajor=True, b_kmajor=True, out_f32=True, split_k=sk)
        dpb = ops.colsum(dh0, M, H)
        wp = self.model._cached(("proj",), [pw], lambda: ops.cast_bf16(pw))
        dn0 = ops.gemm(dh0, wp, M, C, H, b_kmajor=True)
        dfeats, _, dfplw, dfplb = ops.layernorm_bwd(dn0, s.feats, s.m0, s.r0, fplw)
        return (dfeats if x_needs else None), (dfplw, dfplb, dpw, dpb, dembed, dpcg, dpcv, dpcb, delw, delb)


class _FinalLNImpl:
    """encoder.layer_norm after the last layer of pre-LN models (HF:791)."""

    def __init__(self, cfg, geom):
        self.cfg, self.g = cfg, geom

    def fwd(self, x, params, need):
        y, m, r = ops.layernorm_fwd(x, params[0], params[1], self.cfg.layer_norm_eps, save_stats=need)
        return (y,), (SimpleNamespace(x=x, m=m, r=r, w=params[0]) if need else None)

    def bwd(self, s, grads, x_needs):
        dx, _, dg, db = ops.layernorm_bwd(grads[0].contiguous(), s.x, s.m, s.r, s.w)
        return dx, (dg, db)


# =================================================================================== the model
class Wav2Vec2Model(nn.Module):
    """Drop-in for ``transformers.Wav2Vec2Model`` on the reference's call pattern."""

    def __init__(self, config):
        super().__init__()
        cfg = W2V2Config.from_any(config)
        self.config = cfg
        H, I = cfg.hidden_size, cfg.intermediate_size
        if cfg.head_dim != 64:
            raise ValueError("the attention kernels are built for head_dim 64 (wav2vec2 base/large)")
        if any(c != 512 for c in cfg.conv_dim) or cfg.conv_kernel[0] != 10 or cfg.conv_stride[0] != 5:
            raise ValueError("the feature-encoder kernels are built for conv_dim=512, first layer k=10/s=5")
        # ---- feature encoder (HF:382-419)
        fe = _Holder()
        layers = []
        cin = 1
        for i, (c, k) in enumerate(zip(cfg.conv_dim, cfg.conv_kernel)):
            l = _Holder()
            l.conv = _Holder()
            w = torch.empty(c, cin, k)
            nn.init.kaiming_normal_(w)
            l.conv.weight = nn.Parameter(w)
            if cfg.conv_bias:
                bound = math.sqrt(1.0 / (cin * k))
                l.conv.bias = nn.Parameter(torch.empty(c).uniform_(-bound, bound))
            if cfg.feat_extract_norm == "layer" or i == 0:
                l.layer_norm = _norm(c)
            layers.append(l)
            cin = c
        fe.conv_layers = nn.ModuleList(layers)
        self.feature_extractor = fe
        self._fe_requires_grad = True
        # ---- feature projection (HF:422-434)
        fp = _Holder()
        fp.layer_norm = _norm(cin)
        fp.projection = _linear(H, cin)
        self.feature_projection = fp
        if cfg.mask_time_prob > 0.0 or cfg.mask_feature_prob > 0.0:
            self.masked_spec_embed = nn.Parameter(torch.empty(H).uniform_())
        # ---- encoder (HF:657-802)
        enc = _Holder()
        pc = _Holder()
        pc.conv = _Holder()
        k, grp = cfg.num_conv_pos_embeddings, cfg.num_conv_pos_embedding_groups
        pc.conv.bias = nn.Parameter(torch.zeros(H))
        pc.conv.parametrizations = _Holder()
        pc.conv.parametrizations.weight = _Holder()
        v = torch.randn(H, H // grp, k) * (2 * math.sqrt(1 / (k * H)))
        pc.conv.parametrizations.weight.original0 = nn.Parameter(v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt())
        pc.conv.parametrizations.weight.original1 = nn.Parameter(v)
        enc.pos_conv_embed = pc
        enc.layer_norm = _norm(H)
        ls = []
        for _ in range(cfg.num_hidden_layers):
            l = _Holder()
            l.attention = _Holder()
            for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
                setattr(l.attention, n, _linear(H, H))
            l.layer_norm = _norm(H)
            l.feed_forward = _Holder()
            l.feed_forward.intermediate_dense = _linear(I, H)
            l.feed_forward.output_dense = _linear(H, I)
            l.final_layer_norm = _norm(H)
            ls.append(l)
        enc.layers = nn.ModuleList(ls)
        self.encoder = enc
        self._cache = {}
        self._scratch_bufs = {}
        self._scratch_owner = {}
        self._step = 0
        self.base_seed = 0x5EED
        # LayerDrop coins come from a dedicated generator so that data-parallel ranks (same seed) drop the same layers
        self._layerdrop_gen = torch.Generator().manual_seed(0x1A7E)

    # ------------------------------------------------------------------ reference-facing helpers
    @classmethod
    def from_pretrained(cls, model_id_or_path, config=None, cache_dir=None, **kw):
        """Loads a LOCAL HuggingFace checkpoint directory (config.json + model.safetensors | pytorch_model.bin).
        Hub names cannot be resolved offline and raise."""
        path = str(model_id_or_path)
        if not os.path.isdir(path):
            raise FileNotFoundError(
                f"{path!r} is not a local checkpoint directory; hub downloads are not available to this build")
        cfg = W2V2Config.from_any(config) if config is not None else W2V2Config.from_pretrained_dir(path)
        model = cls(cfg)
        sd = None
        st = os.path.join(path, "model.safetensors")
        pb = os.path.join(path, "pytorch_model.bin")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        elif os.path.exists(pb):
            sd = torch.load(pb, map_location="cpu", weights_only=True)
        if sd is None:
            raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin under {path}")
        sd = {(k[len("wav2vec2."):] if k.startswith("wav2vec2.") else k): v for k, v in sd.items()}
        own = model.state_dict()
        # legacy weight-norm names (weight_g / weight_v) -> parametrizations
        ren = {"encoder.pos_conv_embed.conv.weight_g": "encoder.pos_conv_embed.conv.parametrizations.weight.original0",
               "encoder.pos_conv_embed.conv.weight_v": "encoder.pos_conv_embed.conv.parametrizations.weight.original1"}
        sd = {ren.get(k, k): v for k, v in sd.items()}
        missing = [k for k in own if k not in sd]
        if missing:
            raise KeyError(f"checkpoint {path} lacks {len(missing)} tensors, e.g. {missing[:3]}")
        model.load_state_dict({k: sd[k] for k in own})
        return model

    def save_pretrained(self, path):
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(self.config.to_dict(), f, indent=1)
        from safetensors.torch import save_file
        save_file({k: v.contiguous() for k, v in self.state_dict().items()}, os.path.join(path, "model.safetensors"))

    def set_encoder_precision(self, precision: str = "bf16"):
        """"bf16" (default) or "mxfp8": the latter runs the Linear layers of the transformer stack with OCP MXFP8 operands on the
        block-scaled matrix instruction - INFERENCE ONLY (eval mode, no gradients: the frozen recogniser inside Force_APTAI,
        BASELINE configs[4]); any forward that needs gradients keeps the bf16 kernels."""
        # "bf16_f32res": bf16 GEMM operands and attention, but the RESIDUAL STREAM of the transformer stack in fp32 (LayerNorm reads
        # fp32, out-proj / FFN2 add the fp32 residual in their epilogues) and an fp32 last hidden state for the heads - inference
        # only; shrinks the bf16 noise band of the integer outputs downstream (forced-alignment indices) at ~+10 % encoder time
        # "f32x3" / "f32x6": every matrix product of the encoder (conv stack, projection, the four Linear layers, attention, positional
        # conv) at fp32-class accuracy - bf16 split-operand products with fp32 accumulation on the bf16 matrix pipe (3 or 6 bf16 x bf16
        # products per fp32 product, csrc/exact.hip) and the fp32 matrix instruction for attention / positional conv - with fp32
        # LayerNorm, softmax and erf GELU in between: INFERENCE ONLY, ~3-6 x the bf16 encoder time.  For outputs that must equal the
        # reference's INDEX for index (Force_APTAI's alignment argmax, the best-path decode).
        if precision not in ("bf16", "mxfp8", "bf16_f32res", "f32x3", "f32x6"):
            raise ValueError("encoder precision must be 'bf16', 'mxfp8', 'bf16_f32res', 'f32x3' or 'f32x6'")
        self._encoder_precision = precision
        return self

    def _mx_layer_weights(self, i: int):
        """MXFP8 copies (elements, scales) of layer i's four weight matrices, quantised from the bf16 compute copies; rebuilt only
        when the parameters moved (eval mode trusts Tensor._version, see _cached)."""
        e = self._layer_plan().entries[i]

        def build():
            return SimpleNamespace(wqkv=ops.mx_quantize(e.wqkv), wo=ops.mx_quantize(e.wo), w1=ops.mx_quantize(e.w1),
                                   w2=ops.mx_quantize(e.w2))
        return self._cached(("mx", i), self._layer_params(i), build)

    def gradient_checkpointing_enable(self, *a, **kw):
        """Accepted for drop-in compatibility (models/aptai.py:38): 288 GB of HBM make recomputation unnecessary."""
        return None

    def freeze_feature_encoder(self):
        """HF:1265-1270."""
        for p in self.feature_extractor.parameters():
            p.requires_grad = False
        self._fe_requires_grad = False

    def _get_feat_extract_output_lengths(self, input_lengths, add_adapter=None):
        return hostlogic.feat_extract_output_lengths(input_lengths, self.config.conv_kernel, self.config.conv_stride)

    # ------------------------------------------------------------------ bf16 compute copies of the parameters
    def _versions(self, params):
        # Tensor._version alone is not enough: optimisers with their own kernels (torch's fused Adam, aptai_amd.optim.Adam) update
        # parameters in place without bumping it.  `_train_marker` counts training steps of this model (every training-mode forward
        # under autograd, every replayed graph step), so a copy built in eval mode is rebuilt by the first eval forward after any
        # training step - validation then sees the LAST update, not the copies the last training forward made.
        # (training-mode forwards do not key on it: trainable copies are rebuilt on every forward there, frozen ones never move)
        return (0 if self.training else getattr(self, "_train_marker", 0),) + tuple((p.data_ptr(), p._version) for p in params)

    def _cached(self, key, params, build):
        mode = getattr(self, "_cache_mode", None)            # graph capture: "build" = always rebuild, "frozen" = always hit
        if mode == "frozen":
            return self._cache[key][1]
        ver = self._versions(params)
        hit = self._cache.get(key)
        # In training mode the copies are rebuilt on every forward: optimisers with fused kernels (e.g.
        # torch.optim.Adam(fused=True)) update parameters in place WITHOUT bumping Tensor._version, so a version check
        # would silently keep stale weights.  Eval mode trusts the version.  (The transformer-layer copies go through
        # _refresh_layer_copies / ops.CastPlan instead: one launch, or none when the optimiser publishes them.)
        trainable = self.training and any(p.requires_grad for p in params)
        if mode != "build" and not trainable and hit is not None and hit[0] == ver:
            return hit[1]
        with torch.no_grad():
            val = build()
        self._cache[key] = (ver, val)
        return val

    def _layer_params(self, i: int):
        l = self.encoder.layers[i]
        at, ff = l.attention, l.feed_forward
        return [at.q_proj.weight, at.k_proj.weight, at.v_proj.weight, at.q_proj.bias, at.k_proj.bias, at.v_proj.bias,
                at.out_proj.weight, at.out_proj.bias, ff.intermediate_dense.weight, ff.intermediate_dense.bias,
                ff.output_dense.weight, ff.output_dense.bias]

    def _layer_plan(self):
        """Persistent bf16 copies of every transformer-layer weight (+ packed fp32 q/k/v biases) and the ONE-launch cast
        plan (ops.CastPlan) that refreshes them: 12 layers x 9 jobs instead of ~90 separate cast / cat kernels."""
        plan = getattr(self, "_lplan", None)
        if plan is not None and not plan.stale():
            return plan
        H, I = self.config.hidden_size, self.config.intermediate_size
        jobs, entries = [], []
        for i in range(len(self.encoder.layers)):
            p = self._layer_params(i)
            dev = p[0].device
            bf = lambda *shape: torch.empty(shape, device=dev, dtype=torch.bfloat16)
            e = SimpleNamespace(wqkv=bf(3 * H, H), bqkv=torch.empty(3 * H, device=dev, dtype=torch.float32),
                                wo=bf(H, H), w1=bf(I, H), w2=bf(H, I), bo=p[7].detach(), b1=p[9].detach(), b2=p[11].detach())
            for j in range(3):
                jobs.append((p[j].detach(), e.wqkv[j * H:(j + 1) * H]))
                jobs.append((p[3 + j].detach(), e.bqkv[j * H:(j + 1) * H]))
            jobs += [(p[6].detach(), e.wo), (p[8].detach(), e.w1), (p[10].detach(), e.w2)]
            entries.append(e)
        plan = ops.CastPlan(jobs)
        plan.entries = entries
        plan.version = None
        self._lplan = plan
        return plan

    def _refresh_layer_copies(self, force: bool = False):
        """Runs the cast plan when the copies may be stale.  Training mode: every forward (fused optimisers update
        parameters in place without bumping Tensor._version, see _cached); eval mode: when a parameter version moved."""
        mode = getattr(self, "_cache_mode", None)
        if mode == "frozen":
            return
        plan = self._layer_plan()
        params = [p for i in range(len(self.encoder.layers)) for p in self._layer_params(i)]
        ver = self._versions(params)
        trainable = self.training and any(p.requires_grad for p in params)
        # an eval forward right after training must not trust the versions either (the last optimiser step is unseen) -
        # unless the optimiser itself writes the copies (aptai_amd.optim.Adam.publish_to sets plan.optimizer_synced)
        synced = getattr(plan, "optimizer_synced", False)
        stale_by_training = (trainable or getattr(plan, "after_training", False)) and not synced
        if force or mode == "build" or stale_by_training or plan.version != ver:
            with torch.no_grad():
                plan.run()
            plan.version = ver
            plan.after_training = bool(trainable) and not synced

    def _layer_weights(self, i: int, M: int):
        e = self._layer_plan().entries[i]
        e.split_k = self._split_k(M)
        return e, self._layer_params(i)

    def _split_k(self, M: int):
        """split-K factors for the four wgrad GEMMs (qkv, out, ffn1, ffn2): fill ~2 blocks per CU."""
        H, I = self.config.hidden_size, self.config.intermediate_size

        def pick(rows, cols):
            tiles = ((rows + 127) // 128) * ((cols + 127) // 128)
            s = max(1, min(16, 512 // max(tiles, 1)))
            return max(1, min(s, M // 512 if M >= 512 else 1))
        return (pick(3 * H, H), pick(H, H), pick(I, H), pick(H, I))

    def _scratch(self, key, numel, dev):
        """Persistent zero-initialised bf16 scratch (the zero gap rows of the packed positional-conv operands are
        written once here and never again)."""
        # one buffer per (key, size): a buffer is NEVER replaced or freed while the model lives - captured hipGraphs (the
        # graph runners, Force_APTAI.prefetch) hold its address, and a forward with another batch shape in between (validation
        # at batch 1 between training epochs) used to swap it out from under them: the next replays read freed memory
        full = (key, int(numel), str(dev))
        cur = self._scratch_bufs.get(full)
        if cur is None:
            cur = torch.zeros(numel, device=dev, dtype=torch.bfloat16)
            self._scratch_bufs[full] = cur
        return cur

    # ------------------------------------------------------------------ geometry of one batch
    def _geometry(self, B: int, S: int) -> SimpleNamespace:
        cfg = self.config
        Tl = hostlogic.conv_layer_lengths(S, cfg.conv_kernel, cfg.conv_stride)
        T = Tl[-1]
        if T < 1:
            raise ValueError(f"input of {S} samples is shorter than the receptive field of the feature encoder")
        Tp = _round_up(T + 2, 128)
        alloc = [0] * len(Tl)
        alloc[-1] = Tp
        for i in range(len(Tl) - 2, -1, -1):
            alloc[i] = alloc[i + 1] * cfg.conv_stride[i + 1]
        return SimpleNamespace(B=B, S=S, T=T, Tp=Tp, M=B * Tp, Tl=Tl, alloc=alloc)

    # ------------------------------------------------------------------ feature encoder (conv stack)
    def _conv_weights(self):
        cl = self.feature_extractor.conv_layers
        params = [l.conv.weight for l in cl[1:]]
        return self._cached(("convw",), params, lambda: [ops.conv_weight_bf16(p) for p in params])

    def _conv_params(self):
        """Flat parameter list of the conv stack in a fixed order: per layer weight, [bias], [norm weight, norm bias]."""
        cfg = self.config
        out = []
        for i, l in enumerate(self.feature_extractor.conv_layers):
            out.append(l.conv.weight)
            if cfg.conv_bias:
                out.append(l.conv.bias)
            if cfg.feat_extract_norm == "layer" or i == 0:
                out += [l.layer_norm.weight, l.layer_norm.bias]
        return out

    def _conv_forward(self, audio, g, save):
        """7-layer feature encoder (HF:382-419).  Returns (features [M][512] bf16, saved-for-backward | None)."""
        cfg = self.config
        cl = self.feature_extractor.conv_layers
        dev = audio.device
        C = 512
        layer_mode = cfg.feat_extract_norm == "layer"
        sv = SimpleNamespace(bufs=[], pre=[None], stats=[None], audio=audio) if save else None

        def new_buf(i, rows):
            # frozen path: persistent scratch (its slack rows were zeroed once); trainable path: fresh buffers (saved)
            if save:
                return _rows_with_zero_slack(rows, C, dev)
            if i == len(cl) - 1:          # the features outlive this call (saved by the projection's backward): own storage
                return torch.empty((rows, C), device=dev, dtype=torch.bfloat16)
            return self._scratch(("conv", i, rows), rows * C, dev).view(rows, C)
        buf = new_buf(0, g.B * g.alloc[0] + 8)
        l0 = cl[0]
        stats0 = ops.conv0_fwd(audio, l0.conv.weight, l0.conv.bias if cfg.conv_bias else None, l0.layer_norm.weight,
                               l0.layer_norm.bias, 1 if layer_mode else 0, buf, g.Tl[0], g.alloc[0], want_stats=save)
        if save:
            sv.bufs.append(buf)
            sv.stats0 = stats0
        ws = self._conv_weights()
        for i in range(1, len(cl)):
            k, s = cfg.conv_kernel[i], cfg.conv_stride[i]
            Mi = g.B * g.alloc[i]
            out = new_buf(i, Mi + 8)
            bias = cl[i].conv.bias if cfg.conv_bias else None
            if layer_mode:
                u = _rows_with_zero_slack(Mi + 8, C, dev) if save else out
                ops.gemm(buf, ws[i - 1], Mi, C, k * C, lda=s * C, out=u, ldc=C, bias=bias)
                _, m, r = ops.layernorm_fwd(u[:Mi], cl[i].layer_norm.weight, cl[i].layer_norm.bias, 1e-5, gelu_after=True,
                                            save_stats=save, out=out[:Mi])
                if save:
                    sv.pre.append(u)
                    sv.stats.append((m, r))
            else:
                u = _rows_with_zero_slack(Mi + 8, C, dev) if save else None
                ops.gemm(buf, ws[i - 1], Mi, C, k * C, lda=s * C, out=out, ldc=C, bias=bias, gelu=True, out_pre=u)
                if save:
                    sv.pre.append(u)
                    sv.stats.append(None)
            buf = out
            if save:
                sv.bufs.append(buf)
        return buf[:g.M], sv

    def _conv_backward(self, sv, g, dfeats):
        """Gradients of every conv-stack parameter (order of _conv_params).  dgrad of a strided conv = the weight GEMM on the
        SAME overlapping-row view: columns [0, s*C) of the virtual-row gradient own frames s*t .. s*t+s-1 outright, the
        remaining (k-s)*C columns are accumulated one virtual row later."""
        cfg = self.config
        cl = self.feature_extractor.conv_layers
        C = 512
        layer_mode = cfg.feat_extract_norm == "layer"
        ws = self._conv_weights()
        dev = dfeats.device
        grads = {}
        dy = dfeats                                          # gradient w.r.t. the OUTPUT of layer i (post-activation)
        du_ready = False                                     # base mode: dgrad epilogues already applied gelu' of the layer below
        for i in range(len(cl) - 1, 0, -1):
            k, s = cfg.conv_kernel[i], cfg.conv_stride[i]
            Mi = g.B * g.alloc[i]
            x_in = sv.bufs[i - 1]
            if layer_mode:
                m, r = sv.stats[i]
                du, _, dgam, dbet = ops.layernorm_bwd(dy[:Mi], sv.pre[i][:Mi], m, r, cl[i].layer_norm.weight,
                                                      beta_gelu=cl[i].layer_norm.bias)
                grads[(i, "ln_w")], grads[(i, "ln_b")] = dgam, dbet
            else:
                du = dy[:Mi] if du_ready else ops.dgelu(dy[:Mi].contiguous(), sv.pre[i][:Mi])
            if cfg.conv_bias:
                grads[(i, "b")] = ops.colsum(du, Mi, C)
            sk = max(1, min(16, Mi // 4096))
            dw = ops.gemm(du, x_in, C, k * C, Mi, a_kmajor=True, b_kmajor=True, out_f32=True, ldb=s * C, split_k=sk)
            grads[(i, "w")] = dw.view(C, k, C).permute(0, 2, 1).contiguous()          # [N][kw][c] -> nn.Conv1d's [N][c][kw]
            # ---- dgrad into the output of layer i-1
            dx = _rows_with_zero_slack(g.B * g.alloc[i - 1] + 8, C, dev)
            fuse = (not layer_mode) and (i - 1 >= 1)         # base: fold gelu'(u_{i-1}) into the epilogue
            aux = sv.pre[i - 1] if fuse else None
            ops.gemm(du, ws[i - 1], Mi, s * C, C, b_kmajor=True, ldb=k * C, out=dx, ldc=s * C, dgelu_aux=aux,
                     **({"ldaux": s * C} if fuse else {}))
            if k > s:
                n2 = (k - s) * C
                ops.gemm(du, ws[i - 1][:, s * C:], Mi, n2, C, b_kmajor=True, ldb=k * C, out=dx[s:], ldc=s * C, residual=dx[s:],
                         ldr=s * C, dgelu_aux=(aux[s:] if fuse else None), **({"ldaux": s * C} if fuse else {}))
            dy = dx
            du_ready = fuse
        l0 = cl[0]
        dw0, db0, dg0, dbt0 = ops.conv0_bwd(sv.audio, l0.conv.weight, l0.conv.bias if cfg.conv_bias else None, l0.layer_norm.weight,
                                            l0.layer_norm.bias, 1 if layer_mode else 0, dy, g.Tl[0], g.alloc[0], sv.stats0)
        grads[(0, "w")], grads[(0, "b")], grads[(0, "ln_w")], grads[(0, "ln_b")] = dw0, db0, dg0, dbt0
        out = []
        for i in range(len(cl)):
            out.append(grads[(i, "w")])
            if cfg.conv_bias:
                out.append(grads[(i, "b")])
            if layer_mode or i == 0:
                out += [grads[(i, "ln_w")], grads[(i, "ln_b")]]
        return out

    def _feature_encoder(self, audio: torch.Tensor, g) -> torch.Tensor:
        params = self._conv_params()
        trainable = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        if not trainable:
            with torch.no_grad():
                return self._conv_forward(audio, g, save=False)[0]
        return _ConvStackFn.apply(audio, self, g, *params)

    # ------------------------------------------------------------------ exact (fp32-class) inference pass
    def _exact_weights(self, P: int):
        """Split (weight-side) copies of every GEMM weight and the fp32 weight-normed positional-conv weight, built once per
        parameter version (eval mode trusts Tensor._version, see _cached)."""
        cfg = self.config
        cl = self.feature_extractor.conv_layers
        fp = self.feature_projection
        pc = self.encoder.pos_conv_embed.conv
        params = [l.conv.weight for l in cl[1:]] + [fp.projection.weight, pc.parametrizations.weight.original0,
                                                     pc.parametrizations.weight.original1]
        for i in range(len(self.encoder.layers)):
            params += self._layer_params(i)

        def build():
            w = SimpleNamespace(conv=[], layers=[])
            for l in cl[1:]:
                cw = l.conv.weight.detach()                                  # [N][C][kw] -> [N][kw*C] (K index = kw*C + c)
                w.conv.append(ops.split_f32(cw.permute(0, 2, 1).reshape(cw.shape[0], -1).contiguous(), P, weight_side=True))
            w.proj = ops.split_f32(fp.projection.weight.detach().contiguous(), P, weight_side=True)
            # weight_norm(dim=2): w = g * v / ||v||_(0,1) per tap (HF:340-356), fp32, in the kernels' [group][out][kw*Cg + in] layout
            g0, v = pc.parametrizations.weight.original0.detach(), pc.parametrizations.weight.original1.detach()
            G = cfg.num_conv_pos_embedding_groups
            Hh, Cg, Kw = v.shape
            wn = v * (g0 / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt())
            # split form (round 4): the input channels of a group padded 48 -> 64, so that one FRAME of a group is exactly one K-tile of
            # the split layout and the Toeplitz rows of the implicit GEMM stay contiguous ([tap][64 channels] = tap * 64 + c)
            w64 = torch.zeros((G, Cg, Kw, 64), device=wn.device, dtype=torch.float32)
            w64[..., :Cg] = wn.view(G, Cg, Cg, Kw).permute(0, 1, 3, 2)
            w.posconv = ops.split_f32(w64.view(G * Cg, Kw * 64), P, weight_side=True)                 # [G Cg][P Kw 64]
            for i in range(len(self.encoder.layers)):
                p = [t.detach() for t in self._layer_params(i)]
                w.layers.append(SimpleNamespace(
                    wqkv=ops.split_f32(torch.cat(p[0:3]).contiguous(), P, weight_side=True), bqkv=torch.cat(p[3:6]).contiguous(),
                    wo=ops.split_f32(p[6].contiguous(), P, weight_side=True), bo=p[7],
                    w1=ops.split_f32(p[8].contiguous(), P, weight_side=True), b1=p[9],
                    w2=ops.split_f32(p[10].contiguous(), P, weight_side=True), b2=p[11]))
            return w
        return self._cached(("exact", P), params, build)

    def _exact_attention(self, a_s, w, lens_i32, g, P=3):
        """q|k|v projection + softmax(Q K^T / sqrt(d) + key mask) V per head (HF:438-548) at fp32-class accuracy, from the layer input's
        split pieces `a_s` [M][P H], in TWO launches:
          * the q|k|v projection as one split-operand GEMM whose result leaves already split (APTAI_EPI_SPLIT_OUT): Q in the
            activation-side piece order, K and V (columns >= H, `split_out_bcol`) in the weight-side order.  Per head the 64 feature
            columns are exactly one K-tile of the split layout, so head h is the column block [64 P h, 64 P (h + 1)) of its third;
          * aptai_attention_exact_fwd: scores, masked softmax, the split of the probabilities and P . V in one kernel - every product the
            3 (6) leading bf16 piece products in fp32, nothing but the context (as the out-projection's split A operand) stored.
        (Round 3: 24 launches per layer on the fp32 matrix instruction, 10 of the mode's 27 ms; round 4 first ran the scores, the softmax and
        P . V as three launches with the fp32 scores and the split probabilities in memory: 0.29 ms of the layer's 1.0.)"""
        cfg = self.config
        H, heads, B, Tp, M = cfg.hidden_size, cfg.num_attention_heads, g.B, g.Tp, g.M
        d = H // heads
        if d != 64:
            raise NotImplementedError("the exact attention is built for head_dim 64 (one K-tile of the split layout per head)")
        qkvs = ops.gemm_split(a_s, w.wqkv, M, 3 * H, H, P, bias=w.bqkv, split_out=True, split_bcol=H)          # [M][P 3H]: Q | K | V
        return ops.attention_exact_fwd(qkvs, lens_i32, B, Tp, H, heads, P, d ** -0.5)

    def _forward_exact(self, audio, g, lens_i32, P, output_hidden_states):
        """The whole encoder in eval mode at fp32-class accuracy (set_encoder_precision("f32x3" | "f32x6")).  Returns
        (last hidden state fp32 [M][H], list of hidden states fp32)."""
        cfg = self.config
        cl = self.feature_extractor.conv_layers
        dev = audio.device
        C, H, I, M = 512, cfg.hidden_size, cfg.intermediate_size, g.M
        eps = cfg.layer_norm_eps
        W = self._exact_weights(P)
        layer_mode = cfg.feat_extract_norm == "layer"
        # ---- feature encoder (HF:382-419)
        l0 = cl[0]
        stats = None
        if not layer_mode:                                   # GroupNorm statistics (double-precision window moments, csrc/conv.hip)
            scratch = self._scratch(("conv", 0, g.B * g.alloc[0] + 8), (g.B * g.alloc[0] + 8) * C, dev).view(-1, C)
            stats = ops.conv0_fwd(audio, l0.conv.weight, l0.conv.bias if cfg.conv_bias else None, l0.layer_norm.weight,
                                  l0.layer_norm.bias, 0, scratch, g.Tl[0], g.alloc[0], want_stats=True)
        if layer_mode:
            # wav2vec2-large: a LayerNorm sits between the conv layers, so every layer's fp32 output is stored, normalised and split
            buf = torch.empty((g.B * g.alloc[0] + 8, C), device=dev, dtype=torch.float32)
            buf[g.B * g.alloc[0]:].zero_()
            ops.conv0_fwd_f32(audio, l0.conv.weight, l0.conv.bias if cfg.conv_bias else None, l0.layer_norm.weight, l0.layer_norm.bias,
                              1, buf, g.Tl[0], g.alloc[0], stats)
            pending_gelu = False
            for i in range(1, len(cl)):
                k, s = cfg.conv_kernel[i], cfg.conv_stride[i]
                Mi = g.B * g.alloc[i]
                xs = ops.split_f32(buf, P, gelu=pending_gelu)
                out = torch.empty((Mi + 8, C), device=dev, dtype=torch.float32)
                out[Mi:].zero_()
                ops.gemm_split(xs, W.conv[i - 1], Mi, C, k * C, P, lda=s * C, bias=cl[i].conv.bias if cfg.conv_bias else None, out=out, ldc=C)
                _, y32 = ops.layernorm_fwd_f32in(out[:Mi], cl[i].layer_norm.weight, cl[i].layer_norm.bias, 1e-5, want_bf16=False)
                out[:Mi].copy_(y32)
                buf, pending_gelu = out, True
        else:
            # wav2vec2-base (round 4): no fp32 intermediate between the conv layers at all - the first layer writes its activated output as
            # split pieces (aptai_conv0_fwd_split), layers 1..5 leave their GEMMs GELU'd and split (APTAI_EPI_SPLIT_OUT | APTAI_EPI_GELU),
            # only the last layer's output is fp32 (it feeds the feature projection's LayerNorm).  The 8 slack rows the overlapping-row
            # view of the next layer can read are zeroed; frames beyond an utterance are padded frames like everywhere else.
            M0 = g.B * g.alloc[0]
            xs = torch.empty((M0 + 8, P * C), device=dev, dtype=torch.bfloat16)
            xs[M0:].zero_()
            ops.conv0_fwd_split(audio, l0.conv.weight, l0.conv.bias if cfg.conv_bias else None, l0.layer_norm.weight, l0.layer_norm.bias,
                                0, xs, P, g.Tl[0], g.alloc[0], stats)
            for i in range(1, len(cl)):
                k, s = cfg.conv_kernel[i], cfg.conv_stride[i]
                Mi = g.B * g.alloc[i]
                last = i == len(cl) - 1
                bias_i = cl[i].conv.bias if cfg.conv_bias else None
                if last:
                    buf = torch.empty((Mi + 8, C), device=dev, dtype=torch.float32)
                    buf[Mi:].zero_()
                    ops.gemm_split(xs, W.conv[i - 1], Mi, C, k * C, P, lda=s * C, bias=bias_i, out=buf, ldc=C)
                else:
                    nxt = torch.empty((Mi + 8, P * C), device=dev, dtype=torch.bfloat16)
                    nxt[Mi:].zero_()
                    ops.gemm_split(xs, W.conv[i - 1], Mi, C, k * C, P, lda=s * C, bias=bias_i, out=nxt, ldc=P * C, split_out=True, gelu=True)
                    xs = nxt
        feats = ops.bias_act_res_f32(buf[:M], gelu=True)                                               # [M][512] fp32
        # ---- feature projection, padded-frame zeroing, positional conv (HF:422-434, 678-681, 326-379)
        fp = self.feature_projection
        _, n0 = ops.layernorm_fwd_f32in(feats, fp.layer_norm.weight, fp.layer_norm.bias, eps, want_bf16=False)
        h0 = ops.gemm_split(ops.split_f32(n0, P), W.proj, M, H, C, P, bias=fp.projection.bias)
        ops.bias_act_res_f32(h0, lens_i32=lens_i32, rows_per_b=g.Tp, out=h0)
        G, Kw = cfg.num_conv_pos_embedding_groups, cfg.num_conv_pos_embeddings
        Cg, pad = H // G, Kw // 2
        if Cg > 64 or Cg % 8:
            raise NotImplementedError("the exact positional conv is built for at most 64 channels per group (a multiple of 8)")
        rows_p = g.Tp + 2 * pad
        # grouped Conv1d(k = 128, 'same') as ONE batched split-operand NT launch over (group, utterance): row t of group grp is the Kw
        # frames t .. t + Kw - 1 of the zero-padded sequence, 64 (48 real) channels each - lda = one frame.  (Round 3 ran it as 16
        # batched calls on the fp32 matrix instruction: 2.4 ms of the exact-mode step.)
        xg = torch.zeros((G, g.B, rows_p, 64), device=dev, dtype=torch.float32)
        xg[:, :, pad:pad + g.Tp, :Cg].copy_(h0.view(g.B, g.Tp, G, Cg).permute(2, 0, 1, 3))
        xs = ops.split_f32(xg.view(G * g.B * rows_p, 64), P)                                           # [G B rows_p][P 64]
        pc = self.encoder.pos_conv_embed.conv
        conv = torch.empty((M, H), device=dev, dtype=torch.float32)
        KP = P * Kw * 64
        ops.gemm(xs, W.posconv, g.Tp, Cg, KP, lda=P * 64, ldb=KP, out=conv, ldc=H, out_f32=True, bias=pc.bias, tile=128,
                 batch=dict(outer=G, inner=g.B, a=(g.B * rows_p * P * 64, rows_p * P * 64), b=(Cg * KP, 0), c=(Cg, g.Tp * H), bias=(Cg, 0)))
        h = ops.bias_act_res_f32(conv, gelu=True, res=h0)
        h_s = None                                           # split pieces of h where a LayerNorm has just produced them
        if not cfg.do_stable_layer_norm:
            h, h_s = ops.layernorm_fwd_f32in_split(h, self.encoder.layer_norm.weight, self.encoder.layer_norm.bias, eps, P)
        # ---- transformer layers (HF:575-654).  Every LayerNorm whose result feeds a product also writes it split (round 4).
        hidden = []
        for i, layer in enumerate(self.encoder.layers):
            hidden.append(h)
            w = W.layers[i]
            ln1, ln2 = layer.layer_norm, layer.final_layer_norm
            if cfg.do_stable_layer_norm:
                _, a_s = ops.layernorm_fwd_f32in_split(h, ln1.weight, ln1.bias, eps, P, want_f32=False)
            else:
                a_s = h_s if h_s is not None else ops.split_f32(h, P)
            ctx = self._exact_attention(a_s, w, lens_i32, g, P)
            s1 = ops.gemm_split(ctx, w.wo, M, H, H, P, bias=w.bo, residual_f32=h)          # ctx: already split
            if cfg.do_stable_layer_norm:
                _, f_s = ops.layernorm_fwd_f32in_split(s1, ln2.weight, ln2.bias, eps, P, want_f32=False)
                res2 = s1
            else:
                res2, f_s = ops.layernorm_fwd_f32in_split(s1, ln1.weight, ln1.bias, eps, P)
            # FFN1's result leaves its GEMM already GELU'd and split (EPI_SPLIT_OUT): 100 MB of fp32 and a split pass less per layer
            us = ops.gemm_split(f_s, w.w1, M, I, H, P, bias=w.b1, split_out=True, gelu=True)
            s2 = ops.gemm_split(us, w.w2, M, H, I, P, bias=w.b2, residual_f32=res2)
            if cfg.do_stable_layer_norm:
                h, h_s = s2, None
            else:
                h, h_s = ops.layernorm_fwd_f32in_split(s2, ln2.weight, ln2.bias, eps, P)
        if cfg.do_stable_layer_norm:
            _, h = ops.layernorm_fwd_f32in(h, self.encoder.layer_norm.weight, self.encoder.layer_norm.bias, eps, want_bf16=False)
        hidden.append(h)
        return h, hidden

    # ------------------------------------------------------------------ forward
    def forward(self, input_values, attention_mask=None, mask_time_indices=None, output_attentions=None,
                output_hidden_states=None, return_dict=None, **kw):
        """HF:1319-1375.  ``attention_mask`` follows the reference's convention: the (B,1) tensor of sample
        counts (models/aptai.py:77), or a regular (B,S) 0/1 mask; both reduce to lengths via a row sum (HF:1023)."""
        cfg = self.config
        if not input_values.is_cuda:
            raise ops._lib.AptaiHipError("Wav2Vec2Model runs on the MI355X only (no CPU fallback)")
        if output_attentions:
            raise NotImplementedError("attention probabilities never leave the fused kernel")
        audio = input_values.float().contiguous()
        B, S = audio.shape
        g = self._geometry(B, S)
        dev = audio.device
        if attention_mask is not None:
            sample_lens = attention_mask.to(dev).long().sum(-1)
            frame_lens = hostlogic.feat_extract_output_lengths(sample_lens, cfg.conv_kernel, cfg.conv_stride)
            frame_lens = frame_lens.clamp(min=1, max=g.T)
        else:
            frame_lens = torch.full((B,), g.T, device=dev, dtype=torch.long)
        lens_i32 = frame_lens.to(torch.int32).contiguous()
        self._step += 1
        seed = _seed(self.base_seed, self._step)
        training = self.training
        if training and torch.is_grad_enabled():
            self._train_marker = getattr(self, "_train_marker", 0) + 1

        prec = getattr(self, "_encoder_precision", "bf16")
        if prec in ("f32x3", "f32x6") and not training and not torch.is_grad_enabled():
            if mask_time_indices is not None:
                raise NotImplementedError("the exact inference pass takes no SpecAugment mask (eval mode never samples one)")
            h32, hidden32 = self._forward_exact(audio, g, lens_i32, 3 if prec == "f32x3" else 6, output_hidden_states)
            hb = h32.to(torch.bfloat16)
            out = Wav2Vec2BaseModelOutput(last_hidden_state=h32.view(B, g.Tp, -1)[:, :g.T], extract_features=None,
                                          hidden_states=tuple(t.view(B, g.Tp, -1)[:, :g.T] for t in hidden32) if output_hidden_states else None,
                                          attentions=None)
            out._geom, out._frame_lens, out._flat_last, out._flat_last_f32, out._features = g, frame_lens, hb, h32, None
            return out
        feats = self._feature_encoder(audio, g)                                         # [M][512] bf16
        # ---- feature projection + SpecAugment + padded-frame zeroing (HF:429-434, 1272-1316, 678-681)
        spec = None
        if getattr(cfg, "apply_spec_augment", True):
            if mask_time_indices is not None:
                spec = torch.as_tensor(mask_time_indices).to(dev).to(torch.uint8).contiguous()
            elif cfg.mask_time_prob > 0 and training:
                if _SPEC_ON_DEVICE:
                    # sampled by a kernel from the device-resident lengths: no device->host copy (a 2.3 ms stall of the eager
                    # loop per step).  Same rule as HF `_compute_mask_indices`, counter-hash draws instead of numpy's.
                    spec = ops.spec_augment_mask(lens_i32, B, g.T, cfg.mask_time_prob, cfg.mask_time_length, cfg.mask_time_min_masks,
                                                 _seed(seed, 77))
                else:
                    am = None
                    if attention_mask is not None:
                        am = (torch.arange(g.T)[None, :] < frame_lens.cpu()[:, None])
                    m = hostlogic.compute_mask_indices((B, g.T), cfg.mask_time_prob, cfg.mask_time_length, attention_mask=am,
                                                       min_masks=cfg.mask_time_min_masks)
                    spec = torch.from_numpy(m.astype(np.uint8)).to(dev)
        fp = self.feature_projection
        embed = getattr(self, "masked_spec_embed", None)
        front = _FrontImpl(cfg, g, lens_i32, spec, training, seed, self)
        pc = self.encoder.pos_conv_embed.conv
        fparams = [fp.layer_norm.weight, fp.layer_norm.bias, fp.projection.weight, fp.projection.bias,
                   embed if (embed is not None and spec is not None) else None,
                   pc.parametrizations.weight.original0, pc.parametrizations.weight.original1, pc.bias,
                   self.encoder.layer_norm.weight, self.encoder.layer_norm.bias]
        h = _run(front, feats, *fparams)
        # ---- transformer layers with LayerDrop (HF:694-707 / 767-780)
        self._refresh_layer_copies()
        hidden = []
        if getattr(self, "_encoder_precision", "bf16") == "bf16_f32res" and not training and not torch.is_grad_enabled():
            # fp32 residual stream (inference only, see set_encoder_precision)
            h32 = h.float()
            for i, layer in enumerate(self.encoder.layers):
                hidden.append(h if h is not None else h32)
                wt, _lin = self._layer_weights(i, g.M)
                impl = _LayerImpl(cfg, g, lens_i32, wt, False, 0)
                h, h32 = impl.fwd_f32res(h, h32, [layer.layer_norm.weight, layer.layer_norm.bias, layer.final_layer_norm.weight,
                                                   layer.final_layer_norm.bias])
            if cfg.do_stable_layer_norm:
                h, h32 = ops.layernorm_fwd_f32in(h32, self.encoder.layer_norm.weight, self.encoder.layer_norm.bias, cfg.layer_norm_eps)
            hidden.append(h)
            out = Wav2Vec2BaseModelOutput(last_hidden_state=h.view(B, g.Tp, -1)[:, :g.T], extract_features=None,
                                          hidden_states=tuple(t.view(B, g.Tp, -1)[:, :g.T] for t in hidden) if output_hidden_states else None,
                                          attentions=None)
            out._geom, out._frame_lens, out._flat_last, out._flat_last_f32 = g, frame_lens, h, h32
            return out
        for i, layer in enumerate(self.encoder.layers):
            hidden.append(h)
            skip = training and cfg.layerdrop > 0 and (float(torch.rand([], generator=self._layerdrop_gen)) < cfg.layerdrop)
            if skip:
                continue
            w, lin_params = self._layer_weights(i, g.M)
            impl = _LayerImpl(cfg, g, lens_i32, w, training, _seed(seed, 100 + i))
            if getattr(self, "_encoder_precision", "bf16") == "mxfp8" and not training:
                impl.mx = self._mx_layer_weights(i)
            h = _run(impl, h, layer.layer_norm.weight, layer.layer_norm.bias, layer.final_layer_norm.weight,
                            layer.final_layer_norm.bias, *lin_params)
        if cfg.do_stable_layer_norm:
            fin = _FinalLNImpl(cfg, g)
            h = _run(fin, h, self.encoder.layer_norm.weight, self.encoder.layer_norm.bias)
        hidden.append(h)

        def view(t):
            return t.view(B, g.Tp, -1)[:, :g.T]
        out = Wav2Vec2BaseModelOutput(
            last_hidden_state=view(h),
            extract_features=None,
            hidden_states=tuple(view(t) for t in hidden) if output_hidden_states else None,
            attentions=None)
        out._geom = g
        out._frame_lens = frame_lens
        out._flat_last = h
        out._features = feats
        return out


