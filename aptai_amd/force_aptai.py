"""``Force_APTAI`` — drop-in for models/force_aptai.py:19-323 on MI355X: frozen ``Wav2Vec2_PR`` encoder (inference
only, models/w2v2_pr.py:124-127) -> cross-attention forced aligner + BiLSTM regression.  Same constructor
``(pr_model_path, device, vocab)``, ``forward(epoch, **batch)`` dict keys, helpers and state-dict keys.

Differences from the shipped reference, all forced by defects recorded in SURVEY.md §0:
 * batch > 1 works (the reference's ``RNN.forward`` raises NameError at models/modules.py:207; intent followed);
 * the CTC decode inside the step is the best path (torchaudio's beam decoder is absent: parity unpinned) and costs ONE
   device->host transfer per batch; the alignment read-out (:148-161, B*T host syncs in the reference) is one gather
   kernel + one transfer;
 * the encoder runs once per step (the reference ran the conv stack twice, models/w2v2_pr.py:129,132).
"""
from __future__ import annotations

import os
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .config import W2V2Config, load_model_cfg
from .hostlogic import TV_NAMES
from .modules import CrossAttention, ForwardSumLoss, LowPassFilterLayer, PositionalEncoding, RNN
from .w2v2_pr import Wav2Vec2_PR
from .wav2vec2 import _seed

_NPHN = 60


def force_heads_fwd(ac, st, P):
    """Everything of Force_APTAI.forward after the encoder, fp32 on the device (csrc/force.hip, lstm.hip, ctc.hip).  Returns
    ((loss, tv_loss, align_loss, tvs, frame_phns, att_log, att_out, hout, align), saved) - shared by the autograd path
    (_ForceHeadsFn) and the hipGraph runner (graphed.GraphedForceStep)."""
    (fl_w, fl_b, emb_w, q_w, q_b, k_w, k_b, ln_w, ln_b, wih0, whh0, bih0, bhh0, wih1, whh1, bih1, bhh1, l0_w, l0_b, l3_w,
     l3_b) = P
    g = st.g
    B, Tp, T, M, H = g.B, g.Tp, g.T, g.M, ac.shape[1]
    dev = ac.device
    s = SimpleNamespace()
    s.phn = ops.embed_pe_fwd(st.ids, emb_w, st.pe, _NPHN, st.p_hid, _seed(st.seed, 1))                    # [B*60][128]
    fh = ops.linear_f32(ac, fl_w, fl_b, rows=M)                                                           # [M][128]
    s.fhd = ops.dropout_f32(fh, st.p_hid, _seed(st.seed, 2))
    s.cat = torch.empty((M, 256), device=dev, dtype=torch.float32)
    q = s.cat[:, 128:]
    ops.linear_f32(s.fhd, q_w, q_b, out=q, ldc=256)
    s.k = ops.linear_f32(s.phn, k_w, k_b)                                                                  # [B*60][128]
    raw = ops.sgemm(q, 256, 1, s.k, 1, 128, Tp, _NPHN, 128, batch=B, bsa=Tp * 256, bsb=_NPHN * 128, bsc=Tp * _NPHN)
    # forward-sum (CTC) input rows [blank = -1 | att_log | 0], written by the same kernel
    s.pad = torch.empty((M, 64), device=dev, dtype=torch.float32)
    s.energy, s.att, s.att_log, s.align = ops.xattn_softmax_fwd(raw, st.ids, B, Tp, _NPHN, fs_rows=s.pad)
    ops.sgemm(s.att, _NPHN, 1, s.k, 128, 1, Tp, 128, _NPHN, out=s.cat, ldc=256, batch=B, bsa=Tp * _NPHN, bsb=_NPHN * 128,
              bsc=Tp * 256)
    s.att_out, s.lm, s.lr = ops.layernorm_f32_fwd(s.cat, ln_w, ln_b)
    # rows in the LSTM kernels' gate-interleaved order (csrc/lstm.hip: column dir * 1024 + unit * 4 + gate of xproj / gates / dgates)
    perm = ops.lstm_gate_perm(dev)[0]
    s.wih = torch.cat([wih0, wih1])[perm].contiguous()                                                    # [2048][256]
    bsum = torch.cat([bih0 + bhh0, bih1 + bhh1])[perm].contiguous()
    xproj = ops.linear_f32(s.att_out, s.wih, bsum)                                                         # [M][2048]
    s.whh = torch.stack([whh0, whh1]).contiguous()                                                         # [2][1024][256]
    s.hout, s.gates, s.cst = ops.lstm_fwd(xproj, s.whh, st.rnn_lens, B, Tp, T)
    h1 = ops.linear_f32(s.hout, l0_w, l0_b)
    s.h1a = ops.tanh_dropout_fwd(h1, st.p_rnn, _seed(st.seed, 3))
    tv_raw = ops.linear_f32(s.h1a, l3_w, l3_b)                                                             # [M][9]
    n_tv = l3_w.shape[0]
    s.tvs = torch.empty((B, T, n_tv), device=dev, dtype=torch.float32)
    ops.lowpass_fir(tv_raw, n_tv, Tp, st.taps, s.tvs, n_tv, T, B, T, T, n_tv, n_tv)
    s.dummy_logits = torch.zeros((M, 1), device=dev, dtype=torch.float32)
    s.dummy_phn = torch.zeros((B, T), device=dev, dtype=torch.int64)
    s.sc, _ = ops.aptai_loss_fwd(s.tvs, st.tv_tgt, s.dummy_logits, 1, Tp, s.dummy_phn, B, T, n_tv, 1, 1.0, 0.0, want_pred=False)
    # forward-sum (CTC) alignment loss on [blank=-1 | att_log]
    s.fs_loss, s.nll, _, s.alpha = ops.ctc_fwd(s.pad, 64, Tp, st.fs_targets, st.frame_lens, st.text_lens, B, T, _NPHN + 1,
                                               blank=0, reduction="mean", zero_infinity=True, vocab_sizes_i32=st.vocab_sizes,
                                               want_log_probs=False)
    tv_loss = s.sc[1].clone()
    align_loss = s.fs_loss.reshape(()).clone()
    loss = 0.4 * tv_loss + 0.6 * align_loss
    frame_phns = ops.gather_alignment(st.ids, s.align, st.frame_lens, B, Tp, _NPHN)
    return (loss, tv_loss, align_loss, s.tvs, frame_phns, s.att_log, s.att_out, s.hout, s.align), s


def force_heads_bwd(s, st, P, ac, gloss=None):
    """Gradients of the 21 head parameters (order of P) for dLoss = gloss (None = 1)."""
    (fl_w, fl_b, emb_w, q_w, q_b, k_w, k_b, ln_w, ln_b, wih0, whh0, bih0, bhh0, wih1, whh1, bih1, bhh1, l0_w, l0_b, l3_w,
     l3_b) = P
    g = st.g
    B, Tp, T, M, H = g.B, g.Tp, g.T, g.M, ac.shape[1]
    dev = ac.device
    n_tv = l3_w.shape[0]
    gl = torch.ones(1, device=ac.device, dtype=torch.float32) if gloss is None else gloss.float().reshape(1)
    # ---- TV branch
    norm = getattr(st, "norm_scalars", None)             # data parallel: global valid TV count / world (dp.GlobalLossNorm)
    d_tvs, _ = ops.aptai_loss_bwd(s.tvs, st.tv_tgt, s.dummy_logits, 1, Tp, s.dummy_phn, B, T, n_tv, 1, 1.0, 0.0,
                                  norm() if norm is not None else s.sc, (0.4 * gl).contiguous(), ldd=8)
    d_tvraw = torch.empty((M, n_tv), device=dev, dtype=torch.float32)
    ops.lowpass_fir(d_tvs, n_tv, T, st.taps, d_tvraw, n_tv, Tp, B, T, Tp, n_tv, n_tv)
    dl3_w = ops.sgemm(d_tvraw, 1, n_tv, s.h1a, 256, 1, n_tv, 256, M)
    dl3_b = ops.colsum_f32(d_tvraw, M, n_tv)
    dh1a = ops.sgemm(d_tvraw, n_tv, 1, l3_w, 256, 1, M, 256, n_tv)
    dh1 = ops.tanh_dropout_bwd(s.h1a, dh1a, st.p_rnn, _seed(st.seed, 3))
    dl0_w = ops.sgemm(dh1, 1, 256, s.hout, 512, 1, 256, 512, M)
    dl0_b = ops.colsum_f32(dh1, M, 256)
    dhout = ops.sgemm(dh1, 256, 1, l0_w, 512, 1, M, 512, 256)
    dgates = ops.lstm_bwd(dhout, s.whh, st.rnn_lens, s.gates, s.cst, B, Tp, T)                              # [M][2048]
    dwih = ops.sgemm(dgates, 1, 2048, s.att_out, 256, 1, 2048, 256, M)
    dbg = ops.colsum_f32(dgates, M, 2048)
    # dW_hh[dir] = sum_t dgates[t][dir]^T h_prev[t][dir]  (h_prev = previous VISITED frame: t-1 forward, t+1 reverse;
    # the rows in between utterances hold zeros in hout / dgates, so one shifted GEMM over all rows is exact)
    dwhh0 = ops.sgemm(dgates[1:], 1, 2048, s.hout, 512, 1, 1024, 256, M - 1)
    dwhh1 = ops.sgemm(dgates[:, 1024:], 1, 2048, s.hout[1:, 256:], 512, 1, 1024, 256, M - 1)
    datt_out = ops.sgemm(dgates, 2048, 1, s.wih, 256, 1, M, 256, 2048)
    # the four products above have the gate axis in the kernels' interleaved order: back to torch's gate-major rows
    inv = ops.lstm_gate_perm(dev)[1]
    dwih, dbg = dwih[inv], dbg[inv]
    dwhh0, dwhh1 = dwhh0[inv[:1024]], dwhh1[inv[:1024]]
    dcat, dln_w, dln_b = ops.layernorm_f32_bwd(datt_out, s.cat, s.lm, s.lr, ln_w)
    # ---- cross attention
    d_att = ops.sgemm(dcat, 256, 1, s.k, 1, 128, Tp, _NPHN, 128, batch=B, bsa=Tp * 256, bsb=_NPHN * 128, bsc=Tp * _NPHN)
    dk = ops.sgemm(s.att, 1, _NPHN, dcat, 256, 1, _NPHN, 128, Tp, batch=B, bsa=Tp * _NPHN, bsb=Tp * 256, bsc=_NPHN * 128)
    dpad = ops.ctc_bwd(s.pad, 64, Tp, st.fs_targets, st.frame_lens, st.text_lens, B, T, _NPHN + 1, s.alpha, s.nll,
                       (0.6 * gl).contiguous(), blank=0, reduction="mean", zero_infinity=True, vocab_sizes_i32=st.vocab_sizes,
                       ldd=64, out_dtype=torch.float32)
    d_raw = ops.xattn_softmax_bwd(s.att, s.att_log, d_att, dpad[:, 1:], ld_dattlog=64)     # columns 1..60 of the 64-float rows, in place
    dq = dcat[:, 128:].contiguous()
    ops.sgemm(d_raw, _NPHN, 1, s.k, 128, 1, Tp, 128, _NPHN, out=dq, ldc=128, accumulate=True, batch=B, bsa=Tp * _NPHN,
              bsb=_NPHN * 128, bsc=Tp * 128)
    ops.sgemm(d_raw, 1, _NPHN, s.cat[:, 128:], 256, 1, _NPHN, 128, Tp, out=dk, ldc=128, accumulate=True, batch=B,
              bsa=Tp * _NPHN, bsb=Tp * 256, bsc=_NPHN * 128)
    dq_w = ops.sgemm(dq, 1, 128, s.fhd, 128, 1, 128, 128, M)
    dq_b = ops.colsum_f32(dq, M, 128)
    dfhd = ops.sgemm(dq, 128, 1, q_w, 128, 1, M, 128, 128)
    dk_w = ops.sgemm(dk, 1, 128, s.phn, 128, 1, 128, 128, B * _NPHN)
    dk_b = ops.colsum_f32(dk, B * _NPHN, 128)
    dphn = ops.sgemm(dk, 128, 1, k_w, 128, 1, B * _NPHN, 128, 128)
    demb = ops.embed_bwd(st.ids, dphn, emb_w.shape[0], st.p_hid, _seed(st.seed, 1))
    dfh = ops.dropout_f32(dfhd, st.p_hid, _seed(st.seed, 2))
    dfl_wT = ops.sgemm(ac, 1, H, dfh, 128, 1, H, 128, M)                                                    # [H][128]
    dfl_b = ops.colsum_f32(dfh, M, 128)
    return (dfl_wT.t().contiguous(), dfl_b, demb, dq_w, dq_b, dk_w, dk_b, dln_w, dln_b,
            dwih[:1024], dwhh0, dbg[:1024], dbg[:1024], dwih[1024:], dwhh1, dbg[1024:], dbg[1024:], dl0_w, dl0_b, dl3_w, dl3_b)


class _ForceHeadsFn(torch.autograd.Function):
    """force_heads_fwd / force_heads_bwd behind autograd (the eager drop-in path)."""

    @staticmethod
    def forward(ctx, ac, st, *P):
        outs, s = force_heads_fwd(ac, st, P)
        ctx.st, ctx.saved, ctx.P, ctx.ac = st, s, P, ac
        ctx.mark_non_differentiable(*outs[1:])
        return outs

    @staticmethod
    def backward(ctx, gloss, *_):
        grads = force_heads_bwd(ctx.saved, ctx.st, ctx.P, ctx.ac, gloss)
        ctx.saved = None
        return (None, None) + tuple(grads)


class Force_APTAI(nn.Module):
    def __init__(self, pr_model_path, device, vocab):
        super().__init__()
        assert os.path.exists(pr_model_path)
        self.vocab = vocab
        self.device = device
        self.i = 0
        self.dp_loss_norm = None       # aptai_amd.dp.GlobalLossNorm under data parallelism
        self.hidden_drop = 0.2
        self.rnn_drop = 0.1
        self.max_phn_seq_len = _NPHN
        self.frame_hidden_dim = 128
        self.phn_hidden_dim = 128
        self.att_hidden_dim = 128
        self.rnn_in_dim = 2 * self.att_hidden_dim
        self.xatt = CrossAttention(self.frame_hidden_dim, self.phn_hidden_dim, self.att_hidden_dim)
        self.align_loss = ForwardSumLoss()
        # wav2vec2 phoneme recognizer (models/force_aptai.py:60-78): config pickle + state dict written by the PR training script
        self.pr_model_path = pr_model_path
        pr_ckpt_path = os.path.join(pr_model_path, 'best-model-ckpt')
        # restricted unpickler: plain containers / values and configuration objects as attribute bags, nothing callable
        self.w2v2_pr_cfg = load_model_cfg(os.path.join(pr_ckpt_path, 'model_cfg.pkl'))
        self.w2v2_pr = Wav2Vec2_PR(self.w2v2_pr_cfg['pretrain_cfg'], self.w2v2_pr_cfg['cache_dir'],
                                   self.w2v2_pr_cfg['huggingface_model_id'], vocab).to(self.device)
        self.w2v2_pr.load_state_dict(torch.load(os.path.join(pr_ckpt_path, 'pytorch_model.bin'),
                                                map_location=torch.device(self.device), weights_only=True))
        H = self.w2v2_pr.wav2vec2.config.hidden_size          # reference hard-codes 1024 (:43)
        self.frame_lin = nn.Linear(H, self.frame_hidden_dim)
        self.frame_drop = nn.Dropout(self.hidden_drop)
        self.phn_emb_layer = nn.Embedding(len(self.vocab), self.phn_hidden_dim, padding_idx=0)
        self.pe_phn = PositionalEncoding(self.phn_hidden_dim, max_len=_NPHN, dropout=self.hidden_drop)
        self.rnn = RNN(self.rnn_in_dim, 9, self.rnn_drop)
        self.tv_lowpass = LowPassFilterLayer(self.device, 10, 49, 9)
        for param in self.w2v2_pr.parameters():
            param.requires_grad = False
        # the frozen recogniser only ever runs forward here: keep its residual stream in fp32 and hand the heads an fp32 last hidden
        # state (+1.6 % step time, smaller bf16 noise band on the alignment indices; "bf16" restores the plain bf16 stream)
        self.w2v2_pr.wav2vec2.set_encoder_precision("bf16_f32res")
        self._enc_stream = None        # side stream of prefetch()
        self._enc_graphs = {}          # batch shape -> captured encoder pass
        self._enc_seen = {}
        self._prefetched = None

    # ------------------------------------------------------------------ shared body
    def _encode(self, audio_inputs, audio_lengths, phn_pred_list=None):
        """Frozen recogniser (inference) + device best-path decode on the CURRENT stream: everything the heads read from it."""
        pr = self.w2v2_pr
        pr.eval()                                                      # models/w2v2_pr.py:125: the recogniser always runs in eval mode
        with torch.no_grad():
            lens1d = audio_lengths.reshape(-1)
            out, _ = pr._logits_eval(audio_inputs, lens1d[:, None])
            dev = out._flat_last.device
            frame_lens = pr.wav2vec2._get_feat_extract_output_lengths(lens1d.to(dev)).to(torch.int32).contiguous()
            if phn_pred_list is None:
                ids, nlen = pr._decode_device(out, self.max_phn_seq_len)      # int32 [B][60] zero-padded, int32 [B]
            else:
                padded = []
                for lst in phn_pred_list:
                    assert len(lst) < self.max_phn_seq_len, 'Need longer max phoneme sequence length.'
                    padded.append(np.pad(np.asarray(lst, dtype=np.int64), (0, self.max_phn_seq_len - len(lst)), mode='constant'))
                ids = torch.tensor(np.array(padded), dtype=torch.int32, device=dev)
                nlen = torch.tensor([len(l) for l in phn_pred_list], dtype=torch.int32, device=dev)
        # with the fp32 residual stream the heads read the UNROUNDED last hidden state (the fp32 GEMM takes either dtype)
        ac = getattr(out, "_flat_last_f32", None)
        return SimpleNamespace(g=out._geom, ac=ac if ac is not None else out._flat_last, ids=ids, nlen=nlen, frame_lens=frame_lens,
                               step=pr.wav2vec2._step)

    def prefetch(self, audio_inputs, audio_lengths):
        """Run the frozen recogniser for a batch on a side stream NOW; the next forward / _run called with these same tensors
        picks the result up instead of encoding inline.  The recogniser is frozen and runs in eval mode, so WHEN it runs does
        not change any result; running it for batch n+1 beside the heads of batch n fills the CUs the cooperating LSTM
        kernels (32 workgroups for 3.5 ms) leave idle.  Each step still does one encoder pass and one heads pass.
        From the second call with a given batch shape on, the pass is ONE hipGraph launch (the recogniser has no trainable
        state and no host-dependent control flow): the step was otherwise bound by the host's ~350 launches."""
        dev = audio_inputs.device
        if self._enc_stream is None:
            self._enc_stream = torch.cuda.Stream(device=dev)
        cur = torch.cuda.current_stream(dev)
        self._enc_stream.wait_stream(cur)                              # the inputs (and the previous step's frees) are ordered first
        key = (tuple(audio_inputs.shape), tuple(audio_lengths.shape), audio_inputs.dtype, audio_lengths.dtype,
               getattr(self.w2v2_pr.wav2vec2, "_encoder_precision", "bf16"))
        use_graph = os.environ.get("APTAI_FORCE_ENC_GRAPH", "1") != "0"
        with torch.cuda.stream(self._enc_stream):
            ge = self._enc_graphs.get(key) if use_graph else None
            if ge is None and use_graph and self._enc_seen.get(key, 0) >= 1:
                ge = self._capture_encoder(key, audio_inputs, audio_lengths)
            if ge is not None:
                ge.audio.copy_(audio_inputs, non_blocking=True)
                ge.lengths.copy_(audio_lengths, non_blocking=True)
                ge.graph.replay()
                self.w2v2_pr.wav2vec2._step += 1                       # what the eager pass does on the host
                o = ge.out
                # the graph's outputs are overwritten by the next replay: hand out copies (13 MB, 4 us)
                enc = SimpleNamespace(g=o.g, ac=o.ac.clone(), ids=o.ids.clone(), nlen=o.nlen.clone(), frame_lens=o.frame_lens.clone(),
                                      step=self.w2v2_pr.wav2vec2._step)
            else:
                enc = self._encode(audio_inputs, audio_lengths)
                self._enc_seen[key] = self._enc_seen.get(key, 0) + 1
            enc.event = torch.cuda.Event()
            enc.event.record(self._enc_stream)
        self._prefetched = (audio_inputs, audio_lengths, enc)

    def _capture_encoder(self, key, audio_inputs, audio_lengths):
        """hipGraph of _encode for one batch shape, captured on the side stream (an eager pass with this shape has already run:
        weight copies and scratch buffers exist).  Falls back to eager launches if the capture fails."""
        ge = SimpleNamespace(audio=audio_inputs.clone(), lengths=audio_lengths.clone(), graph=torch.cuda.CUDAGraph(), out=None)
        try:
            step0 = self.w2v2_pr.wav2vec2._step
            torch.cuda.synchronize(audio_inputs.device)
            # thread-local capture: under data parallelism the process group's watchdog thread polls its events meanwhile
            # (APTAI_FORCE_ENC_TILE forces one GEMM tile in the side-stream pass - ops.auto_tile; default 0 = the dispatcher's rule, see aptai_amd/graphed.py)
            with torch.cuda.graph(ge.graph, stream=self._enc_stream, capture_error_mode="thread_local"), \
                    ops.auto_tile(int(os.environ.get("APTAI_FORCE_ENC_TILE", "0"))):
                ge.out = self._encode(ge.audio, ge.lengths)
            self.w2v2_pr.wav2vec2._step = step0                        # capturing issued nothing
        except Exception as e:                                         # noqa: BLE001 - keep training, eagerly
            import warnings
            warnings.warn(f"Force_APTAI: encoder hipGraph capture failed ({e!r}); the prefetch stays eager")
            torch.cuda.synchronize(audio_inputs.device)
            self._enc_graphs[key] = None
            self._enc_seen[key] = -(1 << 30)
            return None
        self._enc_graphs[key] = ge
        return ge

    def _take_prefetched(self, audio_inputs, audio_lengths):
        pf, self._prefetched = self._prefetched, None
        if pf is None or pf[0] is not audio_inputs or pf[1] is not audio_lengths:
            return None
        enc = pf[2]
        cur = torch.cuda.current_stream(audio_inputs.device)
        cur.wait_event(enc.event)
        for t in (enc.ac, enc.ids, enc.nlen, enc.frame_lens):         # allocated on the side stream, consumed on this one
            t.record_stream(cur)
        return enc

    def _heads_state(self, g, ids, nlen, frame_lens, tv_targets, step):
        """(st, P) of force_heads_fwd / force_heads_bwd for one encoded batch (all device tensors; no synchronisation)."""
        pr = self.w2v2_pr
        dev = ids.device
        tr = self.training
        n_tv = self.rnn.linear[3].weight.shape[0]
        if tv_targets is None:
            tv_targets = torch.full((g.B, g.T, n_tv), -100.0, device=dev)
        consts = self._consts(g.B, dev)
        # models/modules.py:209-212: the batch-1 branch runs the LSTM unpacked over ALL frames
        rnn_lens = consts["full_T"](g.T) if g.B == 1 else frame_lens
        st = SimpleNamespace(g=g, ids=ids, pe=self.pe_phn.pe.reshape(_NPHN, -1).contiguous(), taps=self.tv_lowpass.taps(),
                             p_hid=self.hidden_drop if tr else 0.0, p_rnn=self.rnn_drop if tr else 0.0,
                             seed=_seed(pr.wav2vec2.base_seed, step, 4242),
                             tv_tgt=tv_targets.contiguous(), fs_targets=consts["fs_targets"], frame_lens=frame_lens, rnn_lens=rnn_lens,
                             text_lens=nlen, vocab_sizes=nlen + 1)
        if getattr(self, "dp_loss_norm", None) is not None and tr:
            # the TV loss is a masked mean over the batch (models/force_aptai.py:137-141); the alignment and CTC terms are
            # means over utterances, which equal-sized shards already average exactly
            self.dp_loss_norm.begin(st.tv_tgt, None)
            st.norm_scalars = self.dp_loss_norm.scalars
        lstm = self.rnn.lstm
        P = (self.frame_lin.weight, self.frame_lin.bias, self.phn_emb_layer.weight, self.xatt.q.weight, self.xatt.q.bias,
             self.xatt.k.weight, self.xatt.k.bias, self.xatt.layer_norm.weight, self.xatt.layer_norm.bias,
             lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0, lstm.weight_ih_l0_reverse,
             lstm.weight_hh_l0_reverse, lstm.bias_ih_l0_reverse, lstm.bias_hh_l0_reverse, self.rnn.linear[0].weight,
             self.rnn.linear[0].bias, self.rnn.linear[3].weight, self.rnn.linear[3].bias)
        return st, P

    def _run(self, audio_inputs, audio_lengths, tv_targets=None, phn_pred_list=None, _ac_override=None, _prefetch_next=None):
        """Encoder (inference) -> decode -> heads.  Nothing in here synchronises host and device: the best-path decode, the
        phoneme slots, every length vector and the alignment read-out stay on the device; `_lists` makes the Python lists the
        reference returns with one round of transfers at the very end."""
        pr = self.w2v2_pr
        enc = self._take_prefetched(audio_inputs, audio_lengths) if phn_pred_list is None else None
        if enc is None:
            enc = self._encode(audio_inputs, audio_lengths, phn_pred_list)
        if _prefetch_next is not None:                                 # the NEXT batch's encoder pass goes out before this batch's heads
            self.prefetch(*_prefetch_next)
        g, ids, nlen, frame_lens = enc.g, enc.ids, enc.nlen, enc.frame_lens
        dev = enc.ac.device
        ac = enc.ac if _ac_override is None else _ac_override          # test hook: heads on given embeddings
        st, P = self._heads_state(g, ids, nlen, frame_lens, tv_targets, enc.step)
        res = _ForceHeadsFn.apply(ac, st, *P)
        return res, g, (ids, nlen, frame_lens, phn_pred_list)

    def _consts(self, B, dev):
        """Batch-size dependent constants (forward-sum targets 1..60, the all-frames length of the batch-1 branch)."""
        key = (B, str(dev))
        c = getattr(self, "_const_cache", {}).get(key)
        if c is None:
            full = {}

            def full_T(T):
                if T not in full:
                    full[T] = torch.full((B,), T, dtype=torch.int32, device=dev)
                return full[T]
            c = {"fs_targets": torch.arange(1, _NPHN + 1, dtype=torch.int32, device=dev)[None, :].repeat(B, 1).contiguous(),
                 "full_T": full_T}
            self._const_cache = dict(getattr(self, "_const_cache", {}))
            self._const_cache[key] = c
        return c

    def _lists(self, dec):
        """Host views of the decode: (decoded id lists, frame lengths, decoded lengths, phoneme table) - the step's only
        device->host transfers.  The status words of the cooperative BiLSTM kernels (csrc/lstm.hip: every cross-workgroup wait
        is bounded and raises a status word on timeout) ride in the same transfer as the lengths: a timed-out wait means
        incomplete hidden states, so it raises here instead of returning a silently wrong `tvs_pred`."""
        ids, nlen, frame_lens, given = dec
        B = int(frame_lens.numel())
        status = ops.lstm_status_words(ids.device)
        parts = [frame_lens.reshape(-1).to(torch.int32), nlen.reshape(-1).to(torch.int32)] + ([status] if status is not None else [])
        host = torch.cat(parts).cpu().tolist()                        # ONE transfer: frame lengths | decoded lengths | LSTM status
        fl, n = [int(v) for v in host[:B]], [int(v) for v in host[B:2 * B]]
        ops.lstm_check(host[2 * B:], ids.device)
        table = ids.cpu().numpy()
        if given is None:
            # models/force_aptai.py:111 (checked once the lengths are on the host; the device decode filled 60 slots at most)
            assert all(v < self.max_phn_seq_len for v in n), 'Need longer max phoneme sequence length.'
            given = [table[b, :n[b]].astype(np.int64) for b in range(len(n))]
        return given, fl, n, table

    def forward(self, epoch, audio_inputs, audio_lengths, phoneme_labels, phn_frames_49hz, LA, LP, JA, TTCL, TTCD, TMCL, TMCD,
                TBCL, TBCD, _phn_pred_list=None, _ac_override=None, _prefetch_next=None):
        """`_prefetch_next = (audio_inputs, audio_lengths)` of the batch the NEXT call will receive (the same tensor objects)
        starts its frozen-encoder pass on a side stream beside this call's heads (see prefetch)."""
        tv_targets = torch.stack([LA, LP, JA, TTCL, TTCD, TMCL, TMCD, TBCL, TBCD], dim=-1).float()
        res, g, dec = self._run(audio_inputs, audio_lengths, tv_targets, _phn_pred_list, _ac_override, _prefetch_next)
        loss, tv_loss, align_loss, tvs, frame_phns = res[:5]
        phn_pred_list, frame_seq_lens, _, _ = self._lists(dec)
        fp = frame_phns.cpu().numpy()                                  # ONE transfer (reference: B*T .cpu() calls)
        pred_frame_phns = [fp[b, :frame_seq_lens[b]].tolist() for b in range(g.B)]
        return {'loss': loss, 'tv_loss': tv_loss, 'align_loss': align_loss, 'tvs_pred': tvs,
                'pred_frame_phns': pred_frame_phns, 'pred_ctc_phn_seq': phn_pred_list}

    def load_state_dict(self, *args, **kwargs):
        # the captured encoder passes of prefetch() hold the addresses of the recogniser's bf16 weight copies, which are rebuilt
        # when the parameters change: drop the graphs (they are re-captured on the next prefetches)
        self._enc_graphs, self._enc_seen, self._prefetched = {}, {}, None
        return super().load_state_dict(*args, **kwargs)

    def set_encoder_precision(self, precision: str = "bf16"):
        """"bf16_f32res" (default here): bf16 GEMMs, fp32 residual stream; "bf16": all-bf16 stream; "mxfp8": the frozen recogniser's
        transformer Linear layers run with MX block-scaled FP8 operands (BASELINE configs[4]).  The heads stay fp32.  See
        Wav2Vec2Model.set_encoder_precision."""
        self.w2v2_pr.wav2vec2.set_encoder_precision(precision)
        return self

    def get_config(self):
        return {'pr_model_path': self.pr_model_path, 'w2v2_pr_cfg': self.w2v2_pr_cfg, 'device': self.device, 'vocab': self.vocab}

    def _wav(self, wav):
        device = next(self.parameters()).device
        if type(wav) is torch.Tensor:
            wav = wav[0]
        return (torch.unsqueeze(torch.Tensor(wav), dim=0).to(device),
                torch.unsqueeze(torch.LongTensor([len(wav)]), dim=0).to(device))

    def get_alignment(self, wav):
        """models/force_aptai.py:188-236: (N x T) log-attention of the decoded phonemes."""
        self.eval()
        with torch.no_grad():
            wav_input, wav_len = self._wav(wav)
            res, g, dec = self._run(wav_input, wav_len.reshape(-1))
            _, frame_seq_lens, phn_seq_lens, _ = self._lists(dec)
            att = res[5].view(g.B, g.Tp, _NPHN)[0]
            return {'alignment': att[0:frame_seq_lens[0], 0:phn_seq_lens[0]].permute(1, 0).cpu().numpy()}

    def get_faptai_output(self, wav):
        """models/force_aptai.py:238-322."""
        self.eval()
        with torch.no_grad():
            wav_input, wav_len = self._wav(wav)
            res, g, dec = self._run(wav_input, wav_len.reshape(-1))
            phn_pred_list, frame_seq_lens, _, table = self._lists(dec)
            tvs_out = res[3].squeeze(dim=0).cpu().numpy()
            tvs_pred_dict = {n: [row[i] for row in tvs_out] for i, n in enumerate(TV_NAMES)}
            align = res[8].view(g.B, g.Tp)[0, :g.T].cpu().numpy()
            return {'tvs_pred': tvs_pred_dict, 'pred_frame_phns': [int(table[0][a]) for a in align],
                    'pred_ctc_phn_seq': phn_pred_list, 'hidden_alignment': res[6].view(g.B, g.Tp, -1)[:, :g.T],
                    'hidden_tvs': res[7].view(g.B, g.Tp, -1)[:, :g.T]}
