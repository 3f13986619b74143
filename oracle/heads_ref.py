"""ORACLE (test infrastructure): CPU restatement of the reference's task heads and model forwards.

models/aptai.py (APTAI), models/w2v2_pr.py (Wav2Vec2_PR), models/force_aptai.py (Force_APTAI) and
models/modules.py (LowPassFilterLayer, ForwardSumLoss, CrossAttention, RNN, PositionalEncoding).
Plain torch/numpy on CPU; differentiable, so reference gradients come from autograd over this code.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from . import w2v2_ref


# ----------------------------------------------------------------------------- M1 low-pass FIR
def lowpass_taps(cutoff: float = 10, sampling_rate: float = 49) -> torch.Tensor:
    """models/modules.py:27-44: N=51 Hann-windowed sinc, unit DC gain, float64."""
    fc = cutoff / sampling_rate
    b = 0.08
    N = int(np.ceil(4 / b))
    if not N % 2:
        N += 1
    n = np.arange(N)
    h = np.sinc(fc * 2 * (n - (N - 1) / 2))
    w = 0.5 * (1 - np.cos(n * 2 * math.pi / (N - 1)))
    h = h * w
    return torch.tensor(h / np.sum(h))


def lowpass_filter(y: torch.Tensor, taps: torch.Tensor) -> torch.Tensor:
    """models/modules.py:46-61: per-channel 'same' cross-correlation in float64, result cast to f32."""
    B, L, C = y.shape
    yd = y.double().permute(0, 2, 1).reshape(B * C, 1, L)
    out = F.conv1d(yd, taps.view(1, 1, -1).double(), padding="same")
    return out.view(B, C, L).permute(0, 2, 1).float()


# ----------------------------------------------------------------------------- A2 APTAI.forward
def aptai_forward(sd, cfg, audio_inputs, audio_lengths, phn_frames_49hz, tv_targets_list,
                  training: bool = False, mask_time_indices=None, layer_keep=None,
                  tv_drop: float = 0.1, phn_drop: float = 0.1) -> Dict[str, torch.Tensor]:
    """models/aptai.py:58-115.  ``tv_targets_list`` = [LA, LP, JA, TTCL, TTCD, TMCL, TMCD, TBCL, TBCD]."""
    tv_targets = torch.stack(list(tv_targets_list), dim=-1).float()
    tv_pad_mask = tv_targets != -100.0
    phn_pad_mask = phn_frames_49hz != 0
    out = w2v2_ref.wav2vec2_forward(sd, cfg, audio_inputs, audio_lengths, "wav2vec2.", training,
                                    mask_time_indices, layer_keep)
    # models/aptai.py:81 reads hidden_states[24] == the last entry for a 24-layer backbone
    h = out["hidden_states"][cfg.num_hidden_layers]
    tv = F.linear(torch.tanh(F.dropout(h, tv_drop, training)), sd["tv_head.2.weight"], sd["tv_head.2.bias"])
    tv = lowpass_filter(tv, sd["tv_lowpass.lowpass.weight"].view(-1))
    logits = F.linear(F.leaky_relu(F.dropout(h, phn_drop, training)), sd["phn_head.2.weight"],
                      sd["phn_head.2.bias"])
    mse = F.mse_loss(tv[tv_pad_mask], tv_targets[tv_pad_mask], reduction="mean")
    flat = phn_pad_mask.flatten()
    ce = F.cross_entropy(logits.view(-1, logits.size(2))[flat], phn_frames_49hz.flatten()[flat],
                         ignore_index=0, reduction="mean")
    loss = 0.5 * mse + 0.5 * ce
    pred = torch.argmax(F.softmax(logits, dim=-1), dim=-1)
    return {"loss": loss, "mse_loss": mse, "ce_loss": ce, "tvs_pred": tv, "phn_fc_pred": pred,
            "phn_logits": logits, "last_hidden": h, "hidden_states": out["hidden_states"]}


# ----------------------------------------------------------------------------- P2 CTC (alpha recursion)
def ctc_loss_ref(log_probs: torch.Tensor, targets: torch.Tensor, input_lengths, target_lengths,
                 blank: int = 0, reduction: str = "mean", zero_infinity: bool = True) -> torch.Tensor:
    """The CTC negative log-likelihood ``F.ctc_loss`` computes for models/w2v2_pr.py:73-81 and
    models/modules.py:110-113 (Graves et al. 2006 alpha recursion in log space).

    log_probs (T,B,V) log-softmaxed; targets (B,Lmax) padded arbitrarily beyond target_lengths.
    reduction 'mean' = mean_b( nll_b / max(target_len_b, 1) ); 'sum' = sum_b nll_b.
    """
    T, B, V = log_probs.shape
    input_lengths = [int(x) for x in input_lengths]
    target_lengths = [int(x) for x in target_lengths]
    NEG = -1e30                     # "log 0": finite, so autograd through logsumexp never sees inf - inf
    losses = []
    for b in range(B):
        L = target_lengths[b]
        Tb = input_lengths[b]
        tgt = targets[b, :L].to(torch.long)
        S = 2 * L + 1
        ext = torch.full((S,), blank, dtype=torch.long)
        ext[1::2] = tgt
        lp = log_probs[:Tb, b, :][:, ext]                            # (Tb, S)
        # allowed skip s-2 -> s: ext[s] != blank and ext[s] != ext[s-2]
        skip = torch.zeros(S, dtype=torch.bool)
        if S > 2:
            skip[2:] = (ext[2:] != blank) & (ext[2:] != ext[:-2])
        neg = lp.new_full((S,), NEG)
        if Tb == 0:
            ll = lp.new_tensor(0.0 if L == 0 else NEG)
        else:
            mask0 = torch.zeros(S, dtype=torch.bool)
            mask0[:2] = True
            alpha = torch.where(mask0, lp[0], neg)
            for t in range(1, Tb):
                a1 = torch.cat([neg[:1], alpha[:-1]])[:S]
                a2 = torch.where(skip, torch.cat([neg[:2], alpha[:-2]])[:S], neg)
                alpha = torch.logsumexp(torch.stack([alpha, a1, a2], dim=0), dim=0) + lp[t]
                alpha = torch.clamp(alpha, min=NEG)
            ll = torch.logsumexp(alpha[max(S - 2, 0):], dim=0)
        nll = -ll
        infeasible = bool(nll.detach() > 1e29)
        if infeasible:
            nll = nll * 0.0 if zero_infinity else nll * 0.0 + float("inf")
        losses.append(nll)
    losses = torch.stack(losses)
    if reduction == "mean":
        tl = torch.tensor(target_lengths, dtype=losses.dtype).clamp(min=1)
        return (losses / tl).mean()
    if reduction == "sum":
        return losses.sum()
    return losses


# ----------------------------------------------------------------------------- P1 Wav2Vec2_PR.forward
def pr_forward(sd, cfg, input_values, input_lengths, phoneme_labels, training: bool = False,
               mask_time_indices=None, layer_keep=None, prefix: str = "") -> Dict[str, torch.Tensor]:
    """models/w2v2_pr.py:40-88."""
    out = w2v2_ref.wav2vec2_forward(sd, cfg, input_values, input_lengths, prefix + "wav2vec2.", training,
                                    mask_time_indices, layer_keep)
    hidden = F.dropout(out["last_hidden_state"], cfg.final_dropout, training)
    state_lens = w2v2_ref.feat_extract_output_lengths(input_lengths.to(torch.long), cfg)
    logits = F.linear(hidden, sd[prefix + "pr_head.weight"], sd[prefix + "pr_head.bias"])
    log_probs = F.log_softmax(logits, dim=-1, dtype=torch.float32).transpose(0, 1)
    target_lengths = (phoneme_labels >= 0).sum(-1)
    loss = ctc_loss_ref(log_probs, phoneme_labels, state_lens, target_lengths, blank=getattr(cfg, "blank", 0),
                        reduction=cfg.ctc_loss_reduction, zero_infinity=cfg.ctc_zero_infinity)
    return {"loss": loss, "phoneme_logits": logits, "log_probs": log_probs, "hidden_states": hidden,
            "state_lens": state_lens}


def pr_get_embeddings(sd, cfg, audio_inputs, audio_lengths, prefix: str = "") -> Dict[str, object]:
    """models/w2v2_pr.py:124-167 without the (absent) beam decoder: eval-mode encoder + pr_head."""
    with torch.no_grad():
        out = w2v2_ref.wav2vec2_forward(sd, cfg, audio_inputs, audio_lengths, prefix + "wav2vec2.", False)
        h = out["last_hidden_state"]
        logits = F.linear(h, sd[prefix + "pr_head.weight"], sd[prefix + "pr_head.bias"])
        lens = w2v2_ref.feat_extract_output_lengths(audio_lengths.to(torch.long), cfg)
    return {"last_transf_hidden": h.permute(0, 2, 1), "phoneme_logits": logits, "frame_seq_lens": lens}


def pr_get_embeddings_grad(sd, cfg, audio_inputs, audio_lengths, intermediate_hidden: int, latter_hidden: int,
                           training: bool = False, prefix: str = "") -> Dict[str, object]:
    """models/w2v2_pr.py:91-122: encoder WITH autograd (no eval() / no_grad in the reference), the last / an intermediate /
    a latter hidden state transposed to (batch, feat, time) and `pr_head` on each; `features_hidden` is the separate
    `feature_extractor` pass of :93 (batch, 512, time)."""
    feats = w2v2_ref.feature_encoder(sd, cfg, audio_inputs, prefix + "wav2vec2.")
    out = w2v2_ref.wav2vec2_forward(sd, cfg, audio_inputs, audio_lengths, prefix + "wav2vec2.", training)
    W, b = sd[prefix + "pr_head.weight"], sd[prefix + "pr_head.bias"]
    last, inter, latter = out["last_hidden_state"], out["hidden_states"][intermediate_hidden], out["hidden_states"][latter_hidden]
    return {"features_hidden": feats, "last_transf_hidden": last.permute(0, 2, 1),
            "phoneme_logits_last": F.linear(last, W, b), "phoneme_logits_inter": F.linear(inter, W, b),
            "phoneme_logits_latter": F.linear(latter, W, b), "intermediate_hidden": inter.permute(0, 2, 1),
            "latter_hidden": latter.permute(0, 2, 1)}


def ctc_best_path(logits: np.ndarray, blank: int = 0) -> np.ndarray:
    """Greedy stand-in for the torchaudio beam decoder (parity unpinned):
    argmax per frame -> collapse repeats -> drop blank.  NOTE: like the reference's call
    (models/w2v2_pr.py:155) it runs over ALL T frames of the padded batch, not ``frame_seq_lens``."""
    ids = np.asarray(logits).argmax(-1)
    keep = np.ones(len(ids), dtype=bool)
    keep[1:] = ids[1:] != ids[:-1]
    ids = ids[keep]
    return ids[ids != blank].astype(np.int64)


# ----------------------------------------------------------------------------- F2..F5 Force_APTAI pieces
def positional_encoding(d_model: int = 128, max_len: int = 60) -> torch.Tensor:
    """models/modules.py:222-228 -> (max_len, 1, d_model)."""
    # angle[t][i] = t * 10000^(-2i/d) in fp32 like the reference; sin in the even columns, cos in the odd ones
    freq = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * (-math.log(10000.0) / d_model))
    angle = torch.arange(max_len, dtype=torch.float32)[:, None] * freq[None, :]
    pe = torch.stack((angle.sin(), angle.cos()), dim=-1).reshape(max_len, 1, d_model).contiguous()
    return pe


def cross_attention(sd, frame_hidden, phn_hidden, labels_att_mask, p: str = "xatt."):
    """models/modules.py:139-153 -> (att_out (B,T,256), energy (B,T,N))."""
    q = F.linear(frame_hidden, sd[p + "q.weight"], sd[p + "q.bias"])
    k = F.linear(phn_hidden, sd[p + "k.weight"], sd[p + "k.bias"])
    energy = torch.bmm(q, k.transpose(2, 1))
    att_mask = (1 - labels_att_mask) * -1000.0
    energy = energy + att_mask.unsqueeze(1).repeat(1, energy.size(1), 1)
    att = torch.softmax(energy, dim=-1)
    out = torch.cat([torch.bmm(att, k), q], dim=-1)
    out = F.layer_norm(out, (out.shape[-1],), sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], 1e-5)
    return out, energy


def forward_sum_loss(attn_logprob: torch.Tensor, text_lens: Sequence[int], mel_lens: Sequence[int],
                     blank_logprob: float = -1) -> torch.Tensor:
    """models/modules.py:77-117.  attn_logprob (B,1,T,N)."""
    pd = F.pad(attn_logprob, (1, 0, 0, 0, 0, 0, 0, 0), value=blank_logprob)
    total = 0.0
    for b in range(attn_logprob.shape[0]):
        n, t = int(text_lens[b]), int(mel_lens[b])
        target = torch.arange(1, n + 1).unsqueeze(0)
        cur = pd[b].permute(1, 0, 2)[:t, :, :n + 1]
        cur = F.log_softmax(cur, dim=-1)
        total = total + ctc_loss_ref(cur, target, [t], [n], blank=0, reduction="mean", zero_infinity=True)
    return total / attn_logprob.shape[0]


def lstm_direction(x, w_ih, w_hh, b_ih, b_hh, lens, reverse: bool):
    """One direction of nn.LSTM(batch_first) over padded x (B,T,D) with per-row lengths
    (packed-sequence semantics: outputs beyond a row's length are zero).  Gate order i,f,g,o."""
    B, T, _ = x.shape
    Hd = w_hh.shape[1]
    outs = [None] * T
    h = x.new_zeros(B, Hd)
    c = x.new_zeros(B, Hd)
    lens_t = torch.as_tensor(lens)
    steps = range(T - 1, -1, -1) if reverse else range(T)
    for t in steps:
        g = F.linear(x[:, t], w_ih, b_ih) + F.linear(h, w_hh, b_hh)
        i, f, gg, o = g.chunk(4, dim=-1)
        c_new = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h_new = torch.sigmoid(o) * torch.tanh(c_new)
        valid = (t < lens_t)[:, None]
        c = torch.where(valid, c_new, c)
        h = torch.where(valid, h_new, h)
        outs[t] = torch.where(valid, h_new, torch.zeros_like(h_new))
    return torch.stack(outs, dim=1)


def rnn_forward(sd, x, lens, p: str = "rnn.", training: bool = False, drop: float = 0.1):
    """models/modules.py:202-214 with the evident intent ``hidden_tvs = packed_output`` on the batch>1
    branch (the shipped line :207 raises NameError).  Returns (out (B,T,9), lstm_out (B,T,512))."""
    if x.shape[0] == 1:
        lens = [x.shape[1]]            # models/modules.py:209-212: the batch-1 branch runs the LSTM unpacked over ALL frames
    fw = lstm_direction(x, sd[p + "lstm.weight_ih_l0"], sd[p + "lstm.weight_hh_l0"],
                        sd[p + "lstm.bias_ih_l0"], sd[p + "lstm.bias_hh_l0"], lens, False)
    bw = lstm_direction(x, sd[p + "lstm.weight_ih_l0_reverse"], sd[p + "lstm.weight_hh_l0_reverse"],
                        sd[p + "lstm.bias_ih_l0_reverse"], sd[p + "lstm.bias_hh_l0_reverse"], lens, True)
    lstm_out = torch.cat([fw, bw], dim=-1)
    lstm_out = lstm_out[:, :int(max(lens))]
    h = F.linear(lstm_out, sd[p + "linear.0.weight"], sd[p + "linear.0.bias"])
    h = torch.tanh(F.dropout(h, drop, training))
    return F.linear(h, sd[p + "linear.3.weight"], sd[p + "linear.3.bias"]), lstm_out


# ----------------------------------------------------------------------------- F1 Force_APTAI.forward
def force_aptai_forward(sd, pr_cfg, audio_inputs, audio_lengths, tv_targets_list,
                        phn_pred_list: Optional[List[np.ndarray]] = None, training: bool = False,
                        max_phn_seq_len: int = 60) -> Dict[str, object]:
    """models/force_aptai.py:80-178.  ``phn_pred_list`` = decoded phoneme ids per utterance; when None the
    greedy best path is used (the reference's beam decoder is absent: parity unpinned)."""
    tv_targets = torch.stack(list(tv_targets_list), dim=-1).float()
    tv_pad_mask = tv_targets != -100.0
    emb = pr_get_embeddings(sd, pr_cfg, audio_inputs, audio_lengths, prefix="w2v2_pr.")
    ac = emb["last_transf_hidden"]                                             # (B,H,T)
    if phn_pred_list is None:
        phn_pred_list = [ctc_best_path(l.numpy()) for l in emb["phoneme_logits"]]
    frame_seq_lens = [int(x) for x in emb["frame_seq_lens"].tolist()]
    phn_seq_lens = [len(l) for l in phn_pred_list]
    padded = []
    for lst in phn_pred_list:
        assert len(lst) < max_phn_seq_len, 'Need longer max phoneme sequence length.'
        padded.append(np.pad(lst, (0, max_phn_seq_len - len(lst)), mode='constant'))
    phn_pred_seq = torch.tensor(np.array(padded), dtype=torch.int32)
    phn_pred_mask = (phn_pred_seq != 0).to(torch.int)
    phn_embs = F.embedding(phn_pred_seq, sd["phn_emb_layer.weight"], padding_idx=0)
    pe = sd["pe_phn.pe"]
    phn_embs = (phn_embs.permute(1, 0, 2) + pe[:phn_embs.size(1)]).permute(1, 0, 2)
    phn_embs = F.dropout(phn_embs, 0.2, training)
    frame_hidden = F.linear(ac.permute(0, 2, 1), sd["frame_lin.weight"], sd["frame_lin.bias"])
    frame_hidden = F.dropout(frame_hidden, 0.2, training)
    att_out, energy = cross_attention(sd, frame_hidden, phn_embs, phn_pred_mask)
    att_mask = ((1 - phn_pred_mask) * -1000.0).unsqueeze(1).repeat(1, energy.size(1), 1)
    att = torch.log_softmax(energy + att_mask, dim=-1)
    rnn_out, lstm_out = rnn_forward(sd, att_out, frame_seq_lens, training=training)
    tvs = lowpass_filter(rnn_out, sd["tv_lowpass.lowpass.weight"].view(-1))
    tv_loss = F.mse_loss(tvs[tv_pad_mask], tv_targets[tv_pad_mask], reduction="mean")
    align_loss = forward_sum_loss(att.unsqueeze(1), phn_seq_lens, frame_seq_lens)
    loss = 0.4 * tv_loss + 0.6 * align_loss
    align_out = torch.max(att, axis=2)[1]
    pred_frame_phns = []
    for b in range(att.shape[0]):
        idx = align_out[b, :frame_seq_lens[b]]
        pred_frame_phns.append([int(v) for v in phn_pred_seq[b][idx]])
    return {"loss": loss, "tv_loss": tv_loss, "align_loss": align_loss, "tvs_pred": tvs,
            "pred_frame_phns": pred_frame_phns, "pred_ctc_phn_seq": phn_pred_list,
            "att": att, "energy": energy, "att_out": att_out, "align_idx": align_out, "lstm_out": lstm_out}
