"""ORACLE (test infrastructure): plain-torch CPU restatement of ``transformers.Wav2Vec2Model.forward``
as the reference calls it (models/aptai.py:75-81, models/w2v2_pr.py:47-52,132-137).

"HF:n" = transformers/models/wav2vec2/modeling_wav2vec2.py line n (5.15.0; un-vendored, unpinned).
Functional style: parameters come from a ``state_dict``-like mapping with the HF key names under
``prefix`` (e.g. ``"wav2vec2."``).  Works under autograd, so the same code gives reference gradients.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F


def feat_extract_output_lengths(lengths: torch.Tensor, cfg) -> torch.Tensor:
    """HF:997-1016."""
    for k, s in zip(cfg.conv_kernel, cfg.conv_stride):
        lengths = torch.div(lengths - k, s, rounding_mode="floor") + 1
    return lengths


def feature_encoder(sd, cfg, input_values: torch.Tensor, prefix: str) -> torch.Tensor:
    """HF:382-419 with the three conv-layer variants HF:254-323.  (B,S) -> (B,C,T)."""
    h = input_values[:, None]
    for i in range(len(cfg.conv_dim)):
        p = f"{prefix}feature_extractor.conv_layers.{i}."
        h = F.conv1d(h, sd[p + "conv.weight"], sd.get(p + "conv.bias"), stride=cfg.conv_stride[i])
        if cfg.feat_extract_norm == "group":
            if i == 0:                                  # GroupNorm(C groups of 1 channel), HF:317-323
                h = F.group_norm(h, h.shape[1], sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], 1e-5)
        else:                                           # LayerNorm over channels, HF:288-299
            h = h.transpose(-2, -1)
            h = F.layer_norm(h, (h.shape[-1],), sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], 1e-5)
            h = h.transpose(-2, -1)
        h = F.gelu(h)
    return h


def pos_conv_weight(sd, prefix: str) -> torch.Tensor:
    """weight_norm(dim=2): w = g * v / ||v||, norm over dims (0,1) per tap — HF:340-356."""
    p = f"{prefix}encoder.pos_conv_embed.conv."
    g = sd[p + "parametrizations.weight.original0"]
    v = sd[p + "parametrizations.weight.original1"]
    norm = v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()
    return v * (g / norm)


def pos_conv_embed(sd, cfg, h: torch.Tensor, prefix: str) -> torch.Tensor:
    """HF:358-379: grouped conv k=128 pad=64, drop the last frame for even k, GELU.  (B,T,H)->(B,T,H)."""
    p = f"{prefix}encoder.pos_conv_embed.conv."
    k = cfg.num_conv_pos_embeddings
    x = h.transpose(1, 2)
    x = F.conv1d(x, pos_conv_weight(sd, prefix), sd[p + "bias"], padding=k // 2,
                 groups=cfg.num_conv_pos_embedding_groups)
    if k % 2 == 0:
        x = x[:, :, :-1]
    return F.gelu(x).transpose(1, 2)


def _ln(sd, name: str, x: torch.Tensor, eps: float) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps)


def attention(sd, cfg, x: torch.Tensor, key_mask: Optional[torch.Tensor], p: str,
              training: bool, drop: float) -> torch.Tensor:
    """HF:438-548 (eager math == the default sdpa).  ``key_mask`` (B,T) True = attend."""
    B, T, H = x.shape
    nh = cfg.num_attention_heads
    d = H // nh
    q = F.linear(x, sd[p + "q_proj.weight"], sd[p + "q_proj.bias"]).view(B, T, nh, d).transpose(1, 2)
    k = F.linear(x, sd[p + "k_proj.weight"], sd[p + "k_proj.bias"]).view(B, T, nh, d).transpose(1, 2)
    v = F.linear(x, sd[p + "v_proj.weight"], sd[p + "v_proj.bias"]).view(B, T, nh, d).transpose(1, 2)
    w = torch.matmul(q, k.transpose(2, 3)) * (d ** -0.5)
    if key_mask is not None:
        w = w.masked_fill(~key_mask[:, None, None, :], torch.finfo(w.dtype).min)
    w = F.softmax(w, dim=-1)
    w = F.dropout(w, p=drop, training=training)
    o = torch.matmul(w, v).transpose(1, 2).reshape(B, T, H)
    return F.linear(o, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])


def feed_forward(sd, cfg, x, p: str, training: bool) -> torch.Tensor:
    """HF:551-572."""
    h = F.linear(x, sd[p + "intermediate_dense.weight"], sd[p + "intermediate_dense.bias"])
    h = F.dropout(F.gelu(h), p=cfg.activation_dropout, training=training)
    h = F.linear(h, sd[p + "output_dense.weight"], sd[p + "output_dense.bias"])
    return F.dropout(h, p=cfg.hidden_dropout, training=training)


def encoder_layer(sd, cfg, x, key_mask, p: str, training: bool) -> torch.Tensor:
    eps = cfg.layer_norm_eps
    if cfg.do_stable_layer_norm:                       # pre-LN, HF:622-644
        res = x
        h = _ln(sd, p + "layer_norm", x, eps)
        h = attention(sd, cfg, h, key_mask, p + "attention.", training, cfg.attention_dropout)
        h = F.dropout(h, p=cfg.hidden_dropout, training=training)
        x = res + h
        return x + feed_forward(sd, cfg, _ln(sd, p + "final_layer_norm", x, eps), p + "feed_forward.", training)
    res = x                                             # post-LN, HF:587-601
    h = attention(sd, cfg, x, key_mask, p + "attention.", training, cfg.attention_dropout)
    h = F.dropout(h, p=cfg.hidden_dropout, training=training)
    x = _ln(sd, p + "layer_norm", res + h, eps)
    x = x + feed_forward(sd, cfg, x, p + "feed_forward.", training)
    return _ln(sd, p + "final_layer_norm", x, eps)


def wav2vec2_forward(sd, cfg, input_values: torch.Tensor, lengths: Optional[torch.Tensor],
                     prefix: str = "wav2vec2.", training: bool = False,
                     mask_time_indices: Optional[torch.Tensor] = None,
                     layer_keep: Optional[List[bool]] = None) -> Dict[str, object]:
    """HF:1319-1375.  ``lengths`` is the (B,) sample-count tensor the reference passes as the
    (B,1) "attention_mask" (models/aptai.py:77; HF:1023 turns it back into lengths).

    ``mask_time_indices`` (B,T) bool applies SpecAugment (HF:1292-1295) — the caller samples it.
    ``layer_keep[i] = False`` drops layer i (LayerDrop, HF:701-703 / 774-776) — caller flips the coins.
    Dropout uses torch's RNG when ``training`` and the probabilities are non-zero.
    """
    eps = cfg.layer_norm_eps
    feats = feature_encoder(sd, cfg, input_values, prefix).transpose(1, 2)        # (B,T,C)
    B, T, _ = feats.shape
    frame_mask = None
    if lengths is not None:
        fl = feat_extract_output_lengths(lengths.to(torch.long), cfg)
        frame_mask = torch.arange(T)[None, :] < fl[:, None]                        # HF:1018-1036
    # feature projection HF:422-434
    normed = _ln(sd, prefix + "feature_projection.layer_norm", feats, eps)
    h = F.linear(normed, sd[prefix + "feature_projection.projection.weight"],
                 sd[prefix + "feature_projection.projection.bias"])
    h = F.dropout(h, p=cfg.feat_proj_dropout, training=training)
    # SpecAugment HF:1272-1316 (time axis only; mask_feature_prob is 0 in every reference config)
    if mask_time_indices is not None and getattr(cfg, "apply_spec_augment", True):
        h = torch.where(mask_time_indices[:, :, None], sd[prefix + "masked_spec_embed"].to(h.dtype), h)
    # encoder HF:657-726 / 729-802
    key_mask = None
    if frame_mask is not None:
        h = h * frame_mask[:, :, None].to(h.dtype)                                 # HF:678-681
        if not bool(frame_mask.all()):
            key_mask = frame_mask
    h = h + pos_conv_embed(sd, cfg, h, prefix)
    if not cfg.do_stable_layer_norm:
        h = _ln(sd, prefix + "encoder.layer_norm", h, eps)
    h = F.dropout(h, p=cfg.hidden_dropout, training=training)
    hidden_states = []
    for i in range(cfg.num_hidden_layers):
        hidden_states.append(h)
        if layer_keep is not None and not layer_keep[i]:
            continue
        h = encoder_layer(sd, cfg, h, key_mask, f"{prefix}encoder.layers.{i}.", training)
    if cfg.do_stable_layer_norm:
        h = _ln(sd, prefix + "encoder.layer_norm", h, eps)
    hidden_states.append(h)
    return {"last_hidden_state": h, "hidden_states": hidden_states, "extract_features": normed,
            "frame_mask": frame_mask}
