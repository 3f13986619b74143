"""ORACLE (test infrastructure): deterministic synthetic weights and batches.

Pretrained checkpoints are unreachable offline, so every parity case uses weights
generated from (tensor name, shape, seed) alone: the golden generator loads them into the reference
modules with ``load_state_dict`` and the tests regenerate the identical tensors.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict, List, Tuple

import numpy as np
import torch

from .heads_ref import lowpass_taps, positional_encoding


def synth_tensor(name: str, shape, seed: int = 0) -> torch.Tensor:
    """Value of parameter ``name`` — a pure function of (name, shape, seed)."""
    g = torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    shape = tuple(int(s) for s in shape)
    leaf = name.rsplit(".", 1)[-1]
    if name.endswith("masked_spec_embed"):
        return torch.rand(shape, generator=g)
    if name.endswith("parametrizations.weight.original0"):          # weight-norm gain g (1,1,k)
        return 0.5 + torch.rand(shape, generator=g)
    if name.endswith("parametrizations.weight.original1"):          # weight-norm direction v
        fan_in = shape[1] * shape[2]
        return torch.randn(shape, generator=g) * (2.0 / fan_in) ** 0.5
    is_norm = ("layer_norm" in name) or name.endswith("layer_norm.weight")
    if len(shape) == 1:
        if is_norm and leaf == "weight":
            return 1.0 + 0.1 * torch.randn(shape, generator=g)
        return 0.05 * torch.randn(shape, generator=g)               # biases, norm betas
    if "phn_emb_layer" in name:
        w = torch.randn(shape, generator=g)
        w[0] = 0.0                                                    # padding_idx=0 row
        return w
    fan_in = int(np.prod(shape[1:]))
    gain = 1.4 if ("conv_layers" in name) else 1.0                   # keep conv-stack activations O(1)
    return torch.randn(shape, generator=g) * gain / fan_in ** 0.5


# ----------------------------------------------------------------------------- key / shape tables
def w2v2_param_shapes(cfg, prefix: str = "wav2vec2.") -> "OrderedDict[str, Tuple[int, ...]]":
    """HF ``Wav2Vec2Model`` state-dict keys and shapes."""
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    H, I = cfg.hidden_size, cfg.intermediate_size
    if cfg.mask_time_prob > 0.0 or cfg.mask_feature_prob > 0.0:
        s[prefix + "masked_spec_embed"] = (H,)
    cin = 1
    for i, (c, k) in enumerate(zip(cfg.conv_dim, cfg.conv_kernel)):
        p = f"{prefix}feature_extractor.conv_layers.{i}."
        s[p + "conv.weight"] = (c, cin, k)
        if cfg.conv_bias:
            s[p + "conv.bias"] = (c,)
        if cfg.feat_extract_norm == "layer" or i == 0:
            s[p + "layer_norm.weight"] = (c,)
            s[p + "layer_norm.bias"] = (c,)
        cin = c
    p = prefix + "feature_projection."
    s[p + "layer_norm.weight"] = (cin,)
    s[p + "layer_norm.bias"] = (cin,)
    s[p + "projection.weight"] = (H, cin)
    s[p + "projection.bias"] = (H,)
    p = prefix + "encoder.pos_conv_embed.conv."
    k, g = cfg.num_conv_pos_embeddings, cfg.num_conv_pos_embedding_groups
    s[p + "bias"] = (H,)
    s[p + "parametrizations.weight.original0"] = (1, 1, k)
    s[p + "parametrizations.weight.original1"] = (H, H // g, k)
    s[prefix + "encoder.layer_norm.weight"] = (H,)
    s[prefix + "encoder.layer_norm.bias"] = (H,)
    for i in range(cfg.num_hidden_layers):
        p = f"{prefix}encoder.layers.{i}."
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            s[p + f"attention.{n}.weight"] = (H, H)
            s[p + f"attention.{n}.bias"] = (H,)
        s[p + "layer_norm.weight"] = (H,)
        s[p + "layer_norm.bias"] = (H,)
        s[p + "feed_forward.intermediate_dense.weight"] = (I, H)
        s[p + "feed_forward.intermediate_dense.bias"] = (I,)
        s[p + "feed_forward.output_dense.weight"] = (H, I)
        s[p + "feed_forward.output_dense.bias"] = (H,)
        s[p + "final_layer_norm.weight"] = (H,)
        s[p + "final_layer_norm.bias"] = (H,)
    return s


def aptai_param_shapes(cfg, n_tv: int = 9, n_phn: int = 46):
    s = w2v2_param_shapes(cfg, "wav2vec2.")
    H = cfg.hidden_size
    s["tv_head.2.weight"] = (n_tv, H)
    s["tv_head.2.bias"] = (n_tv,)
    s["tv_lowpass.lowpass.weight"] = (1, 1, 51)
    s["phn_head.2.weight"] = (n_phn, H)
    s["phn_head.2.bias"] = (n_phn,)
    return s


def pr_param_shapes(cfg, prefix: str = ""):
    s = w2v2_param_shapes(cfg, prefix + "wav2vec2.")
    s[prefix + "pr_head.weight"] = (cfg.vocab_size, cfg.hidden_size)
    s[prefix + "pr_head.bias"] = (cfg.vocab_size,)
    return s


def force_aptai_param_shapes(pr_cfg, vocab_len: int):
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    s["xatt.q.weight"] = (128, 128); s["xatt.q.bias"] = (128,)
    s["xatt.k.weight"] = (128, 128); s["xatt.k.bias"] = (128,)
    s["xatt.layer_norm.weight"] = (256,); s["xatt.layer_norm.bias"] = (256,)
    s["frame_lin.weight"] = (128, pr_cfg.hidden_size); s["frame_lin.bias"] = (128,)
    s["phn_emb_layer.weight"] = (vocab_len, 128)
    s["pe_phn.pe"] = (60, 1, 128)
    for sfx in ("", "_reverse"):
        s[f"rnn.lstm.weight_ih_l0{sfx}"] = (1024, 256)
        s[f"rnn.lstm.weight_hh_l0{sfx}"] = (1024, 256)
        s[f"rnn.lstm.bias_ih_l0{sfx}"] = (1024,)
        s[f"rnn.lstm.bias_hh_l0{sfx}"] = (1024,)
    s["rnn.linear.0.weight"] = (256, 512); s["rnn.linear.0.bias"] = (256,)
    s["rnn.linear.3.weight"] = (9, 256); s["rnn.linear.3.bias"] = (9,)
    s["tv_lowpass.lowpass.weight"] = (1, 1, 51)
    s.update(pr_param_shapes(pr_cfg, "w2v2_pr."))
    return s


def make_state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    sd = OrderedDict()
    for name, shape in shapes.items():
        if name.endswith("tv_lowpass.lowpass.weight"):
            sd[name] = lowpass_taps(10, 49).view(1, 1, -1)                       # f64, not synthetic
        elif name.endswith("pe_phn.pe"):
            sd[name] = positional_encoding(128, 60)
        else:
            sd[name] = synth_tensor(name, shape, seed)
    return sd


# ----------------------------------------------------------------------------- synthetic batches (§8d)
def synth_lengths(batch: int, n_samples: int, g: torch.Generator) -> torch.Tensor:
    """Half the batch full length, half uniform in [0.8 S, S]; the first row is always full so the padded
    width equals S (the reference pads to the longest utterance, train/train_aptai.py:274-278)."""
    lens = torch.full((batch,), n_samples, dtype=torch.long)
    for b in range(batch // 2, batch):
        lens[b] = int(torch.randint(int(0.8 * n_samples), n_samples + 1, (1,), generator=g))
    return lens


def synth_aptai_batch(cfg, batch: int, n_samples: int, seed: int = 1234, n_phn: int = 46, n_tv: int = 9):
    """Batch dict C0 for APTAI / Force_APTAI: N(0,1) audio zero-padded past each length, f64 TV tracks
    with -100.0 beyond each utterance's frames, frame labels in [1, V-1] with 0 padding."""
    from .w2v2_ref import feat_extract_output_lengths
    g = torch.Generator().manual_seed(seed)
    lens = synth_lengths(batch, n_samples, g)
    audio = torch.randn(batch, n_samples, generator=g)
    audio = audio * (torch.arange(n_samples)[None, :] < lens[:, None])
    T = int(feat_extract_output_lengths(torch.tensor(n_samples), cfg))
    fl = feat_extract_output_lengths(lens, cfg)
    valid = torch.arange(T)[None, :] < fl[:, None]
    phn = torch.randint(1, n_phn, (batch, T), generator=g) * valid
    names = ("LA", "LP", "JA", "TTCL", "TTCD", "TMCL", "TMCD", "TBCL", "TBCD")[:n_tv]
    if n_tv > 9:
        names = tuple(f"TV{i}" for i in range(n_tv))
    out = {"audio_inputs": audio, "audio_lengths": lens, "phn_frames_49hz": phn.long()}
    for n in names:
        tv = torch.randn(batch, T, generator=g, dtype=torch.float64)
        out[n] = torch.where(valid, tv, torch.full_like(tv, -100.0))
    return out


def synth_ctc_labels(batch: int, vocab: int, seed: int, lo: int = 20, hi: int = 55) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed + 77)
    lens = torch.randint(lo, hi + 1, (batch,), generator=g)
    lab = torch.full((batch, int(lens.max())), -100, dtype=torch.int32)
    for b in range(batch):
        lab[b, :lens[b]] = torch.randint(1, vocab, (int(lens[b]),), generator=g, dtype=torch.int32)
    return lab


def synth_pr_batch(cfg, batch: int, n_samples: int, seed: int = 1234, lo: int = 20, hi: int = 55):
    g = torch.Generator().manual_seed(seed)
    lens = synth_lengths(batch, n_samples, g)
    audio = torch.randn(batch, n_samples, generator=g)
    audio = audio * (torch.arange(n_samples)[None, :] < lens[:, None])
    return {"input_values": audio, "input_lengths": lens,
            "phoneme_labels": synth_ctc_labels(batch, cfg.vocab_size, seed, lo, hi)}
