"""Model configuration for the wav2vec2 encoder used by APTAI's hot path.

The reference builds a HuggingFace ``Wav2Vec2Config`` (train/train_aptai.py:336-340,
train/train_phoneme_recognizer.py:336-347) and hands it to the model constructors as
``pretrain_cfg``.  ``W2V2Config`` carries the subset of fields the hot path reads
(HF configuration_wav2vec2.py:166-188 for the defaults) and ``W2V2Config.from_any``
accepts an HF config object, a dict, or another ``W2V2Config`` so the constructors stay
drop-in.  Nothing here imports ``transformers``.
"""
from __future__ import annotations

import dataclasses
import io
import json
import os
import pickle
from typing import Any, Tuple


@dataclasses.dataclass
class W2V2Config:
    # transformer
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    hidden_act: str = "gelu"
    layer_norm_eps: float = 1e-5
    do_stable_layer_norm: bool = False          # False: post-LN (base); True: pre-LN (large)
    # conv feature encoder
    feat_extract_norm: str = "group"            # "group" (base) | "layer" (large)
    feat_extract_activation: str = "gelu"
    conv_dim: Tuple[int, ...] = (512, 512, 512, 512, 512, 512, 512)
    conv_stride: Tuple[int, ...] = (5, 2, 2, 2, 2, 2, 2)
    conv_kernel: Tuple[int, ...] = (10, 3, 3, 3, 3, 2, 2)
    conv_bias: bool = False
    num_conv_pos_embeddings: int = 128
    num_conv_pos_embedding_groups: int = 16
    # regularisers (train mode only)
    hidden_dropout: float = 0.1
    activation_dropout: float = 0.1
    attention_dropout: float = 0.1
    feat_proj_dropout: float = 0.0
    final_dropout: float = 0.1
    layerdrop: float = 0.1
    apply_spec_augment: bool = True
    mask_time_prob: float = 0.05
    mask_time_length: int = 10
    mask_time_min_masks: int = 2
    mask_feature_prob: float = 0.0
    mask_feature_length: int = 10
    mask_feature_min_masks: int = 0
    # CTC head (Wav2Vec2_PR)
    vocab_size: int = 32
    ctc_loss_reduction: str = "sum"
    ctc_zero_infinity: bool = False
    blank: int = 0                              # non-HF key added at train_phoneme_recognizer.py:342

    # ------------------------------------------------------------------ helpers
    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def num_feat_extract_layers(self) -> int:
        return len(self.conv_dim)

    def to_dict(self) -> dict:
        d = dataclasses.asdict(self)
        for k in ("conv_dim", "conv_stride", "conv_kernel"):
            d[k] = list(d[k])
        return d

    @classmethod
    def from_any(cls, cfg: Any) -> "W2V2Config":
        if isinstance(cfg, cls):
            return dataclasses.replace(cfg)
        names = {f.name for f in dataclasses.fields(cls)}
        if isinstance(cfg, dict):
            src = cfg
        else:  # HF PretrainedConfig or any attribute bag
            src = {k: getattr(cfg, k) for k in names if hasattr(cfg, k)}
        kw = {k: v for k, v in src.items() if k in names}
        for k in ("conv_dim", "conv_stride", "conv_kernel"):
            if k in kw:
                kw[k] = tuple(int(x) for x in kw[k])
        out = cls(**kw)
        out.validate()
        return out

    @classmethod
    def from_json_file(cls, path: str) -> "W2V2Config":
        with open(path) as f:
            return cls.from_any(json.load(f))

    @classmethod
    def from_pretrained_dir(cls, path: str) -> "W2V2Config":
        return cls.from_json_file(os.path.join(path, "config.json"))

    def validate(self) -> None:
        if self.hidden_size % self.num_attention_heads:
            raise ValueError("hidden_size must be divisible by num_attention_heads")
        if self.feat_extract_norm not in ("group", "layer"):
            raise ValueError(
                f"`config.feat_extract_norm` is {self.feat_extract_norm}, but has to be one of ['group', 'layer']")
        if not (len(self.conv_dim) == len(self.conv_stride) == len(self.conv_kernel)):
            raise ValueError("conv_dim / conv_stride / conv_kernel must have equal length")
        if self.hidden_act != "gelu" or self.feat_extract_activation != "gelu":
            raise ValueError("only the exact-erf 'gelu' activation of the reference configs is built")

    # canonical shapes -------------------------------------------------------
    @classmethod
    def base(cls, **kw) -> "W2V2Config":
        """facebook/wav2vec2-base shape (HF defaults)."""
        return cls(**kw)

    @classmethod
    def large(cls, **kw) -> "W2V2Config":
        """wav2vec2-large(-robust / xlsr) shape: the one APTAI hard-codes (models/aptai.py:46,54,81)."""
        d = dict(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
                 feat_extract_norm="layer", conv_bias=True, do_stable_layer_norm=True)
        d.update(kw)
        return cls(**d)


# ---------------------------------------------------------------------------------------- model_cfg.pkl
class _AttrBag:
    """Stand-in for a pickled configuration OBJECT (the reference pickles a ``transformers.Wav2Vec2Config`` inside
    ``get_config()``, train/train_phoneme_recognizer.py:472-473): receives the instance ``__dict__`` and nothing else."""

    def __setstate__(self, state):
        if isinstance(state, tuple) and len(state) == 2:          # (dict, slots)
            state = {**(state[0] or {}), **(state[1] or {})}
        self.__dict__.update(state or {})


_SAFE_GLOBALS = {
    ("builtins", n): getattr(__import__("builtins"), n)
    for n in ("dict", "list", "tuple", "set", "frozenset", "str", "int", "float", "bool", "bytes", "complex", "object")
}
_SAFE_GLOBALS[("collections", "OrderedDict")] = __import__("collections").OrderedDict


class _ConfigUnpickler(pickle.Unpickler):
    """Unpickler for ``model_cfg.pkl``: containers of plain values, ``torch.device``, and configuration objects, which are
    rebuilt as attribute bags WITHOUT importing or calling their class.  Every other global raises - a checkpoint directory
    cannot make this process run code."""

    def find_class(self, module, name):
        if (module, name) in _SAFE_GLOBALS:
            return _SAFE_GLOBALS[(module, name)]
        if (module, name) == ("copyreg", "_reconstructor"):
            return lambda cls, base, state: cls.__new__(cls)
        if (module, name) == ("torch", "device"):
            import torch
            return torch.device
        if name.endswith("Config") and (module.startswith("transformers.") or module.startswith("aptai_amd.")):
            if module.startswith("aptai_amd.") and name == "W2V2Config":
                return W2V2Config
            return _AttrBag
        if (module, name) == ("argparse", "Namespace"):
            return _AttrBag
        raise pickle.UnpicklingError(f"model_cfg.pkl: global {module}.{name} is not allowed (plain values, torch.device and "
                                     f"configuration objects only)")


def load_model_cfg(path: str) -> dict:
    """The ``get_config()`` dict the training scripts pickle next to ``pytorch_model.bin`` (models/force_aptai.py:61-64)."""
    with open(path, "rb") as f:
        obj = _ConfigUnpickler(io.BytesIO(f.read())).load()
    if not isinstance(obj, dict):
        raise pickle.UnpicklingError(f"{path}: expected the get_config() dict, found {type(obj).__name__}")
    return obj
