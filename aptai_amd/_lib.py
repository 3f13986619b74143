"""ctypes loader for libaptai_hip.so (the C ABI in include/aptai_hip.h).

There is NO CPU fallback: every product op goes through this library and raises if it is missing
or if the call fails.  The library is built in-tree by ``__graft_entry__.build()`` /
``make -C aptai_amd/csrc`` (hipcc cross-compiles gfx950 without a GPU).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import threading

import torch  # noqa: F401  -- MUST precede the CDLL below: torch's bundled HIP runtime has to be the one this process
#                              initialises; loading libaptai_hip.so first pulls in a second runtime that sees no device

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# APTAI_HIP_LIB: A/B a second in-tree build of the same sources (tools/, never a different implementation)
LIB_PATH = os.environ.get("APTAI_HIP_LIB") or os.path.join(CSRC, "libaptai_hip.so")

_lib = None
_lock = threading.Lock()

_P, _I64, _I, _F, _U64 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_uint64
# argument types of every entry point except aptai_gemm_bf16 (descriptor struct, see ops.GemmDesc)
ARGTYPES = {
    "aptai_layernorm_fwd": [_P, _P, _P, _P, _P, _P, _I64, _I64, _F, _I, _P],
    "aptai_layernorm_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _F, _U64, _P, _P, _P, _I64, _I64, _P, _P],
    "aptai_layernorm_bwd_workspace_bytes": [_I64, _I64],
    "aptai_layernorm_bwd_finalize_multi": [_P, _I64, _I64, _P],
    "aptai_layernorm_fwd_f32in": [_P, _P, _P, _P, _P, _I64, _I64, _F, _P],
    "aptai_layernorm_fwd_f32in_split": [_P, _P, _P, _P, _P, _I, _I64, _I64, _F, _P],
    "aptai_attention_fwd": [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _F, _F, _U64, _I, _P],
    "aptai_attention_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _F, _F, _U64, _I, _I, _P],
    "aptai_cast_f32_to_bf16": [_P, _P, _I64, _I64, _I64, _P],
    "aptai_conv_weight_to_bf16": [_P, _P, _I64, _I64, _I64, _P],
    "aptai_posconv_weight": [_P, _P, _P, _P, _P, _I64, _I64, _I64, _P],
    "aptai_posconv_pack": [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _P],
    "aptai_frame_mask_fwd": [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _P],
    "aptai_frame_mask_bwd": [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _P],
    "aptai_frame_mask_bwd_workspace_bytes": [_I64, _I64, _I64],
    "aptai_colsum_bf16": [_P, _I64, _P, _P, _I64, _I64, _I, _P],
    "aptai_colsum_workspace_bytes": [_I64, _I64],
    "aptai_dropout_bf16": [_P, _P, _I64, _F, _U64, _P],
    "aptai_dgelu_bf16": [_P, _P, _P, _I64, _P],
    "aptai_conv0_fwd": [_P, _I64, _I64, _P, _P, _P, _P, _I, _F, _P, _I64, _I64, _I64, _I64, _I64, _P, _P, _P],
    "aptai_conv0_bwd": [_P, _I64, _I64, _P, _P, _P, _P, _I, _F, _P, _I64, _I64, _P, _P, _P, _P, _P, _P, _P],
    "aptai_conv0_bwd_workspace_bytes": [_I64, _I64],
    "aptai_conv0_workspace_bytes": [_I64, _I64],
    "aptai_head_act_fwd": [_P, _P, _P, _I64, _F, _F, _U64, _P],
    "aptai_head_act_bwd": [_P, _P, _P, _P, _I64, _F, _F, _U64, _P],
    "aptai_lowpass_fir": [_P, _I64, _I64, _P, _I64, _P, _I64, _I64, _I, _I64, _I64, _I64, _I64, _I64, _P],
    "aptai_aptai_loss_fwd": [_P, _P, _P, _I64, _I64, _P, _I64, _I64, _I64, _I64, _F, _F, _P, _P, _P, _P],
    "aptai_aptai_loss_bwd": [_P, _P, _P, _I64, _I64, _P, _I64, _I64, _I64, _I64, _F, _F, _P, _P, _P, _P, _I64, _P],
    "aptai_aptai_loss_workspace_bytes": [],
    "aptai_gemm_workspace_bytes": [_I64, _I64, _I],
    "aptai_gemm_sk_workspace_bytes": [],
    "aptai_gemm_sk_status": [_P, _P, _P],
    "aptai_cast_multi": [_P, _I64, _I64, _P],
    "aptai_posconv_weight_bwd": [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _P],
    "aptai_spec_augment_mask": [_P, _P, _I64, _I64, _F, _I64, _I64, _U64, _P],
    "aptai_posconv_wgrad": [_P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _P],
    "aptai_posconv_gemm": [_P, _I64, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _I, _P],
    "aptai_adam_multi": [_P, _P, _I64, _I64, _F, _F, _F, _F, _F, _P],
    "aptai_sgemm_f32": [_P, _I, _I64, _I64, _P, _I64, _I64, _P, _I64, _P, _F, _I, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _P, _P],
    "aptai_sgemm_workspace_bytes": [_I64, _I64, _I64, _I64],
    "aptai_colsum_f32_workspace_bytes": [_I64],
    "aptai_embed_pe_fwd": [_P, _P, _P, _P, _I64, _I64, _I64, _F, _U64, _P],
    "aptai_embed_bwd": [_P, _P, _P, _I64, _I64, _F, _U64, _P],
    "aptai_xattn_softmax_fwd": [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _P],
    "aptai_xattn_softmax_bwd": [_P, _P, _P, _P, _I64, _P, _I64, _I64, _P],
    "aptai_layernorm_f32_fwd": [_P, _P, _P, _P, _P, _P, _I64, _I64, _F, _P],
    "aptai_layernorm_f32_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _P],
    "aptai_layernorm_f32_bwd_workspace_bytes": [_I64],
    "aptai_lstm_fwd": [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _P],
    "aptai_lstm_bwd": [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _P],
    "aptai_lstm_workspace_bytes": [_I64],
    "aptai_lstm_fwd_serial": [_P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _P],
    "aptai_lstm_bwd_serial": [_P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _P],
    "aptai_gather_alignment": [_P, _P, _P, _P, _I64, _I64, _I64, _P],
    "aptai_tanh_dropout_f32": [_P, _P, _P, _P, _I64, _F, _U64, _P],
    "aptai_dropout_f32": [_P, _P, _I64, _F, _U64, _P],
    "aptai_colsum_f32": [_P, _I64, _P, _P, _I64, _I64, _P],
    "aptai_ctc_fwd": [_P, _I64, _I64, _P, _I64, _P, _P, _P, _I64, _I64, _I64, _I, _I, _I, _P, _P, _P, _P, _I, _P],
    "aptai_ctc_bwd": [_P, _I64, _I64, _P, _I64, _P, _P, _P, _I64, _I64, _I64, _I, _I, _I, _P, _P, _P, _F, _P, _I64, _I, _I, _P],
    "aptai_ctc_workspace_bytes": [_I64, _I64, _I64],
    "aptai_ctc_greedy_decode": [_P, _I64, _I64, _I64, _I64, _I64, _I, _P, _I64, _P, _P],
    "aptai_mx_quantize_bf16": [_P, _I64, _P, _I64, _P, _I64, _I64, _I64, _P],
    "aptai_gemm_mxfp8": [_P, _P, _I64, _I64, _P, _P, _I64, _I64, _P, _I64, _P, _I, _P, _I64, _I64, _I64, _I64, _P],
    "aptai_layernorm_fwd_mx": [_P, _P, _P, _P, _P, _I64, _P, _I64, _I64, _I64, _F, _P],
    "aptai_gemm_mxfp8_mxout": [_P, _P, _I64, _I64, _P, _P, _I64, _I64, _P, _I64, _P, _I64, _P, _I, _I64, _I64, _I64, _P],
    "aptai_split_f32": [_P, _I64, _I64, _I64, _I, _I, _I, _P, _I64, _P],
    "aptai_bias_act_res_f32": [_P, _I64, _P, _P, _I64, _P, _I64, _I64, _I64, _I, _P, _I64, _P],
    "aptai_softmax_rows_f32": [_P, _P, _I64, _I64, _I64, _P],
    "aptai_softmax_split_f32": [_P, _P, _I64, _I64, _I64, _I, _P, _I64, _P],
    "aptai_attention_exact_fwd": [_P, _I64, _P, _P, _I64, _I64, _I64, _I64, _I64, _I, _F, _P],
    "aptai_conv0_fwd_f32": [_P, _I64, _I64, _P, _P, _P, _P, _I, _F, _P, _I64, _I64, _P, _P],
    "aptai_conv0_fwd_split": [_P, _I64, _I64, _P, _P, _P, _P, _I, _F, _P, _I, _I64, _I64, _P, _P],
    "aptai_device_check": [ctypes.c_char_p, _I],
    "aptai_set_seed_salt": [_P, _P],
    "aptai_set_frame_bounds": [_P, _P],
}


class AptaiHipError(RuntimeError):
    pass


def build(force: bool = False, jobs: int = 8) -> str:
    """Compile every HIP source for gfx950 and link libaptai_hip.so in-tree."""
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", CSRC, f"-j{jobs}"], check=True)
    return LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise AptaiHipError(
                        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        f"or `make -C {CSRC}` — aptai_amd has no CPU fallback")
                L = ctypes.CDLL(LIB_PATH)
                L.aptai_last_error.restype = ctypes.c_char_p
                for fn in declared_symbols():
                    if fn.endswith('_workspace_bytes'):
                        getattr(L, fn).restype = ctypes.c_int64
                    if fn in ARGTYPES:
                        getattr(L, fn).argtypes = ARGTYPES[fn]
                _lib = L
    return _lib


def call(name: str, *args) -> None:
    """Call an int-returning entry point with its declared argtypes and raise on a non-zero status."""
    check(getattr(lib(), name)(*args), name)


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().aptai_last_error().decode(errors="replace")
        raise AptaiHipError(f"{what or 'aptai_hip call'} failed (status {rc}): {msg}")


def declared_symbols() -> list:
    """Every function name declared in include/aptai_hip.h (used by the CPU export test)."""
    import re
    hdr = os.path.join(os.path.dirname(_HERE), "include", "aptai_hip.h")
    text = open(hdr).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(aptai_[a-z0-9_]+)\s*\(", text)))
