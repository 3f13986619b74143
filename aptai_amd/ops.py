"""Tensor-level wrappers over the C ABI (include/aptai_hip.h).  torch is plumbing here: it owns the
device buffers and the stream; every computation is a hand-written HIP kernel in libaptai_hip.so.
All wrappers raise (no fallback) when the tensors are not on a HIP device.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _lib

EPI_BIAS, EPI_GELU, EPI_RESIDUAL, EPI_DROPOUT, EPI_DGELU, EPI_ALPHA = 1, 2, 4, 8, 16, 32

c_void_p, c_i64, c_int, c_float, c_u64 = (ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float,
                                          ctypes.c_uint64)


class GemmDesc(ctypes.Structure):
    _fields_ = [("A", c_void_p), ("lda", c_i64), ("B", c_void_p), ("ldb", c_i64), ("C", c_void_p), ("ldc", c_i64),
                ("M", c_i64), ("N", c_i64), ("K", c_i64),
                ("a_kmajor", c_int), ("b_kmajor", c_int), ("out_f32", c_int), ("flags", c_int),
                ("bias", c_void_p), ("residual", c_void_p), ("ldr", c_i64), ("out_pre", c_void_p),
                ("aux", c_void_p), ("ldaux", c_i64), ("alpha", c_float), ("dropout_p", c_float), ("seed", c_u64),
                ("split_k", c_int), ("accumulate", c_int), ("workspace", c_void_p), ("workspace_bytes", c_i64)]


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _dev(*ts) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.AptaiHipError("aptai_amd ops need tensors on the MI355X (no CPU fallback)")


def gemm(a: torch.Tensor, b: torch.Tensor, M: int, N: int, K: int, *, lda=None, ldb=None, out=None, ldc=None,
         a_kmajor=False, b_kmajor=False, out_f32=False, bias=None, gelu=False, residual=None, out_pre=None,
         dgelu_aux=None, alpha: Optional[float] = None, dropout_p: float = 0.0, seed: int = 0, split_k: int = 1,
         accumulate: bool = False, workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """C[M,N] = rowop(A)[M,K] . colop(B)[N,K]^T.  See aptai_gemm_bf16 in include/aptai_hip.h."""
    _dev(a, b, out, bias, residual, out_pre, dgelu_aux)
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=torch.float32 if out_f32 else torch.bfloat16)
    d = GemmDesc()
    d.A, d.lda = a.data_ptr(), lda if lda is not None else (a.stride(0))
    d.B, d.ldb = b.data_ptr(), ldb if ldb is not None else (b.stride(0))
    d.C, d.ldc = out.data_ptr(), ldc if ldc is not None else out.stride(0)
    d.M, d.N, d.K = M, N, K
    d.a_kmajor, d.b_kmajor, d.out_f32 = int(a_kmajor), int(b_kmajor), int(out_f32)
    flags = 0
    if bias is not None:
        flags |= EPI_BIAS
        d.bias = bias.data_ptr()
    if gelu:
        flags |= EPI_GELU
    if residual is not None:
        flags |= EPI_RESIDUAL
        d.residual, d.ldr = residual.data_ptr(), residual.stride(0)
    if out_pre is not None:
        d.out_pre = out_pre.data_ptr()
    if dgelu_aux is not None:
        flags |= EPI_DGELU
        d.aux, d.ldaux = dgelu_aux.data_ptr(), dgelu_aux.stride(0)
    if alpha is not None:
        flags |= EPI_ALPHA
        d.alpha = alpha
    if dropout_p > 0:
        flags |= EPI_DROPOUT
        d.dropout_p, d.seed = dropout_p, seed
    d.flags = flags
    d.split_k, d.accumulate = split_k, int(accumulate)
    if out_f32 and (split_k > 1 or accumulate):
        need = _lib.lib().aptai_gemm_workspace_bytes(M, N, split_k)
        if workspace is None or workspace.numel() * workspace.element_size() < need:
            workspace = torch.empty(need, device=a.device, dtype=torch.uint8)
        d.workspace, d.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
    _lib.check(_lib.lib().aptai_gemm_bf16(ctypes.byref(d), c_void_p(_stream())), "aptai_gemm_bf16")
    return out


# ----------------------------------------------------------------------------- LayerNorm
def layernorm_fwd(x, gamma, beta, eps, *, gelu_after=False, save_stats=True, out=None):
    _dev(x, gamma, beta)
    rows, cols = x.shape[0], x.shape[1]
    y = out if out is not None else torch.empty_like(x)
    mean = rstd = None
    if save_stats:
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().aptai_layernorm_fwd(c_void_p(x.data_ptr()), c_void_p(gamma.data_ptr()),
                                              c_void_p(beta.data_ptr()), c_void_p(y.data_ptr()), c_void_p(_ptr(mean)),
                                              c_void_p(_ptr(rstd)), c_i64(rows), c_i64(cols), c_float(eps),
                                              c_int(int(gelu_after)), c_void_p(_stream())), "aptai_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, mean, rstd, gamma, *, dres=None, dropout_p=0.0, seed=0, need_param_grads=True):
    """Returns (dx, dx_drop | None, dgamma | None, dbeta | None)."""
    _dev(dy, x, mean, rstd, gamma, dres)
    rows, cols = x.shape
    dx = torch.empty_like(x)
    dx_drop = torch.empty_like(x) if dropout_p > 0 else None
    dgamma = dbeta = None
    if need_param_grads:
        dgamma = torch.empty(cols, device=x.device, dtype=torch.float32)
        dbeta = torch.empty(cols, device=x.device, dtype=torch.float32)
    ws = torch.empty(_lib.lib().aptai_layernorm_bwd_workspace_bytes(c_i64(rows), c_i64(cols)), device=x.device,
                     dtype=torch.uint8)
    _lib.check(_lib.lib().aptai_layernorm_bwd(
        c_void_p(dy.data_ptr()), c_void_p(x.data_ptr()), c_void_p(mean.data_ptr()), c_void_p(rstd.data_ptr()),
        c_void_p(gamma.data_ptr()), c_void_p(_ptr(dres)), c_void_p(dx.data_ptr()), c_void_p(_ptr(dx_drop)),
        c_float(dropout_p), c_u64(seed), c_void_p(_ptr(dgamma)), c_void_p(_ptr(dbeta)), c_void_p(ws.data_ptr()),
        c_i64(rows), c_i64(cols), c_void_p(_stream())), "aptai_layernorm_bwd")
    return dx, dx_drop, dgamma, dbeta


# ----------------------------------------------------------------------------- attention
def attention_fwd(qkv, lens_i32, B, Tp, H, heads, *, dropout_p=0.0, seed=0, save_lse=True):
    _dev(qkv, lens_i32)
    ctx = torch.empty((B * Tp, H), device=qkv.device, dtype=torch.bfloat16)
    lse2 = torch.empty((B, heads, Tp), device=qkv.device, dtype=torch.float32) if save_lse else None
    _lib.check(_lib.lib().aptai_attention_fwd(
        c_void_p(qkv.data_ptr()), c_void_p(lens_i32.data_ptr()), c_void_p(ctx.data_ptr()), c_void_p(_ptr(lse2)),
        c_i64(B), c_i64(Tp), c_i64(H), c_i64(heads), c_float((H // heads) ** -0.5), c_float(dropout_p), c_u64(seed),
        c_void_p(_stream())), "aptai_attention_fwd")
    return ctx, lse2


def attention_bwd(qkv, lens_i32, ctx, dctx, lse2, B, Tp, H, heads, *, dropout_p=0.0, seed=0, dctx_zero_beyond_len=False):
    _dev(qkv, lens_i32, ctx, dctx, lse2)
    dqkv = torch.empty_like(qkv)
    delta = torch.empty((B, heads, Tp), device=qkv.device, dtype=torch.float32)
    _lib.check(_lib.lib().aptai_attention_bwd(
        c_void_p(qkv.data_ptr()), c_void_p(lens_i32.data_ptr()), c_void_p(ctx.data_ptr()), c_void_p(dctx.data_ptr()),
        c_void_p(lse2.data_ptr()), c_void_p(delta.data_ptr()), c_void_p(dqkv.data_ptr()), c_i64(B), c_i64(Tp),
        c_i64(H), c_i64(heads), c_float((H // heads) ** -0.5), c_float(dropout_p), c_u64(seed),
        c_int(int(dctx_zero_beyond_len)), c_void_p(_stream())), "aptai_attention_bwd")
    return dqkv
