"""Tensor-level wrappers over the C ABI (include/aptai_hip.h).  torch is plumbing here: it owns the
device buffers and the stream; every computation is a hand-written HIP kernel in libaptai_hip.so.
All wrappers raise (no fallback) when the tensors are not on a HIP device.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _lib

EPI_BIAS, EPI_GELU, EPI_RESIDUAL, EPI_DROPOUT, EPI_DGELU, EPI_ALPHA, EPI_PRE_DGELU, EPI_MUL_AUX, EPI_RESIDUAL_F32, EPI_SPLIT_OUT, EPI_BIAS_ROW = 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024

c_void_p, c_i64, c_int, c_float, c_u64 = (ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float,
                                          ctypes.c_uint64)


class GemmDesc(ctypes.Structure):
    _fields_ = [("A", c_void_p), ("lda", c_i64), ("B", c_void_p), ("ldb", c_i64), ("C", c_void_p), ("ldc", c_i64),
                ("M", c_i64), ("N", c_i64), ("K", c_i64),
                ("a_kmajor", c_int), ("b_kmajor", c_int), ("out_f32", c_int), ("flags", c_int),
                ("bias", c_void_p), ("residual", c_void_p), ("ldr", c_i64), ("out_pre", c_void_p),
                ("aux", c_void_p), ("ldaux", c_i64), ("alpha", c_float), ("dropout_p", c_float), ("seed", c_u64),
                ("split_k", c_int), ("accumulate", c_int), ("workspace", c_void_p), ("workspace_bytes", c_i64),
                ("batch_outer", c_int), ("batch_inner", c_int),
                ("batch_stride_a", c_i64 * 2), ("batch_stride_b", c_i64 * 2), ("batch_stride_c", c_i64 * 2),
                ("batch_stride_bias", c_i64 * 2), ("batch_stride_res", c_i64 * 2), ("batch_stride_aux", c_i64 * 2),
                ("tile", c_int), ("colscale_n", c_int), ("colscale", c_float),
                ("sk_workspace", c_void_p), ("sk_workspace_bytes", c_i64), ("split_out_pieces", c_int), ("split_out_bcol", c_int)]


TILE_STREAMK = 257
_SK_WS = {}


def _sk_workspace(device) -> torch.Tensor:
    """Stream-K workspace (one 256 x 256 fp32 slab per CU + self-cleaning ready flags + status word) of the CURRENT stream:
    zeroed once here; launches on one stream are serialised, so they share it; another stream gets its own."""
    key = (str(device), _stream())
    ws = _SK_WS.get(key)
    if ws is None:
        ws = torch.zeros(int(_lib.lib().aptai_gemm_sk_workspace_bytes()), device=device, dtype=torch.uint8)
        _SK_WS[key] = ws
    return ws


def gemm_sk_check(device=None) -> None:
    """Raises if a bounded wait of a stream-K GEMM launch gave up since the last check (its output tile was then incomplete).
    Synchronises: call where the host waits for the device anyway (loss logging, end of an epoch, after a timed region)."""
    for (d, st), ws in list(_SK_WS.items()):
        if device is not None and d != str(device):
            continue
        out = ctypes.c_int(0)
        _lib.check(_lib.lib().aptai_gemm_sk_status(c_void_p(ws.data_ptr()), c_void_p(st), ctypes.byref(out)), "aptai_gemm_sk_status")
        if out.value != 0:
            raise _lib.AptaiHipError("aptai_gemm_bf16 (stream-K): a wait for another workgroup's partial tile timed out; "
                                     "the output of that launch is incomplete")


class GemmProbe:
    """Brackets every aptai_gemm_bf16 launch of one operand-layout family with HIP events on the launch stream and
    accumulates (algorithmic flops, elapsed ms): bench.py's live roofline measurement of the dominant kernel."""

    def __init__(self, a_kmajor=False, b_kmajor=False, out_f32=False):
        self.key = (bool(a_kmajor), bool(b_kmajor), bool(out_f32))
        self.records = []

    def summary(self):
        torch.cuda.synchronize()
        flops = sum(f for f, _, _ in self.records)
        ms = sum(s.elapsed_time(e) for _, s, e in self.records)
        return dict(launches=len(self.records), flops=flops, ms=ms)


_probe: Optional["GemmProbe"] = None


def set_gemm_probe(p: Optional["GemmProbe"]) -> None:
    global _probe
    _probe = p


def _stream() -> int:
    # raw handle of torch's current stream; ~10x cheaper than torch.cuda.current_stream().cuda_stream, which was 0.9 ms of host
    # time per eager train step (1000+ launches)
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _dev(*ts) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.AptaiHipError("aptai_amd ops need tensors on the MI355X (no CPU fallback)")


def _gemm_desc(d: "GemmDesc", a: torch.Tensor, b: torch.Tensor, M: int, N: int, K: int, *, lda=None, ldb=None, out=None, ldc=None,
               a_kmajor=False, b_kmajor=False, out_f32=False, bias=None, gelu=False, residual=None, out_pre=None,
               dgelu_aux=None, alpha: Optional[float] = None, dropout_p: float = 0.0, seed: int = 0, split_k: int = 1,
               accumulate: bool = False, workspace: Optional[torch.Tensor] = None, batch=None, ldr=None, tile: int = 0,
               ldaux=None, pre_dgelu: bool = False, mul_aux=None, colscale=None, residual_f32=None, split_out=None, split_bcol=None, bias_row=None):
    """Fills one aptai_gemm_desc in place; returns (out, workspace) - the caller keeps them alive across the launch."""
    _dev(a, b, out, bias, residual, out_pre, dgelu_aux, mul_aux)
    if out is None:
        if split_out:          # exact-index mode: fp32 result written as `split_out` bf16 pieces (EPI_SPLIT_OUT); ldc in bf16 elements
            out = torch.empty((M, N * int(split_out)), device=a.device, dtype=torch.bfloat16)
        else:
            out = torch.empty((M, N), device=a.device, dtype=torch.float32 if out_f32 else torch.bfloat16)
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
    if residual_f32 is not None:   # fp32 output += fp32 residual (the inference-only encoder's fp32 residual stream)
        _dev(residual_f32)
        flags |= EPI_RESIDUAL_F32
        d.residual, d.ldr = residual_f32.data_ptr(), residual_f32.stride(0)
    if out_pre is not None:
        d.out_pre = out_pre.data_ptr()
    if dgelu_aux is not None:
        flags |= EPI_DGELU
        d.aux, d.ldaux = dgelu_aux.data_ptr(), dgelu_aux.stride(0)
    if mul_aux is not None:        # backward partner of pre_dgelu: *= saved dropmask * gelu'(pre-activation)
        flags |= EPI_MUL_AUX
        d.aux, d.ldaux = mul_aux.data_ptr(), mul_aux.stride(0)
    if pre_dgelu:                  # out_pre receives dropmask/(1-p) * gelu'(pre-activation) instead of the pre-activation
        flags |= EPI_PRE_DGELU
    if alpha is not None:
        flags |= EPI_ALPHA
        d.alpha = alpha
    if dropout_p > 0:
        flags |= EPI_DROPOUT
        d.dropout_p, d.seed = dropout_p, seed
    if bias_row is not None:       # fp32-output launches: bias indexed by the output row (a product evaluated transposed)
        _dev(bias_row)
        flags |= EPI_BIAS_ROW
        d.bias = bias_row.data_ptr()
    if split_out:
        if out.dtype != torch.bfloat16 or not out_f32:
            raise ValueError("split_out writes bf16 pieces from an out_f32 launch")
        flags |= EPI_SPLIT_OUT
        d.split_out_pieces = int(split_out)
        d.split_out_bcol = int(split_bcol) if split_bcol is not None else int(N)
    d.flags = flags
    d.split_k, d.accumulate = split_k, int(accumulate)
    d.tile = tile if tile else _AUTO_TILE
    if tile == TILE_STREAMK:          # opt-in only: measured slower than the tile rule's choice on every hot-path shape
        ws_sk = _sk_workspace(a.device)
        d.sk_workspace, d.sk_workspace_bytes = ws_sk.data_ptr(), ws_sk.numel()
    if colscale is not None:       # (n_cols, factor): output columns [0, n_cols) *= factor after alpha / bias
        d.colscale_n, d.colscale = int(colscale[0]), float(colscale[1])
    if ldaux is not None:
        d.ldaux = ldaux
    if ldr is not None:
        d.ldr = ldr
    if batch is not None:
        # batch = dict(outer=, inner=, a=(so,si), b=(so,si), c=(so,si), bias=(so,si), res=(so,si), aux=(so,si))
        d.batch_outer, d.batch_inner = batch.get("outer", 1), batch.get("inner", 1)
        for key, field in (("a", "batch_stride_a"), ("b", "batch_stride_b"), ("c", "batch_stride_c"),
                           ("bias", "batch_stride_bias"), ("res", "batch_stride_res"), ("aux", "batch_stride_aux")):
            so, si = batch.get(key, (0, 0))
            getattr(d, field)[0], getattr(d, field)[1] = so, si
    if out_f32 and (split_k > 1 or accumulate):
        need = _lib.lib().aptai_gemm_workspace_bytes(M, N, split_k)
        if workspace is None or workspace.numel() * workspace.element_size() < need:
            workspace = torch.empty(need, device=a.device, dtype=torch.uint8)
        d.workspace, d.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
    return out, workspace


_AUTO_TILE = 0        # what `tile = 0` means to _gemm_desc: 0 = the library's rule


class auto_tile:
    """Context: GEMMs that leave the tile to the library run with `tile` instead (0 restores the rule).  Force_APTAI captures its side-stream
    encoder pass under auto_tile(128): a 256 x 256 workgroup owns a whole CU (128 KiB of LDS, 2 x 240 registers per SIMD), so while such a
    launch runs the cooperative BiLSTM workgroups of the heads cannot start beside it - the pipelined step measured 7.70 ms with 128-row
    tiles in the encoder against 8.1-8.4 ms with the rule's 256-row launches, although the encoder alone is slower that way."""

    def __init__(self, tile: int):
        self.tile = int(tile)

    def __enter__(self):
        global _AUTO_TILE
        self.prev, _AUTO_TILE = _AUTO_TILE, self.tile
        return self

    def __exit__(self, *exc):
        global _AUTO_TILE
        _AUTO_TILE = self.prev
        return False


def gemm(a: torch.Tensor, b: torch.Tensor, M: int, N: int, K: int, **kw) -> torch.Tensor:
    """C[M,N] = rowop(A)[M,K] . colop(B)[N,K]^T.  See aptai_gemm_bf16 in include/aptai_hip.h (keywords: _gemm_desc)."""
    d = GemmDesc()
    out, _ws_keep = _gemm_desc(d, a, b, M, N, K, **kw)
    pr = _probe
    if pr is not None and pr.key == (bool(d.a_kmajor), bool(d.b_kmajor), bool(d.out_f32)) and not torch.cuda.is_current_stream_capturing():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(_lib.lib().aptai_gemm_bf16(ctypes.byref(d), c_void_p(_stream())), "aptai_gemm_bf16")
        e1.record()
        batch = kw.get("batch")
        nb = (batch.get("outer", 1) * batch.get("inner", 1)) if batch else 1
        pr.records.append((2.0 * M * N * K * nb, e0, e1))
        return out
    _lib.check(_lib.lib().aptai_gemm_bf16(ctypes.byref(d), c_void_p(_stream())), "aptai_gemm_bf16")
    return out


def gemm_grouped(problems) -> list:
    """One launch for up to 8 independent GEMMs of one operand layout (aptai_gemm_bf16_grouped).
    problems: iterable of (a, b, M, N, K, kwargs-dict) with the keywords of gemm(); returns the outputs in order."""
    problems = list(problems)
    descs = (GemmDesc * len(problems))()
    outs, keep = [], []
    flops = 0.0
    for d, (a, b, M, N, K, kw) in zip(descs, problems):
        out, ws = _gemm_desc(d, a, b, M, N, K, **kw)
        outs.append(out)
        keep.append(ws)
        flops += 2.0 * M * N * K
    pr = _probe
    d0 = descs[0]
    if pr is not None and pr.key == (bool(d0.a_kmajor), bool(d0.b_kmajor), bool(d0.out_f32)) and not torch.cuda.is_current_stream_capturing():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(_lib.lib().aptai_gemm_bf16_grouped(descs, len(problems), c_void_p(_stream())), "aptai_gemm_bf16_grouped")
        e1.record()
        pr.records.append((flops, e0, e1))
        return outs
    _lib.check(_lib.lib().aptai_gemm_bf16_grouped(descs, len(problems), c_void_p(_stream())), "aptai_gemm_bf16_grouped")
    return outs


_ONES = {}


def ones_kmajor(K: int, device) -> torch.Tensor:
    """[K][8] bf16 ones: the A operand that turns a column sum into a grouped-GEMM problem (bias gradients)."""
    key = (K, str(device))
    t = _ONES.get(key)
    if t is None:
        t = torch.ones((K, 8), device=device, dtype=torch.bfloat16)
        _ONES[key] = t
    return t


# ----------------------------------------------------------------------------- MX block-scaled FP8 (inference-only encoder)
def mx_quantize(x: torch.Tensor, out=None):
    """bf16 [rows][K] -> (q uint8 [rows][K], scales uint8 [rows][K/32]) in OCP MXFP8 (E4M3 elements, E8M0 block scales)."""
    _dev(x)
    rows, K = x.shape
    if out is None:
        q = torch.empty((rows, K), device=x.device, dtype=torch.uint8)
        s = torch.empty((rows, K // 32), device=x.device, dtype=torch.uint8)
    else:
        q, s = out
    _lib.call("aptai_mx_quantize_bf16", x.data_ptr(), x.stride(0), q.data_ptr(), q.stride(0), s.data_ptr(), s.stride(0), rows, K, _stream())
    return q, s


def gemm_mxfp8(aq, a_s, bq, b_s, M, N, K, *, bias=None, gelu=False, residual=None, out=None):
    """C bf16 [M][N] = dequant(A) . dequant(B)^T + bias [-> GELU] [+ residual] (aptai_gemm_mxfp8)."""
    _dev(aq, a_s, bq, b_s, bias, residual, out)
    if out is None:
        out = torch.empty((M, N), device=aq.device, dtype=torch.bfloat16)
    _lib.call("aptai_gemm_mxfp8", aq.data_ptr(), a_s.data_ptr(), aq.stride(0), a_s.stride(0), bq.data_ptr(), b_s.data_ptr(), bq.stride(0),
              b_s.stride(0), out.data_ptr(), out.stride(0), _ptr(bias), int(gelu), _ptr(residual), residual.stride(0) if residual is not None else 0,
              M, N, K, _stream())
    return out


def layernorm_fwd_mx(x, gamma, beta, eps, *, want_bf16=False):
    """nn.LayerNorm -> (bf16 | None, q uint8 [rows][cols], scales uint8 [rows][cols/32]) in one launch (aptai_layernorm_fwd_mx)."""
    _dev(x, gamma, beta)
    rows, cols = x.shape
    y = torch.empty_like(x) if want_bf16 else None
    q = torch.empty((rows, cols), device=x.device, dtype=torch.uint8)
    s = torch.empty((rows, cols // 32), device=x.device, dtype=torch.uint8)
    _lib.call("aptai_layernorm_fwd_mx", x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _ptr(y), q.data_ptr(), q.stride(0), s.data_ptr(), s.stride(0),
              rows, cols, eps, _stream())
    return y, q, s


def gemm_mxfp8_mxout(aq, a_s, bq, b_s, M, N, K, *, bias=None, gelu=False, out=None):
    """(q uint8 [M][N], scales uint8 [M][N/32]) = mx_quantize(bf16(dequant(A) . dequant(B)^T + bias [-> GELU])) in one launch
    (aptai_gemm_mxfp8_mxout): the next MX GEMM's A operand without the bf16 tensor in between."""
    _dev(aq, a_s, bq, b_s, bias)
    if out is None:
        q = torch.empty((M, N), device=aq.device, dtype=torch.uint8)
        s = torch.empty((M, N // 32), device=aq.device, dtype=torch.uint8)
    else:
        q, s = out
    _lib.call("aptai_gemm_mxfp8_mxout", aq.data_ptr(), a_s.data_ptr(), aq.stride(0), a_s.stride(0), bq.data_ptr(), b_s.data_ptr(), bq.stride(0),
              b_s.stride(0), q.data_ptr(), q.stride(0), s.data_ptr(), s.stride(0), _ptr(bias), int(gelu), M, N, K, _stream())
    return q, s


# ----------------------------------------------------------------------------- exact (fp32-class) inference path
def split_f32(x32: torch.Tensor, pieces: int, *, weight_side: bool = False, gelu: bool = False, rows=None, cols=None, ldx=None) -> torch.Tensor:
    """fp32 [rows][cols] -> bf16 pieces [rows][cols * pieces] in the K-tile-interleaved layout of aptai_split_f32."""
    _dev(x32)
    rows = x32.shape[0] if rows is None else rows
    cols = x32.shape[1] if cols is None else cols
    out = torch.empty((rows, cols * pieces), device=x32.device, dtype=torch.bfloat16)
    _lib.call("aptai_split_f32", x32.data_ptr(), ldx if ldx is not None else x32.stride(0), rows, cols, int(weight_side), pieces, int(gelu),
              out.data_ptr(), cols * pieces, _stream())
    return out


def bias_act_res_f32(x32, *, bias=None, res=None, gelu=False, lens_i32=None, rows_per_b=0, out=None):
    """y = [res +] gelu_erf?(x + bias) in fp32; rows t >= lens[b] zeroed when lens_i32 is given (aptai_bias_act_res_f32)."""
    _dev(x32, bias, res, lens_i32, out)
    rows, cols = x32.shape
    y = out if out is not None else torch.empty((rows, cols), device=x32.device, dtype=torch.float32)
    _lib.call("aptai_bias_act_res_f32", x32.data_ptr(), x32.stride(0), _ptr(bias), _ptr(res), res.stride(0) if res is not None else 0,
              y.data_ptr(), y.stride(0), rows, cols, int(gelu), _ptr(lens_i32), rows_per_b, _stream())
    return y


def softmax_rows_f32(s32, lens_i32, B, heads, Tp):
    _dev(s32, lens_i32)
    _lib.call("aptai_softmax_rows_f32", s32.data_ptr(), lens_i32.data_ptr(), B, heads, Tp, _stream())
    return s32


def conv0_fwd_split(audio, weight, bias, gamma, beta, mode, out_split, pieces, T_real, T_alloc, stats, eps=1e-5):
    """The exact mode's first conv layer with its (already activated) result written as split bf16 pieces (aptai_conv0_fwd_split)."""
    _dev(audio, weight, bias, gamma, beta, out_split, stats)
    B, S = audio.shape
    _lib.call("aptai_conv0_fwd_split", audio.data_ptr(), B, S, weight.data_ptr(), _ptr(bias), gamma.data_ptr(), beta.data_ptr(), mode, eps,
              out_split.data_ptr(), pieces, T_real, T_alloc, _ptr(stats), _stream())
    return out_split


def softmax_split_f32(s32, lens_i32, B, heads, Tp, pieces):
    """Masked softmax of the fp32 scores [B][heads][Tp][Tp] written as split bf16 pieces [B heads Tp][pieces Tp] (aptai_softmax_split_f32)."""
    _dev(s32, lens_i32)
    out = torch.empty((B * heads * Tp, pieces * Tp), device=s32.device, dtype=torch.bfloat16)
    _lib.call("aptai_softmax_split_f32", s32.data_ptr(), lens_i32.data_ptr(), B, heads, Tp, pieces, out.data_ptr(), pieces * Tp, _stream())
    return out


def attention_exact_fwd(qkv_s, lens_i32, B, Tp, H, heads, pieces, scale, out=None):
    """Fused attention core of the exact-index mode (aptai_attention_exact_fwd): qkv_s bf16 [B Tp][3 pieces H] split Q | K | V as the
    q|k|v projection's split-out epilogue wrote them -> the context as split pieces [B Tp][pieces H] (the out-projection's A operand)."""
    _dev(qkv_s, lens_i32, out)
    if qkv_s.dtype != torch.bfloat16 or qkv_s.stride(1) != 1 or qkv_s.shape[0] != B * Tp or qkv_s.shape[1] != 3 * pieces * H:
        raise _lib.AptaiHipError("attention_exact_fwd: qkv_s must be bf16 [B Tp][3 pieces H] with unit column stride")
    if out is None:
        out = torch.empty((B * Tp, pieces * H), device=qkv_s.device, dtype=torch.bfloat16)
    _lib.call("aptai_attention_exact_fwd", qkv_s.data_ptr(), qkv_s.stride(0), lens_i32.data_ptr(), out.data_ptr(), out.stride(0), B, Tp, H,
              heads, pieces, float(scale), _stream())
    return out


def conv0_fwd_f32(audio, weight, bias, gamma, beta, mode, out32, T_real, T_alloc, stats, eps=1e-5):
    _dev(audio, weight, bias, gamma, beta, out32, stats)
    B, S = audio.shape
    _lib.call("aptai_conv0_fwd_f32", audio.data_ptr(), B, S, weight.data_ptr(), _ptr(bias), gamma.data_ptr(), beta.data_ptr(), mode, eps,
              out32.data_ptr(), T_real, T_alloc, _ptr(stats), _stream())
    return out32


def gemm_split(a_s: torch.Tensor, w_s: torch.Tensor, M: int, N: int, K: int, pieces: int, *, lda=None, bias=None, residual_f32=None,
               out=None, ldc=None, split_out: bool = False, gelu: bool = False, split_bcol=None) -> torch.Tensor:
    """fp32 C[M][N] = A . W^T (+ bias) (+ fp32 residual) from split operands (split_f32): one NT launch of K' = pieces * K.
    split_out: the result (after the erf GELU when `gelu`) is returned as split bf16 pieces [M][pieces N] instead - the next product's A."""
    # fp32-output launches name their tile (include/aptai_hip.h); whole rounds of 256 x 256 tiles where the output has them (the conv
    # stack's [B x 16384 ...] x 512 outputs: 3 x the bf16 work at K' = 3 K is the longest loop of the build), 128-row tiles elsewhere
    t256 = -(-M // 256) * -(-N // 256)
    tile = 256 if (M >= 256 and N >= 256 and t256 >= 1024) else 128
    # one round of full 128 x 192 tiles (the [8192] x 768 outputs of out-proj / FFN2 at K' = 3 K): the 3-stage-ring kernel, as on the bf16 path
    if tile == 128 and M % 128 == 0 and N % 192 == 0 and 192 < (M // 128) * (N // 192) <= 256:
        tile = 192
    return gemm(a_s, w_s, M, N, K * pieces, lda=(lda * pieces if lda is not None else None), out_f32=True, bias=bias, gelu=(gelu and split_out),
                residual_f32=residual_f32, out=out, ldc=ldc, tile=tile, split_out=(pieces if split_out else None), split_bcol=split_bcol)


# ----------------------------------------------------------------------------- LayerNorm
def _ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 16), device=device, dtype=torch.uint8)


def layernorm_fwd(x, gamma, beta, eps, *, gelu_after=False, save_stats=True, out=None):
    _dev(x, gamma, beta)
    rows, cols = x.shape[0], x.shape[1]
    y = out if out is not None else torch.empty_like(x)
    mean = rstd = None
    if save_stats:
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
    _lib.call("aptai_layernorm_fwd", x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), _ptr(mean), _ptr(rstd),
              rows, cols, eps, int(gelu_after), _stream())
    return y, mean, rstd


_LN_DEFER = None      # list of (workspace, dgamma, dbeta, blocks, cols) while a graph runner collects the parameter-gradient reductions


def ln_defer_begin() -> list:
    """From here on layernorm_bwd only writes its per-block partials and registers the reduction; ln_finalize_multi() runs them all in
    ONE launch (aptai_layernorm_bwd_finalize_multi).  Used by the graph runner around the capture of its backward segments: the
    returned dgamma / dbeta tensors are filled by that launch, which the runner replays at the end of the backward pass."""
    global _LN_DEFER
    _LN_DEFER = []
    return _LN_DEFER


def ln_defer_end() -> None:
    global _LN_DEFER
    _LN_DEFER = None


def ln_defer_table(jobs: list):
    """Device job table of the registered reductions (build it OUTSIDE stream capture: it is a host-to-device copy); the caller
    keeps it, and the job list (workspaces, outputs), alive for as long as it launches ln_finalize_multi on it."""
    rows = [[ws.data_ptr(), dg.data_ptr(), db.data_ptr(), blocks, cols] for ws, dg, db, blocks, cols in jobs]
    return torch.tensor(rows, dtype=torch.int64).to(jobs[0][0].device), len(rows), max(r[4] for r in rows)


def ln_finalize_multi(table, n: int, max_cols: int) -> None:
    """Every registered dgamma / dbeta reduction in ONE launch (aptai_layernorm_bwd_finalize_multi); capturable."""
    _lib.call("aptai_layernorm_bwd_finalize_multi", table.data_ptr(), n, max_cols, _stream())


def layernorm_bwd(dy, x, mean, rstd, gamma, *, dres=None, dropout_p=0.0, seed=0, need_param_grads=True, beta_gelu=None):
    """Returns (dx, dx_drop | None, dgamma | None, dbeta | None)."""
    _dev(dy, x, mean, rstd, gamma, dres)
    rows, cols = x.shape
    dx = torch.empty_like(x)
    dx_drop = torch.empty_like(x) if dropout_p > 0 else None
    dgamma = dbeta = None
    if need_param_grads:
        dgamma = torch.empty(cols, device=x.device, dtype=torch.float32)
        dbeta = torch.empty(cols, device=x.device, dtype=torch.float32)
    nbytes = _lib.lib().aptai_layernorm_bwd_workspace_bytes(rows, cols)
    ws = _ws(nbytes, x.device)
    defer = need_param_grads and _LN_DEFER is not None
    _lib.call("aptai_layernorm_bwd", dy.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
              _ptr(dres), dx.data_ptr(), _ptr(dx_drop), dropout_p, seed, None if defer else _ptr(dgamma), None if defer else _ptr(dbeta),
              ws.data_ptr(), rows, cols, _ptr(beta_gelu), _stream())
    if defer:
        _LN_DEFER.append((ws, dgamma, dbeta, nbytes // (8 * cols), cols))
    return dx, dx_drop, dgamma, dbeta


# ----------------------------------------------------------------------------- attention
ATTN_LOG2E = 1.4426950408889634


def layernorm_fwd_f32in(x32, gamma, beta, eps, *, want_bf16=True, want_f32=True):
    """nn.LayerNorm on an fp32 activation [rows][cols] -> (bf16 copy | None, fp32 copy | None) (aptai_layernorm_fwd_f32in)."""
    _dev(x32, gamma, beta)
    rows, cols = x32.shape
    y = torch.empty((rows, cols), device=x32.device, dtype=torch.bfloat16) if want_bf16 else None
    y32 = torch.empty((rows, cols), device=x32.device, dtype=torch.float32) if want_f32 else None
    _lib.call("aptai_layernorm_fwd_f32in", x32.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _ptr(y), _ptr(y32), rows, cols, eps, _stream())
    return y, y32


def layernorm_fwd_f32in_split(x32, gamma, beta, eps, pieces, *, want_f32=True):
    """nn.LayerNorm on an fp32 activation -> (fp32 copy | None, split bf16 pieces [rows][pieces cols]) (aptai_layernorm_fwd_f32in_split)."""
    _dev(x32, gamma, beta)
    rows, cols = x32.shape
    y32 = torch.empty((rows, cols), device=x32.device, dtype=torch.float32) if want_f32 else None
    ys = torch.empty((rows, cols * pieces), device=x32.device, dtype=torch.bfloat16)
    _lib.call("aptai_layernorm_fwd_f32in_split", x32.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _ptr(y32), ys.data_ptr(), pieces, rows, cols, eps, _stream())
    return y32, ys


def attention_qscale(H: int, heads: int) -> float:
    """The factor the fused q|k|v projection applies to its Q columns (gemm(..., colscale=(H, this))) for q_prescaled attention."""
    return (H // heads) ** -0.5 * ATTN_LOG2E


def attention_fwd(qkv, lens_i32, B, Tp, H, heads, *, dropout_p=0.0, seed=0, save_lse=True, q_prescaled=False):
    _dev(qkv, lens_i32)
    ctx = torch.empty((B * Tp, H), device=qkv.device, dtype=torch.bfloat16)
    lse2 = torch.empty((B, heads, Tp), device=qkv.device, dtype=torch.float32) if save_lse else None
    ctx32 = torch.empty((B * Tp, H), device=qkv.device, dtype=torch.float32) if save_lse else None
    _lib.call("aptai_attention_fwd", qkv.data_ptr(), lens_i32.data_ptr(), ctx.data_ptr(), _ptr(lse2), _ptr(ctx32), B, Tp, H,
              heads, (H // heads) ** -0.5, dropout_p, seed, int(q_prescaled), _stream())
    return ctx, (lse2, ctx32) if save_lse else None


def attention_bwd(qkv, lens_i32, ctx, dctx, stats, B, Tp, H, heads, *, dropout_p=0.0, seed=0, dctx_zero_beyond_len=False,
                  q_prescaled=False):
    """``stats`` = the (lse2, ctx_f32) pair returned by attention_fwd."""
    lse2, ctx32 = stats
    _dev(qkv, lens_i32, ctx, dctx, lse2)
    dqkv = torch.empty_like(qkv)
    delta = torch.empty((B, heads, Tp), device=qkv.device, dtype=torch.float32)
    _lib.call("aptai_attention_bwd", qkv.data_ptr(), lens_i32.data_ptr(), ctx.data_ptr(), _ptr(ctx32), dctx.data_ptr(), lse2.data_ptr(),
              delta.data_ptr(), dqkv.data_ptr(), B, Tp, H, heads, (H // heads) ** -0.5, dropout_p, seed,
              int(dctx_zero_beyond_len), int(q_prescaled), _stream())
    return dqkv


# ----------------------------------------------------------------------------- parameter prep
def cast_bf16(src: torch.Tensor, dst: Optional[torch.Tensor] = None, ld_dst: Optional[int] = None) -> torch.Tensor:
    """fp32 [rows][cols] -> bf16 (optionally into a row slice of a wider buffer)."""
    _dev(src, dst)
    src2 = src.reshape(src.shape[0], -1) if src.dim() > 1 else src.reshape(1, -1)
    rows, cols = src2.shape
    if dst is None:
        dst = torch.empty(src.shape, device=src.device, dtype=torch.bfloat16)
    _lib.call("aptai_cast_f32_to_bf16", src2.data_ptr(), dst.data_ptr(), rows, cols, ld_dst if ld_dst else cols, _stream())
    return dst


class CastPlan:
    """A fixed list of (fp32 source, destination) pairs converted by ONE aptai_cast_multi launch.  The destinations are
    persistent, so a plan replays unchanged inside a hipGraph; sources are re-read on every run (parameters move only
    through in-place optimiser updates, so their addresses are stable; `stale()` detects re-allocated parameters)."""

    def __init__(self, jobs):
        # jobs: list of (src fp32 tensor, dst tensor [bf16 or fp32])
        self.jobs = list(jobs)
        rows = []
        for src, dst in self.jobs:
            _dev(src, dst)
            n = src.numel()
            if not (src.is_contiguous() and dst.is_contiguous() and src.dtype == torch.float32 and n == dst.numel() and n % 8 == 0
                    and src.data_ptr() % 16 == 0 and dst.data_ptr() % 16 == 0):
                raise _lib.AptaiHipError("CastPlan: jobs must be contiguous, 16-byte aligned fp32 sources of n % 8 == 0 elements")
            kind = {torch.bfloat16: 0, torch.float32: 1}[dst.dtype]
            rows.append([src.data_ptr(), dst.data_ptr(), n, kind])
        self.max_n = max(r[2] for r in rows)
        self.table = torch.tensor(rows, dtype=torch.int64).to(self.jobs[0][0].device)
        self._src_ptrs = [r[0] for r in rows]

    def stale(self) -> bool:
        return any(src.data_ptr() != p for (src, _), p in zip(self.jobs, self._src_ptrs))

    def run(self) -> None:
        _lib.call("aptai_cast_multi", self.table.data_ptr(), len(self.jobs), self.max_n, _stream())


def conv_weight_bf16(w: torch.Tensor) -> torch.Tensor:
    """[N][C][Kw] fp32 -> [N][Kw*C] bf16."""
    _dev(w)
    N, C, Kw = w.shape
    out = torch.empty((N, Kw * C), device=w.device, dtype=torch.bfloat16)
    _lib.call("aptai_conv_weight_to_bf16", w.data_ptr(), out.data_ptr(), N, C, Kw, _stream())
    return out


def posconv_weight(v, gain, groups, want_dgrad=True):
    _dev(v, gain)
    H, Cg, Kw = v.shape
    norm_ws = torch.empty(257 * Kw, device=v.device, dtype=torch.float32)     # [Kw] result + [256][Kw] reduction scratch
    norm = norm_ws[:Kw]
    wf = torch.empty((groups, Cg, Kw * Cg), device=v.device, dtype=torch.bfloat16)
    wd = torch.empty_like(wf) if want_dgrad else None
    _lib.call("aptai_posconv_weight", v.data_ptr(), gain.data_ptr(), norm_ws.data_ptr(), wf.data_ptr(), _ptr(wd), H, groups, Kw,
              _stream())
    return wf, wd, norm


def posconv_weight_bwd(dw_fwd, v, gain, norm, groups):
    """Weight-norm backward of the positional conv: (dv [H][Cg][Kw], dgain [Kw]) from the weight gradient in the forward layout."""
    _dev(dw_fwd, v, gain, norm)
    H, Cg, Kw = v.shape
    dv = torch.empty_like(v)
    dgain = torch.empty(Kw, device=v.device, dtype=torch.float32)
    ws = torch.empty((H + 1) * Kw, device=v.device, dtype=torch.float32)
    _lib.call("aptai_posconv_weight_bwd", dw_fwd.data_ptr(), v.data_ptr(), gain.data_ptr(), norm.data_ptr(), dv.data_ptr(), dgain.data_ptr(),
              ws.data_ptr(), H, groups, Kw, _stream())
    return dv, dgain


def posconv_gemm(xg, w, out, B, Tp, H, groups, Kw, pad, *, first_row=0, bias=None, gelu=False, residual=None, out_pre=None):
    """Grouped positional convolution on the packed copy (aptai_posconv_gemm: 48 channels per group, 128 taps)."""
    _dev(xg, w, out, bias, residual, out_pre)
    _lib.call("aptai_posconv_gemm", xg.data_ptr(), first_row, w.data_ptr(), _ptr(bias), _ptr(residual), out.data_ptr(), _ptr(out_pre),
              B, Tp, H, groups, Kw, pad, int(gelu), _stream())
    return out


def posconv_wgrad(du_g, x_g, dw, B, Tp, H, groups, Kw, pad):
    """dW of the grouped positional convolution from the two packed copies (aptai_posconv_wgrad); dw fp32 [groups][48][Kw*48]."""
    _dev(du_g, x_g, dw)
    _lib.call("aptai_posconv_wgrad", du_g.data_ptr(), x_g.data_ptr(), dw.data_ptr(), B, Tp, H, groups, Kw, pad, _stream())
    return dw


def posconv_kernel_fits(H: int, groups: int, Kw: int, wgrad: bool = False) -> bool:
    """The dedicated kernels cover 48 channels per group (wav2vec2-base; forward, data and weight gradient) and 64 (large; forward
    and data gradient).  APTAI_POSCONV_KERNEL=0 forces the batched implicit-GEMM path (A/B)."""
    import os
    ok = (H == groups * 48) or (H == groups * 64 and not wgrad)
    return ok and Kw == 128 and os.environ.get("APTAI_POSCONV_KERNEL", "1") != "0"


def posconv_pack(x, xg, B, Tp, H, groups, pad, *, u=None, rowmajor_out=None):
    _dev(x, xg, u, rowmajor_out)
    _lib.call("aptai_posconv_pack", x.data_ptr(), _ptr(u), xg.data_ptr(), _ptr(rowmajor_out), B, Tp, H, groups, pad, _stream())


def spec_augment_mask(frame_lens_i32, B, T, mask_prob, mask_length, min_masks, seed, out=None):
    """SpecAugment time mask [B][T] uint8 sampled on the device (no host round trip of the lengths)."""
    _dev(frame_lens_i32, out)
    m = out if out is not None else torch.empty((B, T), device=frame_lens_i32.device, dtype=torch.uint8)
    _lib.call("aptai_spec_augment_mask", frame_lens_i32.data_ptr(), m.data_ptr(), B, T, float(mask_prob), mask_length, min_masks, seed,
              _stream())
    return m


def frame_mask_fwd(h, lens_i32, spec_mask_u8, embed, B, Tp, T, H):
    _dev(h, lens_i32, spec_mask_u8, embed)
    _lib.call("aptai_frame_mask_fwd", h.data_ptr(), lens_i32.data_ptr(), _ptr(spec_mask_u8), _ptr(embed), B, Tp, T, H, _stream())


def frame_mask_bwd(dy, lens_i32, spec_mask_u8, B, Tp, T, H, want_dembed):
    _dev(dy, lens_i32, spec_mask_u8)
    dembed = ws = None
    if want_dembed and spec_mask_u8 is not None:
        dembed = torch.empty(H, device=dy.device, dtype=torch.float32)
        ws = _ws(_lib.lib().aptai_frame_mask_bwd_workspace_bytes(B, Tp, H), dy.device)
    _lib.call("aptai_frame_mask_bwd", dy.data_ptr(), lens_i32.data_ptr(), _ptr(spec_mask_u8), _ptr(dembed), _ptr(ws), B, Tp, T, H,
              _stream())
    return dembed


def colsum(x, rows, N, *, ld=None, out=None, accumulate=False):
    _dev(x, out)
    if out is None:
        out = torch.empty(N, device=x.device, dtype=torch.float32)
    ws = _ws(_lib.lib().aptai_colsum_workspace_bytes(rows, N), x.device)
    _lib.call("aptai_colsum_bf16", x.data_ptr(), ld if ld else x.stride(0), out.data_ptr(), ws.data_ptr(), rows, N,
              int(accumulate), _stream())
    return out


def dropout(x, p, seed):
    _dev(x)
    y = torch.empty_like(x)
    _lib.call("aptai_dropout_bf16", x.data_ptr(), y.data_ptr(), x.numel(), p, seed, _stream())
    return y


def dgelu(dy, u):
    _dev(dy, u)
    out = torch.empty_like(dy)
    _lib.call("aptai_dgelu_bf16", dy.data_ptr(), u.data_ptr(), out.data_ptr(), dy.numel(), _stream())
    return out


# ----------------------------------------------------------------------------- conv layer 0
def conv0_fwd(audio, weight, bias, gamma, beta, mode, out, T_real, T_alloc, eps=1e-5, want_stats=False):
    """Returns the (mean, rstd) block [B][2][512] in group mode when ``want_stats`` (needed by conv0_bwd)."""
    _dev(audio, weight, bias, gamma, beta, out)
    B, S = audio.shape
    C, _, Kw = weight.shape
    ws = _ws(_lib.lib().aptai_conv0_workspace_bytes(B, T_real), audio.device) if mode == 0 else None
    stats = torch.empty((B, 2, C), device=audio.device, dtype=torch.float32) if (want_stats and mode == 0) else None
    _lib.call("aptai_conv0_fwd", audio.data_ptr(), B, S, weight.data_ptr(), _ptr(bias), gamma.data_ptr(), beta.data_ptr(), mode,
              eps, out.data_ptr(), T_real, T_alloc, C, Kw, 5, _ptr(ws), _ptr(stats), _stream())
    return stats


def conv0_bwd(audio, weight, bias, gamma, beta, mode, dy, T_real, T_alloc, stats, eps=1e-5):
    """Returns (dweight [512][1][10], dbias | None, dgamma, dbeta)."""
    _dev(audio, weight, bias, gamma, beta, dy, stats)
    B, S = audio.shape
    C = weight.shape[0]
    dev = audio.device
    dw = torch.empty_like(weight)
    db = torch.empty(C, device=dev, dtype=torch.float32) if bias is not None else None
    dg = torch.empty(C, device=dev, dtype=torch.float32)
    dbt = torch.empty(C, device=dev, dtype=torch.float32)
    ws = _ws(_lib.lib().aptai_conv0_bwd_workspace_bytes(B, T_real), dev)
    _lib.call("aptai_conv0_bwd", audio.data_ptr(), B, S, weight.data_ptr(), _ptr(bias), gamma.data_ptr(), beta.data_ptr(), mode, eps,
              dy.data_ptr(), T_real, T_alloc, _ptr(stats), dw.data_ptr(), _ptr(db), dg.data_ptr(), dbt.data_ptr(), ws.data_ptr(),
              _stream())
    return dw, db, dg, dbt


# ----------------------------------------------------------------------------- APTAI heads
def head_act_fwd(h, p_tv, p_ph, seed):
    _dev(h)
    a_tv, a_ph = torch.empty_like(h), torch.empty_like(h)
    _lib.call("aptai_head_act_fwd", h.data_ptr(), a_tv.data_ptr(), a_ph.data_ptr(), h.numel(), p_tv, p_ph, seed, _stream())
    return a_tv, a_ph


def head_act_bwd(h, d_tv, d_ph, p_tv, p_ph, seed):
    _dev(h, d_tv, d_ph)
    dh = torch.empty_like(h)
    _lib.call("aptai_head_act_bwd", h.data_ptr(), d_tv.data_ptr(), d_ph.data_ptr(), dh.data_ptr(), h.numel(), p_tv, p_ph, seed,
              _stream())
    return dh


def lowpass_fir(x, ldx, rows_per_b_in, taps_f64, y, ldy, rows_per_b_out, B, T, T_out, C, C_out):
    _dev(x, taps_f64, y)
    _lib.call("aptai_lowpass_fir", x.data_ptr(), ldx, rows_per_b_in, taps_f64.data_ptr(), taps_f64.numel(), y.data_ptr(), ldy,
              rows_per_b_out, int(y.dtype == torch.bfloat16), B, T, T_out, C, C_out, _stream())
    return y


def aptai_loss_fwd(tv_pred, tv_tgt, logits, ldl, rows_per_b, phn_tgt, B, T, n_tv, n_phn, w_mse, w_ce, want_pred=True):
    _dev(tv_pred, tv_tgt, logits, phn_tgt)
    scalars = torch.empty(5, device=tv_pred.device, dtype=torch.float32)
    pred = torch.empty((B, T), device=tv_pred.device, dtype=torch.int64) if want_pred else None
    ws = _ws(_lib.lib().aptai_aptai_loss_workspace_bytes(), tv_pred.device)
    _lib.call("aptai_aptai_loss_fwd", tv_pred.data_ptr(), tv_tgt.data_ptr(), logits.data_ptr(), ldl, rows_per_b,
              phn_tgt.data_ptr(), B, T, n_tv, n_phn, w_mse, w_ce, scalars.data_ptr(), _ptr(pred), ws.data_ptr(), _stream())
    return scalars, pred


def aptai_loss_bwd(tv_pred, tv_tgt, logits, ldl, rows_per_b, phn_tgt, B, T, n_tv, n_phn, w_mse, w_ce, scalars, grad_out, ldd=64):
    _dev(tv_pred, tv_tgt, logits, phn_tgt, scalars, grad_out)
    d_tv = torch.empty((B, T, n_tv), device=tv_pred.device, dtype=torch.float32)
    d_logits = torch.empty((B * rows_per_b, ldd), device=tv_pred.device, dtype=torch.bfloat16)
    _lib.call("aptai_aptai_loss_bwd", tv_pred.data_ptr(), tv_tgt.data_ptr(), logits.data_ptr(), ldl, rows_per_b,
              phn_tgt.data_ptr(), B, T, n_tv, n_phn, w_mse, w_ce, scalars.data_ptr(), _ptr(grad_out), d_tv.data_ptr(),
              d_logits.data_ptr(), ldd, _stream())
    return d_tv, d_logits


# ----------------------------------------------------------------------------- CTC
_REDUCTION = {"none": 0, "mean": 1, "sum": 2}


def ctc_fwd(logits, ldl, rows_per_b, targets_i32, input_lens_i32, target_lens_i32, B, T, V, *, blank=0, reduction="mean",
            zero_infinity=True, vocab_sizes_i32=None, want_log_probs=True, want_beta=None):
    """Returns (loss scalar tensor, nll [B], log_probs (T,B,V) | None, workspace [alpha | beta | per-state log-probs]).
    want_beta (default: when gradients are enabled) also runs the beta recursion, beside alpha in the same launch."""
    _dev(logits, targets_i32, input_lens_i32, target_lens_i32, vocab_sizes_i32)
    dev = logits.device
    ldt = targets_i32.shape[1]
    alpha = torch.empty(_lib.lib().aptai_ctc_workspace_bytes(B, T, ldt) // 4, device=dev, dtype=torch.float32)
    nll = torch.empty(B, device=dev, dtype=torch.float32)
    loss = torch.zeros(1, device=dev, dtype=torch.float32)
    lp = torch.empty((T, B, V), device=dev, dtype=torch.float32) if want_log_probs else None
    if want_beta is None:
        want_beta = True
    _lib.call("aptai_ctc_fwd", logits.data_ptr(), ldl, rows_per_b, targets_i32.data_ptr(), ldt, input_lens_i32.data_ptr(),
              target_lens_i32.data_ptr(), _ptr(vocab_sizes_i32), B, T, V, blank, _REDUCTION[reduction], int(zero_infinity),
              _ptr(lp), alpha.data_ptr(), nll.data_ptr(), loss.data_ptr(), int(want_beta), _stream())
    alpha._beta_ready = bool(want_beta)
    return loss, nll, lp, alpha


def ctc_bwd(logits, ldl, rows_per_b, targets_i32, input_lens_i32, target_lens_i32, B, T, V, alpha, nll, grad_out, *, blank=0,
            reduction="mean", zero_infinity=True, vocab_sizes_i32=None, ldd=64, out_dtype=torch.bfloat16, extra_scale=1.0):
    _dev(logits, targets_i32, alpha, nll, grad_out)
    d = torch.empty((B * rows_per_b, ldd), device=logits.device, dtype=out_dtype)
    _lib.call("aptai_ctc_bwd", logits.data_ptr(), ldl, rows_per_b, targets_i32.data_ptr(), targets_i32.shape[1],
              input_lens_i32.data_ptr(), target_lens_i32.data_ptr(), _ptr(vocab_sizes_i32), B, T, V, blank, _REDUCTION[reduction],
              int(zero_infinity), alpha.data_ptr(), nll.data_ptr(), _ptr(grad_out), extra_scale, d.data_ptr(), ldd,
              int(out_dtype == torch.bfloat16), int(getattr(alpha, "_beta_ready", False)), _stream())
    return d


def ctc_greedy_decode(logits, ldl, rows_per_b, B, T, V, blank, max_n):
    """Device best-path decode: (ids int32 [B][max_n] zero-padded, n int32 [B]) - see aptai_ctc_greedy_decode."""
    _dev(logits)
    ids = torch.empty((B, max_n), device=logits.device, dtype=torch.int32)
    n = torch.empty(B, device=logits.device, dtype=torch.int32)
    _lib.call("aptai_ctc_greedy_decode", logits.data_ptr(), ldl, rows_per_b, B, T, V, blank, ids.data_ptr(), max_n, n.data_ptr(), _stream())
    return ids, n


# ----------------------------------------------------------------------------- Force_APTAI heads (fp32)
_SCRATCH32 = {}


def _scratch_f32(key, numel: int, device) -> torch.Tensor:
    """Persistent fp32 scratch per (purpose, device): split-K slabs and column-sum partials of the fp32 heads (stream-ordered
    reuse: every consumer of a scratch runs on the launch stream right after its producer)."""
    k = (key, str(device))
    t = _SCRATCH32.get(k)
    if t is None or t.numel() < numel:
        t = torch.empty(max(int(numel), 16), device=device, dtype=torch.float32)
        _SCRATCH32[k] = t
    return t


def sgemm(a, sam, sak, b, sbk, sbn, M, N, K, *, out=None, ldc=None, bias=None, alpha=1.0, accumulate=False, batch=1, bsa=0, bsb=0,
          bsc=0, split_k=None):
    """C[m][n] (+)= alpha * sum_k A(m,k) B(k,n) + bias[n] with explicit element strides (see aptai_sgemm_f32).
    split_k=None picks the K split that fills the chip for gradient-shaped problems (small M x N, long K)."""
    _dev(a, b, out, bias)
    if out is None:
        out = torch.empty((batch * M, N) if batch > 1 else (M, N), device=a.device, dtype=torch.float32)
        if batch > 1 and bsc == 0:
            bsc = M * N
    if split_k is None:
        tiles = ((M + 63) // 64) * ((N + 63) // 64) * batch
        split_k = 1 if (tiles >= 256 or K < 256) else max(1, min(K // 128, 512 // tiles))
    ws = None
    if split_k > 1:
        ws = _scratch_f32("sgemm", batch * split_k * M * N, a.device)
    _lib.call("aptai_sgemm_f32", a.data_ptr(), int(a.dtype == torch.bfloat16), sam, sak, b.data_ptr(), sbk, sbn, out.data_ptr(),
              ldc if ldc else N, _ptr(bias), alpha, int(accumulate), M, N, K, batch, bsa, bsb, bsc, split_k, _ptr(ws), _stream())
    return out


def linear_f32(x, w, bias=None, *, rows=None, out=None, ldc=None, ldx=None):
    """y = x W^T + b, x [rows][in] (fp32 or bf16, row stride ldx), W [out][in] fp32."""
    N, K = w.shape
    M = rows if rows is not None else x.shape[0]
    return sgemm(x, ldx if ldx else x.stride(0), 1, w, 1, K, M, N, K, out=out, ldc=ldc, bias=bias)


def embed_pe_fwd(ids_i32, emb, pe, N, p, seed):
    rows, D = ids_i32.numel(), emb.shape[1]
    out = torch.empty((rows, D), device=emb.device, dtype=torch.float32)
    _lib.call("aptai_embed_pe_fwd", ids_i32.data_ptr(), emb.data_ptr(), pe.data_ptr(), out.data_ptr(), rows, N, D, p, seed, _stream())
    return out


def embed_bwd(ids_i32, dout, vocab, p, seed):
    rows, D = dout.shape
    demb = torch.zeros((vocab, D), device=dout.device, dtype=torch.float32)
    _lib.call("aptai_embed_bwd", ids_i32.data_ptr(), dout.data_ptr(), demb.data_ptr(), rows, D, p, seed, _stream())
    return demb


def xattn_softmax_fwd(raw, ids_i32, B, T, N, fs_rows=None):
    """fs_rows (optional [B*T][64] fp32): also writes the forward-sum CTC input rows [-1 | att_log | 0]."""
    dev = raw.device
    energy, att, att_log = (torch.empty((B * T, N), device=dev, dtype=torch.float32) for _ in range(3))
    align = torch.empty((B, T), device=dev, dtype=torch.int64)
    _lib.call("aptai_xattn_softmax_fwd", raw.data_ptr(), ids_i32.data_ptr(), energy.data_ptr(), att.data_ptr(), att_log.data_ptr(),
              align.data_ptr(), _ptr(fs_rows), B, T, N, _stream())
    return energy, att, att_log, align


def xattn_softmax_bwd(att, att_log, d_att, d_attlog, ld_dattlog=0):
    """d_attlog may be a strided view (e.g. columns 1..N of the 64-float forward-sum gradient rows: pass ld_dattlog=64)."""
    rows, N = att.shape
    d_raw = torch.empty_like(att)
    _lib.call("aptai_xattn_softmax_bwd", att.data_ptr(), att_log.data_ptr(), _ptr(d_att), _ptr(d_attlog), ld_dattlog, d_raw.data_ptr(),
              rows, N, _stream())
    return d_raw


def layernorm_f32_fwd(x, gamma, beta, eps=1e-5):
    rows, cols = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(rows, device=x.device, dtype=torch.float32)
    rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
    _lib.call("aptai_layernorm_f32_fwd", x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mean.data_ptr(),
              rstd.data_ptr(), rows, cols, eps, _stream())
    return y, mean, rstd


def layernorm_f32_bwd(dy, x, mean, rstd, gamma):
    rows, cols = x.shape
    dx = torch.empty_like(x)
    dg = torch.empty(cols, device=x.device, dtype=torch.float32)
    db = torch.empty(cols, device=x.device, dtype=torch.float32)
    ws = _scratch_f32("ln32_bwd", 256 * 2 * cols, x.device)
    _lib.call("aptai_layernorm_f32_bwd", dy.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
              dx.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(), rows, cols, _stream())
    return dx, dg, db


_LSTM_WS = {}


def _lstm_workspace(B: int, device) -> torch.Tensor:
    """Exchange area of the cooperating LSTM workgroups + status word, one per (device, STREAM): two Force_APTAI models, or one
    model driven from two streams, never share an exchange area (zeroed once here; every launch re-zeroes what it uses)."""
    n = _lib.lib().aptai_lstm_workspace_bytes(B)
    key = (str(device), _stream())
    ws = _LSTM_WS.get(key)
    if ws is None or ws.numel() < n:
        ws = torch.zeros(n, device=device, dtype=torch.uint8)
        _LSTM_WS[key] = ws
    return ws


def lstm_status_words(device):
    """int32 device tensor with the status word of every LSTM workspace of this device (None if no LSTM has run): lets a caller
    fold the check into a device->host read it makes anyway (Force_APTAI._lists)."""
    words = [ws[:4].view(torch.int32) for (d, _), ws in _LSTM_WS.items() if d == str(device)]
    if not words:
        return None
    return words[0] if len(words) == 1 else torch.cat(words)


def lstm_check(status_host, device) -> None:
    """Raises when a bounded wait of the cooperating LSTM kernels timed out (their output is then incomplete); clears the
    status words so that the next step is judged on its own."""
    if status_host is not None and any(int(v) != 0 for v in status_host):
        for (d, _), ws in _LSTM_WS.items():
            if d == str(device):
                ws[:4].zero_()
        raise _lib.AptaiHipError("aptai_lstm: a cross-workgroup wait of the cooperative BiLSTM kernels timed out "
                                 "(not all 16 workgroups of a cluster were resident together?): hout / dgates are incomplete")


def lstm_status(device) -> int:
    """Non-zero if a bounded wait of the cooperating LSTM kernels ever timed out on this device (synchronises)."""
    w = lstm_status_words(torch.device(device) if not isinstance(device, torch.device) else device)
    return 0 if w is None else int(w.abs().max().item())


_GATE_PERM = {}


def lstm_gate_perm(device):
    """(perm, inv) int64 [2048]: the gate-INTERLEAVED column order of aptai_lstm_fwd / _bwd (xproj, gates, dgates: column
    dir * 1024 + unit * 4 + gate) against torch's gate-major order (dir * 1024 + gate * 256 + unit).  `w[perm]` puts the rows of
    cat(weight_ih_l0, weight_ih_l0_reverse) (or a gate-major vector) into the kernels' order; `t[..., inv]` (or `g[inv]` on rows) brings a
    tensor in the kernels' order back to torch's."""
    key = str(device)
    if key not in _GATE_PERM:
        d, u, g = torch.meshgrid(torch.arange(2), torch.arange(256), torch.arange(4), indexing="ij")
        perm = (d * 1024 + g * 256 + u).reshape(-1)                  # position dir*1024 + unit*4 + gate <- gate-major index
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(2048)
        _GATE_PERM[key] = (perm.to(device), inv.to(device))
    return _GATE_PERM[key]


def lstm_fwd(xproj, whh, lens_i32, B, Tp, T, save=True):
    """xproj in the gate-interleaved column order (lstm_gate_perm); whh [2][1024][256] (weight_hh_l0 | weight_hh_l0_reverse, as stored).
    Returns (hout [B*Tp][512], gates, cstate); gates in the interleaved order."""
    _dev(xproj, whh, lens_i32)
    dev = xproj.device
    hout = torch.empty((B * Tp, 512), device=dev, dtype=torch.float32)
    gates = torch.empty((B * Tp, 2048), device=dev, dtype=torch.float32) if save else None
    cst = torch.empty((B * Tp, 512), device=dev, dtype=torch.float32) if save else None
    ws = _lstm_workspace(B, dev)
    _lib.call("aptai_lstm_fwd", xproj.data_ptr(), whh.data_ptr(), lens_i32.data_ptr(), hout.data_ptr(), _ptr(gates), _ptr(cst),
              ws.data_ptr(), B, Tp, T, 256, _stream())
    return hout, gates, cst


def lstm_bwd(dhout, whh, lens_i32, gates, cst, B, Tp, T):
    _dev(dhout, whh, lens_i32, gates, cst)
    dgates = torch.empty_like(gates)
    ws = _lstm_workspace(B, dhout.device)
    _lib.call("aptai_lstm_bwd", dhout.data_ptr(), whh.data_ptr(), lens_i32.data_ptr(), gates.data_ptr(), cst.data_ptr(),
              dgates.data_ptr(), ws.data_ptr(), B, Tp, T, 256, _stream())
    return dgates


def lstm_fwd_serial(xproj, whhT, lens_i32, B, Tp, T, save=True):
    """One block per (utterance, direction): cross-check of lstm_fwd (aptai_lstm_fwd_serial); whhT [2][256][1024]."""
    dev = xproj.device
    hout = torch.empty((B * Tp, 512), device=dev, dtype=torch.float32)
    gates = torch.zeros((B * Tp, 2048), device=dev, dtype=torch.float32) if save else None
    cst = torch.zeros((B * Tp, 512), device=dev, dtype=torch.float32) if save else None
    _lib.call("aptai_lstm_fwd_serial", xproj.data_ptr(), whhT.data_ptr(), lens_i32.data_ptr(), hout.data_ptr(), _ptr(gates), _ptr(cst),
              B, Tp, T, 256, _stream())
    return hout, gates, cst


def lstm_bwd_serial(dhout, whh, lens_i32, gates, cst, B, Tp, T):
    dgates = torch.empty_like(gates)
    _lib.call("aptai_lstm_bwd_serial", dhout.data_ptr(), whh.data_ptr(), lens_i32.data_ptr(), gates.data_ptr(), cst.data_ptr(),
              dgates.data_ptr(), B, Tp, T, 256, _stream())
    return dgates


def gather_alignment(ids_i32, align, lens_i32, B, T, N):
    out = torch.empty((B, T), device=align.device, dtype=torch.int64)
    _lib.call("aptai_gather_alignment", ids_i32.data_ptr(), align.data_ptr(), lens_i32.data_ptr(), out.data_ptr(), B, T, N, _stream())
    return out


def tanh_dropout_fwd(x, p, seed):
    y = torch.empty_like(x)
    _lib.call("aptai_tanh_dropout_f32", x.data_ptr(), None, None, y.data_ptr(), x.numel(), p, seed, _stream())
    return y


def tanh_dropout_bwd(y, dy, p, seed):
    dx = torch.empty_like(y)
    _lib.call("aptai_tanh_dropout_f32", None, y.data_ptr(), dy.data_ptr(), dx.data_ptr(), y.numel(), p, seed, _stream())
    return dx


def dropout_f32(x, p, seed):
    if p <= 0:
        return x
    y = torch.empty_like(x)
    _lib.call("aptai_dropout_f32", x.data_ptr(), y.data_ptr(), x.numel(), p, seed, _stream())
    return y


def colsum_f32(x, rows, N, ld=None):
    out = torch.empty(N, device=x.device, dtype=torch.float32)
    ws = _scratch_f32("colsum", 64 * N, x.device)
    _lib.call("aptai_colsum_f32", x.data_ptr(), ld if ld else x.stride(0), out.data_ptr(), ws.data_ptr(), rows, N, _stream())
    return out
