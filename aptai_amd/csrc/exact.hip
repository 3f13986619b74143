// fp32-class INFERENCE path of the encoder (Wav2Vec2Model.set_encoder_precision("f32x3" | "f32x6"); gfx950).
//
// Force_APTAI's alignment read-out is an argmax over 60 log-attention scores per frame (models/force_aptai.py:148-161), and
// BASELINE's north star asks for bit-exact alignment indices.  With bf16 GEMM operands in the frozen encoder 2-13 % of the
// decisions sit inside the arithmetic noise.  This file holds what the exact mode needs around the bf16 MFMA GEMM kernels,
// which stay the workhorse:
//   * split operands.  An fp32 value is split into bf16 pieces hi = bf16(x), lo = bf16(x - hi) [, then a third], and the
//     product x . w is evaluated as the exact bf16 x bf16 products hi.hi + hi.lo + lo.hi (3 pieces, relative error ~2^-17)
//     or additionally mid.mid + hi.low + low.hi on a three-way split (6 pieces, ~2^-24), accumulated in fp32 by the MFMA.
//     The pieces are laid out so that the ORDINARY NT kernel runs them as one GEMM of K' = pieces * K: K is cut into 64-wide
//     tiles and each tile is followed by its own pieces, [row][K/64][piece][64].  An activation ("A" pattern: hi,hi,lo |
//     hi,hi,mid,mid,hi,low) meets a weight ("B" pattern: hi,lo,hi | hi,mid,hi,mid,low,hi) piece by piece.  The conv stack's
//     overlapping-row trick survives (a frame is C/64 whole tiles), and 6 x the bf16 rate still beats the fp32 matrix
//     pipe (1/16 of it);
//   * exact element-wise arithmetic in fp32 between the GEMMs: erf-form GELU (the training path's logistic fit has 3e-5
//     error), bias / residual / padded-frame masking, the first conv layer with fp32 output, the attention softmax.
// Attention itself (QK^T, PV) and the positional conv run on the fp32 matrix instruction (aptai_sgemm_f32, force.hip).
#include "common.h"

namespace {

struct SplitArgs {
    const float* x; long ldx;
    bf16_t* out; long ldo;
    long rows; int cols;
    int pattern;            // 0 = activation side (A), 1 = weight side (B)
    int pieces;             // 3 or 6
    int act;                // 0 none, 1 erf-GELU before the split
};

// one thread = 8 consecutive columns of one row (inside one 64-wide K-tile): 2 x 16-byte loads, `pieces` x 16-byte stores
__global__ __launch_bounds__(256) void split_kernel(SplitArgs a) {
    const long chunks_per_row = a.cols / 8;
    const long total = a.rows * chunks_per_row;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long row = i / chunks_per_row;
        const int c0 = (int)(i - row * chunks_per_row) * 8;
        const float* src = a.x + row * a.ldx + c0;
        const f32x4 v0 = *(const f32x4*)src, v1 = *(const f32x4*)(src + 4);
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        float h[8], m[8], l[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (a.act == 1) v[r] = gelu_exact(v[r]);
            split3(v[r], h[r], m[r], l[r]);
        }
        bf16_t* dst = a.out + row * a.ldo + (long)(c0 >> 6) * (64 * a.pieces) + (c0 & 63);
        auto put = [&](int piece, const float (&p)[8]) {
            *(u32x4*)(dst + piece * 64) = (u32x4){pack2bf(p[0], p[1]), pack2bf(p[2], p[3]), pack2bf(p[4], p[5]), pack2bf(p[6], p[7])};
        };
        if (a.pieces == 3) {            // A: hi hi lo      B: hi lo hi
            put(0, h);
            if (a.pattern == 0) { put(1, h); put(2, m); } else { put(1, m); put(2, h); }
        } else {                        // A: h h m m h l   B: h m h m l h
            put(0, h);
            if (a.pattern == 0) { put(1, h); put(2, m); put(3, m); put(4, h); put(5, l); }
            else { put(1, m); put(2, h); put(3, m); put(4, l); put(5, h); }
        }
    }
}

struct EwArgs {
    const float* x; long ldx;
    const float* bias;
    const float* res; long ldr;
    float* y; long ldy;
    long rows; int cols;
    int act;
    const int* lens; long rows_per_b;
};

// y = [res +] act(x + bias); rows at or beyond lens[b] inside each block of rows_per_b rows become 0 (padded frames, HF:678-681)
__global__ __launch_bounds__(256) void ew_kernel(EwArgs a) {
    const long chunks_per_row = a.cols / 4;
    const long total = a.rows * chunks_per_row;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long row = i / chunks_per_row;
        const int c0 = (int)(i - row * chunks_per_row) * 4;
        f32x4 v = *(const f32x4*)(a.x + row * a.ldx + c0);
        if (a.bias) v += *(const f32x4*)(a.bias + c0);
        if (a.act == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = gelu_exact(v[r]);
        }
        if (a.res) v += *(const f32x4*)(a.res + row * a.ldr + c0);
        if (a.lens) {
            const long b = row / a.rows_per_b;
            if (row - b * a.rows_per_b >= a.lens[b]) v = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        *(f32x4*)(a.y + row * a.ldy + c0) = v;
    }
}

// softmax over the keys of one query row, in place: s[b][h][q][k], keys k >= lens[b] masked (HF:452-461 with the additive
// finfo.min mask of HF:1018-1036: their probabilities are exactly 0).  One wave per row.
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ s, const int* __restrict__ lens, long rows_per_b,
                                                           long rows, int Tp) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int len = lens[row / rows_per_b];
    float* p = s + row * Tp;
    float mx = -INFINITY;
    for (int k = lane; k < len; k += 64) mx = fmaxf(mx, p[k]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int k = lane; k < len; k += 64) {
        const float e = expf(p[k] - mx);
        p[k] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int k = lane; k < Tp; k += 64) p[k] = k < len ? p[k] * inv : 0.f;
}

// softmax of one query row straight into the split (activation-side) layout of the probabilities: the P . V product of the exact
// attention reads its A operand as bf16 pieces, so the fp32 probabilities are never stored (round 4: softmax in place + split_kernel
// moved 1.1 GB per layer at 16 x 10 s, this pass 0.5 GB).  One wave per row; a lane owns 8 consecutive keys per 512-key chunk
// (two 16-byte loads, `pieces` 16-byte stores, exactly split_kernel's addressing).  Same arithmetic as softmax_rows_kernel + split3.
template <int NCH>
__global__ __launch_bounds__(256) void softmax_split_kernel(const float* __restrict__ s, const int* __restrict__ lens, long rows_per_b,
                                                            long rows, int Tp, int pieces, bf16_t* __restrict__ out, long ldo) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int len = lens[row / rows_per_b];
    const float* p = s + row * Tp;
    float v[NCH][8];
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int k0 = (c * 64 + lane) * 8;
        if (k0 < Tp) {
            const f32x4 a0 = *(const f32x4*)(p + k0), a1 = *(const f32x4*)(p + k0 + 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[c][r] = a0[r]; v[c][4 + r] = a1[r]; }
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[c][r] = 0.f;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) if (k0 + r < len) mx = fmaxf(mx, v[c][r]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int k0 = (c * 64 + lane) * 8;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float e = (k0 + r < len) ? expf(v[c][r] - mx) : 0.f;
            v[c][r] = e;
            sum += e;
        }
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int k0 = (c * 64 + lane) * 8;
        if (k0 >= Tp) continue;
        float h[8], m[8], l[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) split3((k0 + r < len) ? v[c][r] * inv : 0.f, h[r], m[r], l[r]);
        bf16_t* dst = out + row * ldo + (long)(k0 >> 6) * (64 * pieces) + (k0 & 63);
        auto put = [&](int piece, const float (&q)[8]) {
            *(u32x4*)(dst + piece * 64) = (u32x4){pack2bf(q[0], q[1]), pack2bf(q[2], q[3]), pack2bf(q[4], q[5]), pack2bf(q[6], q[7])};
        };
        put(0, h);
        if (pieces == 3) { put(1, h); put(2, m); }
        else { put(1, h); put(2, m); put(3, m); put(4, h); put(5, l); }
    }
}

// ---- first conv layer with fp32 output and the erf GELU (one wave per frame, lane = 8 channels; the training-path kernel of
// conv.hip writes bf16 and uses the logistic GELU).  mode 0: GroupNorm statistics (mean, rstd per (b, channel)) from `stats`;
// mode 1: LayerNorm over the 512 channels of the frame.
constexpr int C0 = 512, KW = 10, STRIDE = 5;
struct Conv0xArgs {
    const float* audio; long S;
    const float* w; const float* bias; const float* gamma; const float* beta; const float* stats;
    float* out; int T_real, T_alloc, mode;
    float eps;
    bf16_t* out_s; int pieces;      // != null: the result leaves as split bf16 pieces [frame][512 / 64][piece][64] instead of fp32
};
__global__ __launch_bounds__(256) void conv0_exact_kernel(Conv0xArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const float* xb = a.audio + (long)b * a.S;
    float* ob = a.out + (long)b * a.T_alloc * C0;      // (unused when the result leaves split)
    // taps, bias and the affine of this lane's 8 channels live in registers for the whole launch (round 4: the first form re-loaded the
    // 80 taps and the per-channel parameters from memory for every frame: 2.5 ms per call at 16 x 10 s, 8 % of the exact-mode step)
    float w[8][KW], bs[8], sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = lane * 8 + j;
#pragma unroll
        for (int k = 0; k < KW; ++k) w[j][k] = a.w[c * KW + k];
        bs[j] = a.bias ? a.bias[c] : 0.f;
        if (a.mode == 0) {                       // GroupNorm: (v - mean) * rstd * gamma + beta with the statistics of (b, channel), as written
            sc[j] = a.stats[((long)b * 2 + 1) * C0 + c];
            sh[j] = a.stats[((long)b * 2 + 0) * C0 + c];
        } else {
            sc[j] = 0.f; sh[j] = 0.f;
        }
    }
    float gm[8], bt[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { gm[j] = a.gamma[lane * 8 + j]; bt[j] = a.beta[lane * 8 + j]; }
    for (int t = blockIdx.x * 4 + wave; t < a.T_alloc; t += gridDim.x * 4) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = 0.f;
        if (t < a.T_real) {
            float xs[KW], v[8];
#pragma unroll
            for (int k = 0; k < KW; ++k) xs[k] = xb[(long)t * STRIDE + k];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < KW; ++k) acc = fmaf(xs[k], w[j][k], acc);
                v[j] = acc + bs[j];
            }
            if (a.mode == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = gelu_exact((v[j] - sh[j]) * sc[j] * gm[j] + bt[j]);
            } else {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) s += v[j];
                const float mu = wave_sum(s) * (1.0f / C0);
                float q = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float d = v[j] - mu; q += d * d; }
                const float rs = 1.0f / sqrtf(wave_sum(q) * (1.0f / C0) + a.eps);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = gelu_exact((v[j] - mu) * rs * gm[j] + bt[j]);
            }
        }
        if (a.out_s != nullptr) {                            // lane * 8 .. + 7 lie inside one 64-channel K-tile of the split layout
            float h[8], md[8], lw[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) split3(o[j], h[j], md[j], lw[j]);
            const int c0 = lane * 8;
            bf16_t* dst = a.out_s + ((long)b * a.T_alloc + t) * (C0 * a.pieces) + (long)(c0 >> 6) * (64 * a.pieces) + (c0 & 63);
            auto put = [&](int piece, const float (&q)[8]) {
                *(u32x4*)(dst + piece * 64) = (u32x4){pack2bf(q[0], q[1]), pack2bf(q[2], q[3]), pack2bf(q[4], q[5]), pack2bf(q[6], q[7])};
            };
            put(0, h); put(1, h); put(2, md);
            if (a.pieces == 6) { put(3, md); put(4, h); put(5, lw); }
            continue;
        }
        float* dst = ob + (long)t * C0 + lane * 8;
        *(f32x4*)dst = (f32x4){o[0], o[1], o[2], o[3]};
        *(f32x4*)(dst + 4) = (f32x4){o[4], o[5], o[6], o[7]};
    }
}

}  // namespace

extern "C" int aptai_split_f32(const float* x, int64_t ldx, int64_t rows, int64_t cols, int pattern, int pieces, int act, void* out,
                               int64_t ldo, void* stream) {
    APTAI_REQUIRE(x && out, "aptai_split_f32: null pointer");
    APTAI_REQUIRE(rows > 0 && cols > 0 && cols % 64 == 0, "aptai_split_f32: cols=%ld must be a positive multiple of 64", (long)cols);
    APTAI_REQUIRE(pieces == 3 || pieces == 6, "aptai_split_f32: pieces must be 3 or 6");
    APTAI_REQUIRE(pattern == 0 || pattern == 1, "aptai_split_f32: pattern must be 0 (activation) or 1 (weight)");
    APTAI_REQUIRE(ldx % 4 == 0 && ldo % 8 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)out % 16 == 0, "aptai_split_f32: 16-byte alignment");
    APTAI_REQUIRE(act == 0 || act == 1, "aptai_split_f32: act must be 0 or 1");
    SplitArgs a{x, (long)ldx, (bf16_t*)out, (long)ldo, (long)rows, (int)cols, pattern, pieces, act};
    const long total = rows * (cols / 8);
    const int blocks = (int)(total / 256 < 4096 ? (total + 255) / 256 : 4096);
    APTAI_LAUNCH(split_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    APTAI_CHECK_LAUNCH("split_kernel");
    return APTAI_OK;
}

extern "C" int aptai_bias_act_res_f32(const float* x, int64_t ldx, const float* bias, const float* res, int64_t ldr, float* y,
                                      int64_t ldy, int64_t rows, int64_t cols, int act, const int32_t* lens, int64_t rows_per_b,
                                      void* stream) {
    APTAI_REQUIRE(x && y, "aptai_bias_act_res_f32: null pointer");
    APTAI_REQUIRE(rows > 0 && cols > 0 && cols % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && (res == nullptr || ldr % 4 == 0),
                  "aptai_bias_act_res_f32: cols and leading dimensions must be multiples of 4");
    APTAI_REQUIRE(act == 0 || act == 1, "aptai_bias_act_res_f32: act must be 0 or 1");
    if (lens) APTAI_REQUIRE(rows_per_b > 0 && rows % rows_per_b == 0, "aptai_bias_act_res_f32: rows must be whole blocks of rows_per_b");
    EwArgs a{x, (long)ldx, bias, res, (long)ldr, y, (long)ldy, (long)rows, (int)cols, act, lens, (long)rows_per_b};
    const long total = rows * (cols / 4);
    const int blocks = (int)(total / 256 < 4096 ? (total + 255) / 256 : 4096);
    APTAI_LAUNCH(ew_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    APTAI_CHECK_LAUNCH("ew_kernel");
    return APTAI_OK;
}

extern "C" int aptai_softmax_rows_f32(float* s, const int32_t* lens, int64_t B, int64_t heads, int64_t Tp, void* stream) {
    APTAI_REQUIRE(s && lens && B > 0 && heads > 0 && Tp > 0, "aptai_softmax_rows_f32: bad arguments");
    const long rows = B * heads * Tp;
    APTAI_LAUNCH(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, s, lens, (long)(heads * Tp), rows,
                 (int)Tp);
    APTAI_CHECK_LAUNCH("softmax_rows_kernel");
    return APTAI_OK;
}

extern "C" int aptai_softmax_split_f32(const float* s, const int32_t* lens, int64_t B, int64_t heads, int64_t Tp, int pieces, void* out,
                                       int64_t ldo, void* stream) {
    APTAI_REQUIRE(s && lens && out && B > 0 && heads > 0, "aptai_softmax_split_f32: bad arguments");
    APTAI_REQUIRE(Tp > 0 && Tp % 64 == 0 && Tp <= 2048, "aptai_softmax_split_f32: Tp=%ld must be a multiple of 64, at most 2048", (long)Tp);
    APTAI_REQUIRE(pieces == 3 || pieces == 6, "aptai_softmax_split_f32: pieces must be 3 or 6");
    APTAI_REQUIRE(ldo >= Tp * pieces && ldo % 8 == 0 && (uintptr_t)s % 16 == 0 && (uintptr_t)out % 16 == 0, "aptai_softmax_split_f32: layout");
    const long rows = B * heads * Tp;
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    const int nch = (int)((Tp + 511) / 512);
    if (nch == 1) APTAI_LAUNCH(softmax_split_kernel<1>, grid, block, 0, (hipStream_t)stream, s, lens, (long)(heads * Tp), rows, (int)Tp, pieces, (bf16_t*)out, (long)ldo);
    else if (nch == 2) APTAI_LAUNCH(softmax_split_kernel<2>, grid, block, 0, (hipStream_t)stream, s, lens, (long)(heads * Tp), rows, (int)Tp, pieces, (bf16_t*)out, (long)ldo);
    else if (nch == 3) APTAI_LAUNCH(softmax_split_kernel<3>, grid, block, 0, (hipStream_t)stream, s, lens, (long)(heads * Tp), rows, (int)Tp, pieces, (bf16_t*)out, (long)ldo);
    else APTAI_LAUNCH(softmax_split_kernel<4>, grid, block, 0, (hipStream_t)stream, s, lens, (long)(heads * Tp), rows, (int)Tp, pieces, (bf16_t*)out, (long)ldo);
    APTAI_CHECK_LAUNCH("softmax_split_kernel");
    return APTAI_OK;
}

extern "C" int aptai_conv0_fwd_f32(const float* audio, int64_t B, int64_t S, const float* weight, const float* bias, const float* gamma,
                                   const float* beta, int mode, float eps, float* out, int64_t T_real, int64_t T_alloc,
                                   const float* stats, void* stream) {
    APTAI_REQUIRE(audio && weight && gamma && beta && out, "aptai_conv0_fwd_f32: null pointer");
    APTAI_REQUIRE(mode == 0 || mode == 1, "aptai_conv0_fwd_f32: mode must be 0 (group) or 1 (layer)");
    APTAI_REQUIRE(mode == 1 || stats != nullptr, "aptai_conv0_fwd_f32: group mode needs the (mean, rstd) block of aptai_conv0_fwd");
    APTAI_REQUIRE(T_real >= 1 && T_alloc >= T_real && (T_real - 1) * STRIDE + KW <= S, "aptai_conv0_fwd_f32: frames exceed the waveform");
    Conv0xArgs a{audio, (long)S, weight, bias, gamma, beta, stats, out, (int)T_real, (int)T_alloc, mode, eps, nullptr, 0};
    const unsigned bx = (unsigned)((T_alloc + 3) / 4 < 2048 ? (T_alloc + 3) / 4 : 2048);
    APTAI_LAUNCH(conv0_exact_kernel, dim3(bx, (unsigned)B), dim3(256), 0, (hipStream_t)stream, a);
    APTAI_CHECK_LAUNCH("conv0_exact_kernel");
    return APTAI_OK;
}

extern "C" int aptai_conv0_fwd_split(const float* audio, int64_t B, int64_t S, const float* weight, const float* bias, const float* gamma,
                                     const float* beta, int mode, float eps, void* out_split, int pieces, int64_t T_real, int64_t T_alloc,
                                     const float* stats, void* stream) {
    APTAI_REQUIRE(audio && weight && gamma && beta && out_split, "aptai_conv0_fwd_split: null pointer");
    APTAI_REQUIRE(mode == 0 || mode == 1, "aptai_conv0_fwd_split: mode must be 0 (group) or 1 (layer)");
    APTAI_REQUIRE(mode == 1 || stats != nullptr, "aptai_conv0_fwd_split: group mode needs the (mean, rstd) block of aptai_conv0_fwd");
    APTAI_REQUIRE(pieces == 3 || pieces == 6, "aptai_conv0_fwd_split: pieces must be 3 or 6");
    APTAI_REQUIRE(T_real >= 1 && T_alloc >= T_real && (T_real - 1) * STRIDE + KW <= S, "aptai_conv0_fwd_split: frames exceed the waveform");
    Conv0xArgs a{audio, (long)S, weight, bias, gamma, beta, stats, nullptr, (int)T_real, (int)T_alloc, mode, eps, (bf16_t*)out_split, pieces};
    const unsigned bx = (unsigned)((T_alloc + 3) / 4 < 2048 ? (T_alloc + 3) / 4 : 2048);
    APTAI_LAUNCH(conv0_exact_kernel, dim3(bx, (unsigned)B), dim3(256), 0, (hipStream_t)stream, a);
    APTAI_CHECK_LAUNCH("conv0_exact_kernel");
    return APTAI_OK;
}
