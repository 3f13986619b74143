// First layer of the wav2vec2 feature encoder on the raw waveform (HF:260-266 conv, C_in=1 -> 512, k=10, s=5),
// fused with its normalisation and GELU so the 32.8 MB/utterance (10 s) activation is written exactly once:
//   mode 0 "group": GroupNorm(512 groups of 1 channel) over all T0 frames (HF:317-323, wav2vec2-base)
//                   -> per-(utterance, channel) statistics from 65 window moments of the waveform, then ONE conv pass writes.
//   mode 1 "layer": LayerNorm(512) over channels per frame (HF:288-299, wav2vec2-large), single pass.
// HBM-bound by the output write (512 ch x 2 B per frame; the waveform read is 20 B per frame).
// One wave per frame, lane = 8 consecutive channels (16-byte coalesced stores); weights live in registers.
// Layers 1..6 are strided-row implicit GEMMs (gemm.hip).
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int C0 = 512, KW = 10, STRIDE = 5;
constexpr int FRAMES_PER_BLOCK_STATS = 256;

struct Conv0Args {
    const float* audio; long S;
    const float* w; const float* bias; const float* gamma; const float* beta;
    bf16_t* out;
    int B, T_real, T_alloc;
    float eps;
    float* partials;       // [B][nchunks][2][512]
    const float* stats;    // [B][2][512]  (mean, rstd)
    int nchunks;
    const int* bounds;     // bucketed hipGraphs (runtime.hip, aptai_set_frame_bounds): bounds[0] = frames of the batch AS COLLATED, or null
};

// Frames that exist in the batch as the reference collated it.  A graph captured for a bucket length sees T_real = the bucket's frames;
// the window of frame T_batch still covers 5..9 real samples of an utterance that fills the batch (it starts at sample 5 T_batch <
// S_batch), but that frame exists neither in the reference nor in the eager run: it must enter neither the GroupNorm statistics nor the
// weight gradients (round-3 advice: the static-length sums were off by that one frame, ~1/T0 relative).
__device__ __forceinline__ int frames_collated(const int* __restrict__ bounds, int T_real) {
    if (bounds == nullptr) return T_real;
    const int tb = bounds[0];
    return tb < 1 ? 1 : (tb < T_real ? tb : T_real);
}

__device__ __forceinline__ void load_weights(const Conv0Args& a, int lane, float (&w)[8][KW], float (&bias)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = lane * 8 + j;
#pragma unroll
        for (int k = 0; k < KW; ++k) w[j][k] = a.w[c * KW + k];
        bias[j] = a.bias ? a.bias[c] : 0.f;
    }
}

__device__ __forceinline__ void conv_frame(const float* __restrict__ x, const float (&w)[8][KW], const float (&bias)[8],
                                           float (&v)[8]) {
    float s[KW];
#pragma unroll
    for (int k = 0; k < KW; ++k) s[k] = x[k];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float acc = bias[j];
#pragma unroll
        for (int k = 0; k < KW; ++k) acc = fmaf(s[k], w[j][k], acc);
        v[j] = acc;
    }
}

__device__ __forceinline__ void store8(bf16_t* dst, const float (&o)[8]) {
    *(u32x4*)dst = (u32x4){pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7])};
}

// ---- mode 1: conv + bias -> LayerNorm(512) -> GELU, one pass.  grid (blocks, B)
__global__ __launch_bounds__(256) void conv0_layer_kernel(Conv0Args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    float w[8][KW], bias[8], gm[8], bt[8];
    load_weights(a, lane, w, bias);
#pragma unroll
    for (int j = 0; j < 8; ++j) { gm[j] = a.gamma[lane * 8 + j]; bt[j] = a.beta[lane * 8 + j]; }
    const float* xb = a.audio + (long)b * a.S;
    bf16_t* ob = a.out + (long)b * a.T_alloc * C0;
    // next frame's samples prefetched under the current frame's arithmetic (see conv0_group_kernel)
    const int tstep = gridDim.x * 4;
    int t = blockIdx.x * 4 + wave;
    float xs[KW];
#pragma unroll
    for (int k = 0; k < KW; ++k) xs[k] = 0.f;
    if (t < a.T_real) {
#pragma unroll
        for (int k = 0; k < KW; ++k) xs[k] = xb[(long)t * STRIDE + k];
    }
    for (; t < a.T_alloc; t += tstep) {
        float xn[KW];
        const int tn = t + tstep;
        const float* xp = xb + (long)(tn < a.T_real ? tn : 0) * STRIDE;      // clamped: always in bounds
#pragma unroll
        for (int k = 0; k < KW; ++k) xn[k] = xp[k];
        float o[8];
        if (t < a.T_real) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float acc = bias[j];
#pragma unroll
                for (int k = 0; k < KW; ++k) acc = fmaf(xs[k], w[j][k], acc);
                v[j] = acc;
            }
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];
            const float mu = wave_sum(s) * (1.0f / C0);
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = v[j] - mu; q += d * d; }
            const float rs = rsqrtf(wave_sum(q) * (1.0f / C0) + a.eps);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = gelu_fast((v[j] - mu) * rs * gm[j] + bt[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = 0.f;
        }
        store8(ob + (long)t * C0 + lane * 8, o);
#pragma unroll
        for (int k = 0; k < KW; ++k) xs[k] = xn[k];
    }
}

// ---- mode 0 statistics WITHOUT a conv pass.  y_c[t] = b_c + sum_k w_c[k] x[5t+k], so over the frames of one utterance
//   mean_t y_c   = b_c + w_c . m                      m[k]     = mean_t x[5t+k]
//   mean_t y_c^2 = w_c^T R w_c + 2 b_c (w_c . m) + b_c^2   R[k][k'] = mean_t x[5t+k] x[5t+k']
// i.e. the 512 channel statistics are quadratic forms in 10 + 55 window moments of the waveform: 65 FMAs per frame
// instead of 5120 (the recompute pass this replaces took 107 us of the 420 us the first layer cost at 16 x 10 s).
constexpr int AC_TERMS = KW + KW * (KW + 1) / 2;        // 65
constexpr int AC_FRAMES_PER_BLOCK = 1024;

__global__ __launch_bounds__(256) void conv0_moments_kernel(const float* __restrict__ audio, long S, int T_static,
                                                            float* __restrict__ partials, int nch, const int* __restrict__ bounds) {
    const int T_real = frames_collated(bounds, T_static);
    __shared__ float red[4][AC_TERMS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const float* xb = audio + (long)b * S;
    float acc[AC_TERMS];
#pragma unroll
    for (int i = 0; i < AC_TERMS; ++i) acc[i] = 0.f;
    const int t0 = chunk * AC_FRAMES_PER_BLOCK;
    int t1 = t0 + AC_FRAMES_PER_BLOCK;
    t1 = t1 < T_real ? t1 : T_real;
    for (int t = t0 + threadIdx.x; t < t1; t += 256) {
        float x[KW];
#pragma unroll
        for (int k = 0; k < KW; ++k) x[k] = xb[(long)t * STRIDE + k];
        int idx = KW;
#pragma unroll
        for (int k = 0; k < KW; ++k) {
            acc[k] += x[k];
#pragma unroll
            for (int k2 = k; k2 < KW; ++k2) { acc[idx] = fmaf(x[k], x[k2], acc[idx]); ++idx; }
        }
    }
#pragma unroll
    for (int i = 0; i < AC_TERMS; ++i) {
        const float v = wave_sum(acc[i]);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < AC_TERMS)
        partials[((long)b * nch + chunk) * AC_TERMS + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// mean / rstd per (b, c) from the window moments, in double.  grid (B), 512 threads
__global__ void conv0_moments_final_kernel(const float* __restrict__ partials, const float* __restrict__ w,
                                           const float* __restrict__ bias, float* __restrict__ stats, int nch, int T_real,
                                           float eps, const int* __restrict__ bounds) {
    __shared__ double mom[AC_TERMS];
    const int b = blockIdx.x, c = threadIdx.x;
    if (c < AC_TERMS) {
        double s = 0.0;
        for (int k = 0; k < nch; ++k) s += (double)partials[((long)b * nch + k) * AC_TERMS + c];
        // bucketed hipGraphs: sums (conv0_moments_kernel) and frame count both follow the batch as collated
        mom[c] = s / frames_collated(bounds, T_real);
    }
    __syncthreads();
    double wk[KW];
#pragma unroll
    for (int k = 0; k < KW; ++k) wk[k] = (double)w[c * KW + k];
    const double bc = bias ? (double)bias[c] : 0.0;
    double wm = 0.0, q = 0.0;
    int idx = KW;
#pragma unroll
    for (int k = 0; k < KW; ++k) {
        wm += wk[k] * mom[k];
#pragma unroll
        for (int k2 = k; k2 < KW; ++k2) q += (k2 == k ? 1.0 : 2.0) * wk[k] * wk[k2] * mom[idx++];
    }
    const double mean = bc + wm;
    double var = q + 2.0 * bc * wm + bc * bc - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    stats[((long)b * 2 + 0) * C0 + c] = (float)mean;
    stats[((long)b * 2 + 1) * C0 + c] = (float)(1.0 / sqrt(var + (double)eps));
}

// ---- mode 0 pass 2: conv -> (v - mean) * rstd * gamma + beta -> GELU.  grid (blocks, B)
// VALU-bound, not HBM-bound: per output 10 conv FMAs + the affine + ~15 for GELU (two quarter-rate transcendentals) is
// ~28 issue slots against 2 bytes written, i.e. ~0.19 ms of VALU time at 16 x 10 s against 85 us of HBM time.  So the
// normalisation is folded into the taps once per block (w' = w * rstd * gamma, b' = (b - mean) * rstd * gamma + beta)
// and the arithmetic runs on the packed-f32 pipe (v_pk_fma_f32 / v_pk_mul_f32: two channels per issue slot).
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) {
    const f32x2 xc = {__builtin_amdgcn_fmed3f(x.x, -7.0f, 7.0f), __builtin_amdgcn_fmed3f(x.y, -7.0f, 7.0f)};
    const f32x2 x2 = xc * xc;
    f32x2 p = __builtin_elementwise_fma(x2, (f32x2){APTAI_GELU_A5 * APTAI_NLOG2E, APTAI_GELU_A5 * APTAI_NLOG2E},
                                        (f32x2){APTAI_GELU_A3 * APTAI_NLOG2E, APTAI_GELU_A3 * APTAI_NLOG2E});
    p = __builtin_elementwise_fma(p, x2, (f32x2){APTAI_GELU_A1 * APTAI_NLOG2E, APTAI_GELU_A1 * APTAI_NLOG2E});
    const f32x2 z = xc * p;
    const f32x2 e = (f32x2){__builtin_amdgcn_exp2f(z.x), __builtin_amdgcn_exp2f(z.y)} + (f32x2){1.0f, 1.0f};
    return x * (f32x2){__builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y)};
}

__global__ __launch_bounds__(256) void conv0_group_kernel(Conv0Args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    f32x2 w2[4][KW], b2[4];
    {
        float w[8][KW], bias[8];
        load_weights(a, lane, w, bias);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = lane * 8 + j;
            const float mu = a.stats[((long)b * 2 + 0) * C0 + c], rs = a.stats[((long)b * 2 + 1) * C0 + c];
            const float sc = rs * a.gamma[c];
            const float sh = fmaf(bias[j] - mu, sc, a.beta[c]);
#pragma unroll
            for (int k = 0; k < KW; ++k) w2[j >> 1][k][j & 1] = w[j][k] * sc;
            b2[j >> 1][j & 1] = sh;
        }
    }
    const float* xb = a.audio + (long)b * a.S;
    bf16_t* ob = a.out + (long)b * a.T_alloc * C0;
    // the 10 samples of the NEXT frame are loaded before the current one is computed: without the prefetch every frame
    // exposes one global-load latency to its wave
    const int tstep = gridDim.x * 4;
    int t = blockIdx.x * 4 + wave;
    float xs[KW];
#pragma unroll
    for (int k = 0; k < KW; ++k) xs[k] = 0.f;
    if (t < a.T_real) {
#pragma unroll
        for (int k = 0; k < KW; ++k) xs[k] = xb[(long)t * STRIDE + k];
    }
    for (; t < a.T_alloc; t += tstep) {
        float xn[KW];
        const int tn = t + tstep;
        const bool has_next = tn < a.T_real;
        const float* xp = xb + (long)(has_next ? tn : 0) * STRIDE;       // clamped: the load is always in bounds
#pragma unroll
        for (int k = 0; k < KW; ++k) xn[k] = xp[k];
        float o[8];
        if (t < a.T_real) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x2 acc = b2[j];
#pragma unroll
                for (int k = 0; k < KW; ++k) acc = __builtin_elementwise_fma((f32x2){xs[k], xs[k]}, w2[j][k], acc);
                const f32x2 g = gelu_fast2(acc);
                o[2 * j] = g.x; o[2 * j + 1] = g.y;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = 0.f;
        }
        store8(ob + (long)t * C0 + lane * 8, o);
#pragma unroll
        for (int k = 0; k < KW; ++k) xs[k] = xn[k];
    }
}


// ---- mode 0 pass 2 on the fp32 MATRIX pipe (v_mfma_f32_16x16x4_f32: exact fp32 products, k-ordered fp32 accumulation).
// conv0_group_kernel above spends ~28 vector issue slots per output, 10 of them conv FMAs; here the 10-tap dot products run as
// [16 channels] x [12 taps] x [16 frames] MFMA blocks beside the vector pipe, which keeps only the GELU and the packing:
//   A (16 x 4 per k-step) = folded taps  w'[c][k] = w[c][k] * rstd_c * gamma_c   (k = 10: the folded shift, k = 11: 0)
//   B (4 x 16 per k-step) = waveform      x~[k][f] = x[5 (t0 + f) + k]            (k = 10: 1.0 for real frames, k = 11: 0)
//   D[i][f] = pre-activation of channel ch(g, i) at frame t0 + f;  lane (q = lane / 16, f = lane % 16) holds rows 4q .. 4q+3.
// Channel map ch(g, 4q + r) = 32 (g / 2) + 8 q + 4 (g % 2) + r: the lane's outputs of a PAIR of groups are 8 consecutive
// channels = one 16-byte store (a store instruction writes 16 rows x 64 B; the next pair completes the 128-byte lines).
// One wave per 16 frames; the 96 tap registers stay resident, two groups' accumulators are live at a time.
__global__ __launch_bounds__(256) void conv0_group_mfma_kernel(Conv0Args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, f = lane & 15;
    const int b = blockIdx.y;
    float wa[32][3];                                     // A operand: row i = f, k = 4 s + q
#pragma unroll
    for (int g = 0; g < 32; ++g) {
        const int c = 32 * (g >> 1) + 8 * (f >> 2) + 4 * (g & 1) + (f & 3);
        const float mu = a.stats[((long)b * 2 + 0) * C0 + c], rs = a.stats[((long)b * 2 + 1) * C0 + c];
        const float sc = rs * a.gamma[c];
        const float sh = fmaf((a.bias ? a.bias[c] : 0.f) - mu, sc, a.beta[c]);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int k = 4 * s + q;
            wa[g][s] = k < KW ? a.w[c * KW + k] * sc : (k == KW ? sh : 0.f);
        }
    }
    const float* xb = a.audio + (long)b * a.S;
    bf16_t* ob = a.out + (long)b * a.T_alloc * C0;
    const long last = a.S - 1;
    const int nblk = a.T_alloc >> 4;                     // T_alloc is a multiple of 64 (product of the later strides)
    const int step = gridDim.x * 4;
    auto load_x = [&](int blk, float (&x)[3]) {
        const int t = blk * 16 + f;
        const bool real = t < a.T_real;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int k = 4 * s + q;
            long idx = (long)t * STRIDE + k;
            idx = idx < last ? idx : last;               // k >= 10 meets a zero tap: any in-bounds sample will do
            const float v = xb[real ? idx : 0];
            x[s] = !real ? 0.f : (k < KW ? v : (k == KW ? 1.0f : 0.f));
        }
    };
    int blk = blockIdx.x * 4 + wave;
    float xc[3] = {0.f, 0.f, 0.f};
    if (blk < nblk) load_x(blk, xc);
    for (; blk < nblk; blk += step) {
        float xn[3] = {0.f, 0.f, 0.f};
        if (blk + step < nblk) load_x(blk + step, xn);   // next block's samples under this block's arithmetic
        bf16_t* orow = ob + (long)(blk * 16 + f) * C0 + 8 * q;
#pragma unroll
        for (int gp = 0; gp < 16; ++gp) {
            f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[2 * gp][s], xc[s], d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[2 * gp + 1][s], xc[s], d1, 0, 0, 0);
            }
            const f32x2 g0 = gelu_fast2((f32x2){d0[0], d0[1]}), g1 = gelu_fast2((f32x2){d0[2], d0[3]});
            const f32x2 g2 = gelu_fast2((f32x2){d1[0], d1[1]}), g3 = gelu_fast2((f32x2){d1[2], d1[3]});
            *(u32x4*)(orow + 32 * gp) = (u32x4){pack2bf(g0.x, g0.y), pack2bf(g1.x, g1.y), pack2bf(g2.x, g2.y), pack2bf(g3.x, g3.y)};
        }
#pragma unroll
        for (int s = 0; s < 3; ++s) xc[s] = xn[s];
    }
}


// ====================================================================================== backward of layer 0
// The conv output is never stored: every pass recomputes it from the waveform (10 MACs/output) and fuses GELU', the
// norm backward and the weight-gradient accumulation.  Weight/affine gradients are reduced deterministically:
// per-block partials -> one small reduction.
constexpr int BWD_FRAMES_PER_BLOCK = 256;

struct Conv0BwdArgs {
    Conv0Args f;                // forward description (audio, weights, stats, T_real, T_alloc, eps, nchunks)
    const bf16_t* dy;           // [B][T_alloc][512]
    float* gpart;               // group mode: [B][nchunks][2][512] sums of dgn and dgn*xhat
    const float* gmean;         // group mode: [B][2][512] = (mean_t dgn, mean_t dgn*xhat)
    float* wpart;               // [B][nchunks][512][13]: 10 taps, dbias, dgamma, dbeta partials
};

__device__ __forceinline__ void load8bf(const bf16_t* p, float (&o)[8]) {
    const u32x4 v = *(const u32x4*)p;
#pragma unroll
    for (int r = 0; r < 4; ++r) { o[2 * r] = lo_bf(v[r]); o[2 * r + 1] = hi_bf(v[r]); }
}

// group mode, pass A: partial sums over frames of dgn and dgn*xhat per (b, channel)
__global__ __launch_bounds__(256) void conv0_bwd_group_stats_kernel(Conv0BwdArgs a) {
    __shared__ float red[4][2][C0];
    const Conv0Args& f = a.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y, chunk = blockIdx.x;
    float w[8][KW], bias[8], mu[8], rs[8], gm[8], bt[8];
    load_weights(f, lane, w, bias);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = lane * 8 + j;
        mu[j] = f.stats[((long)b * 2 + 0) * C0 + c]; rs[j] = f.stats[((long)b * 2 + 1) * C0 + c];
        gm[j] = f.gamma[c]; bt[j] = f.beta[c];
    }
    const float* xb = f.audio + (long)b * f.S;
    const bf16_t* dyb = a.dy + (long)b * f.T_alloc * C0;
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
    const int t0 = chunk * BWD_FRAMES_PER_BLOCK;
    int t1 = t0 + BWD_FRAMES_PER_BLOCK;
    { const int te = frames_collated(f.bounds, f.T_real); t1 = t1 < te ? t1 : te; }
    for (int t = t0 + wave; t < t1; t += 4) {
        float v[8], d[8];
        conv_frame(xb + (long)t * STRIDE, w, bias, v);
        load8bf(dyb + (long)t * C0 + lane * 8, d);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xh = (v[j] - mu[j]) * rs[j];
            const float dgn = d[j] * gelu_fast_grad(fmaf(xh, gm[j], bt[j]));
            s1[j] += dgn;
            s2[j] = fmaf(dgn, xh, s2[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[wave][0][lane * 8 + j] = s1[j]; red[wave][1][lane * 8 + j] = s2[j]; }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C0; i += 256) {
        const int which = i / C0, c = i % C0;
        a.gpart[(((long)b * f.nchunks + chunk) * 2 + which) * C0 + c] =
            red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
    }
}

// sums over chunks (double) -> per-(b,c) means; also dgamma/dbeta = sums over b of the totals.  The chunk partials of one
// (b, channel) are summed by 16 threads side by side (a serial walk by 512 threads took 645 us at 16 x 10 s: rocprofv3, round 2);
// block = 64 channels x 16 chunk slices, the slices meet in LDS in slice order (deterministic).
__global__ __launch_bounds__(1024) void conv0_bwd_group_final_kernel(const float* __restrict__ gpart, float* __restrict__ gmean,
                                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, int B,
                                                                     int nchunks, int T_real_static, const int* __restrict__ bounds) {
    __shared__ double red[2][16][64];
    const int T_real = frames_collated(bounds, T_real_static);
    const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    double tg = 0.0, tb = 0.0;
    for (int b = 0; b < B; ++b) {
        double s1 = 0.0, s2 = 0.0;
        for (int k = sl; k < nchunks; k += 16) {
            s1 += (double)gpart[(((long)b * nchunks + k) * 2 + 0) * C0 + c];
            s2 += (double)gpart[(((long)b * nchunks + k) * 2 + 1) * C0 + c];
        }
        red[0][sl][cl] = s1;
        red[1][sl][cl] = s2;
        __syncthreads();
        if (sl == 0) {
            s1 = 0.0; s2 = 0.0;
#pragma unroll
            for (int j = 0; j < 16; ++j) { s1 += red[0][j][cl]; s2 += red[1][j][cl]; }
            gmean[((long)b * 2 + 0) * C0 + c] = (float)(s1 / T_real);
            gmean[((long)b * 2 + 1) * C0 + c] = (float)(s2 / T_real);
            tb += s1;
            tg += s2;
        }
        __syncthreads();
    }
    if (sl == 0) {
        if (dgamma) dgamma[c] = (float)tg;
        if (dbeta) dbeta[c] = (float)tb;
    }
}

// pass B (both modes): du per frame -> per-block partials of dW (10 taps), dbias, and (layer mode) dgamma/dbeta
template <int MODE>
__global__ __launch_bounds__(256) void conv0_bwd_weight_kernel(Conv0BwdArgs a) {
    __shared__ float red[4][C0];
    const Conv0Args& f = a.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y, chunk = blockIdx.x;
    float w[8][KW], bias[8], gm[8], bt[8], mu[8], rs[8], m1[8], m2[8];
    load_weights(f, lane, w, bias);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = lane * 8 + j;
        gm[j] = f.gamma[c]; bt[j] = f.beta[c];
        if (MODE == 0) {
            mu[j] = f.stats[((long)b * 2 + 0) * C0 + c]; rs[j] = f.stats[((long)b * 2 + 1) * C0 + c];
            m1[j] = a.gmean[((long)b * 2 + 0) * C0 + c]; m2[j] = a.gmean[((long)b * 2 + 1) * C0 + c];
        }
    }
    const float* xb = f.audio + (long)b * f.S;
    const bf16_t* dyb = a.dy + (long)b * f.T_alloc * C0;
    float acc[8][13];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int k = 0; k < 13; ++k) acc[j][k] = 0.f;
    const int t0 = chunk * BWD_FRAMES_PER_BLOCK;
    int t1 = t0 + BWD_FRAMES_PER_BLOCK;
    { const int te = frames_collated(f.bounds, f.T_real); t1 = t1 < te ? t1 : te; }
    for (int t = t0 + wave; t < t1; t += 4) {
        float v[8], d[8], du[8], smp[KW];
        const float* xs = xb + (long)t * STRIDE;
        conv_frame(xs, w, bias, v);
#pragma unroll
        for (int k = 0; k < KW; ++k) smp[k] = xs[k];
        load8bf(dyb + (long)t * C0 + lane * 8, d);
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xh = (v[j] - mu[j]) * rs[j];
                const float dgn = d[j] * gelu_fast_grad(fmaf(xh, gm[j], bt[j]));
                du[j] = gm[j] * rs[j] * (dgn - m1[j] - xh * m2[j]);
            }
        } else {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];
            const float mean = wave_sum(s) * (1.0f / C0);
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float e = v[j] - mean; q += e * e; }
            const float rstd = rsqrtf(wave_sum(q) * (1.0f / C0) + f.eps);
            float xh[8], gd[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                xh[j] = (v[j] - mean) * rstd;
                const float dln = d[j] * gelu_fast_grad(fmaf(xh[j], gm[j], bt[j]));
                acc[j][11] = fmaf(dln, xh[j], acc[j][11]);      // dgamma
                acc[j][12] += dln;                               // dbeta
                gd[j] = dln * gm[j];
                s1 += gd[j];
                s2 = fmaf(gd[j], xh[j], s2);
            }
            s1 = wave_sum(s1) * (1.0f / C0);
            s2 = wave_sum(s2) * (1.0f / C0);
#pragma unroll
            for (int j = 0; j < 8; ++j) du[j] = rstd * (gd[j] - s1 - xh[j] * s2);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int k = 0; k < KW; ++k) acc[j][k] = fmaf(du[j], smp[k], acc[j][k]);
            acc[j][10] += du[j];                                 // dbias
        }
    }
    // combine the 4 waves, one of the 13 quantities at a time
    float* out = a.wpart + ((long)b * f.nchunks + chunk) * C0 * 13;
    for (int k = 0; k < 13; ++k) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) red[wave][lane * 8 + j] = acc[j][k];
        __syncthreads();
        for (int c = threadIdx.x; c < C0; c += 256) out[(long)c * 13 + k] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
    }
}

// ---- pass A of the group mode on the fp32 matrix pipe: same layout as conv0_bwd_weight_mfma_kernel below (read its comment first) -
// contraction (1) gives xhat^T with the frames on the registers, the vector pipe forms dgn = dy * gelu'(gamma xhat + beta) and each lane
// accumulates sum_f dgn and sum_f dgn * xhat of its 8 channels over its frames; the four frame quarters of a channel (lanes q = 0..3) meet
// in two shuffles at the end.  Writes the same [B][nchunks][2][512] partials as conv0_bwd_group_stats_kernel.
__global__ __launch_bounds__(256) void conv0_bwd_group_stats_mfma_kernel(Conv0BwdArgs a) {
    const Conv0Args& f = a.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, j = lane & 15;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int c0 = 128 * wave + 8 * j;
    float w1[8][3], gm[8], bt[8], s1[8], s2[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const int c = c0 + g;
        const float mu = f.stats[((long)b * 2 + 0) * C0 + c], rs = f.stats[((long)b * 2 + 1) * C0 + c];
        gm[g] = f.gamma[c]; bt[g] = f.beta[c];
        s1[g] = s2[g] = 0.f;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int k = 4 * s + q;
            w1[g][s] = k < KW ? f.w[c * KW + k] * rs : (k == KW ? ((f.bias ? f.bias[c] : 0.f) - mu) * rs : 0.f);
        }
    }
    const float* xb = f.audio + (long)b * f.S;
    const bf16_t* dyb = a.dy + (long)b * f.T_alloc * C0 + c0;
    const long last = f.S - 1;
    const int tb = chunk * BWD_FRAMES_PER_BLOCK;
    int t1 = tb + BWD_FRAMES_PER_BLOCK;
    { const int te = frames_collated(f.bounds, f.T_real); t1 = t1 < te ? t1 : te; }
    for (int t0 = tb; t0 < t1; t0 += 16) {
        float x1[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int k = 4 * s + q;
            long idx = (long)(t0 + j) * STRIDE + k;
            idx = idx < last ? idx : last;
            const float v = xb[idx];
            x1[s] = k < KW ? v : (k == KW ? 1.0f : 0.f);
        }
        u32x4 dq[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = t0 + 4 * q + r;
            dq[r] = *(const u32x4*)(dyb + (long)(t < f.T_alloc ? t : f.T_alloc - 1) * C0);
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            f32x4 xh = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 3; ++s) xh = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[s], w1[g][s], xh, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t pair = dq[r][g >> 1];
                const float d = (g & 1) ? hi_bf(pair) : lo_bf(pair);
                float dgn = d * gelu_fast_grad(fmaf(xh[r], gm[g], bt[g]));
                dgn = (t0 + 4 * q + r < t1) ? dgn : 0.f;
                s1[g] += dgn;
                s2[g] = fmaf(dgn, xh[r], s2[g]);
            }
        }
    }
    float* out = a.gpart + ((long)b * f.nchunks + chunk) * 2 * C0;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        float u = s1[g], v = s2[g];
        u += __shfl_xor(u, 16, 64); v += __shfl_xor(v, 16, 64);
        u += __shfl_xor(u, 32, 64); v += __shfl_xor(v, 32, 64);
        if (q == 0) { out[c0 + g] = u; out[C0 + c0 + g] = v; }
    }
}

// ---- pass B of the group mode on the fp32 MATRIX pipe.  conv0_bwd_weight_kernel<0> spends ~45 vector issue slots per output element
// (10 conv FMAs to recompute it, the norm / GELU derivative, 11 FMAs into the tap accumulators): 629 us per call at 16 x 10 s, six
// times its HBM floor (the 524 MB of dy).  Here both contractions run as v_mfma_f32_16x16x4_f32 blocks and the vector pipe keeps
// only the ~20 slots of the derivative:
//   (1) xhat^T[f][c]  = sum_k x~[f][k] w'[k][c]      A = waveform window (16 frames x 12), B = taps folded with rstd and the shift
//                        (k = 10: (bias - mean) rstd against x~ = 1, k = 11: 0): the NORMALISED conv output, frames on the registers
//   (2) dW^T[k][c]   += sum_f x^T[k][f] du[f][c]     register r of the lane that holds frames 4 q + r of (1) IS the B operand of K-step
//                        r (frames {r, 4 + r, 8 + r, 12 + r}); A = x[5 (t0 + 4 q + r) + k], k = 10: 1.0 -> dbias, k >= 11: 0
// One wave = one quarter of the channels (8 groups of 16) for the block's 256 frames, so the four waves write disjoint channels and
// nothing is reduced across them.  Channel map c(g, j) = 128 w + 8 j + g: a lane's 8 groups are 8 consecutive channels = one 16-byte
// dy load per frame row (a wave instruction reads 256-byte row segments).
__global__ __launch_bounds__(256) void conv0_bwd_weight_mfma_kernel(Conv0BwdArgs a) {
    const Conv0Args& f = a.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, j = lane & 15;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int c0 = 128 * wave + 8 * j;                   // this lane's 8 channels c0 + g
    float w1[8][3], gm[8], bt[8], gr[8], m1g[8], m2g[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const int c = c0 + g;
        const float mu = f.stats[((long)b * 2 + 0) * C0 + c], rs = f.stats[((long)b * 2 + 1) * C0 + c];
        gm[g] = f.gamma[c]; bt[g] = f.beta[c];
        gr[g] = gm[g] * rs;
        m1g[g] = gr[g] * a.gmean[((long)b * 2 + 0) * C0 + c];
        m2g[g] = gr[g] * a.gmean[((long)b * 2 + 1) * C0 + c];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int k = 4 * s + q;
            w1[g][s] = k < KW ? f.w[c * KW + k] * rs : (k == KW ? ((f.bias ? f.bias[c] : 0.f) - mu) * rs : 0.f);
        }
    }
    const float* xb = f.audio + (long)b * f.S;
    const bf16_t* dyb = a.dy + (long)b * f.T_alloc * C0 + c0;
    const long last = f.S - 1;
    f32x4 acc[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int tb = chunk * BWD_FRAMES_PER_BLOCK;
    int t1 = tb + BWD_FRAMES_PER_BLOCK;
    { const int te = frames_collated(f.bounds, f.T_real); t1 = t1 < te ? t1 : te; }
    for (int t0 = tb; t0 < t1; t0 += 16) {
        // operands of both contractions from the waveform (L1-resident: 16 frames = 85 samples); a software prefetch of the next block's
        // operands measured 4 % slower (645 -> 674 us for the whole call): the waves of a SIMD already cover each other's loads
        float x1[3], x2[4];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int k = 4 * s + q;
            long idx = (long)(t0 + j) * STRIDE + k;
            idx = idx < last ? idx : last;
            const float v = xb[idx];
            x1[s] = k < KW ? v : (k == KW ? 1.0f : 0.f);
        }
        u32x4 dq[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = t0 + 4 * q + r;
            long idx = (long)t * STRIDE + j;
            idx = idx < last ? idx : last;
            const float v = xb[idx];
            x2[r] = j < KW ? v : (j == KW ? 1.0f : 0.f);
            dq[r] = *(const u32x4*)(dyb + (long)(t < f.T_alloc ? t : f.T_alloc - 1) * C0);
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            f32x4 xh = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 3; ++s) xh = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[s], w1[g][s], xh, 0, 0, 0);
            float du[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t pair = dq[r][g >> 1];
                const float d = (g & 1) ? hi_bf(pair) : lo_bf(pair);
                const float dgn = d * gelu_fast_grad(fmaf(xh[r], gm[g], bt[g]));
                const float v = fmaf(dgn, gr[g], -fmaf(xh[r], m2g[g], m1g[g]));
                du[r] = (t0 + 4 * q + r < t1) ? v : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(x2[r], du[r], acc[g], 0, 0, 0);
        }
    }
    // acc[g][r] = partial of quantity k = 4 q + r (taps 0..9, dbias 10, zeros above) of channel c0 + g
    float* out = a.wpart + ((long)b * f.nchunks + chunk) * C0 * 13;
#pragma unroll
    for (int g = 0; g < 8; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = 4 * q + r;
            if (k < 13) out[(long)(c0 + g) * 13 + k] = acc[g][r];
        }
}

// dweight [512][10], dbias [512], (layer mode) dgamma/dbeta [512] = sum over (b, chunk) partials: 64 outputs x 16 partial
// slices per block, slices combined in LDS in slice order (the serial walk over 8 000 partials took 631 us)
__global__ __launch_bounds__(1024) void conv0_bwd_reduce_kernel(const float* __restrict__ wpart, int nparts, float* __restrict__ dweight,
                                                                float* __restrict__ dbias, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta) {
    __shared__ double red[16][64];
    const int il = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + il;
    double s = 0.0;
    if (i < C0 * 13)
        for (int p = sl; p < nparts; p += 16) s += (double)wpart[(long)p * C0 * 13 + i];
    red[sl][il] = s;
    __syncthreads();
    if (sl != 0 || i >= C0 * 13) return;
    s = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += red[j][il];
    const int c = i / 13, k = i % 13;
    if (k < KW) dweight[c * KW + k] = (float)s;
    else if (k == 10) { if (dbias) dbias[c] = (float)s; }
    else if (k == 11) { if (dgamma) dgamma[c] = (float)s; }
    else { if (dbeta) dbeta[c] = (float)s; }
}

}  // namespace

extern "C" int64_t aptai_conv0_workspace_bytes(int64_t B, int64_t T_real) {
    const long nchunks = ceil_div(T_real, FRAMES_PER_BLOCK_STATS);
    return (B * nchunks * 2 * C0 + B * 2 * C0) * 4;
}

extern "C" int aptai_conv0_fwd(const float* audio, int64_t B, int64_t S, const float* weight, const float* bias,
                               const float* gamma, const float* beta, int mode, float eps, void* out, int64_t T_real,
                               int64_t T_alloc, int64_t C, int64_t Kw, int64_t stride, void* workspace, float* stats_out,
                               void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APTAI_REQUIRE(audio && weight && gamma && beta && out, "aptai_conv0_fwd: null pointer");
    APTAI_REQUIRE(C == C0 && Kw == KW && stride == STRIDE, "aptai_conv0_fwd: built for C=512, k=10, s=5 (got %ld,%ld,%ld)",
                  (long)C, (long)Kw, (long)stride);
    APTAI_REQUIRE(B > 0 && T_real > 0 && T_alloc >= T_real, "aptai_conv0_fwd: bad frame counts");
    APTAI_REQUIRE((T_real - 1) * STRIDE + KW <= S, "aptai_conv0_fwd: T_real=%ld frames need more than S=%ld samples", (long)T_real, (long)S);
    APTAI_REQUIRE(mode == 0 || mode == 1, "aptai_conv0_fwd: mode must be 0 (group) or 1 (layer)");
    Conv0Args a;
    memset(&a, 0, sizeof(a));
    a.audio = audio; a.S = S; a.w = weight; a.bias = bias; a.gamma = gamma; a.beta = beta;
    a.out = (bf16_t*)out; a.B = (int)B; a.T_real = (int)T_real; a.T_alloc = (int)T_alloc; a.eps = eps;
    long blocks = ceil_div(T_alloc, 4 * 8);           // 8 frames per wave
    if (blocks > 1024) blocks = 1024;
    if (mode == 1) {
        APTAI_LAUNCH(conv0_layer_kernel, dim3((unsigned)blocks, (unsigned)B), dim3(256), 0, stream, a);
        APTAI_CHECK_LAUNCH("conv0_layer_kernel");
        return APTAI_OK;
    }
    APTAI_REQUIRE(workspace != nullptr, "aptai_conv0_fwd: group mode needs a workspace");
    a.nchunks = (int)ceil_div(T_real, FRAMES_PER_BLOCK_STATS);
    a.partials = (float*)workspace;
    float* stats = stats_out ? stats_out : (float*)workspace + (long)B * a.nchunks * 2 * C0;
    a.stats = stats;
    const int nch = (int)ceil_div(T_real, AC_FRAMES_PER_BLOCK);   // [B][nch][65] floats: fits the conv-pass workspace
    APTAI_LAUNCH(conv0_moments_kernel, dim3((unsigned)nch, (unsigned)B), dim3(256), 0, stream, audio, (long)S, (int)T_real,
                 a.partials, nch, (const int*)aptai_frame_bounds(stream_));
    APTAI_CHECK_LAUNCH("conv0_moments_kernel");
    APTAI_LAUNCH(conv0_moments_final_kernel, dim3((unsigned)B), dim3(C0), 0, stream, (const float*)a.partials, weight, bias,
                 stats, nch, (int)T_real, eps, (const int*)aptai_frame_bounds(stream_));
    APTAI_CHECK_LAUNCH("conv0_moments_final_kernel");
    // the conv on the fp32 matrix pipe beside the vector pipe's GELU (APTAI_CONV0_MFMA=0: the all-vector kernel, A/B)
    static int use_mfma = -1;
    if (use_mfma < 0) {
        const char* e = getenv("APTAI_CONV0_MFMA");
        use_mfma = e ? atoi(e) : 1;
    }
    if (use_mfma && T_alloc % 16 == 0) {
        // ~two rounds of the 768 block slots (3 blocks per CU at 145 registers), each wave amortising its 96 tap registers over
        // several 16-frame blocks
        long mb = ceil_div(1536, B);
        const long mb_max = ceil_div(T_alloc / 16, 4);
        if (mb > mb_max) mb = mb_max;
        APTAI_LAUNCH(conv0_group_mfma_kernel, dim3((unsigned)mb, (unsigned)B), dim3(256), 0, stream, a);
        APTAI_CHECK_LAUNCH("conv0_group_mfma_kernel");
        return APTAI_OK;
    }
    APTAI_LAUNCH(conv0_group_kernel, dim3((unsigned)blocks, (unsigned)B), dim3(256), 0, stream, a);
    APTAI_CHECK_LAUNCH("conv0_group_kernel");
    return APTAI_OK;
}

extern "C" int64_t aptai_conv0_bwd_workspace_bytes(int64_t B, int64_t T_real) {
    const long nch = ceil_div(T_real, BWD_FRAMES_PER_BLOCK);
    return (B * nch * 2 * C0 + B * 2 * C0 + B * nch * C0 * 13) * 4;
}

/* stats: the [B][2][512] (mean, rstd) block the forward left at the END of its workspace (group mode); null in layer mode */
extern "C" int aptai_conv0_bwd(const float* audio, int64_t B, int64_t S, const float* weight, const float* bias, const float* gamma,
                               const float* beta, int mode, float eps, const void* dy, int64_t T_real, int64_t T_alloc,
                               const float* fwd_stats, float* dweight, float* dbias, float* dgamma, float* dbeta, void* workspace,
                               void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APTAI_REQUIRE(audio && weight && gamma && beta && dy && dweight && workspace, "aptai_conv0_bwd: null pointer");
    APTAI_REQUIRE(mode == 0 || mode == 1, "aptai_conv0_bwd: mode");
    APTAI_REQUIRE(mode == 1 || fwd_stats, "aptai_conv0_bwd: group mode needs the forward statistics");
    APTAI_REQUIRE((T_real - 1) * STRIDE + KW <= S && T_alloc >= T_real, "aptai_conv0_bwd: bad frame counts");
    Conv0BwdArgs a;
    memset(&a, 0, sizeof(a));
    a.f.audio = audio; a.f.S = S; a.f.w = weight; a.f.bias = bias; a.f.gamma = gamma; a.f.beta = beta; a.f.B = (int)B;
    a.f.T_real = (int)T_real; a.f.T_alloc = (int)T_alloc; a.f.eps = eps; a.f.stats = fwd_stats;
    a.f.nchunks = (int)ceil_div(T_real, BWD_FRAMES_PER_BLOCK);
    a.f.bounds = mode == 0 ? (const int*)aptai_frame_bounds(stream_) : nullptr;
    a.dy = (const bf16_t*)dy;
    float* ws = (float*)workspace;
    a.gpart = ws;
    float* gmean = ws + (long)B * a.f.nchunks * 2 * C0;
    a.gmean = gmean;
    a.wpart = gmean + (long)B * 2 * C0;
    dim3 grid((unsigned)a.f.nchunks, (unsigned)B);
    if (mode == 0) {
        // both passes' contractions on the fp32 matrix pipe (APTAI_CONV0_BWD_MFMA=0: the all-vector kernels, A/B)
        static const bool bwd_mfma = !(getenv("APTAI_CONV0_BWD_MFMA") && atoi(getenv("APTAI_CONV0_BWD_MFMA")) == 0);
        if (bwd_mfma) APTAI_LAUNCH(conv0_bwd_group_stats_mfma_kernel, grid, dim3(256), 0, stream, a);
        else APTAI_LAUNCH(conv0_bwd_group_stats_kernel, grid, dim3(256), 0, stream, a);
        APTAI_CHECK_LAUNCH("conv0_bwd_group_stats_kernel");
        APTAI_LAUNCH(conv0_bwd_group_final_kernel, dim3(C0 / 64), dim3(1024), 0, stream, (const float*)a.gpart, gmean, dgamma, dbeta, (int)B,
                     a.f.nchunks, (int)T_real, (const int*)aptai_frame_bounds((const void*)stream));
        APTAI_CHECK_LAUNCH("conv0_bwd_group_final_kernel");
        if (bwd_mfma) APTAI_LAUNCH(conv0_bwd_weight_mfma_kernel, grid, dim3(256), 0, stream, a);
        else APTAI_LAUNCH(conv0_bwd_weight_kernel<0>, grid, dim3(256), 0, stream, a);
    } else {
        APTAI_LAUNCH(conv0_bwd_weight_kernel<1>, grid, dim3(256), 0, stream, a);
    }
    APTAI_CHECK_LAUNCH("conv0_bwd_weight_kernel");
    APTAI_LAUNCH(conv0_bwd_reduce_kernel, dim3((unsigned)ceil_div(C0 * 13, 64)), dim3(1024), 0, stream, (const float*)a.wpart,
                 (int)(B * a.f.nchunks), dweight, bias ? dbias : nullptr, mode == 1 ? dgamma : nullptr, mode == 1 ? dbeta : nullptr);
    APTAI_CHECK_LAUNCH("conv0_bwd_reduce_kernel");
    return APTAI_OK;
}
