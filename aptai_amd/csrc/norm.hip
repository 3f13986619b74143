// LayerNorm forward/backward over the channel axis of channels-last bf16 activations (gfx950).
// Replaces nn.LayerNorm at HF:288-299 (conv-stack LN, large), HF:425-431 (feature projection),
// HF:587-601 / 622-644 (encoder layers), HF:691 / 791 (encoder LN) and models/modules.py:134,151.
// HBM-bound: one wave64 per row, 8-byte vector loads, statistics in fp32 with wave shuffles, no LDS
// on the forward path.  Algorithmic bytes/row: 2*cols in + 2*cols out (+8 for mean/rstd).
#include <stdlib.h>

#include "common.h"

namespace {

// fp32 residual stream of the inference-only encoder (Force_APTAI's frozen recogniser, opt-in): the row arrives in fp32 (the
// GEMM before it added bias and residual in fp32) and leaves twice - bf16 for the next GEMM's A operand, fp32 for the next
// residual add - so the residual path is never rounded to bf16.  Same statistics, same lane mapping as ln_fwd_kernel.
template <int NCH>
__global__ __launch_bounds__(256) void ln_fwd_f32_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                         float* __restrict__ y32, long rows, float eps,
                                                         bf16_t* __restrict__ ys = nullptr, int pieces = 0) {
    constexpr int COLS = NCH * 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float gm[NCH][4], bt[NCH][4];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const f32x4 g4 = *(const f32x4*)(gamma + (j * 64 + lane) * 4);
        const f32x4 b4 = *(const f32x4*)(beta + (j * 64 + lane) * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) { gm[j][r] = g4[r]; bt[j][r] = b4[r]; }
    }
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
        const float* xr = x + row * COLS;
        float v[NCH][4];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const f32x4 p = *(const f32x4*)(xr + (j * 64 + lane) * 4);
            v[j][0] = p[0]; v[j][1] = p[1]; v[j][2] = p[2]; v[j][3] = p[3];
            s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
        }
        const float mu = wave_sum(s) * (1.0f / COLS);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = v[j][r] - mu; q += d * d; }
        const float rs = rsqrtf(wave_sum(q) * (1.0f / COLS) + eps);
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (v[j][r] - mu) * rs * gm[j][r] + bt[j][r];
            if (y) *(u32x2*)(y + row * COLS + (j * 64 + lane) * 4) = (u32x2){pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
            if (y32) *(f32x4*)(y32 + row * COLS + (j * 64 + lane) * 4) = (f32x4){o[0], o[1], o[2], o[3]};
            if (ys) {               // exact-index mode: the normalised row also leaves as the next product's split A operand (activation side)
                float h[4], md[4], lw[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) split3(o[r], h[r], md[r], lw[r]);
                const int c = (j * 64 + lane) * 4;
                bf16_t* dst = ys + row * ((long)COLS * pieces) + (long)(c >> 6) * (64 * pieces) + (c & 63);
                const u32x2 ph = {pack2bf(h[0], h[1]), pack2bf(h[2], h[3])}, pm = {pack2bf(md[0], md[1]), pack2bf(md[2], md[3])};
                *(u32x2*)dst = ph;
                *(u32x2*)(dst + 64) = ph;
                *(u32x2*)(dst + 128) = pm;
                if (pieces == 6) {
                    *(u32x2*)(dst + 192) = pm;
                    *(u32x2*)(dst + 256) = ph;
                    *(u32x2*)(dst + 320) = (u32x2){pack2bf(lw[0], lw[1]), pack2bf(lw[2], lw[3])};
                }
            }
        }
    }
}

// MX = true: the result also (or only: y may be null) leaves as MXFP8 - E4M3 elements q + one E8M0 scale per 32 columns - the A operand
// of the fp8 encoder's q|k|v / FFN1 GEMMs (csrc/mxgemm.hip) without the quantiser's own pass.  Bit-identical to the MX = false kernel
// followed by mx_quantize_kernel: it IS the same kernel up to the store (statistics from the same instruction sequence - a separately
// written copy differed in one bf16 value per 3 M), and the normalised values are rounded to bf16 before they are quantised.  A lane
// holds 4 consecutive columns per 256-column chunk, so a 32-column block is 8 neighbouring lanes.
template <int NCH, bool MX = false>   // cols = NCH * 256
__global__ __launch_bounds__(256) void ln_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, long rows,
                                                     float eps, int gelu_after, uint8_t* __restrict__ mq = nullptr, long ldq = 0,
                                                     uint8_t* __restrict__ msc = nullptr, long lds = 0) {
    constexpr int COLS = NCH * 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float gm[NCH][4], bt[NCH][4];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const f32x4 g4 = *(const f32x4*)(gamma + (j * 64 + lane) * 4);
        const f32x4 b4 = *(const f32x4*)(beta + (j * 64 + lane) * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) { gm[j][r] = g4[r]; bt[j][r] = b4[r]; }
    }
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
        const bf16_t* xr = x + row * COLS;
        float v[NCH][4];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const u32x2 p = *(const u32x2*)(xr + (j * 64 + lane) * 4);
            v[j][0] = lo_bf(p[0]); v[j][1] = hi_bf(p[0]); v[j][2] = lo_bf(p[1]); v[j][3] = hi_bf(p[1]);
            s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
        }
        const float mu = wave_sum(s) * (1.0f / COLS);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = v[j][r] - mu; q += d * d; }
        const float rs = rsqrtf(wave_sum(q) * (1.0f / COLS) + eps);
        bf16_t* yr = y + row * COLS;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                o[r] = fmaf((v[j][r] - mu) * rs, gm[j][r], bt[j][r]);      // (spelled out: ln_fwd_mx_kernel must round the same way)
                if (gelu_after) o[r] = gelu_fast(o[r]);
            }
            const u32x2 packed = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
            if (!MX || y) *(u32x2*)(yr + (j * 64 + lane) * 4) = packed;
            if (MX) {
                o[0] = lo_bf(packed[0]); o[1] = hi_bf(packed[0]); o[2] = lo_bf(packed[1]); o[3] = hi_bf(packed[1]);
                float amax = fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3])));
                amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
                amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
                amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
                int e = (int)((__float_as_uint(amax) >> 23) & 0xffu) - 8;   // as mx_quantize_kernel: 2^(floor(log2 amax) - 8), biased
                e = e < 1 ? 1 : (e > 254 ? 254 : e);
                const float inv = __uint_as_float((uint32_t)(254 - e) << 23);
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = __builtin_amdgcn_fmed3f(o[r] * inv, -448.f, 448.f);
                int w = __builtin_amdgcn_cvt_pk_fp8_f32(o[0], o[1], 0, false);
                w = __builtin_amdgcn_cvt_pk_fp8_f32(o[2], o[3], w, true);
                *(uint32_t*)(mq + row * ldq + (j * 64 + lane) * 4) = (uint32_t)w;
                if ((lane & 7) == 0) msc[row * lds + ((j * 64 + lane) >> 3)] = (uint8_t)e;
            }
        }
        if (lane == 0) {
            if (mean) mean[row] = mu;
            if (rstd) rstd[row] = rs;
        }
    }
}

// dx = rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat)) [+ dres];  partial dgamma/dbeta per block.
template <int NCH, bool GELU>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const bf16_t* __restrict__ dres,
                                                     bf16_t* __restrict__ dx, bf16_t* __restrict__ dx_drop,
                                                     uint32_t seed0, uint32_t seed1, uint32_t thr16, float dscale,
                                                     float* __restrict__ partials, long rows, const float* __restrict__ beta_gelu,
                                                     const uint32_t* __restrict__ salt) {
    if (thr16) apply_salt(salt, seed0, seed1);
    constexpr int COLS = NCH * 256;
    __shared__ float red[4][2][COLS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float gm[NCH][4], dg[NCH][4], db[NCH][4], bt[NCH][4];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const f32x4 g4 = *(const f32x4*)(gamma + (j * 64 + lane) * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) { gm[j][r] = g4[r]; dg[j][r] = 0.f; db[j][r] = 0.f; bt[j][r] = 0.f; }
        if (GELU) {
            const f32x4 b4 = *(const f32x4*)(beta_gelu + (j * 64 + lane) * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) bt[j][r] = b4[r];
        }
    }
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
        const float mu = mean[row], rs = rstd[row];
        float xh[NCH][4], gd[NCH][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const long off = row * COLS + (j * 64 + lane) * 4;
            const u32x2 px = *(const u32x2*)(x + off);
            const u32x2 pd = *(const u32x2*)(dy + off);
            const float xv[4] = {lo_bf(px[0]), hi_bf(px[0]), lo_bf(px[1]), hi_bf(px[1])};
            float dv[4] = {lo_bf(pd[0]), hi_bf(pd[0]), lo_bf(pd[1]), hi_bf(pd[1])};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                xh[j][r] = (xv[r] - mu) * rs;
                if (GELU) dv[r] *= gelu_fast_grad(fmaf(xh[j][r], gm[j][r], bt[j][r]));   // y = gelu(LN(x)): fold gelu' into dy
                gd[j][r] = dv[r] * gm[j][r];
                s1 += gd[j][r];
                s2 += gd[j][r] * xh[j][r];
                dg[j][r] += dv[r] * xh[j][r];
                db[j][r] += dv[r];
            }
        }
        s1 = wave_sum(s1) * (1.0f / COLS);
        s2 = wave_sum(s2) * (1.0f / COLS);
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const long off = row * COLS + (j * 64 + lane) * 4;
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = rs * (gd[j][r] - s1 - xh[j][r] * s2);
            if (dres) {
                const u32x2 pr = *(const u32x2*)(dres + off);
                o[0] += lo_bf(pr[0]); o[1] += hi_bf(pr[0]); o[2] += lo_bf(pr[1]); o[3] += hi_bf(pr[1]);
            }
            *(u32x2*)(dx + off) = (u32x2){pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
            if (dx_drop) {
                float m[4];
#pragma unroll
                for (int r = 0; r < 4; r += 2) {                   // off is a multiple of 4: one hash per element pair
                    const uint32_t hsh = drop_hash_pair((uint64_t)off + r, seed0, seed1);
                    m[r] = (hsh & 0xffffu) >= thr16 ? o[r] * dscale : 0.f;
                    m[r + 1] = (hsh >> 16) >= thr16 ? o[r + 1] * dscale : 0.f;
                }
                *(u32x2*)(dx_drop + off) = (u32x2){pack2bf(m[0], m[1]), pack2bf(m[2], m[3])};
            }
        }
    }
    // block reduction of the per-wave column partials
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            red[wave][0][(j * 64 + lane) * 4 + r] = dg[j][r];
            red[wave][1][(j * 64 + lane) * 4 + r] = db[j][r];
        }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * COLS; c += 256) {
        const int which = c / COLS, col = c % COLS;
        const float s = red[0][which][col] + red[1][which][col] + red[2][which][col] + red[3][which][col];
        partials[((long)blockIdx.x * 2 + which) * COLS + col] = s;
    }
}

// out[which][col] = sum_b partials[b][which][col].  block = 16 columns x 64 slices of the block list (1024 threads): 96
// blocks for 2 x 768 columns, 8 independent loads per thread for 512 partial blocks, fixed-order LDS combine.
__global__ __launch_bounds__(1024) void colsum_partials_kernel(const float* __restrict__ partials, float* __restrict__ out0,
                                                               float* __restrict__ out1, int nblocks, int cols) {
    __shared__ float red[64][17];
    const int lc = threadIdx.x & 15, slice = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + lc;                // index into the concatenated [2][cols] vector
    float s = 0.f;
    if (c < 2 * cols) {
        const int which = c / cols, col = c % cols;
        const float* p = partials + (long)which * cols + col;
#pragma unroll 8
        for (int b = slice; b < nblocks; b += 64) s += p[(long)b * 2 * cols];
    }
    red[slice][lc] = s;
    __syncthreads();
    if (slice == 0 && c < 2 * cols) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 64; ++k) t += red[k][lc];
        float* o = (c / cols) ? out1 : out0;
        if (o) o[c % cols] = t;
    }
}

// Many finalisations in ONE launch: job j = {partials [nblocks][2][cols], out0 [cols] | 0, out1 [cols] | 0, nblocks, cols} (device table
// of int64).  The graph runner defers the dgamma / dbeta reductions of all ~24 LayerNorm backwards of a step to one launch at the end
// of the backward pass: as separate launches they were 22 x (4.8 us + a kernel boundary) per step for 3 MB of partials each.
__global__ __launch_bounds__(1024) void colsum_partials_multi_kernel(const long* __restrict__ table) {
    __shared__ float red[64][17];
    const long* job = table + (long)blockIdx.y * 5;
    const float* partials = (const float*)job[0];
    float* out0 = (float*)job[1];
    float* out1 = (float*)job[2];
    const int nblocks = (int)job[3], cols = (int)job[4];
    const int lc = threadIdx.x & 15, slice = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + lc;
    if (blockIdx.x * 16 >= 2 * cols) return;             // block-uniform: this job has fewer columns than the widest one
    float s = 0.f;
    if (c < 2 * cols) {
        const int which = c / cols, col = c % cols;
        const float* p = partials + (long)which * cols + col;
#pragma unroll 8
        for (int b = slice; b < nblocks; b += 64) s += p[(long)b * 2 * cols];
    }
    red[slice][lc] = s;
    __syncthreads();
    if (slice == 0 && c < 2 * cols) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 64; ++k) t += red[k][lc];
        float* o = (c / cols) ? out1 : out0;
        if (o) o[c % cols] = t;
    }
}

constexpr int LN_BWD_MAX_BLOCKS_DEFAULT = 512;
static long ln_bwd_max_blocks() {
    static long v = 0;
    if (v == 0) {
        const char* e = getenv("APTAI_LN_BWD_BLOCKS");          // tuning knob (tools/ln_bench.py)
        v = e ? atol(e) : LN_BWD_MAX_BLOCKS_DEFAULT;
        if (v < 1) v = LN_BWD_MAX_BLOCKS_DEFAULT;
    }
    return v;
}
#define LN_BWD_MAX_BLOCKS ln_bwd_max_blocks()

}  // namespace

extern "C" int aptai_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                                   float* rstd, int64_t rows, int64_t cols, float eps, int gelu_after, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APTAI_REQUIRE(x && gamma && beta && y, "aptai_layernorm_fwd: null pointer");
    APTAI_REQUIRE(rows > 0, "aptai_layernorm_fwd: rows=%ld", (long)rows);
    APTAI_REQUIRE(cols % 256 == 0 && cols >= 256 && cols <= 1024, "aptai_layernorm_fwd: cols=%ld (need 256..1024, %%256)", (long)cols);
    long blocks = ceil_div(rows, 4);
    if (blocks > 2048) blocks = 2048;
#define LN_FWD(NCH) APTAI_LAUNCH(ln_fwd_kernel<NCH>, dim3((unsigned)blocks), dim3(256), 0, stream, (const bf16_t*)x, gamma, beta, (bf16_t*)y, mean, rstd, (long)rows, eps, gelu_after)
    switch (cols / 256) {
        case 1: LN_FWD(1); break;
        case 2: LN_FWD(2); break;
        case 3: LN_FWD(3); break;
        default: LN_FWD(4); break;
    }
#undef LN_FWD
    APTAI_CHECK_LAUNCH("ln_fwd_kernel");
    return APTAI_OK;
}

extern "C" int aptai_layernorm_fwd_mx(const void* x, const float* gamma, const float* beta, void* y_bf16, void* q, int64_t ldq,
                                      void* scales, int64_t lds, int64_t rows, int64_t cols, float eps, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APTAI_REQUIRE(x && gamma && beta && q && scales, "aptai_layernorm_fwd_mx: null pointer");
    APTAI_REQUIRE(rows > 0 && cols % 256 == 0 && cols >= 256 && cols <= 1024, "aptai_layernorm_fwd_mx: rows=%ld cols=%ld (cols: 256..1024, %%256)",
                  (long)rows, (long)cols);
    APTAI_REQUIRE(ldq >= cols && ldq % 4 == 0 && (uintptr_t)q % 4 == 0 && lds >= cols / 32, "aptai_layernorm_fwd_mx: output rows must keep 4-byte alignment");
    long blocks = ceil_div(rows, 4);
    if (blocks > 2048) blocks = 2048;
#define LN_MX(NCH) APTAI_LAUNCH((ln_fwd_kernel<NCH, true>), dim3((unsigned)blocks), dim3(256), 0, stream, (const bf16_t*)x, gamma, beta, (bf16_t*)y_bf16, (float*)nullptr, (float*)nullptr, (long)rows, eps, 0, (uint8_t*)q, (long)ldq, (uint8_t*)scales, (long)lds)
    switch (cols / 256) {
        case 1: LN_MX(1); break;
        case 2: LN_MX(2); break;
        case 3: LN_MX(3); break;
        default: LN_MX(4); break;
    }
#undef LN_MX
    APTAI_CHECK_LAUNCH("ln_fwd_kernel<MX>");
    return APTAI_OK;
}

extern "C" int aptai_layernorm_bwd_finalize_multi(const int64_t* table_dev, int64_t njobs, int64_t max_cols, void* stream) {
    APTAI_REQUIRE(table_dev != nullptr && njobs > 0 && njobs <= 65535 && max_cols > 0, "aptai_layernorm_bwd_finalize_multi: bad arguments");
    APTAI_LAUNCH(colsum_partials_multi_kernel, dim3((unsigned)((2 * max_cols + 15) / 16), (unsigned)njobs), dim3(1024), 0, (hipStream_t)stream,
                 (const long*)table_dev);
    APTAI_CHECK_LAUNCH("colsum_partials_multi_kernel");
    return APTAI_OK;
}

extern "C" int64_t aptai_layernorm_bwd_workspace_bytes(int64_t rows, int64_t cols) {
    long blocks = ceil_div(rows, 4);
    if (blocks > LN_BWD_MAX_BLOCKS) blocks = LN_BWD_MAX_BLOCKS;
    return blocks * 2 * cols * 4;
}

extern "C" int aptai_layernorm_fwd_f32in(const float* x, const float* gamma, const float* beta, void* y_bf16, float* y_f32,
                                         int64_t rows, int64_t cols, float eps, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APTAI_REQUIRE(x && gamma && beta && (y_bf16 || y_f32), "aptai_layernorm_fwd_f32in: null pointer");
    APTAI_REQUIRE(rows > 0, "aptai_layernorm_fwd_f32in: rows=%ld", (long)rows);
    APTAI_REQUIRE(cols % 256 == 0 && cols >= 256 && cols <= 1024, "aptai_layernorm_fwd_f32in: cols=%ld (need 256..1024, %%256)", (long)cols);
    long blocks = ceil_div(rows, 4);
    if (blocks > 2048) blocks = 2048;
#define LN_FWD32(NCH) APTAI_LAUNCH(ln_fwd_f32_kernel<NCH>, dim3((unsigned)blocks), dim3(256), 0, stream, x, gamma, beta, (bf16_t*)y_bf16, y_f32, (long)rows, eps, (bf16_t*)nullptr, 0)
    switch (cols / 256) {
        case 1: LN_FWD32(1); break;
        case 2: LN_FWD32(2); break;
        case 3: LN_FWD32(3); break;
        default: LN_FWD32(4); break;
    }
#undef LN_FWD32
    APTAI_CHECK_LAUNCH("ln_fwd_f32_kernel");
    return APTAI_OK;
}

extern "C" int aptai_layernorm_fwd_f32in_split(const float* x, const float* gamma, const float* beta, float* y_f32, void* y_split, int pieces,
                                               int64_t rows, int64_t cols, float eps, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APTAI_REQUIRE(x && gamma && beta && y_split, "aptai_layernorm_fwd_f32in_split: null pointer");
    APTAI_REQUIRE(rows > 0 && (pieces == 3 || pieces == 6), "aptai_layernorm_fwd_f32in_split: rows=%ld pieces=%d", (long)rows, pieces);
    APTAI_REQUIRE(cols % 256 == 0 && cols >= 256 && cols <= 1024, "aptai_layernorm_fwd_f32in_split: cols=%ld (need 256..1024, %%256)", (long)cols);
    long blocks = ceil_div(rows, 4);
    if (blocks > 2048) blocks = 2048;
#define LN_FWD32S(NCH) APTAI_LAUNCH(ln_fwd_f32_kernel<NCH>, dim3((unsigned)blocks), dim3(256), 0, stream, x, gamma, beta, (bf16_t*)nullptr, y_f32, (long)rows, eps, (bf16_t*)y_split, pieces)
    switch (cols / 256) {
        case 1: LN_FWD32S(1); break;
        case 2: LN_FWD32S(2); break;
        case 3: LN_FWD32S(3); break;
        default: LN_FWD32S(4); break;
    }
#undef LN_FWD32S
    APTAI_CHECK_LAUNCH("ln_fwd_f32_kernel (split)");
    return APTAI_OK;
}

extern "C" int aptai_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd,
                                   const float* gamma, const void* dres, void* dx, void* dx_drop, float dropout_p,
                                   uint64_t seed, float* dgamma, float* dbeta, void* workspace, int64_t rows,
                                   int64_t cols, const float* beta_if_gelu_after, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APTAI_REQUIRE(dy && x && mean && rstd && gamma && dx && workspace, "aptai_layernorm_bwd: null pointer");
    APTAI_REQUIRE(cols % 256 == 0 && cols >= 256 && cols <= 1024, "aptai_layernorm_bwd: cols=%ld", (long)cols);
    APTAI_REQUIRE(rows > 0, "aptai_layernorm_bwd: rows=%ld", (long)rows);
    long blocks = ceil_div(rows, 4);
    if (blocks > LN_BWD_MAX_BLOCKS) blocks = LN_BWD_MAX_BLOCKS;
    const uint32_t thr = drop_thr16(dropout_p);
    void* dxd = thr ? dx_drop : nullptr;
    if (dx_drop && !thr) APTAI_FAIL(APTAI_ERR_INVALID, "aptai_layernorm_bwd: dx_drop given with dropout_p == 0");
#define LN_BWD_G(NCH, G) APTAI_LAUNCH((ln_bwd_kernel<NCH, G>), dim3((unsigned)blocks), dim3(256), 0, stream, (const bf16_t*)dy, (const bf16_t*)x, mean, rstd, gamma, (const bf16_t*)dres, (bf16_t*)dx, (bf16_t*)dxd, (uint32_t)seed, (uint32_t)(seed >> 32), thr, drop_scale(thr), (float*)workspace, (long)rows, beta_if_gelu_after, aptai_seed_salt(stream))
#define LN_BWD(NCH) do { if (beta_if_gelu_after) LN_BWD_G(NCH, true); else LN_BWD_G(NCH, false); } while (0)
    switch (cols / 256) {
        case 1: LN_BWD(1); break;
        case 2: LN_BWD(2); break;
        case 3: LN_BWD(3); break;
        default: LN_BWD(4); break;
    }
#undef LN_BWD_G
#undef LN_BWD
    APTAI_CHECK_LAUNCH("ln_bwd_kernel");
    if (dgamma || dbeta) {
        const int n = 2 * (int)cols;
        APTAI_LAUNCH(colsum_partials_kernel, dim3((n + 15) / 16), dim3(1024), 0, stream,
                           (const float*)workspace, dgamma, dbeta, (int)blocks, (int)cols);
        APTAI_CHECK_LAUNCH("colsum_partials_kernel");
    }
    return APTAI_OK;
}
