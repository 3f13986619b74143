// HBM-bound helper kernels of the wav2vec2 hot path (gfx950): parameter casts / re-layouts, frame masking
// (HF:678-681 zeroing of padded frames + HF:1292-1295 SpecAugment fill), bias-gradient column sums,
// positional-conv packing (HF:326-379) and the activation shims of the APTAI heads (models/aptai.py:43-55).
// All are grid-stride, 8/16-byte vectorised where the layout allows.
#include "common.h"

namespace {

__global__ void cast_f32_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long n4) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const f32x4 v = ((const f32x4*)src)[i];
        ((u32x2*)dst)[i] = (u32x2){pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
    }
}

// job table row: {src, dst, n (multiple of 8), kind}; kind 0: fp32 -> bf16, kind 1: fp32 -> fp32 copy.  grid (chunks, jobs):
// every block converts one 8192-element chunk of its job, so the ~90 parameter casts of a train step are ONE launch.
__global__ void cast_multi_kernel(const int64_t* __restrict__ table) {
    const int64_t* job = table + (long)blockIdx.y * 4;
    const long n = job[2];
    const long base = (long)blockIdx.x * 8192;
    if (base >= n) return;
    const float* src = (const float*)job[0];
    const bool copy = job[3] != 0;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const long i = base + (long)(it * 256 + threadIdx.x) * 8;
        if (i >= n) break;
        const f32x4 a = *(const f32x4*)(src + i), b = *(const f32x4*)(src + i + 4);
        if (copy) {
            float* dst = (float*)job[1];
            *(f32x4*)(dst + i) = a;
            *(f32x4*)(dst + i + 4) = b;
        } else {
            bf16_t* dst = (bf16_t*)job[1];
            *(u32x4*)(dst + i) = (u32x4){pack2bf(a[0], a[1]), pack2bf(a[2], a[3]), pack2bf(b[0], b[1]), pack2bf(b[2], b[3])};
        }
    }
}

// rows x cols fp32 -> bf16 with a destination leading dimension (packs q/k/v weights into one [3H][H] buffer)
__global__ void cast_rows_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long rows, long cols, long ldd) {
    const long n = rows * cols;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const long r = i / cols, c = i % cols;
        dst[r * ldd + c] = f2bf(src[i]);
    }
}

// conv weight [N][C][Kw] fp32 -> [N][Kw][C] bf16 (K index = kw*C + c matches channels-last frames)
__global__ void conv_weight_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int N, int C, int Kw) {
    const long n = (long)N * C * Kw;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int c = (int)(i % C);
        const int kw = (int)((i / C) % Kw);
        const int nn = (int)(i / ((long)C * Kw));
        dst[i] = f2bf(src[((long)nn * C + c) * Kw + kw]);
    }
}

// ---- weight-norm (dim=2) of the positional conv: norm[kk] = ||v[:, :, kk]||_2   (HF:340-356)
// Two deterministic stages over coalesced reads: block b sums its contiguous slice of v with thread t owning tap t % Kw
// (blockDim % Kw == 0 and slice % blockDim == 0, so a thread's tap never changes), then Kw threads add the block partials.
constexpr int PCN_BLOCKS = 256;            // one block per CU: 64 blocks read the 19 MB at 0.9 TB/s (22 us)
__global__ __launch_bounds__(256) void posconv_norm_partial_kernel(const float* __restrict__ v, float* __restrict__ partial,
                                                                   long n, int Kw) {
    __shared__ float red[256];
    const long per = ((n + PCN_BLOCKS - 1) / PCN_BLOCKS + 255) / 256 * 256;
    const long i0 = (long)blockIdx.x * per;
    long i1 = i0 + per;
    i1 = i1 < n ? i1 : n;
    float s8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) s8[u] = 0.f;
    long i = i0 + threadIdx.x;
    for (; i + 7 * 256 < i1; i += 8 * 256) {            // 8 independent loads in flight per thread
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = v[i + u * 256];
#pragma unroll
        for (int u = 0; u < 8; ++u) s8[u] = fmaf(x[u], x[u], s8[u]);
    }
    for (; i < i1; i += 256) {
        const float x = v[i];
        s8[0] = fmaf(x, x, s8[0]);
    }
    const float s = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
    red[threadIdx.x] = s;
    __syncthreads();
    if ((int)threadIdx.x < Kw) {
        float t = 0.f;
        for (int j = threadIdx.x; j < 256; j += Kw) t += red[j];
        partial[(long)blockIdx.x * Kw + threadIdx.x] = t;
    }
}
// sum_r partial[r][kk] for one tap kk per block: 256 threads take rows r = t, t + 256, ... (independent loads), then a
// fixed-order LDS tree.  (A thread per tap walking all rows serially is a chain of dependent L2 round trips: 178 us for 768 rows.)
__device__ __forceinline__ float tap_total(const float* __restrict__ partial, int R, int Kw, int kk, float* red) {
    float s = 0.f;
    for (int r = threadIdx.x; r < R; r += 256) s += partial[(long)r * Kw + kk];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    return red[0];
}
__global__ __launch_bounds__(256) void posconv_norm_final_kernel(const float* __restrict__ partial, float* __restrict__ norm, int Kw) {
    __shared__ float red[256];
    const float t = tap_total(partial, PCN_BLOCKS, Kw, blockIdx.x, red);
    if (threadIdx.x == 0) norm[blockIdx.x] = sqrtf(t);
}

// w = g*v/norm ->  fwd layout  Wf[grp][n][kk*Cg + c]            = w[grp*Cg+n][c][kk]
//                  dgrad layout Wd[grp][c][kk'*Cg + n], kk'=Kw-1-kk (flipped taps, in/out swapped)
// One block per OUTPUT row (H rows of Wf, then H rows of Wd): the Cg x Kw source slab is read in 512-byte runs, transposed
// through LDS and written as one contiguous bf16 row (both sides coalesced; element-wise scatter took 47 us).
__global__ __launch_bounds__(256) void posconv_weight_kernel(const float* __restrict__ v, const float* __restrict__ gain,
                                                             const float* __restrict__ norm, bf16_t* __restrict__ wf,
                                                             bf16_t* __restrict__ wd, int H, int Cg, int Kw) {
    extern __shared__ float slab[];                    // [Cg][Kw + 1]
    const bool dgrad = (int)blockIdx.x >= H;
    const int row = dgrad ? blockIdx.x - H : blockIdx.x;
    const int grp = row / Cg, r = row % Cg;            // fwd: r = out channel n; dgrad: r = in channel c
    const int n = Cg * Kw;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int a = i / Kw, kk = i % Kw;             // fwd: a = c (source row o = row);  dgrad: a = n (source o = grp*Cg + a, c = r)
        const long src = dgrad ? (((long)(grp * Cg + a) * Cg + r) * Kw + kk) : ((long)row * n + i);
        slab[a * (Kw + 1) + kk] = v[src] * gain[kk] / norm[kk];
    }
    __syncthreads();
    bf16_t* out = (dgrad ? wd : wf) + (long)row * n;
    for (int j = threadIdx.x; j < n; j += 256) {
        const int kk = j / Cg, a = j % Cg;             // output index kk*Cg + a
        out[j] = f2bf(slab[a * (Kw + 1) + (dgrad ? Kw - 1 - kk : kk)]);
    }
}

// ---- weight-norm backward of the positional conv (w = gain * v / ||v||_(0,1) per tap, HF:340-356) from the weight gradient in the
// forward GEMM layout dwf[grp][n][kk*Cg + c]:   dot[kk] = sum_{o,c} dW[o][c][kk] v[o][c][kk],   dgain[kk] = dot / norm,
// dv = gain / norm * (dW - v * dot / norm^2).  One block per output channel o: its dwf row is transposed through LDS so both the
// dwf reads and the v / dv accesses are contiguous.  Pass 1 writes per-block tap sums, pass 2 (after the tiny final sum) writes dv.
__global__ __launch_bounds__(256) void posconv_wn_dot_kernel(const float* __restrict__ dwf, const float* __restrict__ v,
                                                             float* __restrict__ partial, int Cg, int Kw) {
    extern __shared__ float slab[];                    // [Kw][Cg + 1]: dwf row o as [kk][c]
    __shared__ float red[256];
    const int o = blockIdx.x, n = Cg * Kw;
    for (int i = threadIdx.x; i < n; i += 256) slab[(i / Cg) * (Cg + 1) + i % Cg] = dwf[(long)o * n + i];
    __syncthreads();
    float s = 0.f;                                     // thread t owns tap t % Kw (256 % Kw == 0)
    for (int i = threadIdx.x; i < n; i += 256) {
        const int c = i / Kw, kk = i % Kw;
        s = fmaf(slab[kk * (Cg + 1) + c], v[(long)o * n + i], s);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if ((int)threadIdx.x < Kw) {
        float t = 0.f;
        for (int j = threadIdx.x; j < 256; j += Kw) t += red[j];
        partial[(long)o * Kw + threadIdx.x] = t;
    }
}
__global__ __launch_bounds__(256) void posconv_wn_final_kernel(const float* __restrict__ partial, const float* __restrict__ norm,
                                                               float* __restrict__ dot, float* __restrict__ dgain, int H, int Kw) {
    __shared__ float red[256];
    const int kk = blockIdx.x;
    const float s = tap_total(partial, H, Kw, kk, red);
    if (threadIdx.x == 0) {
        dot[kk] = s;
        dgain[kk] = s / norm[kk];
    }
}
__global__ __launch_bounds__(256) void posconv_wn_apply_kernel(const float* __restrict__ dwf, const float* __restrict__ v,
                                                               const float* __restrict__ gain, const float* __restrict__ norm,
                                                               const float* __restrict__ dot, float* __restrict__ dv, int Cg, int Kw) {
    extern __shared__ float slab[];
    const int o = blockIdx.x, n = Cg * Kw;
    for (int i = threadIdx.x; i < n; i += 256) slab[(i / Cg) * (Cg + 1) + i % Cg] = dwf[(long)o * n + i];
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) {
        const int c = i / Kw, kk = i % Kw;
        const float nr = norm[kk];
        dv[(long)o * n + i] = gain[kk] / nr * (slab[kk * (Cg + 1) + c] - v[(long)o * n + i] * dot[kk] / (nr * nr));
    }
}

// x [B*Tp][H] -> Xg[grp][b][pad | Tp | pad][Cg]  (gap rows are never written: the buffer is zeroed once)
// optional: multiply by gelu'(u) first (du = dy * gelu'(u)) and also emit the row-major product.
__global__ void posconv_pack_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ u, bf16_t* __restrict__ xg,
                                    bf16_t* __restrict__ rowmajor_out, int B, int Tp, int H, int Cg, int pad) {
    const long n4 = (long)B * Tp * H / 4;
    const long stride = (long)gridDim.x * blockDim.x;
    const int G = H / Cg;
    const long rows_p = Tp + 2 * pad;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const long e = i * 4;
        const long row = e / H;
        const int col = (int)(e % H);
        u32x2 p = *(const u32x2*)(x + e);
        if (u) {
            const u32x2 q = *(const u32x2*)(u + e);
            p = (u32x2){pack2bf(lo_bf(p[0]) * gelu_fast_grad(lo_bf(q[0])), hi_bf(p[0]) * gelu_fast_grad(hi_bf(q[0]))),
                        pack2bf(lo_bf(p[1]) * gelu_fast_grad(lo_bf(q[1])), hi_bf(p[1]) * gelu_fast_grad(hi_bf(q[1])))};
            if (rowmajor_out) *(u32x2*)(rowmajor_out + e) = p;
        }
        const int grp = col / Cg, c = col % Cg;          // Cg % 4 == 0: the 4 elements stay in one group
        const long b = row / Tp, t = row % Tp;
        *(u32x2*)(xg + (((long)grp * B + b) * rows_p + pad + t) * Cg + c) = p;
    }
    (void)G;
}

// ---- SpecAugment span sampler on the device (HF:101-217 `_compute_mask_indices`): same span-count rule (probabilistic
// rounding with ONE epsilon per call, min_masks, the two length clamps), span starts drawn WITHOUT replacement from
// [0, len - (mask_length - 1)) by Floyd's algorithm, spans clamped to the last frame.  Random stream: the counter hash instead of
// numpy's Mersenne twister - same distribution, different draws (parity tests pass explicit masks).  Exists so the training
// forward needs no device->host copy of the utterance lengths: that copy stalled the eager loop for 2.3 ms per step.
// One block of 64 threads per utterance.
constexpr int SPEC_MAX_SPANS = 256;
__device__ __forceinline__ uint32_t spec_rand(uint32_t ctr, uint32_t s0, uint32_t s1) { return rng_hash(ctr * 2654435761u + 12345u, s0, s1); }

__global__ __launch_bounds__(64) void spec_mask_kernel(const int* __restrict__ lens, uint8_t* __restrict__ mask, int T, float mask_prob,
                                                       int mask_length, int min_masks, uint32_t s0, uint32_t s1,
                                                       const uint32_t* __restrict__ salt) {
    __shared__ int starts[SPEC_MAX_SPANS];
    __shared__ int nspan_sh;
    apply_salt(salt, s0, s1);
    const int b = blockIdx.x, lane = threadIdx.x;
    uint8_t* row = mask + (long)b * T;
    for (int t = lane; t < T; t += 64) row[t] = 0;
    int len = lens[b];
    len = len < 0 ? 0 : (len > T ? T : len);
    if (lane == 0) {
        const float eps = (float)(spec_rand(0xE951u, s0, s1) >> 8) * (1.0f / 16777216.0f);      // shared by all utterances
        int n = (int)(mask_prob * (float)len / (float)mask_length + eps);
        n = n > min_masks ? n : min_masks;
        if (n * mask_length > T) n = T / mask_length;
        const int range = len - (mask_length - 1);
        if (range < n) n = range > 0 ? range : 0;
        n = n < SPEC_MAX_SPANS ? n : SPEC_MAX_SPANS;
        // Floyd: for j = range-n .. range-1: t = U[0, j]; take t unless already taken, else j
        for (int i = 0; i < n; ++i) {
            const int j = range - n + i;
            const int t = (int)(spec_rand((uint32_t)(b * SPEC_MAX_SPANS + i + 1), s0, s1) % (uint32_t)(j + 1));
            bool taken = false;
            for (int k = 0; k < i; ++k) taken |= (starts[k] == t);
            starts[i] = taken ? j : t;
        }
        nspan_sh = n;
    }
    __syncthreads();
    const int n = nspan_sh;
    for (int i = 0; i < n; ++i) {
        const int st = starts[i];
        for (int o = lane; o < mask_length; o += 64) {
            int t = st + o;
            t = t < T - 1 ? t : T - 1;
            row[t] = 1;
        }
    }
}

// ---- frame masking: out = pad ? 0 : (spec ? embed : h)   (in place)
__global__ void frame_mask_kernel(bf16_t* __restrict__ h, const int* __restrict__ lens, const uint8_t* __restrict__ spec,
                                  const float* __restrict__ embed, int B, int Tp, int T, int H) {
    const long n4 = (long)B * Tp * H / 4;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const long e = i * 4;
        const long row = e / H;
        const int col = (int)(e % H);
        const int b = (int)(row / Tp), t = (int)(row % Tp);
        int len = lens[b];
        len = len < T ? len : T;
        if (t >= len) {
            *(u32x2*)(h + e) = (u32x2){0u, 0u};
        } else if (spec && spec[(long)b * T + t]) {
            const f32x4 m = *(const f32x4*)(embed + col);
            *(u32x2*)(h + e) = (u32x2){pack2bf(m[0], m[1]), pack2bf(m[2], m[3])};
        }
    }
}

// backward: dh = (pad | spec) ? 0 : dy (in place); dembed partials over spec & !pad rows
__global__ void frame_mask_bwd_kernel(bf16_t* __restrict__ dy, const int* __restrict__ lens,
                                      const uint8_t* __restrict__ spec, float* __restrict__ partials, int B, int Tp, int T,
                                      int H, int rows_per_block) {
    // block handles rows [blockIdx.x*rpb, +rpb); thread handles 4 columns per pass
    const long rows = (long)B * Tp;
    const long r0 = (long)blockIdx.x * rows_per_block;
    for (int c4 = threadIdx.x; c4 < H / 4; c4 += blockDim.x) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int rr = 0; rr < rows_per_block; ++rr) {
            const long row = r0 + rr;
            if (row >= rows) break;
            const int b = (int)(row / Tp), t = (int)(row % Tp);
            int len = lens[b];
            len = len < T ? len : T;
            const long e = row * H + c4 * 4;
            if (t >= len) {
                *(u32x2*)(dy + e) = (u32x2){0u, 0u};
            } else if (spec && spec[(long)b * T + t]) {
                const u32x2 p = *(const u32x2*)(dy + e);
                acc[0] += lo_bf(p[0]); acc[1] += hi_bf(p[0]); acc[2] += lo_bf(p[1]); acc[3] += hi_bf(p[1]);
                *(u32x2*)(dy + e) = (u32x2){0u, 0u};
            }
        }
        if (partials) *(f32x4*)(partials + (long)blockIdx.x * H + c4 * 4) = (f32x4){acc[0], acc[1], acc[2], acc[3]};
    }
}

// ---- column sums of a bf16 matrix (bias gradients), HBM-bound: block = strip of 256 columns x COLSUM_RPB rows;
// each wave streams rows with 8-byte loads (lane = 4 columns), the 4 waves combine through LDS -> partials[rowblk][N]
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16_t* __restrict__ x, long ld, float* __restrict__ partials,
                                                          long rows, int N, int rows_per_block) {
    __shared__ float red[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.x * 256 + lane * 4;
    const long r0 = (long)blockIdx.y * rows_per_block;
    long r1 = r0 + rows_per_block;
    r1 = r1 < rows ? r1 : rows;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (col < N) {
        long row = r0 + wave;
        for (; row + 12 < r1; row += 16) {              // 4 independent 8-byte loads in flight per lane
            const u32x2 p0 = *(const u32x2*)(x + row * ld + col);
            const u32x2 p1 = *(const u32x2*)(x + (row + 4) * ld + col);
            const u32x2 p2 = *(const u32x2*)(x + (row + 8) * ld + col);
            const u32x2 p3 = *(const u32x2*)(x + (row + 12) * ld + col);
            acc[0] += (lo_bf(p0[0]) + lo_bf(p1[0])) + (lo_bf(p2[0]) + lo_bf(p3[0]));
            acc[1] += (hi_bf(p0[0]) + hi_bf(p1[0])) + (hi_bf(p2[0]) + hi_bf(p3[0]));
            acc[2] += (lo_bf(p0[1]) + lo_bf(p1[1])) + (lo_bf(p2[1]) + lo_bf(p3[1]));
            acc[3] += (hi_bf(p0[1]) + hi_bf(p1[1])) + (hi_bf(p2[1]) + hi_bf(p3[1]));
        }
        for (; row < r1; row += 4) {
            const u32x2 p = *(const u32x2*)(x + row * ld + col);
            acc[0] += lo_bf(p[0]); acc[1] += hi_bf(p[0]); acc[2] += lo_bf(p[1]); acc[3] += hi_bf(p[1]);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][lane * 4 + r] = acc[r];
    __syncthreads();
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < N)
        partials[(long)blockIdx.y * N + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// out[c] (+)= sum_b partials[b][c]: block = 64 columns x 4 slices
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partials, float* __restrict__ out, int nblocks,
                                                              int N, int accumulate) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (c < N)
        for (int b = slice; b < nblocks; b += 4) s += partials[(long)b * N + c];
    red[slice][lane] = s;
    __syncthreads();
    if (slice == 0 && c < N) {
        const float t = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
        out[c] = accumulate ? out[c] + t : t;
    }
}

// ---- APTAI head activations: a_tv = tanh(drop(h)), a_ph = leaky_relu(drop(h))   (models/aptai.py:43-55)
// 8 elements per thread (16-byte loads and stores), one hash per element PAIR (drop_hash_pair: the same mask drop_keep
// defines), tanh as 1 - 2 / (1 + 2^(2 log2e x)) on the bare v_exp / v_rcp (|err| < 1e-6, the output is bf16).
__device__ __forceinline__ float tanh_fast(float x) {
    const float xc = __builtin_amdgcn_fmed3f(x, -10.f, 10.f);
    return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(xc * 2.8853900817779268f));
}
__device__ __forceinline__ void keep_scales8(long e0, uint32_t s0, uint32_t s1, uint32_t thr, float sc, float* m) {
    if (!thr) {
#pragma unroll
        for (int r = 0; r < 8; ++r) m[r] = 1.f;
        return;
    }
#pragma unroll
    for (int r = 0; r < 8; r += 2) {
        const uint32_t hsh = drop_hash_pair((uint64_t)(e0 + r), s0, s1);
        m[r] = (hsh & 0xffffu) >= thr ? sc : 0.f;
        m[r + 1] = (hsh >> 16) >= thr ? sc : 0.f;
    }
}
__device__ __forceinline__ void unpack8(const u32x4 v, float* x) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { x[2 * r] = lo_bf(v[r]); x[2 * r + 1] = hi_bf(v[r]); }
}
__device__ __forceinline__ u32x4 pack8f(const float* x) {
    return (u32x4){pack2bf(x[0], x[1]), pack2bf(x[2], x[3]), pack2bf(x[4], x[5]), pack2bf(x[6], x[7])};
}

__global__ void head_act_fwd_kernel(const bf16_t* __restrict__ h, bf16_t* __restrict__ a_tv, bf16_t* __restrict__ a_ph,
                                    long n, uint32_t s0, uint32_t s1, uint32_t thr_tv, uint32_t thr_ph, float sc_tv,
                                    float sc_ph, const uint32_t* __restrict__ salt) {
    if (thr_tv | thr_ph) apply_salt(salt, s0, s1);
    const long stride = (long)gridDim.x * blockDim.x;
    const long n8 = n >> 3;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        float x[8], mt[8], mp[8], t[8], p[8];
        unpack8(*(const u32x4*)(h + i * 8), x);
        keep_scales8(i * 8, s0, s1, thr_tv, sc_tv, mt);
        keep_scales8(i * 8, s0 ^ 0x5bd1e995u, s1, thr_ph, sc_ph, mp);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float xp = x[r] * mp[r];
            t[r] = tanh_fast(x[r] * mt[r]);
            p[r] = xp > 0.f ? xp : 0.01f * xp;
        }
        *(u32x4*)(a_tv + i * 8) = pack8f(t);
        *(u32x4*)(a_ph + i * 8) = pack8f(p);
    }
    for (long i = (n8 << 3) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {      // n % 8 tail
        const float x = bf2f(h[i]);
        float xt = x, xp = x;
        if (thr_tv) xt = drop_keep((uint64_t)i, s0, s1, thr_tv) ? x * sc_tv : 0.f;
        if (thr_ph) xp = drop_keep((uint64_t)i, s0 ^ 0x5bd1e995u, s1, thr_ph) ? x * sc_ph : 0.f;
        a_tv[i] = f2bf(tanh_fast(xt));
        a_ph[i] = f2bf(xp > 0.f ? xp : 0.01f * xp);
    }
}

// dh = d_tv * (1 - tanh^2) * mask_tv + d_ph * leaky' * mask_ph
__global__ void head_act_bwd_kernel(const bf16_t* __restrict__ h, const bf16_t* __restrict__ d_tv,
                                    const bf16_t* __restrict__ d_ph, bf16_t* __restrict__ dh, long n, uint32_t s0,
                                    uint32_t s1, uint32_t thr_tv, uint32_t thr_ph, float sc_tv, float sc_ph,
                                    const uint32_t* __restrict__ salt) {
    if (thr_tv | thr_ph) apply_salt(salt, s0, s1);
    const long stride = (long)gridDim.x * blockDim.x;
    const long n8 = n >> 3;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        float x[8], mt[8], mp[8], gt[8], gp[8], g[8];
        unpack8(*(const u32x4*)(h + i * 8), x);
        unpack8(*(const u32x4*)(d_tv + i * 8), gt);
        unpack8(*(const u32x4*)(d_ph + i * 8), gp);
        keep_scales8(i * 8, s0, s1, thr_tv, sc_tv, mt);
        keep_scales8(i * 8, s0 ^ 0x5bd1e995u, s1, thr_ph, sc_ph, mp);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float th = tanh_fast(x[r] * mt[r]);
            g[r] = gt[r] * (1.f - th * th) * mt[r] + gp[r] * (x[r] * mp[r] > 0.f ? 1.f : 0.01f) * mp[r];
        }
        *(u32x4*)(dh + i * 8) = pack8f(g);
    }
    for (long i = (n8 << 3) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float x = bf2f(h[i]);
        float xt = x, xp = x, mt = 1.f, mp = 1.f;
        if (thr_tv) { mt = drop_keep((uint64_t)i, s0, s1, thr_tv) ? sc_tv : 0.f; xt = x * mt; }
        if (thr_ph) { mp = drop_keep((uint64_t)i, s0 ^ 0x5bd1e995u, s1, thr_ph) ? sc_ph : 0.f; xp = x * mp; }
        const float th = tanh_fast(xt);
        dh[i] = f2bf(bf2f(d_tv[i]) * (1.f - th * th) * mt + bf2f(d_ph[i]) * (xp > 0.f ? 1.f : 0.01f) * mp);
    }
}

// generic elementwise dropout apply (forward or backward): y = keep ? x*scale : 0
__global__ void dropout_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, long n, uint32_t s0, uint32_t s1,
                               uint32_t thr, float sc, const uint32_t* __restrict__ salt) {
    apply_salt(salt, s0, s1);
    const long stride = (long)gridDim.x * blockDim.x;
    const long n8 = n >> 3;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        float v[8], m[8];
        unpack8(*(const u32x4*)(x + i * 8), v);
        keep_scales8(i * 8, s0, s1, thr, sc, m);
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] *= m[r];
        *(u32x4*)(y + i * 8) = pack8f(v);
    }
    for (long i = (n8 << 3) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        y[i] = drop_keep((uint64_t)i, s0, s1, thr) ? f2bf(bf2f(x[i]) * sc) : (bf16_t)0;
}

// out = dy * gelu'(u)   (backward of a standalone GELU; 8 elements per thread)
__global__ void dgelu_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ u, bf16_t* __restrict__ out, long n8) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        const u32x4 d = ((const u32x4*)dy)[i], x = ((const u32x4*)u)[i];
        u32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            o[r] = pack2bf(lo_bf(d[r]) * gelu_fast_grad(lo_bf(x[r])), hi_bf(d[r]) * gelu_fast_grad(hi_bf(x[r])));
        ((u32x4*)out)[i] = o;
    }
}

inline unsigned grid_for(long n, int block = 256, int max_blocks = 4096) {
    long b = ceil_div(n, block);
    return (unsigned)(b < 1 ? 1 : (b > max_blocks ? max_blocks : b));
}

}  // namespace

extern "C" int aptai_cast_f32_to_bf16(const float* src, void* dst, int64_t rows, int64_t cols, int64_t ld_dst, void* stream) {
    APTAI_REQUIRE(src && dst && rows > 0 && cols > 0, "aptai_cast_f32_to_bf16: bad arguments");
    if (ld_dst == cols && (rows * cols) % 4 == 0 && ((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 8 == 0)) {
        const long n4 = rows * cols / 4;
        APTAI_LAUNCH(cast_f32_bf16_kernel, dim3(grid_for(n4)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n4);
    } else {
        APTAI_LAUNCH(cast_rows_kernel, dim3(grid_for(rows * cols)), dim3(256), 0, (hipStream_t)stream, src,
                           (bf16_t*)dst, (long)rows, (long)cols, (long)ld_dst);
    }
    APTAI_CHECK_LAUNCH("cast kernel");
    return APTAI_OK;
}

extern "C" int aptai_cast_multi(const int64_t* table_dev, int64_t njobs, int64_t max_n, void* stream) {
    APTAI_REQUIRE(table_dev && njobs > 0 && njobs <= 65535 && max_n > 0, "aptai_cast_multi: bad arguments");
    const long chunks = ceil_div(max_n, 8192);
    APTAI_LAUNCH(cast_multi_kernel, dim3((unsigned)chunks, (unsigned)njobs), dim3(256), 0, (hipStream_t)stream, table_dev);
    APTAI_CHECK_LAUNCH("cast_multi_kernel");
    return APTAI_OK;
}

extern "C" int aptai_conv_weight_to_bf16(const float* src, void* dst, int64_t N, int64_t C, int64_t Kw, void* stream) {
    APTAI_REQUIRE(src && dst && N > 0 && C > 0 && Kw > 0, "aptai_conv_weight_to_bf16: bad arguments");
    APTAI_LAUNCH(conv_weight_kernel, dim3(grid_for(N * C * Kw)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst,
                       (int)N, (int)C, (int)Kw);
    APTAI_CHECK_LAUNCH("conv_weight_kernel");
    return APTAI_OK;
}

extern "C" int aptai_posconv_weight(const float* v, const float* gain, float* norm_ws, void* w_fwd, void* w_dgrad,
                                    int64_t H, int64_t groups, int64_t Kw, void* stream) {
    APTAI_REQUIRE(v && gain && norm_ws && w_fwd, "aptai_posconv_weight: null pointer");
    APTAI_REQUIRE(groups > 0 && H % groups == 0 && (H / groups) % 8 == 0, "aptai_posconv_weight: H=%ld groups=%ld", (long)H, (long)groups);
    const int Cg = (int)(H / groups);
    APTAI_REQUIRE(Kw > 0 && 256 % Kw == 0, "aptai_posconv_weight: Kw=%ld must divide 256", (long)Kw);
    const long nel = (long)H * Cg * Kw;
    float* partial = norm_ws + Kw;                      // [256][Kw] scratch behind the result
    APTAI_LAUNCH(posconv_norm_partial_kernel, dim3(PCN_BLOCKS), dim3(256), 0, (hipStream_t)stream, v, partial, nel, (int)Kw);
    APTAI_CHECK_LAUNCH("posconv_norm_partial_kernel");
    APTAI_LAUNCH(posconv_norm_final_kernel, dim3((unsigned)Kw), dim3(256), 0, (hipStream_t)stream, (const float*)partial, norm_ws, (int)Kw);
    APTAI_CHECK_LAUNCH("posconv_norm_final_kernel");
    const size_t slab_bytes = (size_t)Cg * (Kw + 1) * 4;
    APTAI_REQUIRE(slab_bytes <= 64 * 1024, "aptai_posconv_weight: Cg*Kw slab of %ld bytes exceeds the LDS budget", (long)slab_bytes);
    APTAI_LAUNCH(posconv_weight_kernel, dim3((unsigned)(w_dgrad ? 2 * H : H)), dim3(256), slab_bytes, (hipStream_t)stream, v, gain,
                       (const float*)norm_ws, (bf16_t*)w_fwd, (bf16_t*)w_dgrad, (int)H, Cg, (int)Kw);
    APTAI_CHECK_LAUNCH("posconv_weight_kernel");
    return APTAI_OK;
}

extern "C" int aptai_spec_augment_mask(const int32_t* frame_lens, void* mask_u8, int64_t B, int64_t T, float mask_prob, int64_t mask_length,
                                       int64_t min_masks, uint64_t seed, void* stream) {
    APTAI_REQUIRE(frame_lens && mask_u8 && B > 0 && T > 0, "aptai_spec_augment_mask: bad arguments");
    APTAI_REQUIRE(mask_length >= 1 && mask_length <= T, "aptai_spec_augment_mask: mask_length=%ld must lie in [1, T=%ld]", (long)mask_length, (long)T);
    APTAI_REQUIRE(mask_prob >= 0.f && mask_prob <= 1.f && min_masks >= 0, "aptai_spec_augment_mask: bad mask_prob / min_masks");
    APTAI_LAUNCH(spec_mask_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, frame_lens, (uint8_t*)mask_u8, (int)T, mask_prob,
                 (int)mask_length, (int)min_masks, (uint32_t)seed, (uint32_t)(seed >> 32), aptai_seed_salt(stream));
    APTAI_CHECK_LAUNCH("spec_mask_kernel");
    return APTAI_OK;
}

extern "C" int aptai_posconv_weight_bwd(const float* dw_fwd, const float* v, const float* gain, const float* norm, float* dv, float* dgain,
                                       float* workspace, int64_t H, int64_t groups, int64_t Kw, void* stream) {
    APTAI_REQUIRE(dw_fwd && v && gain && norm && dv && dgain && workspace, "aptai_posconv_weight_bwd: null pointer");
    APTAI_REQUIRE(groups > 0 && H % groups == 0 && Kw > 0 && 256 % Kw == 0, "aptai_posconv_weight_bwd: H=%ld groups=%ld Kw=%ld", (long)H, (long)groups, (long)Kw);
    const int Cg = (int)(H / groups);
    const size_t slab_bytes = (size_t)Kw * (Cg + 1) * 4;
    APTAI_REQUIRE(slab_bytes <= 64 * 1024, "aptai_posconv_weight_bwd: slab of %ld bytes exceeds the LDS budget", (long)slab_bytes);
    float* partial = workspace;                         // [H][Kw]
    float* dot = workspace + H * Kw;                    // [Kw]
    hipStream_t st = (hipStream_t)stream;
    APTAI_LAUNCH(posconv_wn_dot_kernel, dim3((unsigned)H), dim3(256), slab_bytes, st, dw_fwd, v, partial, Cg, (int)Kw);
    APTAI_CHECK_LAUNCH("posconv_wn_dot_kernel");
    APTAI_LAUNCH(posconv_wn_final_kernel, dim3((unsigned)Kw), dim3(256), 0, st, (const float*)partial, norm, dot, dgain, (int)H, (int)Kw);
    APTAI_CHECK_LAUNCH("posconv_wn_final_kernel");
    APTAI_LAUNCH(posconv_wn_apply_kernel, dim3((unsigned)H), dim3(256), slab_bytes, st, dw_fwd, v, gain, norm, (const float*)dot, dv, Cg, (int)Kw);
    APTAI_CHECK_LAUNCH("posconv_wn_apply_kernel");
    return APTAI_OK;
}

extern "C" int aptai_posconv_pack(const void* x, const void* u, void* xg, void* rowmajor_out, int64_t B, int64_t Tp,
                                  int64_t H, int64_t groups, int64_t pad, void* stream) {
    APTAI_REQUIRE(x && xg, "aptai_posconv_pack: null pointer");
    APTAI_REQUIRE(groups > 0 && H % groups == 0 && (H / groups) % 4 == 0, "aptai_posconv_pack: H=%ld groups=%ld", (long)H, (long)groups);
    APTAI_LAUNCH(posconv_pack_kernel, dim3(grid_for(B * Tp * H / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, (const bf16_t*)u, (bf16_t*)xg, (bf16_t*)rowmajor_out, (int)B, (int)Tp, (int)H,
                       (int)(H / groups), (int)pad);
    APTAI_CHECK_LAUNCH("posconv_pack_kernel");
    return APTAI_OK;
}

extern "C" int aptai_frame_mask_fwd(void* h, const int32_t* lens, const uint8_t* spec_mask, const float* embed, int64_t B,
                                    int64_t Tp, int64_t T, int64_t H, void* stream) {
    APTAI_REQUIRE(h && lens && H % 4 == 0, "aptai_frame_mask_fwd: bad arguments");
    APTAI_REQUIRE(!spec_mask || embed, "aptai_frame_mask_fwd: spec mask without masked_spec_embed");
    APTAI_LAUNCH(frame_mask_kernel, dim3(grid_for(B * Tp * H / 4)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)h, lens,
                       spec_mask, embed, (int)B, (int)Tp, (int)T, (int)H);
    APTAI_CHECK_LAUNCH("frame_mask_kernel");
    return APTAI_OK;
}

static const int MASK_BWD_RPB = 32;
extern "C" int64_t aptai_frame_mask_bwd_workspace_bytes(int64_t B, int64_t Tp, int64_t H) {
    return ceil_div(B * Tp, MASK_BWD_RPB) * H * 4;
}
extern "C" int aptai_frame_mask_bwd(void* dy, const int32_t* lens, const uint8_t* spec_mask, float* dembed, void* workspace,
                                    int64_t B, int64_t Tp, int64_t T, int64_t H, void* stream) {
    APTAI_REQUIRE(dy && lens && H % 4 == 0, "aptai_frame_mask_bwd: bad arguments");
    const long blocks = ceil_div(B * Tp, MASK_BWD_RPB);
    const bool want = spec_mask && dembed;
    APTAI_REQUIRE(!want || workspace, "aptai_frame_mask_bwd: workspace needed for dembed");
    APTAI_LAUNCH(frame_mask_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (bf16_t*)dy, lens,
                       spec_mask, want ? (float*)workspace : nullptr, (int)B, (int)Tp, (int)T, (int)H, MASK_BWD_RPB);
    APTAI_CHECK_LAUNCH("frame_mask_bwd_kernel");
    if (want) {
        APTAI_LAUNCH(reduce_partials_kernel, dim3((unsigned)ceil_div(H, 64)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)workspace, dembed, (int)blocks, (int)H, 0);
        APTAI_CHECK_LAUNCH("reduce_partials_kernel");
    }
    return APTAI_OK;
}

static const int COLSUM_RPB = 128;
extern "C" int64_t aptai_colsum_workspace_bytes(int64_t rows, int64_t N) { return ceil_div(rows, COLSUM_RPB) * N * 4; }
extern "C" int aptai_colsum_bf16(const void* x, int64_t ld, float* out, void* workspace, int64_t rows, int64_t N,
                                 int accumulate, void* stream) {
    APTAI_REQUIRE(x && out && workspace && rows > 0 && N % 4 == 0 && ld % 4 == 0, "aptai_colsum_bf16: bad arguments");
    const long blocks = ceil_div(rows, COLSUM_RPB);
    APTAI_LAUNCH(colsum_bf16_kernel, dim3((unsigned)ceil_div(N, 256), (unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                 (const bf16_t*)x, (long)ld, (float*)workspace, (long)rows, (int)N, COLSUM_RPB);
    APTAI_CHECK_LAUNCH("colsum_bf16_kernel");
    APTAI_LAUNCH(reduce_partials_kernel, dim3((unsigned)ceil_div(N, 64)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)workspace, out, (int)blocks, (int)N, accumulate);
    APTAI_CHECK_LAUNCH("reduce_partials_kernel");
    return APTAI_OK;
}

extern "C" int aptai_head_act_fwd(const void* h, void* a_tv, void* a_ph, int64_t n, float p_tv, float p_ph, uint64_t seed,
                                  void* stream) {
    APTAI_REQUIRE(h && a_tv && a_ph && n > 0, "aptai_head_act_fwd: bad arguments");
    APTAI_REQUIRE((((uintptr_t)h | (uintptr_t)a_tv | (uintptr_t)a_ph) & 15) == 0, "aptai_head_act_fwd: buffers must be 16-byte aligned");
    const uint32_t t1 = drop_thr16(p_tv), t2 = drop_thr16(p_ph);
    APTAI_LAUNCH(head_act_fwd_kernel, dim3(grid_for(n / 8 + 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)h,
                       (bf16_t*)a_tv, (bf16_t*)a_ph, (long)n, (uint32_t)seed, (uint32_t)(seed >> 32), t1, t2, drop_scale(t1),
                       drop_scale(t2), aptai_seed_salt(stream));
    APTAI_CHECK_LAUNCH("head_act_fwd_kernel");
    return APTAI_OK;
}

extern "C" int aptai_head_act_bwd(const void* h, const void* d_tv, const void* d_ph, void* dh, int64_t n, float p_tv,
                                  float p_ph, uint64_t seed, void* stream) {
    APTAI_REQUIRE(h && d_tv && d_ph && dh && n > 0, "aptai_head_act_bwd: bad arguments");
    APTAI_REQUIRE((((uintptr_t)h | (uintptr_t)d_tv | (uintptr_t)d_ph | (uintptr_t)dh) & 15) == 0, "aptai_head_act_bwd: buffers must be 16-byte aligned");
    const uint32_t t1 = drop_thr16(p_tv), t2 = drop_thr16(p_ph);
    APTAI_LAUNCH(head_act_bwd_kernel, dim3(grid_for(n / 8 + 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)h,
                       (const bf16_t*)d_tv, (const bf16_t*)d_ph, (bf16_t*)dh, (long)n, (uint32_t)seed, (uint32_t)(seed >> 32),
                       t1, t2, drop_scale(t1), drop_scale(t2), aptai_seed_salt(stream));
    APTAI_CHECK_LAUNCH("head_act_bwd_kernel");
    return APTAI_OK;
}

extern "C" int aptai_dropout_bf16(const void* x, void* y, int64_t n, float p, uint64_t seed, void* stream) {
    APTAI_REQUIRE(x && y && n > 0, "aptai_dropout_bf16: bad arguments");
    APTAI_REQUIRE((((uintptr_t)x | (uintptr_t)y) & 15) == 0, "aptai_dropout_bf16: buffers must be 16-byte aligned");
    const uint32_t t = drop_thr16(p);
    APTAI_LAUNCH(dropout_kernel, dim3(grid_for(n / 8 + 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y,
                       (long)n, (uint32_t)seed, (uint32_t)(seed >> 32), t, drop_scale(t), aptai_seed_salt(stream));
    APTAI_CHECK_LAUNCH("dropout_kernel");
    return APTAI_OK;
}

extern "C" int aptai_dgelu_bf16(const void* dy, const void* u, void* out, int64_t n, void* stream) {
    APTAI_REQUIRE(dy && u && out && n > 0 && n % 8 == 0, "aptai_dgelu_bf16: bad arguments (n %% 8 == 0)");
    APTAI_LAUNCH(dgelu_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, (const bf16_t*)u,
                 (bf16_t*)out, (long)(n / 8));
    APTAI_CHECK_LAUNCH("dgelu_kernel");
    return APTAI_OK;
}
