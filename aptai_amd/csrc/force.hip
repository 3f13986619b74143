// Force_APTAI aligner heads in fp32 (gfx950): everything after the frozen wav2vec2 encoder in
// models/force_aptai.py:108-161 — Embedding + sinusoidal PE (:118-119), frame Linear (:122), CrossAttention
// (models/modules.py:139-153), log-softmax alignment + argmax read-out (:128-130,148-161), BiLSTM + MLP
// (models/modules.py:195-214).  These layers are small (128/256 wide) and feed an ARGMAX whose indices must match the
// reference, so they run in fp32 (VALU FMA), not bf16 MFMA.  The forward-sum loss reuses the CTC kernels (ctc.hip).
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------ generic fp32 GEMM
// C[m][n] (+)= alpha * sum_k A(m,k) * B(k,n) + bias[n];  A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn].
// 64x64x16 tile, 256 threads, 4x4 outputs per thread.  A may be bf16 (the encoder's hidden states).
struct SgemmArgs {
    const void* A; const float* B; float* C; const float* bias;
    long sam, sak, sbk, sbn, ldc;
    long bsa, bsb, bsc;           // batch strides (elements)
    int M, N, K, a_bf16, accumulate;
    float alpha;
};

__global__ __launch_bounds__(256) void sgemm_kernel(SgemmArgs g) {
    __shared__ float As[16][68];
    __shared__ float Bs[16][68];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const long boff = blockIdx.z;
    const float* Af = (const float*)g.A + boff * g.bsa;
    const bf16_t* Ab = (const bf16_t*)g.A + boff * g.bsa;
    const float* B = g.B + boff * g.bsb;
    float* C = g.C + boff * g.bsc;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int k0 = 0; k0 < g.K; k0 += 16) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = e * 256 + tid;
            {   // A tile: 64 (m) x 16 (k); consecutive threads walk the contiguous axis
                int mm, kk;
                if (g.sak == 1) { kk = idx & 15; mm = idx >> 4; } else { mm = idx & 63; kk = idx >> 6; }
                const int m = m0 + mm, k = k0 + kk;
                float v = 0.f;
                if (m < g.M && k < g.K) {
                    const long off = (long)m * g.sam + (long)k * g.sak;
                    v = g.a_bf16 ? bf2f(Ab[off]) : Af[off];
                }
                As[kk][mm] = v;
            }
            {   // B tile: 16 (k) x 64 (n)
                int nn, kk;
                if (g.sbn == 1) { nn = idx & 63; kk = idx >> 6; } else { kk = idx & 15; nn = idx >> 4; }
                const int n = n0 + nn, k = k0 + kk;
                float v = 0.f;
                if (n < g.N && k < g.K) v = B[(long)k * g.sbk + (long)n * g.sbn];
                Bs[kk][nn] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[kk][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bs[kk][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty * 4 + i;
        if (m >= g.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n >= g.N) continue;
            float v = acc[i][j] * g.alpha + (g.bias ? g.bias[n] : 0.f);
            float* c = C + (long)m * g.ldc + n;
            *c = g.accumulate ? *c + v : v;
        }
    }
}

// ------------------------------------------------------------------------------------------ embedding + PE
__global__ void embed_pe_fwd_kernel(const int* __restrict__ ids, const float* __restrict__ emb, const float* __restrict__ pe,
                                    float* __restrict__ out, int rows, int N, int D, float scale_keep, uint32_t s0, uint32_t s1,
                                    uint32_t thr) {
    const long n = (long)rows * D;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / D), d = (int)(i % D);
        float v = emb[(long)ids[r] * D + d] + pe[(long)(r % N) * D + d];
        if (thr) v = drop_keep((uint64_t)i, s0, s1, thr) ? v * scale_keep : 0.f;
        out[i] = v;
    }
}
__global__ void embed_bwd_kernel(const int* __restrict__ ids, const float* __restrict__ dout, float* __restrict__ demb, int rows,
                                 int D, float scale_keep, uint32_t s0, uint32_t s1, uint32_t thr) {
    const long n = (long)rows * D;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / D), d = (int)(i % D);
        const int id = ids[r];
        if (id == 0) continue;                                      // padding_idx = 0 receives no gradient
        float v = dout[i];
        if (thr) v = drop_keep((uint64_t)i, s0, s1, thr) ? v * scale_keep : 0.f;
        atomicAdd(&demb[(long)id * D + d], v);
    }
}

// ------------------------------------------------------------------------------------------ cross-attention softmaxes
// one wave per (b,t) row, lane = phoneme slot (N <= 64).  raw -> energy = raw + mask1; att = softmax(energy);
// att_log = log_softmax(energy + mask1) (the reference adds the -1000 mask twice); align = argmax(att_log) (first max)
__global__ __launch_bounds__(256) void xattn_softmax_fwd_kernel(const float* __restrict__ raw, const int* __restrict__ ids,
                                                                float* __restrict__ energy, float* __restrict__ att,
                                                                float* __restrict__ att_log, int64_t* __restrict__ align, int B,
                                                                int T, int N) {
    const long row = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (row >= (long)B * T) return;
    const int b = (int)(row / T);
    const bool ok = lane < N;
    const float m1 = ok ? ((ids[b * N + lane] != 0) ? 0.f : -1000.f) : 0.f;
    const float e = ok ? raw[row * N + lane] + m1 : -INFINITY;
    float mx = wave_max(e);
    float ex = ok ? __expf(e - mx) : 0.f;
    float se = wave_sum(ex);
    if (ok) { energy[row * N + lane] = e; att[row * N + lane] = ex / se; }
    const float e2 = ok ? e + m1 : -INFINITY;
    mx = wave_max(e2);
    ex = ok ? expf(e2 - mx) : 0.f;
    se = wave_sum(ex);
    const float al = e2 - (mx + logf(se));
    if (ok) att_log[row * N + lane] = al;
    // argmax with first-index tie break
    float best = ok ? al : -INFINITY;
    int bi = lane;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0 && align) align[row] = bi;
}
// d_raw = att*(d_att - sum(att*d_att)) + d_attlog - exp(att_log)*sum(d_attlog)
__global__ __launch_bounds__(256) void xattn_softmax_bwd_kernel(const float* __restrict__ att, const float* __restrict__ att_log,
                                                                const float* __restrict__ d_att, const float* __restrict__ d_attlog,
                                                                float* __restrict__ d_raw, long rows, int N) {
    const long row = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const bool ok = lane < N;
    const float a = ok ? att[row * N + lane] : 0.f, da = (ok && d_att) ? d_att[row * N + lane] : 0.f;
    const float al = ok ? att_log[row * N + lane] : 0.f, dl = (ok && d_attlog) ? d_attlog[row * N + lane] : 0.f;
    const float s1 = wave_sum(a * da), s2 = wave_sum(dl);
    if (ok) d_raw[row * N + lane] = a * (da - s1) + dl - __expf(al) * s2;
}

// ------------------------------------------------------------------------------------------ fp32 LayerNorm (cols % 64 == 0, <= 1024)
__global__ __launch_bounds__(256) void ln32_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ y, float* __restrict__ mean,
                                                       float* __restrict__ rstd, long rows, int cols, float eps) {
    const long row = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int per = cols / 64;
    float v[16];
    float s = 0.f;
    for (int j = 0; j < per; ++j) { v[j] = x[row * cols + j * 64 + lane]; s += v[j]; }
    const float mu = wave_sum(s) / cols;
    float q = 0.f;
    for (int j = 0; j < per; ++j) { const float d = v[j] - mu; q += d * d; }
    const float rs = rsqrtf(wave_sum(q) / cols + eps);
    for (int j = 0; j < per; ++j) {
        const int c = j * 64 + lane;
        y[row * cols + c] = (v[j] - mu) * rs * gamma[c] + beta[c];
    }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}
__global__ __launch_bounds__(256) void ln32_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma, float* __restrict__ dx, float* __restrict__ dgamma,
                                                       float* __restrict__ dbeta, long rows, int cols) {
    const long row = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int per = cols / 64;
    const float mu = mean[row], rs = rstd[row];
    float xh[16], gd[16];
    float s1 = 0.f, s2 = 0.f;
    for (int j = 0; j < per; ++j) {
        const int c = j * 64 + lane;
        const float d = dy[row * cols + c];
        xh[j] = (x[row * cols + c] - mu) * rs;
        gd[j] = d * gamma[c];
        s1 += gd[j];
        s2 += gd[j] * xh[j];
        atomicAdd(&dgamma[c], d * xh[j]);
        atomicAdd(&dbeta[c], d);
    }
    s1 = wave_sum(s1) / cols;
    s2 = wave_sum(s2) / cols;
    for (int j = 0; j < per; ++j) dx[row * cols + j * 64 + lane] = rs * (gd[j] - s1 - xh[j] * s2);
}

// ------------------------------------------------------------------------------------------ BiLSTM (hidden 256)
// grid (B, 2 directions), 256 threads = hidden units.  xproj [B*Tp][2][4*HID] holds x W_ih^T + b_ih + b_hh;
// whhT [2][HID][4*HID] (transposed: coalesced over the gate column).  Packed-sequence semantics: only t < len[b].
constexpr int HID = 256;
__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(HID) void lstm_fwd_kernel(const float* __restrict__ xproj, const float* __restrict__ whhT,
                                                       const int* __restrict__ lens, float* __restrict__ hout, float* __restrict__ gates,
                                                       float* __restrict__ cstate, int Tp, int T) {
    __shared__ float h[HID];
    const int b = blockIdx.x, dir = blockIdx.y, j = threadIdx.x;
    int len = lens[b];
    len = len < T ? len : T;
    const float* W = whhT + (long)dir * HID * 4 * HID;
    float c = 0.f;
    h[j] = 0.f;
    __syncthreads();
    for (int step = 0; step < len; ++step) {
        const int t = dir ? len - 1 - step : step;
        const long row = (long)b * Tp + t;
        const float* xp = xproj + (row * 2 + dir) * 4 * HID;
        float pi = xp[j], pf = xp[HID + j], pg = xp[2 * HID + j], po = xp[3 * HID + j];
#pragma unroll 4
        for (int k = 0; k < HID; ++k) {
            const float hk = h[k];
            const float* w = W + (long)k * 4 * HID;
            pi = fmaf(hk, w[j], pi);
            pf = fmaf(hk, w[HID + j], pf);
            pg = fmaf(hk, w[2 * HID + j], pg);
            po = fmaf(hk, w[3 * HID + j], po);
        }
        const float gi = sigm(pi), gf = sigm(pf), gg = tanhf(pg), go = sigm(po);
        c = gf * c + gi * gg;
        const float hn = go * tanhf(c);
        __syncthreads();
        h[j] = hn;
        __syncthreads();
        hout[row * 2 * HID + dir * HID + j] = hn;
        if (gates) {
            float* gp = gates + (row * 2 + dir) * 4 * HID;
            gp[j] = gi; gp[HID + j] = gf; gp[2 * HID + j] = gg; gp[3 * HID + j] = go;
            cstate[(row * 2 + dir) * HID + j] = c;
        }
    }
    // frames beyond the utterance: zeros (pad_packed_sequence)
    for (int t = len; t < Tp; ++t) hout[((long)b * Tp + t) * 2 * HID + dir * HID + j] = 0.f;
}

// backward through time: dgates (pre-activation grads) [B*Tp][2][4*HID]; whh [2][4*HID][HID] (row = gate column)
__global__ __launch_bounds__(HID) void lstm_bwd_kernel(const float* __restrict__ dhout, const float* __restrict__ whh,
                                                       const int* __restrict__ lens, const float* __restrict__ gates,
                                                       const float* __restrict__ cstate, float* __restrict__ dgates, int Tp, int T) {
    __shared__ float dg[4 * HID];
    const int b = blockIdx.x, dir = blockIdx.y, j = threadIdx.x;
    int len = lens[b];
    len = len < T ? len : T;
    const float* W = whh + (long)dir * 4 * HID * HID;
    float dh_rec = 0.f, dc = 0.f;
    for (int step = len - 1; step >= 0; --step) {
        const int t = dir ? len - 1 - step : step;                  // reverse of the forward visiting order
        const long row = (long)b * Tp + t;
        const float* gp = gates + (row * 2 + dir) * 4 * HID;
        const float gi = gp[j], gf = gp[HID + j], gg = gp[2 * HID + j], go = gp[3 * HID + j];
        const float c = cstate[(row * 2 + dir) * HID + j];
        float cprev = 0.f;
        if (step > 0) {
            const int tp = dir ? t + 1 : t - 1;
            cprev = cstate[(((long)b * Tp + tp) * 2 + dir) * HID + j];
        }
        const float dh = dhout[row * 2 * HID + dir * HID + j] + dh_rec;
        const float tc = tanhf(c);
        const float d_o = dh * tc;
        dc += dh * go * (1.f - tc * tc);
        const float d_i = dc * gg, d_g = dc * gi, d_f = dc * cprev;
        const float pi = d_i * gi * (1.f - gi), pf = d_f * gf * (1.f - gf), pg = d_g * (1.f - gg * gg), po = d_o * go * (1.f - go);
        dc = dc * gf;
        float* dgp = dgates + (row * 2 + dir) * 4 * HID;
        dgp[j] = pi; dgp[HID + j] = pf; dgp[2 * HID + j] = pg; dgp[3 * HID + j] = po;
        __syncthreads();
        dg[j] = pi; dg[HID + j] = pf; dg[2 * HID + j] = pg; dg[3 * HID + j] = po;
        __syncthreads();
        float acc = 0.f;
#pragma unroll 4
        for (int col = 0; col < 4 * HID; ++col) acc = fmaf(dg[col], W[(long)col * HID + j], acc);
        dh_rec = acc;
    }
    for (int t = len; t < Tp; ++t) {
        float* dgp = dgates + (((long)b * Tp + t) * 2 + dir) * 4 * HID;
        dgp[j] = 0.f; dgp[HID + j] = 0.f; dgp[2 * HID + j] = 0.f; dgp[3 * HID + j] = 0.f;
    }
}

// out[b][t] = table[b][idx[b][t]]  (int64), rows beyond len -> -1
__global__ void gather_rows_kernel(const int* __restrict__ table, const int64_t* __restrict__ idx, const int* __restrict__ lens,
                                   int64_t* __restrict__ out, int B, int T, int N) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * T) return;
    const int b = (int)(i / T), t = (int)(i % T);
    out[i] = t < lens[b] ? (int64_t)table[b * N + (int)idx[i]] : -1;
}

// elementwise fp32 helpers: y = tanh(drop(x)); dy -> dx
__global__ void tanh_drop_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n, float sc, uint32_t s0, uint32_t s1,
                                     uint32_t thr) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float v = x[i];
        if (thr) v = drop_keep((uint64_t)i, s0, s1, thr) ? v * sc : 0.f;
        y[i] = tanhf(v);
    }
}
__global__ void tanh_drop_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx, long n, float sc,
                                     uint32_t s0, uint32_t s1, uint32_t thr) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float g = dy[i] * (1.f - y[i] * y[i]);
        if (thr) g = drop_keep((uint64_t)i, s0, s1, thr) ? g * sc : 0.f;
        dx[i] = g;
    }
}
__global__ void drop32_kernel(const float* __restrict__ x, float* __restrict__ y, long n, float sc, uint32_t s0, uint32_t s1, uint32_t thr) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        y[i] = drop_keep((uint64_t)i, s0, s1, thr) ? x[i] * sc : 0.f;
}
__global__ void colsum32_kernel(const float* __restrict__ x, long ld, float* __restrict__ out, long rows, int N) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    float s = 0.f;
    for (long r = 0; r < rows; ++r) s += x[r * ld + c];
    out[c] = s;
}

inline unsigned gridn(long n, int block = 256, int maxb = 2048) {
    long b = ceil_div(n, block);
    return (unsigned)(b < 1 ? 1 : (b > maxb ? maxb : b));
}

}  // namespace

extern "C" int aptai_sgemm_f32(const void* A, int a_bf16, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn,
                               float* C, int64_t ldc, const float* bias, float alpha, int accumulate, int64_t M, int64_t N,
                               int64_t K, int64_t batch, int64_t bsa, int64_t bsb, int64_t bsc, void* stream) {
    APTAI_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && batch > 0, "aptai_sgemm_f32: bad arguments");
    SgemmArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.sam = sam; g.sak = sak; g.sbk = sbk; g.sbn = sbn; g.ldc = ldc;
    g.bsa = bsa; g.bsb = bsb; g.bsc = bsc; g.M = (int)M; g.N = (int)N; g.K = (int)K; g.a_bf16 = a_bf16; g.accumulate = accumulate;
    g.alpha = alpha;
    APTAI_LAUNCH(sgemm_kernel, dim3((unsigned)ceil_div(N, 64), (unsigned)ceil_div(M, 64), (unsigned)batch), dim3(256), 0,
                 (hipStream_t)stream, g);
    APTAI_CHECK_LAUNCH("sgemm_kernel");
    return APTAI_OK;
}

extern "C" int aptai_embed_pe_fwd(const int32_t* ids, const float* emb, const float* pe, float* out, int64_t rows, int64_t N,
                                  int64_t D, float dropout_p, uint64_t seed, void* stream) {
    APTAI_REQUIRE(ids && emb && pe && out && rows > 0, "aptai_embed_pe_fwd: bad arguments");
    const uint32_t thr = drop_thr16(dropout_p);
    APTAI_LAUNCH(embed_pe_fwd_kernel, dim3(gridn(rows * D)), dim3(256), 0, (hipStream_t)stream, ids, emb, pe, out, (int)rows, (int)N,
                 (int)D, drop_scale(thr), (uint32_t)seed, (uint32_t)(seed >> 32), thr);
    APTAI_CHECK_LAUNCH("embed_pe_fwd_kernel");
    return APTAI_OK;
}
extern "C" int aptai_embed_bwd(const int32_t* ids, const float* dout, float* demb_zeroed, int64_t rows, int64_t D, float dropout_p,
                               uint64_t seed, void* stream) {
    APTAI_REQUIRE(ids && dout && demb_zeroed && rows > 0, "aptai_embed_bwd: bad arguments");
    const uint32_t thr = drop_thr16(dropout_p);
    APTAI_LAUNCH(embed_bwd_kernel, dim3(gridn(rows * D)), dim3(256), 0, (hipStream_t)stream, ids, dout, demb_zeroed, (int)rows, (int)D,
                 drop_scale(thr), (uint32_t)seed, (uint32_t)(seed >> 32), thr);
    APTAI_CHECK_LAUNCH("embed_bwd_kernel");
    return APTAI_OK;
}

extern "C" int aptai_xattn_softmax_fwd(const float* raw, const int32_t* phn_ids, float* energy, float* att, float* att_log,
                                       int64_t* align, int64_t B, int64_t T, int64_t N, void* stream) {
    APTAI_REQUIRE(raw && phn_ids && energy && att && att_log && N > 0 && N <= 64, "aptai_xattn_softmax_fwd: bad arguments (N <= 64)");
    APTAI_LAUNCH(xattn_softmax_fwd_kernel, dim3((unsigned)ceil_div(B * T * 64, 256)), dim3(256), 0, (hipStream_t)stream, raw, phn_ids,
                 energy, att, att_log, align, (int)B, (int)T, (int)N);
    APTAI_CHECK_LAUNCH("xattn_softmax_fwd_kernel");
    return APTAI_OK;
}
extern "C" int aptai_xattn_softmax_bwd(const float* att, const float* att_log, const float* d_att, const float* d_attlog, float* d_raw,
                                       int64_t rows, int64_t N, void* stream) {
    APTAI_REQUIRE(att && att_log && d_raw && N > 0 && N <= 64, "aptai_xattn_softmax_bwd: bad arguments");
    APTAI_LAUNCH(xattn_softmax_bwd_kernel, dim3((unsigned)ceil_div(rows * 64, 256)), dim3(256), 0, (hipStream_t)stream, att, att_log,
                 d_att, d_attlog, d_raw, (long)rows, (int)N);
    APTAI_CHECK_LAUNCH("xattn_softmax_bwd_kernel");
    return APTAI_OK;
}

extern "C" int aptai_layernorm_f32_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                                       int64_t rows, int64_t cols, float eps, void* stream) {
    APTAI_REQUIRE(x && gamma && beta && y && mean && rstd && cols % 64 == 0 && cols <= 1024, "aptai_layernorm_f32_fwd: bad arguments");
    APTAI_LAUNCH(ln32_fwd_kernel, dim3((unsigned)ceil_div(rows * 64, 256)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, mean,
                 rstd, (long)rows, (int)cols, eps);
    APTAI_CHECK_LAUNCH("ln32_fwd_kernel");
    return APTAI_OK;
}
extern "C" int aptai_layernorm_f32_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                                       float* dx, float* dgamma_zeroed, float* dbeta_zeroed, int64_t rows, int64_t cols, void* stream) {
    APTAI_REQUIRE(dy && x && mean && rstd && gamma && dx && dgamma_zeroed && dbeta_zeroed && cols % 64 == 0 && cols <= 1024,
                  "aptai_layernorm_f32_bwd: bad arguments");
    APTAI_LAUNCH(ln32_bwd_kernel, dim3((unsigned)ceil_div(rows * 64, 256)), dim3(256), 0, (hipStream_t)stream, dy, x, mean, rstd, gamma,
                 dx, dgamma_zeroed, dbeta_zeroed, (long)rows, (int)cols);
    APTAI_CHECK_LAUNCH("ln32_bwd_kernel");
    return APTAI_OK;
}

extern "C" int aptai_lstm_fwd(const float* xproj, const float* whhT, const int32_t* lens, float* hout, float* gates, float* cstate,
                              int64_t B, int64_t Tp, int64_t T, int64_t hidden, void* stream) {
    APTAI_REQUIRE(xproj && whhT && lens && hout, "aptai_lstm_fwd: null pointer");
    APTAI_REQUIRE(hidden == HID, "aptai_lstm_fwd: built for hidden size 256");
    APTAI_REQUIRE((gates == nullptr) == (cstate == nullptr), "aptai_lstm_fwd: gates and cstate go together");
    APTAI_LAUNCH(lstm_fwd_kernel, dim3((unsigned)B, 2), dim3(HID), 0, (hipStream_t)stream, xproj, whhT, lens, hout, gates, cstate,
                 (int)Tp, (int)T);
    APTAI_CHECK_LAUNCH("lstm_fwd_kernel");
    return APTAI_OK;
}
extern "C" int aptai_lstm_bwd(const float* dhout, const float* whh, const int32_t* lens, const float* gates, const float* cstate,
                              float* dgates, int64_t B, int64_t Tp, int64_t T, int64_t hidden, void* stream) {
    APTAI_REQUIRE(dhout && whh && lens && gates && cstate && dgates, "aptai_lstm_bwd: null pointer");
    APTAI_REQUIRE(hidden == HID, "aptai_lstm_bwd: built for hidden size 256");
    APTAI_LAUNCH(lstm_bwd_kernel, dim3((unsigned)B, 2), dim3(HID), 0, (hipStream_t)stream, dhout, whh, lens, gates, cstate, dgates,
                 (int)Tp, (int)T);
    APTAI_CHECK_LAUNCH("lstm_bwd_kernel");
    return APTAI_OK;
}

extern "C" int aptai_gather_alignment(const int32_t* phn_table, const int64_t* align, const int32_t* lens, int64_t* out, int64_t B,
                                      int64_t T, int64_t N, void* stream) {
    APTAI_REQUIRE(phn_table && align && lens && out, "aptai_gather_alignment: null pointer");
    APTAI_LAUNCH(gather_rows_kernel, dim3((unsigned)ceil_div(B * T, 256)), dim3(256), 0, (hipStream_t)stream, phn_table, align, lens, out,
                 (int)B, (int)T, (int)N);
    APTAI_CHECK_LAUNCH("gather_rows_kernel");
    return APTAI_OK;
}

extern "C" int aptai_tanh_dropout_f32(const float* x, const float* y_or_null, const float* dy_or_null, float* out, int64_t n,
                                      float dropout_p, uint64_t seed, void* stream) {
    // forward: out = tanh(dropout(x));  backward (dy given): out = dropout'(dy * (1 - y^2)) with y = saved forward output
    APTAI_REQUIRE(out && n > 0, "aptai_tanh_dropout_f32: bad arguments");
    const uint32_t thr = drop_thr16(dropout_p);
    if (dy_or_null) {
        APTAI_REQUIRE(y_or_null, "aptai_tanh_dropout_f32: backward needs the saved output");
        APTAI_LAUNCH(tanh_drop_bwd_kernel, dim3(gridn(n)), dim3(256), 0, (hipStream_t)stream, y_or_null, dy_or_null, out, (long)n,
                     drop_scale(thr), (uint32_t)seed, (uint32_t)(seed >> 32), thr);
    } else {
        APTAI_REQUIRE(x, "aptai_tanh_dropout_f32: forward needs x");
        APTAI_LAUNCH(tanh_drop_fwd_kernel, dim3(gridn(n)), dim3(256), 0, (hipStream_t)stream, x, out, (long)n, drop_scale(thr),
                     (uint32_t)seed, (uint32_t)(seed >> 32), thr);
    }
    APTAI_CHECK_LAUNCH("tanh_dropout kernel");
    return APTAI_OK;
}
extern "C" int aptai_dropout_f32(const float* x, float* y, int64_t n, float dropout_p, uint64_t seed, void* stream) {
    APTAI_REQUIRE(x && y && n > 0 && dropout_p > 0.f, "aptai_dropout_f32: bad arguments");
    const uint32_t thr = drop_thr16(dropout_p);
    APTAI_LAUNCH(drop32_kernel, dim3(gridn(n)), dim3(256), 0, (hipStream_t)stream, x, y, (long)n, drop_scale(thr), (uint32_t)seed,
                 (uint32_t)(seed >> 32), thr);
    APTAI_CHECK_LAUNCH("drop32_kernel");
    return APTAI_OK;
}
extern "C" int aptai_colsum_f32(const float* x, int64_t ld, float* out, int64_t rows, int64_t N, void* stream) {
    APTAI_REQUIRE(x && out && rows > 0 && N > 0, "aptai_colsum_f32: bad arguments");
    APTAI_LAUNCH(colsum32_kernel, dim3((unsigned)ceil_div(N, 64)), dim3(64), 0, (hipStream_t)stream, x, (long)ld, out, (long)rows, (int)N);
    APTAI_CHECK_LAUNCH("colsum32_kernel");
    return APTAI_OK;
}
