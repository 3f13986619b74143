// Force_APTAI aligner heads in fp32 (gfx950): everything after the frozen wav2vec2 encoder in
// models/force_aptai.py:108-161 — Embedding + sinusoidal PE (:118-119), frame Linear (:122), CrossAttention
// (models/modules.py:139-153), log-softmax alignment + argmax read-out (:128-130,148-161), BiLSTM + MLP
// (models/modules.py:195-214).  These layers are small (128/256 wide) and feed an ARGMAX whose indices must match the
// reference, so they run in fp32 (VALU FMA), not bf16 MFMA.  The forward-sum loss reuses the CTC kernels (ctc.hip).
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------ generic fp32 GEMM on the matrix cores
// C[m][n] (+)= alpha * sum_k A(m,k) * B(k,n) + bias[n];  A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn].
// v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, bit-for-bit a k-ordered fmaf chain (exact fp32, guide "FP32-input MFMA"), at
// the fp32 vector peak but with ONE operand register per lane and no LDS traffic per FMA.  64 x 64 x 32 tile, 4 waves, each a
// 32 x 32 sub-tile = 16 MFMAs per K-tile.  Operands go through LDS as [k][m] / [k][n] (pitch 65: conflict-free scatter from
// either source orientation, conflict-free 32-lane row reads).  16-byte global loads whenever the contiguous axis allows it,
// scalar guarded loads on ragged tiles.  A may be bf16 (the encoder's hidden states).  split_k > 1: K is cut into slabs whose
// partial tiles go to a workspace and are summed in slab order by sgemm_reduce_kernel (deterministic).
constexpr int SG_BK = 32, SG_P = 65;
typedef __attribute__((ext_vector_type(4))) float sg_f4;
struct SgemmArgs {
    const void* A; const float* B; float* C; const float* bias; float* ws;
    long sam, sak, sbk, sbn, ldc;
    long bsa, bsb, bsc;           // batch strides (elements)
    long kchunk;
    int M, N, K, a_bf16, accumulate, split_k, vec_a, vec_b;
    float alpha;
};

__global__ __launch_bounds__(256) void sgemm_mfma_kernel(SgemmArgs g) {
    __shared__ float As[SG_BK * SG_P];
    __shared__ float Bs[SG_BK * SG_P];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int bz = blockIdx.z / g.split_k, sz = blockIdx.z % g.split_k;
    const long kbeg = (long)sz * g.kchunk;
    const long kend = (kbeg + g.kchunk < g.K) ? kbeg + g.kchunk : g.K;
    const float* Af = (const float*)g.A + (long)bz * g.bsa;
    const bf16_t* Ab = (const bf16_t*)g.A + (long)bz * g.bsa;
    const float* Bp = g.B + (long)bz * g.bsb;
    f32x16 acc = (f32x16)(0.f);
    const bool m_full = m0 + 64 <= g.M, n_full = n0 + 64 <= g.N;
    // staging goes through registers: the loads of K-tile t+1 are issued before the MFMAs of tile t and land under them
    // (8 values per thread and operand).  pa / pb = how this tile was fetched: 0 scalar guarded, 1 16-byte loads along k,
    // 2 16-byte loads along m / n, 3 (A only) 8 bf16 along k.
    float ra[8], rb[8];
    int pa = 0, pb = 0;
    auto fetch = [&](long k0) {
        const bool k_full = k0 + SG_BK <= kend;
        pa = 0;
        if (g.vec_a && m_full && k_full) pa = g.sak == 1 ? (g.a_bf16 ? 3 : 1) : ((g.sam == 1 && !g.a_bf16) ? 2 : 0);
        if (pa == 3) {
            const int row = tid >> 2, k8 = (tid & 3) * 8;
            const u32x4 v = *(const u32x4*)(Ab + (long)(m0 + row) * g.sam + k0 + k8);
#pragma unroll
            for (int j = 0; j < 4; ++j) { ra[2 * j] = lo_bf(v[j]); ra[2 * j + 1] = hi_bf(v[j]); }
        } else if (pa == 1) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int idx = e * 256 + tid, row = idx >> 3, k4 = (idx & 7) * 4;
                const sg_f4 v = *(const sg_f4*)(Af + (long)(m0 + row) * g.sam + k0 + k4);
                ra[4 * e] = v[0]; ra[4 * e + 1] = v[1]; ra[4 * e + 2] = v[2]; ra[4 * e + 3] = v[3];
            }
        } else if (pa == 2) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int idx = e * 256 + tid, kk = idx >> 4, m4 = (idx & 15) * 4;
                const sg_f4 v = *(const sg_f4*)(Af + (k0 + kk) * g.sak + m0 + m4);
                ra[4 * e] = v[0]; ra[4 * e + 1] = v[1]; ra[4 * e + 2] = v[2]; ra[4 * e + 3] = v[3];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int idx = e * 256 + tid;
                int mm, kk;
                if (g.sak == 1) { kk = idx & 31; mm = idx >> 5; } else { mm = idx & 63; kk = idx >> 6; }
                const int m = m0 + mm;
                const long k = k0 + kk;
                float v = 0.f;
                if (m < g.M && k < kend) {
                    const long off = (long)m * g.sam + k * g.sak;
                    v = g.a_bf16 ? bf2f(Ab[off]) : Af[off];
                }
                ra[e] = v;
            }
        }
        pb = 0;
        if (g.vec_b && n_full && k_full) pb = g.sbn == 1 ? 2 : (g.sbk == 1 ? 1 : 0);
        if (pb == 2) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int idx = e * 256 + tid, kk = idx >> 4, n4 = (idx & 15) * 4;
                const sg_f4 v = *(const sg_f4*)(Bp + (k0 + kk) * g.sbk + n0 + n4);
                rb[4 * e] = v[0]; rb[4 * e + 1] = v[1]; rb[4 * e + 2] = v[2]; rb[4 * e + 3] = v[3];
            }
        } else if (pb == 1) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int idx = e * 256 + tid, col = idx >> 3, k4 = (idx & 7) * 4;
                const sg_f4 v = *(const sg_f4*)(Bp + (long)(n0 + col) * g.sbn + k0 + k4);
                rb[4 * e] = v[0]; rb[4 * e + 1] = v[1]; rb[4 * e + 2] = v[2]; rb[4 * e + 3] = v[3];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int idx = e * 256 + tid;
                int nn, kk;
                if (g.sbn == 1) { nn = idx & 63; kk = idx >> 6; } else { kk = idx & 31; nn = idx >> 5; }
                const int n = n0 + nn;
                const long k = k0 + kk;
                float v = 0.f;
                if (n < g.N && k < kend) v = Bp[k * g.sbk + (long)n * g.sbn];
                rb[e] = v;
            }
        }
    };
    auto stage = [&]() {                                        // registers -> As[k][m], Bs[k][n]
        if (pa == 3) {
            const int row = tid >> 2, k8 = (tid & 3) * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) As[(k8 + j) * SG_P + row] = ra[j];
        } else if (pa == 1) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int idx = e * 256 + tid, row = idx >> 3, k4 = (idx & 7) * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) As[(k4 + j) * SG_P + row] = ra[4 * e + j];
            }
        } else if (pa == 2) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int idx = e * 256 + tid, kk = idx >> 4, m4 = (idx & 15) * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) As[kk * SG_P + m4 + j] = ra[4 * e + j];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int idx = e * 256 + tid;
                int mm, kk;
                if (g.sak == 1) { kk = idx & 31; mm = idx >> 5; } else { mm = idx & 63; kk = idx >> 6; }
                As[kk * SG_P + mm] = ra[e];
            }
        }
        if (pb == 2) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int idx = e * 256 + tid, kk = idx >> 4, n4 = (idx & 15) * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) Bs[kk * SG_P + n4 + j] = rb[4 * e + j];
            }
        } else if (pb == 1) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int idx = e * 256 + tid, col = idx >> 3, k4 = (idx & 7) * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) Bs[(k4 + j) * SG_P + col] = rb[4 * e + j];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int idx = e * 256 + tid;
                int nn, kk;
                if (g.sbn == 1) { nn = idx & 63; kk = idx >> 6; } else { kk = idx & 31; nn = idx >> 5; }
                Bs[kk * SG_P + nn] = rb[e];
            }
        }
    };
    if (kbeg < kend) fetch(kbeg);
    for (long k0 = kbeg; k0 < kend; k0 += SG_BK) {
        stage();
        __syncthreads();
        if (k0 + SG_BK < kend) fetch(k0 + SG_BK);
        const float* ap = As + (lane >> 5) * SG_P + wm * 32 + (lane & 31);
        const float* bp = Bs + (lane >> 5) * SG_P + wn * 32 + (lane & 31);
#pragma unroll
        for (int kk = 0; kk < SG_BK / 2; ++kk)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * kk * SG_P], bp[2 * kk * SG_P], acc, 0, 0, 0);
        __syncthreads();
    }
    const int n = n0 + wn * 32 + (lane & 31);
    if (n >= g.N) return;
    const float bias = (g.bias && g.split_k == 1) ? g.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m >= g.M) continue;
        if (g.split_k > 1) {
            g.ws[((long)blockIdx.z * g.M + m) * g.N + n] = acc[r];
        } else {
            float* c = g.C + (long)bz * g.bsc + (long)m * g.ldc + n;
            const float v = acc[r] * g.alpha + bias;
            *c = g.accumulate ? *c + v : v;
        }
    }
}

__global__ __launch_bounds__(256) void sgemm_reduce_kernel(SgemmArgs g, int batch) {
    const long total = (long)batch * g.M * g.N;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i % g.N);
        const long mb = i / g.N;
        const int m = (int)(mb % g.M), bz = (int)(mb / g.M);
        float s = 0.f;
        for (int z = 0; z < g.split_k; ++z) s += g.ws[(((long)bz * g.split_k + z) * g.M + m) * g.N + n];
        float* c = g.C + (long)bz * g.bsc + (long)m * g.ldc + n;
        const float v = s * g.alpha + (g.bias ? g.bias[n] : 0.f);
        *c = g.accumulate ? *c + v : v;
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
// ORDER-FIXED (round 4; was one float atomicAdd per element: the sum over the rows that share a phoneme id had no defined order, so
// phn_emb_layer.weight.grad changed in its last bits from launch to launch).  One block per row; the block of the FIRST row that
// carries an id owns that id's gradient row, collects which later rows carry the same id as ballot masks (row order), then every
// thread adds its column of those rows in mask order, four independent loads at a time (a fixed order: same inputs, same bits).
__global__ __launch_bounds__(128) void embed_bwd_kernel(const int* __restrict__ ids, const float* __restrict__ dout, float* __restrict__ demb,
                                                        int rows, int D, float scale_keep, uint32_t s0, uint32_t s1, uint32_t thr) {
    __shared__ unsigned long long masks[128];                       // rows r0 + 64 m .. + 64 : up to 8192 rows behind r0
    __shared__ int list[8192];                                      // the rows that carry this id, ascending
    __shared__ int nlist;
    const int r0 = blockIdx.x;
    const int id = ids[r0];
    if (id == 0) return;                                            // padding_idx = 0 receives no gradient (block-uniform)
    int dup = 0;
    for (int i = threadIdx.x; i < r0; i += blockDim.x) dup |= ids[i] == id;
    if (__syncthreads_or(dup)) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int nmask = (rows - r0 + 63) >> 6;
    nmask = nmask < 128 ? nmask : 128;
    for (int m = wave; m < nmask; m += 2) {
        const int r = r0 + m * 64 + lane;
        const unsigned long long bal = __ballot(r < rows && ids[r] == id);
        if (lane == 0) masks[m] = bal;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int n = 0;
        for (int m = 0; m < nmask; ++m) {
            unsigned long long bits = masks[m];
            while (bits) {
                list[n++] = r0 + m * 64 + __builtin_ctzll(bits);
                bits &= bits - 1;
            }
        }
        nlist = n;
    }
    __syncthreads();
    const int n = nlist;
    auto value = [&](int k, int d) {
        if (k >= n) return 0.f;
        const long e = (long)list[k] * D + d;
        float v = dout[e];
        if (thr) v = drop_keep((uint64_t)e, s0, s1, thr) ? v * scale_keep : 0.f;
        return v;
    };
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float acc = 0.f;
        for (int k = 0; k < n; k += 4) {                            // four independent loads, added in list order
            const float v0 = value(k, d), v1 = value(k + 1, d), v2 = value(k + 2, d), v3 = value(k + 3, d);
            acc = (((acc + v0) + v1) + v2) + v3;
        }
        demb[(long)id * D + d] = acc;
    }
}

// ------------------------------------------------------------------------------------------ cross-attention softmaxes
// one wave per (b,t) row, lane = phoneme slot (N <= 64).  raw -> energy = raw + mask1; att = softmax(energy);
// att_log = log_softmax(energy + mask1) (the reference adds the -1000 mask twice); align = argmax(att_log) (first max)
__global__ __launch_bounds__(256) void xattn_softmax_fwd_kernel(const float* __restrict__ raw, const int* __restrict__ ids,
                                                                float* __restrict__ energy, float* __restrict__ att,
                                                                float* __restrict__ att_log, int64_t* __restrict__ align,
                                                                float* __restrict__ fs_rows, int B, int T, int N) {
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
    // forward-sum input row (models/modules.py:90-98): [blank log-prob -1 | att_log | zero padding] in a 64-float row
    if (fs_rows) {
        const float up = __shfl_up(al, 1, 64);
        fs_rows[row * 64 + lane] = lane == 0 ? -1.f : (lane <= N ? up : 0.f);
    }
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
                                                                long ld_dattlog, float* __restrict__ d_raw, long rows, int N) {
    const long row = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const bool ok = lane < N;
    const float a = ok ? att[row * N + lane] : 0.f, da = (ok && d_att) ? d_att[row * N + lane] : 0.f;
    const float al = ok ? att_log[row * N + lane] : 0.f, dl = (ok && d_attlog) ? d_attlog[row * ld_dattlog + lane] : 0.f;
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
// dgamma / dbeta: per-lane register partials over the rows a wave walks, one [2][cols] slab per block (no atomics: the 2 M
// atomicAdds of the first version took 343 us), summed in block order by ln32_bwd_final_kernel
constexpr int LN32_BLOCKS = 256;
__global__ __launch_bounds__(256) void ln32_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma, float* __restrict__ dx, float* __restrict__ ws,
                                                       long rows, int cols) {
    __shared__ float red[4][2][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per = cols / 64;
    float ag[16], ab[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { ag[j] = 0.f; ab[j] = 0.f; }
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
        const float mu = mean[row], rs = rstd[row];
        float xh[16], gd[16];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (j >= per) break;
            const int c = j * 64 + lane;
            const float d = dy[row * cols + c];
            xh[j] = (x[row * cols + c] - mu) * rs;
            gd[j] = d * gamma[c];
            s1 += gd[j];
            s2 += gd[j] * xh[j];
            ag[j] += d * xh[j];
            ab[j] += d;
        }
        s1 = wave_sum(s1) / cols;
        s2 = wave_sum(s2) / cols;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (j >= per) break;
            dx[row * cols + j * 64 + lane] = rs * (gd[j] - s1 - xh[j] * s2);
        }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (j >= per) break;
        red[wave][0][j * 64 + lane] = ag[j];
        red[wave][1][j * 64 + lane] = ab[j];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * cols; i += 256) {
        const int which = i / cols, c = i % cols;
        ws[((long)blockIdx.x * 2 + which) * cols + c] = (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]);
    }
}
__global__ void ln32_bwd_final_kernel(const float* __restrict__ ws, float* __restrict__ dgamma, float* __restrict__ dbeta, int blocks,
                                      int cols) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * cols) return;
    const int which = i / cols, c = i % cols;
    float s = 0.f;
    for (int b = 0; b < blocks; ++b) s += ws[((long)b * 2 + which) * cols + c];
    (which ? dbeta : dgamma)[c] = s;
}

// ------------------------------------------------------------------------------------------ BiLSTM (hidden 256), serial form
// The first build's kernels: one block per (utterance, direction) that re-streams W_hh from L2 on every frame.  Kept as the
// on-device cross-check of the cooperating-workgroup kernels in lstm.hip (tests/test_gpu_force.py); not on the hot path.
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
// column sums of an fp32 matrix in two deterministic stages: [chunks][N] partials (lane = column, the 4 waves of a block
// interleave rows, coalesced 256-byte row segments), then one thread per column adds the chunks in order
constexpr int CS_CHUNKS = 64;
__global__ __launch_bounds__(256) void colsum32_part_kernel(const float* __restrict__ x, long ld, float* __restrict__ ws, long rows, int N,
                                                            long rows_per_chunk) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const long r0 = (long)blockIdx.y * rows_per_chunk;
    const long r1 = (r0 + rows_per_chunk < rows) ? r0 + rows_per_chunk : rows;
    float s0 = 0.f, s1 = 0.f;
    if (c < N) {
        long r = r0 + wave;
        for (; r + 4 < r1; r += 8) { s0 += x[r * ld + c]; s1 += x[(r + 4) * ld + c]; }
        for (; r < r1; r += 4) s0 += x[r * ld + c];
    }
    part[wave][lane] = s0 + s1;
    __syncthreads();
    if (wave == 0 && c < N) ws[(long)blockIdx.y * N + c] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}
__global__ void colsum32_final_kernel(const float* __restrict__ ws, float* __restrict__ out, int chunks, int N) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    float s = 0.f;
    for (int k = 0; k < chunks; ++k) s += ws[(long)k * N + c];
    out[c] = s;
}

inline unsigned gridn(long n, int block = 256, int maxb = 2048) {
    long b = ceil_div(n, block);
    return (unsigned)(b < 1 ? 1 : (b > maxb ? maxb : b));
}

}  // namespace

extern "C" int64_t aptai_sgemm_workspace_bytes(int64_t M, int64_t N, int64_t batch, int64_t split_k) {
    return split_k > 1 ? batch * split_k * M * N * 4 : 0;
}

extern "C" int aptai_sgemm_f32(const void* A, int a_bf16, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn,
                               float* C, int64_t ldc, const float* bias, float alpha, int accumulate, int64_t M, int64_t N,
                               int64_t K, int64_t batch, int64_t bsa, int64_t bsb, int64_t bsc, int64_t split_k, float* workspace,
                               void* stream) {
    APTAI_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && batch > 0, "aptai_sgemm_f32: bad arguments");
    APTAI_REQUIRE(split_k >= 1 && (split_k == 1 || workspace != nullptr), "aptai_sgemm_f32: split_k > 1 needs a workspace");
    SgemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.sam = sam; g.sak = sak; g.sbk = sbk; g.sbn = sbn; g.ldc = ldc;
    g.bsa = bsa; g.bsb = bsb; g.bsc = bsc; g.M = (int)M; g.N = (int)N; g.K = (int)K; g.a_bf16 = a_bf16; g.accumulate = accumulate;
    g.alpha = alpha;
    g.ws = workspace;
    long kchunk = (long)ceil_div(ceil_div(K, split_k), SG_BK) * SG_BK;      // slabs start on K-tile boundaries
    g.split_k = (int)ceil_div(K, kchunk);
    g.kchunk = kchunk;
    const long ea = a_bf16 ? 8 : 4;                                         // elements per 16-byte load
    const long esz = a_bf16 ? 2 : 4;
    g.vec_a = ((uintptr_t)A % 16 == 0) && (bsa % ea == 0) &&
              ((sak == 1 && sam % ea == 0) || (sam == 1 && !a_bf16 && sak % 4 == 0));
    (void)esz;
    g.vec_b = ((uintptr_t)B % 16 == 0) && (bsb % 4 == 0) && ((sbn == 1 && sbk % 4 == 0) || (sbk == 1 && sbn % 4 == 0));
    APTAI_LAUNCH(sgemm_mfma_kernel, dim3((unsigned)ceil_div(N, 64), (unsigned)ceil_div(M, 64), (unsigned)(batch * g.split_k)),
                 dim3(256), 0, (hipStream_t)stream, g);
    APTAI_CHECK_LAUNCH("sgemm_mfma_kernel");
    if (g.split_k > 1) {
        APTAI_LAUNCH(sgemm_reduce_kernel, dim3(gridn(batch * M * N)), dim3(256), 0, (hipStream_t)stream, g, (int)batch);
        APTAI_CHECK_LAUNCH("sgemm_reduce_kernel");
    }
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
    APTAI_REQUIRE(ids && dout && demb_zeroed && rows > 0 && rows <= 8192, "aptai_embed_bwd: bad arguments (at most 8192 rows)");
    const uint32_t thr = drop_thr16(dropout_p);
    APTAI_LAUNCH(embed_bwd_kernel, dim3((unsigned)rows), dim3(128), 0, (hipStream_t)stream, ids, dout, demb_zeroed, (int)rows, (int)D,
                 drop_scale(thr), (uint32_t)seed, (uint32_t)(seed >> 32), thr);
    APTAI_CHECK_LAUNCH("embed_bwd_kernel");
    return APTAI_OK;
}

extern "C" int aptai_xattn_softmax_fwd(const float* raw, const int32_t* phn_ids, float* energy, float* att, float* att_log,
                                       int64_t* align, float* fs_rows, int64_t B, int64_t T, int64_t N, void* stream) {
    APTAI_REQUIRE(raw && phn_ids && energy && att && att_log && N > 0 && N <= 64, "aptai_xattn_softmax_fwd: bad arguments (N <= 64)");
    APTAI_REQUIRE(fs_rows == nullptr || N <= 63, "aptai_xattn_softmax_fwd: the forward-sum rows hold at most 63 phoneme slots");
    APTAI_LAUNCH(xattn_softmax_fwd_kernel, dim3((unsigned)ceil_div(B * T * 64, 256)), dim3(256), 0, (hipStream_t)stream, raw, phn_ids,
                 energy, att, att_log, align, fs_rows, (int)B, (int)T, (int)N);
    APTAI_CHECK_LAUNCH("xattn_softmax_fwd_kernel");
    return APTAI_OK;
}
extern "C" int aptai_xattn_softmax_bwd(const float* att, const float* att_log, const float* d_att, const float* d_attlog,
                                       int64_t ld_dattlog, float* d_raw, int64_t rows, int64_t N, void* stream) {
    APTAI_REQUIRE(att && att_log && d_raw && N > 0 && N <= 64, "aptai_xattn_softmax_bwd: bad arguments");
    APTAI_LAUNCH(xattn_softmax_bwd_kernel, dim3((unsigned)ceil_div(rows * 64, 256)), dim3(256), 0, (hipStream_t)stream, att, att_log,
                 d_att, d_attlog, (long)(ld_dattlog > 0 ? ld_dattlog : N), d_raw, (long)rows, (int)N);
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
extern "C" int64_t aptai_layernorm_f32_bwd_workspace_bytes(int64_t cols) { return (int64_t)LN32_BLOCKS * 2 * cols * 4; }
extern "C" int aptai_layernorm_f32_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                                       float* dx, float* dgamma, float* dbeta, float* workspace, int64_t rows, int64_t cols,
                                       void* stream) {
    APTAI_REQUIRE(dy && x && mean && rstd && gamma && dx && dgamma && dbeta && workspace && cols % 64 == 0 && cols <= 1024,
                  "aptai_layernorm_f32_bwd: bad arguments");
    long blocks = ceil_div(rows, 4);
    if (blocks > LN32_BLOCKS) blocks = LN32_BLOCKS;
    APTAI_LAUNCH(ln32_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dy, x, mean, rstd, gamma, dx, workspace,
                 (long)rows, (int)cols);
    APTAI_CHECK_LAUNCH("ln32_bwd_kernel");
    APTAI_LAUNCH(ln32_bwd_final_kernel, dim3((unsigned)ceil_div(2 * cols, 256)), dim3(256), 0, (hipStream_t)stream,
                 (const float*)workspace, dgamma, dbeta, (int)blocks, (int)cols);
    APTAI_CHECK_LAUNCH("ln32_bwd_final_kernel");
    return APTAI_OK;
}

extern "C" int aptai_lstm_fwd_serial(const float* xproj, const float* whhT, const int32_t* lens, float* hout, float* gates, float* cstate,
                              int64_t B, int64_t Tp, int64_t T, int64_t hidden, void* stream) {
    APTAI_REQUIRE(xproj && whhT && lens && hout, "aptai_lstm_fwd_serial: null pointer");
    APTAI_REQUIRE(hidden == HID, "aptai_lstm_fwd_serial: built for hidden size 256");
    APTAI_REQUIRE((gates == nullptr) == (cstate == nullptr), "aptai_lstm_fwd_serial: gates and cstate go together");
    APTAI_LAUNCH(lstm_fwd_kernel, dim3((unsigned)B, 2), dim3(HID), 0, (hipStream_t)stream, xproj, whhT, lens, hout, gates, cstate,
                 (int)Tp, (int)T);
    APTAI_CHECK_LAUNCH("lstm_fwd_kernel");
    return APTAI_OK;
}
extern "C" int aptai_lstm_bwd_serial(const float* dhout, const float* whh, const int32_t* lens, const float* gates, const float* cstate,
                              float* dgates, int64_t B, int64_t Tp, int64_t T, int64_t hidden, void* stream) {
    APTAI_REQUIRE(dhout && whh && lens && gates && cstate && dgates, "aptai_lstm_bwd_serial: null pointer");
    APTAI_REQUIRE(hidden == HID, "aptai_lstm_bwd_serial: built for hidden size 256");
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
extern "C" int64_t aptai_colsum_f32_workspace_bytes(int64_t N) { return (int64_t)CS_CHUNKS * N * 4; }
extern "C" int aptai_colsum_f32(const float* x, int64_t ld, float* out, float* workspace, int64_t rows, int64_t N, void* stream) {
    APTAI_REQUIRE(x && out && workspace && rows > 0 && N > 0, "aptai_colsum_f32: bad arguments");
    long chunks = ceil_div(rows, 64);
    if (chunks > CS_CHUNKS) chunks = CS_CHUNKS;
    const long rpc = ceil_div(rows, chunks);
    chunks = ceil_div(rows, rpc);
    APTAI_LAUNCH(colsum32_part_kernel, dim3((unsigned)ceil_div(N, 64), (unsigned)chunks), dim3(256), 0, (hipStream_t)stream, x, (long)ld,
                 workspace, (long)rows, (int)N, rpc);
    APTAI_CHECK_LAUNCH("colsum32_part_kernel");
    APTAI_LAUNCH(colsum32_final_kernel, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, out,
                 (int)chunks, (int)N);
    APTAI_CHECK_LAUNCH("colsum32_final_kernel");
    return APTAI_OK;
}
