// APTAI task-head kernels (gfx950): the fp64 low-pass FIR of models/modules.py:46-61 (no host round trip),
// the masked MSE + masked cross-entropy + frame argmax of models/aptai.py:89-106, and their gradients.
// Tiny, latency-bound work: everything stays on the device and on the caller's stream.
#include "common.h"

namespace {

constexpr int MAXTAPS = 64;

// y[b][t][c] = sum_j taps[j] * x[b][t + j - N/2][c]   ('same' zero padding over t in [0,T)), fp64 accumulate.
// in:  fp32 with row stride ldx (rows b*Tp_in + t);  out: fp32 (ldy) or bf16 (ldy) with zero fill of cols >= C.
template <bool OUT_BF16>
__global__ void fir_kernel(const float* __restrict__ x, long ldx, long rows_per_b_in, const double* __restrict__ taps, int ntaps,
                           void* __restrict__ y, long ldy, long rows_per_b_out, int B, int T_static, int T_out, int C, int C_out,
                           const int* __restrict__ bounds) {
    __shared__ double tp[MAXTAPS];
    // frames at or beyond the bound do not exist for the filter (zero padding starts there) and are written as zeros: a graph
    // captured for a bucket length then equals the eager run on the batch's own padded length (runtime.hip, aptai_set_frame_bounds)
    int T = T_static;
    if (bounds != nullptr) { const int tb = bounds[1]; T = tb < 1 ? 1 : (tb < T_static ? tb : T_static); }
    if ((int)threadIdx.x < ntaps) tp[threadIdx.x] = taps[threadIdx.x];
    __syncthreads();
    const long n = (long)B * T_out * C_out;
    const long stride = (long)gridDim.x * blockDim.x;
    const int half = ntaps / 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int c = (int)(i % C_out);
        const int t = (int)((i / C_out) % T_out);
        const int b = (int)(i / ((long)C_out * T_out));
        double acc = 0.0;
        if (c < C && t < T) {
            const float* xb = x + (long)b * rows_per_b_in * ldx + c;
            // branch-free so the 51 strided loads are in flight together (the guarded form waited for each in turn, 20 us);
            // a tap outside [0, T) reads frame t with weight 0: the sum and its order are unchanged
#pragma unroll 17
            for (int j = 0; j < ntaps; ++j) {
                const int tt = t + j - half;
                const bool in = (unsigned)tt < (unsigned)T;
                acc += (in ? tp[j] : 0.0) * (double)xb[(long)(in ? tt : t) * ldx];
            }
        }
        const long o = ((long)b * rows_per_b_out + t) * ldy + c;
        if (OUT_BF16) ((bf16_t*)y)[o] = f2bf((float)acc);
        else ((float*)y)[o] = (float)acc;
    }
}

struct LossArgs {
    const float* tv_pred;      // [B][T][n_tv]
    const float* tv_tgt;       // [B][T][n_tv]  (-100.0 = padding)
    const float* logits; long ldl; long rows_per_b;   // rows b*rows_per_b + t
    const int64_t* phn_tgt;    // [B][T] (0 = padding)
    int B, T, n_tv, n_phn;
    float* partials;           // [blocks][4]
    float* scalars;            // loss, mse, ce, n_tv_valid, n_phn_valid
    int64_t* pred;             // [B][T]
    float w_mse, w_ce;
};

// 8 lanes per frame (lane `sub` owns logits sub, sub+8, ... and tracks sub, sub+8): cross-entropy + argmax over n_phn logits,
// squared error over n_tv tracks.  One thread per frame left 32 blocks of latency-bound serial loops (20 + 34 us fwd / bwd).
__device__ __forceinline__ float grp8_sum(float v) {
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
    return v;
}
__device__ __forceinline__ float grp8_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 1, 64)); v = fmaxf(v, __shfl_xor(v, 2, 64)); v = fmaxf(v, __shfl_xor(v, 4, 64));
    return v;
}

__global__ void loss_fwd_kernel(LossArgs a) {
    __shared__ float red[4][256];
    const long frames = (long)a.B * a.T;
    const int sub = threadIdx.x & 7;
    float sse = 0.f, ntv = 0.f, ce = 0.f, nph = 0.f;
    // every lane of an 8-lane group takes the same trip count (the group index decides), so the shuffles are convergent
    for (long f = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 3; f < frames; f += ((long)gridDim.x * blockDim.x) >> 3) {
        const int b = (int)(f / a.T), t = (int)(f % a.T);
        for (int c = sub; c < a.n_tv; c += 8) {
            const float tg = a.tv_tgt[f * a.n_tv + c];
            if (tg != -100.0f) {
                const float d = a.tv_pred[f * a.n_tv + c] - tg;
                sse = fmaf(d, d, sse);
                ntv += 1.f;
            }
        }
        const float* lg = a.logits + ((long)b * a.rows_per_b + t) * a.ldl;
        float mx = -INFINITY;
        int am = 0x7fffffff;
        for (int k = sub; k < a.n_phn; k += 8) {
            const float v = lg[k];
            if (v > mx || am == 0x7fffffff) { mx = v; am = k; }      // first maximum of this lane's (ascending) classes
        }
        // first maximum over the group: larger value wins, equal values -> smaller index (torch.argmax)
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            const float ov = __shfl_xor(mx, o, 64);
            const int oi = __shfl_xor(am, o, 64);
            if (ov > mx || (ov == mx && oi < am)) { mx = ov; am = oi; }
        }
        if (a.pred && sub == 0) a.pred[f] = am;
        const int64_t tg = a.phn_tgt[f];
        if (tg != 0) {                                              // uniform within the group
            float se = 0.f;
            for (int k = sub; k < a.n_phn; k += 8) se += __expf(lg[k] - mx);
            se = grp8_sum(se);
            if (sub == 0) {
                ce += (mx + __logf(se)) - lg[tg];
                nph += 1.f;
            }
        }
    }
    red[0][threadIdx.x] = sse; red[1][threadIdx.x] = ntv; red[2][threadIdx.x] = ce; red[3][threadIdx.x] = nph;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
            for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x < 4) a.partials[blockIdx.x * 4 + threadIdx.x] = red[threadIdx.x][0];
}

// one wave: lane l sums the partials of blocks l, l + 64, ... in double, then a fixed-order butterfly
__global__ void loss_final_kernel(const float* __restrict__ partials, int nblocks, float* __restrict__ scalars, float w_mse,
                                  float w_ce) {
    double s[4] = {0, 0, 0, 0};
    for (int b = threadIdx.x; b < nblocks; b += 64)
        for (int k = 0; k < 4; ++k) s[k] += partials[b * 4 + k];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s[k] += __shfl_xor(s[k], o, 64);
    if (threadIdx.x != 0) return;
    const float mse = (float)(s[0] / s[1]);        // 0/0 -> nan like F.mse_loss on an empty selection
    const float ce = (w_ce == 0.f && s[3] == 0.0) ? 0.f : (float)(s[2] / s[3]);   // tv-only use (Force_APTAI): no phoneme term
    scalars[0] = w_mse * mse + w_ce * ce;
    scalars[1] = mse;
    scalars[2] = ce;
    scalars[3] = (float)s[1];
    scalars[4] = (float)s[3];
}

// d_tv [B][T][n_tv] fp32, d_logits bf16 [B*rows_per_b][ldd] (zero outside the valid region); 8 lanes per row
__global__ void loss_bwd_kernel(LossArgs a, const float* __restrict__ gout, float* __restrict__ d_tv, bf16_t* __restrict__ d_logits,
                                long ldd) {
    const float g = gout ? gout[0] : 1.f;
    const float k_mse = g * a.w_mse * 2.f / a.scalars[3];
    const float k_ce = g * a.w_ce / a.scalars[4];
    const long rows = (long)a.B * a.rows_per_b;
    const int sub = threadIdx.x & 7;
    for (long r = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 3; r < rows; r += ((long)gridDim.x * blockDim.x) >> 3) {
        const int b = (int)(r / a.rows_per_b), t = (int)(r % a.rows_per_b);
        bf16_t* dl = d_logits + r * ldd;
        if (t >= a.T) {
            for (int k = sub; k < ldd; k += 8) dl[k] = 0;
            continue;
        }
        const long f = (long)b * a.T + t;
        for (int c = sub; c < a.n_tv; c += 8) {
            const float tg = a.tv_tgt[f * a.n_tv + c];
            d_tv[f * a.n_tv + c] = tg != -100.0f ? k_mse * (a.tv_pred[f * a.n_tv + c] - tg) : 0.f;
        }
        const int64_t tg = a.phn_tgt[f];
        if (tg == 0) {
            for (int k = sub; k < ldd; k += 8) dl[k] = 0;
            continue;
        }
        const float* lg = a.logits + r * a.ldl;
        float mx = -INFINITY;
        for (int k = sub; k < a.n_phn; k += 8) mx = fmaxf(mx, lg[k]);
        mx = grp8_max(mx);
        float se = 0.f;
        for (int k = sub; k < a.n_phn; k += 8) se += __expf(lg[k] - mx);
        se = grp8_sum(se);
        const float inv = 1.f / se;
        for (int k = sub; k < ldd; k += 8) {
            float v = 0.f;
            if (k < a.n_phn) v = k_ce * (__expf(lg[k] - mx) * inv - (k == tg ? 1.f : 0.f));
            dl[k] = f2bf(v);
        }
    }
}

constexpr int LOSS_BLOCKS = 128;

}  // namespace

extern "C" int aptai_lowpass_fir(const float* x, int64_t ldx, int64_t rows_per_b_in, const double* taps, int64_t ntaps,
                                 void* y, int64_t ldy, int64_t rows_per_b_out, int out_bf16, int64_t B, int64_t T,
                                 int64_t T_out, int64_t C, int64_t C_out, void* stream) {
    APTAI_REQUIRE(x && taps && y, "aptai_lowpass_fir: null pointer");
    APTAI_REQUIRE(ntaps > 0 && ntaps <= MAXTAPS && (ntaps & 1), "aptai_lowpass_fir: ntaps=%ld (odd, <= %d)", (long)ntaps, MAXTAPS);
    APTAI_REQUIRE(B > 0 && T > 0 && T_out >= T && C > 0 && C_out >= C, "aptai_lowpass_fir: bad sizes");
    const long n = B * T_out * C_out;
    unsigned blocks = (unsigned)(ceil_div(n, 256) > 2048 ? 2048 : ceil_div(n, 256));
    if (out_bf16)
        APTAI_LAUNCH(fir_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, (long)ldx, (long)rows_per_b_in,
                           taps, (int)ntaps, y, (long)ldy, (long)rows_per_b_out, (int)B, (int)T, (int)T_out, (int)C, (int)C_out,
                           (const int*)aptai_frame_bounds(stream));
    else
        APTAI_LAUNCH(fir_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, (long)ldx, (long)rows_per_b_in,
                           taps, (int)ntaps, y, (long)ldy, (long)rows_per_b_out, (int)B, (int)T, (int)T_out, (int)C, (int)C_out,
                           (const int*)aptai_frame_bounds(stream));
    APTAI_CHECK_LAUNCH("fir_kernel");
    return APTAI_OK;
}

extern "C" int64_t aptai_aptai_loss_workspace_bytes(void) { return LOSS_BLOCKS * 4 * 4; }

static int fill_loss(LossArgs& a, const float* tv_pred, const float* tv_tgt, const float* logits, int64_t ldl,
                     int64_t rows_per_b, const int64_t* phn_tgt, int64_t B, int64_t T, int64_t n_tv, int64_t n_phn,
                     float w_mse, float w_ce) {
    APTAI_REQUIRE(tv_pred && tv_tgt && logits && phn_tgt, "aptai_aptai_loss: null pointer");
    APTAI_REQUIRE(B > 0 && T > 0 && rows_per_b >= T && n_tv > 0 && n_phn > 0 && ldl >= n_phn, "aptai_aptai_loss: bad sizes");
    memset(&a, 0, sizeof(a));
    a.tv_pred = tv_pred; a.tv_tgt = tv_tgt; a.logits = logits; a.ldl = ldl; a.rows_per_b = rows_per_b; a.phn_tgt = phn_tgt;
    a.B = (int)B; a.T = (int)T; a.n_tv = (int)n_tv; a.n_phn = (int)n_phn; a.w_mse = w_mse; a.w_ce = w_ce;
    return APTAI_OK;
}

extern "C" int aptai_aptai_loss_fwd(const float* tv_pred, const float* tv_tgt, const float* logits, int64_t ldl,
                                    int64_t rows_per_b, const int64_t* phn_tgt, int64_t B, int64_t T, int64_t n_tv,
                                    int64_t n_phn, float w_mse, float w_ce, float* scalars, int64_t* pred, void* workspace,
                                    void* stream) {
    LossArgs a;
    int rc = fill_loss(a, tv_pred, tv_tgt, logits, ldl, rows_per_b, phn_tgt, B, T, n_tv, n_phn, w_mse, w_ce);
    if (rc) return rc;
    APTAI_REQUIRE(scalars && workspace, "aptai_aptai_loss_fwd: null pointer");
    a.partials = (float*)workspace; a.scalars = scalars; a.pred = pred;
    APTAI_LAUNCH(loss_fwd_kernel, dim3(LOSS_BLOCKS), dim3(256), 0, (hipStream_t)stream, a);
    APTAI_CHECK_LAUNCH("loss_fwd_kernel");
    APTAI_LAUNCH(loss_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const float*)workspace, LOSS_BLOCKS, scalars,
                       w_mse, w_ce);
    APTAI_CHECK_LAUNCH("loss_final_kernel");
    return APTAI_OK;
}

extern "C" int aptai_aptai_loss_bwd(const float* tv_pred, const float* tv_tgt, const float* logits, int64_t ldl,
                                    int64_t rows_per_b, const int64_t* phn_tgt, int64_t B, int64_t T, int64_t n_tv,
                                    int64_t n_phn, float w_mse, float w_ce, const float* scalars, const float* grad_out,
                                    float* d_tv, void* d_logits_bf16, int64_t ldd, void* stream) {
    LossArgs a;
    int rc = fill_loss(a, tv_pred, tv_tgt, logits, ldl, rows_per_b, phn_tgt, B, T, n_tv, n_phn, w_mse, w_ce);
    if (rc) return rc;
    APTAI_REQUIRE(scalars && d_tv && d_logits_bf16 && ldd >= n_phn, "aptai_aptai_loss_bwd: bad arguments");
    a.scalars = (float*)scalars;
    const long rows = B * rows_per_b;
    APTAI_LAUNCH(loss_bwd_kernel, dim3((unsigned)ceil_div(rows * 8, 256)), dim3(256), 0, (hipStream_t)stream, a, grad_out, d_tv,
                       (bf16_t*)d_logits_bf16, (long)ldd);
    APTAI_CHECK_LAUNCH("loss_bwd_kernel");
    return APTAI_OK;
}
