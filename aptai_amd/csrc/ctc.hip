// CTC negative log-likelihood, forward (alpha) and backward (beta + gradient w.r.t. the LOGITS), gfx950.
// Replaces log_softmax + F.ctc_loss at models/w2v2_pr.py:59,73-81 (blank 0, 'mean', zero_infinity) and the
// per-sample nn.CTCLoss loop of ForwardSumLoss (models/modules.py:99-116: per-sample vocabulary, targets 1..N).
//
// One wave64 per utterance: the 2L+1 extended-label states live in registers (NS consecutive states per lane,
// neighbours through one lane shuffle), the per-frame log-softmax row goes through a V-float LDS array, the
// T-step recursion is sequential and barrier-free beyond that.  alpha is kept in HBM for the backward sweep
// (B*T*S fp32: 4 MB at B=16, T=499, L<=60).  Latency-bound by construction; fp32 like the reference.
#include "common.h"

namespace {

constexpr int MAXV = 256;
#define NEG_INF (-INFINITY)

__device__ __forceinline__ float lse2(float a, float b) {
    const float m = fmaxf(a, b);
    if (m == NEG_INF) return NEG_INF;
    return m + __logf(__expf(a - m) + __expf(b - m));
}
__device__ __forceinline__ float lse3(float a, float b, float c) {
    const float m = fmaxf(a, fmaxf(b, c));
    if (m == NEG_INF) return NEG_INF;
    return m + __logf(__expf(a - m) + __expf(b - m) + __expf(c - m));
}

struct CtcArgs {
    const float* logits; long ldl; long rows_per_b;
    const int* targets; long ldt;
    const int* input_lens; const int* target_lens; const int* vocab_sizes;
    int B, T, V, blank, S_max;
    float* log_probs;      // (T,B,V) or null
    float* alpha;          // [B][T][S_max]
    float* nll;            // [B]  (inf when infeasible)
};

// log-softmax of frame t of utterance b over its first Vb columns -> lp[] (LDS), optional global copy
__device__ __forceinline__ void frame_log_softmax(const CtcArgs& a, int b, int t, int Vb, float* lp, int lane) {
    const float* row = a.logits + ((long)b * a.rows_per_b + t) * a.ldl;
    float mx = NEG_INF;
    for (int v = lane; v < Vb; v += 64) mx = fmaxf(mx, row[v]);
    mx = wave_max(mx);
    float se = 0.f;
    for (int v = lane; v < Vb; v += 64) se += __expf(row[v] - mx);
    se = wave_sum(se);
    const float lz = mx + __logf(se);
    for (int v = lane; v < Vb; v += 64) lp[v] = row[v] - lz;
}

template <int NS>
__global__ __launch_bounds__(64) void ctc_alpha_kernel(CtcArgs a) {
    __shared__ float lp[MAXV];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int L = a.target_lens[b];
    int Tb = a.input_lens[b];
    Tb = Tb < a.T ? Tb : a.T;
    const int S = 2 * L + 1;
    const int Vb = a.vocab_sizes ? a.vocab_sizes[b] : a.V;
    int ext[NS];
    bool skip[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int s = lane * NS + i;
        int e = a.blank;
        if (s < S && (s & 1)) e = a.targets[(long)b * a.ldt + (s >> 1)];
        ext[i] = e;
        bool sk = false;
        if (s < S && (s & 1) && s >= 3) sk = e != a.targets[(long)b * a.ldt + (s >> 1) - 1];
        skip[i] = sk;                                            // s-2 -> s allowed (s odd, labels differ)
    }
    float al[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) al[i] = NEG_INF;
    float* aw = a.alpha + (long)b * a.T * a.S_max;
    for (int t = 0; t < Tb; ++t) {
        __syncthreads();
        frame_log_softmax(a, b, t, Vb, lp, lane);
        __syncthreads();
        if (a.log_probs)
            for (int v = lane; v < a.V; v += 64) a.log_probs[((long)t * a.B + b) * a.V + v] = v < Vb ? lp[v] : NEG_INF;
        float nw[NS];
        if (t == 0) {
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int s = lane * NS + i;
                nw[i] = (s < 2 && s < S) ? lp[ext[i]] : NEG_INF;
            }
        } else {
            float p1 = __shfl_up(al[NS - 1], 1, 64), p2 = __shfl_up(al[NS - 2], 1, 64);
            if (lane == 0) { p1 = NEG_INF; p2 = NEG_INF; }
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int s = lane * NS + i;
                const float a1 = i >= 1 ? al[i - 1] : p1;
                const float a2 = i >= 2 ? al[i - 2] : (i == 1 ? p1 : p2);
                const float v = lse3(al[i], a1, skip[i] ? a2 : NEG_INF);
                nw[i] = s < S ? v + lp[ext[i]] : NEG_INF;
            }
        }
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            al[i] = nw[i];
            const int s = lane * NS + i;
            if (s < a.S_max) aw[(long)t * a.S_max + s] = nw[i];
        }
    }
    // frames beyond the utterance in the (T,B,V) output: the reference's log_softmax covers every frame
    if (a.log_probs)
        for (int t = Tb; t < a.T; ++t) {
            __syncthreads();
            frame_log_softmax(a, b, t, Vb, lp, lane);
            __syncthreads();
            for (int v = lane; v < a.V; v += 64) a.log_probs[((long)t * a.B + b) * a.V + v] = v < Vb ? lp[v] : NEG_INF;
        }
    // log-likelihood = lse(alpha_{T-1}[S-1], alpha_{T-1}[S-2])
    float fin = NEG_INF;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int s = lane * NS + i;
        if (s == S - 1 || s == S - 2) fin = lse2(fin, al[i]);
    }
    float m = wave_max(fin);
    float ll = NEG_INF;
    if (m != NEG_INF) ll = m + __logf(wave_sum(fin == NEG_INF ? 0.f : __expf(fin - m)));
    if (Tb == 0) ll = (L == 0) ? 0.f : NEG_INF;
    if (lane == 0) a.nll[b] = -ll;
}

__global__ void ctc_reduce_kernel(const float* __restrict__ nll, const int* __restrict__ target_lens, int B, int reduction,
                                  int zero_infinity, float* __restrict__ loss) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0.0;
    for (int b = 0; b < B; ++b) {
        float v = nll[b];
        if (zero_infinity && isinf(v)) v = 0.f;
        if (reduction == 1) {
            int tl = target_lens[b];
            tl = tl < 1 ? 1 : tl;
            s += (double)v / tl;
        } else {
            s += (double)v;
        }
    }
    loss[0] = (float)(reduction == 1 ? s / B : s);
}

// backward sweep: beta recursion + gradient w.r.t. logits:  scale_b * (softmax - occupancy)
template <int NS, bool OUT_BF16>
__global__ __launch_bounds__(64) void ctc_beta_kernel(CtcArgs a, const float* __restrict__ grad_out, int reduction,
                                                      int zero_infinity, void* __restrict__ dlogits, long ldd, float extra_scale) {
    __shared__ float lp[MAXV];
    __shared__ float occ[MAXV];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int L = a.target_lens[b];
    int Tb = a.input_lens[b];
    Tb = Tb < a.T ? Tb : a.T;
    const int S = 2 * L + 1;
    const int Vb = a.vocab_sizes ? a.vocab_sizes[b] : a.V;
    const float nll = a.nll[b];
    float scale = (grad_out ? grad_out[0] : 1.f) * extra_scale;
    if (reduction == 1) scale /= (float)((L < 1 ? 1 : L)) * (float)a.B;
    const bool dead = isinf(nll) || isnan(nll);                  // infeasible alignment: zero_infinity -> zero gradient
    if (dead && zero_infinity) scale = 0.f;
    auto store_row = [&](int t, bool active) {
        const long r = (long)b * a.rows_per_b + t;
        for (int v = lane; v < (int)ldd; v += 64) {
            float gv = 0.f;
            if (active && v < Vb) gv = scale * (__expf(lp[v]) - occ[v]);
            if (OUT_BF16) ((bf16_t*)dlogits)[r * ldd + v] = f2bf(gv);
            else ((float*)dlogits)[r * ldd + v] = gv;
        }
    };
    for (int t = (int)a.rows_per_b - 1; t >= Tb; --t) store_row(t, false);
    int ext[NS];
    bool skipn[NS];                                              // s -> s+2 allowed
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int s = lane * NS + i;
        int e = a.blank;
        if (s < S && (s & 1)) e = a.targets[(long)b * a.ldt + (s >> 1)];
        ext[i] = e;
        bool sk = false;
        if ((s & 1) && s + 2 < S) sk = e != a.targets[(long)b * a.ldt + (s >> 1) + 1];
        skipn[i] = sk;
    }
    float be[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) be[i] = NEG_INF;
    const float* aw = a.alpha + (long)b * a.T * a.S_max;
    for (int t = Tb - 1; t >= 0; --t) {
        __syncthreads();
        frame_log_softmax(a, b, t, Vb, lp, lane);
        for (int v = lane; v < MAXV; v += 64) occ[v] = 0.f;
        __syncthreads();
        float nw[NS];
        if (t == Tb - 1) {
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int s = lane * NS + i;
                nw[i] = (s < S && (s == S - 1 || s == S - 2)) ? lp[ext[i]] : NEG_INF;
            }
        } else {
            float n1 = __shfl_down(be[0], 1, 64), n2 = __shfl_down(be[1], 1, 64);
            if (lane == 63) { n1 = NEG_INF; n2 = NEG_INF; }
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int s = lane * NS + i;
                const float b1 = i + 1 < NS ? be[i + 1 < NS ? i + 1 : 0] : n1;
                const float b2 = i + 2 < NS ? be[i + 2 < NS ? i + 2 : 0] : (i + 2 == NS ? n1 : n2);
                const float v = lse3(be[i], b1, skipn[i] ? b2 : NEG_INF);
                nw[i] = s < S ? v + lp[ext[i]] : NEG_INF;
            }
        }
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            be[i] = nw[i];
            const int s = lane * NS + i;
            if (s < S && !dead) {
                const float al = aw[(long)t * a.S_max + s];
                const float lg = al + nw[i] - lp[ext[i]] + nll;   // log occupancy, <= 0
                if (lg > -80.f) atomicAdd(&occ[ext[i]], __expf(lg));
            }
        }
        __syncthreads();
        store_row(t, !dead);
    }
}

int fill(CtcArgs& a, const char* who, const float* logits, int64_t ldl, int64_t rows_per_b, const int32_t* targets, int64_t ldt,
         const int32_t* input_lens, const int32_t* target_lens, const int32_t* vocab_sizes, int64_t B, int64_t T, int64_t V,
         int blank, float* alpha_ws, float* nll) {
    APTAI_REQUIRE(logits && targets && input_lens && target_lens && alpha_ws && nll, "%s: null pointer", who);
    APTAI_REQUIRE(B > 0 && T > 0 && V > 0 && V <= MAXV && ldl >= V && rows_per_b >= T, "%s: bad sizes (V=%ld, max %d)", who, (long)V, MAXV);
    APTAI_REQUIRE(ldt >= 1 && 2 * ldt + 1 <= 512, "%s: at most 255 labels per utterance (got row length %ld)", who, (long)ldt);
    APTAI_REQUIRE(blank >= 0 && blank < V, "%s: blank out of range", who);
    memset(&a, 0, sizeof(a));
    a.logits = logits; a.ldl = ldl; a.rows_per_b = rows_per_b; a.targets = targets; a.ldt = ldt;
    a.input_lens = input_lens; a.target_lens = target_lens; a.vocab_sizes = vocab_sizes;
    a.B = (int)B; a.T = (int)T; a.V = (int)V; a.blank = blank; a.S_max = (int)(2 * ldt + 1);
    a.alpha = alpha_ws; a.nll = nll;
    return APTAI_OK;
}

}  // namespace

extern "C" int64_t aptai_ctc_workspace_bytes(int64_t B, int64_t T, int64_t ldt) { return B * T * (2 * ldt + 1) * 4; }

extern "C" int aptai_ctc_fwd(const float* logits, int64_t ldl, int64_t rows_per_b, const int32_t* targets, int64_t ldt,
                             const int32_t* input_lens, const int32_t* target_lens, const int32_t* vocab_sizes, int64_t B,
                             int64_t T, int64_t V, int blank, int reduction, int zero_infinity, float* log_probs_out,
                             float* alpha_ws, float* nll, float* loss, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CtcArgs a;
    int rc = fill(a, "aptai_ctc_fwd", logits, ldl, rows_per_b, targets, ldt, input_lens, target_lens, vocab_sizes, B, T, V, blank,
                  alpha_ws, nll);
    if (rc) return rc;
    APTAI_REQUIRE(loss != nullptr && reduction >= 0 && reduction <= 2, "aptai_ctc_fwd: bad loss/reduction");
    a.log_probs = log_probs_out;
    const int ns = (a.S_max + 63) / 64;
    if (ns <= 2) APTAI_LAUNCH(ctc_alpha_kernel<2>, dim3((unsigned)B), dim3(64), 0, stream, a);
    else if (ns <= 4) APTAI_LAUNCH(ctc_alpha_kernel<4>, dim3((unsigned)B), dim3(64), 0, stream, a);
    else APTAI_LAUNCH(ctc_alpha_kernel<8>, dim3((unsigned)B), dim3(64), 0, stream, a);
    APTAI_CHECK_LAUNCH("ctc_alpha_kernel");
    if (reduction != 0) {
        APTAI_LAUNCH(ctc_reduce_kernel, dim3(1), dim3(64), 0, stream, (const float*)nll, target_lens, (int)B, reduction,
                     zero_infinity, loss);
        APTAI_CHECK_LAUNCH("ctc_reduce_kernel");
    }
    return APTAI_OK;
}

extern "C" int aptai_ctc_bwd(const float* logits, int64_t ldl, int64_t rows_per_b, const int32_t* targets, int64_t ldt,
                             const int32_t* input_lens, const int32_t* target_lens, const int32_t* vocab_sizes, int64_t B,
                             int64_t T, int64_t V, int blank, int reduction, int zero_infinity, const float* alpha_ws,
                             const float* nll, const float* grad_out, float extra_scale, void* dlogits, int64_t ldd,
                             int out_bf16, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CtcArgs a;
    int rc = fill(a, "aptai_ctc_bwd", logits, ldl, rows_per_b, targets, ldt, input_lens, target_lens, vocab_sizes, B, T, V, blank,
                  (float*)alpha_ws, (float*)nll);
    if (rc) return rc;
    APTAI_REQUIRE(dlogits != nullptr && ldd >= V, "aptai_ctc_bwd: bad dlogits");
    const int ns = (a.S_max + 63) / 64;
#define CTC_B(NSV)                                                                                                              \
    do {                                                                                                                        \
        if (out_bf16) APTAI_LAUNCH((ctc_beta_kernel<NSV, true>), dim3((unsigned)B), dim3(64), 0, stream, a, grad_out, reduction,  \
                                   zero_infinity, dlogits, (long)ldd, extra_scale);                                             \
        else APTAI_LAUNCH((ctc_beta_kernel<NSV, false>), dim3((unsigned)B), dim3(64), 0, stream, a, grad_out, reduction,          \
                          zero_infinity, dlogits, (long)ldd, extra_scale);                                                      \
    } while (0)
    if (ns <= 2) CTC_B(2);
    else if (ns <= 4) CTC_B(4);
    else CTC_B(8);
#undef CTC_B
    APTAI_CHECK_LAUNCH("ctc_beta_kernel");
    return APTAI_OK;
}
