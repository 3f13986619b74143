// CTC negative log-likelihood, forward (alpha) and backward (beta + gradient w.r.t. the LOGITS), gfx950.
// Replaces log_softmax + F.ctc_loss at models/w2v2_pr.py:59,73-81 (blank 0, 'mean', zero_infinity) and the
// per-sample nn.CTCLoss loop of ForwardSumLoss (models/modules.py:99-116: per-sample vocabulary, targets 1..N).
//
// The T-step recursions are inherently sequential, so everything that is NOT sequential is taken out of them:
//   1. ctc_lpe_kernel    (parallel, one wave per frame): log-softmax of every frame; the (T,B,V) log_probs output; and
//                         lpe[b][t][s] = log p_t(ext_s), the only values the recursions read (2L+1 per frame instead of V);
//   2. ctc_recur_kernel  (one wave per utterance and direction; alpha and beta run SIDE BY SIDE in one launch): states in
//                         registers (NS consecutive states per lane, neighbours by one lane shuffle), lpe rows prefetched four
//                         frames ahead so no step waits for memory; ~150 cycles per frame instead of two barriers, a
//                         dependent global load and a log-softmax per frame (2.6 us per frame before: rocprofv3, round 2);
//   3. ctc_grad_kernel   (parallel, one wave per frame): state occupancies exp(alpha + beta - lpe + nll) scattered to their
//                         labels in LDS, gradient row = scale * (softmax - occupancy).
// alpha | beta | lpe live in one caller workspace (3 x B*T*S fp32: 12 MB at B=16, T=499, L<=60).  fp32 like the reference.
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

// ---- 1. per-frame log-softmax, (T,B,V) output, lpe gather.  grid = ceil(B*T/4) blocks of 4 waves, one frame per wave.
__global__ __launch_bounds__(256) void ctc_lpe_kernel(CtcArgs a, float* __restrict__ lpe) {
    __shared__ float lps[4][MAXV];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long f = (long)blockIdx.x * 4 + wave;
    if (f >= (long)a.B * a.T) return;
    const int b = (int)(f / a.T), t = (int)(f % a.T);
    const int Vb = a.vocab_sizes ? a.vocab_sizes[b] : a.V;
    const float* row = a.logits + ((long)b * a.rows_per_b + t) * a.ldl;
    float x[MAXV / 64];
    float mx = NEG_INF;
#pragma unroll
    for (int j = 0; j < MAXV / 64; ++j) {
        const int v = j * 64 + lane;
        x[j] = v < Vb ? row[v] : NEG_INF;
        mx = fmaxf(mx, x[j]);
    }
    mx = wave_max(mx);
    float se = 0.f;
#pragma unroll
    for (int j = 0; j < MAXV / 64; ++j) se += (j * 64 + lane < Vb) ? __expf(x[j] - mx) : 0.f;
    se = wave_sum(se);
    const float lz = mx + __logf(se);
    float* lp = lps[wave];
#pragma unroll
    for (int j = 0; j < MAXV / 64; ++j) {
        const int v = j * 64 + lane;
        if (v < a.V) {
            const float l = v < Vb ? x[j] - lz : NEG_INF;
            lp[v] = l;
            if (a.log_probs) a.log_probs[((long)t * a.B + b) * a.V + v] = l;
        }
    }
    int Tb = a.input_lens[b];
    Tb = Tb < a.T ? Tb : a.T;
    if (t >= Tb) return;                                          // the recursions never read frames beyond the utterance
    const int L = a.target_lens[b], S = 2 * L + 1;
    float* out = lpe + ((long)b * a.T + t) * a.S_max;
    for (int s = lane; s < a.S_max; s += 64) {
        float l = NEG_INF;
        if (s < S) {
            const int e = (s & 1) ? a.targets[(long)b * a.ldt + (s >> 1)] : a.blank;
            l = (e >= 0 && e < a.V) ? lp[e] : NEG_INF;
        }
        out[s] = l;
    }
}

// ---- 2. the recursions.  grid (B, 1 or 2): y = 0 alpha (+ nll), y = 1 beta.  One wave each; NS states per lane.
template <int NS>
__device__ __forceinline__ void load_row(const float* __restrict__ lpe, long row, int S_max, int lane, float (&dst)[NS]) {
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int s = lane * NS + i;
        dst[i] = s < S_max ? lpe[row * S_max + s] : NEG_INF;
    }
}

template <int NS>
__global__ __launch_bounds__(64) void ctc_recur_kernel(CtcArgs a, const float* __restrict__ lpe, float* __restrict__ beta, int first_dir) {
    const int b = blockIdx.x, lane = threadIdx.x, dir = first_dir + blockIdx.y;
    const int L = a.target_lens[b];
    int Tb = a.input_lens[b];
    Tb = Tb < a.T ? Tb : a.T;
    const int S = 2 * L + 1;
    bool skip[NS];                       // alpha: s-2 -> s allowed; beta: s -> s+2 allowed
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int s = lane * NS + i;
        bool sk = false;
        if (s < S && (s & 1)) {
            const int e = a.targets[(long)b * a.ldt + (s >> 1)];
            if (dir == 0) { if (s >= 3) sk = e != a.targets[(long)b * a.ldt + (s >> 1) - 1]; }
            else { if (s + 2 < S) sk = e != a.targets[(long)b * a.ldt + (s >> 1) + 1]; }
        }
        skip[i] = sk;
    }
    const long base = (long)b * a.T;
    float st[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) st[i] = NEG_INF;
    float* outw = (dir == 0 ? a.alpha : beta) + base * a.S_max;
    // four-deep prefetch ring of lpe rows (static register names: runtime-indexed arrays would go to scratch)
    float r0[NS], r1[NS], r2[NS], r3[NS];
    auto frame = [&](int step) { return dir == 0 ? step : Tb - 1 - step; };
    auto fetch = [&](int step, float (&dst)[NS]) {
        if (step < Tb) load_row<NS>(lpe, base + frame(step), a.S_max, lane, dst);
    };
    fetch(0, r0); fetch(1, r1); fetch(2, r2); fetch(3, r3);
    auto advance = [&](int step, const float (&lp)[NS]) {
        float nw[NS];
        if (step == 0) {
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int s = lane * NS + i;
                const bool init = dir == 0 ? (s < 2) : (s == S - 1 || s == S - 2);
                nw[i] = (init && s < S) ? lp[i] : NEG_INF;
            }
        } else if (dir == 0) {
            float p1 = __shfl_up(st[NS - 1], 1, 64), p2 = __shfl_up(st[NS - 2], 1, 64);
            if (lane == 0) { p1 = NEG_INF; p2 = NEG_INF; }
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int s = lane * NS + i;
                const float a1 = i >= 1 ? st[i >= 1 ? i - 1 : 0] : p1;
                const float a2 = i >= 2 ? st[i >= 2 ? i - 2 : 0] : (i == 1 ? p1 : p2);
                const float v = lse3(st[i], a1, skip[i] ? a2 : NEG_INF);
                nw[i] = s < S ? v + lp[i] : NEG_INF;
            }
        } else {
            float n1 = __shfl_down(st[0], 1, 64), n2 = __shfl_down(st[1], 1, 64);
            if (lane == 63) { n1 = NEG_INF; n2 = NEG_INF; }
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int s = lane * NS + i;
                const float b1 = i + 1 < NS ? st[i + 1 < NS ? i + 1 : 0] : n1;
                const float b2 = i + 2 < NS ? st[i + 2 < NS ? i + 2 : 0] : (i + 2 == NS ? n1 : n2);
                const float v = lse3(st[i], b1, skip[i] ? b2 : NEG_INF);
                nw[i] = s < S ? v + lp[i] : NEG_INF;
            }
        }
        const long t = frame(step);
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            st[i] = nw[i];
            const int s = lane * NS + i;
            if (s < a.S_max) outw[t * a.S_max + s] = nw[i];
        }
    };
    for (int step = 0; step < Tb; step += 4) {
        advance(step, r0);
        fetch(step + 4, r0);
        if (step + 1 < Tb) { advance(step + 1, r1); fetch(step + 5, r1); }
        if (step + 2 < Tb) { advance(step + 2, r2); fetch(step + 6, r2); }
        if (step + 3 < Tb) { advance(step + 3, r3); fetch(step + 7, r3); }
    }
    if (dir != 0) return;
    // log-likelihood = lse(alpha_{T-1}[S-1], alpha_{T-1}[S-2])
    float fin = NEG_INF;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int s = lane * NS + i;
        if (s == S - 1 || s == S - 2) fin = lse2(fin, st[i]);
    }
    const float m = wave_max(fin);
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

// ---- 3. gradient rows:  scale_b * (softmax - occupancy).  One wave per row (b, t) of the [B][rows_per_b] output.
// The occupancy of a label is a sum over the states that carry it.  Round 3 scattered exp(.) into LDS with float atomicAdd: the order
// of those additions is not defined, so the last bits of the gradient changed from launch to launch (and Adam's sign-like first steps
// turned that into O(lr) parameter differences: the graph-vs-eager loop test of round 3).  Now ORDER-FIXED: every state's occupancy is
// written to its own LDS word; the blank's sum runs per lane over its (even) states in increasing order and then through the fixed
// shuffle tree; the lane that owns vocabulary entry v adds the states of the labels equal to v in label order.  Same inputs -> same bits.
constexpr int CTC_MAXS = 512;                                     // 2 * 255 + 1 states at most (checked in fill())
template <bool OUT_BF16>
__global__ __launch_bounds__(256) void ctc_grad_kernel(CtcArgs a, const float* __restrict__ lpe, const float* __restrict__ beta,
                                                       const float* __restrict__ grad_out, int reduction, int zero_infinity,
                                                       void* __restrict__ dlogits, long ldd, float extra_scale) {
    __shared__ float ev[4][CTC_MAXS];                             // occupancy of state s
    __shared__ int lab[4][CTC_MAXS / 2];                          // label of odd state 2 i + 1
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long f = (long)blockIdx.x * 4 + wave;
    const bool valid = f < (long)a.B * a.rows_per_b;              // every thread reaches the barrier below
    const int b = valid ? (int)(f / a.rows_per_b) : 0, t = valid ? (int)(f % a.rows_per_b) : 0;
    int Tb = a.input_lens[b];
    Tb = Tb < a.T ? Tb : a.T;
    const int L = a.target_lens[b], S = 2 * L + 1;
    const int Vb = a.vocab_sizes ? a.vocab_sizes[b] : a.V;
    const float nll = a.nll[b];
    float scale = (grad_out ? grad_out[0] : 1.f) * extra_scale;
    if (reduction == 1) scale /= (float)((L < 1 ? 1 : L)) * (float)a.B;
    const bool dead = isinf(nll) || isnan(nll);                  // infeasible alignment: zero_infinity -> zero gradient
    if (dead && zero_infinity) scale = 0.f;
    const bool active = valid && t < Tb && !dead;
    const long r = (long)b * a.rows_per_b + t;
    float blank_part = 0.f;
    if (active) {
        const long off = ((long)b * a.T + t) * a.S_max;
        for (int s = lane; s < S; s += 64) {                     // s and lane have the same parity: even lanes own the blank states
            const float lg = a.alpha[off + s] + beta[off + s] - lpe[off + s] + nll;        // log occupancy, <= 0
            const float o = lg > -80.f ? __expf(lg) : 0.f;
            ev[wave][s] = o;
            if (s & 1) lab[wave][s >> 1] = a.targets[(long)b * a.ldt + (s >> 1)];
            else blank_part += o;
        }
    }
    const float blank_occ = wave_sum(blank_part);                // fixed shuffle tree
    __syncthreads();
    // softmax of the row (recomputed: V <= 256 values)
    const float* row = a.logits + r * a.ldl;
    float x[MAXV / 64];
    float mx = NEG_INF;
#pragma unroll
    for (int j = 0; j < MAXV / 64; ++j) {
        const int v = j * 64 + lane;
        x[j] = (active && v < Vb) ? row[v] : NEG_INF;
        mx = fmaxf(mx, x[j]);
    }
    mx = wave_max(mx);
    float se = 0.f;
#pragma unroll
    for (int j = 0; j < MAXV / 64; ++j) se += (active && j * 64 + lane < Vb) ? __expf(x[j] - mx) : 0.f;
    se = wave_sum(se);
    const float inv = active ? 1.f / se : 0.f;
    if (!valid) return;
#pragma unroll
    for (int j = 0; j < MAXV / 64; ++j) {
        const int v = j * 64 + lane;
        if (j * 64 >= (int)ldd) break;                           // wave-uniform
        float occ = 0.f;
        if (active && j * 64 < a.V) {
            occ = v == a.blank ? blank_occ : 0.f;
            for (int i = 0; i < L; ++i) occ += lab[wave][i] == v ? ev[wave][2 * i + 1] : 0.f;   // LDS broadcasts, label order
        }
        if (v >= (int)ldd) continue;
        float gv = 0.f;
        if (active && v < Vb) gv = scale * (__expf(x[j] - mx) * inv - occ);
        if (OUT_BF16) ((bf16_t*)dlogits)[r * ldd + v] = f2bf(gv);
        else ((float*)dlogits)[r * ldd + v] = gv;
    }
}

// ---- best-path (greedy) CTC decode on the device: frame argmax (first maximum, like torch.argmax) -> collapse repeats -> drop
// blank, over ALL T frames of every row of the padded batch (the reference's decoder call passes no lengths,
// models/w2v2_pr.py:155).  One wave per utterance, 64 frames per round: ballot + prefix popcount compacts the kept labels.
// ids_out int32 [B][max_n] zero-padded; n_out int32 [B] = decoded length (may exceed max_n: the caller checks).
__global__ __launch_bounds__(64) void ctc_greedy_decode_kernel(const float* __restrict__ logits, long ldl, long rows_per_b, int T, int V,
                                                               int blank, int* __restrict__ ids_out, int max_n, int* __restrict__ n_out) {
    const int b = blockIdx.x, lane = threadIdx.x;
    int count = 0, carry = -1;                                   // label of the last frame of the previous round
    for (int l = lane; l < max_n; l += 64) ids_out[(long)b * max_n + l] = 0;
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int t = t0 + lane;
        int arg = -1;
        if (t < T) {
            const float* row = logits + ((long)b * rows_per_b + t) * ldl;
            float best = row[0];
            arg = 0;
            for (int v = 1; v < V; ++v) {
                const float x = row[v];
                if (x > best) { best = x; arg = v; }
            }
        }
        int prev = __shfl_up(arg, 1, 64);
        if (lane == 0) prev = carry;
        const bool keep = t < T && arg != prev && arg != blank;
        const unsigned long long m = __ballot(keep);
        const int pos = count + __popcll(m & ((1ull << lane) - 1ull));
        if (keep && pos < max_n) ids_out[(long)b * max_n + pos] = arg;
        count += __popcll(m);
        carry = __shfl(arg, 63, 64);
    }
    if (lane == 0) n_out[b] = count;
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

extern "C" int64_t aptai_ctc_workspace_bytes(int64_t B, int64_t T, int64_t ldt) { return 3 * B * T * (2 * ldt + 1) * 4; }

template <int NS>
static void launch_recur(const CtcArgs& a, const float* lpe, float* beta, int first_dir, int ndir, hipStream_t stream) {
    APTAI_LAUNCH(ctc_recur_kernel<NS>, dim3((unsigned)a.B, (unsigned)ndir), dim3(64), 0, stream, a, lpe, beta, first_dir);
}
static void launch_recur_ns(const CtcArgs& a, const float* lpe, float* beta, int first_dir, int ndir, hipStream_t stream) {
    const int ns = (a.S_max + 63) / 64;
    if (ns <= 2) launch_recur<2>(a, lpe, beta, first_dir, ndir, stream);
    else if (ns <= 4) launch_recur<4>(a, lpe, beta, first_dir, ndir, stream);
    else launch_recur<8>(a, lpe, beta, first_dir, ndir, stream);
}

extern "C" int aptai_ctc_fwd(const float* logits, int64_t ldl, int64_t rows_per_b, const int32_t* targets, int64_t ldt,
                             const int32_t* input_lens, const int32_t* target_lens, const int32_t* vocab_sizes, int64_t B,
                             int64_t T, int64_t V, int blank, int reduction, int zero_infinity, float* log_probs_out,
                             float* workspace, float* nll, float* loss, int want_beta, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CtcArgs a;
    int rc = fill(a, "aptai_ctc_fwd", logits, ldl, rows_per_b, targets, ldt, input_lens, target_lens, vocab_sizes, B, T, V, blank,
                  workspace, nll);
    if (rc) return rc;
    APTAI_REQUIRE(loss != nullptr && reduction >= 0 && reduction <= 2, "aptai_ctc_fwd: bad loss/reduction");
    a.log_probs = log_probs_out;
    const long plane = (long)B * T * a.S_max;
    float* beta = workspace + plane;
    float* lpe = workspace + 2 * plane;
    APTAI_LAUNCH(ctc_lpe_kernel, dim3((unsigned)ceil_div(B * T, 4)), dim3(256), 0, stream, a, lpe);
    APTAI_CHECK_LAUNCH("ctc_lpe_kernel");
    launch_recur_ns(a, lpe, beta, 0, want_beta ? 2 : 1, stream);
    APTAI_CHECK_LAUNCH("ctc_recur_kernel");
    if (reduction != 0) {
        APTAI_LAUNCH(ctc_reduce_kernel, dim3(1), dim3(64), 0, stream, (const float*)nll, target_lens, (int)B, reduction,
                     zero_infinity, loss);
        APTAI_CHECK_LAUNCH("ctc_reduce_kernel");
    }
    return APTAI_OK;
}

extern "C" int aptai_ctc_bwd(const float* logits, int64_t ldl, int64_t rows_per_b, const int32_t* targets, int64_t ldt,
                             const int32_t* input_lens, const int32_t* target_lens, const int32_t* vocab_sizes, int64_t B,
                             int64_t T, int64_t V, int blank, int reduction, int zero_infinity, const float* workspace,
                             const float* nll, const float* grad_out, float extra_scale, void* dlogits, int64_t ldd,
                             int out_bf16, int beta_ready, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CtcArgs a;
    int rc = fill(a, "aptai_ctc_bwd", logits, ldl, rows_per_b, targets, ldt, input_lens, target_lens, vocab_sizes, B, T, V, blank,
                  (float*)workspace, (float*)nll);
    if (rc) return rc;
    APTAI_REQUIRE(dlogits != nullptr && ldd >= V && ldd <= MAXV, "aptai_ctc_bwd: bad dlogits (V <= ldd <= %d)", MAXV);
    const long plane = (long)B * T * a.S_max;
    float* beta = (float*)workspace + plane;
    const float* lpe = workspace + 2 * plane;
    if (!beta_ready) {
        launch_recur_ns(a, lpe, beta, 1, 1, stream);
        APTAI_CHECK_LAUNCH("ctc_recur_kernel (beta)");
    }
    const unsigned blocks = (unsigned)ceil_div(B * rows_per_b, 4);
    if (out_bf16) APTAI_LAUNCH((ctc_grad_kernel<true>), dim3(blocks), dim3(256), 0, stream, a, lpe, (const float*)beta, grad_out,
                               reduction, zero_infinity, dlogits, (long)ldd, extra_scale);
    else APTAI_LAUNCH((ctc_grad_kernel<false>), dim3(blocks), dim3(256), 0, stream, a, lpe, (const float*)beta, grad_out, reduction,
                      zero_infinity, dlogits, (long)ldd, extra_scale);
    APTAI_CHECK_LAUNCH("ctc_grad_kernel");
    return APTAI_OK;
}

extern "C" int aptai_ctc_greedy_decode(const float* logits, int64_t ldl, int64_t rows_per_b, int64_t B, int64_t T, int64_t V, int blank,
                                       int32_t* ids_out, int64_t max_n, int32_t* n_out, void* stream) {
    APTAI_REQUIRE(logits && ids_out && n_out && B > 0 && T > 0 && V > 0 && ldl >= V && rows_per_b >= T && max_n > 0,
                  "aptai_ctc_greedy_decode: bad arguments");
    APTAI_LAUNCH(ctc_greedy_decode_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, logits, (long)ldl, (long)rows_per_b, (int)T,
                 (int)V, blank, ids_out, (int)max_n, n_out);
    APTAI_CHECK_LAUNCH("ctc_greedy_decode_kernel");
    return APTAI_OK;
}
