// 256 x 192 x 64 bf16 MFMA GEMM tile for the [8192] x {2304, 3072} outputs of an encoder layer (q|k|v projection, FFN1 forward, FFN2 data
// gradient; wav2vec2-large: [4096] x 3072) - round 4.
//
// Why this tile.  Those outputs are 384 / 288 tiles of 256 x 256 (1.5 / 1.125 rounds of the 256 CUs) and ran on 128-row tiles at
// 0.34 of the MFMA rate in the loop.  Bytes per flop: a 256 x 192 tile stages 56 KB per 6.3 MFLOP (8.9 B/kflop) against 12.9
// (128 x 192) and 15.2 (128 x 128); [8192] x 3072 is exactly two rounds of it (512 tiles), [4096] x 3072 (large) exactly one.
//
// Structure (the 256 x 256 kernel's ingredients - two wave groups one barrier apart, LDS-DMA stages, counted vmcnt, raw s_barrier - with
// ONE phase per K-tile):
//   * 8 waves = 4 (M) x 2 (N); wave tile 64 x 96 = 4 x 6 MFMA tiles (96 accumulator registers).  Wave (wr, wc) owns rows
//     wr * 64 .. + 64 and columns wc * 96 .. + 96.
//   * per K-tile a wave runs ONE load section - its 8 A + 12 B fragment reads, then the LDS-DMA of its group's half of A(t + 1) (4 pieces
//     per thread) and of B(t + 2) (3), `s_waitcnt vmcnt(7)` (B(t + 1) landed), lgkmcnt(0), barrier - and ONE MFMA section - 48 MFMAs,
//     `s_waitcnt vmcnt(3)` (A(t + 1) landed, B(t + 2) stays in flight), barrier.  Never vmcnt(0) in the loop.  The two waves of a SIMD
//     (w, w + 4) sit in different groups, group 1 one barrier behind group 0: while one issues its 48 MFMAs (~800 cycles) the other runs
//     its load section.
//   * A is double-buffered (2 x 32 KB) and PRIVATE to a group: group g stages and reads only rows g * 128 .. of the tile, so its waits
//     alone order its hazards and A(t + 1) may land as late as the end of the group's own MFMA section t.  B is read by both groups
//     and triple-buffered (3 x 24 KB; K-major B: 3 x 32 KB): B(t + 2) has two K-tiles to land.  136 KB (NT) / 160 KB (NN).
//   Why one phase.  In-kernel stamps (tools/gemm_t4_stamps.py, profiles/r04_gemm_t4_stamps.txt) on two-phase forms of this loop (24 MFMAs
//   per section) showed load sections of 500-600 cycles against 430 for the partner's MFMAs: a section's cost is mostly LATENCY
//   (fragment reads issue -> data ~300 cycles under the LDS-DMA write traffic, ~40-50 cycles per DMA piece) and hardly shrinks with
//   its size, so halving the number of sections per K-tile balances load against MFMA time (and halves the barriers).
//   Hazards (arrival numbers of the workgroup barrier: group 0 ends load section t with arrival 2 + 2t and MFMA section t with
//   3 + 2t; group 1 one later).  RAW B: B(t + 1) is retired by every wave in load section t, in front of arrival 2 + 2t / 3 + 2t; its
//   first read is group 0's load section t + 1, behind arrival 3 + 2t.  RAW A (own group only): retired in front of the barrier that
//   ends the group's MFMA section t, read behind it.  WAR: B(t + 2) overwrites B(t - 1), whose reads completed (lgkmcnt(0) sits in
//   FRONT of the barrier) before arrival 2t / 2t + 1; issued behind arrival 2t + 1 / 2t + 2.  A(t + 1) overwrites the group's own
//   A(t - 1), complete before the barrier that ended its load section t - 1.
//
// K-major B (data gradients, dX = dY W): the B image is two sub-images of 64 k-rows x 256 B (columns 0..127 and 128..191, the
// second half empty) so that the transposing reads and their swizzle are exactly the 256-row kernel's (ds_read_b64_tr_b16 through
// inline asm, counted by hand: csrc/gemm_common.h).
//
// Epilogue: two passes of 128 rows x 192 columns of fp32 through the free staging LDS (pitch 784 B: conflict-free for the
// 16-row accumulator writes), every thread 6 chunks of 8 outputs per pass through the shared per-flag-word epilogue
// (gemm_common.h), residual / aux rows prefetched one group ahead.
#include "gemm_common.h"

using namespace aptai_gemm;

namespace {

constexpr int T4_THREADS = 512;
constexpr int T4_BM = 256, T4_BN = 192;
constexpr int T4_APART = 128 * BK * 2;                  // 16 KiB: one group's half of the A tile
constexpr int T4_ABUF = 2 * T4_APART;                   // 32 KiB
constexpr int T4_EPI_PITCH = T4_BN * 4 + 16;            // 784 B
template <bool B_KM> constexpr int t4_bbuf() { return B_KM ? 2 * 64 * 256 : T4_BN * BK * 2; }       // 32 KiB / 24 KiB
template <bool B_KM> constexpr int t4_smem() { return 2 * T4_ABUF + 3 * t4_bbuf<B_KM>(); }          // 136 KiB / 160 KiB
static_assert(128 * T4_EPI_PITCH <= t4_smem<false>(), "an epilogue pass must fit the staging LDS");

#ifdef APTAI_T4_STAMPS
// development only (tools/ab builds, -DAPTAI_T4_STAMPS; never the product library): shader-clock time of the 7 segments of a K-tile
// (reads issued + landed | DMA issue | vmcnt + lgkmcnt wait | barrier | MFMA issue | vmcnt wait | barrier), summed over the K-tiles, for wave 0
// (group 0) and wave 4 (group 1) of every block; read back by tools/gemm_t4_stamps.py.  The stamp is the guide's one-statement form.
__device__ unsigned long long g_t4_seg[1024 * 2 * 16];
#define T4_SEG_INIT()                                                                                   \
    unsigned long long seg_[16], tp_, rt0_;                                                             \
    for (int q_ = 0; q_ < 16; ++q_) seg_[q_] = 0;                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt0_)::"memory");                     \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tp_)::"memory");                          \
    const unsigned long long t0_ = tp_;                                                                 \
    __builtin_amdgcn_sched_barrier(0)
#define T4_SEG(i)                                                                                       \
    do {                                                                                                \
        unsigned long long tn_;                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tn_)::"memory");                      \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        seg_[i] += tn_ - tp_;                                                                           \
        tp_ = tn_;                                                                                      \
    } while (0)
#define T4_SEG_DONE()                                                                                   \
    do {                                                                                                \
        unsigned long long rt1_;                                                                        \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt1_)::"memory");                 \
        seg_[14] = tp_ - t0_;            /* shader cycles over the loop */                              \
        seg_[15] = rt1_ - rt0_;          /* 100 MHz ticks over the loop */                              \
        if ((threadIdx.x & 255) == 0 && blockIdx.x < 1024)                                              \
            for (int q_ = 0; q_ < 16; ++q_) g_t4_seg[(blockIdx.x * 2 + (threadIdx.x >> 8)) * 16 + q_] = seg_[q_]; \
    } while (0)
extern "C" int aptai_debug_read_t4_stamps(void* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_t4_seg), sizeof(g_t4_seg)) == hipSuccess ? 0 : 1;
}
#else
#define T4_SEG_INIT() do {} while (0)
#define T4_SEG(i) do {} while (0)
#define T4_SEG_DONE() do {} while (0)
#endif

// ---- main loop over K-tiles [kt_begin, kt_begin + nk).  Leaves every wave behind a workgroup barrier, no LDS-DMA outstanding.
template <bool B_KM>
__device__ __forceinline__ void t4_mainloop(const GemmArgs& g, char* smem, const int m0, const int n0, const int kt_begin, const int nk,
                                            f32x4 (&acc)[4][6], const int tid, const int lane, const int wave, const int wr,
                                            const int wc) {
    constexpr int BBUF = t4_bbuf<B_KM>();
    constexpr int NB_IT = B_KM ? 4 : 3;                 // B staging instructions per thread and K-tile
    char* const sA = smem;
    char* const sB = smem + 2 * T4_ABUF;
    const int grp = wave >> 2;                          // wave group = A half
    // ---- staging sources (advanced by one K-tile per call)
    const bf16_t* pa[4];
    const bf16_t* pb[NB_IT];
    bool pb_on[NB_IT];
    {
        const int tg = tid & 255;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int cid = it * 256 + tg;
            const int row = cid >> 3, pc = cid & 7;
            int grow = m0 + grp * 128 + row;
            grow = grow < g.M ? grow : g.M - 1;
            pa[it] = g.A + (long)grow * g.lda + (long)kt_begin * BK + ((pc ^ (row & 7)) << 3);
        }
#pragma unroll
        for (int it = 0; it < NB_IT; ++it) {
            const int cid = it * T4_THREADS + tid;
            if (!B_KM) {
                const int row = cid >> 3, pc = cid & 7;
                int grow = n0 + row;
                grow = grow < g.N ? grow : g.N - 1;
                pb[it] = g.B + (long)grow * g.ldb + (long)kt_begin * BK + ((pc ^ (row & 7)) << 3);
                pb_on[it] = true;
            } else {
                const int sub = it >> 1;                // sub-image: columns sub * 128 ..
                const int c2 = cid & 1023;
                const int krow = c2 >> 4, pc = c2 & 15;
                const int lc = pc ^ km_swz(krow);       // logical 8-column chunk held at LDS position pc of this k-row
                int col = n0 + sub * 128 + (lc << 3);
                pb_on[it] = sub == 0 || lc < 8;         // the second sub-image holds columns 128..191 only
                col = col <= g.N - 8 ? col : g.N - 8;
                pb[it] = g.B + ((long)kt_begin * BK + krow) * g.ldb + col;
            }
        }
    }
    const long stepB = B_KM ? (long)BK * g.ldb : (long)BK;
    auto stage_a = [&](int t) {                         // own half of A(t) -> A buffer t % 2
        char* dA = sA + (t & 1) * T4_ABUF + grp * T4_APART + (wave & 3) * 1024;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            __builtin_amdgcn_global_load_lds(GLB_PTR(pa[it]), LDS_PTR(dA + it * 4096), 16, 0, 0);
            pa[it] += BK;
        }
    };
    auto stage_b = [&](int t) {                         // B(t) -> B buffer t % 3
        char* dB = sB + (t % 3) * BBUF + wave * 1024;
#pragma unroll
        for (int it = 0; it < NB_IT; ++it) {
            // (K-major B: a lane whose chunk lies beyond column 191 issues nothing - every wave keeps 32 active lanes, so the
            //  instruction still counts once per wave)
            if (!B_KM || pb_on[it]) __builtin_amdgcn_global_load_lds(GLB_PTR(pb[it]), LDS_PTR(dB + it * 8192), 16, 0, 0);
            pb[it] += stepB;
        }
    };
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- prologue: A(0), B(0), B(1) in flight, the first two retired
    stage_a(0);
    stage_b(0);
    if (nk > 1) {
        stage_b(1);
        if (B_KM) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();         // group 1 runs one barrier behind group 0

    Frag<false> af[4][2];
    Frag<B_KM> bf[6][2];
    const int arow = (wr & 1) * 64;                     // this wave's rows inside its group's A half
    T4_SEG_INIT();
    for (int t = 0; t < nk; ++t) {
        const char* tA = sA + (t & 1) * T4_ABUF + grp * T4_APART;
        const char* tB = sB + (t % 3) * BBUF;
        // ---------------- load section
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int jt = wc * 6 + j;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (!B_KM) bf[j][ks].read(tB, jt * 16, ks, lane);
                else bf[j][ks].read(tB + (jt >> 3) * (64 * 256), (jt & 7) * 16, ks, lane);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) af[i][ks].read(tA, arow + i * 16, ks, lane);
        __builtin_amdgcn_sched_barrier(0);
        T4_SEG(0);
        if (t + 2 < nk) {
            stage_a(t + 1);
            stage_b(t + 2);
            T4_SEG(1);
            if (B_KM) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // B(t + 1) landed; A(t + 1), B(t + 2) in flight
            else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        } else if (t + 1 < nk) {
            stage_a(t + 1);
            T4_SEG(1);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                  // B(t + 1) landed; A(t + 1) in flight
        } else {
            T4_SEG(1);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // in FRONT of the barrier: a buffer is free once every wave has passed it
        T4_SEG(2);
        __builtin_amdgcn_s_barrier();
        T4_SEG(3);
        __builtin_amdgcn_sched_barrier(0);
        // ---------------- MFMA section: 64 rows x 96 columns x K = 64
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j][ks].get(), af[i][ks].get(), acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        // the MFMAs are register-only: without a use that is ordered against the barrier the compiler may sink them into the next
        // load section (it did in a stamped build) - an empty volatile statement on the accumulators pins them here
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) asm volatile("" : "+v"(acc[i][j]));
        __builtin_amdgcn_sched_barrier(0);
        T4_SEG(4);
        if (t + 2 < nk) {
            if (B_KM) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");       // the group's A(t + 1) landed; B(t + 2) stays in flight
            else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        } else if (t + 1 < nk) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        T4_SEG(5);
        __builtin_amdgcn_s_barrier();
        T4_SEG(6);
    }
    T4_SEG_DONE();
    if (grp == 0) __builtin_amdgcn_s_barrier();         // re-align the two groups
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// ---- epilogue; ends behind a barrier
template <int FM>
__device__ __forceinline__ void t4_epilogue_body(const GemmArgs& g, char* smem, const int m0, const int n0, const f32x4 (&acc)[4][6],
                                                 const int tid, const int lane, const int wr, const int wc, const int flags_rt) {
    const int flags = FM >= 0 ? FM : flags_rt;
    // chunk ids of this thread in a pass: tid + 512 k (k = 0..5) -> row = id / 24, column chunk = id % 24 = (c0 + 8 k) % 24
    const int c0 = tid % 24, r0 = tid / 24;
    int cc[3], rr[6];
#pragma unroll
    for (int s = 0; s < 3; ++s) cc[s] = (c0 + 8 * s) % 24;
#pragma unroll
    for (int k = 0; k < 6; ++k) rr[k] = (tid + 512 * k) / 24;
    (void)r0;
    float bias8[3][8];
    bool n_ok[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int n = n0 + cc[s] * 8;
        n_ok[s] = n < g.N;
#pragma unroll
        for (int r = 0; r < 8; ++r) bias8[s][r] = 0.f;
        if ((flags & APTAI_EPI_BIAS) && n_ok[s]) {
            const f32x4 b0 = *(const f32x4*)(g.bias + n), b1 = *(const f32x4*)(g.bias + n + 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) { bias8[s][r] = b0[r]; bias8[s][4 + r] = b1[r]; }
        }
    }
    const float alpha = (flags & APTAI_EPI_ALPHA) ? g.alpha : 1.0f;
    uint32_t sd0 = g.seed0, sd1 = g.seed1;
    if (flags & APTAI_EPI_DROPOUT) apply_salt(g.salt, sd0, sd1);
    const bool want_res = (flags & APTAI_EPI_RESIDUAL) != 0, want_aux = (flags & (APTAI_EPI_DGELU | APTAI_EPI_MUL_AUX)) != 0;
    constexpr bool HEAVIEST = FM < 0 || ((FM & APTAI_EPI_GELU) && (FM & (APTAI_EPI_DROPOUT | APTAI_EPI_PRE_DGELU)));
    constexpr int GS = HEAVIEST ? 1 : 3, NG = 6 / GS;   // chunks per group, groups per pass
    u32x4 resv[2][GS], auxv[2][GS];
    auto prefetch = [&](int slot, int pq, int grp) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < GS; ++k) {
            const int kk = grp * GS + k;
            const int m = m0 + pq * 128 + rr[kk], n = n0 + cc[kk % 3] * 8;
            const bool ok = n_ok[kk % 3] && m < g.M;
            u32x4 rq = {0u, 0u, 0u, 0u}, aq = {0u, 0u, 0u, 0u};
            if (ok && want_res) rq = *(const u32x4*)(g.residual + (long)m * g.ldr + n);
            if (ok && want_aux) aq = *(const u32x4*)(g.aux + (long)m * g.ldaux + n);
            resv[slot][k] = rq;
            auxv[slot][k] = aq;
        }
    };
    if (want_res || want_aux) prefetch(0, 0, 0);
#pragma unroll
    for (int pq = 0; pq < 2; ++pq) {
        if ((wr >> 1) == pq) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const int lr = (wr & 1) * 64 + i * 16 + (lane & 15);
                    const int c16 = (wc * 6 + j) * 4 + (lane >> 4);
                    *(f32x4*)(smem + lr * T4_EPI_PITCH + c16 * 16) = acc[i][j];
                }
        }
        __syncthreads();
#pragma unroll
        for (int grp = 0; grp < NG; ++grp) {
            const int q = pq * NG + grp, cur = q & 1;
            if ((want_res || want_aux) && q + 1 < 2 * NG) prefetch(cur ^ 1, (q + 1) / NG, (q + 1) % NG);
            f32x4 v0[GS], v1[GS];
#pragma unroll
            for (int k = 0; k < GS; ++k) {
                const int kk = grp * GS + k;
                const char* p = smem + rr[kk] * T4_EPI_PITCH + cc[kk % 3] * 32;
                v0[k] = *(const f32x4*)p;
                v1[k] = *(const f32x4*)(p + 16);
            }
#pragma unroll
            for (int k = 0; k < GS; ++k) {
                const int kk = grp * GS + k, s = kk % 3;
                const int m = m0 + pq * 128 + rr[kk], n = n0 + cc[s] * 8;
                if (m >= g.M || !n_ok[s]) continue;
                float v[8] = {v0[k][0], v0[k][1], v0[k][2], v0[k][3], v1[k][0], v1[k][1], v1[k][2], v1[k][3]};
                if (flags & (APTAI_EPI_BIAS | APTAI_EPI_ALPHA)) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] = fmaf(v[r], alpha, bias8[s][r]);
                }
                epilogue_chunk<FM>(v, g, flags, (long)m, n, auxv[cur][k], resv[cur][k], sd0, sd1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
}

template <bool B_KM>
__global__ __launch_bounds__(T4_THREADS, 2) void gemm_t4_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;            // waves w and w + 4 share a SIMD: rows 0..127 (group 0) / 128..255 (group 1)
    const int nwg = g.tiles_m * g.tiles_n;
    const int bid = xcd_remap(blockIdx.x, nwg);
    int tile_m, tile_n;
    raster2d(bid, g.tiles_m, g.tiles_n, g.raster_gm, tile_m, tile_n);
    const int m0 = tile_m * T4_BM, n0 = tile_n * T4_BN;
    f32x4 acc[4][6];                                    // [i][j]: rows wr * 64 + i * 16, columns wc * 96 + j * 16
    t4_mainloop<B_KM>(g, smem, m0, n0, 0, g.K / BK, acc, tid, lane, wave, wr, wc);
    // (indices through opaque copies: keeps the specialised epilogue bodies' address arithmetic out of the main loop, see gemm.hip)
    int tid_e = tid, lane_e = lane, m0_e = __builtin_amdgcn_readfirstlane(m0), n0_e = __builtin_amdgcn_readfirstlane(n0);
    asm volatile("" : "+v"(tid_e), "+v"(lane_e), "+s"(m0_e), "+s"(n0_e));
    const int fx = epi_flag_word(g);
    epi_dispatch(fx, [&](auto w) { t4_epilogue_body<decltype(w)::value>(g, smem, m0_e, n0_e, acc, tid_e, lane_e, wr, wc, fx); });
}

}  // namespace

namespace aptai_gemm {

int launch_gemm_t4(GemmArgs g, bool b_km, hipStream_t stream) {
    g.tiles_m = (int)((g.M + T4_BM - 1) / T4_BM);
    g.tiles_n = (int)((g.N + T4_BN - 1) / T4_BN);
    dim3 grid(g.tiles_m * g.tiles_n, 1, 1);
    if (!b_km) {
        auto kern = gemm_t4_kernel<false>;
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, t4_smem<false>());
            attr_set = true;
        }
        APTAI_LAUNCH(kern, grid, dim3(T4_THREADS), t4_smem<false>(), stream, g);
    } else {
        auto kern = gemm_t4_kernel<true>;
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, t4_smem<true>());
            attr_set = true;
        }
        APTAI_LAUNCH(kern, grid, dim3(T4_THREADS), t4_smem<true>(), stream, g);
    }
    APTAI_CHECK_LAUNCH("gemm_t4_kernel");
    return APTAI_OK;
}

}  // namespace aptai_gemm
