// bf16 MFMA GEMM family for the wav2vec2 hot path (gfx950).
//
//   C[M,N] = rowop[M,K] . colop[N,K]^T     (fp32 accumulate on v_mfma_f32_16x16x32_bf16)
//
// Each operand is either K-contiguous in memory ([rows][K], "KC") or K-major ([K][rows], "KM"):
//   NT  A=KC  B=KC : Linear forward  Y = X W^T            (HF nn.Linear call sites),
//                    strided-row implicit GEMM for conv L1..L6 (K4: lda = stride*C < K = k*C)
//   NN  A=KC  B=KM : dgrad          dX = dY W
//   TN  A=KM  B=KM : wgrad          dW = dY^T X           (fp32 out, optional split-K slabs)
//
// Tile 128x128x64, 256 threads = 4 waves (2x2), wave tile 64x64 = 4x4 MFMA tiles.  Tiles are staged
// global->LDS by LDS-DMA (global_load_lds_dwordx4, 16 B/lane) into two stages; the XOR swizzle sits on the
// per-lane SOURCE address and on the LDS read (LDS-DMA writes are lane-linear).  KC fragments are read with
// ds_read_b128, KM fragments with ds_read_b64_tr_b16 (hardware transpose).  MFMA operands are swapped
// (D = colfrag x rowfrag) so each lane ends up with 4 consecutive output columns -> 8/16-byte stores.
#include <stdlib.h>

#include <mutex>
#include <set>

#include "common.h"
#include "gemm_common.h"

using namespace aptai_gemm;

namespace {

#ifdef APTAI_STAMPS
// development only (tools/ab builds, -DAPTAI_STAMPS): per-block time stamps of the 256 x 256 kernel, read back by tools/gemm256_stamps.py
__device__ unsigned long long g_stamps[4096 * 8];
#define APTAI_STAMP(slot)                                                                          \
    do {                                                                                           \
        if (threadIdx.x == 0 && blockIdx.x < 4096 && blockIdx.y == 0 && blockIdx.z == 0)           \
            g_stamps[blockIdx.x * 8 + (slot)] = wall_clock64();                                    \
    } while (0)
extern "C" int aptai_debug_read_stamps(void* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), sizeof(g_stamps)) == hipSuccess ? 0 : 1;
}
#else
#define APTAI_STAMP(slot) do {} while (0)
#endif

// host side, APTAI_EPI_TRACE=1: report each flag word of a bf16-output launch once, and whether it has a compiled body
void epi_trace(const GemmArgs& g, int tile) {
    static const bool on = getenv("APTAI_EPI_TRACE") != nullptr;
    if (!on) return;
    static std::mutex mu;
    static std::set<int> seen;
    const int w = g.flags | (g.out_pre ? EPX_PRE : 0) | (g.colscale_n > 0 ? EPX_CS : 0);
    std::lock_guard<std::mutex> lk(mu);
    if (!seen.insert(w).second) return;
    bool listed = false;
#define APTAI_EPI_CASE(W) listed = listed || (w == (W));
    APTAI_EPI_WORDS(APTAI_EPI_CASE)
#undef APTAI_EPI_CASE
    fprintf(stderr, "[aptai epi] flag word 0x%x (tile %d, %d x %d x %d): %s\n", w, tile, g.M, g.N, g.K, listed ? "compiled" : "RUN-TIME form");
}

// one BM_T x 128 output tile: `bid` is the (already remapped) tile index of problem g, `batch` < 0 = not batched.
// BM_T = 128: wave tile 64 x 64, 2 blocks per CU.  BM_T = 64 (K-contiguous A only): wave tile 32 x 64, 24 KiB stages, 3 blocks
// per CU = 768 slots - for [8192] x 768 outputs, whose 384 big tiles fill only 0.75 of one round of 512 slots while their 768
// half tiles are exactly one round of 768.
template <bool A_KM, bool B_KM, bool OUT_F32, int BM_T = 128>
__device__ __forceinline__ void gemm_tile_body(GemmArgs g, const int bid, const int batch, const int split) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    APTAI_STAMP(0);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    int tile_m, tile_n;
    raster2d(bid, g.tiles_m, g.tiles_n, g.raster_gm, tile_m, tile_n);
    constexpr int NI = BM_T / 32;                       // 16-row MFMA tiles per wave
    constexpr int WM = BM_T / 2;                        // wave tile height
    constexpr int NA_IT = BM_T / 32;                    // staging instructions per thread for the A tile
    constexpr int A_BYTES_T = BM_T * BK * 2;
    constexpr int STAGE_T = A_BYTES_T + BN * BK * 2;
    constexpr int NPASS = BM_T / 16;                    // epilogue passes of 16 rows
    static_assert(BM_T == 128 || (BM_T == 64 && !A_KM), "64-row tiles are built for K-contiguous A only");
    constexpr bool RING5 = A_KM && B_KM && BM_T == 128 && APTAI_GEMM_RING5;
    const int m0 = tile_m * BM_T, n0 = tile_n * BN;

    if (batch >= 0) {
        const int bo = batch / g.nb_inner, bi = batch % g.nb_inner;
        g.A += bo * g.sA[0] + bi * g.sA[1];
        g.B += bo * g.sB[0] + bi * g.sB[1];
        const long co = bo * g.sC[0] + bi * g.sC[1];
        g.C = (OUT_F32 && !(g.flags & APTAI_EPI_SPLIT_OUT)) ? (void*)((float*)g.C + co) : (void*)((bf16_t*)g.C + co);
        if (g.out_pre) g.out_pre += co;
        if (g.bias) g.bias += bo * g.sBias[0] + bi * g.sBias[1];
        if (g.residual) g.residual += bo * g.sR[0] + bi * g.sR[1];
        if (g.aux) g.aux += bo * g.sAux[0] + bi * g.sAux[1];
    }
    const int total_kt = g.K / BK;
    const int kt_begin = split * g.ktiles_per_split;
    int kt_end = kt_begin + g.ktiles_per_split;
    kt_end = kt_end < total_kt ? kt_end : total_kt;
    const int nk = kt_end - kt_begin;

    f32x4 acc[NI][4];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int wave_base_tid = wave * 64;
    const bool wave_active = (m0 + wm * WM < g.M) && (n0 + wn * 64 < g.N);
    const bf16_t* pa[NA_IT];
    const bf16_t* pb[4];
    const long stepA = A_KM ? (long)BK * g.lda : (long)BK, stepB = B_KM ? (long)BK * g.ldb : (long)BK;
#pragma unroll
    for (int it = 0; it < NA_IT; ++it) pa[it] = stage_src<A_KM>(g.A, g.lda, m0, g.M, it, tid) + kt_begin * stepA;
#pragma unroll
    for (int it = 0; it < 4; ++it) pb[it] = stage_src<B_KM>(g.B, g.ldb, n0, g.N, it, tid) + kt_begin * stepB;
    auto stage = [&](char* buf) {                     // K-tiles are staged in order: every call advances the sources
        char* dst = buf + wave_base_tid * 16;          // wave-uniform; the hardware adds lane * 16
#pragma unroll
        for (int it = 0; it < NA_IT; ++it) {
            __builtin_amdgcn_global_load_lds(GLB_PTR(pa[it]), LDS_PTR(dst + it * (NTHREADS * 16)), 16, 0, 0);
            pa[it] += stepA;
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            __builtin_amdgcn_global_load_lds(GLB_PTR(pb[it]), LDS_PTR(dst + A_BYTES_T + it * (NTHREADS * 16)), 16, 0, 0);
            pb[it] += stepB;
        }
    };
    if constexpr (RING5) {
        // Both operands K-major (weight gradients, K = all frames of the batch): the loop is bound by operand delivery, not by
        // the MFMA pipe (with cache-resident operands the grouped launch runs 33 % faster), and delivery per CU is
        // bytes-in-flight / latency.  The K-tile is staged as two 32-row half-stages of 16 KiB through a ring of five
        // (80 KiB per block, two blocks per CU = all of the LDS): three half-stages = 48 KiB per block stay in flight under
        // the MFMAs instead of one 32-KiB stage, still one barrier per 64 k.  Half h lives in slot h % 5.
        //   iteration t: wait for halves <= 2t+1 (counted vmcnt: 4 loads per thread and half-stage, only half 2t+2 may be
        //   outstanding), barrier, issue halves 2t+3 and 2t+4 into the slots of halves 2t-2 and 2t-1 (every wave is past its
        //   reads of those: it entered the barrier with lgkmcnt(0)), then the MFMAs of halves 2t and 2t+1.
        constexpr int HALF_T = RING_HALF_BYTES;
        const bf16_t* ha[2];
        const bf16_t* hb[2];
        const long hstepA = 32 * (long)g.lda, hstepB = 32 * (long)g.ldb;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            ha[it] = stage_src<true>(g.A, g.lda, m0, g.M, it, tid) + kt_begin * stepA;
            hb[it] = stage_src<true>(g.B, g.ldb, n0, g.N, it, tid) + kt_begin * stepB;
        }
        auto stage_half = [&](int slot) {
            char* dst = smem + slot * HALF_T + wave_base_tid * 16;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                __builtin_amdgcn_global_load_lds(GLB_PTR(ha[it]), LDS_PTR(dst + it * (NTHREADS * 16)), 16, 0, 0);
                ha[it] += hstepA;
            }
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                __builtin_amdgcn_global_load_lds(GLB_PTR(hb[it]), LDS_PTR(dst + HALF_T / 2 + it * (NTHREADS * 16)), 16, 0, 0);
                hb[it] += hstepB;
            }
        };
        // per-lane byte offsets of the 16 transposing reads of one half-stage (read_frag<true> with ks = 0): A fragments
        // 0..3 (lo, hi) in [0, 8 KiB), B fragments in [8 KiB, 16 KiB)
        uint32_t frag_off[16];
        {
            const int gq = lane >> 4, li = lane & 15, qq = li >> 2, pp = li & 3;
            const int k_lo = gq * 8 + qq, k_hi = k_lo + 4;
            const int sub = (pp & 1) << 3;
#pragma unroll
            for (int f = 0; f < 8; ++f) {
                const int is_b = f >> 2, idx = f & 3;
                const int row_base = is_b ? wn * 64 + idx * 16 : wm * WM + idx * 16;
                const int ch = (row_base >> 3) + (pp >> 1);
                frag_off[2 * f] = (uint32_t)(is_b * (HALF_T / 2) + k_lo * 256 + ((ch ^ km_swz(k_lo)) << 4) + sub);
                frag_off[2 * f + 1] = (uint32_t)(is_b * (HALF_T / 2) + k_hi * 256 + ((ch ^ km_swz(k_hi)) << 4) + sub);
            }
        }
        const uint32_t ring_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)smem;
        const int nh = 2 * nk;
        if (nh > 0) { stage_half(0); stage_half(1); }
        if (nh > 2) stage_half(2);
        int s0 = 0;                                        // slot of half 2t
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const int s1 = s0 + 1 >= 5 ? s0 - 4 : s0 + 1, s2 = s0 + 2 >= 5 ? s0 - 3 : s0 + 2, s3 = s0 + 3 >= 5 ? s0 - 2 : s0 + 3,
                      s4 = s0 + 4 >= 5 ? s0 - 1 : s0 + 4;
            if (2 * kt + 3 < nh) stage_half(s3);
            if (2 * kt + 4 < nh) stage_half(s4);
            if (wave_active) {
                // The transposing reads go through inline asm: behind an LDS-DMA the compiler puts `s_waitcnt vmcnt(0)` in
                // front of every ds_read_tr builtin (it cannot tell the slots apart), which would drain the ring at each
                // K-tile.  So the waits are written here: all 32 reads of the K-tile are issued, the MFMAs of the first half
                // start once at most 15 are outstanding (LDS reads return in order: the first 17, i.e. the first half's 16,
                // are back), those of the second half at lgkmcnt(0).  The "+v" operands tie the waits to the fragments.
                short4v fr[2][16];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const uint32_t base = ring_base + (uint32_t)((hh ? s1 : s0) * HALF_T);
#pragma unroll
                    for (int f = 0; f < 16; ++f)
                        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fr[hh][f]) : "v"(base + frag_off[f]));
                }
#define APTAI_TIE8(h, o) "+v"(fr[h][o]), "+v"(fr[h][o + 1]), "+v"(fr[h][o + 2]), "+v"(fr[h][o + 3]), "+v"(fr[h][o + 4]), \
                         "+v"(fr[h][o + 5]), "+v"(fr[h][o + 6]), "+v"(fr[h][o + 7])
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    if (hh == 0) asm volatile("s_waitcnt lgkmcnt(15)" : APTAI_TIE8(0, 0), APTAI_TIE8(0, 8));
                    else asm volatile("s_waitcnt lgkmcnt(0)" : APTAI_TIE8(1, 0), APTAI_TIE8(1, 8));    // (pinning the first half's MFMAs in front of this wait: 10.22 vs 10.19 ms/step)
                    bf16x8 af[NI], bfr[4];
#pragma unroll
                    for (int i = 0; i < NI; ++i) {
                        const short4v lo4 = fr[hh][2 * i], hi4 = fr[hh][2 * i + 1];
                        af[i] = __builtin_bit_cast(bf16x8, (short8v){lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]});
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const short4v lo4 = fr[hh][8 + 2 * j], hi4 = fr[hh][8 + 2 * j + 1];
                        bfr[j] = __builtin_bit_cast(bf16x8, (short8v){lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]});
                    }
#pragma unroll
                    for (int i = 0; i < NI; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
                }
#undef APTAI_TIE8
            }
            s0 = s2;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();                                   // every wave is done with the ring before the epilogue reuses it
    } else {
    if (nk > 0) stage(smem);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(smem + (cur ^ 1) * STAGE_T);
        const char* sa = smem + cur * STAGE_T;
        const char* sb = sa + A_BYTES_T;
        // a wave whose 64 x 64 quadrant lies wholly outside the problem (grouped positional conv: 48 channels per group;
        // bias-gradient problems: M = 8; heads: N = 64) stages and synchronises but issues no LDS reads and no MFMAs
        if (wave_active) {
            if constexpr (!A_KM && !B_KM && APTAI_GEMM_HOIST) {
                // K-contiguous operands: all 16 ds_read_b128 of the K-tile are issued before the first MFMA (sched_barrier
                // keeps the compiler from sinking them next to their uses): one exposed LDS latency per K-tile instead of
                // one per fragment group (+9 % at K = 3072).  With transposing reads (K-major operands) the same hoist
                // measured 4-7 % slower, so those variants keep the compiler's order.
                bf16x8 af[2][NI], bfr[2][4];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) bfr[ks][j] = read_frag<B_KM>(sb, wn * 64 + j * 16, ks, lane);
#pragma unroll
                    for (int i = 0; i < NI; ++i) af[ks][i] = read_frag<A_KM>(sa, wm * WM + i * 16, ks, lane);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < NI; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ks][j], af[ks][i], acc[i][j], 0, 0, 0);
            } else if constexpr (BM_T == 64 && !A_KM && B_KM && APTAI_GEMM_M64_ASM) {
                // 64-row NN tiles (the dgrads): every LDS read of the K-tile through inline asm, issued up front in the order
                // B(k-half 0) x8, A(0) x2, B(1) x8, A(1) x2; LDS reads return in order, so lgkmcnt(10) = the first half is back
                short4v bl[2][4], bh[2][4];
                u32x4 aa[2][2];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) read_frag_tr_asm(sb, wn * 64 + j * 16, ks, lane, bl[ks][j], bh[ks][j]);
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int row = wm * WM + i * 16 + (lane & 15), q = ks * 4 + (lane >> 4);
                        const uint32_t addr = lds_u32(sa) + (uint32_t)(row * 128 + ((q ^ (row & 7)) << 4));
                        asm volatile("ds_read_b128 %0, %1" : "=v"(aa[ks][i]) : "v"(addr));
                    }
                }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    if (ks == 0)
                        asm volatile("s_waitcnt lgkmcnt(10)" : "+v"(bl[0][0]), "+v"(bl[0][1]), "+v"(bl[0][2]), "+v"(bl[0][3]), "+v"(bh[0][0]),
                                                               "+v"(bh[0][1]), "+v"(bh[0][2]), "+v"(bh[0][3]), "+v"(aa[0][0]), "+v"(aa[0][1]));
                    else
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bl[1][0]), "+v"(bl[1][1]), "+v"(bl[1][2]), "+v"(bl[1][3]), "+v"(bh[1][0]),
                                                              "+v"(bh[1][1]), "+v"(bh[1][2]), "+v"(bh[1][3]), "+v"(aa[1][0]), "+v"(aa[1][1]));
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(combine_tr(bl[ks][j], bh[ks][j]),
                                                                                __builtin_bit_cast(bf16x8, aa[ks][i]), acc[i][j], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bf16x8 af[NI], bfr[4];
                    // K-major fragments of the 128-row tiles arrive through read_frag_tr_asm (no compiler vmcnt(0) behind the
                    // staging loads: NN FFN2 dgrad 58 -> 55 us).  The 64-row NN tiles have their own all-asm branch above; with
                    // only the B reads in asm and one lgkmcnt(0) per k-half they lost to the builtin (57 vs 60 us).
                    constexpr bool ASM_TR = BM_T == 128;
                    short4v tl[NI + 4], th[NI + 4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if constexpr (B_KM && ASM_TR) read_frag_tr_asm(sb, wn * 64 + j * 16, ks, lane, tl[NI + j], th[NI + j]);
                        else bfr[j] = read_frag<B_KM>(sb, wn * 64 + j * 16, ks, lane);
                    }
#pragma unroll
                    for (int i = 0; i < NI; ++i) {
                        if constexpr (A_KM && ASM_TR) read_frag_tr_asm(sa, wm * WM + i * 16, ks, lane, tl[i], th[i]);
                        else af[i] = read_frag<A_KM>(sa, wm * WM + i * 16, ks, lane);
                    }
                    if constexpr (B_KM && ASM_TR) {
                        asm volatile("s_waitcnt lgkmcnt(0)"
                                     : "+v"(tl[NI]), "+v"(tl[NI + 1]), "+v"(tl[NI + 2]), "+v"(tl[NI + 3]), "+v"(th[NI]), "+v"(th[NI + 1]),
                                       "+v"(th[NI + 2]), "+v"(th[NI + 3]));
#pragma unroll
                        for (int j = 0; j < 4; ++j) bfr[j] = combine_tr(tl[NI + j], th[NI + j]);
                    }
                    if constexpr (A_KM && ASM_TR) {
                        static_assert(!A_KM || NI == 4, "K-major A is built for 128-row tiles");
                        asm volatile("s_waitcnt lgkmcnt(0)"
                                     : "+v"(tl[0]), "+v"(tl[1]), "+v"(tl[2]), "+v"(tl[3]), "+v"(th[0]), "+v"(th[1]), "+v"(th[2]), "+v"(th[3]));
#pragma unroll
                        for (int i = 0; i < NI; ++i) af[i] = combine_tr(tl[i], th[i]);
                    }
#pragma unroll
                    for (int i = 0; i < NI; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
                }
            }
        }
    }
    }

    // ------------------------------------------------------------------ epilogue
    // Phase 0: issue the residual / aux row loads of ALL 8 passes now (16 B each): their HBM latency hides under the
    //          LDS staging instead of serialising pass after pass.
    // Phase 1: accumulators -> fp32 LDS tile [128][128] (row pitch 528 B: conflict-free 16-B writes of 16 rows).
    // acc[i][j][r]: m = wm*64 + i*16 + (lane&15);  n = wn*64 + j*16 + (lane>>4)*4 + r
    // The body is compiled per flag word (epi_dispatch; block-uniform branch); thread / tile indices enter through opaque copies so
    // that the addresses of all those bodies are not hoisted above the main loop as loop invariants.
    APTAI_STAMP(2);
    int tid_e = tid, lane_e = lane, m0_e = __builtin_amdgcn_readfirstlane(m0), n0_e = __builtin_amdgcn_readfirstlane(n0);
    asm volatile("" : "+v"(tid_e), "+v"(lane_e), "+s"(m0_e), "+s"(n0_e));
    const int fx = epi_flag_word(g);
    auto body = [&](auto w) {
        constexpr int FM = decltype(w)::value;
        const int flags = FM >= 0 ? FM : fx;
        const int cl = (tid_e & 15) * 8;                     // column inside the tile, fixed per thread
        const int n = n0_e + cl;
        const bool n_ok = n < g.N;
        // residual / aux rows of all passes, issued ahead of the LDS staging - in the compiled bodies that read them; the run-time form
        // (a flag word nobody listed) loads them pass by pass in a rolled loop: it has to fit the register budget, not to be fast
        constexpr bool PREF = !OUT_F32 && FM >= 0 && (FM & (APTAI_EPI_RESIDUAL | APTAI_EPI_DGELU | APTAI_EPI_MUL_AUX)) != 0;
        constexpr int NPF = PREF ? NPASS : 1;
        u32x4 resv[NPF], auxv[NPF];
#pragma unroll
        for (int pass = 0; pass < NPF; ++pass) {
            const int m = m0_e + pass * 16 + (tid_e >> 4);
            const bool ok = PREF && n_ok && m < g.M;
            u32x4 rq = {0u, 0u, 0u, 0u}, aq = {0u, 0u, 0u, 0u};   // values first, one unconditional array store each (no stack arrays)
            if (ok && (flags & APTAI_EPI_RESIDUAL)) rq = *(const u32x4*)(g.residual + (long)m * g.ldr + n);
            if (ok && (flags & (APTAI_EPI_DGELU | APTAI_EPI_MUL_AUX))) aq = *(const u32x4*)(g.aux + (long)m * g.ldaux + n);
            resv[pass] = rq;
            auxv[pass] = aq;
        }
        __syncthreads();                                   // every wave is done reading the staging buffers
        {
            char* ct = smem;
    #pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int ml = wm * WM + i * 16 + (lane_e & 15);
    #pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int nl = wn * 64 + j * 16 + (lane_e >> 4) * 4;
                    *(f32x4*)(ct + ml * EPI_PITCH + nl * 4) = acc[i][j];
                }
            }
        }
        __syncthreads();
        // Phase 2: thread -> 8 consecutive columns of one row per pass (16 threads cover a 512-B row): every global
        // access of the epilogue (residual, aux, out_pre, C) is a coalesced 16/32-byte-per-lane row segment.
        float bias8[8];
    #pragma unroll
        for (int r = 0; r < 8; ++r) bias8[r] = 0.f;
        if ((flags & APTAI_EPI_BIAS) && n_ok) {
            const f32x4 b0 = *(const f32x4*)(g.bias + n), b1 = *(const f32x4*)(g.bias + n + 4);
    #pragma unroll
            for (int r = 0; r < 4; ++r) { bias8[r] = b0[r]; bias8[4 + r] = b1[r]; }
        }
        const float alpha = (flags & APTAI_EPI_ALPHA) ? g.alpha : 1.0f;
        uint32_t sd0 = g.seed0, sd1 = g.seed1;
        if (flags & APTAI_EPI_DROPOUT) apply_salt(g.salt, sd0, sd1);
        auto one_pass = [&](const int pass, const u32x4 auxq, const u32x4 resq) __attribute__((always_inline)) {
            const int ml = pass * 16 + (tid_e >> 4);
            const int m = m0_e + ml;
            if (m >= g.M || !n_ok) return;
            const f32x4 v0 = *(const f32x4*)(smem + ml * EPI_PITCH + cl * 4);
            const f32x4 v1 = *(const f32x4*)(smem + ml * EPI_PITCH + cl * 4 + 16);
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = (flags & (APTAI_EPI_BIAS | APTAI_EPI_ALPHA)) ? fmaf(v[r], alpha, bias8[r]) : v[r];
            if (OUT_F32) {
                if (flags & APTAI_EPI_BIAS_ROW) {
                    const float bm = g.bias[m];
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] += bm;
                }
                if (flags & APTAI_EPI_RESIDUAL_F32) {             // fp32 residual stream (inference-only encoder): += res32[m][n..n+7]
                    const float* R = (const float*)g.residual + (long)m * g.ldr + n;
                    const f32x4 r0 = *(const f32x4*)R, r1 = *(const f32x4*)(R + 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { v[r] += r0[r]; v[4 + r] += r1[r]; }
                }
                if (flags & APTAI_EPI_SPLIT_OUT) {               // exact-index mode: the result leaves as the next GEMM's split A operand
                    split_out_store(g, flags, v, (long)m, n);
                    return;
                }
                float* C = (float*)g.C + (long)split * g.slab_stride + (long)m * g.ldc + n;
                *(f32x4*)C = (f32x4){v[0], v[1], v[2], v[3]};
                *(f32x4*)(C + 4) = (f32x4){v[4], v[5], v[6], v[7]};
                return;
            }
            epilogue_chunk<FM>(v, g, flags, (long)m, n, auxq, resq, sd0, sd1);
        };
        if constexpr (!OUT_F32 && FM < 0) {
#pragma unroll 1
            for (int pass = 0; pass < NPASS; ++pass) {
                const int m = m0_e + pass * 16 + (tid_e >> 4);
                const bool ok = n_ok && m < g.M;
                u32x4 rq = {0u, 0u, 0u, 0u}, aq = {0u, 0u, 0u, 0u};
                if (ok && (flags & APTAI_EPI_RESIDUAL)) rq = *(const u32x4*)(g.residual + (long)m * g.ldr + n);
                if (ok && (flags & (APTAI_EPI_DGELU | APTAI_EPI_MUL_AUX))) aq = *(const u32x4*)(g.aux + (long)m * g.ldaux + n);
                one_pass(pass, aq, rq);
            }
        } else {
#pragma unroll
            for (int pass = 0; pass < NPASS; ++pass) {
                one_pass(pass, auxv[PREF ? pass : 0], resv[PREF ? pass : 0]);
                asm volatile("" ::: "memory");           // keep the passes apart: hoisting every pass's LDS reads costs the third block's
            }                                            // worth of registers (launch bounds of gemm_kernel)
        }
    };
    if (OUT_F32) body(EpiWord<-1>{});
    else epi_dispatch(fx, body);
    APTAI_STAMP(3);
#ifdef APTAI_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    APTAI_STAMP(4);
#endif
}

template <bool A_KM, bool B_KM, bool OUT_F32>
__global__ __launch_bounds__(NTHREADS, 3) void gemm_kernel(GemmArgs g) {
#ifdef APTAI_EXP_STAGGER
    if (g.exp_sleep > 0) {
        bool late = true;
        if (g.exp_cu_count != nullptr) {
            unsigned hwid, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            __shared__ unsigned arrival;
            if (threadIdx.x == 0) arrival = atomicAdd(g.exp_cu_count + ((xcc & 7u) * 256u + ((hwid >> 8) & 0xffu)), 1u);
            __syncthreads();
            late = arrival == 1u;        // the CU's second block of the FIRST round only: later blocks inherit the offset
        }
        if (late) for (int i = 0; i < g.exp_sleep; ++i) __builtin_amdgcn_s_sleep(31);   // ~1 us each
    }
#endif
    gemm_tile_body<A_KM, B_KM, OUT_F32>(g, xcd_remap(blockIdx.x, g.tiles_m * g.tiles_n), gridDim.y > 1 ? (int)blockIdx.y : -1,
                                        blockIdx.z);
}

constexpr int SMEM_M64_BYTES = 2 * (64 + BN) * BK * 2;      // 48 KiB >= the 64 x 128 fp32 epilogue tile (33 KiB): 3 blocks per CU
template <bool B_KM, bool OUT_F32>
__global__ __launch_bounds__(NTHREADS, 3) void gemm_kernel_m64(GemmArgs g) {
    gemm_tile_body<false, B_KM, OUT_F32, 64>(g, xcd_remap(blockIdx.x, g.tiles_m * g.tiles_n), gridDim.y > 1 ? (int)blockIdx.y : -1,
                                             blockIdx.z);
}

// Grouped launch: the tiles of up to MAX_GROUP independent problems of one operand layout in ONE grid.  Made for the four
// weight gradients of a transformer layer (dW = dY^T X, K = B*Tp = 8192 rows each): separately each is 36..144 tiles and
// needs split-K slabs plus a reduce launch to fill 256 CUs; together they are 432 full-K tiles = one round of the 512
// block slots, with no slabs, no reduce kernels and a 128-K-tile main loop per block.
constexpr int MAX_GROUP = 8;
struct GroupArgs {
    GemmArgs p[MAX_GROUP];
    int tile_end[MAX_GROUP];        // running sum of tiles_m * tiles_n
    int n, total;
};

template <bool A_KM, bool B_KM, bool OUT_F32>
__global__ __launch_bounds__(NTHREADS, 3) void gemm_grouped_kernel(GroupArgs ga) {
    const int bid = xcd_remap(blockIdx.x, ga.total);
    int pi = 0;
#pragma unroll
    for (int i = 1; i < MAX_GROUP; ++i)
        if (i < ga.n && bid >= ga.tile_end[i - 1]) pi = i;
    const int first = pi ? ga.tile_end[pi - 1] : 0;
    gemm_tile_body<A_KM, B_KM, OUT_F32>(ga.p[pi], bid - first, -1, 0);
}

// out[i] = (accumulate ? out[i] : 0) + sum_s slabs[s][i]
__global__ void splitk_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ out, long n4, long slab_stride4,
                                     int nsplit, int accumulate) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 s = accumulate ? ((const f32x4*)out)[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < nsplit; ++k) s += ((const f32x4*)slabs)[(long)k * slab_stride4 + i];
        ((f32x4*)out)[i] = s;
    }
}

template <bool B_KM, bool OUT_F32>
int launch_gemm_m64(GemmArgs g, int nbatch, int nsplit, hipStream_t stream) {
    auto kern = gemm_kernel_m64<B_KM, OUT_F32>;
    g.tiles_m = (int)ceil_div(g.M, 64);
    dim3 grid(g.tiles_m * g.tiles_n, nbatch, nsplit);
    APTAI_LAUNCH(kern, grid, dim3(NTHREADS), SMEM_M64_BYTES, stream, g);
    APTAI_CHECK_LAUNCH("gemm_kernel_m64");
    return APTAI_OK;
}

template <bool A_KM, bool B_KM, bool OUT_F32>
int launch_gemm(const GemmArgs& g, int nbatch, int nsplit, hipStream_t stream) {
    auto kern = gemm_kernel<A_KM, B_KM, OUT_F32>;
    constexpr int smem = smem_for<A_KM, B_KM>();
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_set = true;
    }
    dim3 grid(g.tiles_m * g.tiles_n, nbatch, nsplit);
#ifdef APTAI_EXP_STAGGER
    {   // development: per-call sleep and LDS request (occupancy) from the environment (tools/stagger_probe.py)
        GemmArgs ge = g;
        const char* e = getenv("APTAI_EXP_SLEEP");
        ge.exp_sleep = e ? atoi(e) : 0;
        const char* c = getenv("APTAI_EXP_CU_COUNT");          // device pointer (decimal) of 2048 zeroed words
        ge.exp_cu_count = c ? (unsigned*)(uintptr_t)strtoull(c, nullptr, 10) : nullptr;
        const char* m = getenv("APTAI_EXP_SMEM");
        const int sm = m ? atoi(m) : smem;
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, sm > smem ? sm : smem);
        APTAI_LAUNCH(kern, grid, dim3(NTHREADS), sm > smem ? sm : smem, stream, ge);
        APTAI_CHECK_LAUNCH("gemm_kernel");
        return APTAI_OK;
    }
#endif
    APTAI_LAUNCH(kern, grid, dim3(NTHREADS), smem, stream, g);
    APTAI_CHECK_LAUNCH("gemm_kernel");
    return APTAI_OK;
}


// =====================================================================================================================
// 256 x 256 x 64 tile, 512 threads = 8 waves (2 M-groups x 4 N-columns), one block per CU, 128 KiB of LDS.
//
// Schedule (after the "8-phase" idea of the CDNA4 playbook, re-derived for this kernel):
//  * a K-tile is consumed in 4 PHASES, one 64x32 quadrant of the wave's 128x64 output per phase (16 MFMAs):
//      p0: (qm0,qn0) reads A-half0 + B-half0 fragments   p1: (qm0,qn1) reads B-half1
//      p2: (qm1,qn1) reads A-half1                        p3: (qm1,qn0) reads nothing (B-half0 kept in registers)
//    so every 16-KiB half-tile {A0,B0,B1,A1} has a LAST ds_read phase (0,0,1,2) and is restaged by LDS-DMA exactly
//    two or more phases later: half-tile h (stream order A0,B0,B1,A1 per K-tile) is issued in global phase h-6.
//  * one half-tile (2 x global_load_lds_dwordx4 per thread) is issued per phase and FOUR half-tiles stay in
//    flight: `s_waitcnt vmcnt(8)` per phase, never 0 in the main loop; raw s_barrier (no compiler vmcnt(0)).
//  * the two waves of a SIMD belong to different M-groups, and group 1 runs one barrier behind group 0:
//    while one wave of the SIMD issues its MFMA cluster the other issues ds_reads + LDS-DMA.
//  Hazards: RAW - the wait that retires half-tile h sits in the phase BEFORE its first read, in front of a barrier
//  every wave crosses; WAR - a region is restaged >= 2 phases after its last read (covers the group stagger).
constexpr int T2_THREADS = 512;
constexpr int T2_BM = 256, T2_BN = 256;
constexpr int T2_HALF_BYTES = 128 * BK * 2;             // 16 KiB
constexpr int T2_BUF_BYTES = 4 * T2_HALF_BYTES;         // A0 A1 B0 B1
constexpr int T2_SMEM = 2 * T2_BUF_BYTES;               // 128 KiB
static_assert(128 * T2_BN * 4 <= T2_SMEM, "an epilogue pass (128 fp32 rows, unpadded) must fit the staging LDS");

template <bool KM>
__device__ __forceinline__ void stage_half(const bf16_t* __restrict__ base, long ld, int row0, int rows_total, int k0,
                                           char* lds_half, int tid, int wave_base_tid) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int cid = it * T2_THREADS + tid;
        const bf16_t* src;
        if (!KM) {
            const int row = cid >> 3, pc = cid & 7;
            int grow = row0 + row;
            grow = grow < rows_total ? grow : rows_total - 1;
            src = base + (long)grow * ld + k0 + ((pc ^ (row & 7)) << 3);
        } else {
            const int krow = cid >> 4, pc = cid & 15;
            int col = row0 + ((pc ^ km_swz(krow)) << 3);
            col = col <= rows_total - 8 ? col : rows_total - 8;
            src = base + (long)(k0 + krow) * ld + col;
        }
        char* dst = lds_half + (it * T2_THREADS + wave_base_tid) * 16;
        __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(dst), 16, 0, 0);
    }
}

// ---- main loop of one 256 x 256 tile over K-tiles [kt_begin, kt_begin + nk): prologue, 4-phase loop, drain.  Leaves every
// wave behind a workgroup barrier with no LDS-DMA outstanding (the staging LDS is free for the epilogue).
template <bool A_KM, bool B_KM>
__device__ __forceinline__ void gemm256_mainloop(const GemmArgs& g, char* smem, const int m0, const int n0, const int kt_begin,
                                                 const int nk, f32x4 (&acc)[2][2][4][2], const int tid, const int lane, const int wr,
                                                 const int wc, const int wave_base_tid, unsigned* publish_flag = nullptr) {
    const int total_h = 4 * nk;                         // half-tiles in stream order A0 B0 B1 A1 per K-tile

    // issue half-tile h of the stream (uniform control flow)
    auto stage = [&](int h) {
        const int t = h >> 2, part = h & 3;
        char* buf = smem + (t & 1) * T2_BUF_BYTES;
        const int k0 = (kt_begin + t) * BK;
        if (part == 0) stage_half<A_KM>(g.A, g.lda, m0, g.M, k0, buf, tid, wave_base_tid);
        else if (part == 3) stage_half<A_KM>(g.A, g.lda, m0 + 128, g.M, k0, buf + T2_HALF_BYTES, tid, wave_base_tid);
        else if (part == 1) stage_half<B_KM>(g.B, g.ldb, n0, g.N, k0, buf + 2 * T2_HALF_BYTES, tid, wave_base_tid);
        else stage_half<B_KM>(g.B, g.ldb, n0 + 128, g.N, k0, buf + 3 * T2_HALF_BYTES, tid, wave_base_tid);
    };

#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][b][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- prologue: 6 half-tiles in flight, the first two retired
#pragma unroll
    for (int h = 0; h < 6; ++h)
        if (h < total_h) stage(h);
    if (total_h >= 6) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    APTAI_STAMP(1);
    // stream-K: the slab stores of the PREVIOUS segment are older than this prologue's loads, vector-memory operations retire in
    // order, so every wave that passed the counted wait above has drained its write-through stores, and every wave has passed
    // it once the barrier releases: the ready flag goes out here, its drain hidden under this prologue's own latency
    if (publish_flag != nullptr && tid == 0) __hip_atomic_store(publish_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wr == 1) __builtin_amdgcn_s_barrier();          // group 1 runs one barrier behind group 0

    // K-major fragments come through inline asm (Frag<true>): the compiler would put `s_waitcnt vmcnt(0)` in front of the
    // ds_read_tr builtins and drain the four half-tiles this schedule keeps in flight.  They are complete at the explicit
    // lgkmcnt(0) in front of each MFMA section (the sched_barrier behind it keeps the MFMAs below).
    Frag<A_KM> af[4][2];
    Frag<B_KM> b0f[2][2], b1f[2][2];
    for (int t = 0; t < nk; ++t) {
        const char* buf = smem + (t & 1) * T2_BUF_BYTES;
        const char* sA0 = buf, *sA1 = buf + T2_HALF_BYTES, *sB0 = buf + 2 * T2_HALF_BYTES, *sB1 = buf + 3 * T2_HALF_BYTES;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            // ---------------- load section: fragment reads of this phase, then the LDS-DMA of half-tile G+6
            if (p == 0) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) b0f[j][ks].read(sB0, wc * 32 + j * 16, ks, lane);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) af[i][ks].read(sA0, wr * 64 + i * 16, ks, lane);
            } else if (p == 1) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) b1f[j][ks].read(sB1, wc * 32 + j * 16, ks, lane);
            } else if (p == 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) af[i][ks].read(sA1, wr * 64 + i * 16, ks, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
            const int G = 4 * t + p;
            const int hn = G + 6;
            if (hn < total_h) {
                stage(hn);
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            } else {
                const int allow = total_h - G - 3;      // half-tiles that may stay in flight (need h <= G+2 landed)
                if (allow >= 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else if (allow == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else if (allow == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            // ---------------- MFMA section: one quadrant x K=64
            __builtin_amdgcn_s_setprio(1);
            constexpr int QM[4] = {0, 0, 1, 1}, QN[4] = {0, 1, 1, 0};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const bf16x8 bfrag = (QN[p] == 0) ? b0f[j][ks].get() : b1f[j][ks].get();
                        acc[QM[p]][QN[p]][i][j] =
                            __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfrag, af[i][ks].get(), acc[QM[p]][QN[p]][i][j], 0, 0, 0);
                    }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        }
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();          // re-align the two groups
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// ---- epilogue of one 256 x 256 tile; ends behind a barrier.
// Two passes of 128 rows x 256 cols of fp32 through the (free) staging LDS.  In-kernel stamps (tools/gemm256_stamps.py) put the first
// form of this epilogue - four 64-row passes written by half of the waves, a rolled loop of [2 LDS reads -> arithmetic -> store] per row
// with the residual / aux loads inside it - at 7.5 us of a 27 us tile at K = 768 (13-14 us with a GELU): a chain of latencies, not a
// bandwidth.  Now: every wave stages its half in each pass (4 barriers instead of 8); rows are read back in groups of four with all
// eight LDS reads and the residual / aux row loads of the group issued before the arithmetic of the first; the staging rows carry no
// padding (exactly 128 KiB) and are conflict-free both ways - 16-byte chunk c of row r sits at slot (pi(c) ^ (r & 15)) with
// pi(c) = (c & 32) | ((c & 1) << 4) | ((c >> 1) & 15): the 16 lanes of a write (same chunk, 16 rows) and the 16 lanes of a read (one row,
// chunks 2L, then 2L + 1) both cover 16 distinct slots of one 256-byte group.
__device__ __forceinline__ int epi256_off(int row, int c) {
    const int p = (c & 32) | ((c & 1) << 4) | ((c >> 1) & 15);
    return row * (T2_BN * 4) + ((p ^ (row & 15)) << 4);
}

template <bool OUT_F32, int FM>
__device__ __forceinline__ void gemm256_epilogue_body(const GemmArgs& g, char* smem, const int m0, const int n0, const int split,
                                                      const f32x4 (&acc)[2][2][4][2], const int tid, const int lane, const int wr,
                                                      const int wc, const int flags_rt) {
    const int flags = FM >= 0 ? FM : flags_rt;
    const int L = tid & 31;
    const int cl = L * 8;
    const int n = n0 + cl;
    const bool n_ok = n < g.N;
    float bias8[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) bias8[r] = 0.f;
    if ((flags & APTAI_EPI_BIAS) && n_ok) {
        const f32x4 bb0 = *(const f32x4*)(g.bias + n), bb1 = *(const f32x4*)(g.bias + n + 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) { bias8[r] = bb0[r]; bias8[4 + r] = bb1[r]; }
    }
    const float alpha = (flags & APTAI_EPI_ALPHA) ? g.alpha : 1.0f;
    uint32_t sd0 = g.seed0, sd1 = g.seed1;
    if (!OUT_F32 && (flags & APTAI_EPI_DROPOUT)) apply_salt(g.salt, sd0, sd1);
    const bool want_res = !OUT_F32 && (flags & APTAI_EPI_RESIDUAL), want_aux = !OUT_F32 && (flags & (APTAI_EPI_DGELU | APTAI_EPI_MUL_AUX));
    const int row_in_step = tid >> 5;                   // 16 rows per step, 8 steps per pass
    // Steps are read back in groups: all LDS reads (and the residual / aux row loads, one group ahead) of a group are issued before the
    // arithmetic of its first step.  Four steps per group for the light flag words; two for the heavy ones and the run-time form, whose
    // arithmetic needs the registers (with four they spilled ~190 VGPRs - and a kernel that needs scratch memory no longer overlaps
    // cleanly with another queue's kernels: Force_APTAI's pipelined step lost 6 % to it, measured by swapping this file alone).
    constexpr bool HEAVY = OUT_F32 ? false : (FM < 0 || (FM & (APTAI_EPI_GELU | APTAI_EPI_DROPOUT | APTAI_EPI_DGELU | APTAI_EPI_RESIDUAL | APTAI_EPI_MUL_AUX)) != 0);
    constexpr bool HEAVIEST = !OUT_F32 && (FM < 0 || ((FM & APTAI_EPI_GELU) && (FM & (APTAI_EPI_DROPOUT | APTAI_EPI_PRE_DGELU))));
    constexpr int GS = HEAVIEST ? 1 : HEAVY ? 2 : 4, NG = 8 / GS;      // steps per group, groups per pass
    u32x4 resv[2][GS], auxv[2][GS];
    auto prefetch = [&](int slot, int pqm, int grp) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < GS; ++k) {
            const int m = m0 + pqm * 128 + (grp * GS + k) * 16 + row_in_step;
            const bool ok = n_ok && m < g.M;
            u32x4 rq = {0u, 0u, 0u, 0u}, aq = {0u, 0u, 0u, 0u};   // (values first, ONE unconditional array store each: two conditional
            if (ok && want_res) rq = *(const u32x4*)(g.residual + (long)m * g.ldr + n);   // stores were merged into a store through a
            if (ok && want_aux) aq = *(const u32x4*)(g.aux + (long)m * g.ldaux + n);       // selected stack address: scratch memory)
            resv[slot][k] = rq;
            auxv[slot][k] = aq;
        }
    };
    if (want_res || want_aux) prefetch(0, 0, 0);
#pragma unroll
    for (int pqm = 0; pqm < 2; ++pqm) {
#pragma unroll
        for (int qn = 0; qn < 2; ++qn)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    *(f32x4*)(smem + epi256_off(wr * 64 + i * 16 + (lane & 15), qn * 32 + wc * 8 + j * 4 + (lane >> 4))) = acc[pqm][qn][i][j];
        __syncthreads();
#pragma unroll
        for (int grp = 0; grp < NG; ++grp) {
            const int q = pqm * NG + grp, cur = q & 1;
            if ((want_res || want_aux) && q + 1 < 2 * NG) prefetch(cur ^ 1, (q + 1) / NG, (q + 1) % NG);
            f32x4 v0[GS], v1[GS];
#pragma unroll
            for (int k = 0; k < GS; ++k) {
                const int lr = (grp * GS + k) * 16 + row_in_step;
                v0[k] = *(const f32x4*)(smem + epi256_off(lr, 2 * L));
                v1[k] = *(const f32x4*)(smem + epi256_off(lr, 2 * L + 1));
            }
#pragma unroll
            for (int k = 0; k < GS; ++k) {
                const int m = m0 + pqm * 128 + (grp * GS + k) * 16 + row_in_step;
                if (m >= g.M || !n_ok) continue;
                float v[8] = {v0[k][0], v0[k][1], v0[k][2], v0[k][3], v1[k][0], v1[k][1], v1[k][2], v1[k][3]};
                if (flags & (APTAI_EPI_BIAS | APTAI_EPI_ALPHA)) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] = fmaf(v[r], alpha, bias8[r]);
                }
                if (OUT_F32) {
                    if (flags & APTAI_EPI_RESIDUAL_F32) {             // fp32 residual stream (inference-only encoder)
                        const float* R = (const float*)g.residual + (long)m * g.ldr + n;
                        const f32x4 r0 = *(const f32x4*)R, r1 = *(const f32x4*)(R + 4);
#pragma unroll
                        for (int r = 0; r < 4; ++r) { v[r] += r0[r]; v[4 + r] += r1[r]; }
                    }
                    if (flags & APTAI_EPI_BIAS_ROW) { const float bm = g.bias[m];
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] += bm; }
                    if (flags & APTAI_EPI_BIAS_ROW) { const float bm = g.bias[m];
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] += bm; }
                if (flags & APTAI_EPI_SPLIT_OUT) { split_out_store(g, flags, v, (long)m, n); continue; }
                    float* C = (float*)g.C + (long)split * g.slab_stride + (long)m * g.ldc + n;
                    *(f32x4*)C = (f32x4){v[0], v[1], v[2], v[3]};
                    *(f32x4*)(C + 4) = (f32x4){v[4], v[5], v[6], v[7]};
                    continue;
                }
                epilogue_chunk<FM>(v, g, flags, (long)m, n, auxv[cur][k], resv[cur][k], sd0, sd1);
            }
            if (HEAVY) __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
}

template <bool OUT_F32>
__device__ __forceinline__ void gemm256_epilogue(const GemmArgs& g, char* smem, const int m0, const int n0, const int split,
                                                 const f32x4 (&acc)[2][2][4][2], const int tid, const int lane, const int wr,
                                                 const int wc) {
    // Everything the epilogue derives from the thread / tile indices is recomputed from opaque copies: otherwise the addresses of
    // all the specialised bodies are hoisted above the main loop as loop invariants and spill INSIDE it (measured: main loop 18.4 ->
    // 23.4 us per tile at K = 768).
    int tid_e = tid, lane_e = lane, m0_e = __builtin_amdgcn_readfirstlane(m0), n0_e = __builtin_amdgcn_readfirstlane(n0);
    asm volatile("" : "+v"(tid_e), "+v"(lane_e), "+s"(m0_e), "+s"(n0_e));
    const int fx = epi_flag_word(g);
    if (OUT_F32) {                                      // fp32 outputs (weight gradients, exact mode): alpha / bias / fp32 residual only
        gemm256_epilogue_body<true, -1>(g, smem, m0_e, n0_e, split, acc, tid_e, lane_e, wr, wc, fx);
    } else {
        epi_dispatch(fx, [&](auto w) {
            gemm256_epilogue_body<false, decltype(w)::value>(g, smem, m0_e, n0_e, split, acc, tid_e, lane_e, wr, wc, fx);
        });
    }
}

template <bool A_KM, bool B_KM, bool OUT_F32>
__global__ __launch_bounds__(T2_THREADS, 2) void gemm256_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;            // waves w and w+4 share a SIMD and sit in different groups
    const int wave_base_tid = wave * 64;

    const int nwg = g.tiles_m * g.tiles_n;
    const int bid = xcd_remap(blockIdx.x, nwg);
    int tile_m, tile_n;
    raster2d(bid, g.tiles_m, g.tiles_n, g.raster_gm, tile_m, tile_n);
    const int m0 = tile_m * T2_BM, n0 = tile_n * T2_BN;
    if (gridDim.y > 1) {
        const int bo = blockIdx.y / g.nb_inner, bi = blockIdx.y % g.nb_inner;
        g.A += bo * g.sA[0] + bi * g.sA[1];
        g.B += bo * g.sB[0] + bi * g.sB[1];
        const long co = bo * g.sC[0] + bi * g.sC[1];
        g.C = (OUT_F32 && !(g.flags & APTAI_EPI_SPLIT_OUT)) ? (void*)((float*)g.C + co) : (void*)((bf16_t*)g.C + co);
        if (g.out_pre) g.out_pre += co;
        if (g.bias) g.bias += bo * g.sBias[0] + bi * g.sBias[1];
        if (g.residual) g.residual += bo * g.sR[0] + bi * g.sR[1];
        if (g.aux) g.aux += bo * g.sAux[0] + bi * g.sAux[1];
    }
    const int split = blockIdx.z;
    const int total_kt = g.K / BK;
    const int kt_begin = split * g.ktiles_per_split;
    int kt_end = kt_begin + g.ktiles_per_split;
    kt_end = kt_end < total_kt ? kt_end : total_kt;
    f32x4 acc[2][2][4][2];                              // [qm][qn][i][j]
    APTAI_STAMP(0);
    gemm256_mainloop<A_KM, B_KM>(g, smem, m0, n0, kt_begin, kt_end - kt_begin, acc, tid, lane, wr, wc, wave_base_tid);
    APTAI_STAMP(2);
    gemm256_epilogue<OUT_F32>(g, smem, m0, n0, split, acc, tid, lane, wr, wc);
    APTAI_STAMP(3);
#ifdef APTAI_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    APTAI_STAMP(4);
    if (threadIdx.x == 0 && blockIdx.x < 4096 && blockIdx.y == 0 && blockIdx.z == 0) {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_stamps[blockIdx.x * 8 + 5] = ((unsigned long long)xcc << 32) | hwid;
    }
#endif
}

// =====================================================================================================================
// Stream-K form of the 256 x 256 kernel: ONE persistent workgroup per CU, every workgroup gets the same number (+-1) of
// (tile, K-tile) iterations, so [8192] x {2304, 3072} outputs (288 / 384 tiles = 1.125 / 1.5 rounds of 256 CUs) cost
// 1.125 / 1.5 rounds' worth of main loop instead of 2.  Workgroup c owns the iteration range [c T / G, (c + 1) T / G) of the
// tile-major sequence (T = tiles x K-tiles, G = grid).  A range cuts at most one tile at its start and one at its end:
//   * the END cut (the K-head of a tile another workgroup finishes) is computed FIRST and published as an fp32 slab in the
//     accumulator's own register layout (write-through 16-byte stores, then one flag word: guide Guideline 16, R1);
//   * whole tiles follow, with the ordinary epilogue;
//   * the START cut (the K-tail of a tile: this workgroup is its OWNER) comes LAST: by then the lower-numbered workgroups that
//     hold the rest of that tile have long published, the owner adds their slabs to its accumulators and runs the epilogue.
// A tile's contributors always have lower numbers than its owner and publish before anything else, so a waiting owner never
// depends on a workgroup that is queued behind it; every wait is bounded all the same (status word, give-up = garbage tile,
// never a hang).  Flags are self-cleaning: the owner re-zeroes what it consumed, the workspace is zeroed once at allocation.
struct SkArgs {
    float* slabs;             // [G][256 * 256] fp32
    unsigned* flags;          // [G] ready flags (0 / 1); flags[-1 .. ] : see launch
    unsigned* status;         // set to 1 when a bounded wait gives up
    int kt_per_tile;
    int total_iters;          // tiles x K-tiles, < 2^23
};
constexpr unsigned SK_SPIN_LIMIT = 1u << 20;        // ~1 s of polling before a wait gives up
constexpr long SK_SLAB_FLOATS = (long)T2_BM * T2_BN;

template <bool A_KM, bool B_KM, bool OUT_F32>
__global__ __launch_bounds__(T2_THREADS, 2) void gemm256_sk_kernel(GemmArgs g, SkArgs sk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int wave_base_tid = wave * 64;
    const int G = gridDim.x;
    const int c = xcd_remap(blockIdx.x, G);             // consecutive workgroups (shared tiles, shared panels) on one XCD
    const int KT = sk.kt_per_tile;
    const int total = sk.total_iters;                   // < 2^23 (launcher), so c * total fits 32 bits
    const int it0 = (int)((unsigned)c * (unsigned)total / (unsigned)G), it1 = (int)((unsigned)(c + 1) * (unsigned)total / (unsigned)G);
    f32x4 acc[2][2][4][2];

    auto tile_origin = [&](int tile, int& m0, int& n0) {
        int tile_m, tile_n;
        raster2d(tile, g.tiles_m, g.tiles_n, g.raster_gm, tile_m, tile_n);
        m0 = tile_m * T2_BM;
        n0 = tile_n * T2_BN;
    };

    // ---- 1. the END cut: K-tiles [kb, kb + n) of a tile whose tail belongs to a later workgroup -> slab c
    int end_main = it1;
    unsigned* pending_flag = nullptr;
    if (it1 % KT != 0) {
        const int tile = (it1 - 1) / KT;
        const int seg0 = it0 > tile * KT ? it0 : tile * KT;
        int m0, n0;
        tile_origin(tile, m0, n0);
        gemm256_mainloop<A_KM, B_KM>(g, smem, m0, n0, seg0 - tile * KT, it1 - seg0, acc, tid, lane, wr, wc, wave_base_tid);
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc((void*)(sk.slabs + (long)c * SK_SLAB_FLOATS), 0, (int)(SK_SLAB_FLOATS * 4), 0x00020000);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const unsigned idx = (unsigned)(((a * 2 + b) * 4 + i) * 2 + j);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[a][b][i][j]), rsrc,
                                                               (idx * T2_THREADS + (unsigned)tid) * 16u, 0, 16);      // aux 16 = sc1
                    }
        end_main = seg0;
        pending_flag = sk.flags + c;                                     // published behind the NEXT segment's prologue wait
        if (it0 >= end_main) {                                           // nothing follows: drain and publish here
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // EVERY storing wave drains its write-through stores
            __syncthreads();
            if (tid == 0) __hip_atomic_store(pending_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pending_flag = nullptr;
        }
    }
    // ---- 2. whole tiles (the first segment is deferred when it starts inside a tile)
    const bool first_partial = (it0 % KT != 0) && it0 < end_main;
    const int first_end = (it0 / KT + 1) * KT;                           // <= end_main whenever first_partial
    for (int it = first_partial ? first_end : it0; it < end_main; it += KT) {
        int m0, n0;
        tile_origin(it / KT, m0, n0);
        gemm256_mainloop<A_KM, B_KM>(g, smem, m0, n0, 0, KT, acc, tid, lane, wr, wc, wave_base_tid, pending_flag);
        pending_flag = nullptr;
        gemm256_epilogue<OUT_F32>(g, smem, m0, n0, 0, acc, tid, lane, wr, wc);
    }
    // ---- 3. the START cut: this workgroup owns the tile; add the slabs of the workgroups that hold its K-head
    if (first_partial) {
        const int tile = it0 / KT;
        int m0, n0;
        tile_origin(tile, m0, n0);
        gemm256_mainloop<A_KM, B_KM>(g, smem, m0, n0, it0 - tile * KT, first_end - it0, acc, tid, lane, wr, wc, wave_base_tid, pending_flag);
        int cfirst = c - 1;                                              // contributors cfirst .. c-1 (it1(c-1) = it0 > tile * KT)
        while (cfirst > 0 && (int)((unsigned)cfirst * (unsigned)total / (unsigned)G) > tile * KT) --cfirst;
        if (tid == 0) {
            bool ok = true;
            for (int cp = cfirst; cp < c && ok; ++cp) {
                unsigned spins = 0;
                while (__hip_atomic_load(sk.flags + cp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                    if (++spins > SK_SPIN_LIMIT) { atomicExch(sk.status, 1u); ok = false; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");           // ONE invalidate of this CU's L1 after the polls
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        // The slabs come in through LDS-DMA into the (free) staging LDS, 64 KiB quarters two deep: no registers are tied up, so a
        // whole quarter is in flight per wave instead of the 4-8 loads the 256-register budget left room for (a plain-load
        // version spent ~15 us here).  A lane reads back exactly the bytes its own DMA instructions wrote (wave-uniform base +
        // lane * 16), so its counted vmcnt is the only ordering needed - no barrier.
        {
            constexpr int QI = 8;                                        // accumulator f32x4 per lane and quarter
            const int nq = 4 * (c - cfirst);
            auto issue_quarter = [&](int q) {
                const float* slab = sk.slabs + (long)(cfirst + (q >> 2)) * SK_SLAB_FLOATS;
                char* dst = smem + (q & 1) * (QI * T2_THREADS * 16);
#pragma unroll
                for (int e = 0; e < QI; ++e) {
                    const int idx = (q & 3) * QI + e;
                    __builtin_amdgcn_global_load_lds(GLB_PTR(slab + ((long)idx * T2_THREADS + tid) * 4),
                                                     LDS_PTR(dst + (e * T2_THREADS + wave_base_tid) * 16), 16, 0, 0);
                }
            };
            issue_quarter(0);
            if (nq > 1) issue_quarter(1);
            for (int q = 0; q < nq; ++q) {
                if (q + 1 < nq) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const char* src = smem + (q & 1) * (QI * T2_THREADS * 16);
                f32x4 part[QI];
#pragma unroll
                for (int e = 0; e < QI; ++e) part[e] = *(const f32x4*)(src + (e * T2_THREADS + tid) * 16);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the reads are done before the buffer is refilled
                if (q + 2 < nq) issue_quarter(q + 2);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                const int idx = ((a * 2 + b) * 4 + i) * 2 + j;
                                if ((idx >> 3) == (q & 3)) acc[a][b][i][j] += part[idx & 7];
                            }
            }
            __syncthreads();                                             // the epilogue reuses this LDS: every wave is done reading
        }
        if (tid == 0)
            for (int cp = cfirst; cp < c; ++cp) __hip_atomic_store(sk.flags + cp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        gemm256_epilogue<OUT_F32>(g, smem, m0, n0, 0, acc, tid, lane, wr, wc);
    }
}

template <bool A_KM, bool B_KM, bool OUT_F32>
int launch_gemm256(GemmArgs g, int nbatch, int nsplit, hipStream_t stream) {
    auto kern = gemm256_kernel<A_KM, B_KM, OUT_F32>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, T2_SMEM);
        attr_set = true;
    }
    g.tiles_m = (int)ceil_div(g.M, T2_BM);
    g.tiles_n = (int)ceil_div(g.N, T2_BN);
    dim3 grid(g.tiles_m * g.tiles_n, nbatch, nsplit);
    APTAI_LAUNCH(kern, grid, dim3(T2_THREADS), T2_SMEM, stream, g);
    APTAI_CHECK_LAUNCH("gemm256_kernel");
    return APTAI_OK;
}

constexpr int SK_MAX_GRID = 256;                         // one workgroup per CU of an MI355X
constexpr int64_t SK_HEADER_BYTES = 4096;                // status word @0, flags @256 .. 256 + 4 * SK_MAX_GRID
static inline int64_t sk_workspace_bytes() { return SK_HEADER_BYTES + (int64_t)SK_MAX_GRID * SK_SLAB_FLOATS * 4; }

template <bool A_KM, bool B_KM, bool OUT_F32>
int launch_gemm256_sk(GemmArgs g, void* ws, hipStream_t stream) {
    auto kern = gemm256_sk_kernel<A_KM, B_KM, OUT_F32>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, T2_SMEM);
        attr_set = true;
    }
    g.tiles_m = (int)ceil_div(g.M, T2_BM);
    g.tiles_n = (int)ceil_div(g.N, T2_BN);
    SkArgs sk;
    sk.status = (unsigned*)ws;
    sk.flags = (unsigned*)((char*)ws + 256);
    sk.slabs = (float*)((char*)ws + SK_HEADER_BYTES);
    sk.kt_per_tile = g.K / BK;
    const long total = (long)g.tiles_m * g.tiles_n * sk.kt_per_tile;
    if (total >= (1L << 23)) APTAI_FAIL(APTAI_ERR_INVALID, "aptai_gemm_bf16: stream-K iteration count %ld out of range", total);
    sk.total_iters = (int)total;
    const int grid = total < SK_MAX_GRID ? (int)total : SK_MAX_GRID;
    APTAI_LAUNCH(kern, dim3(grid), dim3(T2_THREADS), T2_SMEM, stream, g, sk);
    APTAI_CHECK_LAUNCH("gemm256_sk_kernel");
    return APTAI_OK;
}

// =====================================================================================================================
// 128 x 192 x 64 tile, 512 threads = 8 waves (2 M-rows x 4 N-columns, wave tile 64 x 48 = 4 x 3 MFMA tiles), ONE block
// per CU, 3-stage LDS ring (120 KiB).
//
// Why this shape: the transformer GEMMs of the hot path are [B*Tp = 8192] x {768, 2304, 3072}.  With 128 x 192 tiles
// that is 64 x {4, 12, 16} = 256 / 768 / 1024 tiles = exactly 1 / 3 / 4 rounds of the 256 CUs, where the 128 x 128 kernel
// (2 blocks per CU, 512 slots) runs 0.75 / 2.25 / 3 rounds and the 256 x 256 kernel 0.375 / 1.125 / 1.5.
//
// Pipeline: K-tile kt lives in ring slot kt % 3 and is staged two K-tiles ahead by LDS-DMA; every thread issues exactly
// 5 x 16 B per stage in every operand layout, so `s_waitcnt vmcnt(5)` retires "all but the newest stage".  One barrier
// per K-tile, placed BETWEEN the two k-halves: the fragments of (kt+1, half 0) are read while the MFMAs of (kt, half 1)
// run, and those of (kt+1, half 1) under the MFMAs of (kt+1, half 0), so no LDS latency sits on the MFMA chain.
//   RAW: a thread waits for its own stage-(kt+1) loads, then crosses the barrier every wave crosses, before anyone reads
//        slot (kt+1) % 3.   WAR: slot kt % 3 is restaged (stage kt+3) after that same barrier, which each wave enters
//        with lgkmcnt(0), i.e. with all its reads of slot kt % 3 complete.
// K-major operands are stored as 64-row panels [64 k][128 B] (5 = 2 + 3 panels per stage) with the 32-byte-granular
// swizzle pn_swz: the 8 k-rows one ds_read_b64_tr_b16 lane group touches land in 8 distinct 32-B bank slots.
constexpr int T3_THREADS = 512;
constexpr int T3_BM = 128, T3_BN = 192;
constexpr int T3_A_BYTES = T3_BM * BK * 2;              // 16 KiB
constexpr int T3_B_BYTES = T3_BN * BK * 2;              // 24 KiB
constexpr int T3_STAGE = T3_A_BYTES + T3_B_BYTES;       // 40 KiB
constexpr int T3_SMEM = 3 * T3_STAGE;                   // 120 KiB
constexpr int T3_EPI_PITCH = T3_BN * 4 + 16;            // 784 B: 196 dwords = 4 mod 32 -> conflict-free 16-B row writes
static_assert(T3_BM * T3_EPI_PITCH <= T3_SMEM, "epilogue tile must fit the staging LDS");

__device__ __forceinline__ int pn_swz(int krow) { return (((krow >> 1) & 1) | (((krow >> 3) & 1) << 1)) << 1; }

// per-thread source pointer of staging instruction `it` of an operand tile at K offset 0
template <bool KM>
__device__ __forceinline__ const bf16_t* stage3_src(const bf16_t* __restrict__ base, long ld, int row0, int rows_total, int it,
                                                    int tid) {
    if (!KM) {
        const int slot = it * T3_THREADS + tid, row = slot >> 3, pc = slot & 7;
        int grow = row0 + row;
        grow = grow < rows_total ? grow : rows_total - 1;
        return base + (long)grow * ld + ((pc ^ (row & 7)) << 3);
    } else {
        const int krow = tid >> 3, pc = tid & 7;
        int col = row0 + it * 64 + ((pc ^ pn_swz(krow)) << 3);
        col = col <= rows_total - 8 ? col : rows_total - 8;
        return base + (long)krow * ld + col;
    }
}

template <bool KM>
__device__ __forceinline__ bf16x8 read_frag3(const char* lds_tile, int row_base, int ks, int lane) {
    if (!KM) {
        return read_frag<false>(lds_tile, row_base, ks, lane);
    } else {
        const char* panel = lds_tile + (row_base >> 6) * 8192;
        const int rb = row_base & 63;
        const int gq = lane >> 4, i = lane & 15, qq = i >> 2, p = i & 3;
        const int ch = (rb >> 3) + (p >> 1);
        const int sub = (p & 1) << 3;
        const int k_lo = ks * 32 + gq * 8 + qq;
        const int k_hi = k_lo + 4;
        const char* a0 = panel + k_lo * 128 + ((ch ^ pn_swz(k_lo)) << 4) + sub;
        const char* a1 = panel + k_hi * 128 + ((ch ^ pn_swz(k_hi)) << 4) + sub;
        short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)a0);
        short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)a1);
        short8v r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, r);
    }
}

template <bool A_KM, bool B_KM, bool OUT_F32>
__global__ __launch_bounds__(T3_THREADS, 1) void gemm192_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    APTAI_STAMP(0);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int wave_base_tid = wave * 64;

    const int nwg = g.tiles_m * g.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    int tile_m, tile_n;
    raster2d(bid, g.tiles_m, g.tiles_n, g.raster_gm, tile_m, tile_n);
    const int m0 = tile_m * T3_BM, n0 = tile_n * T3_BN;
    if (gridDim.y > 1) {
        const int bo = blockIdx.y / g.nb_inner, bi = blockIdx.y % g.nb_inner;
        g.A += bo * g.sA[0] + bi * g.sA[1];
        g.B += bo * g.sB[0] + bi * g.sB[1];
        const long co = bo * g.sC[0] + bi * g.sC[1];
        g.C = (OUT_F32 && !(g.flags & APTAI_EPI_SPLIT_OUT)) ? (void*)((float*)g.C + co) : (void*)((bf16_t*)g.C + co);
        if (g.out_pre) g.out_pre += co;
        if (g.bias) g.bias += bo * g.sBias[0] + bi * g.sBias[1];
        if (g.residual) g.residual += bo * g.sR[0] + bi * g.sR[1];
        if (g.aux) g.aux += bo * g.sAux[0] + bi * g.sAux[1];
    }
    const int split = blockIdx.z;
    const int total_kt = g.K / BK;
    const int kt_begin = split * g.ktiles_per_split;
    int kt_end = kt_begin + g.ktiles_per_split;
    kt_end = kt_end < total_kt ? kt_end : total_kt;
    const int nk = kt_end - kt_begin;

    // staging sources advance by one K-tile per stage() call (stages are issued in K order)
    const bf16_t* pa[2];
    const bf16_t* pb[3];
    const long stepA = A_KM ? (long)BK * g.lda : (long)BK;
    const long stepB = B_KM ? (long)BK * g.ldb : (long)BK;
#pragma unroll
    for (int it = 0; it < 2; ++it) pa[it] = stage3_src<A_KM>(g.A, g.lda, m0, g.M, it, tid) + (long)kt_begin * stepA;
#pragma unroll
    for (int it = 0; it < 3; ++it) pb[it] = stage3_src<B_KM>(g.B, g.ldb, n0, g.N, it, tid) + (long)kt_begin * stepB;
    auto stage = [&](int slot) {
        char* buf = smem + slot * T3_STAGE + wave_base_tid * 16;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            __builtin_amdgcn_global_load_lds(GLB_PTR(pa[it]), LDS_PTR(buf + it * (T3_THREADS * 16)), 16, 0, 0);
            pa[it] += stepA;
        }
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            __builtin_amdgcn_global_load_lds(GLB_PTR(pb[it]), LDS_PTR(buf + T3_A_BYTES + it * (T3_THREADS * 16)), 16, 0, 0);
            pb[it] += stepB;
        }
    };

    f32x4 acc[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if constexpr (!A_KM && B_KM && APTAI_GEMM192_ASM) {
        // dgrad layout (A K-contiguous, B K-major): every LDS read through inline asm with counted waits.  Behind an LDS-DMA the
        // compiler puts `s_waitcnt vmcnt(0)` in front of each ds_read_tr builtin (section "asm transposing reads" of the 128-tile
        // kernel), which drains this kernel's three-stage ring at every K-tile: the builtin form ran at the speed of the 64-row
        // tiles, so the tile rule never picked it.  One read_half = 4 ds_read_b128 (A) + 6 ds_read_b64_tr_b16 (B) = 10 LDS
        // operations, returned in order: `lgkmcnt(10)` = "everything but the newest half is back".
        u32x4 aa[2][4];
        short4v bl[2][3], bh[2][3];
        auto read_half = [&](int slot, int ks) {
            const char* sa = smem + slot * T3_STAGE;
            const char* sb = sa + T3_A_BYTES;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wm * 64 + i * 16 + (lane & 15), q = ks * 4 + (lane >> 4);
                const uint32_t addr = lds_u32(sa) + (uint32_t)(row * 128 + ((q ^ (row & 7)) << 4));
                asm volatile("ds_read_b128 %0, %1" : "=v"(aa[ks][i]) : "v"(addr));
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int row_base = wn * 48 + j * 16;
                const char* panel = sb + (row_base >> 6) * 8192;
                const int rb = row_base & 63;
                const int gq = lane >> 4, li = lane & 15, qq = li >> 2, pp = li & 3;
                const int ch = (rb >> 3) + (pp >> 1);
                const int sub = (pp & 1) << 3;
                const int k_lo = ks * 32 + gq * 8 + qq, k_hi = k_lo + 4;
                const uint32_t a0 = lds_u32(panel) + (uint32_t)(k_lo * 128 + ((ch ^ pn_swz(k_lo)) << 4) + sub);
                const uint32_t a1 = lds_u32(panel) + (uint32_t)(k_hi * 128 + ((ch ^ pn_swz(k_hi)) << 4) + sub);
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(bl[ks][j]) : "v"(a0));
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(bh[ks][j]) : "v"(a1));
            }
        };
        auto mfma_half = [&](int ks) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(combine_tr(bl[ks][j], bh[ks][j]), __builtin_bit_cast(bf16x8, aa[ks][i]),
                                                                        acc[i][j], 0, 0, 0);
        };
#define APTAI_TIE_HALF(K) "+v"(aa[K][0]), "+v"(aa[K][1]), "+v"(aa[K][2]), "+v"(aa[K][3]), "+v"(bl[K][0]), "+v"(bl[K][1]), "+v"(bl[K][2]), \
                          "+v"(bh[K][0]), "+v"(bh[K][1]), "+v"(bh[K][2])
        if (nk > 0) {
            stage(0);
            if (nk > 1) stage(1);
            if (nk > 2) stage(2);
            if (nk > 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else if (nk > 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            read_half(0, 0);
            read_half(0, 1);
        }
        int cur = 0;
        for (int kt = 0; kt < nk; ++kt) {
            const int nxt = cur == 2 ? 0 : cur + 1;
            asm volatile("s_waitcnt lgkmcnt(10)" : APTAI_TIE_HALF(0));          // half 0 of this K-tile is back; half 1 may be in flight
            mfma_half(0);
            if (kt + 1 < nk) {
                if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (kt + 3 < nk) stage(cur);
                asm volatile("" : APTAI_TIE_HALF(1));                            // half 1 completed at the wait above: MFMAs stay below it
                read_half(nxt, 0);
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : APTAI_TIE_HALF(1));
            }
            mfma_half(1);
            if (kt + 1 < nk) read_half(nxt, 1);
            cur = nxt;
        }
#undef APTAI_TIE_HALF
    } else {
    bf16x8 af[2][4], bfr[2][3];
    auto read_half = [&](int slot, int ks) {
        const char* sa = smem + slot * T3_STAGE;
        const char* sb = sa + T3_A_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) af[ks][i] = read_frag3<A_KM>(sa, wm * 64 + i * 16, ks, lane);
#pragma unroll
        for (int j = 0; j < 3; ++j) bfr[ks][j] = read_frag3<B_KM>(sb, wn * 48 + j * 16, ks, lane);
    };
    auto mfma_half = [&](int ks) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ks][j], af[ks][i], acc[i][j], 0, 0, 0);
    };

    if (nk > 0) {
        stage(0);
        if (nk > 1) stage(1);
        if (nk > 2) stage(2);
        if (nk > 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (nk > 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        read_half(0, 0);
        read_half(0, 1);
    }
    int cur = 0;
#ifdef APTAI_STAMPS
    // development: shader-clock time of the K-tile's five segments, summed over the K-tiles (thread 0 of each block; slots 1, 5, 6, 7 and
    // the high word of slot 5 hold cycles: tools/gemm256_stamps.py loop192)
    unsigned long long seg[5] = {0, 0, 0, 0, 0}, tp = __builtin_amdgcn_s_memtime();
#define SEG(i) do { const unsigned long long tn = __builtin_amdgcn_s_memtime(); seg[i] += tn - tp; tp = tn; } while (0)
#else
#define SEG(i) do {} while (0)
#endif
    for (int kt = 0; kt < nk; ++kt) {
        const int nxt = cur == 2 ? 0 : cur + 1;
        mfma_half(0);
        SEG(0);
        if (kt + 1 < nk) {
            if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            SEG(1);
            if (kt + 3 < nk) stage(cur);
            read_half(nxt, 0);
            SEG(2);
        }
        mfma_half(1);
        SEG(3);
        if (kt + 1 < nk) read_half(nxt, 1);
        SEG(4);
        cur = nxt;
    }
#ifdef APTAI_STAMPS
    if (threadIdx.x == 0 && blockIdx.x < 4096 && blockIdx.y == 0 && blockIdx.z == 0) {
        g_stamps[blockIdx.x * 8 + 1] = seg[0];
        g_stamps[blockIdx.x * 8 + 5] = seg[1];
        g_stamps[blockIdx.x * 8 + 6] = seg[2];
        g_stamps[blockIdx.x * 8 + 7] = seg[3] | (seg[4] << 32);
    }
#endif
#undef SEG
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();

    // ------------------------------------------------------------------ epilogue (same scheme as gemm_kernel)
    // 128 x 192 outputs = 3072 chunks of 8 columns = 6 passes x 512 threads; chunk c -> row c / 24, column 8 * (c % 24)
    // compiled per flag word, indices through opaque copies (see gemm_tile_body)
    APTAI_STAMP(2);
    int tid_e = tid, lane_e = lane, m0_e = __builtin_amdgcn_readfirstlane(m0), n0_e = __builtin_amdgcn_readfirstlane(n0);
    asm volatile("" : "+v"(tid_e), "+v"(lane_e), "+s"(m0_e), "+s"(n0_e));
    const int fx = epi_flag_word(g);
    auto body = [&](auto w) {
        constexpr int FM = decltype(w)::value;
        const int flags = FM >= 0 ? FM : fx;
        u32x4 resv[6], auxv[6];
    #pragma unroll
        for (int pass = 0; pass < 6; ++pass) {
            resv[pass] = (u32x4){0u, 0u, 0u, 0u};
            auxv[pass] = (u32x4){0u, 0u, 0u, 0u};
        }
        if (!OUT_F32 && (flags & (APTAI_EPI_RESIDUAL | APTAI_EPI_DGELU | APTAI_EPI_MUL_AUX))) {     // uniform; addresses clamped, no per-lane_e branches
    #pragma unroll
            for (int pass = 0; pass < 6; ++pass) {
                const int c = pass * T3_THREADS + tid_e, ml = c / 24, cl = (c - ml * 24) * 8;
                int m = m0_e + ml, n = n0_e + cl;
                m = m < g.M ? m : g.M - 1;
                n = n <= g.N - 8 ? n : g.N - 8;
                if (flags & APTAI_EPI_RESIDUAL) resv[pass] = *(const u32x4*)(g.residual + (long)m * g.ldr + n);
                if (flags & (APTAI_EPI_DGELU | APTAI_EPI_MUL_AUX)) auxv[pass] = *(const u32x4*)(g.aux + (long)m * g.ldaux + n);
            }
        }
    #pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ml = wm * 64 + i * 16 + (lane_e & 15);
    #pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int nl = wn * 48 + j * 16 + (lane_e >> 4) * 4;
                *(f32x4*)(smem + ml * T3_EPI_PITCH + nl * 4) = acc[i][j];
            }
        }
        __syncthreads();
        const float alpha = (flags & APTAI_EPI_ALPHA) ? g.alpha : 1.0f;
        uint32_t sd0 = g.seed0, sd1 = g.seed1;
        if (flags & APTAI_EPI_DROPOUT) apply_salt(g.salt, sd0, sd1);
    #pragma unroll
        for (int pass = 0; pass < 6; ++pass) {
            const int c = pass * T3_THREADS + tid_e, ml = c / 24, cl = (c - ml * 24) * 8;
            const int m = m0_e + ml, n = n0_e + cl;
            if (m >= g.M || n >= g.N) continue;
            const f32x4 v0 = *(const f32x4*)(smem + ml * T3_EPI_PITCH + cl * 4);
            const f32x4 v1 = *(const f32x4*)(smem + ml * T3_EPI_PITCH + cl * 4 + 16);
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            if (flags & APTAI_EPI_BIAS) {
                const f32x4 b0 = *(const f32x4*)(g.bias + n), b1 = *(const f32x4*)(g.bias + n + 4);
    #pragma unroll
                for (int r = 0; r < 4; ++r) { v[r] = fmaf(v[r], alpha, b0[r]); v[4 + r] = fmaf(v[4 + r], alpha, b1[r]); }
            } else if (flags & APTAI_EPI_ALPHA) {
    #pragma unroll
                for (int r = 0; r < 8; ++r) v[r] *= alpha;
            }
            if (OUT_F32) {
                if (flags & APTAI_EPI_RESIDUAL_F32) {             // fp32 residual stream (inference-only encoder): += res32[m][n..n+7]
                    const float* R = (const float*)g.residual + (long)m * g.ldr + n;
                    const f32x4 r0 = *(const f32x4*)R, r1 = *(const f32x4*)(R + 4);
    #pragma unroll
                    for (int r = 0; r < 4; ++r) { v[r] += r0[r]; v[4 + r] += r1[r]; }
                }
                if (flags & APTAI_EPI_BIAS_ROW) { const float bm = g.bias[m];
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] += bm; }
                if (flags & APTAI_EPI_SPLIT_OUT) { split_out_store(g, flags, v, (long)m, n); continue; }
                float* C = (float*)g.C + (long)split * g.slab_stride + (long)m * g.ldc + n;
                *(f32x4*)C = (f32x4){v[0], v[1], v[2], v[3]};
                *(f32x4*)(C + 4) = (f32x4){v[4], v[5], v[6], v[7]};
                continue;
            }
            epilogue_chunk<FM>(v, g, flags, (long)m, n, auxv[pass], resv[pass], sd0, sd1);
            __builtin_amdgcn_sched_barrier(0);             // keep the passes apart: interleaving all six spills
        }
    };
    if (OUT_F32) body(EpiWord<-1>{});
    else epi_dispatch(fx, body);
    APTAI_STAMP(3);
#ifdef APTAI_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    APTAI_STAMP(4);
#endif
}

template <bool A_KM, bool B_KM, bool OUT_F32>
int launch_gemm192(GemmArgs g, int nbatch, int nsplit, hipStream_t stream) {
    auto kern = gemm192_kernel<A_KM, B_KM, OUT_F32>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, T3_SMEM);
        attr_set = true;
    }
    g.tiles_m = (int)ceil_div(g.M, T3_BM);
    g.tiles_n = (int)ceil_div(g.N, T3_BN);
    dim3 grid(g.tiles_m * g.tiles_n, nbatch, nsplit);
    APTAI_LAUNCH(kern, grid, dim3(T3_THREADS), T3_SMEM, stream, g);
    APTAI_CHECK_LAUNCH("gemm192_kernel");
    return APTAI_OK;
}

}  // namespace

// validates one descriptor and fills the kernel arguments (tiles for the 128-tile kernel; the others recompute them)
static int build_args(const aptai_gemm_desc* d, GemmArgs& g, int& nbatch, int& nsplit, const void* stream) {
    APTAI_REQUIRE(d != nullptr, "aptai_gemm_bf16: null descriptor");
    APTAI_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "aptai_gemm_bf16: empty problem M=%ld N=%ld K=%ld", (long)d->M,
                  (long)d->N, (long)d->K);
    APTAI_REQUIRE(d->K % BK == 0, "aptai_gemm_bf16: K=%ld must be a multiple of %d", (long)d->K, BK);
    APTAI_REQUIRE(d->N % 8 == 0, "aptai_gemm_bf16: N=%ld must be a multiple of 8", (long)d->N);
    APTAI_REQUIRE(d->A && d->B && d->C, "aptai_gemm_bf16: null operand");
    APTAI_REQUIRE(d->lda % 8 == 0 && d->ldb % 8 == 0 && d->ldc % 8 == 0, "aptai_gemm_bf16: leading dims must keep 16-B alignment");
    if (d->residual) APTAI_REQUIRE(d->ldr % 8 == 0 && (uintptr_t)d->residual % 16 == 0, "aptai_gemm_bf16: residual must be 16-B aligned");
    if (d->aux) APTAI_REQUIRE(d->ldaux % 8 == 0 && (uintptr_t)d->aux % 16 == 0, "aptai_gemm_bf16: aux must be 16-B aligned");
    if (d->bias) APTAI_REQUIRE((uintptr_t)d->bias % 16 == 0, "aptai_gemm_bf16: bias must be 16-B aligned");
    APTAI_REQUIRE(((uintptr_t)d->A % 16 == 0) && ((uintptr_t)d->B % 16 == 0) && ((uintptr_t)d->C % 16 == 0),
                  "aptai_gemm_bf16: operands must be 16-byte aligned");
    if (d->a_kmajor) APTAI_REQUIRE(d->M % 8 == 0 && d->M >= 8, "aptai_gemm_bf16: K-major A needs M %% 8 == 0");
    if (d->b_kmajor) APTAI_REQUIRE(d->N >= 8, "aptai_gemm_bf16: K-major B needs N >= 8");
    if (d->flags & APTAI_EPI_BIAS) APTAI_REQUIRE(d->bias != nullptr, "aptai_gemm_bf16: EPI_BIAS without bias");
    if (d->flags & APTAI_EPI_RESIDUAL) APTAI_REQUIRE(d->residual != nullptr, "aptai_gemm_bf16: EPI_RESIDUAL without residual");
    if (d->flags & APTAI_EPI_RESIDUAL_F32)
        APTAI_REQUIRE(d->residual != nullptr && d->out_f32 && !(d->flags & APTAI_EPI_RESIDUAL) && d->split_k <= 1 && !d->accumulate &&
                      (d->tile == 128 || d->tile == 64 || d->tile == 192 || d->tile == 256 || d->tile == 257),
                      "aptai_gemm_bf16: EPI_RESIDUAL_F32 needs fp32 output, a residual, no split-K / accumulate and an explicit tile");
    if (d->flags & (APTAI_EPI_DGELU | APTAI_EPI_MUL_AUX)) APTAI_REQUIRE(d->aux != nullptr, "aptai_gemm_bf16: EPI_DGELU / EPI_MUL_AUX without aux");
    if (d->flags & APTAI_EPI_PRE_DGELU) APTAI_REQUIRE(d->out_pre != nullptr && (d->flags & APTAI_EPI_GELU), "aptai_gemm_bf16: EPI_PRE_DGELU needs EPI_GELU and out_pre");

    if (d->flags & APTAI_EPI_BIAS_ROW)
        APTAI_REQUIRE(d->out_f32 && d->bias != nullptr && !(d->flags & APTAI_EPI_BIAS) && (d->tile == 128 || d->tile == 192 || d->tile == 256) &&
                      d->split_k <= 1 && !d->accumulate, "aptai_gemm_bf16: EPI_BIAS_ROW needs out_f32, a bias of length M, no EPI_BIAS, tile 128 / 192 / 256");
    if (d->flags & APTAI_EPI_SPLIT_OUT)
        APTAI_REQUIRE(d->out_f32 && (d->tile == 128 || d->tile == 192 || d->tile == 256) && !d->a_kmajor && d->split_k <= 1 && !d->accumulate && (d->split_out_pieces == 3 || d->split_out_pieces == 6) &&
                      d->ldc >= d->split_out_pieces * d->N && d->N % 8 == 0,
                      "aptai_gemm_bf16: EPI_SPLIT_OUT needs out_f32, tile 128 / 192 / 256, 3 or 6 pieces and ldc >= pieces * N (bf16 elements)");
    APTAI_REQUIRE(d->colscale_n >= 0 && d->colscale_n % 8 == 0 && d->colscale_n <= d->N && (d->colscale_n == 0 || !d->out_f32),
                  "aptai_gemm_bf16: colscale_n must be a multiple of 8 within N, bf16 output only");
    memset(&g, 0, sizeof(g));
    g.A = (const bf16_t*)d->A; g.lda = d->lda;
    g.B = (const bf16_t*)d->B; g.ldb = d->ldb;
    g.C = d->C; g.ldc = d->ldc;
    g.M = (int)d->M; g.N = (int)d->N; g.K = (int)d->K;
    g.bias = d->bias;
    g.residual = (const bf16_t*)d->residual; g.ldr = d->ldr;
    g.out_pre = (bf16_t*)d->out_pre;
    g.aux = (const bf16_t*)d->aux; g.ldaux = d->ldaux;
    g.flags = d->flags;
    g.seed0 = (uint32_t)d->seed; g.seed1 = (uint32_t)(d->seed >> 32);
    g.salt = aptai_seed_salt(stream);
    g.thr16 = drop_thr16(d->dropout_p);
    g.dscale = drop_scale(g.thr16);
    if (g.thr16 == 0) g.flags &= ~APTAI_EPI_DROPOUT;
    g.alpha = d->alpha;
    g.colscale_n = d->colscale_n; g.colscale = d->colscale;
    g.split_pieces = d->split_out_pieces;
    g.split_bcol = (d->flags & APTAI_EPI_SPLIT_OUT) ? d->split_out_bcol : 0;
    g.hash_ld = d->N; g.hash_n0 = 0;
    g.tiles_m = (int)ceil_div(d->M, BM);
    g.tiles_n = (int)ceil_div(d->N, BN);
    {
        // default (-1 here, resolved per kernel in raster_default below): groups of 8 tile rows for the kernels that gain from
        // it, the plain row-major walk for the others; APTAI_GEMM_RASTER=<n> forces one value everywhere (A/B)
        static int raster = -2;
        if (raster == -2) {
            const char* e = getenv("APTAI_GEMM_RASTER");
            raster = e ? atoi(e) : -1;
        }
        g.raster_gm = raster;
    }
    const int total_kt = g.K / BK;
    nsplit = d->split_k > 0 ? d->split_k : 1;
    if (nsplit > total_kt) nsplit = total_kt;
    g.ktiles_per_split = (int)ceil_div(total_kt, nsplit);
    nsplit = (int)ceil_div(total_kt, g.ktiles_per_split);

    nbatch = 1;
    g.nb_inner = 1;
    if (d->batch_outer > 1 || d->batch_inner > 1) {
        const int bo = d->batch_outer > 0 ? d->batch_outer : 1, bi = d->batch_inner > 0 ? d->batch_inner : 1;
        nbatch = bo * bi;
        g.nb_inner = bi;
        for (int l = 0; l < 2; ++l) {
            g.sA[l] = d->batch_stride_a[l]; g.sB[l] = d->batch_stride_b[l]; g.sC[l] = d->batch_stride_c[l];
            g.sBias[l] = d->batch_stride_bias[l]; g.sR[l] = d->batch_stride_res[l]; g.sAux[l] = d->batch_stride_aux[l];
            APTAI_REQUIRE(g.sA[l] % 8 == 0 && g.sB[l] % 8 == 0 && g.sC[l] % 8 == 0 && g.sBias[l] % 4 == 0 &&
                              g.sR[l] % 8 == 0 && g.sAux[l] % 8 == 0,
                          "aptai_gemm_bf16: batch strides must keep vector alignment");
        }
        APTAI_REQUIRE(nbatch <= 65535, "aptai_gemm_bf16: too many batches");
    }
    const bool f32 = d->out_f32 != 0;
    if (nbatch > 1) APTAI_REQUIRE(!(f32 && (d->split_k > 1 || d->accumulate)), "aptai_gemm_bf16: batched GEMM cannot split-K/accumulate");
    if (!f32) APTAI_REQUIRE(nsplit == 1, "aptai_gemm_bf16: split-K needs fp32 output");
    if (f32 && (nsplit > 1 || d->accumulate)) {
        APTAI_REQUIRE(d->workspace != nullptr, "aptai_gemm_bf16: split-K / accumulate needs a workspace");
        APTAI_REQUIRE(d->ldc == d->N, "aptai_gemm_bf16: split-K output must be dense (ldc == N)");
        APTAI_REQUIRE((size_t)d->workspace_bytes >= (size_t)nsplit * d->M * d->N * 4, "aptai_gemm_bf16: workspace too small");
        g.C = d->workspace;
        g.slab_stride = (long)d->M * d->N;
    }
    return APTAI_OK;
}

static int gemm_bf16_one(const aptai_gemm_desc* d, void* stream_, long hash_ld, int hash_n0);

extern "C" int aptai_gemm_bf16(const aptai_gemm_desc* d, void* stream_) {
    // Outputs that make x.5 rounds of 256 x 256 tiles (base FFN1 forward / FFN2 dgrad: 8192 x 3072 = 384 tiles on 256 CUs) run as TWO
    // launches over column ranges: whole rounds of 256-row tiles (0.63 of the MFMA rate in the loop), then the remainder as one round of
    // 128-row tiles - instead of three rounds of 128-row tiles (0.34).  Measured both ways in one call: standalone the single launch is the
    // faster one since the 128-row kernels walk the tiles in 2-D groups (FFN1 forward 56-58 vs 60.8 us), in the step the split is (8.82-8.85
    // -> 8.78-8.79 ms, three interleaved pairs).  OFF by default (APTAI_GEMM_SPLITN=1 turns it on): 0.5 % is not worth a whole-CU 256-row launch
    // in the backward pass, where anything that runs beside it - Force_APTAI's BiLSTM clusters did, a gradient all-reduce would - cannot get a CU.  Bit-identical outputs and dropout masks
    // (tests/test_gpu_gemm.py: a tile's K walk does not depend on its size; the element index is the whole output's, hash_ld / hash_n0).
    static const bool split_on = getenv("APTAI_GEMM_SPLITN") && atoi(getenv("APTAI_GEMM_SPLITN")) != 0;
    if (split_on && d != nullptr && d->tile == 0 && !d->out_f32 && !d->a_kmajor && d->split_k <= 1 && !d->accumulate && d->batch_outer <= 1 &&
        d->batch_inner <= 1 && d->colscale_n == 0 && d->sk_workspace == nullptr && d->M % 256 == 0 && d->N % 256 == 0 && d->K <= 1024 &&
        getenv("APTAI_GEMM_TILE") == nullptr) {
        const long tm = d->M / 256, tn = d->N / 256, T = tm * tn;
        if (T > 256 && T % 256 == 128 && (T - 128) % tm == 0) {
            const long n1 = (T - 128) / tm * 256, rem = d->N - n1;
            if ((d->M / 128) * (rem / 128) <= 512) {
                aptai_gemm_desc a = *d, b = *d;
                a.N = n1; a.tile = 256;
                b.N = rem; b.tile = 128;
                b.B = (const char*)d->B + (d->b_kmajor ? n1 * 2 : n1 * d->ldb * 2);
                b.C = (char*)d->C + n1 * 2;
                if (d->out_pre) b.out_pre = (char*)d->out_pre + n1 * 2;
                if (d->bias) b.bias = d->bias + n1;
                if (d->residual) b.residual = (const char*)d->residual + n1 * 2;
                if (d->aux) b.aux = (const char*)d->aux + n1 * 2;
                const int rc = gemm_bf16_one(&a, stream_, d->N, 0);
                if (rc != APTAI_OK) return rc;
                return gemm_bf16_one(&b, stream_, d->N, (int)n1);
            }
        }
    }
    return gemm_bf16_one(d, stream_, d ? d->N : 0, 0);
}

static int gemm_bf16_one(const aptai_gemm_desc* d, void* stream_, long hash_ld, int hash_n0) {
    hipStream_t stream = (hipStream_t)stream_;
    GemmArgs g;
    int nbatch = 1, nsplit = 1;
    const int brc = build_args(d, g, nbatch, nsplit, stream_);
    if (brc != APTAI_OK) return brc;
    g.hash_ld = hash_ld; g.hash_n0 = hash_n0;
    const bool f32 = d->out_f32 != 0;
    float* final_out = (float*)d->C;
    // tile selection: the 256x256 deep-pipelined kernel when the grid still fills the chip, else 128x128 (2 blocks/CU)
    int tile = d->tile;
    if (tile == 0) {
        static int env_tile = -1;
        if (env_tile < 0) {
            const char* e = getenv("APTAI_GEMM_TILE");
            env_tile = e ? atoi(e) : 0;
        }
        tile = env_tile;
    }
    if (tile == 0) {
        // wave-quantisation model fitted to tools/gemm_bench*.py on MI355X: the 256-tile kernel sustains ~1.25x the
        // 128-tile kernel per busy CU (1 block/CU, 256 slots) but needs the grid to fill whole rounds of 256 tiles;
        // the 128-tile kernel runs 2 blocks/CU (512 slots).
        const long t256 = ceil_div(d->M, T2_BM) * ceil_div(d->N, T2_BN) * nbatch * nsplit;
        const long t128 = ceil_div(d->M, BM) * ceil_div(d->N, BN) * nbatch * nsplit;
        const double e256 = (double)t256 / (double)(ceil_div(t256, 256) * 256);
        const double e128 = (double)t128 / (double)(ceil_div(t128, 512) * 512);
        static double f256 = -1.0;
        if (f256 < 0) {
            const char* e = getenv("APTAI_GEMM_F256");      // A/B knob for the 256-tile advantage factor (default 1.25)
            f256 = e ? atof(e) : 1.25;
        }
        tile = (d->M >= 256 && d->N >= 256 && f256 * e256 > e128) ? 256 : 128;
        // 128x192 tiles (one block per CU): measured faster than the 128-tile kernel only where the whole K-contiguous GEMM
        // is ONE round of full tiles (8192 x 768: 16.9 vs 19.7 us at K = 768, 43.6 vs 51.2 us at K = 3072)
        const long t192 = ceil_div(d->M, T3_BM) * ceil_div(d->N, T3_BN) * nbatch * nsplit;
        // (round 3: also with a K-major B, the dgrads [8192] x 768 - since its LDS reads go through inline asm with counted waits the
        //  192-tile kernel no longer drains its ring at every K-tile: 35.6 vs 38.6 us at K = 2304, 45.1 vs 49.0 us at K = 3072)
        if (tile == 128 && !d->a_kmajor && d->M % T3_BM == 0 && d->N % T3_BN == 0 && t192 > 192 && t192 <= 256)
            tile = 192;
        // ... and where it is a whole number of rounds with a light epilogue (one block per CU leaves GELU / dropout arithmetic
        // exposed: 8192 x 2304 x 768 with bias only 38.3 vs 40.7-42.8 us for the 64- / 128-row tiles, but 58.0 vs 52.5 us with
        // bias + GELU + dropout + second output; tools/gemm_round.py)
        const bool light_epi = !(d->flags & (APTAI_EPI_GELU | APTAI_EPI_DROPOUT | APTAI_EPI_DGELU)) && d->out_pre == nullptr;
        const bool t192_rounds = tile == 128 && !d->a_kmajor && !d->b_kmajor && d->M % T3_BM == 0 && d->N % T3_BN == 0 && t192 % 256 == 0 &&
                                 t192 <= 768 && d->K <= 1024 && light_epi && nbatch == 1 && nsplit == 1;
        if (t192_rounds) tile = 192;
        // 256 x 192 tiles (round 4, csrc/gemm_t4.hip; one block per CU, 8.9 staged bytes per kflop against 12.9 / 15.2): measured faster only
        // where the whole K-contiguous GEMM is whole rounds of full tiles with a light epilogue - wav2vec2-large's q|k|v projection,
        // [4096] x 3072 x 1024 = exactly 256 tiles: 26-28 us against 31-32 (128-row tiles) and 29-30 (vendor library), tools/gemm_t4_bench.py,
        // profiles/r04_gemm_t4_bench.txt.  On the base model's [8192] x {2304, 3072} outputs it ties the 128-row kernels (43-44 us): every tile
        // of this family sits on the same LDS / delivery balance, and its exposed epilogue loses with GELU / dropout.
        const long t448 = ceil_div(d->M, 256) * ceil_div(d->N, 192);
        if (!d->a_kmajor && !d->b_kmajor && !f32 && nbatch == 1 && nsplit == 1 && !d->accumulate && d->M % 256 == 0 && d->N % 192 == 0 &&
            t448 % 256 == 0 && t448 <= 512 && light_epi && d->K >= 1024 && (tile == 128 || tile == 64 || tile == 256))
            tile = 448;
        // GEMMs that fill the 512 slots of the 128-tile kernel badly run as 64 x 128 tiles on 768 slots (3 blocks per CU): base
        // dgrads [8192] x 768 (384 tiles -> 768: 49.1 vs 54.9 us at K = 3072, 38.0 vs 42.9 us at K = 2304), base QKV (1152 tiles =
        // 2.25 rounds -> 2304 = 3 rounds), large [4096] x 1024 outputs (256 tiles -> 512: 13.7 vs 17.5 us at K = 1024, 42.4 vs
        // 47.5 us at K = 4096, dgrads 41-52 vs 52-65 us).  Whole step: -0.25 ms from the multi-round cases alone.
        // APTAI_GEMM_M64=0 disables the rule, =2 restricts it to single-round shapes (A/B).
        static int m64 = -1;
        if (m64 < 0) {
            const char* e = getenv("APTAI_GEMM_M64");
            m64 = e ? atoi(e) : 1;
        }
        const long t64 = ceil_div(d->M, 64) * ceil_div(d->N, BN) * nbatch * nsplit;
        const double e64 = (double)t64 / (double)(ceil_div(t64, 768) * 768);
        if (m64 && tile == 128 && !t192_rounds && !d->a_kmajor && d->M % 64 == 0 && (t128 <= 512 || m64 != 2) && e64 >= 1.2 * e128) tile = 64;
    }
    const bool sk_ok = d->sk_workspace != nullptr && d->sk_workspace_bytes >= sk_workspace_bytes() && nbatch == 1 && nsplit == 1 &&
                       !d->accumulate && d->M >= T2_BM && d->N >= T2_BN;
    if (tile == 257) APTAI_REQUIRE(sk_ok, "aptai_gemm_bf16: tile 257 (stream-K) needs sk_workspace (aptai_gemm_sk_workspace_bytes), no batching / "
                                          "split-K / accumulate and M, N >= 256");
    if (tile == 64 && d->a_kmajor) tile = 128;            // 64-row tiles need a K-contiguous A
    // 2-D rasterisation (tile groups of 8 rows per XCD) for every kernel.  Rounds 2 and 3a kept the row-major walk for the 128- and 64-row
    // kernels with a K-contiguous A, which then LOST 5-7 % to the raster (FFN1 forward 71.5 -> 76.7 us) while it cut their fabric-side
    // re-reads (FFN1: 129 MB fetched for 17 MB of operands); with the per-flag-word epilogues the balance is the other way: per forced
    // 128-row tile FFN1 forward 61.5-64.2 -> 59.3 us, FFN2 dgrad 50.4 -> 48.8 us, the layer's ten bf16-output GEMMs 388-392 -> 382-383 us
    // (tools/step_gemm_tiles.py, APTAI_GEMM_RASTER = -1 / 4 / 8 / 16 in one call; 64-row tiles 425 -> 416 us).  APTAI_GEMM_RASTER=0 = row-major.
    if (g.raster_gm < 0) g.raster_gm = 8;
    static const bool epi_runtime = getenv("APTAI_EPI_RUNTIME") && atoi(getenv("APTAI_EPI_RUNTIME")) != 0;
    if (epi_runtime) g.flags |= EPX_RUNTIME;
    if (!f32) epi_trace(g, tile);
    int rc;
    if (tile == 257) {
        if (!d->a_kmajor && !d->b_kmajor) rc = f32 ? launch_gemm256_sk<false, false, true>(g, d->sk_workspace, stream) : launch_gemm256_sk<false, false, false>(g, d->sk_workspace, stream);
        else if (!d->a_kmajor && d->b_kmajor) rc = f32 ? launch_gemm256_sk<false, true, true>(g, d->sk_workspace, stream) : launch_gemm256_sk<false, true, false>(g, d->sk_workspace, stream);
        else if (d->a_kmajor && d->b_kmajor) rc = f32 ? launch_gemm256_sk<true, true, true>(g, d->sk_workspace, stream) : launch_gemm256_sk<true, true, false>(g, d->sk_workspace, stream);
        else APTAI_FAIL(APTAI_ERR_INVALID, "aptai_gemm_bf16: A K-major with B K-contiguous is not built");
    } else
    if (tile == 64) {
        if (!d->b_kmajor) rc = f32 ? launch_gemm_m64<false, true>(g, nbatch, nsplit, stream) : launch_gemm_m64<false, false>(g, nbatch, nsplit, stream);
        else rc = f32 ? launch_gemm_m64<true, true>(g, nbatch, nsplit, stream) : launch_gemm_m64<true, false>(g, nbatch, nsplit, stream);
    } else
    if (tile == 448) {
        APTAI_REQUIRE(!f32 && !d->a_kmajor && nbatch == 1 && nsplit == 1 && !d->accumulate,
                      "aptai_gemm_bf16: tile 448 (256 x 192) is built for bf16 output, K-contiguous A, no batching / split-K");
        rc = launch_gemm_t4(g, d->b_kmajor != 0, stream);
    } else
    if (tile == 192) {
        if (!d->a_kmajor && !d->b_kmajor) rc = f32 ? launch_gemm192<false, false, true>(g, nbatch, nsplit, stream) : launch_gemm192<false, false, false>(g, nbatch, nsplit, stream);
        else if (!d->a_kmajor && d->b_kmajor) rc = f32 ? launch_gemm192<false, true, true>(g, nbatch, nsplit, stream) : launch_gemm192<false, true, false>(g, nbatch, nsplit, stream);
        else if (d->a_kmajor && d->b_kmajor) rc = f32 ? launch_gemm192<true, true, true>(g, nbatch, nsplit, stream) : launch_gemm192<true, true, false>(g, nbatch, nsplit, stream);
        else APTAI_FAIL(APTAI_ERR_INVALID, "aptai_gemm_bf16: A K-major with B K-contiguous is not built");
    } else
    if (tile == 256) {
        if (!d->a_kmajor && !d->b_kmajor) rc = f32 ? launch_gemm256<false, false, true>(g, nbatch, nsplit, stream) : launch_gemm256<false, false, false>(g, nbatch, nsplit, stream);
        else if (!d->a_kmajor && d->b_kmajor) rc = f32 ? launch_gemm256<false, true, true>(g, nbatch, nsplit, stream) : launch_gemm256<false, true, false>(g, nbatch, nsplit, stream);
        else if (d->a_kmajor && d->b_kmajor) rc = f32 ? launch_gemm256<true, true, true>(g, nbatch, nsplit, stream) : launch_gemm256<true, true, false>(g, nbatch, nsplit, stream);
        else APTAI_FAIL(APTAI_ERR_INVALID, "aptai_gemm_bf16: A K-major with B K-contiguous is not built");
    } else
    if (!d->a_kmajor && !d->b_kmajor) rc = f32 ? launch_gemm<false, false, true>(g, nbatch, nsplit, stream) : launch_gemm<false, false, false>(g, nbatch, nsplit, stream);
    else if (!d->a_kmajor && d->b_kmajor) rc = f32 ? launch_gemm<false, true, true>(g, nbatch, nsplit, stream) : launch_gemm<false, true, false>(g, nbatch, nsplit, stream);
    else if (d->a_kmajor && d->b_kmajor) rc = f32 ? launch_gemm<true, true, true>(g, nbatch, nsplit, stream) : launch_gemm<true, true, false>(g, nbatch, nsplit, stream);
    else APTAI_FAIL(APTAI_ERR_INVALID, "aptai_gemm_bf16: A K-major with B K-contiguous is not built");
    if (rc != APTAI_OK) return rc;
    if (f32 && (nsplit > 1 || d->accumulate)) {
        const long n4 = (long)d->M * d->N / 4;
        int blocks = (int)(n4 / 256 < 2048 ? (n4 + 255) / 256 : 2048);
        APTAI_LAUNCH(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, (const float*)d->workspace, final_out,
                           n4, g.slab_stride / 4, nsplit, d->accumulate ? 1 : 0);
        APTAI_CHECK_LAUNCH("splitk_reduce_kernel");
    }
    return APTAI_OK;
}

extern "C" int aptai_gemm_bf16_grouped(const aptai_gemm_desc* descs, int n, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    APTAI_REQUIRE(descs != nullptr && n >= 1 && n <= MAX_GROUP, "aptai_gemm_bf16_grouped: need 1..%d problems, got %d", MAX_GROUP, n);
    GroupArgs ga;
    memset(&ga, 0, sizeof(ga));
    int total = 0;
    for (int i = 0; i < n; ++i) {
        const aptai_gemm_desc* d = descs + i;
        int nbatch = 1, nsplit = 1;
        const int brc = build_args(d, ga.p[i], nbatch, nsplit, stream_);
        if (brc != APTAI_OK) return brc;
        if (ga.p[i].raster_gm < 0) ga.p[i].raster_gm = (d->a_kmajor && d->b_kmajor) ? 8 : 0;     // see aptai_gemm_bf16
        APTAI_REQUIRE(nbatch == 1 && nsplit == 1 && !d->accumulate, "aptai_gemm_bf16_grouped: problem %d: no batching, split-K or accumulate", i);
        APTAI_REQUIRE(d->a_kmajor == descs[0].a_kmajor && d->b_kmajor == descs[0].b_kmajor && (d->out_f32 != 0) == (descs[0].out_f32 != 0),
                      "aptai_gemm_bf16_grouped: problem %d: all problems must share the operand layout and output type", i);
        total += ga.p[i].tiles_m * ga.p[i].tiles_n;
        ga.tile_end[i] = total;
    }
    ga.n = n;
    ga.total = total;
    const bool f32 = descs[0].out_f32 != 0, akm = descs[0].a_kmajor != 0, bkm = descs[0].b_kmajor != 0;
    APTAI_REQUIRE(!(akm && !bkm), "aptai_gemm_bf16_grouped: A K-major with B K-contiguous is not built");
#define APTAI_GROUPED(AK, BK_, F)                                                                                      \
    do {                                                                                                              \
        auto kern = gemm_grouped_kernel<AK, BK_, F>;                                                                  \
        constexpr int smem = smem_for<AK, BK_>();                                                                     \
        static bool attr_set = false;                                                                                 \
        if (!attr_set) {                                                                                              \
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);                  \
            attr_set = true;                                                                                          \
        }                                                                                                             \
        APTAI_LAUNCH(kern, dim3(total), dim3(NTHREADS), smem, stream, ga);                                            \
    } while (0)
    if (!akm && !bkm) { if (f32) APTAI_GROUPED(false, false, true); else APTAI_GROUPED(false, false, false); }
    else if (!akm && bkm) { if (f32) APTAI_GROUPED(false, true, true); else APTAI_GROUPED(false, true, false); }
    else { if (f32) APTAI_GROUPED(true, true, true); else APTAI_GROUPED(true, true, false); }
#undef APTAI_GROUPED
    APTAI_CHECK_LAUNCH("gemm_grouped_kernel");
    return APTAI_OK;
}

extern "C" int64_t aptai_gemm_sk_workspace_bytes(void) { return sk_workspace_bytes(); }

extern "C" int aptai_gemm_sk_status(void* sk_workspace, void* stream, int* status_out) {
    APTAI_REQUIRE(sk_workspace != nullptr && status_out != nullptr, "aptai_gemm_sk_status: null pointer");
    unsigned v = 0;
    if (hipMemcpyAsync(&v, sk_workspace, 4, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
        hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
        APTAI_FAIL(APTAI_ERR_LAUNCH, "aptai_gemm_sk_status: reading the status word failed");
    if (v != 0 && hipMemsetAsync(sk_workspace, 0, 4, (hipStream_t)stream) != hipSuccess)
        APTAI_FAIL(APTAI_ERR_LAUNCH, "aptai_gemm_sk_status: clearing the status word failed");
    *status_out = (int)v;
    return APTAI_OK;
}

extern "C" int64_t aptai_gemm_workspace_bytes(int64_t M, int64_t N, int split_k) {
    return (int64_t)(split_k > 0 ? split_k : 1) * M * N * 4;
}
