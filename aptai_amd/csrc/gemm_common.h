// Shared pieces of the bf16 MFMA GEMM family (gemm.hip, gemm_t4.hip): the kernel argument block, the per-flag-word epilogue, LDS-DMA
// staging sources, fragment reads, tile-index remaps.  Device code only; everything is forceinline / constexpr / templates.
#pragma once
#include "common.h"

namespace aptai_gemm {


#ifndef APTAI_GEMM_HOIST
#define APTAI_GEMM_HOIST 1
#endif
constexpr int BM = 128, BN = 128, BK = 64;
constexpr int NTHREADS = 256;
constexpr int STAGE_BYTES = (BM * BK + BN * BK) * 2;   // 32 KiB
constexpr int OPER_BYTES = BM * BK * 2;                // 16 KiB per operand tile
constexpr int EPI_PITCH = BN * 4 + 16;                 // fp32 epilogue tile row pitch (bytes)
constexpr int SMEM_BYTES = BM * EPI_PITCH > 2 * STAGE_BYTES ? BM * EPI_PITCH : 2 * STAGE_BYTES;
#ifndef APTAI_GEMM_RING5
#define APTAI_GEMM_RING5 1
#endif
#ifndef APTAI_GEMM_M64_ASM
#define APTAI_GEMM_M64_ASM 1
#endif
#ifndef APTAI_GEMM192_ASM
#define APTAI_GEMM192_ASM 1
#endif
constexpr int RING_HALF_BYTES = (BM + BN) * 32 * 2;    // 16 KiB: 32 k-rows of both operands
constexpr int SMEM_RING_BYTES = 5 * RING_HALF_BYTES;   // 80 KiB: two blocks per CU use all of the 160 KiB
static_assert(SMEM_RING_BYTES >= BM * EPI_PITCH, "the epilogue tile must fit the ring");
template <bool A_KM, bool B_KM> constexpr int smem_for() { return (A_KM && B_KM && APTAI_GEMM_RING5) ? SMEM_RING_BYTES : SMEM_BYTES; }

struct GemmArgs {
    const bf16_t* A; long lda;
    const bf16_t* B; long ldb;
    void* C; long ldc;
    int M, N, K;
    const float* bias;
    const bf16_t* residual; long ldr;
    bf16_t* out_pre;            // pre-activation copy (same ld as C)
    const bf16_t* aux; long ldaux;
    int flags;
    uint32_t seed0, seed1, thr16;
    const uint32_t* salt;
    float dscale;
    float alpha;
    int ktiles_per_split;
    long slab_stride;           // elements between split-K slabs (fp32 out only)
    int tiles_m, tiles_n;
    int raster_gm;              // tile rows per raster group (raster2d); 0 = row-major walk
    long hash_ld; int hash_n0;  // dropout element index = m * hash_ld + hash_n0 + n: a launch that covers columns [hash_n0, hash_n0 + N) of a
                                // wider output (aptai_gemm_bf16 splits some) draws the masks of the whole one
    int colscale_n; float colscale;   // columns [0, colscale_n) of the bf16 output are multiplied by colscale (after alpha / bias)
    int split_pieces;                 // APTAI_EPI_SPLIT_OUT: 3 or 6 bf16 pieces per fp32 result (C is bf16, ldc in bf16 elements)
    int split_bcol;                   // ... columns >= split_bcol in the weight-side piece order
#ifdef APTAI_EXP_STAGGER
    int exp_sleep;                    // development (tools/ab builds): s_sleep(127) iterations at the start of every SECOND block to arrive on a CU
    unsigned* exp_cu_count;           // ... per-CU arrival counters (2048 words, zeroed by the caller before the launch), or null = every block sleeps
#endif
    // 2-level batching: blockIdx.y = outer * nb_inner + inner; element offsets per level
    int nb_inner;
    long sA[2], sB[2], sC[2], sBias[2], sR[2], sAux[2];
};



__device__ __forceinline__ int km_swz(int krow) { return ((krow & 3) << 2) | ((krow >> 2) & 3); }

// gelu(x) and gelu'(x) together (shared sigmoid): forward epilogues that save the activation derivative for the backward
__device__ __forceinline__ void gelu_fast_both(float x, float& y, float& dy) {
    const float xc = __builtin_amdgcn_fmed3f(x, -7.0f, 7.0f);
    const float x2 = xc * xc;
    const float s = gelu_sig(xc, x2);
    float q = fmaf(5.0f * APTAI_GELU_A5, x2, 3.0f * APTAI_GELU_A3);
    q = fmaf(q, x2, APTAI_GELU_A1);
    y = x * s;
    dy = fmaf(fmaf(-s, s, s), xc * q, s);
}

// Everything behind alpha/bias for 8 consecutive columns (n..n+7) of output row m, bf16-output kernels: optional copy of
// the pre-activation (or, with EPI_PRE_DGELU, of dropmask * gelu'(pre-activation): the factor the backward multiplies by,
// so the dgrad epilogue is ONE multiply per element instead of a dropout hash and a gelu'), GELU, dropout, x gelu'(aux) or
// x aux, + residual, packed 16-byte store.
//
// The flag word.  Measured with in-kernel stamps (tools/gemm256_stamps.py): with the flags tested at RUN time this function is ~100
// issued instructions and ~15 scalar branches per 8 outputs, and a 256 x 256 tile's epilogue took 7.5 us of 27 us at K = 768 with NO
// option set.  So the epilogues branch ONCE per block (epi_dispatch, block-uniform) into a body compiled for the exact flag word of the
// launch - the words the model's own launches use are listed there - and everything else runs the run-time form (FM = -1).  The word is
// the descriptor's flags plus two internal bits for what the descriptor says with pointers / counts.
constexpr int EPX_PRE = 1 << 16;                               // out_pre != nullptr
constexpr int EPX_CS = 1 << 17;                                // colscale_n > 0
constexpr int EPX_RUNTIME = 1 << 18;                           // set by the host under APTAI_EPI_RUNTIME=1 (A/B): matches no compiled word
__device__ __forceinline__ int epi_flag_word(const GemmArgs& g) {
    return g.flags | (g.out_pre ? EPX_PRE : 0) | (g.colscale_n > 0 ? EPX_CS : 0);
}
template <int FM> struct EpiWord { static constexpr int value = FM; };
#define APTAI_EPI_WORDS(X)                                                                                                  \
    X(0)                                                                  /* dgrads */                                      \
    X(APTAI_EPI_BIAS)                                                                                                       \
    X(APTAI_EPI_BIAS | EPX_CS)                                            /* q|k|v projection */                            \
    X(APTAI_EPI_RESIDUAL)                                                 /* dgrads joining the residual gradient */        \
    X(APTAI_EPI_BIAS | APTAI_EPI_RESIDUAL)                                /* out-proj / FFN2, evaluation */                 \
    X(APTAI_EPI_BIAS | APTAI_EPI_RESIDUAL | APTAI_EPI_DROPOUT)            /* out-proj / FFN2, training */                   \
    X(APTAI_EPI_BIAS | APTAI_EPI_DROPOUT)                                 /* feature projection */                          \
    X(APTAI_EPI_GELU)                                                     /* frozen conv stack */                           \
    X(APTAI_EPI_BIAS | APTAI_EPI_GELU)                                    /* FFN1, evaluation */                            \
    X(APTAI_EPI_GELU | EPX_PRE)                                           /* trainable conv stack */                        \
    X(APTAI_EPI_BIAS | APTAI_EPI_GELU | EPX_PRE)                                                                            \
    X(APTAI_EPI_BIAS | APTAI_EPI_GELU | APTAI_EPI_DROPOUT | APTAI_EPI_PRE_DGELU | EPX_PRE)   /* FFN1, training */           \
    X(APTAI_EPI_BIAS | APTAI_EPI_GELU | APTAI_EPI_PRE_DGELU | EPX_PRE)    /* FFN1, training, activation dropout 0 */        \
    X(APTAI_EPI_MUL_AUX)                                                  /* FFN2 dgrad */                                  \
    X(APTAI_EPI_BIAS | APTAI_EPI_GELU | APTAI_EPI_DROPOUT)                /* FFN1, training, derivative not saved */        \
    X(APTAI_EPI_DGELU | APTAI_EPI_RESIDUAL)                               /* trainable conv stack, dgrad joining a tap */   \
    X(APTAI_EPI_DGELU)
template <class F>
__device__ __forceinline__ void epi_dispatch(const int fx, F&& f) {
#define APTAI_EPI_CASE(W)  \
    if (fx == (W)) {        \
        f(EpiWord<(W)>{});  \
        return;             \
    }
    APTAI_EPI_WORDS(APTAI_EPI_CASE)
#undef APTAI_EPI_CASE
    f(EpiWord<-1>{});
}
template <int FM>
__device__ __forceinline__ void epilogue_chunk(float (&v)[8], const GemmArgs& g, const int flags_rt, const long m, const int n,
                                               const u32x4 auxq, const u32x4 resq, const uint32_t sd0, const uint32_t sd1) {
    const int flags = FM >= 0 ? FM : flags_rt;                 // flags_rt = epi_flag_word(g)
    const bool pre_dgelu = (flags & APTAI_EPI_PRE_DGELU) != 0;
    const bool has_pre = (flags & EPX_PRE) != 0;
    if ((flags & EPX_CS) && n < g.colscale_n) {                // 8-column chunks: colscale_n % 8 == 0 (checked on the host)
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] *= g.colscale;
    }
    float d[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) d[r] = 1.0f;
    if (has_pre && !pre_dgelu)
        *(u32x4*)(g.out_pre + m * g.ldc + n) =
            (u32x4){pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
    if (flags & APTAI_EPI_GELU) {
        if (pre_dgelu) {
#pragma unroll
            for (int r = 0; r < 8; ++r) gelu_fast_both(v[r], v[r], d[r]);
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = gelu_fast(v[r]);
        }
    }
    if (flags & APTAI_EPI_DROPOUT) {
        // the chunk starts at a multiple of 8 (n % 8 == 0, N % 8 == 0: 16-byte stores), so the four pairs share the folded high
        // word of drop_hash_pair and their low words are consecutive: same masks, without 64-bit arithmetic and a quarter-rate
        // 32-bit multiply per pair
        const uint64_t e = (uint64_t)m * (uint64_t)g.hash_ld + (uint64_t)(g.hash_n0 + n);
        const uint32_t e_lo = (uint32_t)(e >> 1), e_hi = (uint32_t)(e >> 33) * 0x85ebca6bu;
#pragma unroll
        for (int r = 0; r < 8; r += 2) {
            const uint32_t hsh = rng_hash((e_lo + (uint32_t)(r >> 1)) ^ e_hi, sd0, sd1);
            const float k0 = (hsh & 0xffffu) >= g.thr16 ? g.dscale : 0.f, k1 = (hsh >> 16) >= g.thr16 ? g.dscale : 0.f;
            v[r] *= k0;
            v[r + 1] *= k1;
            d[r] *= k0;
            d[r + 1] *= k1;
        }
    }
    if (has_pre && pre_dgelu)
        *(u32x4*)(g.out_pre + m * g.ldc + n) =
            (u32x4){pack2bf(d[0], d[1]), pack2bf(d[2], d[3]), pack2bf(d[4], d[5]), pack2bf(d[6], d[7])};
    if (flags & APTAI_EPI_DGELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v[2 * r] *= gelu_fast_grad(lo_bf(auxq[r]));
            v[2 * r + 1] *= gelu_fast_grad(hi_bf(auxq[r]));
        }
    }
    if (flags & APTAI_EPI_MUL_AUX) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[2 * r] *= lo_bf(auxq[r]); v[2 * r + 1] *= hi_bf(auxq[r]); }
    }
    if (flags & APTAI_EPI_RESIDUAL) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[2 * r] += lo_bf(resq[r]); v[2 * r + 1] += hi_bf(resq[r]); }
    }
#ifdef APTAI_EXP_NOSTORE
    if (g.M < 0)
#endif
    *(u32x4*)((bf16_t*)g.C + m * g.ldc + n) =
        (u32x4){pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
}

// ---- global -> LDS staging of one operand tile (1024 x 16-B chunks, 4 per thread).
// stage_src: the per-thread source of staging instruction `it` at K offset 0.  The kernel keeps the 8 pointers (4 per operand)
// in registers and advances them by one K-tile per stage: recomputing `(k0 + krow) * ld` for K-major operands cost two
// v_mul_lo_u32 + one v_mad_u64_u32 (quarter-rate) per load per K-tile.
template <bool KM>
__device__ __forceinline__ const bf16_t* stage_src(const bf16_t* __restrict__ base, long ld, int row0, int rows_total, int it,
                                                   int tid) {
    const int cid = it * NTHREADS + tid;
    if (!KM) {
        const int row = cid >> 3, pc = cid & 7;
        int grow = row0 + row;
        grow = grow < rows_total ? grow : rows_total - 1;
        return base + (long)grow * ld + ((pc ^ (row & 7)) << 3);
    } else {
        const int krow = cid >> 4, pc = cid & 15;
        int col = row0 + ((pc ^ km_swz(krow)) << 3);
        col = col <= rows_total - 8 ? col : rows_total - 8;
        return base + (long)krow * ld + col;
    }
}

// ---- fragment reads (16 rows x 32 k) for MFMA 16x16x32
template <bool KM>
__device__ __forceinline__ bf16x8 read_frag(const char* lds_tile, int row_base, int ks, int lane) {
    if (!KM) {
        const int row = row_base + (lane & 15);
        const int q = ks * 4 + (lane >> 4);
        return *(const bf16x8*)(lds_tile + row * 128 + ((q ^ (row & 7)) << 4));
    } else {
        const int g = lane >> 4, i = lane & 15, qq = i >> 2, p = i & 3;
        const int ch = (row_base >> 3) + (p >> 1);
        const int sub = (p & 1) << 3;
        const int k_lo = ks * 32 + g * 8 + qq;
        const int k_hi = k_lo + 4;
        const char* a0 = lds_tile + k_lo * 256 + ((ch ^ km_swz(k_lo)) << 4) + sub;
        const char* a1 = lds_tile + k_hi * 256 + ((ch ^ km_swz(k_hi)) << 4) + sub;
        short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)a0);
        short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)a1);
        short8v r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, r);
    }
}

// The same K-major fragment through inline asm, in two 4-element halves that the caller completes with combine_tr() after its
// own `s_waitcnt lgkmcnt`.  Why: behind an LDS-DMA the compiler puts `s_waitcnt vmcnt(0)` in front of every
// __builtin_amdgcn_ds_read_tr16_b64 (it cannot tell which LDS bytes the DMA writes), i.e. right after the staging loads of the
// NEXT K-tile have been issued - the block then waits for them before it computes the current one and only the other blocks of
// the CU hide the memory latency.  Plain ds_read_b128 (K-contiguous operands) do not get that wait.  A compiler-inserted
// lgkmcnt for its own reads stays safe next to these: LDS operations return in order, so it can only over-wait.
__device__ __forceinline__ uint32_t lds_u32(const char* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
__device__ __forceinline__ void read_frag_tr_asm(const char* lds_tile, int row_base, int ks, int lane, short4v& lo, short4v& hi) {
    const int g = lane >> 4, i = lane & 15, qq = i >> 2, p = i & 3;
    const int ch = (row_base >> 3) + (p >> 1);
    const int sub = (p & 1) << 3;
    const int k_lo = ks * 32 + g * 8 + qq;
    const int k_hi = k_lo + 4;
    const uint32_t a0 = lds_u32(lds_tile) + (uint32_t)(k_lo * 256 + ((ch ^ km_swz(k_lo)) << 4) + sub);
    const uint32_t a1 = lds_u32(lds_tile) + (uint32_t)(k_hi * 256 + ((ch ^ km_swz(k_hi)) << 4) + sub);
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a0));
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(a1));
}
__device__ __forceinline__ bf16x8 combine_tr(const short4v lo, const short4v hi) {
    return __builtin_bit_cast(bf16x8, (short8v){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
}

// one MFMA operand fragment: K-contiguous -> plain ds_read_b128 (compiler-scheduled); K-major -> the asm pair above, valid only
// after the caller's own `s_waitcnt lgkmcnt`
template <bool KM>
struct Frag {
    bf16x8 v;
    short4v lo, hi;
    __device__ __forceinline__ void read(const char* lds_tile, int row_base, int ks, int lane) {
        if constexpr (KM) read_frag_tr_asm(lds_tile, row_base, ks, lane, lo, hi);
        else v = read_frag<false>(lds_tile, row_base, ks, lane);
    }
    __device__ __forceinline__ bf16x8 get() const {
        if constexpr (KM) return combine_tr(lo, hi);
        else return v;
    }
};

// XCD-aware bijective remap: blocks b, b+8, ... share an XCD; give each XCD a contiguous run of tiles
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// 2-D rasterisation of a (remapped) tile index: groups of `gm` tile rows are walked row-fastest, then along N, so the blocks
// that run TOGETHER on one XCD (its contiguous run of indices, ~64 resident tiles) touch gm A-panels and ~64/gm B-panels instead of
// a few A-panels and EVERY B-panel: for [8192 x 768] x [3072 x 768]^T the row-major walk re-streamed the whole 4.7 MB weight
// matrix past each XCD's 4 MB L2 once per row panel (7.4x the algorithmic bytes on the L2 fabric side, rocprofv3 FETCH_SIZE,
// round 1).  gm = 0 keeps the row-major walk (APTAI_GEMM_RASTER=0, A/B).
__device__ __forceinline__ void raster2d(int bid, int tiles_m, int tiles_n, int gm, int& tile_m, int& tile_n) {
    if (gm <= 1) { tile_m = bid / tiles_n; tile_n = bid % tiles_n; return; }
    const int group = gm * tiles_n;
    const int first = (bid / group) * gm;
    const int rows = tiles_m - first < gm ? tiles_m - first : gm;
    const int r = bid % group;
    tile_m = first + r % rows;
    tile_n = r / rows;
}

// APTAI_EPI_SPLIT_OUT (exact-index mode): 8 consecutive fp32 results of output row m leave as `split_pieces` bf16 pieces in the
// activation-side layout of aptai_split_f32 ([m][N/64][piece][64]), after the erf GELU when the launch carries APTAI_EPI_GELU.
__device__ __forceinline__ void split_out_store(const GemmArgs& g, const int flags, float (&v)[8], const long m, const int n) {
    float h[8], md[8], lw[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        if (flags & APTAI_EPI_GELU) v[r] = gelu_exact(v[r]);
        split3(v[r], h[r], md[r], lw[r]);
    }
    bf16_t* dst = (bf16_t*)g.C + m * g.ldc + (long)(n >> 6) * (64 * g.split_pieces) + (n & 63);
    auto put = [&](int piece, const float (&q)[8]) {
        *(u32x4*)(dst + piece * 64) = (u32x4){pack2bf(q[0], q[1]), pack2bf(q[2], q[3]), pack2bf(q[4], q[5]), pack2bf(q[6], q[7])};
    };
    put(0, h);
    if (n < g.split_bcol) {                              // activation side: hi hi lo | hi hi mid mid hi low
        put(1, h); put(2, md);
        if (g.split_pieces == 6) { put(3, md); put(4, h); put(5, lw); }
    } else {                                             // weight side: hi lo hi | hi mid hi mid low hi
        put(1, md); put(2, h);
        if (g.split_pieces == 6) { put(3, md); put(4, lw); put(5, h); }
    }
}

// 256 x 192 tile kernel (gemm_t4.hip): bf16 output, K-contiguous A, B K-contiguous or K-major; no batching / split-K
int launch_gemm_t4(GemmArgs g, bool b_km, hipStream_t stream);

}  // namespace aptai_gemm
