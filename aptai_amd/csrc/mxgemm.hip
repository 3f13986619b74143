// MX block-scaled FP8 GEMM for the FROZEN, inference-only encoder of Force_APTAI (BASELINE configs[4]: "fp8 MFMA weights"), gfx950.
//
//   C[M,N] = A[M,K] . B[N,K]^T      A, B: OCP FP8 E4M3 elements, one E8M0 scale per 32 consecutive k (OCP MX "MXFP8")
//
// Only the block-scaled matrix instruction runs FP8 at twice the bf16 rate on gfx950 (v_mfma_scale_f32_32x32x64_f8f6f4: 65 536
// MACs in the cycles the bf16 32x32x16 form needs for 32 768; the plain fp8 MFMA issues at the bf16 rate - guide, "Matrix cores").
// Operand map of the 64-deep step, established with exact-integer probes on the hardware (tools/mx_operand_map.py): lane l = (row
// l & 31, half h = l >> 5) holds k in [16 h, 16 h + 16) in its first 16 bytes and k in [32 + 16 h, 48 + 16 h) in its second 16
// bytes, and the scale byte supplied by lane (row, h) applies to the k block [32 h, 32 h + 32) of that row - i.e. to the first
// 16 bytes of BOTH halves for h = 0 and to the second 16 bytes of both for h = 1.  Fragments are therefore read as two 16-byte
// pieces 32 bytes apart, so that memory-contiguous 32-element blocks meet their own E8M0 scale.
//
// Kernel: 128 x 128 x 128 tile, 4 waves (2 x 2), wave tile 64 x 64 = 2 x 2 MFMA tiles, two 64-deep MFMA steps per K-tile;
// operands staged global -> registers -> LDS (144-byte row pitch: conflict-free 16-byte fragment reads), next K-tile's loads
// in flight under the current tile's MFMAs; operands swapped (D^T = B A^T) so a lane owns 4 consecutive output columns;
// epilogue: + bias, GELU, + residual, bf16 store.  Quantisation (bf16 -> MXFP8) is its own HBM-bound kernel: 4 lanes per
// 32-element block, scale = 2^(floor(log2 amax) - 8), elements saturated to +-448.
#include <stdlib.h>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) int i32x8;

constexpr int MX_BM = 128, MX_BN = 128, MX_BK = 128;      // BK in elements = bytes
constexpr int MX_PITCH = MX_BK + 16;                      // LDS row pitch in bytes
constexpr int MX_TILE_BYTES = MX_BM * MX_PITCH;           // 18 432
constexpr int MX_SC_BYTES = MX_BM * 4;                    // 4 scale bytes per row and K-tile
constexpr int MX_STAGE = 2 * MX_TILE_BYTES + 2 * MX_SC_BYTES;

struct MxArgs {
    const uint8_t* A; const uint8_t* As; long lda, ldas;   // elements [M][lda], scales [M][ldas] (one byte per 32 k)
    const uint8_t* B; const uint8_t* Bs; long ldb, ldbs;
    bf16_t* C; long ldc;
    const float* bias; const bf16_t* residual; long ldr;
    int M, N, K, gelu;
    uint8_t* Cq; uint8_t* Cs; long ldcq, ldcs;            // LDS-DMA kernel only: MXFP8 output (elements, scales) instead of bf16 C
};

__global__ __launch_bounds__(256, 2) void mxgemm_kernel(MxArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (g.N + MX_BN - 1) / MX_BN;
    const int m0 = (blockIdx.x / tiles_n) * MX_BM, n0 = (blockIdx.x % tiles_n) * MX_BN;
    // staging map: 4 x 16 bytes per thread and operand: row = 32 i + (tid >> 3), 16-byte chunk tid & 7
    const int srow = tid >> 3, sch = tid & 7;
    u32x4 ra[4], rb[4];
    uint32_t rs = 0;                                        // threads 0..127: A scales of row tid; 128..255: B scales of row tid - 128
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ar = m0 + 32 * i + srow, br = n0 + 32 * i + srow;
            ra[i] = ar < g.M ? *(const u32x4*)(g.A + (long)ar * g.lda + k0 + sch * 16) : (u32x4){0u, 0u, 0u, 0u};
            rb[i] = br < g.N ? *(const u32x4*)(g.B + (long)br * g.ldb + k0 + sch * 16) : (u32x4){0u, 0u, 0u, 0u};
        }
        if (tid < 128) {
            const int r = m0 + tid;
            rs = r < g.M ? *(const uint32_t*)(g.As + (long)r * g.ldas + (k0 >> 5)) : 0x7f7f7f7fu;
        } else {
            const int r = n0 + tid - 128;
            rs = r < g.N ? *(const uint32_t*)(g.Bs + (long)r * g.ldbs + (k0 >> 5)) : 0x7f7f7f7fu;
        }
    };
    auto stage = [&](char* st) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *(u32x4*)(st + (32 * i + srow) * MX_PITCH + sch * 16) = ra[i];
            *(u32x4*)(st + MX_TILE_BYTES + (32 * i + srow) * MX_PITCH + sch * 16) = rb[i];
        }
        *(uint32_t*)(st + 2 * MX_TILE_BYTES + tid * 4) = rs;
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16)(0.f);
    const int nk = g.K / MX_BK;
    fetch(0);
    const int r31 = lane & 31, kb = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        char* st = smem + (kt & 1) * MX_STAGE;
        stage(st);
        __syncthreads();
        if (kt + 1 < nk) fetch((kt + 1) * MX_BK);
        const char* sA = st;
        const char* sB = st + MX_TILE_BYTES;
        const uint8_t* scA = (const uint8_t*)(st + 2 * MX_TILE_BYTES);
        const uint8_t* scB = scA + MX_SC_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {                        // two 64-deep MFMA steps; this lane's k block = 2 ks + kb
            i32x8 af[2], bfr[2];
            int sa[2], sb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = wm * 64 + i * 32 + r31;
                const u32x4 lo = *(const u32x4*)(sA + row * MX_PITCH + ks * 64 + kb * 16);
                const u32x4 hi = *(const u32x4*)(sA + row * MX_PITCH + ks * 64 + 32 + kb * 16);
                af[i] = (i32x8){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
                sa[i] = scA[row * 4 + 2 * ks + kb];
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = wn * 64 + j * 32 + r31;
                const u32x4 lo = *(const u32x4*)(sB + row * MX_PITCH + ks * 64 + kb * 16);
                const u32x4 hi = *(const u32x4*)(sB + row * MX_PITCH + ks * 64 + 32 + kb * 16);
                bfr[j] = (i32x8){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
                sb[j] = scB[row * 4 + 2 * ks + kb];
            }
            // D^T = B A^T: first operand rows = output columns n, second operand = output rows m (the lane's column)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(bfr[j], af[i], acc[i][j], 0, 0, 0, sb[j], 0, sa[i]);
        }
        // double-buffered stages: the next iteration writes the OTHER stage; the barrier at its top orders those writes behind
        // every wave's reads of it two iterations ago
    }
    // ---- epilogue: lane owns output row m = .. + r31 and columns n = 8 g + 4 kb + (0..3) of each 32-wide MFMA tile
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + wm * 64 + i * 32 + r31;
        if (m >= g.M) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int n = n0 + wn * 64 + j * 32 + 8 * gq + 4 * kb;
                if (n >= g.N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = acc[i][j][4 * gq + e];
                    if (g.bias) x += g.bias[n + e];
                    if (g.gelu) x = gelu_fast(x);
                    v[e] = x;
                }
                if (g.residual) {
                    const u32x2 r = *(const u32x2*)(g.residual + (long)m * g.ldr + n);
                    v[0] += lo_bf(r[0]); v[1] += hi_bf(r[0]); v[2] += lo_bf(r[1]); v[3] += hi_bf(r[1]);
                }
                *(u32x2*)(g.C + (long)m * g.ldc + n) = (u32x2){pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
            }
    }
}

__device__ __forceinline__ uint32_t pack4_fp8(float a, float b, float c, float d) {
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);        // bytes 0, 1
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);             // bytes 2, 3
    return (uint32_t)w;
}

// ---- round 3: the same tile and the same MFMA loop fed by LDS-DMA (global_load_lds, 16 B / lane) into an unpadded, XOR-swizzled
// LDS image, two stages, one barrier per K-tile, and a coalesced epilogue through LDS.  Why it matters here: the bf16 GEMMs of this
// library are bound by operand delivery (~50 KB/us per CU, DESIGN.md sections 8 and 9); a K-tile of 128 fp8 elements is the 32 KB the bf16
// 128-tile stages for 64 elements, and the scaled MFMA retires it in the same cycles - so the same delivery rate carries twice the MACs.
// LDS image of an operand tile: row r = 128 bytes = 8 chunks of 16 B, chunk c at slot c ^ (r & 7) (the 32 rows of a fragment read
// cover the 8 slots evenly); the hardware writes lane-linear, so thread t of instruction `it` FETCHES chunk ((t & 7) ^ (row & 7)) of
// row (it * 256 + t) >> 3.  Scales: 4 bytes per row and K-tile, staged by 4-byte LDS-DMA (waves 0-1: A rows, waves 2-3: B rows).
constexpr int MXD_TILE = MX_BM * MX_BK;                   // 16 384 B per operand tile, unpadded
constexpr int MXD_STAGE = 2 * MXD_TILE + 2 * MX_SC_BYTES; // 33 792 B
constexpr int MXD_SMEM = 2 * MXD_STAGE;                   // 67 584 B: two blocks per CU
constexpr int MXD_EPI_PITCH = MX_BN * 4 + 16;             // fp32 row of 128 columns + pad (conflict-free 16-B row writes)
static_assert(MX_BM * MXD_EPI_PITCH <= MXD_SMEM, "the epilogue tile must fit the ring");

__global__ __launch_bounds__(256, 2) void mxgemm_dma_kernel(MxArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (g.N + MX_BN - 1) / MX_BN;
    const int m0 = (blockIdx.x / tiles_n) * MX_BM, n0 = (blockIdx.x % tiles_n) * MX_BN;
    // staging sources (rows outside the problem are clamped: their products land in accumulator rows / columns nobody stores)
    const uint8_t* pa[4];
    const uint8_t* pb[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int cid = it * 256 + tid, row = cid >> 3, c = (cid & 7) ^ (row & 7);
        int ar = m0 + row, br = n0 + row;
        ar = ar < g.M ? ar : g.M - 1;
        br = br < g.N ? br : g.N - 1;
        pa[it] = g.A + (long)ar * g.lda + c * 16;
        pb[it] = g.B + (long)br * g.ldb + c * 16;
    }
    const uint8_t* ps;                                      // this thread's scale dword: waves 0-1 A rows, waves 2-3 B rows
    {
        const int r = tid & 127;
        int ar = m0 + r, br = n0 + r;
        ar = ar < g.M ? ar : g.M - 1;
        br = br < g.N ? br : g.N - 1;
        ps = wave < 2 ? g.As + (long)ar * g.ldas : g.Bs + (long)br * g.ldbs;
    }
    auto stage = [&](char* st) {                            // one K-tile; every call advances the sources by 128 k
        char* dst = st + wave * 1024;                       // wave-uniform; the hardware adds lane * 16
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            __builtin_amdgcn_global_load_lds(GLB_PTR(pa[it]), LDS_PTR(dst + it * 4096), 16, 0, 0);
            pa[it] += MX_BK;
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            __builtin_amdgcn_global_load_lds(GLB_PTR(pb[it]), LDS_PTR(dst + MXD_TILE + it * 4096), 16, 0, 0);
            pb[it] += MX_BK;
        }
        __builtin_amdgcn_global_load_lds(GLB_PTR(ps), LDS_PTR(st + 2 * MXD_TILE + wave * 256), 4, 0, 0);   // + lane * 4
        ps += MX_BK / 32;
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16)(0.f);
    const int nk = g.K / MX_BK;
    const int r31 = lane & 31, kb = lane >> 5;
    if (nk > 0) stage(smem);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                    // stage kt has landed for every wave; the other slot is free
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(smem + (cur ^ 1) * MXD_STAGE);
        const char* sA = smem + cur * MXD_STAGE;
        const char* sB = sA + MXD_TILE;
        const uint8_t* scA = (const uint8_t*)(sA + 2 * MXD_TILE);
        const uint8_t* scB = scA + MX_SC_BYTES;
        i32x8 af[2][2], bfr[2][2];
        int sa[2][2], sb[2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {                    // all fragment reads of the K-tile ahead of its MFMAs
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = wm * 64 + i * 32 + r31, sw = row & 7;
                const u32x4 lo = *(const u32x4*)(sA + row * 128 + (((ks * 4 + kb) ^ sw) << 4));
                const u32x4 hi = *(const u32x4*)(sA + row * 128 + (((ks * 4 + 2 + kb) ^ sw) << 4));
                af[ks][i] = (i32x8){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
                sa[ks][i] = scA[row * 4 + 2 * ks + kb];
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = wn * 64 + j * 32 + r31, sw = row & 7;
                const u32x4 lo = *(const u32x4*)(sB + row * 128 + (((ks * 4 + kb) ^ sw) << 4));
                const u32x4 hi = *(const u32x4*)(sB + row * 128 + (((ks * 4 + 2 + kb) ^ sw) << 4));
                bfr[ks][j] = (i32x8){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
                sb[ks][j] = scB[row * 4 + 2 * ks + kb];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(bfr[ks][j], af[ks][i], acc[i][j], 0, 0, 0, sb[ks][j], 0, sa[ks][i]);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();                                        // the ring becomes the epilogue's staging tile
    // ---- epilogue through LDS: acc[i][j][4 gq + e] = C[m = wm 64 + 32 i + r31][n = wn 64 + 32 j + 8 gq + 4 kb + e] (D^T = B A^T);
    // staged as fp32 rows, read back as 8 consecutive columns of one row per thread and pass (16 threads cover a 512-byte row), so
    // bias, residual and C are 16 / 32-byte-per-lane row segments (the first form stored 8 bytes per lane at a row stride)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                *(f32x4*)(smem + (wm * 64 + i * 32 + r31) * MXD_EPI_PITCH + (wn * 64 + j * 32 + 8 * gq + 4 * kb) * 4) =
                    (f32x4){acc[i][j][4 * gq], acc[i][j][4 * gq + 1], acc[i][j][4 * gq + 2], acc[i][j][4 * gq + 3]};
    __syncthreads();
    const int cl = (tid & 15) * 8, n = n0 + cl;
    if (n < g.N) {
        float bias8[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) bias8[r] = g.bias ? g.bias[n + r] : 0.f;      // (N % 8 == 0 on this path: checked by the launcher)
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const int ml = pass * 16 + (tid >> 4), m = m0 + ml;
            if (m >= g.M) continue;
            const f32x4 v0 = *(const f32x4*)(smem + ml * MXD_EPI_PITCH + cl * 4);
            const f32x4 v1 = *(const f32x4*)(smem + ml * MXD_EPI_PITCH + cl * 4 + 16);
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                v[r] += bias8[r];
                if (g.gelu) v[r] = gelu_fast(v[r]);
            }
            if (g.residual) {
                const u32x4 rq = *(const u32x4*)(g.residual + (long)m * g.ldr + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[2 * r] += lo_bf(rq[r]); v[2 * r + 1] += hi_bf(rq[r]); }
            }
            const u32x4 packed = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
            if (g.Cq == nullptr) {
                *(u32x4*)(g.C + (long)m * g.ldc + n) = packed;
                continue;
            }
            // MXFP8 output (the next Linear's A operand): exactly what aptai_mx_quantize_bf16 makes of the bf16 result - the values are
            // rounded to bf16 first, a 32-column block is four neighbouring threads of the row (N % 32 == 0 on this path)
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[2 * r] = lo_bf(packed[r]); v[2 * r + 1] = hi_bf(packed[r]); }
            float amax = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) amax = fmaxf(amax, fabsf(v[r]));
            amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
            amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
            int e = (int)((__float_as_uint(amax) >> 23) & 0xffu) - 8;
            e = e < 1 ? 1 : (e > 254 ? 254 : e);
            const float inv = __uint_as_float((uint32_t)(254 - e) << 23);
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = __builtin_amdgcn_fmed3f(v[r] * inv, -448.f, 448.f);
            *(u32x2*)(g.Cq + (long)m * g.ldcq + n) = (u32x2){pack4_fp8(v[0], v[1], v[2], v[3]), pack4_fp8(v[4], v[5], v[6], v[7])};
            if ((tid & 3) == 0) g.Cs[(long)m * g.ldcs + (n >> 5)] = (uint8_t)e;
        }
    }
}

// ---- bf16 [rows][K] -> MXFP8: elements [rows][ldq] (1 byte), scales [rows][lds] (1 byte per 32 k).  4 lanes per block of 32.
__global__ __launch_bounds__(256) void mx_quantize_kernel(const bf16_t* __restrict__ x, long ldx, uint8_t* __restrict__ q, long ldq,
                                                          uint8_t* __restrict__ s, long lds, long rows, int K) {
    const long per_row = K / 8;                                 // 8-element units per row
    const long total = rows * per_row;
    for (long u = (long)blockIdx.x * blockDim.x + threadIdx.x; u < ((total + 63) / 64) * 64; u += (long)gridDim.x * blockDim.x) {
        const bool live = u < total;
        const long r = live ? u / per_row : 0;
        const int c8 = live ? (int)(u % per_row) : 0;
        float v[8];
        u32x4 in = {0u, 0u, 0u, 0u};
        if (live) in = *(const u32x4*)(x + r * ldx + c8 * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[2 * j] = lo_bf(in[j]); v[2 * j + 1] = hi_bf(in[j]); }
        float amax = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(v[j]));
        amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
        amax = fmaxf(amax, __shfl_xor(amax, 2, 64));            // the 4 lanes of a block are consecutive (K % 32 == 0)
        // shared exponent = floor(log2 amax) - 8 (E4M3: emax = 8); biased E8M0 byte; amax = 0 or subnormal -> smallest scale used
        int e = (int)((__float_as_uint(amax) >> 23) & 0xffu) - 8;
        e = e < 1 ? 1 : (e > 254 ? 254 : e);
        const float inv = __uint_as_float((uint32_t)(254 - e) << 23);          // 2^-(e - 127)
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = __builtin_amdgcn_fmed3f(v[j] * inv, -448.f, 448.f);
        if (live) {
            *(u32x2*)(q + r * ldq + c8 * 8) = (u32x2){pack4_fp8(v[0], v[1], v[2], v[3]), pack4_fp8(v[4], v[5], v[6], v[7])};
            if ((c8 & 3) == 0) s[r * lds + (c8 >> 2)] = (uint8_t)e;
        }
    }
}

}  // namespace

extern "C" int aptai_mx_quantize_bf16(const void* x, int64_t ldx, void* q, int64_t ldq, void* scales, int64_t lds, int64_t rows,
                                      int64_t K, void* stream) {
    APTAI_REQUIRE(x && q && scales && rows > 0 && K > 0 && K % 32 == 0, "aptai_mx_quantize_bf16: K=%ld must be a positive multiple of 32", (long)K);
    APTAI_REQUIRE(ldx % 8 == 0 && ldq % 8 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)q % 8 == 0),
                  "aptai_mx_quantize_bf16: rows must keep 16-byte (input) / 8-byte (output) alignment");
    const long units = rows * (K / 8);
    long blocks = ceil_div(units, 256);
    if (blocks > 4096) blocks = 4096;
    APTAI_LAUNCH(mx_quantize_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (long)ldx, (uint8_t*)q,
                 (long)ldq, (uint8_t*)scales, (long)lds, (long)rows, (int)K);
    APTAI_CHECK_LAUNCH("mx_quantize_kernel");
    return APTAI_OK;
}

static int mx_launch(MxArgs& g, bool need_dma, hipStream_t stream);

// C as MXFP8 (elements [M][ldcq], scales [M][ldcs]): the GEMM + bias (+ GELU) whose result is the next MX GEMM's A operand, bit-identical
// to aptai_gemm_mxfp8 followed by aptai_mx_quantize_bf16 without the bf16 round trip through HBM and the quantiser's launch
extern "C" int aptai_gemm_mxfp8_mxout(const void* A, const void* A_scales, int64_t lda, int64_t ldas, const void* B, const void* B_scales,
                                      int64_t ldb, int64_t ldbs, void* Cq, int64_t ldcq, void* C_scales, int64_t ldcs, const float* bias,
                                      int gelu, int64_t M, int64_t N, int64_t K, void* stream) {
    APTAI_REQUIRE(A && A_scales && B && B_scales && Cq && C_scales && M > 0 && N > 0 && K > 0, "aptai_gemm_mxfp8_mxout: bad arguments");
    APTAI_REQUIRE(K % MX_BK == 0 && N % 32 == 0, "aptai_gemm_mxfp8_mxout: K=%ld must be a multiple of %d and N=%ld of 32", (long)K, MX_BK, (long)N);
    APTAI_REQUIRE(lda % 16 == 0 && ldb % 16 == 0 && ldas % 4 == 0 && ldbs % 4 == 0 && ldcq % 8 == 0 && ldcs >= N / 32,
                  "aptai_gemm_mxfp8_mxout: leading dimensions must keep the vector accesses aligned");
    APTAI_REQUIRE(((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && ((uintptr_t)Cq % 8 == 0) && ((uintptr_t)A_scales % 4 == 0) &&
                      ((uintptr_t)B_scales % 4 == 0) && (!bias || (uintptr_t)bias % 4 == 0), "aptai_gemm_mxfp8_mxout: operands must be 16-byte aligned (scales 4)");
    MxArgs g;
    memset(&g, 0, sizeof(g));
    g.A = (const uint8_t*)A; g.As = (const uint8_t*)A_scales; g.lda = lda; g.ldas = ldas;
    g.B = (const uint8_t*)B; g.Bs = (const uint8_t*)B_scales; g.ldb = ldb; g.ldbs = ldbs;
    g.Cq = (uint8_t*)Cq; g.Cs = (uint8_t*)C_scales; g.ldcq = ldcq; g.ldcs = ldcs; g.bias = bias;
    g.M = (int)M; g.N = (int)N; g.K = (int)K; g.gelu = gelu;
    return mx_launch(g, true, (hipStream_t)stream);
}

extern "C" int aptai_gemm_mxfp8(const void* A, const void* A_scales, int64_t lda, int64_t ldas, const void* B, const void* B_scales,
                                int64_t ldb, int64_t ldbs, void* C, int64_t ldc, const float* bias, int gelu, const void* residual,
                                int64_t ldr, int64_t M, int64_t N, int64_t K, void* stream) {
    APTAI_REQUIRE(A && A_scales && B && B_scales && C && M > 0 && N > 0 && K > 0, "aptai_gemm_mxfp8: bad arguments");
    APTAI_REQUIRE(K % MX_BK == 0, "aptai_gemm_mxfp8: K=%ld must be a multiple of %d", (long)K, MX_BK);
    APTAI_REQUIRE(N % 4 == 0 && ldc % 4 == 0 && lda % 16 == 0 && ldb % 16 == 0 && ldas % 4 == 0 && ldbs % 4 == 0,
                  "aptai_gemm_mxfp8: leading dimensions must keep the vector accesses aligned");
    APTAI_REQUIRE(((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && ((uintptr_t)C % 8 == 0) && ((uintptr_t)A_scales % 4 == 0) &&
                      ((uintptr_t)B_scales % 4 == 0), "aptai_gemm_mxfp8: operands must be 16-byte aligned (scales 4)");
    if (residual) APTAI_REQUIRE(ldr % 4 == 0 && (uintptr_t)residual % 8 == 0, "aptai_gemm_mxfp8: residual must be 8-byte aligned");
    MxArgs g;
    memset(&g, 0, sizeof(g));
    g.A = (const uint8_t*)A; g.As = (const uint8_t*)A_scales; g.lda = lda; g.ldas = ldas;
    g.B = (const uint8_t*)B; g.Bs = (const uint8_t*)B_scales; g.ldb = ldb; g.ldbs = ldbs;
    g.C = (bf16_t*)C; g.ldc = ldc; g.bias = bias; g.residual = (const bf16_t*)residual; g.ldr = ldr;
    g.M = (int)M; g.N = (int)N; g.K = (int)K; g.gelu = gelu;
    static const bool dma_off = getenv("APTAI_MX_DMA") && atoi(getenv("APTAI_MX_DMA")) == 0;
    // the LDS-DMA kernel wherever its 16-byte row segments are aligned (every shape of the encoder); APTAI_MX_DMA=0 = the round-2 kernel (A/B)
    const bool dma_ok = !dma_off && N % 8 == 0 && ldc % 8 == 0 && (uintptr_t)C % 16 == 0 && (!residual || (ldr % 8 == 0 && (uintptr_t)residual % 16 == 0)) &&
                        (!bias || (uintptr_t)bias % 4 == 0);
    return mx_launch(g, dma_ok, (hipStream_t)stream);
}

static int mx_launch(MxArgs& g, bool dma, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)mxgemm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * MX_STAGE) != hipSuccess ||
            hipFuncSetAttribute((const void*)mxgemm_dma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MXD_SMEM) != hipSuccess)
            APTAI_FAIL(APTAI_ERR_LAUNCH, "aptai_gemm_mxfp8: cannot raise the dynamic LDS limit");
        attr_set = true;
    }
    const long tiles = ceil_div((long)g.M, (long)MX_BM) * ceil_div((long)g.N, (long)MX_BN);
    if (dma) {
        APTAI_LAUNCH(mxgemm_dma_kernel, dim3((unsigned)tiles), dim3(256), MXD_SMEM, stream, g);
        APTAI_CHECK_LAUNCH("mxgemm_dma_kernel");
        return APTAI_OK;
    }
    APTAI_LAUNCH(mxgemm_kernel, dim3((unsigned)tiles), dim3(256), 2 * MX_STAGE, stream, g);
    APTAI_CHECK_LAUNCH("mxgemm_kernel");
    return APTAI_OK;
}
