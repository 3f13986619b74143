// Positional convolution of the wav2vec2 encoder (HF:326-379: grouped Conv1d, k = 128, 16 groups, "same" padding) as a
// dedicated MFMA kernel for 48 channels per group (wav2vec2-base).
//
// As an implicit GEMM (gemm.hip, lda = Cg < K = 128*Cg) row t of the A operand is the 6144 contiguous elements starting at
// frame t of the group-major, zero-gapped copy: consecutive rows overlap in all but 48 elements, yet every K-tile re-stages
// 128 rows x 128 B through the texture path, and 48-channel outputs fill 37.5 % of a 128-wide tile.  Here a block keeps the
// whole input window of its 128 output frames in LDS ONCE (255 frames x 96 B = 24 KB) and reads MFMA A fragments straight out
// of it at the Toeplitz row stride: A[t][k] sits at byte 96 t + 2 k whatever k = kw*48 + c is.  Only the weights stream (6 KB
// per K-tile); all MFMA tiles are real (2 x 3 tiles of 16 x 16 per wave).  A ds_read_b128 lane group {0-3,12-15,20-27} then
// touches start banks 24 r + 4 q (mod 64) = 16 distinct multiples of 4: conflict-free without any swizzle.
//
// The same kernel is the data gradient (flipped-tap weights, input = packed dY one row later) - exactly the two calls the
// model made through aptai_gemm_bf16.
#include "common.h"

namespace {

constexpr int CG = 48, KW = 128, FR = 128;              // channels per group (base), taps, output frames per block
constexpr int KTOT = KW * CG;                           // 6144
constexpr int XROWS = FR + KW - 1;                      // 255 input frames
constexpr int X_BYTES = 6 * 256 * 16;                   // 24 KiB (255 * 96 = 24480 B used)
constexpr int W_STAGE = 512 * 16;                       // 8 KiB (48 rows x 128 B used)

struct PosconvArgs {
    const bf16_t* xg; long x_group_stride, x_batch_stride;   // elements: [G][B][rows_p][Cg]
    const bf16_t* w;                                           // [G][Cg][128*Cg]
    const float* bias;                                         // [H] or null
    const bf16_t* residual;                                    // [B*Tp][H] or null
    bf16_t* out;                                               // [B*Tp][H]
    bf16_t* out_pre;                                           // pre-activation copy or null
    int Tp, H, gelu;
};

// epilogue of one 16 x 16 MFMA tile: the lane owns 4 consecutive channels of one frame
__device__ __forceinline__ void posconv_store(const PosconvArgs& a, const f32x4 accv, long row, int col) {
    float v[4] = {accv[0], accv[1], accv[2], accv[3]};
    if (a.bias) {
        const f32x4 bv = *(const f32x4*)(a.bias + col);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += bv[r];
    }
    if (a.out_pre) *(u32x2*)(a.out_pre + row * a.H + col) = (u32x2){pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
    if (a.gelu) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_fast(v[r]);
    }
    if (a.residual) {
        const u32x2 rv = *(const u32x2*)(a.residual + row * a.H + col);
        v[0] += lo_bf(rv[0]); v[1] += hi_bf(rv[0]); v[2] += lo_bf(rv[1]); v[3] += hi_bf(rv[1]);
    }
    *(u32x2*)(a.out + row * a.H + col) = (u32x2){pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
}

__global__ __launch_bounds__(256, 3) void posconv_kernel(PosconvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sX = smem;
    char* sW = smem + X_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t0 = blockIdx.x * FR, b = blockIdx.y, grp = blockIdx.z;

    // ---- input window: 1530 x 16 B, contiguous in the packed copy (chunks beyond it re-read the last one into LDS padding)
    const bf16_t* xsrc = a.xg + (long)grp * a.x_group_stride + (long)b * a.x_batch_stride + (long)t0 * CG;
#pragma unroll
    for (int it = 0; it < 6; ++it) {
        int c = it * 256 + tid;
        c = c < (XROWS * CG * 2) / 16 ? c : (XROWS * CG * 2) / 16 - 1;
        __builtin_amdgcn_global_load_lds(GLB_PTR(xsrc + (long)c * 8), LDS_PTR(sX + (it * 256 + wave * 64) * 16), 16, 0, 0);
    }
    // ---- weight K-tiles: [48][64] bf16, 128-B rows, chunk index XORed with (row & 7); 2 chunks per thread (rows >= 48 clamp)
    const bf16_t* wg = a.w + (long)grp * CG * KTOT;
    const bf16_t* pw[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int c = it * 256 + tid, row = c >> 3, pc = c & 7;
        const int r = row < CG ? row : CG - 1;
        pw[it] = wg + (long)r * KTOT + ((pc ^ (row & 7)) << 3);
    }
    auto stage_w = [&](char* buf) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            __builtin_amdgcn_global_load_lds(GLB_PTR(pw[it]), LDS_PTR(buf + (it * 256 + wave * 64) * 16), 16, 0, 0);
            pw[it] += 64;
        }
    };
    stage_w(sW);

    f32x4 acc[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // per-lane fragment offsets (loop invariant): A row = wave*32 + i*16 + (lane&15), 16-B chunk (lane>>4) of the 32-k slice
    const int a_off = (wave * 32 + (lane & 15)) * (CG * 2) + (lane >> 4) * 16;
    int b_off[3][2];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int row = j * 16 + (lane & 15), q = ks * 4 + (lane >> 4);
            b_off[j][ks] = row * 128 + ((q ^ (row & 7)) << 4);
        }

    constexpr int NKT = KTOT / 64;                      // 96
    for (int kt = 0; kt < NKT; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int cur = kt & 1;
        if (kt + 1 < NKT) stage_w(sW + (cur ^ 1) * W_STAGE);
        const char* sw = sW + cur * W_STAGE;
        const char* sa = sX + a_off + kt * 128;         // k advances by 64 elements = 128 B per K-tile
        bf16x8 af[2][2], bfr[2][3];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < 2; ++i) af[ks][i] = *(const bf16x8*)(sa + i * (16 * CG * 2) + ks * 64);
#pragma unroll
            for (int j = 0; j < 3; ++j) bfr[ks][j] = *(const bf16x8*)(sw + b_off[j][ks]);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ks][j], af[ks][i], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const long row = (long)b * a.Tp + t0 + wave * 32 + i * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < 3; ++j) posconv_store(a, acc[i][j], row, grp * CG + j * 16 + (lane >> 4) * 4);
    }
}

// ---- 64 channels per group (wav2vec2-large).  A 128-byte Toeplitz stride would put the 16 rows of a fragment read on the same
// banks, so the window is stored with a 144-byte frame pitch (16 distinct 4-bank groups again) and every K-tile is exactly one
// tap: A[t][kw, c] = window[(t + kw) * 144 + 2 c].  The padded pitch rules out the linear LDS-DMA copy: the window is loaded
// through registers once.
constexpr int CG2 = 64, PITCH2 = 144;
constexpr int X2_BYTES = (XROWS + 1) * PITCH2;          // 36,864 B
constexpr int W2_STAGE = CG2 * 128;                     // 8 KiB: [64][64] bf16

__global__ __launch_bounds__(256, 3) void posconv64_kernel(PosconvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sX = smem;
    char* sW = smem + X2_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t0 = blockIdx.x * FR, b = blockIdx.y, grp = blockIdx.z;
    const bf16_t* xsrc = a.xg + (long)grp * a.x_group_stride + (long)b * a.x_batch_stride + (long)t0 * CG2;
    for (int c = tid; c < XROWS * 8; c += 256) {        // 255 frames x 8 chunks of 16 B
        const int fr = c >> 3, ch = c & 7;
        *(u32x4*)(sX + fr * PITCH2 + ch * 16) = *(const u32x4*)(xsrc + (long)fr * CG2 + ch * 8);
    }
    const bf16_t* wg = a.w + (long)grp * CG2 * (KW * CG2);
    const bf16_t* pw[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int c = it * 256 + tid, row = c >> 3, pc = c & 7;
        pw[it] = wg + (long)row * (KW * CG2) + ((pc ^ (row & 7)) << 3);
    }
    auto stage_w = [&](char* buf) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            __builtin_amdgcn_global_load_lds(GLB_PTR(pw[it]), LDS_PTR(buf + (it * 256 + wave * 64) * 16), 16, 0, 0);
            pw[it] += 64;
        }
    };
    stage_w(sW);
    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int a_off = (wave * 32 + (lane & 15)) * PITCH2 + (lane >> 4) * 16;
    int b_off[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int row = j * 16 + (lane & 15), q = ks * 4 + (lane >> 4);
            b_off[j][ks] = row * 128 + ((q ^ (row & 7)) << 4);
        }
    for (int kt = 0; kt < KW; ++kt) {                   // one tap per K-tile
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int cur = kt & 1;
        if (kt + 1 < KW) stage_w(sW + (cur ^ 1) * W2_STAGE);
        const char* sw = sW + cur * W2_STAGE;
        const char* sa = sX + a_off + kt * PITCH2;
        bf16x8 af[2][2], bfr[2][4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < 2; ++i) af[ks][i] = *(const bf16x8*)(sa + i * (16 * PITCH2) + ks * 64);
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[ks][j] = *(const bf16x8*)(sw + b_off[j][ks]);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ks][j], af[ks][i], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const long row = (long)b * a.Tp + t0 + wave * 32 + i * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < 4; ++j) posconv_store(a, acc[i][j], row, grp * CG2 + j * 16 + (lane >> 4) * 4);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient: dW[grp][co][kw*48 + ci] = sum over the frames f of the packed copies of dU[f][co] * X[f + kw][ci]
// (the zero gap rows between utterances make one long frame axis per group, exactly as the TN GEMM call did).
// Both operands are frame-major ("K-major"): A[f][co] = dU rows (96 B), B[f][n] = X at byte 96 f + 2 n - the Toeplitz
// window again, 64 frames x 128 columns = 6.3 KB instead of a 16 KB tile.  One block = 48 x 128 outputs of one group,
// 4 waves x 32 columns, fragments through ds_read_b64_tr_b16 (per-lane addresses, so the 96-byte row stride needs no
// re-layout); every MFMA tile is real where the 128 x 128 TN tile computed 128 rows for 48.
constexpr int WG_NT = 128;                                // output columns per block
constexpr int WG_STAGE = 2 * 512 * 16;                    // A (6144 B used) + B window (6304 B used), 8 KiB each

struct PosconvWgradArgs {
    const bf16_t* du; const bf16_t* x;                    // group-major packed copies, [G][frames_total][48]
    long group_stride;                                    // elements
    float* dw;                                            // [G][48][6144]
    int nkt;                                              // K-tiles of 64 frames
};

// The transposing reads go through inline asm with hand-placed lgkmcnt waits: behind an LDS-DMA the compiler puts
// `s_waitcnt vmcnt(0)` in front of every ds_read_tr builtin, i.e. it waits for the NEXT K-tile's staging loads before it computes
// the current one (see read_frag_tr_asm in gemm.hip).
__device__ __forceinline__ void tr_frag_asm(const char* base, int lane, int ks, short4v& lo, short4v& hi) {
    // 16 rows (the 32 contiguous bytes at `base`, 8-B piece p) x 32 k of a frame-major tile with 96-byte rows
    const int gq = lane >> 4, i = lane & 15, qq = i >> 2, p = i & 3;
    const uint32_t a0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)base +
                        (uint32_t)((ks * 32 + gq * 8 + qq) * (CG * 2) + p * 8);
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a0));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a0), "n"(4 * CG * 2));
}
__device__ __forceinline__ bf16x8 tr_join(const short4v lo, const short4v hi) {
    return __builtin_bit_cast(bf16x8, (short8v){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
}

__global__ __launch_bounds__(256, 3) void posconv_wgrad_kernel(PosconvWgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.x * WG_NT, grp = blockIdx.y;
    const bf16_t* du = a.du + (long)grp * a.group_stride;
    const bf16_t* xb = a.x + (long)grp * a.group_stride + n0;
    // per-thread staging sources, advanced by 64 frames per K-tile; chunks beyond the used bytes re-read the last one
    const bf16_t* pa[2];
    const bf16_t* pb[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int c = it * 256 + tid;
        pa[it] = du + (long)(c < 384 ? c : 383) * 8;
        pb[it] = xb + (long)(c < 394 ? c : 393) * 8;
    }
    auto stage = [&](char* buf) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            __builtin_amdgcn_global_load_lds(GLB_PTR(pa[it]), LDS_PTR(buf + (it * 256 + wave * 64) * 16), 16, 0, 0);
            pa[it] += 64 * CG;
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            __builtin_amdgcn_global_load_lds(GLB_PTR(pb[it]), LDS_PTR(buf + 8192 + (it * 256 + wave * 64) * 16), 16, 0, 0);
            pb[it] += 64 * CG;
        }
    };
    stage(smem);
    f32x4 acc[3][2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < a.nkt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int cur = kt & 1;
        if (kt + 1 < a.nkt) stage(smem + (cur ^ 1) * WG_STAGE);
        const char* sa = smem + cur * WG_STAGE;
        const char* sb = sa + 8192 + wave * 64;            // this wave's 32 columns (2 bytes each)
        short4v fl[2][5], fh[2][5];                        // fragments 0..2: dU rows, 3..4: this wave's X columns
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < 3; ++i) tr_frag_asm(sa + i * 32, lane, ks, fl[ks][i], fh[ks][i]);
#pragma unroll
            for (int j = 0; j < 2; ++j) tr_frag_asm(sb + j * 32, lane, ks, fl[ks][3 + j], fh[ks][3 + j]);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            // 20 reads issued, in-order return: at most 10 outstanding = the first k-half is back
            if (ks == 0)
                asm volatile("s_waitcnt lgkmcnt(10)" : "+v"(fl[0][0]), "+v"(fl[0][1]), "+v"(fl[0][2]), "+v"(fl[0][3]), "+v"(fl[0][4]),
                                                       "+v"(fh[0][0]), "+v"(fh[0][1]), "+v"(fh[0][2]), "+v"(fh[0][3]), "+v"(fh[0][4]));
            else
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fl[1][0]), "+v"(fl[1][1]), "+v"(fl[1][2]), "+v"(fl[1][3]), "+v"(fl[1][4]),
                                                      "+v"(fh[1][0]), "+v"(fh[1][1]), "+v"(fh[1][2]), "+v"(fh[1][3]), "+v"(fh[1][4]));
            bf16x8 af[3], bfr[2];
#pragma unroll
            for (int i = 0; i < 3; ++i) af[i] = tr_join(fl[ks][i], fh[ks][i]);
#pragma unroll
            for (int j = 0; j < 2; ++j) bfr[j] = tr_join(fl[ks][3 + j], fh[ks][3 + j]);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
    }
    float* out = a.dw + (long)grp * CG * KTOT;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int m = i * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wave * 32 + j * 16 + (lane >> 4) * 4;
            *(f32x4*)(out + (long)m * KTOT + n) = acc[i][j];
        }
    }
}

}  // namespace

extern "C" int aptai_posconv_wgrad(const void* du_g, const void* x_g, float* dw, int64_t B, int64_t Tp, int64_t H, int64_t groups,
                                   int64_t Kw, int64_t pad, void* stream) {
    APTAI_REQUIRE(du_g && x_g && dw, "aptai_posconv_wgrad: null pointer");
    APTAI_REQUIRE(groups > 0 && H == groups * CG && Kw == KW && 2 * pad == KW, "aptai_posconv_wgrad: built for 48 channels per group, 128 taps, pad 64");
    const long rows_p = Tp + 2 * pad;
    const long frames = B * rows_p - 2 * pad;              // dU frame f pairs with X frames f .. f+127
    APTAI_REQUIRE(B > 0 && Tp > 0 && frames % 64 == 0, "aptai_posconv_wgrad: %ld frames per group is not a multiple of 64", (long)frames);
    PosconvWgradArgs a;
    a.du = (const bf16_t*)du_g + pad * CG;                 // dU of output frame t lives at packed row pad + t
    a.x = (const bf16_t*)x_g;
    a.group_stride = B * rows_p * CG;
    a.dw = dw;
    a.nkt = (int)(frames / 64);
    APTAI_LAUNCH(posconv_wgrad_kernel, dim3((unsigned)(KTOT / WG_NT), (unsigned)groups), dim3(256), 2 * WG_STAGE, (hipStream_t)stream, a);
    APTAI_CHECK_LAUNCH("posconv_wgrad_kernel");
    return APTAI_OK;
}

extern "C" int aptai_posconv_gemm(const void* xg, int64_t first_row, const void* w, const float* bias, const void* residual, void* out,
                                  void* out_pre, int64_t B, int64_t Tp, int64_t H, int64_t groups, int64_t Kw, int64_t pad, int gelu,
                                  void* stream) {
    APTAI_REQUIRE(xg && w && out, "aptai_posconv_gemm: null pointer");
    APTAI_REQUIRE(groups > 0 && (H == groups * CG || H == groups * CG2) && Kw == KW,
                  "aptai_posconv_gemm: built for 48 or 64 channels per group and 128 taps (H=%ld groups=%ld Kw=%ld)", (long)H, (long)groups, (long)Kw);
    APTAI_REQUIRE(B > 0 && Tp > 0 && Tp % FR == 0, "aptai_posconv_gemm: Tp=%ld must be a positive multiple of %d", (long)Tp, FR);
    APTAI_REQUIRE(first_row >= 0 && first_row + KW - 1 <= 2 * pad, "aptai_posconv_gemm: first_row=%ld leaves the padded window (pad=%ld)",
                  (long)first_row, (long)pad);
    const long cg = H / groups;
    const long rows_p = Tp + 2 * pad;
    PosconvArgs a;
    a.xg = (const bf16_t*)xg + first_row * cg;
    a.x_batch_stride = rows_p * cg;
    a.x_group_stride = B * rows_p * cg;
    a.w = (const bf16_t*)w; a.bias = bias; a.residual = (const bf16_t*)residual; a.out = (bf16_t*)out; a.out_pre = (bf16_t*)out_pre;
    a.Tp = (int)Tp; a.H = (int)H; a.gelu = gelu;
    const dim3 grid((unsigned)(Tp / FR), (unsigned)B, (unsigned)groups);
    if (cg == CG) APTAI_LAUNCH(posconv_kernel, grid, dim3(256), X_BYTES + 2 * W_STAGE, (hipStream_t)stream, a);
    else APTAI_LAUNCH(posconv64_kernel, grid, dim3(256), X2_BYTES + 2 * W2_STAGE, (hipStream_t)stream, a);
    APTAI_CHECK_LAUNCH("posconv_kernel");
    return APTAI_OK;
}
