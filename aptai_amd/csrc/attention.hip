// Multi-head self-attention core for wav2vec2 (head_dim 64), forward + backward, gfx950.
// Replaces HF:452-461 / the default sdpa path (softmax(Q K^T d^-1/2 + key-padding mask) V, dropout on the
// probabilities in training) and its autograd.  Flash-style: scores never reach HBM.
//
// Layout: qkv [B*Tp][3H] bf16 (Q | K | V, head h at columns h*64..h*64+63 of each third), Tp = frames per
// utterance padded to a multiple of 128, lens[b] = valid frames (keys >= lens[b] are masked; query rows
// beyond lens[b] are still computed, like the reference).  ctx [B*Tp][H] bf16.  lse2 [B][heads][Tp] fp32 holds
// log2-sum-exp2 of the scaled scores (x = s * scale * log2 e), shared with the backward kernels.
//
// MFMA plan (v_mfma_f32_32x32x16_bf16, one wave = 32 query rows or 32 key rows):
//   fwd : S^T = K Q^T (key rows from LDS, Q fragment in registers)  -> each lane owns ONE query column, so the
//         online-softmax row statistics are lane-local (+ one lane^32 exchange); P stays in registers and is
//         fed straight back as the B operand of O^T = V^T P^T (accumulator-as-operand, permuted k order);
//         V^T fragments come from the row-major V tile through ds_read_b64_tr_b16.
//   bwd : dK/dV kernel keeps keys on the lanes (S = Q K^T, dP = dO V^T; dV^T += dO^T P, dK^T += Q^T dS);
//         dQ kernel keeps queries on the lanes (S^T, dP^T; dQ^T += K^T dS^T).  P is recomputed from lse2.
// K/V (or Q/dO) tiles are staged global -> registers -> LDS (16-B chunks, XOR-swizzled by row&7).
#include "common.h"

namespace {

#ifdef APTAI_STAMPS
// development only (tools/ab builds): per-block wall-clock stamps (100 MHz), see tools/attn_stamps.py
__device__ unsigned long long g_attn_stamps[3 * 1024 * 8];
#define ATTN_STAMP(k, slot)                                                                                       \
    do {                                                                                                          \
        if (threadIdx.x == 0) {                                                                                   \
            const int bl = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);                        \
            if (bl < 1024) g_attn_stamps[((k) * 1024 + bl) * 8 + (slot)] = wall_clock64();                         \
        }                                                                                                         \
    } while (0)
#else
#define ATTN_STAMP(k, slot) do {} while (0)
#endif

constexpr int HD = 64;          // head dim
constexpr float LOG2E = 1.4426950408889634f;

struct AttnArgs {
    const bf16_t* qkv; long ld;     // ld = 3H
    const int* lens;
    bf16_t* ctx; long ldo;          // H
    float* lse2;
    float* o32;                     // optional fp32 copy of the context (keeps delta = rowsum(dO*O) accurate)
    int B, Tp, H, heads;
    float c;                        // softmax_scale * log2(e)
    float scale;
    uint32_t seed0, seed1, thr16; float dscale;
    const uint32_t* salt;
    // backward
    const bf16_t* dctx;
    float* delta;
    bf16_t* dqkv;
    int skip_pad_q;
    int stagger;                    // experiment: s_sleep units (64 clocks) per wave-slot index at kernel start
    int xcd_remap;
    int q_prescaled;                // the Q third of qkv already carries scale * log2(e) (QKV GEMM epilogue): scores arrive in the exp2 domain
};

// ---- LDS tile [rows][64] bf16, 128-B rows, 16-B chunk index XORed with tile_swz(row) = row bits (2, 3, 1) -> swizzle bits
// (0, 1, 2).  Found by exhaustive search over the GF(2)-linear 3x5 maps: it is the simplest one for which BOTH access shapes
// of these kernels are conflict-free - the 32-row ds_read_b128 fragments (lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31}:
// 16 distinct 16-byte bank groups) and the ds_read_b64_tr_b16 column reads (4 rows x 64 B per 32-lane group: rows of equal
// parity must differ in chunk bit 2).  With the usual (row & 7) rocprofv3 counted 1.3-1.8 conflict cycles per LDS cycle here.
// All per-lane LDS offsets are loop invariant and are computed ONCE per kernel (the per-tile instruction count, not the
// MFMA rate, is what bounds these kernels at T ~ 500: every VALU instruction in the tile loop costs ~1/500 of it).
__device__ __forceinline__ int tile_swz(int row) { return ((row >> 2) & 3) | (((row >> 1) & 1) << 2); }
__device__ __forceinline__ int tile_off(int row, int chunk) { return row * 128 + ((chunk ^ tile_swz(row)) << 4); }

struct LaneOffs {
    int rowk[4];      // row reads: byte offset of this lane's 16-B chunk for d-step ds (tile row = lane&31); + row_block*4096
    int tr[2][2];     // transposed reads: [d-tile][lo / hi half], depth block 0 (rows 4h+qq and 4h+qq+8); + s*2048
};
__device__ __forceinline__ LaneOffs lane_offs(int lane) {
    LaneOffs o;
    const int r31 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) o.rowk[ds] = tile_off(r31, 2 * ds + h);
    const int dgrp = (lane >> 4) & 1, i = lane & 15, qq = i >> 2, p = i & 3;
    const int trow = 4 * h + qq;                          // rows 16*s + trow (+8): the swizzle reads row bits 1..3 only
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int hi = 0; hi < 2; ++hi) o.tr[dt][hi] = tile_off(trow + 8 * hi, dt * 4 + 2 * dgrp + (p >> 1)) + ((p & 1) << 3);
    return o;
}
__device__ __forceinline__ bf16x8 rd_row(const char* tile, const LaneOffs& o, int row_block, int ds) {
    return *(const bf16x8*)(tile + o.rowk[ds] + row_block * 4096);
}
// A operand (rows = tile columns d, depth = tile rows) for 32x32x16: element j <-> tile row 16*s + 8*(j>>2) + 4*h + (j&3)
__device__ __forceinline__ bf16x8 rd_tr(const char* tile, const LaneOffs& o, int s, int dt) {
    short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(tile + o.tr[dt][0] + s * 2048));
    short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(tile + o.tr[dt][1] + s * 2048));
    short8v r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, r);
}

__device__ __forceinline__ bf16x8 pack8(const float* v) {
    u32x4 u = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
    return __builtin_bit_cast(bf16x8, u);
}

// accumulator register r of a 32x32 tile <-> row (r&3) + 8*(r>>2) + 4*h
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ---- coalesced block prologue / epilogue helpers (the per-BLOCK cost: a lane owns one ROW in these kernels, so direct
// fragment loads / accumulator stores touch 32-64 cache lines per instruction)
constexpr int OUT_PITCH = 144;                                  // bytes per staged row (128 + 16: odd number of 16-B units)
constexpr int STAGE_BYTES = 4 * 32 * OUT_PITCH;                 // 18 432 B: 32 rows per wave
constexpr int DQ_STAGE = 2 * 128 * 128;                           // dQ kernel: Q and dO tiles side by side (>= STAGE_BYTES)
constexpr int DKDV_BLOCKS = 3;                                  // dK/dV kernel, pre-scaled Q (the model's path): blocks per CU (register budget 168)
constexpr int DKDV_KV_BYTES = 2 * 128 * 128;                    // its resident K and V tiles

// [128 rows][64] bf16 (row stride ld elements) -> LDS tile layout (tile_off), 8 lanes per 128-byte row
__device__ __forceinline__ void stage_rows128(char* dst, const bf16_t* src, long ld, int tid) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int cid = it * 256 + tid, row = cid >> 3, ch = cid & 7;
        *(u32x4*)(dst + tile_off(row, ch)) = *(const u32x4*)(src + (long)row * ld + ch * 8);
    }
}
// the wave's transposed accumulator pair acc[dt][4 g4 + j] = X^T[d = 32 dt + 8 g4 + 4 h + j][row = lane & 31], scaled, as bf16
// rows of 128 bytes: through the wave's own staging area sw (32 x OUT_PITCH), then 16-byte chunks of whole rows
__device__ __forceinline__ void store_rows_bf16(char* sw, const f32x16 (&acc)[2], float scale, bf16_t* g_row0, long ld, int lane) {
    const int r31 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int d = dt * 32 + 8 * g4 + 4 * h;
            *(u32x2*)(sw + r31 * OUT_PITCH + d * 2) = (u32x2){pack2bf(acc[dt][4 * g4] * scale, acc[dt][4 * g4 + 1] * scale),
                                                              pack2bf(acc[dt][4 * g4 + 2] * scale, acc[dt][4 * g4 + 3] * scale)};
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // same wave, LDS in order: only the compiler must not reorder
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = i * 64 + lane, r = idx >> 3, ch = idx & 7;
        *(u32x4*)(g_row0 + (long)r * ld + ch * 8) = *(const u32x4*)(sw + r * OUT_PITCH + ch * 16);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// dropout on P: one hash per PAIR of consecutive keys of one query row; pair index = (bh*Tp + q)*(Tp/2) + key/2 (32-bit)
__device__ __forceinline__ bool attn_keep(uint32_t pair, int key, uint32_t s0, uint32_t s1, uint32_t thr16) {
    const uint32_t hsh = rng_hash(pair, s0, s1);
    return ((key & 1) ? (hsh >> 16) : (hsh & 0xffffu)) >= thr16;
}

// ================================================================================== forward
// grid (Tp/128, heads, B), 256 threads; wave w: queries q0 = qt*128 + w*32 .. +31
// Block -> (tile, head, utterance).  Workgroups are dealt round-robin over the 8 XCDs in launch order, so the Tp/128 tiles of one
// (utterance, head) - which all read the same K / V (forward, dQ) or Q / dO (dK/dV) - would land on as many different L2s.
// The remap gives every XCD a contiguous run of (utterance, head) groups with all their tiles (speed only: any dispatch order is
// correct).  APTAI_ATTN_XCD_REMAP=0 keeps the launch order (A/B).
__device__ __forceinline__ void attn_block(const AttnArgs& a, int& tile, int& hd, int& b) {
    const int nt = gridDim.x, total = nt * gridDim.y * gridDim.z;
    int L = blockIdx.x + nt * (blockIdx.y + gridDim.y * blockIdx.z);
    if (a.xcd_remap && (total & 7) == 0) L = (L & 7) * (total >> 3) + (L >> 3);
    tile = L % nt;
    const int grp = L / nt;
    hd = grp % (int)gridDim.y;
    b = grp / (int)gridDim.y;
}

// The per-BLOCK cost matters as much as the tile loop at T ~ 500 (8 key tiles): the epilogue goes through LDS so that every
// store instruction writes whole 128-byte row segments (a lane owns one QUERY, so storing from the accumulators directly is
// 8 or 16 bytes per lane at a row stride: 32-64 distinct lines per instruction, 32 instructions per wave).
//
// Tile loop.  rocprofv3 counters (profiles/r02_attn_pmc.json) show the SIMD time of a tile is the SUM of its matrix time
// (32 cycles per MFMA) and its vector-issue time: co-resident waves run the same phases in lockstep, so nothing overlaps
// unless the interleave is in each wave's own instruction stream (an MFMA holds the vector issue port for 8 of its 32
// cycles; up to 24 cycles of VALU issue behind it are free).  Hence:
//   * reference-point softmax: p = exp2(c s - ref) with a per-query reference that is the true maximum of the FIRST tile
//     and afterwards only moves (by an exact power of two) when a tile's row sum exceeds 2^14.  No per-tile maximum, no
//     O / l rescale, and - the point - no dependency of the exponentials on the whole tile: the softmax of keys 0..31 is
//     issued between the MFMAs of keys 32..63, the softmax of keys 32..63 between the first O^T MFMAs;
//   * all fragment reads of a phase are issued before its first MFMA, the two accumulator chains alternate;
//   * 2-deep LDS ring, ONE barrier per tile.
constexpr int FWD_BUF = 2 * 64 * 128;                           // one K tile + one V tile (64 keys each)
constexpr int FWD_SMEM = 2 * FWD_BUF;                           // 2-deep ring, 32 KB (>= the 18 KB of epilogue staging)
// The reference sits REF_HEADROOM binades ABOVE the maximum it was taken from (floating point is scale-free: p ~ 2^-40 costs no
// accuracy in fp32 sums or in the bf16 P operand, which has the fp32 exponent range), so a score has to jump 2^(127+40) over
// everything seen before it, within one 64-key tile, to overflow; scores 2^86 below the reference flush to zero, as they should.
constexpr float REF_HEADROOM = 40.0f;
constexpr float REF_SUM_LIMIT = 1.4901161193847656e-08f;         // 2^-26 = 2^(14 - REF_HEADROOM)

template <bool V> struct Flag { static constexpr bool value = V; };

template <bool DROP>
__global__ __launch_bounds__(256, 3) void attn_fwd_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[FWD_SMEM];
    ATTN_STAMP(0, 0);
    if (DROP) apply_salt(a.salt, a.seed0, a.seed1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    int tile_x, hd, b;
    attn_block(a, tile_x, hd, b);
    const int q0 = tile_x * 128 + wave * 32;
    int len = a.lens[b];
    len = len < 1 ? 1 : (len > a.Tp ? a.Tp : len);
    const int ntiles = (len + 63) >> 6;
    const long rowbase = (long)b * a.Tp;
    const int q = q0 + (lane & 31);
    const bf16_t* Qg = a.qkv + (rowbase + q) * a.ld + hd * HD;
    const bf16_t* Kg = a.qkv + rowbase * a.ld + a.H + hd * HD;
    const bf16_t* Vg = Kg + a.H;
    const LaneOffs lo = lane_offs(lane);

    bf16x8 qf[4];
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) qf[ds] = *(const bf16x8*)(Qg + ds * 16 + h * 8);
    // keep the Q loads OLDER than every K/V load: the wait for tile 0 then covers them, and the tile loop carries no
    // vmcnt for them (the scheduler had moved two of them behind the tile loads: a vmcnt(1) inside the S chain, per tile)
    __builtin_amdgcn_sched_barrier(0);

    f32x16 oT[2];
    oT[0] = (f32x16)(0.f);
    oT[1] = (f32x16)(0.f);
    float ref = 0.f, l_run = 0.f;              // reference in the exp2 domain (c * raw score), running sum relative to it

    // staging: 512 16-B chunks per tile, 2 per thread per tensor; offsets are loop invariant
    int soff[2];
    long goff[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int cid = it * 256 + tid, row = cid >> 3, ch = cid & 7;
        soff[it] = tile_off(row, ch);
        goff[it] = (long)row * a.ld + ch * 8;
    }
    u32x4 kreg[2], vreg[2];
    const long tile_stride = 64 * a.ld;
#pragma unroll
    for (int it = 0; it < 2; ++it) { kreg[it] = *(const u32x4*)(Kg + goff[it]); vreg[it] = *(const u32x4*)(Vg + goff[it]); }
    // dropout: row hash of this lane's query + (pair index) * K1; registers r, r+1 are the two keys of pair
    // (kbase + 32 kt2 + acc_row(r, h)) / 2 = kbase / 2 + 2 h + 16 kt2 + acc_row(r, 0) / 2
    uint32_t hbase = 0;
    if (DROP) hbase = rng_hash((uint32_t)((b * a.heads + hd) * a.Tp + q), a.seed0, a.seed1) + (uint32_t)(2 * h) * ATTN_K1;
    const float c = a.c;
#pragma unroll
    for (int it = 0; it < 2; ++it) { *(u32x4*)(smem + soff[it]) = kreg[it]; *(u32x4*)(smem + 64 * 128 + soff[it]) = vreg[it]; }
    if (ntiles > 1) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            kreg[it] = *(const u32x4*)(Kg + tile_stride + goff[it]);
            vreg[it] = *(const u32x4*)(Vg + tile_stride + goff[it]);
        }
    }
    __syncthreads();
    ATTN_STAMP(0, 1);

    // one key tile; FIRST: take the reference from this tile's maximum; MASK: keys >= len are excluded (last tile only)
    auto tile = [&](auto first_flag, auto mask_flag, const int t) {
        constexpr bool FIRST = decltype(first_flag)::value, MASK = decltype(mask_flag)::value;
        const char* sK = smem + (t & 1) * FWD_BUF;
        const char* sV = sK + 64 * 128;
        const int kbase = t * 64;
        bf16x8 kfr[2][4];
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) kfr[0][ds] = rd_row(sK, lo, 0, ds);
        if (t + 1 < ntiles) {
            char* nb = smem + ((t + 1) & 1) * FWD_BUF;
#pragma unroll
            for (int it = 0; it < 2; ++it) { *(u32x4*)(nb + soff[it]) = kreg[it]; *(u32x4*)(nb + 64 * 128 + soff[it]) = vreg[it]; }
            if (t + 2 < ntiles) {
                const bf16_t* kn = Kg + (long)(t + 2) * tile_stride;
                const bf16_t* vn = Vg + (long)(t + 2) * tile_stride;
#pragma unroll
                for (int it = 0; it < 2; ++it) { kreg[it] = *(const u32x4*)(kn + goff[it]); vreg[it] = *(const u32x4*)(vn + goff[it]); }
            }
        }
        // S^T tiles: keys (kt2*32 + acc_row) x queries (lane&31)
        f32x16 sT[2];
        sT[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[0][0], qf[0], (f32x16)(0.f), 0, 0, 0);
#pragma unroll
        for (int ds = 1; ds < 4; ++ds) sT[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[0][ds], qf[ds], sT[0], 0, 0, 0);
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) kfr[1][ds] = rd_row(sK, lo, 1, ds);
        sT[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[1][0], qf[0], (f32x16)(0.f), 0, 0, 0);
#pragma unroll
        for (int ds = 1; ds < 4; ++ds) sT[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[1][ds], qf[ds], sT[1], 0, 0, 0);
        if (MASK) {
#pragma unroll
            for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kbase + kt2 * 32 + acc_row(r, h) >= len) sT[kt2][r] = -INFINITY;
        }
        if (FIRST) {                                            // the tile holds at least one valid key: the maximum is finite
            float mt = sT[0][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mt = fmaxf(mt, sT[0][r]);
#pragma unroll
            for (int r = 0; r < 16; ++r) mt = fmaxf(mt, sT[1][r]);
            mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
            ref = fmaf(mt, c, REF_HEADROOM);
        }
        float psum = 0.f;
#pragma unroll
        for (int kt2 = 0; kt2 < 2; ++kt2) {
            // V^T fragments of this half's 32 keys: their LDS latency hides under the half's softmax arithmetic
            bf16x8 vfr[2][2];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) vfr[s][dt] = rd_tr(sV, lo, 2 * kt2 + s, dt);
            float x[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) { x[r] = fast_exp2(fmaf(sT[kt2][r], c, -ref)); psum += x[r]; }
            if (DROP) {
#pragma unroll
                for (int r = 0; r < 16; r += 2) {              // registers r, r+1 are consecutive keys: one hash per pair
                    const uint32_t hsh = attn_mix(hbase + (uint32_t)(kbase / 2 + 16 * kt2 + acc_row(r, 0) / 2) * ATTN_K1);
                    x[r] = (hsh & 0xffffu) >= a.thr16 ? x[r] : 0.f;                    // the 1/(1-p) factor rides on the
                    x[r + 1] = (hsh >> 16) >= a.thr16 ? x[r + 1] : 0.f;                // final 1/l normalisation
                }
            }
            const bf16x8 pf0 = pack8(&x[0]), pf1 = pack8(&x[8]);
            // O^T += V^T P^T : 2 k-steps of 16 keys
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) oT[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[0][dt], pf0, oT[dt], 0, 0, 0);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) oT[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[1][dt], pf1, oT[dt], 0, 0, 0);
        }
        psum += __shfl_xor(psum, 32, 64);
        l_run += psum;
        // the reference moves only when a row sum says some p passed 2^14 times the level of the previous maximum: exact
        // power-of-two rescale of O and l (this tile's products included) back to the headroom level, applied after the fact -
        // rare, wave-uniform.  (A row that overflowed all the same - a 2^167 jump inside one tile - turns into inf / NaN in lse
        // and ctx rather than into a silently wrong row.)
        if (!FIRST && __builtin_amdgcn_ballot_w64(!(psum <= REF_SUM_LIMIT))) {
            if (!(psum <= REF_SUM_LIMIT)) {
                const float e = floorf(fast_log2(psum)) + REF_HEADROOM;
                const float sc = fast_exp2(-e);
                ref += e;
                l_run *= sc;
#pragma unroll
                for (int r = 0; r < 16; ++r) { oT[0][r] *= sc; oT[1][r] *= sc; }
            }
        }
        __syncthreads();           // every wave has read buffer t&1 (its fragments are consumed); buffer (t+1)&1 is written
    };
    if (ntiles == 1) {
        tile(Flag<true>{}, Flag<true>{}, 0);
    } else {
        tile(Flag<true>{}, Flag<false>{}, 0);
        for (int t = 1; t + 1 < ntiles; ++t) tile(Flag<false>{}, Flag<false>{}, t);
        tile(Flag<false>{}, Flag<true>{}, ntiles - 1);
    }
    ATTN_STAMP(0, 2);
    const float l_tot = l_run;                 // both half-waves' keys are in (psum was exchanged per tile)
    const float inv = __frcp_rn(l_tot) * a.dscale;
    if (a.lse2 && h == 0) a.lse2[((long)b * a.heads + hd) * a.Tp + q] = ref + fast_log2(l_tot);
    // epilogue through LDS: [32 queries][OUT_PITCH] per wave, written 8 / 16 bytes per lane, read back as 16-byte chunks of
    // whole rows so that 8 lanes cover one 128-byte line
    char* sw = smem + wave * (32 * OUT_PITCH);                 // (the loop's last barrier released the K / V tiles)
    const int r31 = lane & 31;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            oT[dt][4 * g4 + 0] *= inv; oT[dt][4 * g4 + 1] *= inv; oT[dt][4 * g4 + 2] *= inv; oT[dt][4 * g4 + 3] *= inv;
            const int d = dt * 32 + 8 * g4 + 4 * h;
            *(u32x2*)(sw + r31 * OUT_PITCH + d * 2) = (u32x2){pack2bf(oT[dt][4 * g4], oT[dt][4 * g4 + 1]),
                                                              pack2bf(oT[dt][4 * g4 + 2], oT[dt][4 * g4 + 3])};
        }
    __syncthreads();
    const long orow = (rowbase + q0) * a.ldo + hd * HD;         // first query of this wave
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = i * 64 + lane, r = idx >> 3, ch = idx & 7;
        *(u32x4*)(a.ctx + orow + (long)r * a.ldo + ch * 8) = *(const u32x4*)(sw + r * OUT_PITCH + ch * 16);
    }
    if (a.o32) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            __syncthreads();
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                *(f32x4*)(sw + r31 * OUT_PITCH + (8 * g4 + 4 * h) * 4) =
                    (f32x4){oT[dt][4 * g4], oT[dt][4 * g4 + 1], oT[dt][4 * g4 + 2], oT[dt][4 * g4 + 3]};
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = i * 64 + lane, r = idx >> 3, ch = idx & 7;
                *(f32x4*)(a.o32 + orow + (long)r * a.ldo + dt * 32 + ch * 4) = *(const f32x4*)(sw + r * OUT_PITCH + ch * 16);
            }
        }
    }
    ATTN_STAMP(0, 3);
#ifdef APTAI_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ATTN_STAMP(0, 4);
#endif
}

// ================================================================================== backward: dK, dV
// grid (Tp/128, heads, B); wave w owns keys key0 = kt*128 + w*32 .. +31; loops over 32-query tiles.
// PRE: Q arrives pre-multiplied by scale * log2(e), so S = Q K^T is already the exp2 argument up to the row constant; the S and dP
// accumulators START at -lse2 and -delta (read from LDS straight into the MFMA C operand: no instruction), which removes the
// FMA in front of every exponential and the subtraction behind every dP.
template <bool PRE>
__global__ __launch_bounds__(256, PRE ? DKDV_BLOCKS : 2) void attn_bwd_dkdv_kernel(AttnArgs a) {
    // the block's K and V tiles (128 keys) stay in LDS and their fragments are re-read per query tile: holding them in registers
    // (32 VGPRs) put the kernel at 206 VGPRs = 2 blocks per CU = 512 slots for the 768 blocks of B = 16, Tp = 512: two rounds, the
    // second half empty.  At <= 168 VGPRs all 768 blocks are resident at once.
    __shared__ __attribute__((aligned(16))) char smem[STAGE_BYTES + 384 + DKDV_KV_BYTES];   // staging; Q tile, dO tile (32 queries each)
    char* sQ = smem;
    char* sD = smem + 32 * 128;
    float* sL = (float*)(smem + STAGE_BYTES);           // [0,32) lse2, [32,64) delta, [64,96) dropout row hash of the tile's queries
    char* sKall = smem + STAGE_BYTES + 384;
    char* sVall = sKall + 128 * 128;
    ATTN_STAMP(1, 0);
    if (a.thr16) apply_salt(a.salt, a.seed0, a.seed1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    int tile_x, hd, b;
    attn_block(a, tile_x, hd, b);
    const int kb0 = tile_x * 128;
    const int key0 = kb0 + wave * 32;
    int len = a.lens[b];
    len = len < 1 ? 1 : (len > a.Tp ? a.Tp : len);
    const long rowbase = (long)b * a.Tp;
    const int key = key0 + (lane & 31);
    bf16_t* dK0 = a.dqkv + (rowbase + kb0) * a.ld + a.H + hd * HD;        // row kb0 of this block's dK / dV columns
    bf16_t* dV0 = dK0 + a.H;
    if (kb0 >= len) {                       // whole block beyond the utterance: gradients are exactly zero
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int cid = it * 256 + tid, row = cid >> 3, ch = cid & 7;
            *(u32x4*)(dK0 + (long)row * a.ld + ch * 8) = z;
            *(u32x4*)(dV0 + (long)row * a.ld + ch * 8) = z;
        }
        return;
    }
    const LaneOffs lo = lane_offs(lane);
    const bf16_t* K0 = a.qkv + (rowbase + kb0) * a.ld + a.H + hd * HD;
    stage_rows128(sKall, K0, a.ld, tid);
    stage_rows128(sVall, K0 + a.H, a.ld, tid);
    // (the tile loop opens with a barrier before it overwrites the staging area)
    f32x16 dKT[2], dVT[2];
    dKT[0] = dKT[1] = dVT[0] = dVT[1] = (f32x16)(0.f);

    // dO rows of padded frames are exactly zero in the model (no loss term touches them): skip them on request
    const int nq = a.skip_pad_q ? ((len + 31) >> 5) : (a.Tp >> 5);
    const bf16_t* Qb = a.qkv + rowbase * a.ld + hd * HD;
    const bf16_t* Db = a.dctx + rowbase * a.ldo + hd * HD;
    const float* lse = a.lse2 + ((long)b * a.heads + hd) * a.Tp;
    const float* del = a.delta + ((long)b * a.heads + hd) * a.Tp;
    const int srow = tid >> 3, sch = tid & 7;          // 256 chunks per tile: one per thread
    const int soff = tile_off(srow, sch);
    const long gq = (long)srow * a.ld + sch * 8, gd = (long)srow * a.ldo + sch * 8;
    u32x4 qreg = *(const u32x4*)(Qb + gq), dreg = *(const u32x4*)(Db + gd);
    // row statistics ride along with the tile: threads 0..7 carry lse2, 8..15 delta (one float4 each)
    const float* stat = (tid < 8 ? lse : del) + (tid & 7) * 4;
    f32x4 sreg = (f32x4)(0.f);
    if (tid < 16) sreg = *(const f32x4*)stat;
    const bool key_ok = key < len;
    const bool wave_boundary = key0 + 32 > len;
    const uint32_t hshift = (key & 1) ? 16u : 0u;
    const float c = a.c;
    // dropout: the keys sit on the lanes here, so the row hashes of the tile's 32 queries (level 1 of the two-level hash)
    // ride along with lse2 / delta: threads 32..63 compute one each for the NEXT tile while this one is consumed
    const uint32_t pairk = (uint32_t)(key >> 1) * ATTN_K1;
    const uint32_t rowid0 = (uint32_t)((b * a.heads + hd) * a.Tp) + (uint32_t)(tid & 31);
    uint32_t hreg = 0;
    if (a.thr16 && tid >= 32 && tid < 64) hreg = rng_hash(rowid0, a.seed0, a.seed1);
    ATTN_STAMP(1, 1);
    for (int t = 0; t < nq; ++t) {
        __syncthreads();
        *(u32x4*)(sQ + soff) = qreg;
        *(u32x4*)(sD + soff) = dreg;
        if (tid < 16) *(f32x4*)(sL + tid * 4) = PRE ? -sreg : sreg;
        if (a.thr16 && tid >= 32 && tid < 64) ((uint32_t*)sL)[64 + (tid & 31)] = hreg;
        __syncthreads();
        if (t + 1 < nq) {
            qreg = *(const u32x4*)(Qb + (long)(t + 1) * 32 * a.ld + gq);
            dreg = *(const u32x4*)(Db + (long)(t + 1) * 32 * a.ldo + gd);
            if (tid < 16) sreg = *(const f32x4*)(stat + (t + 1) * 32);
            if (a.thr16 && tid >= 32 && tid < 64) hreg = rng_hash(rowid0 + (uint32_t)((t + 1) * 32), a.seed0, a.seed1);
        }
        // S[q][key] and dP[q][key]
        f32x16 cs = (f32x16)(0.f), cd = (f32x16)(0.f);        // PRE: -lse2 / -delta of the 16 query rows this lane holds
        if (PRE) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 l4 = *(const f32x4*)(sL + 8 * g4 + 4 * h);
                const f32x4 d4 = *(const f32x4*)(sL + 32 + 8 * g4 + 4 * h);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) { cs[4 * g4 + rr] = l4[rr]; cd[4 * g4 + rr] = d4[rr]; }
            }
        }
        f32x16 s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd_row(sQ, lo, 0, 0), rd_row(sKall, lo, wave, 0), cs, 0, 0, 0);
        f32x16 dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd_row(sD, lo, 0, 0), rd_row(sVall, lo, wave, 0), cd, 0, 0, 0);
#pragma unroll
        for (int ds = 1; ds < 4; ++ds) {
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd_row(sQ, lo, 0, ds), rd_row(sKall, lo, wave, ds), s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd_row(sD, lo, 0, ds), rd_row(sVall, lo, wave, ds), dp, 0, 0, 0);
        }
        float pd[16], dsv[16];                                 // P and dS / dscale (see below); dsv starts as delta (!PRE)
        if (PRE) {
#pragma unroll
            for (int r = 0; r < 16; ++r) pd[r] = fast_exp2(s[r]);
        } else {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 l4 = *(const f32x4*)(sL + 8 * g4 + 4 * h);
                const f32x4 d4 = *(const f32x4*)(sL + 32 + 8 * g4 + 4 * h);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    pd[4 * g4 + rr] = fast_exp2(fmaf(s[4 * g4 + rr], c, -l4[rr]));
                    dsv[4 * g4 + rr] = d4[rr];
                }
            }
        }
        if (wave_boundary) {                                   // wave-uniform: only the wave holding key len-1 masks
#pragma unroll
            for (int r = 0; r < 16; ++r) pd[r] = key_ok ? pd[r] : 0.f;
        }
        if (a.thr16) {
            // element (query = register, key = lane): level-2 mix of the query's row hash with this lane's pair index; the
            // lane's key parity picks the half (bfe)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const u32x4 rh = *(const u32x4*)((const uint32_t*)sL + 64 + 8 * g4 + 4 * h);
                // -delta of the rows again from LDS: keeping the 16 registers of the MFMA's C operand alive to here costs the
                // third block per CU
                f32x4 nd4 = (f32x4)(0.f);
                if (PRE) nd4 = *(const f32x4*)(sL + 32 + 8 * g4 + 4 * h);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int r = 4 * g4 + rr;
                    const uint32_t y = attn_mix(rh[rr] + pairk);
                    const bool k = ((y >> hshift) & 0xffffu) >= a.thr16;
                    const float p0 = pd[r];
                    // dS = dscale * P (keep * dP - delta / dscale), P_drop = dscale * keep * P: the dQ kernel left delta / dscale
                    // in a.delta and the 1/(1-p) factors are applied once to dK and dV in the epilogue
                    if (PRE) dsv[r] = p0 * (k ? dp[r] : nd4[rr]);        // dp = dP - delta already; a dropped element keeps -delta
                    else dsv[r] = p0 * ((k ? dp[r] : 0.f) - dsv[r]);
                    pd[r] = k ? p0 : 0.f;
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) dsv[r] = PRE ? pd[r] * dp[r] : pd[r] * (dp[r] - dsv[r]);
        }
#pragma unroll
        for (int sstep = 0; sstep < 2; ++sstep) {
            const bf16x8 pf = pack8(&pd[8 * sstep]);
            const bf16x8 df = pack8(&dsv[8 * sstep]);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                dVT[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd_tr(sD, lo, sstep, dt), pf, dVT[dt], 0, 0, 0);
                dKT[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd_tr(sQ, lo, sstep, dt), df, dKT[dt], 0, 0, 0);
            }
        }
    }
    __syncthreads();                                           // the Q / dO tiles become the epilogue staging area
    ATTN_STAMP(1, 2);
    char* sw = smem + wave * (32 * OUT_PITCH);
    // dK = scale * dS^T Q; with PRE the staged rows are Q' = scale * log2(e) * Q, so the factor left over is ln 2
    store_rows_bf16(sw, dKT, (PRE ? 0.6931471805599453f : a.scale) * a.dscale, dK0 + (long)(wave * 32) * a.ld, a.ld, lane);
    store_rows_bf16(sw, dVT, a.dscale, dV0 + (long)(wave * 32) * a.ld, a.ld, lane);
    ATTN_STAMP(1, 3);
#ifdef APTAI_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ATTN_STAMP(1, 4);
#endif
}

// ================================================================================== backward: dQ
// grid (Tp/128, heads, B); wave w owns queries q0 = qt*128 + w*32 .. +31; loops over 32-key tiles.
template <bool PRE>
__global__ __launch_bounds__(256, 3) void attn_bwd_dq_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[DQ_STAGE + 512];      // prologue: Q and dO tiles; K, V tiles (32 keys); epilogue staging
    char* sK = smem;
    char* sV = smem + 32 * 128;
    float* sDel = (float*)(smem + DQ_STAGE);                                // delta of the block's 128 queries
    ATTN_STAMP(2, 0);
    if (a.thr16) apply_salt(a.salt, a.seed0, a.seed1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    int tile_x, hd, b;
    attn_block(a, tile_x, hd, b);
    const int qb0 = tile_x * 128;
    const int q = qb0 + wave * 32 + (lane & 31);
    int len = a.lens[b];
    len = len < 1 ? 1 : (len > a.Tp ? a.Tp : len);
    const long rowbase = (long)b * a.Tp;
    const LaneOffs lo = lane_offs(lane);
    // Prologue.  In-kernel stamps (tools/attn_stamps.py) put its first form - stage Q, barrier, fragments, barrier, stage dO, the delta
    // loop with its own loads, barrier - at 8.2 us of a 27 us block: three global-memory latencies in series.  Now every global load of
    // the prologue is issued before the first wait (Q and dO rows, the dO / O values of delta, lse2, the first K / V tile further down),
    // Q and dO tiles sit side by side in LDS, and one barrier separates the writes from the reads.
    bf16x8 qf[4], df[4];
    const bf16_t* Kb = a.qkv + rowbase * a.ld + a.H + hd * HD;
    const bf16_t* Vb = Kb + a.H;
    const int srow = tid >> 3, sch = tid & 7;
    const int soff = tile_off(srow, sch);
    const long gk = (long)srow * a.ld + sch * 8;
    u32x4 kreg = *(const u32x4*)(Kb + gk), vreg = *(const u32x4*)(Vb + gk);
    const bf16_t* Qrows = a.qkv + (rowbase + qb0) * a.ld + hd * HD;
    const bf16_t* Drows = a.dctx + (rowbase + qb0) * a.ldo + hd * HD;
    u32x4 qst[4], dst[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int cid = it * 256 + tid, row = cid >> 3, ch = cid & 7;
        qst[it] = *(const u32x4*)(Qrows + (long)row * a.ld + ch * 8);
        dst[it] = *(const u32x4*)(Drows + (long)row * a.ldo + ch * 8);
    }
    // delta = rowsum(dO * O) of the block's queries, 16 lanes per row (the fp32 context keeps it accurate): every load
    // instruction covers 4 whole rows.  Written once for the dK/dV kernel, which is launched after.
    const int part = lane & 15, rsub = lane >> 4;
    f32x4 ovv[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int row = wave * 32 + it * 4 + rsub;
        const long g = (rowbase + qb0 + row) * a.ldo + hd * HD + part * 4;
        if (a.o32) {
            ovv[it] = *(const f32x4*)(a.o32 + g);
        } else {
            const u32x2 ov = *(const u32x2*)(a.ctx + g);
            ovv[it] = (f32x4){lo_bf(ov[0]), hi_bf(ov[0]), lo_bf(ov[1]), hi_bf(ov[1])};
        }
    }
    const float lse_q = a.lse2[((long)b * a.heads + hd) * a.Tp + q];
    char* sQt = smem;
    char* sDt = smem + 128 * 128;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int cid = it * 256 + tid, row = cid >> 3, ch = cid & 7;
        *(u32x4*)(sQt + tile_off(row, ch)) = qst[it];
        *(u32x4*)(sDt + tile_off(row, ch)) = dst[it];
    }
    ATTN_STAMP(2, 5);
    __syncthreads();
    ATTN_STAMP(2, 6);
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) { qf[ds] = rd_row(sQt, lo, wave, ds); df[ds] = rd_row(sDt, lo, wave, ds); }
    // the dO values of delta come from the staged tile (the prologue is an HBM burst of every block at once: 12.6 MB less of it)
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int row = wave * 32 + it * 4 + rsub;
        const u32x2 dv = *(const u32x2*)(sDt + tile_off(row, part >> 1) + ((part & 1) << 3));
        float acc = lo_bf(dv[0]) * ovv[it][0];
        acc = fmaf(hi_bf(dv[0]), ovv[it][1], acc);
        acc = fmaf(lo_bf(dv[1]), ovv[it][2], acc);
        acc = fmaf(hi_bf(dv[1]), ovv[it][3], acc);
        acc += __shfl_xor(acc, 8, 64);
        acc += __shfl_xor(acc, 4, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 1, 64);
        if (part == 0) sDel[row] = acc;
    }
    ATTN_STAMP(2, 7);
    __syncthreads();
    float del_q = sDel[wave * 32 + (lane & 31)];
    __syncthreads();                                           // the staging area becomes the K / V tiles
    // with attention dropout dS = dscale * P (keep * dP - delta / dscale): both backward kernels work with delta / dscale and
    // apply dscale = 1/(1-p) once in their epilogues (dscale = 1 without dropout)
    del_q *= __frcp_rn(a.dscale);
    if (h == 0) a.delta[((long)b * a.heads + hd) * a.Tp + q] = del_q;
    f32x16 dQT[2];
    dQT[0] = dQT[1] = (f32x16)(0.f);
    const int nk = (len + 31) >> 5;
    uint32_t hbase = 0;                      // dropout: row hash of this lane's query + 2 h K1 (see the forward kernel)
    if (a.thr16) hbase = rng_hash((uint32_t)((b * a.heads + hd) * a.Tp + q), a.seed0, a.seed1) + (uint32_t)(2 * h) * ATTN_K1;
    const float c = a.c;
    const f32x16 cs = (f32x16)(PRE ? -lse_q : 0.f), cd = (f32x16)(PRE ? -del_q : 0.f);
    ATTN_STAMP(2, 1);
    for (int t = 0; t < nk; ++t) {
        __syncthreads();
        *(u32x4*)(sK + soff) = kreg;
        *(u32x4*)(sV + soff) = vreg;
        __syncthreads();
        if (t + 1 < nk) {
            kreg = *(const u32x4*)(Kb + (long)(t + 1) * 32 * a.ld + gk);
            vreg = *(const u32x4*)(Vb + (long)(t + 1) * 32 * a.ld + gk);
        }
        f32x16 sT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd_row(sK, lo, 0, 0), qf[0], cs, 0, 0, 0);      // PRE: starts at -lse2
        f32x16 dpT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd_row(sV, lo, 0, 0), df[0], cd, 0, 0, 0);     // PRE: starts at -delta
#pragma unroll
        for (int ds = 1; ds < 4; ++ds) {
            sT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd_row(sK, lo, 0, ds), qf[ds], sT, 0, 0, 0);
            dpT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd_row(sV, lo, 0, ds), df[ds], dpT, 0, 0, 0);
        }
        float dsv[16];
        const bool boundary = t * 32 + 32 > len;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float p = PRE ? fast_exp2(sT[r]) : fast_exp2(fmaf(sT[r], c, -lse_q));
            if (boundary && (t * 32 + acc_row(r, h) >= len)) p = 0.f;
            dsv[r] = p;
        }
        if (a.thr16) {
            const float dropped = PRE ? -del_q : 0.f;                 // PRE: dpT = dP - delta already; a dropped element keeps -delta
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const uint32_t hsh = attn_mix(hbase + (uint32_t)(t * 16 + acc_row(r, 0) / 2) * ATTN_K1);
                dpT[r] = (hsh & 0xffffu) >= a.thr16 ? dpT[r] : dropped;
                dpT[r + 1] = (hsh >> 16) >= a.thr16 ? dpT[r + 1] : dropped;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) dsv[r] *= PRE ? dpT[r] : (dpT[r] - del_q);
#pragma unroll
        for (int sstep = 0; sstep < 2; ++sstep) {
            const bf16x8 dsf = pack8(&dsv[8 * sstep]);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
                dQT[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd_tr(sK, lo, sstep, dt), dsf, dQT[dt], 0, 0, 0);
        }
    }
    __syncthreads();                                           // the K / V tiles become the epilogue staging area
    ATTN_STAMP(2, 2);
    store_rows_bf16(smem + wave * (32 * OUT_PITCH), dQT, a.scale * a.dscale,
                    a.dqkv + (rowbase + qb0 + wave * 32) * a.ld + hd * HD, a.ld, lane);
    ATTN_STAMP(2, 3);
#ifdef APTAI_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ATTN_STAMP(2, 4);
#endif
}

__device__ __forceinline__ uint32_t lds_u32x(const char* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}

// ================================================================================== exact-index mode: fused forward (inference)
// softmax(Q K^T d^-1/2 + key mask) V at fp32-class accuracy from SPLIT bf16 operands (include/aptai_hip.h, aptai_split_f32): every
// fp32 value is a sum of 2 (f32x3) or 3 (f32x6) bf16 pieces and a product of two values is the 3 (6) leading piece products,
// accumulated in fp32 by the same 32x32x16 MFMA.  Round 4 ran this as three launches per layer with the fp32 scores (201 MB) and
// the split probabilities (302 MB) in memory: 88 + 87 + 113 us per layer at 16 x 10 s, a quarter of the exact encoder pass.  Here
// the scores never leave the registers: same block / wave decomposition as attn_fwd_kernel (4 waves x 32 queries, S^T = K Q^T so
// a lane owns one query and P^T is the B operand of O^T += V^T P^T as it stands), the probabilities are split IN REGISTERS
// (p_h = bf16(p), p_m = bf16(p - p_h), ...) and the context leaves as the out-projection's split A operand.
//
// qkv: [B*Tp][ld] bf16, thirds Q | K | V of `slots * H` columns each (slots = 3 | 6), head h = K-tile h of its third:
//   [slot][64].  Q in the activation-side slot order (h h m | h h m m h l), K and V in the weight-side order
//   (h m h | h m h m l h) - what one aptai_gemm_bf16 launch with APTAI_EPI_SPLIT_OUT and split_out_bcol = H writes.
// ctx: [B*Tp][ldo] bf16, activation-side split layout ([H/64][slot][64]).
// K / V tiles (64 keys x 64 features x NPQ pieces each) arrive by LDS-DMA into a 2-deep ring; the tile swizzle (tile_off) sits on
// the global side of the copy.  The transposing V reads go through inline asm + an explicit lgkmcnt: behind an LDS-DMA the
// compiler puts vmcnt(0) in front of every ds_read_tr builtin (gemm_common.h), which would serialise the ring.
struct AttnXArgs {
    const bf16_t* qkv; long ld;
    const int* lens;
    bf16_t* ctx; long ldo;
    int B, Tp, H, heads, slots;
    float c;                        // d^-1/2 * log2(e)
    int xcd_remap;
};

__device__ __forceinline__ void tr_pair_asm(uint32_t a_lo, uint32_t a_hi, short4v& lo, short4v& hi) {
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a_lo));
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(a_hi));
}
__device__ __forceinline__ bf16x8 tr_join(const short4v lo, const short4v hi) {
    return __builtin_bit_cast(bf16x8, (short8v){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
}
// pieces of 8 fp32 values as packed bf16 fragments: h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)
template <int NPQ>
__device__ __forceinline__ void split8(const float* x, bf16x8 (&out)[NPQ]) {
    u32x4 hq, mq, lq;
    float r1[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        hq[i] = pack2bf(x[2 * i], x[2 * i + 1]);
        r1[2 * i] = x[2 * i] - lo_bf(hq[i]);                 // exact in fp32
        r1[2 * i + 1] = x[2 * i + 1] - hi_bf(hq[i]);
        mq[i] = pack2bf(r1[2 * i], r1[2 * i + 1]);
        if (NPQ == 3) lq[i] = pack2bf(r1[2 * i] - lo_bf(mq[i]), r1[2 * i + 1] - hi_bf(mq[i]));
    }
    out[0] = __builtin_bit_cast(bf16x8, hq);
    out[1] = __builtin_bit_cast(bf16x8, mq);
    if constexpr (NPQ == 3) out[2] = __builtin_bit_cast(bf16x8, lq);
}

template <int NPQ>                  // distinct pieces per value: 2 (f32x3: 3 products) or 3 (f32x6: 6 products)
__global__ __launch_bounds__(256, NPQ == 2 ? 2 : 1) void attn_exact_fwd_kernel(AttnXArgs a) {
    extern __shared__ __attribute__((aligned(16))) char xsmem[];
    constexpr int TILE = 64 * 128;                              // one piece of one tensor: 64 keys x 64 features
    constexpr int STAGE = 2 * NPQ * TILE;                       // K pieces, then V pieces
    constexpr int NPROD = NPQ == 2 ? 3 : 6;
    constexpr int PA[6] = {0, 0, 1, 1, 0, 2}, PB[6] = {0, 1, 0, 1, 2, 0};       // (activation piece, weight piece) of product i
    constexpr int ASLOT[3] = {0, 2, 5}, BSLOT[3] = {0, 1, 4};                     // where piece h / m / l sits in each slot order
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int nt = gridDim.x, total = nt * gridDim.y * gridDim.z;
    int L = blockIdx.x + nt * (blockIdx.y + gridDim.y * blockIdx.z);
    if (a.xcd_remap && (total & 7) == 0) L = (L & 7) * (total >> 3) + (L >> 3);   // all query tiles of a (utterance, head) on one XCD
    const int tile_x = L % nt, grp = L / nt, hd = grp % (int)gridDim.y, b = grp / (int)gridDim.y;
    const int q0 = tile_x * 128 + wave * 32;
    int len = a.lens[b];
    len = len < 1 ? 1 : (len > a.Tp ? a.Tp : len);
    const int ntiles = (len + 63) >> 6;
    const long rowbase = (long)b * a.Tp, third = (long)a.slots * a.H;
    const int q = q0 + (lane & 31);
    const bf16_t* Qg = a.qkv + (rowbase + q) * a.ld + (long)hd * (HD * a.slots);
    const bf16_t* Kg = a.qkv + rowbase * a.ld + third + (long)hd * (HD * a.slots);
    const bf16_t* Vg = Kg + third;
    const LaneOffs lo = lane_offs(lane);

    bf16x8 qf[NPQ][4];
#pragma unroll
    for (int i = 0; i < NPQ; ++i)
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) qf[i][ds] = *(const bf16x8*)(Qg + ASLOT[i] * 64 + ds * 16 + h * 8);

    // LDS-DMA staging: instruction `it` of a piece fills LDS bytes [(it * 256 + tid) * 16, +16) of its tile = (row, chunk') of the
    // swizzled layout, i.e. global chunk chunk' ^ tile_swz(row) of that row
    long goff[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int cid = it * 256 + tid, row = cid >> 3, cp = cid & 7;
        goff[it] = (long)row * a.ld + ((cp ^ tile_swz(row)) << 3);
    }
    const long tile_stride = 64 * a.ld;
    auto issue = [&](int t) {
        char* dst = xsmem + (t & 1) * STAGE + wave * 1024;      // wave-uniform; the hardware adds lane * 16
        const bf16_t* kt = Kg + (long)t * tile_stride;
        const bf16_t* vt = Vg + (long)t * tile_stride;
#pragma unroll
        for (int j = 0; j < NPQ; ++j)
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                __builtin_amdgcn_global_load_lds(GLB_PTR(kt + BSLOT[j] * 64 + goff[it]), LDS_PTR(dst + j * TILE + it * 4096), 16, 0, 0);
                __builtin_amdgcn_global_load_lds(GLB_PTR(vt + BSLOT[j] * 64 + goff[it]), LDS_PTR(dst + (NPQ + j) * TILE + it * 4096), 16, 0, 0);
            }
    };

    f32x16 oT[2];
    oT[0] = (f32x16)(0.f);
    oT[1] = (f32x16)(0.f);
    float ref = 0.f, l_run = 0.f;
    const float c = a.c;
    issue(0);

    auto tile = [&](auto first_flag, auto mask_flag, const int t) {
        constexpr bool FIRST = decltype(first_flag)::value, MASK = decltype(mask_flag)::value;
        const char* sK = xsmem + (t & 1) * STAGE;
        const char* sV = sK + NPQ * TILE;
        const uint32_t vbase = lds_u32x(sV);
        const int kbase = t * 64;
        float psum = 0.f;
#pragma unroll
        for (int kt2 = 0; kt2 < 2; ++kt2) {
            // V^T fragments of this half's 32 keys, every piece: issued first, they land under the S^T products
            short4v vl[NPQ][2][2], vh[NPQ][2][2];
#pragma unroll
            for (int j = 0; j < NPQ; ++j)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
                        tr_pair_asm(vbase + (uint32_t)(j * TILE + (2 * kt2 + s) * 2048 + lo.tr[dt][0]),
                                    vbase + (uint32_t)(j * TILE + (2 * kt2 + s) * 2048 + lo.tr[dt][1]), vl[j][s][dt], vh[j][s][dt]);
            // S^T = sum over the piece products of K_piece Q_piece^T: keys (kt2 * 32 + acc_row) x queries (lane & 31)
            f32x16 sT = (f32x16)(0.f);
#pragma unroll
            for (int ds = 0; ds < 4; ++ds) {
                bf16x8 kfr[NPQ];
#pragma unroll
                for (int j = 0; j < NPQ; ++j) kfr[j] = rd_row(sK + j * TILE, lo, kt2, ds);
#pragma unroll
                for (int pr = NPROD - 1; pr >= 0; --pr)          // smallest products first
                    sT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[PB[pr]], qf[PA[pr]][ds], sT, 0, 0, 0);
            }
            if (MASK) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kbase + kt2 * 32 + acc_row(r, h) >= len) sT[r] = -INFINITY;
            }
            if (FIRST && kt2 == 0) {                            // key 0 is valid: the maximum of the first 32 keys is finite
                float mt = sT[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mt = fmaxf(mt, sT[r]);
                mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
                ref = fmaf(mt, c, REF_HEADROOM);
            }
            float x[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) { x[r] = fast_exp2(fmaf(sT[r], c, -ref)); psum += x[r]; }
            bf16x8 pf[2][NPQ];
            split8<NPQ>(&x[0], pf[0]);
            split8<NPQ>(&x[8], pf[1]);
            // every V fragment of the half is back (LDS returns in order; the "+v" operands keep the consumers behind the wait)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < NPQ; ++j)
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    asm volatile("" : "+v"(vl[j][s][0]), "+v"(vh[j][s][0]), "+v"(vl[j][s][1]), "+v"(vh[j][s][1]));
            // O^T += sum over the piece products of V_piece^T P_piece^T: 2 k-steps of 16 keys
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int pr = NPROD - 1; pr >= 0; --pr)
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
                        oT[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(vl[PB[pr]][s][dt], vh[PB[pr]][s][dt]), pf[s][PA[pr]],
                                                                         oT[dt], 0, 0, 0);
        }
        psum += __shfl_xor(psum, 32, 64);
        l_run += psum;
        // the reference moves (by an exact power of two) only when a row sum says some p passed 2^14 times the level of the maximum
        // it was taken from: see attn_fwd_kernel
        if (!FIRST && __builtin_amdgcn_ballot_w64(!(psum <= REF_SUM_LIMIT))) {
            if (!(psum <= REF_SUM_LIMIT)) {
                const float e = floorf(fast_log2(psum)) + REF_HEADROOM;
                const float sc = fast_exp2(-e);
                ref += e;
                l_run *= sc;
#pragma unroll
                for (int r = 0; r < 16; ++r) { oT[0][r] *= sc; oT[1][r] *= sc; }
            }
        }
    };
    for (int t = 0; t < ntiles; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this thread's pieces of tile t (and its Q fragments) have landed
        __syncthreads();                                        // ... everyone's; and everyone has finished reading buffer (t + 1) & 1
        if (t + 1 < ntiles) issue(t + 1);                       // flies under this tile's products
        const bool last = t + 1 == ntiles;
        if (t == 0) { if (last) tile(Flag<true>{}, Flag<true>{}, 0); else tile(Flag<true>{}, Flag<false>{}, 0); }
        else if (last) tile(Flag<false>{}, Flag<true>{}, t);
        else tile(Flag<false>{}, Flag<false>{}, t);
    }
    __syncthreads();                                           // the ring becomes the epilogue staging area
    const float inv = 1.0f / l_run;
    // context -> pieces -> the activation-side slots of this head's K-tile, through the wave's staging area as whole 128-byte rows
    f32x16 pc[3][2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float o = oT[dt][r] * inv;
            const float hh = bf_round(o), r1 = o - hh, mm = bf_round(r1);
            pc[0][dt][r] = hh;
            pc[1][dt][r] = mm;
            pc[2][dt][r] = NPQ == 3 ? bf_round(r1 - mm) : 0.f;
        }
    char* sw = xsmem + wave * (32 * OUT_PITCH);
    bf16_t* dst = a.ctx + (rowbase + q0) * a.ldo + (long)hd * (HD * a.slots);
    // activation-side order: h h m | h h m m h l
    store_rows_bf16(sw, pc[0], 1.0f, dst, a.ldo, lane);
    store_rows_bf16(sw, pc[0], 1.0f, dst + 64, a.ldo, lane);
    store_rows_bf16(sw, pc[1], 1.0f, dst + 128, a.ldo, lane);
    if constexpr (NPQ == 3) {
        store_rows_bf16(sw, pc[1], 1.0f, dst + 192, a.ldo, lane);
        store_rows_bf16(sw, pc[0], 1.0f, dst + 256, a.ldo, lane);
        store_rows_bf16(sw, pc[2], 1.0f, dst + 320, a.ldo, lane);
    }
}

int fill_args(AttnArgs& a, const char* who, const void* qkv, const int32_t* lens, int64_t B, int64_t Tp, int64_t H,
              int64_t heads, float scale, float dropout_p, uint64_t seed, int q_prescaled, const void* stream) {
    APTAI_REQUIRE(qkv && lens, "%s: null pointer", who);
    APTAI_REQUIRE(B > 0 && Tp > 0 && Tp % 128 == 0, "%s: frames per utterance (%ld) must be a positive multiple of 128", who, (long)Tp);
    APTAI_REQUIRE(heads > 0 && H == heads * HD, "%s: head_dim must be 64 (H=%ld heads=%ld)", who, (long)H, (long)heads);
    memset(&a, 0, sizeof(a));
    a.qkv = (const bf16_t*)qkv; a.ld = 3 * H; a.lens = lens; a.ldo = H;
    a.B = (int)B; a.Tp = (int)Tp; a.H = (int)H; a.heads = (int)heads;
    a.scale = scale; a.c = q_prescaled ? 1.0f : scale * LOG2E;
    a.q_prescaled = q_prescaled;
    a.thr16 = drop_thr16(dropout_p); a.dscale = drop_scale(a.thr16);
    a.seed0 = (uint32_t)seed; a.seed1 = (uint32_t)(seed >> 32);
    a.salt = aptai_seed_salt(stream);
    static const int stagger = getenv("APTAI_ATTN_STAGGER") ? atoi(getenv("APTAI_ATTN_STAGGER")) : 0;
    a.stagger = stagger;
    static const int remap = getenv("APTAI_ATTN_XCD_REMAP") ? atoi(getenv("APTAI_ATTN_XCD_REMAP")) : 1;
    a.xcd_remap = remap;
    return APTAI_OK;
}

}  // namespace

#ifdef APTAI_STAMPS
extern "C" int aptai_debug_read_attn_stamps(void* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_attn_stamps), sizeof(g_attn_stamps)) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int aptai_attention_fwd(const void* qkv, const int32_t* lens, void* ctx, float* lse2, float* ctx_f32, int64_t B,
                                   int64_t Tp, int64_t H, int64_t heads, float scale, float dropout_p, uint64_t seed,
                                   int q_prescaled, void* stream_) {
    AttnArgs a;
    int rc = fill_args(a, "aptai_attention_fwd", qkv, lens, B, Tp, H, heads, scale, dropout_p, seed, q_prescaled, stream_);
    if (rc) return rc;
    APTAI_REQUIRE(ctx != nullptr, "aptai_attention_fwd: null ctx");
    a.ctx = (bf16_t*)ctx; a.lse2 = lse2; a.o32 = ctx_f32;
    if (a.thr16) {
        APTAI_LAUNCH(attn_fwd_kernel<true>, dim3((unsigned)(Tp / 128), (unsigned)heads, (unsigned)B), dim3(256), 0,
                     (hipStream_t)stream_, a);
    } else {
        APTAI_LAUNCH(attn_fwd_kernel<false>, dim3((unsigned)(Tp / 128), (unsigned)heads, (unsigned)B), dim3(256), 0,
                     (hipStream_t)stream_, a);
    }
    APTAI_CHECK_LAUNCH("attn_fwd_kernel");
    return APTAI_OK;
}

extern "C" int aptai_attention_bwd(const void* qkv, const int32_t* lens, const void* ctx, const float* ctx_f32, const void* dctx,
                                   const float* lse2, float* delta_ws, void* dqkv, int64_t B, int64_t Tp, int64_t H,
                                   int64_t heads, float scale, float dropout_p, uint64_t seed, int dctx_zero_beyond_len,
                                   int q_prescaled, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AttnArgs a;
    int rc = fill_args(a, "aptai_attention_bwd", qkv, lens, B, Tp, H, heads, scale, dropout_p, seed, q_prescaled, stream_);
    if (rc) return rc;
    APTAI_REQUIRE(ctx && dctx && lse2 && delta_ws && dqkv, "aptai_attention_bwd: null pointer");
    a.ctx = (bf16_t*)ctx; a.dctx = (const bf16_t*)dctx; a.lse2 = (float*)lse2; a.delta = delta_ws; a.o32 = (float*)ctx_f32; a.dqkv = (bf16_t*)dqkv;
    a.skip_pad_q = dctx_zero_beyond_len;
    dim3 grid((unsigned)(Tp / 128), (unsigned)heads, (unsigned)B);
    // dQ first: it computes delta for its own queries and leaves it in delta_ws for the dK/dV kernel
    if (q_prescaled) {
        APTAI_LAUNCH(attn_bwd_dq_kernel<true>, grid, dim3(256), 0, stream, a);
        APTAI_CHECK_LAUNCH("attn_bwd_dq_kernel");
        APTAI_LAUNCH(attn_bwd_dkdv_kernel<true>, grid, dim3(256), 0, stream, a);
    } else {
        APTAI_LAUNCH(attn_bwd_dq_kernel<false>, grid, dim3(256), 0, stream, a);
        APTAI_CHECK_LAUNCH("attn_bwd_dq_kernel");
        APTAI_LAUNCH(attn_bwd_dkdv_kernel<false>, grid, dim3(256), 0, stream, a);
    }
    APTAI_CHECK_LAUNCH("attn_bwd_dkdv_kernel");
    return APTAI_OK;
}

extern "C" int aptai_attention_exact_fwd(const void* qkv_split, int64_t ld, const int32_t* lens, void* ctx_split, int64_t ldo, int64_t B,
                                         int64_t Tp, int64_t H, int64_t heads, int pieces, float scale, void* stream_) {
    APTAI_REQUIRE(qkv_split && lens && ctx_split, "aptai_attention_exact_fwd: null pointer");
    APTAI_REQUIRE(B > 0 && Tp > 0 && Tp % 128 == 0, "aptai_attention_exact_fwd: frames per utterance (%ld) must be a positive multiple of 128", (long)Tp);
    APTAI_REQUIRE(heads > 0 && H == heads * HD, "aptai_attention_exact_fwd: head_dim must be 64 (H=%ld heads=%ld)", (long)H, (long)heads);
    APTAI_REQUIRE(pieces == 3 || pieces == 6, "aptai_attention_exact_fwd: pieces must be 3 (f32x3) or 6 (f32x6)");
    APTAI_REQUIRE(ld >= 3 * pieces * H && ld % 8 == 0 && ldo >= pieces * H && ldo % 8 == 0, "aptai_attention_exact_fwd: row pitches (ld=%ld, ldo=%ld) too small "
                  "for %d slots of H=%ld or not a multiple of 8", (long)ld, (long)ldo, pieces, (long)H);
    APTAI_REQUIRE((uintptr_t)qkv_split % 16 == 0 && (uintptr_t)ctx_split % 16 == 0, "aptai_attention_exact_fwd: operands must be 16-byte aligned");
    AttnXArgs a;
    a.qkv = (const bf16_t*)qkv_split; a.ld = ld; a.lens = lens; a.ctx = (bf16_t*)ctx_split; a.ldo = ldo;
    a.B = (int)B; a.Tp = (int)Tp; a.H = (int)H; a.heads = (int)heads; a.slots = pieces;
    a.c = scale * LOG2E;
    static const int remap = getenv("APTAI_ATTN_XCD_REMAP") ? atoi(getenv("APTAI_ATTN_XCD_REMAP")) : 1;
    a.xcd_remap = remap;
    const dim3 grid((unsigned)(Tp / 128), (unsigned)heads, (unsigned)B);
    if (pieces == 3) {
        constexpr int smem = 2 * 2 * 2 * 64 * 128;              // 2 stages x (K, V) x 2 pieces x 8 KiB
        static const hipError_t attr = hipFuncSetAttribute((const void*)attn_exact_fwd_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        APTAI_REQUIRE(attr == hipSuccess, "aptai_attention_exact_fwd: cannot reserve %d bytes of LDS", smem);
        APTAI_LAUNCH(attn_exact_fwd_kernel<2>, grid, dim3(256), smem, (hipStream_t)stream_, a);
    } else {
        constexpr int smem = 2 * 2 * 3 * 64 * 128;
        static const hipError_t attr = hipFuncSetAttribute((const void*)attn_exact_fwd_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        APTAI_REQUIRE(attr == hipSuccess, "aptai_attention_exact_fwd: cannot reserve %d bytes of LDS", smem);
        APTAI_LAUNCH(attn_exact_fwd_kernel<3>, grid, dim3(256), smem, (hipStream_t)stream_, a);
    }
    APTAI_CHECK_LAUNCH("attn_exact_fwd_kernel");
    return APTAI_OK;
}
