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

// dropout on P: one hash per PAIR of consecutive keys of one query row; pair index = (bh*Tp + q)*(Tp/2) + key/2 (32-bit)
__device__ __forceinline__ bool attn_keep(uint32_t pair, int key, uint32_t s0, uint32_t s1, uint32_t thr16) {
    const uint32_t hsh = rng_hash(pair, s0, s1);
    return ((key & 1) ? (hsh >> 16) : (hsh & 0xffffu)) >= thr16;
}

// ================================================================================== forward
// grid (Tp/128, heads, B), 256 threads; wave w: queries q0 = qt*128 + w*32 .. +31
__global__ __launch_bounds__(256, 3) void attn_fwd_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 64 * 128];   // K tile, V tile (64 keys each)
    char* sK = smem;
    char* sV = smem + 64 * 128;
    if (a.thr16) apply_salt(a.salt, a.seed0, a.seed1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int b = blockIdx.z, hd = blockIdx.y;
    const int q0 = blockIdx.x * 128 + wave * 32;
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

    f32x16 oT[2];
    oT[0] = (f32x16)(0.f);
    oT[1] = (f32x16)(0.f);
    float m_run = -INFINITY, l_run = 0.f;      // running max in the RAW score domain, running sum in the exp2 domain

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
    const uint32_t pbase = (uint32_t)(((b * a.heads + hd) * a.Tp + q) * (a.Tp >> 1));
    const float c = a.c;

    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 2; ++it) { *(u32x4*)(sK + soff[it]) = kreg[it]; *(u32x4*)(sV + soff[it]) = vreg[it]; }
        __syncthreads();
        if (t + 1 < ntiles) {
            const bf16_t* kn = Kg + (long)(t + 1) * tile_stride;
            const bf16_t* vn = Vg + (long)(t + 1) * tile_stride;
#pragma unroll
            for (int it = 0; it < 2; ++it) { kreg[it] = *(const u32x4*)(kn + goff[it]); vreg[it] = *(const u32x4*)(vn + goff[it]); }
        }
        // S^T tiles: keys (kt2*32 + acc_row) x queries (lane&31)
        f32x16 sT[2];
#pragma unroll
        for (int kt2 = 0; kt2 < 2; ++kt2) {
            sT[kt2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd_row(sK, lo, kt2, 0), qf[0], (f32x16)(0.f), 0, 0, 0);
#pragma unroll
            for (int ds = 1; ds < 4; ++ds)
                sT[kt2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd_row(sK, lo, kt2, ds), qf[ds], sT[kt2], 0, 0, 0);
        }
        const int kbase = t * 64;
        if (kbase + 64 > len) {                               // boundary tile only: mask keys >= len (wave-uniform branch)
#pragma unroll
            for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kbase + kt2 * 32 + acc_row(r, h) >= len) sT[kt2][r] = -INFINITY;
        }
        float mt = sT[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mt = fmaxf(mt, sT[0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mt = fmaxf(mt, sT[1][r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = fast_exp2((m_run - m_new) * c);
        const float mc = m_new * c;
        m_run = m_new;
        float x[2][16];
        float psum = 0.f;
#pragma unroll
        for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = fast_exp2(fmaf(sT[kt2][r], c, -mc));
                psum += p;
                x[kt2][r] = p;
            }
        if (a.thr16) {
#pragma unroll
            for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {              // registers r, r+1 are consecutive keys: one hash per pair
                    const int key = kbase + kt2 * 32 + acc_row(r, h);
                    const uint32_t hsh = rng_hash(pbase + (uint32_t)(key >> 1), a.seed0, a.seed1);
                    x[kt2][r] = (hsh & 0xffffu) >= a.thr16 ? x[kt2][r] : 0.f;          // the 1/(1-p) factor rides on the
                    x[kt2][r + 1] = (hsh >> 16) >= a.thr16 ? x[kt2][r + 1] : 0.f;      // final 1/l normalisation
                }
        }
        l_run = fmaf(l_run, alpha, psum);
#pragma unroll
        for (int r = 0; r < 16; ++r) { oT[0][r] *= alpha; oT[1][r] *= alpha; }
        // O^T += V^T P^T : 4 k-steps of 16 keys
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bf16x8 pf = pack8(&x[s >> 1][8 * (s & 1)]);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
                oT[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd_tr(sV, lo, s, dt), pf, oT[dt], 0, 0, 0);
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = __frcp_rn(l_tot) * a.dscale;
    bf16_t* og = a.ctx + (rowbase + q) * a.ldo + hd * HD;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int d = dt * 32 + 8 * g4 + 4 * h;
            const float o0 = oT[dt][4 * g4 + 0] * inv, o1 = oT[dt][4 * g4 + 1] * inv, o2 = oT[dt][4 * g4 + 2] * inv,
                        o3 = oT[dt][4 * g4 + 3] * inv;
            *(u32x2*)(og + d) = (u32x2){pack2bf(o0, o1), pack2bf(o2, o3)};
            if (a.o32) *(f32x4*)(a.o32 + (rowbase + q) * a.ldo + hd * HD + d) = (f32x4){o0, o1, o2, o3};
        }
    if (a.lse2 && h == 0) a.lse2[((long)b * a.heads + hd) * a.Tp + q] = fmaf(m_run, c, fast_log2(l_tot));
}

// ================================================================================== backward: dK, dV
// grid (Tp/128, heads, B); wave w owns keys key0 = kt*128 + w*32 .. +31; loops over 32-query tiles.
__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 32 * 128 + 256];   // Q tile, dO tile (32 queries each), lse|delta
    char* sQ = smem;
    char* sD = smem + 32 * 128;
    float* sL = (float*)(smem + 2 * 32 * 128);                             // [0,32) lse2, [32,64) delta of the tile's queries
    if (a.thr16) apply_salt(a.salt, a.seed0, a.seed1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int b = blockIdx.z, hd = blockIdx.y;
    const int key0 = blockIdx.x * 128 + wave * 32;
    int len = a.lens[b];
    len = len < 1 ? 1 : (len > a.Tp ? a.Tp : len);
    const long rowbase = (long)b * a.Tp;
    const int key = key0 + (lane & 31);
    bf16_t* dKg = a.dqkv + (rowbase + key) * a.ld + a.H + hd * HD;
    bf16_t* dVg = dKg + a.H;
    if (blockIdx.x * 128 >= len) {          // whole block beyond the utterance: gradients are exactly zero
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *(u32x4*)(dKg + (h * 4 + i) * 8) = z;
            *(u32x4*)(dVg + (h * 4 + i) * 8) = z;
        }
        return;
    }
    const LaneOffs lo = lane_offs(lane);
    const bf16_t* Kg = a.qkv + (rowbase + key) * a.ld + a.H + hd * HD;
    const bf16_t* Vg = Kg + a.H;
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) {
        kf[ds] = *(const bf16x8*)(Kg + ds * 16 + h * 8);
        vf[ds] = *(const bf16x8*)(Vg + ds * 16 + h * 8);
    }
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
    const uint32_t half = (uint32_t)(a.Tp >> 1);
    const uint32_t pkey = (uint32_t)((b * a.heads + hd) * a.Tp) * half + (uint32_t)(key >> 1);   // + q*half per element
    for (int t = 0; t < nq; ++t) {
        __syncthreads();
        *(u32x4*)(sQ + soff) = qreg;
        *(u32x4*)(sD + soff) = dreg;
        if (tid < 16) *(f32x4*)(sL + tid * 4) = sreg;
        __syncthreads();
        if (t + 1 < nq) {
            qreg = *(const u32x4*)(Qb + (long)(t + 1) * 32 * a.ld + gq);
            dreg = *(const u32x4*)(Db + (long)(t + 1) * 32 * a.ldo + gd);
            if (tid < 16) sreg = *(const f32x4*)(stat + (t + 1) * 32);
        }
        // S[q][key] and dP[q][key]
        f32x16 s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd_row(sQ, lo, 0, 0), kf[0], (f32x16)(0.f), 0, 0, 0);
        f32x16 dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd_row(sD, lo, 0, 0), vf[0], (f32x16)(0.f), 0, 0, 0);
#pragma unroll
        for (int ds = 1; ds < 4; ++ds) {
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd_row(sQ, lo, 0, ds), kf[ds], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd_row(sD, lo, 0, ds), vf[ds], dp, 0, 0, 0);
        }
        float pd[16], dsv[16];
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
        if (wave_boundary) {                                   // wave-uniform: only the wave holding key len-1 masks
#pragma unroll
            for (int r = 0; r < 16; ++r) pd[r] = key_ok ? pd[r] : 0.f;
        }
        if (a.thr16) {
            // the pair (keys 2k, 2k+1) of one query shares a hash and sits on lanes l, l^1: the even lane hashes the even
            // register's query, the odd lane the odd register's, and one DPP quad-permute hands each its partner's value
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const uint32_t mine = rng_hash(pkey + (uint32_t)(t * 32 + acc_row(r, h) + (lane & 1)) * half, a.seed0, a.seed1);
                const uint32_t other = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0xB1, 0xf, 0xf, false);
                const uint32_t h0 = (lane & 1) ? other : mine, h1 = (lane & 1) ? mine : other;
                const bool k0 = ((h0 >> hshift) & 0xffffu) >= a.thr16, k1 = ((h1 >> hshift) & 0xffffu) >= a.thr16;
                const float p0 = pd[r], p1 = pd[r + 1];
                // dS = dscale * P (keep * dP - delta / dscale), P_drop = dscale * keep * P: the dQ kernel left delta / dscale in
                // a.delta and the 1/(1-p) factors are applied once to dK and dV in the epilogue
                dsv[r] = p0 * ((k0 ? dp[r] : 0.f) - dsv[r]);
                dsv[r + 1] = p1 * ((k1 ? dp[r + 1] : 0.f) - dsv[r + 1]);
                pd[r] = k0 ? p0 : 0.f;
                pd[r + 1] = k1 ? p1 : 0.f;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) dsv[r] = pd[r] * (dp[r] - dsv[r]);
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
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int d = dt * 32 + 8 * g4 + 4 * h;
            const float ks = a.scale * a.dscale, vs = a.dscale;
            *(u32x2*)(dKg + d) = (u32x2){pack2bf(dKT[dt][4 * g4] * ks, dKT[dt][4 * g4 + 1] * ks),
                                         pack2bf(dKT[dt][4 * g4 + 2] * ks, dKT[dt][4 * g4 + 3] * ks)};
            *(u32x2*)(dVg + d) = (u32x2){pack2bf(dVT[dt][4 * g4] * vs, dVT[dt][4 * g4 + 1] * vs),
                                         pack2bf(dVT[dt][4 * g4 + 2] * vs, dVT[dt][4 * g4 + 3] * vs)};
        }
}

// ================================================================================== backward: dQ
// grid (Tp/128, heads, B); wave w owns queries q0 = qt*128 + w*32 .. +31; loops over 32-key tiles.
__global__ __launch_bounds__(256, 3) void attn_bwd_dq_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 32 * 128];   // K tile, V tile (32 keys each)
    char* sK = smem;
    char* sV = smem + 32 * 128;
    if (a.thr16) apply_salt(a.salt, a.seed0, a.seed1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int b = blockIdx.z, hd = blockIdx.y;
    const int q = blockIdx.x * 128 + wave * 32 + (lane & 31);
    int len = a.lens[b];
    len = len < 1 ? 1 : (len > a.Tp ? a.Tp : len);
    const long rowbase = (long)b * a.Tp;
    const bf16_t* Qg = a.qkv + (rowbase + q) * a.ld + hd * HD;
    const bf16_t* Dg = a.dctx + (rowbase + q) * a.ldo + hd * HD;
    const LaneOffs lo = lane_offs(lane);
    bf16x8 qf[4], df[4];
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) {
        qf[ds] = *(const bf16x8*)(Qg + ds * 16 + h * 8);
        df[ds] = *(const bf16x8*)(Dg + ds * 16 + h * 8);
    }
    const float lse_q = a.lse2[((long)b * a.heads + hd) * a.Tp + q];
    // delta = rowsum(dO * O) of this lane's query: each half-wave lane holds 32 of the 64 head dims (the dO fragments it
    // feeds the MFMAs with); the fp32 context keeps it accurate.  Written once for the dK/dV kernel, which is launched after.
    float del_q = 0.f;
    if (a.o32) {
        const float* Og = a.o32 + (rowbase + q) * a.ldo + hd * HD;
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) {
            const f32x4 o0 = *(const f32x4*)(Og + ds * 16 + h * 8), o1 = *(const f32x4*)(Og + ds * 16 + h * 8 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) del_q = fmaf((float)df[ds][j], o0[j], fmaf((float)df[ds][j + 4], o1[j], del_q));
        }
    } else {
        const bf16_t* Og = a.ctx + (rowbase + q) * a.ldo + hd * HD;
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) {
            const bf16x8 o = *(const bf16x8*)(Og + ds * 16 + h * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) del_q = fmaf((float)df[ds][j], (float)o[j], del_q);
        }
    }
    del_q += __shfl_xor(del_q, 32, 64);
    // with attention dropout dS = dscale * P (keep * dP - delta / dscale): both backward kernels work with delta / dscale and
    // apply dscale = 1/(1-p) once in their epilogues (dscale = 1 without dropout)
    del_q *= __frcp_rn(a.dscale);
    if (h == 0) a.delta[((long)b * a.heads + hd) * a.Tp + q] = del_q;
    f32x16 dQT[2];
    dQT[0] = dQT[1] = (f32x16)(0.f);
    const int nk = (len + 31) >> 5;
    const bf16_t* Kb = a.qkv + rowbase * a.ld + a.H + hd * HD;
    const bf16_t* Vb = Kb + a.H;
    const int srow = tid >> 3, sch = tid & 7;
    const int soff = tile_off(srow, sch);
    const long gk = (long)srow * a.ld + sch * 8;
    u32x4 kreg = *(const u32x4*)(Kb + gk), vreg = *(const u32x4*)(Vb + gk);
    const uint32_t pbase = (uint32_t)(((b * a.heads + hd) * a.Tp + q) * (a.Tp >> 1));
    const float c = a.c;
    for (int t = 0; t < nk; ++t) {
        __syncthreads();
        *(u32x4*)(sK + soff) = kreg;
        *(u32x4*)(sV + soff) = vreg;
        __syncthreads();
        if (t + 1 < nk) {
            kreg = *(const u32x4*)(Kb + (long)(t + 1) * 32 * a.ld + gk);
            vreg = *(const u32x4*)(Vb + (long)(t + 1) * 32 * a.ld + gk);
        }
        f32x16 sT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd_row(sK, lo, 0, 0), qf[0], (f32x16)(0.f), 0, 0, 0);
        f32x16 dpT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd_row(sV, lo, 0, 0), df[0], (f32x16)(0.f), 0, 0, 0);
#pragma unroll
        for (int ds = 1; ds < 4; ++ds) {
            sT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd_row(sK, lo, 0, ds), qf[ds], sT, 0, 0, 0);
            dpT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd_row(sV, lo, 0, ds), df[ds], dpT, 0, 0, 0);
        }
        float dsv[16];
        const bool boundary = t * 32 + 32 > len;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float p = fast_exp2(fmaf(sT[r], c, -lse_q));
            if (boundary && (t * 32 + acc_row(r, h) >= len)) p = 0.f;
            dsv[r] = p;
        }
        if (a.thr16) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const int key = t * 32 + acc_row(r, h);
                const uint32_t hsh = rng_hash(pbase + (uint32_t)(key >> 1), a.seed0, a.seed1);
                dpT[r] = (hsh & 0xffffu) >= a.thr16 ? dpT[r] : 0.f;
                dpT[r + 1] = (hsh >> 16) >= a.thr16 ? dpT[r + 1] : 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) dsv[r] *= (dpT[r] - del_q);
#pragma unroll
        for (int sstep = 0; sstep < 2; ++sstep) {
            const bf16x8 dsf = pack8(&dsv[8 * sstep]);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
                dQT[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd_tr(sK, lo, sstep, dt), dsf, dQT[dt], 0, 0, 0);
        }
    }
    bf16_t* dQg = a.dqkv + (rowbase + q) * a.ld + hd * HD;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int d = dt * 32 + 8 * g4 + 4 * h;
            const float qs = a.scale * a.dscale;
            *(u32x2*)(dQg + d) = (u32x2){pack2bf(dQT[dt][4 * g4] * qs, dQT[dt][4 * g4 + 1] * qs),
                                         pack2bf(dQT[dt][4 * g4 + 2] * qs, dQT[dt][4 * g4 + 3] * qs)};
        }
}

int fill_args(AttnArgs& a, const char* who, const void* qkv, const int32_t* lens, int64_t B, int64_t Tp, int64_t H,
              int64_t heads, float scale, float dropout_p, uint64_t seed, const void* stream) {
    APTAI_REQUIRE(qkv && lens, "%s: null pointer", who);
    APTAI_REQUIRE(B > 0 && Tp > 0 && Tp % 128 == 0, "%s: frames per utterance (%ld) must be a positive multiple of 128", who, (long)Tp);
    APTAI_REQUIRE(heads > 0 && H == heads * HD, "%s: head_dim must be 64 (H=%ld heads=%ld)", who, (long)H, (long)heads);
    memset(&a, 0, sizeof(a));
    a.qkv = (const bf16_t*)qkv; a.ld = 3 * H; a.lens = lens; a.ldo = H;
    a.B = (int)B; a.Tp = (int)Tp; a.H = (int)H; a.heads = (int)heads;
    a.scale = scale; a.c = scale * LOG2E;
    a.thr16 = drop_thr16(dropout_p); a.dscale = drop_scale(a.thr16);
    a.seed0 = (uint32_t)seed; a.seed1 = (uint32_t)(seed >> 32);
    a.salt = aptai_seed_salt(stream);
    return APTAI_OK;
}

}  // namespace

extern "C" int aptai_attention_fwd(const void* qkv, const int32_t* lens, void* ctx, float* lse2, float* ctx_f32, int64_t B,
                                   int64_t Tp, int64_t H, int64_t heads, float scale, float dropout_p, uint64_t seed,
                                   void* stream_) {
    AttnArgs a;
    int rc = fill_args(a, "aptai_attention_fwd", qkv, lens, B, Tp, H, heads, scale, dropout_p, seed, stream_);
    if (rc) return rc;
    APTAI_REQUIRE(ctx != nullptr, "aptai_attention_fwd: null ctx");
    a.ctx = (bf16_t*)ctx; a.lse2 = lse2; a.o32 = ctx_f32;
    APTAI_LAUNCH(attn_fwd_kernel, dim3((unsigned)(Tp / 128), (unsigned)heads, (unsigned)B), dim3(256), 0,
                       (hipStream_t)stream_, a);
    APTAI_CHECK_LAUNCH("attn_fwd_kernel");
    return APTAI_OK;
}

extern "C" int aptai_attention_bwd(const void* qkv, const int32_t* lens, const void* ctx, const float* ctx_f32, const void* dctx,
                                   const float* lse2, float* delta_ws, void* dqkv, int64_t B, int64_t Tp, int64_t H,
                                   int64_t heads, float scale, float dropout_p, uint64_t seed, int dctx_zero_beyond_len, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AttnArgs a;
    int rc = fill_args(a, "aptai_attention_bwd", qkv, lens, B, Tp, H, heads, scale, dropout_p, seed, stream_);
    if (rc) return rc;
    APTAI_REQUIRE(ctx && dctx && lse2 && delta_ws && dqkv, "aptai_attention_bwd: null pointer");
    a.ctx = (bf16_t*)ctx; a.dctx = (const bf16_t*)dctx; a.lse2 = (float*)lse2; a.delta = delta_ws; a.o32 = (float*)ctx_f32; a.dqkv = (bf16_t*)dqkv;
    a.skip_pad_q = dctx_zero_beyond_len;
    dim3 grid((unsigned)(Tp / 128), (unsigned)heads, (unsigned)B);
    // dQ first: it computes delta for its own queries and leaves it in delta_ws for the dK/dV kernel
    APTAI_LAUNCH(attn_bwd_dq_kernel, grid, dim3(256), 0, stream, a);
    APTAI_CHECK_LAUNCH("attn_bwd_dq_kernel");
    APTAI_LAUNCH(attn_bwd_dkdv_kernel, grid, dim3(256), 0, stream, a);
    APTAI_CHECK_LAUNCH("attn_bwd_dkdv_kernel");
    return APTAI_OK;
}
