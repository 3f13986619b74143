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
    const float* delta;
    bf16_t* dqkv;
    int skip_pad_q;
};

// ---- LDS tile [rows][64] bf16, 128-B rows, 16-B chunk index XORed with (row & 7)
__device__ __forceinline__ int tile_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

__device__ __forceinline__ bf16x8 frag_row(const char* tile, int row, int chunk) {
    return *(const bf16x8*)(tile + tile_off(row, chunk));
}

// A operand (rows = tile columns d, depth = tile rows) for 32x32x16: element j <-> tile row
// rbase + 8*(j>>2) + 4*h + (j&3)   (h = lane>>5), matching the accumulator-as-operand k order.
__device__ __forceinline__ bf16x8 frag_tr(const char* tile, int rbase, int dt, int lane) {
    const int dgrp = (lane >> 4) & 1, h = lane >> 5, i = lane & 15, qq = i >> 2, p = i & 3;
    const int ch = dt * 4 + 2 * dgrp + (p >> 1);
    const int sub = (p & 1) << 3;
    const int r_lo = rbase + 4 * h + qq, r_hi = r_lo + 8;
    short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) short4v*)(tile + tile_off(r_lo, ch) + sub));
    short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) short4v*)(tile + tile_off(r_hi, ch) + sub));
    short8v r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, r);
}

__device__ __forceinline__ bf16x8 pack8(const float* v) {
    u32x4 u = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
    return __builtin_bit_cast(bf16x8, u);
}

// accumulator register r of a 32x32 tile <-> row (r&3) + 8*(r>>2) + 4*h
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ================================================================================== forward
// grid (Tp/128, heads, B), 256 threads; wave w: queries q0 = qt*128 + w*32 .. +31
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnArgs a) {
    if (a.thr16) apply_salt(a.salt, a.seed0, a.seed1);
    __shared__ __attribute__((aligned(16))) char smem[2 * 64 * 128];   // K tile, V tile (64 keys each)
    char* sK = smem;
    char* sV = smem + 64 * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int b = blockIdx.z, hd = blockIdx.y;
    const int q0 = blockIdx.x * 128 + wave * 32;
    int len = a.lens[b];
    len = len < 1 ? 1 : (len > a.Tp ? a.Tp : len);
    const int ntiles = (len + 63) >> 6;
    const long rowbase = (long)b * a.Tp;
    const bf16_t* Qg = a.qkv + (rowbase + q0 + (lane & 31)) * a.ld + hd * HD;
    const bf16_t* Kg = a.qkv + rowbase * a.ld + a.H + hd * HD;
    const bf16_t* Vg = Kg + a.H;

    bf16x8 qf[4];
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) qf[ds] = *(const bf16x8*)(Qg + ds * 16 + h * 8);

    f32x16 oT[2];
    oT[0] = (f32x16)(0.f);
    oT[1] = (f32x16)(0.f);
    float m_run = -INFINITY, l_run = 0.f;

    // staging registers: 512 chunks per tile, 2 per thread per tensor
    u32x4 kreg[2], vreg[2];
    auto load_tile = [&](int t) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int cid = it * 256 + tid, row = cid >> 3, ch = cid & 7;
            const long g = (long)(t * 64 + row) * a.ld + ch * 8;
            kreg[it] = *(const u32x4*)(Kg + g);
            vreg[it] = *(const u32x4*)(Vg + g);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int cid = it * 256 + tid, row = cid >> 3, ch = cid & 7;
            *(u32x4*)(sK + tile_off(row, ch)) = kreg[it];
            *(u32x4*)(sV + tile_off(row, ch)) = vreg[it];
        }
    };
    load_tile(0);
    const uint64_t ebase = ((uint64_t)(b * a.heads + hd) * a.Tp + (uint64_t)(q0 + (lane & 31))) * (uint64_t)a.Tp;

    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();
        store_tile();
        __syncthreads();
        if (t + 1 < ntiles) load_tile(t + 1);

        // S^T tiles: keys (kt2*32 + acc_row) x queries (lane&31)
        f32x16 sT[2];
#pragma unroll
        for (int kt2 = 0; kt2 < 2; ++kt2) {
            sT[kt2] = (f32x16)(0.f);
#pragma unroll
            for (int ds = 0; ds < 4; ++ds) {
                const bf16x8 kf = frag_row(sK, kt2 * 32 + (lane & 31), 2 * ds + h);
                sT[kt2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ds], sT[kt2], 0, 0, 0);
            }
        }
        // online softmax in the exp2 domain
        float x[2][16];
        float mt = -INFINITY;
        const int kbase = t * 64;
#pragma unroll
        for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kbase + kt2 * 32 + acc_row(r, h);
                const float v = key < len ? sT[kt2][r] * a.c : -INFINITY;
                x[kt2][r] = v;
                mt = fmaxf(mt, v);
            }
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = exp2f(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = exp2f(x[kt2][r] - m_new);
                psum += p;
                float pd = p;
                if (a.thr16) {
                    const int key = kbase + kt2 * 32 + acc_row(r, h);
                    pd = drop_keep(ebase + key, a.seed0, a.seed1, a.thr16) ? p * a.dscale : 0.f;
                }
                x[kt2][r] = pd;
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int r = 0; r < 16; ++r) { oT[0][r] *= alpha; oT[1][r] *= alpha; }
        // O^T += V^T P^T : 4 k-steps of 16 keys
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bf16x8 pf = pack8(&x[s >> 1][8 * (s & 1)]);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const bf16x8 vf = frag_tr(sV, 16 * s, dt, lane);
                oT[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oT[dt], 0, 0, 0);
            }
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int q = q0 + (lane & 31);
    bf16_t* og = a.ctx + (rowbase + q) * a.ldo + hd * HD;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int d = dt * 32 + 8 * g4 + 4 * h;
            u32x2 o = {pack2bf(oT[dt][4 * g4 + 0] * inv, oT[dt][4 * g4 + 1] * inv),
                       pack2bf(oT[dt][4 * g4 + 2] * inv, oT[dt][4 * g4 + 3] * inv)};
            *(u32x2*)(og + d) = o;
            if (a.o32)
                *(f32x4*)(a.o32 + (rowbase + q) * a.ldo + hd * HD + d) =
                    (f32x4){oT[dt][4 * g4 + 0] * inv, oT[dt][4 * g4 + 1] * inv, oT[dt][4 * g4 + 2] * inv, oT[dt][4 * g4 + 3] * inv};
        }
    if (a.lse2 && h == 0) a.lse2[((long)b * a.heads + hd) * a.Tp + q] = m_run + log2f(l_tot);
}

// ================================================================================== delta = rowsum(dO * O)
// one wave per (row, head): 64 elements
__global__ void attn_delta_kernel(const bf16_t* __restrict__ dctx, const bf16_t* __restrict__ ctx, const float* __restrict__ o32,
                                  float* __restrict__ delta, int B, int Tp, int H, int heads) {
    const long gw = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const long total = (long)B * Tp * heads;
    if (gw >= total) return;
    const long row = gw / heads;
    const int hd = (int)(gw % heads);
    const long off = row * H + hd * HD + lane;
    float v = bf2f(dctx[off]) * (o32 ? o32[off] : bf2f(ctx[off]));
    v = wave_sum(v);
    if (lane == 0) {
        const long bb = row / Tp, q = row % Tp;
        delta[(bb * heads + hd) * Tp + q] = v;
    }
}

// ================================================================================== backward: dK, dV
// grid (Tp/128, heads, B); wave w owns keys key0 = kt*128 + w*32 .. +31; loops over 32-query tiles.
__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv_kernel(AttnArgs a) {
    if (a.thr16) apply_salt(a.salt, a.seed0, a.seed1);
    __shared__ __attribute__((aligned(16))) char smem[2 * 32 * 128];   // Q tile, dO tile (32 queries each)
    char* sQ = smem;
    char* sD = smem + 32 * 128;
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
    u32x4 qreg, dreg;
    const int srow = tid >> 3, sch = tid & 7;          // 256 chunks per tile: one per thread
    auto load_tile = [&](int t) {
        qreg = *(const u32x4*)(Qb + (long)(t * 32 + srow) * a.ld + sch * 8);
        dreg = *(const u32x4*)(Db + (long)(t * 32 + srow) * a.ldo + sch * 8);
    };
    load_tile(0);
    const bool key_ok = key < len;
    for (int t = 0; t < nq; ++t) {
        __syncthreads();
        *(u32x4*)(sQ + tile_off(srow, sch)) = qreg;
        *(u32x4*)(sD + tile_off(srow, sch)) = dreg;
        __syncthreads();
        if (t + 1 < nq) load_tile(t + 1);
        // S[q][key] and dP[q][key]
        f32x16 s = (f32x16)(0.f), dp = (f32x16)(0.f);
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) {
            const bf16x8 qa = frag_row(sQ, lane & 31, 2 * ds + h);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[ds], s, 0, 0, 0);
            const bf16x8 da = frag_row(sD, lane & 31, 2 * ds + h);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, vf[ds], dp, 0, 0, 0);
        }
        float pd[16], dsv[16];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int qrow = t * 32 + 8 * g4 + 4 * h;
            const f32x4 l4 = *(const f32x4*)(lse + qrow);
            const f32x4 d4 = *(const f32x4*)(del + qrow);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int r = 4 * g4 + rr;
                float p = key_ok ? exp2f(s[r] * a.c - l4[rr]) : 0.f;
                float dpe = dp[r];
                float pdrop = p;
                if (a.thr16) {
                    const uint64_t e = ((uint64_t)(b * a.heads + hd) * a.Tp + (uint64_t)(qrow + rr)) * (uint64_t)a.Tp + key;
                    const bool keep = drop_keep(e, a.seed0, a.seed1, a.thr16);
                    pdrop = keep ? p * a.dscale : 0.f;
                    dpe = keep ? dpe * a.dscale : 0.f;
                }
                pd[r] = pdrop;
                dsv[r] = p * (dpe - d4[rr]);
            }
        }
#pragma unroll
        for (int sstep = 0; sstep < 2; ++sstep) {
            const bf16x8 pf = pack8(&pd[8 * sstep]);
            const bf16x8 df = pack8(&dsv[8 * sstep]);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const bf16x8 doT = frag_tr(sD, 16 * sstep, dt, lane);
                dVT[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(doT, pf, dVT[dt], 0, 0, 0);
                const bf16x8 qT = frag_tr(sQ, 16 * sstep, dt, lane);
                dKT[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qT, df, dKT[dt], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int d = dt * 32 + 8 * g4 + 4 * h;
            *(u32x2*)(dKg + d) = (u32x2){pack2bf(dKT[dt][4 * g4] * a.scale, dKT[dt][4 * g4 + 1] * a.scale),
                                         pack2bf(dKT[dt][4 * g4 + 2] * a.scale, dKT[dt][4 * g4 + 3] * a.scale)};
            *(u32x2*)(dVg + d) = (u32x2){pack2bf(dVT[dt][4 * g4], dVT[dt][4 * g4 + 1]),
                                         pack2bf(dVT[dt][4 * g4 + 2], dVT[dt][4 * g4 + 3])};
        }
}

// ================================================================================== backward: dQ
// grid (Tp/128, heads, B); wave w owns queries q0 = qt*128 + w*32 .. +31; loops over 32-key tiles.
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(AttnArgs a) {
    if (a.thr16) apply_salt(a.salt, a.seed0, a.seed1);
    __shared__ __attribute__((aligned(16))) char smem[2 * 32 * 128];   // K tile, V tile (32 keys each)
    char* sK = smem;
    char* sV = smem + 32 * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int b = blockIdx.z, hd = blockIdx.y;
    const int q = blockIdx.x * 128 + wave * 32 + (lane & 31);
    int len = a.lens[b];
    len = len < 1 ? 1 : (len > a.Tp ? a.Tp : len);
    const long rowbase = (long)b * a.Tp;
    const bf16_t* Qg = a.qkv + (rowbase + q) * a.ld + hd * HD;
    const bf16_t* Dg = a.dctx + (rowbase + q) * a.ldo + hd * HD;
    bf16x8 qf[4], df[4];
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) {
        qf[ds] = *(const bf16x8*)(Qg + ds * 16 + h * 8);
        df[ds] = *(const bf16x8*)(Dg + ds * 16 + h * 8);
    }
    const float lse_q = a.lse2[((long)b * a.heads + hd) * a.Tp + q];
    const float del_q = a.delta[((long)b * a.heads + hd) * a.Tp + q];
    f32x16 dQT[2];
    dQT[0] = dQT[1] = (f32x16)(0.f);
    const int nk = (len + 31) >> 5;
    const bf16_t* Kb = a.qkv + rowbase * a.ld + a.H + hd * HD;
    const bf16_t* Vb = Kb + a.H;
    u32x4 kreg, vreg;
    const int srow = tid >> 3, sch = tid & 7;
    auto load_tile = [&](int t) {
        kreg = *(const u32x4*)(Kb + (long)(t * 32 + srow) * a.ld + sch * 8);
        vreg = *(const u32x4*)(Vb + (long)(t * 32 + srow) * a.ld + sch * 8);
    };
    load_tile(0);
    const uint64_t ebase = ((uint64_t)(b * a.heads + hd) * a.Tp + (uint64_t)q) * (uint64_t)a.Tp;
    for (int t = 0; t < nk; ++t) {
        __syncthreads();
        *(u32x4*)(sK + tile_off(srow, sch)) = kreg;
        *(u32x4*)(sV + tile_off(srow, sch)) = vreg;
        __syncthreads();
        if (t + 1 < nk) load_tile(t + 1);
        f32x16 sT = (f32x16)(0.f), dpT = (f32x16)(0.f);
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) {
            const bf16x8 ka = frag_row(sK, lane & 31, 2 * ds + h);
            sT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qf[ds], sT, 0, 0, 0);
            const bf16x8 va = frag_row(sV, lane & 31, 2 * ds + h);
            dpT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, df[ds], dpT, 0, 0, 0);
        }
        float dsv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = t * 32 + acc_row(r, h);
            const float p = key < len ? exp2f(sT[r] * a.c - lse_q) : 0.f;
            float dpe = dpT[r];
            if (a.thr16) dpe = drop_keep(ebase + key, a.seed0, a.seed1, a.thr16) ? dpe * a.dscale : 0.f;
            dsv[r] = p * (dpe - del_q);
        }
#pragma unroll
        for (int sstep = 0; sstep < 2; ++sstep) {
            const bf16x8 dsf = pack8(&dsv[8 * sstep]);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const bf16x8 kT = frag_tr(sK, 16 * sstep, dt, lane);
                dQT[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kT, dsf, dQT[dt], 0, 0, 0);
            }
        }
    }
    bf16_t* dQg = a.dqkv + (rowbase + q) * a.ld + hd * HD;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int d = dt * 32 + 8 * g4 + 4 * h;
            *(u32x2*)(dQg + d) = (u32x2){pack2bf(dQT[dt][4 * g4] * a.scale, dQT[dt][4 * g4 + 1] * a.scale),
                                         pack2bf(dQT[dt][4 * g4 + 2] * a.scale, dQT[dt][4 * g4 + 3] * a.scale)};
        }
}

int fill_args(AttnArgs& a, const char* who, const void* qkv, const int32_t* lens, int64_t B, int64_t Tp, int64_t H,
              int64_t heads, float scale, float dropout_p, uint64_t seed) {
    APTAI_REQUIRE(qkv && lens, "%s: null pointer", who);
    APTAI_REQUIRE(B > 0 && Tp > 0 && Tp % 128 == 0, "%s: frames per utterance (%ld) must be a positive multiple of 128", who, (long)Tp);
    APTAI_REQUIRE(heads > 0 && H == heads * HD, "%s: head_dim must be 64 (H=%ld heads=%ld)", who, (long)H, (long)heads);
    memset(&a, 0, sizeof(a));
    a.qkv = (const bf16_t*)qkv; a.ld = 3 * H; a.lens = lens; a.ldo = H;
    a.B = (int)B; a.Tp = (int)Tp; a.H = (int)H; a.heads = (int)heads;
    a.scale = scale; a.c = scale * LOG2E;
    a.thr16 = drop_thr16(dropout_p); a.dscale = drop_scale(a.thr16);
    a.seed0 = (uint32_t)seed; a.seed1 = (uint32_t)(seed >> 32);
    a.salt = aptai_seed_salt();
    return APTAI_OK;
}

}  // namespace

extern "C" int aptai_attention_fwd(const void* qkv, const int32_t* lens, void* ctx, float* lse2, float* ctx_f32, int64_t B,
                                   int64_t Tp, int64_t H, int64_t heads, float scale, float dropout_p, uint64_t seed,
                                   void* stream_) {
    AttnArgs a;
    int rc = fill_args(a, "aptai_attention_fwd", qkv, lens, B, Tp, H, heads, scale, dropout_p, seed);
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
    int rc = fill_args(a, "aptai_attention_bwd", qkv, lens, B, Tp, H, heads, scale, dropout_p, seed);
    if (rc) return rc;
    APTAI_REQUIRE(ctx && dctx && lse2 && delta_ws && dqkv, "aptai_attention_bwd: null pointer");
    a.ctx = (bf16_t*)ctx; a.dctx = (const bf16_t*)dctx; a.lse2 = (float*)lse2; a.delta = delta_ws; a.dqkv = (bf16_t*)dqkv;
    a.skip_pad_q = dctx_zero_beyond_len;
    const long waves = (long)B * Tp * heads;
    APTAI_LAUNCH(attn_delta_kernel, dim3((unsigned)ceil_div(waves * 64, 256)), dim3(256), 0, stream,
                       (const bf16_t*)dctx, (const bf16_t*)ctx, ctx_f32, delta_ws, (int)B, (int)Tp, (int)H, (int)heads);
    APTAI_CHECK_LAUNCH("attn_delta_kernel");
    dim3 grid((unsigned)(Tp / 128), (unsigned)heads, (unsigned)B);
    APTAI_LAUNCH(attn_bwd_dkdv_kernel, grid, dim3(256), 0, stream, a);
    APTAI_CHECK_LAUNCH("attn_bwd_dkdv_kernel");
    APTAI_LAUNCH(attn_bwd_dq_kernel, grid, dim3(256), 0, stream, a);
    APTAI_CHECK_LAUNCH("attn_bwd_dq_kernel");
    return APTAI_OK;
}
