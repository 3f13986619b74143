// BiLSTM(256 -> 256, one layer, packed sequences) of the Force_APTAI regression head (models/modules.py:195,204-206), forward
// and backward through time, as COOPERATING workgroups (gfx950).
//
// Gate layout in memory (round 4): the input projections, the saved gates and the pre-activation gradients are [frame][direction][unit][gate]
// - GATE-INTERLEAVED, column dir * 1024 + unit * 4 + gate (gate = i, f, g, o), not torch's gate-major [gate][unit]: a lane then moves the
// four gates of its (utterance, unit) as ONE 16-byte access and a wave's input-projection load is 64 contiguous bytes per utterance
// (gate-major: four 4-byte accesses 1 KB apart per lane, 16-byte segments).  The host permutes the ROWS of W_ih / b_ih + b_hh once per
// step (aptai_amd/modules.py, ops.lstm_gate_perm), so x W_ih^T arrives in this order and dW_ih / db leave in it; W_hh is read as stored.
//
// The recurrence is sequential over T = 499 frames and tiny per frame (16 utterances x 256 x 1024 MACs per direction), so the
// step latency is everything.  The first build ran one 256-thread block per (utterance, direction) and re-streamed the 1 MB
// fp32 W_hh from L2 on every frame: 12 us per frame, 6 ms forward + 13.7 ms backward per training step (rocprofv3, round 2).
// Here the 16 utterances of a batch group advance TOGETHER, and W_hh never moves:
//   * a cluster = 16 workgroups per (batch group of 16 utterances, direction); workgroup k owns hidden units [16k, 16k+16)
//     = 64 gate columns, whose W_hh slice (64 KB) lives in REGISTERS for the whole kernel as the B operand of
//     v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate: exact fp32, the M = 16 rows are the 16 utterances);
//   * per frame every workgroup needs the whole h_{t-1} [16][256]: each publishes its [16][16] slice as self-tagged 4-byte
//     words (one write-through store per lane) into a double-buffered exchange area and sweeps all 4096 words of the previous
//     frame (four 16-byte L1-bypassing loads per thread) until every tag matches (guide, Guideline 16 R2: the data is the flag,
//     no fences, placement-independent).  The tag is ONE bit inside the float: |h| <= 1 leaves bit 30 of its fp32 pattern
//     clear, so the forward stores the tag there losslessly; the backward's partial sums give up their mantissa LSB (<= 1 ulp).
//     One bit suffices because a buffer is rewritten every second frame and its tag flips between consecutive rewrites
//     (tau(s) = ((s >> 1) & 1) ^ 1; the area is zeroed before every launch, so the first writes already differ);
//   * backward: workgroup k forms the pre-activation gradients of its own 64 gate columns, multiplies them with its W_hh rows
//     ([16 x 64] . [64 x 256]) and publishes the partial dh_{t-1} [16][256]; every workgroup sums the 16 partials of its own
//     units in workgroup order (deterministic).
// The exchange area is zeroed by a memset node ahead of every launch, so a replayed hipGraph is safe.  Every spin is bounded; a timeout raises the status word at the end of the workspace.
// All 16 workgroups of a cluster must be resident together: a launch holds at most 12 clusters = 192 workgroups of one wave
// per SIMD on a 256-CU chip.
#include "common.h"

namespace {

constexpr int LH = 256;                 // hidden size
constexpr int LC_WG = 16;               // workgroups per cluster
constexpr long LC_EX_GRANULES = 2L * 16 * 16 * 256;      // backward needs 2 parities x [dest 16][src 16][256]; forward 2 x 4096
constexpr long LC_EX_BYTES = LC_EX_GRANULES * 8;        // 1 MiB per cluster
constexpr int LC_MAX_BGROUPS = 6;       // batch groups (x 2 directions x 16 workgroups) per launch
constexpr unsigned LC_SPIN_LIMIT = 1u << 22;

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((ext_vector_type(2))) u64 u64x2;

struct LstmArgs {
    const float* xproj; const float* whh; const int* lens;
    float* hout; float* gates; float* cstate;
    const float* dhout; const float* gates_in; const float* cstate_in; float* dgates;
    u64* ex; unsigned* status;
    int B, Tp, T, bgroup0, nclusters;
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() waits for EVERY outstanding memory operation of the wave (vmcnt(0)):
// inside the frame loop that would put the latency of the frame-ahead loads and frame-late stores (below) back on the critical path.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }
// Gate nonlinearities on the step's critical path: bare v_exp_f32 / v_rcp_f32 (1 ulp each) instead of the libm expf / tanhf and
// an IEEE divide (~150 instructions per frame and lane): sigmoid(x) = 1 / (1 + 2^(-x log2 e)), tanh(x) = 1 - 2 / (1 + 2^(2 x log2 e))
// (absolute error ~1e-7; saturates correctly through inf / 0).
__device__ __forceinline__ float sigm_fast(float x) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float tanh_fast(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(2.8853900817779268f * x)); }
template <int SRC>
__device__ __forceinline__ float quad_bcast(float v) {                   // value of lane (quad base + SRC)
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), SRC * 0x55, 0xf, 0xf, true));
}
template <int SRC>
__device__ __forceinline__ float quad_pick(const f32x4& acc, int slot) {  // register `slot` of lane (quad base + SRC)
    const float v0 = quad_bcast<SRC>(acc[0]), v1 = quad_bcast<SRC>(acc[1]), v2 = quad_bcast<SRC>(acc[2]), v3 = quad_bcast<SRC>(acc[3]);
    return slot == 0 ? v0 : slot == 1 ? v1 : slot == 2 ? v2 : v3;
}
__device__ __forceinline__ unsigned tau(int step) { return (((unsigned)step >> 1) & 1u) ^ 1u; }
// forward words: tag in bit 30 (clear for every |v| < 2); backward words: tag in the mantissa LSB
__device__ __forceinline__ unsigned word30(unsigned tag, float v) { return (__float_as_uint(v) & ~(1u << 30)) | (tag << 30); }
__device__ __forceinline__ unsigned word0(unsigned tag, float v) { return (__float_as_uint(v) & ~1u) | tag; }
// aux = 16: sc1 (write-through store / L1-bypassing load) on the raw buffer instructions (guide, Guideline 16 R1)
__device__ __forceinline__ void spin_or_fail(unsigned& spins, unsigned* status) {
    if (++spins > LC_SPIN_LIMIT) {
        if ((threadIdx.x & 63) == 0) atomicExch(status, 1u);
        spins = 0x80000000u;                                         // tells the caller to give up
    } else {
        __builtin_amdgcn_s_sleep(1);
    }
}

// Placement (speed only, never correctness): workgroups b and b + 8 share an XCD under the observed round-robin dispatch, so
// a cluster takes the 16 workgroups {8 i + x : i = 0..15} of one XCD label x and exchanges through ONE L2; the launch spans
// 8 x 16 x ceil(clusters / 8) workgroups and the labels without a cluster exit at once.
__device__ __forceinline__ bool cluster_of_block(int nclusters, int& cluster, int& k) {
    const int x = blockIdx.x & 7, i = blockIdx.x >> 3;
    cluster = x + 8 * (i >> 4);
    k = i & 15;
    return cluster < nclusters;
}

// ================================================================================================ forward
// grid = clusters x 16, 256 threads.  wave w, lane (j = lane & 15, q = lane >> 4): gate column j of the wave's n-tile =
// (unit 16k + 4w + (j >> 2), gate j & 3); MFMA k index (q, ss) <-> hidden unit q*64 + ss; C rows = utterances 4q + reg.
__global__ __launch_bounds__(256, 1) void lstm_cluster_fwd_kernel(LstmArgs a) {
    __shared__ __attribute__((aligned(16))) float hs[16 * LH];       // h_{t-1} as MFMA A image: [c 0..15][lane][4 floats]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int cluster, k;
    if (!cluster_of_block(a.nclusters, cluster, k)) return;
    // a latency chain of 499 dependent frames: when other streams' kernels share the CU (the frozen encoder of the next batch
    // in the pipelined Force_APTAI step) these waves should win every issue arbitration
    __builtin_amdgcn_s_setprio(3);
    const int dir = cluster & 1, bg = a.bgroup0 + (cluster >> 1);
    const int j = lane & 15, q = lane >> 4;
    const int unit = k * 16 + wave * 4 + (j >> 2), gate = j & 3;
    const int col = gate * LH + unit;

    float w[64];                                                      // W_hh[col][q*64 .. q*64+63]: resident for the whole kernel
    {
        const float* wrow = a.whh + ((long)dir * 4 * LH + col) * LH + q * 64;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const f32x4 v = *(const f32x4*)(wrow + 4 * c);
            w[4 * c] = v[0]; w[4 * c + 1] = v[1]; w[4 * c + 2] = v[2]; w[4 * c + 3] = v[3];
        }
    }
    int lenr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int b = bg * 16 + 4 * q + r;
        int l = b < a.B ? a.lens[b] : 0;
        lenr[r] = l < a.T ? l : a.T;
    }
    int maxlen = 0;
    for (int i = 0; i < 16; ++i) {
        const int b = bg * 16 + i;
        int l = b < a.B ? a.lens[b] : 0;
        l = l < a.T ? l : a.T;
        maxlen = l > maxlen ? l : maxlen;
    }
    const int myrow = 4 * q + gate, myb = bg * 16 + myrow;            // the (utterance, unit) this lane finalises
    const int mylen = gate == 0 ? lenr[0] : gate == 1 ? lenr[1] : gate == 2 ? lenr[2] : lenr[3];
    float c_st = 0.f, h_st = 0.f;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.ex + (long)cluster * LC_EX_GRANULES), 0,
                                                                        (int)LC_EX_BYTES, 0x00020000);

    // The frame's own traffic beside the exchange - its input projection (4 loads per lane) and its results (h, four gates, c: 6
    // stores per lane) - shares the vector-memory queue with the polling loads, and that queue returns IN ORDER: issued where they are
    // needed, a frame's loads and the previous frame's store acknowledgements stand in front of every poll (0.5 us of 3.2 per frame with
    // warm operands, 0.9 of 3.7 with cold ones: tools/lstm_probe.py with the -DAPTAI_EXP_LSTM builds).  So both are moved to the one
    // place where nothing waits behind them: right after a poll has SUCCEEDED - the next frame's projection is fetched there, a frame
    // ahead, and the previous frame's results are written there, a frame late; they have the whole matrix / gate / publish phase plus the
    // peers' publish latency to complete before the next poll could succeed anyway.
    // (UNCONDITIONAL loads from a clamped row, masked where the value is used: a load under a condition ends in a select or a branch
    // merge right behind it, i.e. in a wait for the data where it was issued - on the critical path)
    auto load_x = [&](int s, f32x4& x) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int b = bg * 16 + 4 * q + r;
            b = b < a.B ? b : a.B - 1;
            int t = dir ? lenr[r] - 1 - s : s;
            t = t < 0 ? 0 : (t >= a.T ? a.T - 1 : t);
#if defined(APTAI_EXP_LSTM) && (APTAI_EXP_LSTM & 1)                   // development (tools/lstm_probe.py): timing without the input loads
            x[r] = 0.01f * (float)(t & 3);
#else
            x[r] = a.xproj[(((long)b * a.Tp + t) * 2 + dir) * 4 * LH + unit * 4 + gate];     // 16 lanes = 64 contiguous bytes
#endif
        }
    };
    float p_gi = 0.f, p_gf = 0.f, p_gg = 0.f, p_go = 0.f, p_c = 0.f, p_h = 0.f;       // results of the previous frame, not yet stored
    long p_row = -1;
    auto flush = [&]() {
        if (p_row >= 0) {
            a.hout[p_row * 2 * LH + dir * LH + unit] = p_h;
            if (a.gates) {
                *(f32x4*)(a.gates + (p_row * 2 + dir) * 4 * LH + unit * 4) = (f32x4){p_gi, p_gf, p_gg, p_go};
                a.cstate[(p_row * 2 + dir) * LH + unit] = p_c;
            }
        }
    };
    f32x4 x_next;
    load_x(0, x_next);
    for (int s = 0; s < maxlen; ++s) {
        const f32x4 x_cur = x_next;
        f32x4 acc;
        if (s == 0) {
            if (maxlen > 1) load_x(1, x_next);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = 0 < lenr[r] ? x_cur[r] : 0.f;
        }
        if (s > 0) {
            // ---- gather h_{s-1}: word (unit u, batch i) at u*16 + i.  Thread reads the 16-byte pieces L = r*256 + tid, r = 0..3:
            // unit u = r*64 + (tid >> 2), batches 4 (tid & 3) .. + 3
            const unsigned want = tau(s - 1);
            const unsigned gbase = (unsigned)(((s - 1) & 1) * 4096 * 4) + (unsigned)tid * 16u;
            u32x4 g[4];
            for (unsigned spins = 0;;) {
                bool ok = true;
#pragma unroll
                for (int r = 0; r < 4; ++r) g[r] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, gbase + (unsigned)r * 4096u, 0, 16);
                __builtin_amdgcn_sched_barrier(0);                     // all four loads in flight before the first tag is looked at
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    ok &= ((g[r][0] >> 30) & 1u) == want && ((g[r][1] >> 30) & 1u) == want && ((g[r][2] >> 30) & 1u) == want &&
                          ((g[r][3] >> 30) & 1u) == want;
                if (__all(ok)) break;
                spin_or_fail(spins, a.status);
                if (spins == 0x80000000u) break;
            }
#if !(defined(APTAI_EXP_LSTM) && (APTAI_EXP_LSTM & 2))                // development: timing without the per-frame result stores
            flush();                                                    // frame s-1's results, and
#endif
            if (s + 1 < maxlen) load_x(s + 1, x_next);                  // frame s+1's projection: nothing polls behind them for a while
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = s < lenr[r] ? x_cur[r] : 0.f;       // (first touch of the prefetched values: behind the poll)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // element (i, u): MFMA image word = c*256 + (qq*16 + i)*4 + e with qq = u >> 6 = r, c = (u & 63) >> 2 = tid >> 4,
                // e = u & 3 = (tid >> 2) & 3
                const int base = (tid >> 4) * 256 + (r * 16 + 4 * (tid & 3)) * 4 + ((tid >> 2) & 3);
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) hs[base + 4 * e4] = __uint_as_float(g[r][e4] & ~(1u << 30));
            }
            lds_barrier();
            // two accumulator chains: a dependent v_mfma_f32_16x16x4_f32 waits 40 cycles, an independent one issues after 32
            f32x4 acc2 = (f32x4)(0.f);
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const f32x4 av = *(const f32x4*)&hs[c * 256 + lane * 4];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], w[4 * c], acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], w[4 * c + 1], acc2, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], w[4 * c + 2], acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], w[4 * c + 3], acc2, 0, 0, 0);
            }
            acc += acc2;
            lds_barrier();                                            // the image is free for the next frame's gather
        }
        // ---- the four gates of (utterance 4q + slot, unit) sit in the four lanes of the quad, register = slot: 4 x 4 transpose
        const float gi = sigm_fast(quad_pick<0>(acc, gate)), gf = sigm_fast(quad_pick<1>(acc, gate)),
                    gg = tanh_fast(quad_pick<2>(acc, gate)), go = sigm_fast(quad_pick<3>(acc, gate));
        const float c_new = gf * c_st + gi * gg;
        const float h_new = go * tanh_fast(c_new);
        const bool act = s < mylen;
        if (act) { c_st = c_new; h_st = h_new; }
        __builtin_amdgcn_raw_buffer_store_b32(word30(tau(s), h_st), rsrc, (unsigned)(((s & 1) * 4096 + unit * 16 + myrow) * 4), 0, 16);
        p_row = -1;
        if (act) {
            const int t = dir ? mylen - 1 - s : s;
            p_row = (long)myb * a.Tp + t;
            p_gi = gi; p_gf = gf; p_gg = gg; p_go = go; p_c = c_new; p_h = h_new;
        }
    }
    flush();
    // frames beyond each utterance: zeros (pad_packed_sequence) for this workgroup's 16 units
    for (int i = 0; i < 16; ++i) {
        const int b = bg * 16 + i;
        if (b >= a.B) break;
        int l = a.lens[b];
        l = l < a.T ? l : a.T;
        for (int t = l + (tid >> 4); t < a.Tp; t += 16) a.hout[((long)b * a.Tp + t) * 2 * LH + dir * LH + k * 16 + (tid & 15)] = 0.f;
    }
}

// ================================================================================================ backward through time
// thread (eb = tid & 15, eu = tid >> 4) owns (utterance eb, unit 16k + eu) of the elementwise part; MFMA: A = the workgroup's
// pre-activation gradients [16 utterances][64 gate columns] (k index (q, ss) <-> gate q, unit 16k + ss), B = W_hh rows of those
// columns [64][256 units] in registers, wave w -> output units 64w .. 64w+63 (4 n-tiles = 4 destination workgroups).
__global__ __launch_bounds__(256, 1) void lstm_cluster_bwd_kernel(LstmArgs a) {
    __shared__ __attribute__((aligned(16))) float dgs[16 * 64];      // A image: [c 0..3][lane][4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int cluster, k;
    if (!cluster_of_block(a.nclusters, cluster, k)) return;
    // a latency chain of 499 dependent frames: when other streams' kernels share the CU (the frozen encoder of the next batch
    // in the pipelined Force_APTAI step) these waves should win every issue arbitration
    __builtin_amdgcn_s_setprio(3);
    const int dir = cluster & 1, bg = a.bgroup0 + (cluster >> 1);
    const int j = lane & 15, q = lane >> 4;
    const int eb = tid & 15, eu = tid >> 4, unit = k * 16 + eu;
    const int b = bg * 16 + eb;
    int len = b < a.B ? a.lens[b] : 0;
    len = len < a.T ? len : a.T;
    int maxlen = 0;
    for (int i = 0; i < 16; ++i) {
        const int bb = bg * 16 + i;
        int l = bb < a.B ? a.lens[bb] : 0;
        l = l < a.T ? l : a.T;
        maxlen = l > maxlen ? l : maxlen;
    }
    float w[4][16];                                                   // W_hh[q*256 + 16k + ss][64w + 16nt + j]
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int ss = 0; ss < 16; ++ss)
            w[nt][ss] = a.whh[((long)dir * 4 * LH + q * LH + k * 16 + ss) * LH + wave * 64 + nt * 16 + j];
    constexpr long PAR = 16L * 16 * 256;                              // words per parity
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.ex + (long)cluster * LC_EX_GRANULES), 0,
                                                                        (int)LC_EX_BYTES, 0x00020000);
    float dc = 0.f;
    for (int n = 0; n < maxlen; ++n) {
        const int s = maxlen - 1 - n;
        const bool act = s < len;
        const int t = dir ? len - 1 - s : s;
        const long row = (long)b * a.Tp + t;
        float gi = 0.f, gf = 0.f, gg = 0.f, go = 0.f, c = 0.f, cprev = 0.f, dho = 0.f;
#if defined(APTAI_EXP_LSTM) && (APTAI_EXP_LSTM & 4)                   // development: timing without the saved-state loads
        if (act) { gi = 0.4f; gf = 0.6f; gg = 0.1f; go = 0.5f; c = 0.2f; cprev = 0.1f; dho = 0.01f * (float)(t & 3); }
        if (false) {
#else
        if (act) {
#endif
            const f32x4 g4 = *(const f32x4*)(a.gates_in + (row * 2 + dir) * 4 * LH + unit * 4);
            gi = g4[0]; gf = g4[1]; gg = g4[2]; go = g4[3];
            c = a.cstate_in[(row * 2 + dir) * LH + unit];
            if (s > 0) cprev = a.cstate_in[((row + (dir ? 1 : -1)) * 2 + dir) * LH + unit];
            dho = a.dhout[row * 2 * LH + dir * LH + unit];
        }
        // (The frame-ahead / frame-late treatment of the forward kernel was tried here as well - 7 loads fetched a frame ahead, 4 stores
        // written a frame late, both right behind a successful poll: 3.04-3.16 us per frame against 3.06-3.09 as written, tools/lstm_probe.py -
        // no gain: this kernel's 11 per-frame accesses are 16 segments of 16 bytes each per wave, and it is their number, not their latency,
        // that the next poll waits behind.  Without them the frame takes 2.63 us.)
        const float tc = tanh_fast(c);                                // needs only this frame's loads: off the exchange's critical path
        float dh_rec = 0.f;
        if (n > 0) {
            // partial sums of iteration n-1 for this workgroup's units: [src 16][unit_l 16][batch 16], thread = (eu, eb)
            const unsigned want = tau(n - 1);
            const unsigned gbase = (unsigned)((((n - 1) & 1) * PAR + (long)k * 16 * 256 + tid) * 4);
            unsigned g[16];
            for (unsigned spins = 0;;) {
                bool ok = true;
#pragma unroll
                for (int src = 0; src < 16; ++src) g[src] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, gbase + (unsigned)src * 1024u, 0, 16);
                // all sixteen loads in flight before the first tag is looked at: left to itself the scheduler may pair each compare with
                // its load and wait in between (seen once while this loop was being restructured: eight L2 round trips one after the
                // other, 4.7 instead of 3.1 us per frame)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int src = 0; src < 16; ++src) ok &= (g[src] & 1u) == want;
                if (__all(ok)) break;
                spin_or_fail(spins, a.status);
                if (spins == 0x80000000u) break;
            }
#pragma unroll
            for (int src = 0; src < 16; ++src) dh_rec += __uint_as_float(g[src] & ~1u);
        }
        float pi = 0.f, pf = 0.f, pg = 0.f, po = 0.f;
        if (act) {
            const float dh = dho + dh_rec;
            const float d_o = dh * tc;
            dc += dh * go * (1.f - tc * tc);
            const float d_i = dc * gg, d_g = dc * gi, d_f = dc * cprev;
            pi = d_i * gi * (1.f - gi); pf = d_f * gf * (1.f - gf); pg = d_g * (1.f - gg * gg); po = d_o * go * (1.f - go);
            dc = dc * gf;
#if defined(APTAI_EXP_LSTM) && (APTAI_EXP_LSTM & 8)                   // development: timing without the per-frame gradient stores
            if (n + 1 == maxlen)
#endif
            {
                *(f32x4*)(a.dgates + (row * 2 + dir) * 4 * LH + unit * 4) = (f32x4){pi, pf, pg, po};
            }
        }
        if (n + 1 < maxlen) {
            // ---- A image: element (i = eb, kk = gate*16 + eu): word = c*256 + (gate*16 + eb)*4 + e, c = eu >> 2, e = eu & 3
            const int base = (eu >> 2) * 256 + eb * 4 + (eu & 3);
            dgs[base] = pi; dgs[base + 64] = pf; dgs[base + 128] = pg; dgs[base + 192] = po;
            lds_barrier();
            f32x4 acc[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = (f32x4)(0.f);
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const f32x4 av = *(const f32x4*)&dgs[cc * 256 + lane * 4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], w[nt][4 * cc + e], acc[nt], 0, 0, 0);
            }
            lds_barrier();
            // ---- publish the partial dh_{t-1}: element (utterance 4q + r, unit 64w + 16nt + j) -> dest 4w + nt, unit_l j
            // the lane's four utterances 4q .. 4q+3 of one (dest, unit) are four ADJACENT words: one 16-byte store
            const unsigned t = tau(n);
            const unsigned obase = (unsigned)(((n & 1) * PAR) * 4);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const unsigned off = obase + (unsigned)((((wave * 4 + nt) * 16 + k) * 16 + j) * 16 + 4 * q) * 4u;
                const u32x4 v = {word0(t, acc[nt][0]), word0(t, acc[nt][1]), word0(t, acc[nt][2]), word0(t, acc[nt][3])};
                __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, off, 0, 16);
            }
        }
    }
    // frames beyond each utterance: zero pre-activation gradients for this workgroup's 64 gate columns
    for (int i = 0; i < 16; ++i) {
        const int bb = bg * 16 + i;
        if (bb >= a.B) break;
        int l = a.lens[bb];
        l = l < a.T ? l : a.T;
        for (int t = l + (tid >> 6); t < a.Tp; t += 4) {
            float* dgp = a.dgates + (((long)bb * a.Tp + t) * 2 + dir) * 4 * LH + k * 64 + lane;       // units 16k .. 16k+15, four gates each
            *dgp = 0.f;
        }
    }
}

int launch_cluster(bool backward, LstmArgs a, void* workspace, int64_t B, hipStream_t stream) {
    const int nb = (int)ceil_div(B, 16);
    a.status = (unsigned*)workspace;                       // first 256 bytes: status word; the exchange areas follow
    for (int bg0 = 0; bg0 < nb; bg0 += LC_MAX_BGROUPS) {
        const int n = nb - bg0 < LC_MAX_BGROUPS ? nb - bg0 : LC_MAX_BGROUPS;
        a.bgroup0 = bg0;
        a.ex = (u64*)((char*)workspace + 256 + (long)bg0 * 2 * LC_EX_BYTES);
        // tags restart at 1 in every launch: the exchange area must not hold a previous launch's granules
        if (hipMemsetAsync(a.ex, 0, (size_t)n * 2 * LC_EX_BYTES, stream) != hipSuccess)
            APTAI_FAIL(APTAI_ERR_LAUNCH, "aptai_lstm: hipMemsetAsync of the exchange area failed");
        a.nclusters = n * 2;
        const unsigned grid = 8u * LC_WG * (unsigned)ceil_div(a.nclusters, 8);
        // An LDS request the kernels do not use (APTAI_LSTM_LDS_KB, default 136 of the CU's 160 KB; 0 = none), so that no LDS-staged
        // kernel of another stream - the GEMM and attention blocks of the frozen encoder in the pipelined Force_APTAI step - shares a
        // compute unit with these latency chains: beside them a frame took 1.8 x its stand-alone time (their LDS-DMA traffic fills the
        // CU's vector-memory path in front of the polling loads; wave priority does not reach there).  Measured, interleaved on one box:
        // Force_APTAI bf16 step 6.73 / 6.75 ms without, 6.36 / 6.37 ms with; exact-index f32x3 12.84 -> 12.64 ms.  Results unchanged.
        static const int lds_kb = getenv("APTAI_LSTM_LDS_KB") ? atoi(getenv("APTAI_LSTM_LDS_KB")) : 136;
        const int smem = lds_kb > 0 ? (lds_kb > 140 ? 140 : lds_kb) * 1024 : 0;
        if (smem > 0) {
            static const hipError_t e1 = hipFuncSetAttribute((const void*)lstm_cluster_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
            static const hipError_t e2 = hipFuncSetAttribute((const void*)lstm_cluster_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
            APTAI_REQUIRE(e1 == hipSuccess && e2 == hipSuccess, "aptai_lstm: cannot reserve the LDS of APTAI_LSTM_LDS_KB");
        }
        if (backward) APTAI_LAUNCH(lstm_cluster_bwd_kernel, dim3(grid), dim3(256), smem, stream, a);
        else APTAI_LAUNCH(lstm_cluster_fwd_kernel, dim3(grid), dim3(256), smem, stream, a);
        APTAI_CHECK_LAUNCH(backward ? "lstm_cluster_bwd_kernel" : "lstm_cluster_fwd_kernel");
    }
    return APTAI_OK;
}

}  // namespace

extern "C" int64_t aptai_lstm_workspace_bytes(int64_t B) { return ceil_div(B, 16) * 2 * LC_EX_BYTES + 256; }

extern "C" int aptai_lstm_fwd(const float* xproj, const float* whh, const int32_t* lens, float* hout, float* gates, float* cstate,
                              void* workspace, int64_t B, int64_t Tp, int64_t T, int64_t hidden, void* stream) {
    APTAI_REQUIRE(xproj && whh && lens && hout && workspace, "aptai_lstm_fwd: null pointer");
    APTAI_REQUIRE(hidden == LH, "aptai_lstm_fwd: built for hidden size 256");
    APTAI_REQUIRE((gates == nullptr) == (cstate == nullptr), "aptai_lstm_fwd: gates and cstate go together");
    APTAI_REQUIRE(B > 0 && T > 0 && Tp >= T, "aptai_lstm_fwd: bad sizes");
    APTAI_REQUIRE((uintptr_t)whh % 16 == 0 && (uintptr_t)workspace % 16 == 0 && (uintptr_t)gates % 16 == 0,
                  "aptai_lstm_fwd: whh / workspace / gates must be 16-byte aligned");
    LstmArgs a;
    memset(&a, 0, sizeof(a));
    a.xproj = xproj; a.whh = whh; a.lens = lens; a.hout = hout; a.gates = gates; a.cstate = cstate;
    a.B = (int)B; a.Tp = (int)Tp; a.T = (int)T;
    return launch_cluster(false, a, workspace, B, (hipStream_t)stream);
}

extern "C" int aptai_lstm_bwd(const float* dhout, const float* whh, const int32_t* lens, const float* gates, const float* cstate,
                              float* dgates, void* workspace, int64_t B, int64_t Tp, int64_t T, int64_t hidden, void* stream) {
    APTAI_REQUIRE(dhout && whh && lens && gates && cstate && dgates && workspace, "aptai_lstm_bwd: null pointer");
    APTAI_REQUIRE(hidden == LH, "aptai_lstm_bwd: built for hidden size 256");
    APTAI_REQUIRE(B > 0 && T > 0 && Tp >= T, "aptai_lstm_bwd: bad sizes");
    APTAI_REQUIRE((uintptr_t)gates % 16 == 0 && (uintptr_t)dgates % 16 == 0, "aptai_lstm_bwd: gates / dgates must be 16-byte aligned");
    LstmArgs a;
    memset(&a, 0, sizeof(a));
    a.dhout = dhout; a.whh = whh; a.lens = lens; a.gates_in = gates; a.cstate_in = cstate; a.dgates = dgates;
    a.B = (int)B; a.Tp = (int)Tp; a.T = (int)T;
    return launch_cluster(true, a, workspace, B, (hipStream_t)stream);
}
