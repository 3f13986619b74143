// Shared device/host helpers for the aptai_hip kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/aptai_hip.h"

typedef uint16_t bf16_t;   // raw bfloat16 storage

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(8))) short short8v;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// ---------------------------------------------------------------------------------- error plumbing
void aptai_set_error(const char* fmt, ...);
const uint32_t* aptai_seed_salt(const void* stream);
const int32_t* aptai_frame_bounds(const void* stream);  // device pointer to {conv0 GroupNorm frame count, FIR frame bound} bound to this stream (or null), see runtime.hip   // device pointer to the dropout salt bound to this stream (or null), see runtime.hip
#define APTAI_FAIL(code, ...)            \
    do {                                 \
        aptai_set_error(__VA_ARGS__);    \
        return (code);                   \
    } while (0)
#define APTAI_REQUIRE(cond, ...)                                   \
    do {                                                           \
        if (!(cond)) APTAI_FAIL(APTAI_ERR_INVALID, __VA_ARGS__);   \
    } while (0)
// clear any stale error another library left on this thread, then launch (APTAI_CHECK_LAUNCH reads the launch's own status)
#define APTAI_LAUNCH(...)             \
    do {                              \
        (void)hipGetLastError();      \
        hipLaunchKernelGGL(__VA_ARGS__); \
    } while (0)
#define APTAI_CHECK_LAUNCH(name)                                                          \
    do {                                                                                  \
        hipError_t e__ = hipGetLastError();                                               \
        if (e__ != hipSuccess) APTAI_FAIL(APTAI_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e__)); \
    } while (0)

// ---------------------------------------------------------------------------------- bf16 <-> f32
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
    // plain cast: hipcc emits v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN preserved)
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    // vector fptrunc: ONE v_cvt_pk_bf16_f32 for the pair (two scalar casts + or/shift cost 3.5 issues per pair)
    typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
    typedef float f2_t __attribute__((ext_vector_type(2)));
    const f2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf2_t));
}
__device__ __forceinline__ float lo_bf(uint32_t v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float hi_bf(uint32_t v) { return __uint_as_float(v & 0xffff0000u); }

// ---------------------------------------------------------------------------------- wave64 reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---------------------------------------------------------------------------------- math
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }     // bare v_exp_f32
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }      // bare v_log_f32 (log2)
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    const float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7, far below the bf16 output grid): one v_rcp + one v_exp + 6 fma
__device__ __forceinline__ float erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = 1.0f - p * t * __expf(-ax * ax);
    return copysignf(e, x);
}
// GELU in the epilogues (25 M elements per FFN GEMM: the erf form cost 12.5 us of VALU time per launch, measured).
// Phi(x) ~= sigmoid(x * (a1 + a3 x^2 + a5 x^4)) on x clamped to [-7, 7]: a minimax-style logistic fit of the normal CDF
// (tools/gelu_fit.py).  |gelu - erf form| <= 3.3e-5 and |gelu' - exact| <= 1.3e-4 over all x: below the bf16 resolution of
// the stored activations for |y| > 0.01.  7 VALU + v_exp + v_rcp (was 18 + 2), the gradient adds 5 FMAs and no
// transcendental (was a second v_exp).  Coefficients carry -log2(e) so the exponential is a bare v_exp_f32.
#define APTAI_GELU_A1 1.59499531f
#define APTAI_GELU_A3 7.40885562e-2f
#define APTAI_GELU_A5 -7.23764583e-4f
#define APTAI_NLOG2E -1.4426950408889634f
__device__ __forceinline__ float gelu_sig(float xc, float x2) {      // Phi(xc), xc clamped, x2 = xc*xc
    float p = fmaf(APTAI_GELU_A5 * APTAI_NLOG2E, x2, APTAI_GELU_A3 * APTAI_NLOG2E);
    p = fmaf(p, x2, APTAI_GELU_A1 * APTAI_NLOG2E);
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(xc * p));     // bare v_rcp_f32 (1 ulp); __frcp_rn expands to the 10-instruction IEEE divide
}
__device__ __forceinline__ float gelu_fast(float x) {
    const float xc = __builtin_amdgcn_fmed3f(x, -7.0f, 7.0f);
    return x * gelu_sig(xc, xc * xc);
}
__device__ __forceinline__ float gelu_fast_grad(float x) {
    const float xc = __builtin_amdgcn_fmed3f(x, -7.0f, 7.0f);
    const float x2 = xc * xc;
    const float s = gelu_sig(xc, x2);
    float q = fmaf(5.0f * APTAI_GELU_A5, x2, 3.0f * APTAI_GELU_A3);
    q = fmaf(q, x2, APTAI_GELU_A1);
    return fmaf(fmaf(-s, s, s), xc * q, s);                             // s + x s (1 - s) d/dx[x p(x^2)]
}

// ---------------------------------------------------------------------------------- exact-index mode (csrc/exact.hip, split-out epilogue)
__device__ __forceinline__ float bf_round(float x) { return bf2f(f2bf(x)); }
// pieces of one value: p[0] >= p[1] >= p[2] in magnitude, p[0] + p[1] (+ p[2]) = x up to 2^-17 (2^-25)
__device__ __forceinline__ void split3(float x, float& h, float& m, float& l) {
    h = bf_round(x);
    const float r1 = x - h;             // exact in fp32
    m = bf_round(r1);
    l = bf_round(r1 - m);
}
// erf-form GELU of the exact mode.  erf by Abramowitz-Stegun 7.1.26 with an IEEE reciprocal and the fast exponential (erf_fast above:
// |error| <= 1.5e-7 absolute, about one fp32 ulp of the GELU value for |x| of order 1): libm's erff cost ~40 vector instructions per
// element and made the passes that apply a GELU to the conv stack's 260 M activations VALU-bound (conv0_exact_kernel 0.61 ms per call
// for 1 GB of output; tests/test_gpu_exact.py pins the mode end to end, indices and hidden states).
__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752f)); }

// ---------------------------------------------------------------------------------- counter RNG (dropout)
// One 32-bit hash per PAIR of elements (16 bits each): keep iff half >= thr16.  Forward and backward
// regenerate the same mask from (seed, logical element index), whatever their thread mapping.
__device__ __forceinline__ uint32_t rng_hash(uint32_t idx, uint32_t s0, uint32_t s1) {
    // three full-rate 24-bit multiplies (v_mul_u32_u24 / v_mad_u32_u24; v_mul_lo_u32 is quarter rate), each followed by a
    // fold of the high bits back into the low 24 that the next multiply reads.  tools/hash_quality.py measures keep-rate,
    // lag/row/seed correlations (< 1e-3) and a chi-square of both 16-bit halves for sequential indices.
    uint32_t x = idx + s0;
    x ^= x >> 15;
    x = __umul24(x, 0x9E3779u) + s1;
    x ^= x >> 13;
    x = __umul24(x, 0xC2B2AFu) + ((x >> 7) | (x << 25));
    x ^= x >> 11;
    x = __umul24(x, 0x85EBCBu) + (x >> 5);
    return x ^ (x >> 16);
}
// element index e (64-bit logical index folded to 32 bits by the caller when rows*cols < 2^32)
__device__ __forceinline__ bool drop_keep(uint64_t e, uint32_t s0, uint32_t s1, uint32_t thr16) {
    const uint32_t h = rng_hash((uint32_t)(e >> 1) ^ (uint32_t)(e >> 33) * 0x85ebca6bu, s0, s1);
    const uint32_t half = (e & 1) ? (h >> 16) : (h & 0xffffu);
    return half >= thr16;
}
// both 16-bit lanes of the pair containing element e (e even)
__device__ __forceinline__ uint32_t drop_hash_pair(uint64_t e, uint32_t s0, uint32_t s1) {
    return rng_hash((uint32_t)(e >> 1) ^ (uint32_t)(e >> 33) * 0x85ebca6bu, s0, s1);
}
// Attention dropout: two-level counter hash.  Level 1 is rng_hash of the QUERY ROW id (once per row per kernel); level 2 mixes
// in the key-pair index with one 24-bit multiply and two 16-bit folds (single SDWA xors): ~5 instructions per pair of keys
// instead of 12.  The low half decides the even key, the high half the odd key of the pair.  tools/hash2_quality.py: keep
// rate, pair/row/seed correlations (< 2e-3 at 1e6 samples) and chi-square of both halves match the one-level hash.
constexpr uint32_t ATTN_K1 = 0x9E3779u, ATTN_K2 = 0xC2B2AFu;
__device__ __forceinline__ uint32_t attn_mix(uint32_t t0) {            // t0 = rowhash + pair * ATTN_K1 (mod 2^32)
    uint32_t x = t0 ^ (t0 >> 16);
    x = __umul24(x, ATTN_K2);
    return x ^ (x >> 16);
}

__device__ __forceinline__ void apply_salt(const uint32_t* salt, uint32_t& s0, uint32_t& s1) {
    if (salt) { s0 ^= salt[0]; s1 ^= salt[1]; }
}
static inline uint32_t drop_thr16(float p) {
    if (p <= 0.f) return 0;
    long t = (long)(p * 65536.0 + 0.5);
    if (t > 65535) t = 65535;
    return (uint32_t)t;
}
static inline float drop_scale(uint32_t thr16) { return thr16 ? 65536.0f / (65536.0f - (float)thr16) : 1.0f; }

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
