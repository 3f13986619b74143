// Multi-tensor Adam (torch.optim.Adam semantics: train/train_aptai.py:350-356 - betas, eps OUTSIDE the square root divided by
// sqrt(bias_correction2), L2 weight decay added to the gradient, no amsgrad) over a device job table, one launch per
// parameter group.  HBM-bound: 16 B read + 12 B written per parameter, plus the optional refreshed compute copy
// (bf16 for GEMM weights, fp32 for packed biases), which replaces a separate cast pass over the parameters.
#include "common.h"

namespace {

struct AdamArgs {
    const int64_t* table;       // static rows of 6: {param, exp_avg, exp_avg_sq, copy_dst (0 = none), n, copy_kind (0 bf16 / 1 fp32)}
    const int64_t* dyn;         // per-step rows of 2: {grad (0 = no gradient this step: skip), step count of THIS update (>= 1)}
    float lr, beta1, beta2, eps, weight_decay;
    float bc1, bc2_sqrt;        // filled per block from the job's own step count
};

constexpr int ADAM_CHUNK = 4096;    // elements per block: 256 threads x 4 x 4

__device__ __forceinline__ float adam_one(float p, float g, float& m, float& v, const AdamArgs& a) {
    g = fmaf(a.weight_decay, p, g);
    m = fmaf(a.beta1, m, (1.0f - a.beta1) * g);
    v = fmaf(a.beta2, v, (1.0f - a.beta2) * g * g);
    const float denom = __fsqrt_rn(v) / a.bc2_sqrt + a.eps;
    return p - (a.lr / a.bc1) * (m / denom);
}

__global__ __launch_bounds__(256) void adam_multi_kernel(AdamArgs a) {
    const int64_t* job = a.table + (long)blockIdx.y * 6;
    const int64_t* dyn = a.dyn + (long)blockIdx.y * 2;
    const long n = job[4];
    const long base = (long)blockIdx.x * ADAM_CHUNK;
    const float* g = (const float*)dyn[0];
    if (base >= n || g == nullptr) return;
    // bias corrections of this parameter's own step count (parameters skipped by LayerDrop lag behind), in double like
    // the Python reference; every thread computes the same two values
    // (one lane per block evaluates the two double-precision powers - 256 threads each did, ~200 instructions beside 16 elements of work)
    __shared__ float bc[2];
    if (threadIdx.x == 0) {
        const double step = (double)dyn[1];
        bc[0] = (float)(1.0 - pow((double)a.beta1, step));
        bc[1] = (float)sqrt(1.0 - pow((double)a.beta2, step));
    }
    __syncthreads();
    a.bc1 = bc[0];
    a.bc2_sqrt = bc[1];
    float* p = (float*)job[0];
    float* m = (float*)job[1];
    float* v = (float*)job[2];
    void* copy = (void*)job[3];
    const bool copy_f32 = job[5] != 0;
    const bool vec = (n % 4 == 0) && ((job[0] | job[1] | job[2] | dyn[0]) % 16 == 0) && (job[3] % 8 == 0);
    if (vec) {
        // all 16 loads of the thread's four chunks first: the pointers may alias as far as the compiler knows, so a rolled form waits
        // for chunk k's stores before it issues chunk k + 1's loads
        f32x4 pq[4], mq[4], vq[4], gq[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const long i = base + (long)(it * 256 + threadIdx.x) * 4;
            if (i < n) { pq[it] = *(const f32x4*)(p + i); mq[it] = *(const f32x4*)(m + i); vq[it] = *(const f32x4*)(v + i); gq[it] = *(const f32x4*)(g + i); }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const long i = base + (long)(it * 256 + threadIdx.x) * 4;
            if (i >= n) break;
            f32x4 pp = pq[it], mm = mq[it], vv = vq[it];
            const f32x4 gg = gq[it];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float mr = mm[r], vr = vv[r];
                pp[r] = adam_one(pp[r], gg[r], mr, vr, a);
                mm[r] = mr;
                vv[r] = vr;
            }
            *(f32x4*)(p + i) = pp;
            *(f32x4*)(m + i) = mm;
            *(f32x4*)(v + i) = vv;
            if (copy) {
                if (copy_f32) *(f32x4*)((float*)copy + i) = pp;
                else *(u32x2*)((bf16_t*)copy + i) = (u32x2){pack2bf(pp[0], pp[1]), pack2bf(pp[2], pp[3])};
            }
        }
    } else {
        long end = base + ADAM_CHUNK;
        end = end < n ? end : n;
        for (long i = base + threadIdx.x; i < end; i += 256) {
            float mr = m[i], vr = v[i];
            const float pn = adam_one(p[i], g[i], mr, vr, a);
            p[i] = pn;
            m[i] = mr;
            v[i] = vr;
            if (copy) {
                if (copy_f32) ((float*)copy)[i] = pn;
                else ((bf16_t*)copy)[i] = f2bf(pn);
            }
        }
    }
}

}  // namespace

extern "C" int aptai_adam_multi(const int64_t* table_dev, const int64_t* dyn_dev, int64_t njobs, int64_t max_n, float lr, float beta1,
                                float beta2, float eps, float weight_decay, void* stream) {
    APTAI_REQUIRE(table_dev && dyn_dev && njobs > 0 && njobs <= 65535 && max_n > 0, "aptai_adam_multi: bad arguments");
    APTAI_REQUIRE(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f, "aptai_adam_multi: betas must lie in [0, 1)");
    AdamArgs a;
    a.table = table_dev;
    a.dyn = dyn_dev;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay;
    a.bc1 = a.bc2_sqrt = 1.f;
    APTAI_LAUNCH(adam_multi_kernel, dim3((unsigned)ceil_div(max_n, ADAM_CHUNK), (unsigned)njobs), dim3(256), 0, (hipStream_t)stream, a);
    APTAI_CHECK_LAUNCH("adam_multi_kernel");
    return APTAI_OK;
}
