// Development micro-benchmark: how fast can one CU pull L2-resident data into LDS - by LDS-DMA (global_load_lds_dwordx4) and by
// global_load_dwordx4 -> VGPR -> ds_write_b128?  Every workgroup streams the same 2 MiB window (L2 / MALL resident after the first pass)
// NITER times; 256 or 512 threads per workgroup, 1 or 2 workgroups per CU.   hipcc --offload-arch=gfx950 -O3 -o lds_dma_rate lds_dma_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int MODE, int THREADS, int STRIDE_BYTES = 0>
__global__ __launch_bounds__(THREADS) void stream_kernel(const char* __restrict__ src, long window, int niter, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6;
    constexpr int STAGE = THREADS * 16 * 4;                   // bytes per stage: 4 instructions per thread
    // every workgroup walks the window from a different offset (so the L2 sees distinct lines at a time) unless SAME is set
    long off = ((long)blockIdx.x * 65536) % window;
    unsigned acc = 0;
    for (int it = 0; it < niter; ++it) {
        char* buf = smem + (it & 1) * STAGE;
        if (MODE == 2) {
            // a GEMM operand tile: thread t fetches 16 B of row (k * THREADS + t) >> 3 at row stride `window_stride` (passed in niter's upper bits)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int cid = k * THREADS + tid, row = cid >> 3, ch = cid & 7;
                const char* p = src + off + (long)row * STRIDE_BYTES + ch * 16;
                __builtin_amdgcn_global_load_lds(GLB_PTR(p), LDS_PTR(buf + k * THREADS * 16 + wave * 1024), 16, 0, 0);
            }
        } else if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const char* p = src + off + (long)k * THREADS * 16 + tid * 16;
                __builtin_amdgcn_global_load_lds(GLB_PTR(p), LDS_PTR(buf + k * THREADS * 16 + wave * 1024), 16, 0, 0);
            }
        } else {
            u32x4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = *(const u32x4*)(src + off + (long)k * THREADS * 16 + tid * 16);
#pragma unroll
            for (int k = 0; k < 4; ++k) *(u32x4*)(buf + k * THREADS * 16 + tid * 16) = v[k];
        }
        off += MODE == 2 ? 128 : STAGE;                     // strided mode: the next K-tile = the next 128 bytes of every row
        if (MODE == 2 ? (off % STRIDE_BYTES) + 128 > STRIDE_BYTES || off + (long)(THREADS / 2) * STRIDE_BYTES + 128 > window : off + STAGE > window) off = ((long)blockIdx.x * 65536) % (window / 4);
        if ((it & 7) == 7) {                                   // touch the data now and then so nothing is optimised away
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __syncthreads();
            acc += *(const unsigned*)(smem + tid * 4);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int THREADS, int STRIDE_BYTES = 0>
void run(const char* name, const char* d, long window, int blocks_per_cu, unsigned* sink) {
    const int niter = 4096, ncu = 256;
    const int smem = 2 * THREADS * 64;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((stream_kernel<MODE, THREADS, STRIDE_BYTES>), dim3(ncu * blocks_per_cu), dim3(THREADS), smem, 0, d, window, niter, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)ncu * blocks_per_cu * niter * THREADS * 64;
    printf("%-28s threads %d x %d per CU, window %4ld KiB: %7.2f TB/s chip, %6.1f KB/us per CU\n", name, THREADS, blocks_per_cu, window >> 10,
           bytes / ms / 1e9, bytes / ms / 1e6 / ncu);
}

int main() {
    char* d; unsigned* sink;
    const long total = 64L << 20;
    hipMalloc(&d, total); hipMemset(d, 1, total); hipMalloc(&sink, 4);
    // GEMM-like strided tiles out of a 6 MiB window (rows of 1536 / 6144 bytes: K = 768 / 3072 bf16): 4 x THREADS / 8 rows x 128 B per stage
    run<2, 512, 1536>("LDS-DMA, rows @ 1536 B", d, 6L << 20, 1, sink);
    run<2, 512, 6144>("LDS-DMA, rows @ 6144 B", d, 6L << 20, 1, sink);
    run<2, 256, 1536>("LDS-DMA, rows @ 1536 B", d, 6L << 20, 2, sink);
    run<0, 512>("LDS-DMA contiguous", d, 6L << 20, 1, sink);
    for (long window : {2L << 20, 32L << 20}) {
        run<0, 256>("LDS-DMA", d, window, 1, sink);
        run<0, 256>("LDS-DMA", d, window, 2, sink);
        run<0, 512>("LDS-DMA", d, window, 1, sink);
        run<1, 256>("global_load + ds_write", d, window, 1, sink);
        run<1, 256>("global_load + ds_write", d, window, 2, sink);
        run<1, 512>("global_load + ds_write", d, window, 1, sink);
    }
    return 0;
}
