/* aptai_hip.h — C ABI of libaptai_hip.so: the MI355X (gfx950) kernels behind APTAI's hot path.
 *
 * The reference (tobwei/APTAI) has no native/FFI layer: its hot path is Python calling torch/ATen operators
 * through HuggingFace's Wav2Vec2Model.  Each entry point below therefore cites the reference
 * *operator call site* it replaces ("HF:n" = transformers/models/wav2vec2/modeling_wav2vec2.py line n).
 *
 * Conventions
 *  - the caller owns every buffer; pointers are raw DEVICE pointers unless a parameter says "host";
 *    nothing here allocates device memory (workspaces are caller-provided, sized by *_workspace_bytes);
 *  - all launches go to the hipStream_t passed as `stream` (void*); no internal synchronisation;
 *  - return 0 (APTAI_OK) or a negative aptai_status; aptai_last_error() gives a thread-local message;
 *  - "bf16" = raw bfloat16 (uint16_t storage); activations are row-major [rows][cols] with frames as rows
 *    (channels-last); statistics, biases, norm parameters, losses and gradients of parameters are fp32;
 *  - integer outputs (lengths, argmax / alignment indices) are int64 and bit-exact by construction.
 */
#ifndef APTAI_HIP_H
#define APTAI_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    APTAI_OK = 0,
    APTAI_ERR_INVALID = -1,  /* bad argument / unsupported shape */
    APTAI_ERR_LAUNCH = -2,   /* HIP launch failure */
    APTAI_ERR_NO_DEVICE = -3
} aptai_status;

const char* aptai_last_error(void);
int aptai_version(void);
/* device sanity: returns APTAI_OK iff a gfx950 device is current; fills name (<=63 chars) if non-null */
int aptai_device_check(char* name, int name_len);

/* ------------------------------------------------------------------------------------------------ GEMM
 * C[M,N] = A . B^T with fp32 accumulation (MFMA), replacing nn.Linear / nn.Conv1d(k>1 as implicit GEMM):
 *   q/k/v/out_proj HF:495-498,522-527,546; FFN HF:556-572; feature projection HF:429-434;
 *   conv layers 1..6 HF:260-266 (A rows overlap: lda = stride*C_in < K = kernel*C_in, channels-last frames).
 * a_kmajor/b_kmajor: operand stored [K][rows] instead of [rows][K] (dgrad: B K-major; wgrad: both).
 * Epilogue order: alpha, +bias, (store out_pre), GELU, dropout, *gelu'(aux), +residual, cast. */
enum {
    APTAI_EPI_BIAS = 1,
    APTAI_EPI_GELU = 2,      /* exact erf GELU (ACT2FN["gelu"], HF:267-272,560) */
    APTAI_EPI_RESIDUAL = 4,  /* += residual[m*ldr+n] (bf16) */
    APTAI_EPI_DROPOUT = 8,   /* counter-based mask from (seed, m*N+n); scaled by 1/(1-p) */
    APTAI_EPI_DGELU = 16,    /* *= gelu'(aux[m*ldaux+n]) — backward of the FFN activation */
    APTAI_EPI_ALPHA = 32
};

typedef struct {
    const void* A; int64_t lda;     /* bf16 */
    const void* B; int64_t ldb;     /* bf16 */
    void* C; int64_t ldc;           /* bf16, or fp32 when out_f32 */
    int64_t M, N, K;                /* K % 64 == 0, N % 8 == 0 */
    int a_kmajor, b_kmajor, out_f32;
    int flags;
    const float* bias;              /* [N] fp32 */
    const void* residual; int64_t ldr;
    void* out_pre;                  /* optional bf16 [M][ldc]: value before GELU (saved for backward) */
    const void* aux; int64_t ldaux;
    float alpha;
    float dropout_p; uint64_t seed;
    int split_k;                    /* fp32 output only: K split into slabs in `workspace`, then reduced */
    int accumulate;                 /* fp32 output only: C += result */
    void* workspace; int64_t workspace_bytes;
} aptai_gemm_desc;

int aptai_gemm_bf16(const aptai_gemm_desc* desc, void* stream);
int64_t aptai_gemm_workspace_bytes(int64_t M, int64_t N, int split_k);

#ifdef __cplusplus
}
#endif
#endif /* APTAI_HIP_H */
