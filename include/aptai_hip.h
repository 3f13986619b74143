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
/* Per-step dropout salt, bound to ONE stream: device pointer to two uint32 words that every seeded kernel launched (or
 * captured) on `stream` XORs into its dropout seed; null clears the binding.  A captured hipGraph thereby draws fresh masks
 * on each replay; forward and backward of one step stay consistent.  Launches on other streams are unaffected (no
 * process-global state: two runners / two models on two streams are independent).  The caller keeps the two words alive
 * until it clears the binding. */
int aptai_set_seed_salt(void* stream, const void* device_ptr_2xu32);
/* Per-step frame bounds, bound to ONE stream like the salt: device pointer to two int32 words {frames of the first conv layer that
 * count for its GroupNorm statistics (HF:317-323), frames that exist for LowPassFilterLayer's zero padding (models/modules.py:46-61)};
 * null clears the binding.  The reference pads every batch to ITS OWN longest utterance (train/train_aptai.py:268-285), and both
 * operations see that padded length; a hipGraph captured for a longer bucket length replays with the bounds of the batch it is fed
 * (aptai_conv0_fwd / aptai_conv0_bwd in group mode and aptai_lowpass_fir read them), so it equals the eager run on the batch's
 * own shape.  The caller keeps the two words alive until it clears the binding. */
int aptai_set_frame_bounds(void* stream, const void* device_ptr_2xi32);

/* ------------------------------------------------------------------------------------------------ GEMM
 * C[M,N] = A . B^T with fp32 accumulation (MFMA), replacing nn.Linear / nn.Conv1d(k>1 as implicit GEMM):
 *   q/k/v/out_proj HF:495-498,522-527,546; FFN HF:556-572; feature projection HF:429-434;
 *   conv layers 1..6 HF:260-266 (A rows overlap: lda = stride*C_in < K = kernel*C_in, channels-last frames).
 * a_kmajor/b_kmajor: operand stored [K][rows] instead of [rows][K] (dgrad: B K-major; wgrad: both).
 * Epilogue order: alpha, +bias, (store out_pre), GELU, dropout, *gelu'(aux), +residual, cast. */
enum {
    APTAI_EPI_BIAS = 1,
    APTAI_EPI_GELU = 2,      /* GELU in place of ACT2FN["gelu"] (HF:267-272,560): x * sigmoid(x (a1 + a3 x^2 + a5 x^4)), a logistic fit of the
                              * normal CDF: |y - erf form| <= 3.3e-5, |y' - exact| <= 1.3e-4 (below the bf16 grid of the stored output) */
    APTAI_EPI_RESIDUAL = 4,  /* += residual[m*ldr+n] (bf16) */
    APTAI_EPI_DROPOUT = 8,   /* counter-based mask from (seed, m*N+n); scaled by 1/(1-p) */
    APTAI_EPI_DGELU = 16,    /* *= gelu'(aux[m*ldaux+n]) — backward of the FFN activation */
    APTAI_EPI_ALPHA = 32,
    APTAI_EPI_PRE_DGELU = 64, /* with EPI_GELU: out_pre receives dropmask/(1-p) * gelu'(pre-activation) instead of the pre-activation */
    APTAI_EPI_MUL_AUX = 128,  /* *= aux[m*ldaux+n] (bf16): the backward partner of EPI_PRE_DGELU */
    APTAI_EPI_RESIDUAL_F32 = 256, /* fp32 output only (tile 64 / 128 / 192): += residual[m*ldr+n] read as FP32 - the residual stream of the
                                   * inference-only encoder kept in fp32 (HF:594-601: hidden_states = attn_residual + hidden_states) */
    APTAI_EPI_BIAS_ROW = 1024,    /* out_f32 launches only: bias is indexed by the output ROW (bias[m], length M) - a product evaluated transposed,
                                   * C^T = W . X^T, carries its Linear bias along the rows (the exact mode's V^T projection) */
    APTAI_EPI_SPLIT_OUT = 512     /* out_f32 launches on tile 128 / 192 / 256 (exact-index mode): the fp32 result [-> erf GELU with EPI_GELU] leaves as
                                   * `split_out_pieces` bf16 pieces in the activation-side layout of aptai_split_f32, [m][N/64][piece][64],
                                   * C = bf16*, ldc (and the C batch strides) in bf16 elements, ldc >= pieces * N: the next split-operand
                                   * GEMM's A operand, written once instead of an fp32 store, a re-read and a split pass */
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
    /* optional 2-level batching (grouped positional conv: outer = utterance, inner = group); strides in elements */
    int batch_outer, batch_inner;
    int64_t batch_stride_a[2], batch_stride_b[2], batch_stride_c[2], batch_stride_bias[2], batch_stride_res[2],
        batch_stride_aux[2];
    int tile;                       /* 0 = auto, 64 = 64x128 tile (3 blocks/CU, K-contiguous A), 128 = 128x128 tile kernel (2 blocks/CU), 192 = 128x192 tile, 3-stage ring
                                       (1 block/CU), 256 = 256x256 deep-pipelined kernel, 448 = 256x192 tile (bf16 out, K-contiguous A, one problem), 257 = the 256x256 tile as a persistent stream-K
                                       kernel (one workgroup per CU, equal shares of (tile, K-tile) iterations; needs sk_workspace) */
    int colscale_n; float colscale; /* bf16 output: columns [0, colscale_n) *= colscale after alpha / bias (colscale_n % 8 == 0).  The fused
                                       q|k|v projection (HF:495-498) hands the attention kernels Q already multiplied by
                                       head_dim^-0.5 * log2(e) (HF:522 applies the scaling to the projected query too), rounded once */
    void* sk_workspace;             /* optional, aptai_gemm_sk_workspace_bytes() bytes, ZEROED ONCE by the caller and then owned by ONE stream:
                                       fp32 partial-tile slabs + ready flags (self-cleaning) + status word of the stream-K kernel.  With it the
                                       auto rule may pick the stream-K form; without it tile 257 is refused and auto never picks it */
    int64_t sk_workspace_bytes;
    int split_out_pieces;           /* with APTAI_EPI_SPLIT_OUT: 3 or 6 */
    int split_out_bcol;             /* with APTAI_EPI_SPLIT_OUT: output columns n >= split_out_bcol are written in the WEIGHT-side piece order
                                       (hi lo hi | hi mid hi mid low hi), columns below it in the activation-side order; N (or more) = all
                                       activation-side.  A fused q|k projection hands Q (A operand of Q K^T) and K (its B operand) over at once */
} aptai_gemm_desc;

int aptai_gemm_bf16(const aptai_gemm_desc* desc, void* stream);
/* n (<= 8) independent problems of ONE operand layout / output type in a single launch (no batching, split-K or
 * accumulate).  Made for the per-layer weight and bias gradients of the encoder backward, which autograd computes as
 * separate `grad_out.t() @ input` / `grad_out.sum(0)` kernels behind HF:478-480,575-654 (nn.Linear backward): together
 * their full-K tiles fill the 256 CUs once, where each alone needs split-K slabs and a reduce pass.  A bias gradient is
 * the problem M = 8, A = ones[K][8] (K-major), row 0 of the [8][N] fp32 result. */
int aptai_gemm_bf16_grouped(const aptai_gemm_desc* descs, int n, void* stream);
int64_t aptai_gemm_workspace_bytes(int64_t M, int64_t N, int split_k);
/* Stream-K workspace (independent of the problem: one 256 x 256 fp32 slab per CU + flags) and its status word: non-zero after a
 * launch in which a bounded wait for another workgroup's slab gave up (that launch's output is then incomplete; never a hang).
 * aptai_gemm_sk_status copies the word to the host behind everything queued on `stream` (synchronises) and clears it. */
int64_t aptai_gemm_sk_workspace_bytes(void);
int aptai_gemm_sk_status(void* sk_workspace, void* stream, int* status_out);

/* ------------------------------------------------------------------------------------------------ exact (fp32-class) inference path
 * Wav2Vec2Model.set_encoder_precision("f32x3" | "f32x6"): the frozen recogniser inside Force_APTAI (models/force_aptai.py:60-78,
 * 124-127, inference only) with every matrix product at fp32-class accuracy, so that the alignment argmax (:148-161) and the
 * best-path decode see the reference's fp32 scores.  aptai_split_f32 turns an fp32 operand [rows][cols] into bf16 pieces in the
 * K-tile-interleaved layout [row][cols/64][pieces][64] (pattern 0 = activation side: hi,hi,lo | hi,hi,mid,mid,hi,low; pattern 1 =
 * weight side: hi,lo,hi | hi,mid,hi,mid,low,hi), which aptai_gemm_bf16 (NT, fp32 output, K' = pieces * K, lda' = pieces * lda)
 * multiplies as hi.hi + hi.lo + lo.hi (+ mid.mid + hi.low + low.hi): relative error ~2^-17 (3 pieces) / ~2^-24 (6 pieces) with
 * fp32 accumulation.  act = 1 applies the erf-form GELU (ACT2FN["gelu"], HF:267-272,560) before the split. */
int aptai_split_f32(const float* x, int64_t ldx, int64_t rows, int64_t cols, int pattern, int pieces, int act, void* out, int64_t ldo,
                    void* stream);
/* y = [res +] act(x + bias) in fp32 (act 0 none, 1 erf GELU); with lens (int32 [rows / rows_per_b]) rows t >= lens[b] of each block
 * of rows_per_b rows are zeroed (padded frames, HF:678-681).  In place (y == x) allowed. */
int aptai_bias_act_res_f32(const float* x, int64_t ldx, const float* bias, const float* res, int64_t ldr, float* y, int64_t ldy,
                           int64_t rows, int64_t cols, int act, const int32_t* lens, int64_t rows_per_b, void* stream);
/* softmax over keys, in place on fp32 scores s[B][heads][Tp][Tp]; keys >= lens[b] get probability 0 (HF:452-461 under the
 * finfo.min key mask of HF:1018-1036). */
int aptai_softmax_rows_f32(float* s, const int32_t* lens, int64_t B, int64_t heads, int64_t Tp, void* stream);
/* The same masked softmax written straight into the split (activation-side) layout [row][Tp/64][piece][64] of its probabilities - the A
 * operand of the exact attention's P . V product (aptai_amd/wav2vec2.py _exact_attention; replaces torch softmax at HF:452-461 followed by
 * aptai_split_f32: the fp32 probabilities are never stored).  s fp32 [B][heads][Tp][Tp]; out bf16, row pitch ldo >= pieces * Tp. */
int aptai_softmax_split_f32(const float* s, const int32_t* lens, int64_t B, int64_t heads, int64_t Tp, int pieces, void* out, int64_t ldo,
                            void* stream);
/* The whole attention core of the exact-index mode in ONE launch: softmax(Q K^T * scale + key mask) V per (utterance, head), every product
 * as the 3 (pieces = 3, "f32x3") or 6 (pieces = 6, "f32x6") leading bf16 piece products accumulated in fp32, the scores and probabilities
 * never stored (replaces HF:452-461 on the path of models/force_aptai.py:148-161 whose argmax indices must be the reference's; round 4's
 * three launches - aptai_gemm_bf16 scores, aptai_softmax_split_f32, aptai_gemm_bf16 P . V - moved ~0.8 GB per layer at 16 x 10 s).
 * qkv_split bf16 [B*Tp][ld >= 3 * pieces * H]: thirds Q | K | V of pieces * H columns, head h = columns [64 * pieces * h, +64 * pieces)
 * of its third as [slot][64]; Q in the activation-side slot order, K and V in the weight-side order - what one aptai_gemm_bf16 launch with
 * APTAI_EPI_SPLIT_OUT, split_out_pieces = pieces and split_out_bcol = H writes for the concatenated q|k|v weight.  lens int32 [B]: keys
 * >= lens[b] get probability 0, query rows beyond it are computed like the reference's.  ctx_split bf16 [B*Tp][ldo >= pieces * H] in the
 * activation-side split layout: the out-projection's A operand.  Tp % 128 == 0, head_dim 64.  Inference only (no dropout, no backward). */
int aptai_attention_exact_fwd(const void* qkv_split, int64_t ld, const int32_t* lens, void* ctx_split, int64_t ldo, int64_t B, int64_t Tp,
                              int64_t H, int64_t heads, int pieces, float scale, void* stream);
/* Conv1d(1,512,10,5) + GroupNorm / LayerNorm + erf GELU with FP32 output [B][T_alloc][512] (HF:260-323); `stats` = the (mean, rstd)
 * block [B][2][512] of aptai_conv0_fwd in group mode (mode 0), unused in layer mode (mode 1). */
int aptai_conv0_fwd_f32(const float* audio, int64_t B, int64_t S, const float* weight, const float* bias, const float* gamma,
                        const float* beta, int mode, float eps, float* out, int64_t T_real, int64_t T_alloc, const float* stats,
                        void* stream);
/* The same layer with its result written as split bf16 pieces [B][T_alloc][512 / 64][piece][64] - the A operand of the exact mode's second
 * conv layer (round 4: the 1 GB fp32 intermediate at 16 x 10 s is never stored).  Frames in [T_real, T_alloc) are written as zeros. */
int aptai_conv0_fwd_split(const float* audio, int64_t B, int64_t S, const float* weight, const float* bias, const float* gamma,
                          const float* beta, int mode, float eps, void* out_split, int pieces, int64_t T_real, int64_t T_alloc,
                          const float* stats, void* stream);

/* ------------------------------------------------------------------------------------------------ LayerNorm
 * y = (x - mean) * rstd * gamma + beta over the channel axis (cols in {256,512,768,1024}), one wave per row.
 * Replaces nn.LayerNorm at HF:288-299 (conv layers, "layer" mode; gelu_after=1 fuses the GELU of HF:299),
 * HF:425-431, HF:587-601 / 622-644, HF:691 / 791 and models/modules.py:134,151.  x,y bf16; mean/rstd fp32 (may be null). */
int aptai_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                        int64_t rows, int64_t cols, float eps, int gelu_after, void* stream);
/* dx = LN'(dy) [+ dres]; optional dx_drop = dropout-masked copy of dx (mask of (seed, element index), for the
 * branch that went through nn.Dropout, HF:591,628); dgamma/dbeta fp32 [cols] (overwritten). */
/* nn.LayerNorm forward on an FP32 row (the fp32 residual stream of the inference-only encoder): y_bf16 and / or y_f32 receive the
 * normalised row (bf16 for the next GEMM operand, fp32 for the next residual add).  cols in {256, 512, 768, 1024}. */
int aptai_layernorm_fwd_f32in(const float* x, const float* gamma, const float* beta, void* y_bf16, float* y_f32, int64_t rows,
                              int64_t cols, float eps, void* stream);
/* The same LayerNorm with its result ALSO written as split bf16 pieces [rows][cols / 64][piece][64] (activation-side order): the A operand of
 * the exact mode's next split-operand product, without a split pass of its own.  y_f32 may be null. */
int aptai_layernorm_fwd_f32in_split(const float* x, const float* gamma, const float* beta, float* y_f32, void* y_split, int pieces,
                                    int64_t rows, int64_t cols, float eps, void* stream);
int aptai_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                        const void* dres, void* dx, void* dx_drop, float dropout_p, uint64_t seed, float* dgamma,
                        float* dbeta, void* workspace, int64_t rows, int64_t cols, const float* beta_if_gelu_after,
                        void* stream);   /* beta_if_gelu_after != null: dy is the gradient of gelu(LN(x)) (conv stack, HF:299) */
int64_t aptai_layernorm_bwd_workspace_bytes(int64_t rows, int64_t cols);
/* The dgamma / dbeta reductions of MANY aptai_layernorm_bwd calls (each called with dgamma = dbeta = null, its workspace kept) in one
 * launch: table_dev = device int64 [njobs][5] = {workspace of that call, dgamma | 0, dbeta | 0, blocks = workspace_bytes / (8 * cols),
 * cols}; max_cols = the widest job.  Replaces the per-call finalisation behind nn.LayerNorm's parameter gradients (HF:587-601). */
int aptai_layernorm_bwd_finalize_multi(const int64_t* table_dev, int64_t njobs, int64_t max_cols, void* stream);

/* ------------------------------------------------------------------------------------------------ attention
 * softmax(Q K^T * scale + key-padding mask) V per head (head_dim 64), flash-style, replacing HF:452-461 / sdpa
 * as called from HF:522-546, and its backward.  qkv [B*Tp][3H] bf16; lens int32 [B] valid frames (keys beyond are
 * masked, HF:678-688); Tp % 128 == 0; ctx [B*Tp][H] bf16; lse2 fp32 [B][heads][Tp] (log2-domain log-sum-exp).
 * dropout_p: attention-probability dropout (HF:458), mask from (seed, (b,h,q,k)).
 * ctx_f32 (optional, [B*Tp][H] fp32): unrounded context; the backward's delta = rowsum(dO*O) is a small-difference term
 * (dS = P*(dP - delta)) whose bf16 rounding would otherwise dominate the q/k gradients when attention is diffuse.
 * q_prescaled != 0: the Q third of qkv already carries scale * log2(e) (aptai_gemm_desc.colscale of the fused q|k|v projection):
 * the scores arrive in the exp2 domain, the backward starts its S / dP accumulators at -lse2 / -delta, and dqkv's Q third is
 * still the gradient with respect to the UNSCALED projection output (what the projection's own backward expects). */
int aptai_attention_fwd(const void* qkv, const int32_t* lens, void* ctx, float* lse2, float* ctx_f32, int64_t B, int64_t Tp,
                        int64_t H, int64_t heads, float scale, float dropout_p, uint64_t seed, int q_prescaled, void* stream);
/* delta_ws: fp32 [B][heads][Tp] scratch.  dctx_zero_beyond_len=1 lets the kernel skip query rows >= lens[b]
 * (their incoming gradient is exactly zero in the models: no loss term touches padded frames). */
int aptai_attention_bwd(const void* qkv, const int32_t* lens, const void* ctx, const float* ctx_f32, const void* dctx,
                        const float* lse2, float* delta_ws, void* dqkv, int64_t B, int64_t Tp, int64_t H, int64_t heads, float scale,
                        float dropout_p, uint64_t seed, int dctx_zero_beyond_len, int q_prescaled, void* stream);

/* ------------------------------------------------------------------------------------------------ parameter prep
 * fp32 master parameters -> bf16 compute copies (and the layouts the kernels want). */
int aptai_cast_f32_to_bf16(const float* src, void* dst, int64_t rows, int64_t cols, int64_t ld_dst, void* stream);
/* Many casts in one launch.  table_dev: device int64 [njobs][4] = {src fp32 ptr, dst ptr, n, kind}; n % 8 == 0, both
 * pointers 32-/16-byte aligned; kind 0 = fp32 -> bf16, 1 = fp32 -> fp32 copy (packs q/k/v biases); max_n = largest n. */
int aptai_cast_multi(const int64_t* table_dev, int64_t njobs, int64_t max_n, void* stream);
/* torch.optim.Adam step (train/train_aptai.py:350-356, 443) for many tensors in one launch.  table_dev: device int64
 * [njobs][6] = {param, exp_avg, exp_avg_sq, copy_dst or 0, n, copy_kind (0 = bf16, 1 = fp32)} (static across steps);
 * dyn_dev: device int64 [njobs][2] = {grad or 0 (no gradient this step: the job is skipped), step count of this update >= 1}
 * (rewritten every step).  All tensors fp32 with n elements; copy_dst receives the updated parameter (the compute copy the
 * forward reads).  Bias corrections 1 - beta^step are evaluated per job in double, as torch does on the host. */
int aptai_adam_multi(const int64_t* table_dev, const int64_t* dyn_dev, int64_t njobs, int64_t max_n, float lr, float beta1,
                     float beta2, float eps, float weight_decay, void* stream);
/* nn.Conv1d weight [N][C][Kw] (HF:260-266) -> [N][Kw*C] bf16, K index = kw*C + c (channels-last frames) */
int aptai_conv_weight_to_bf16(const float* src, void* dst, int64_t N, int64_t C, int64_t Kw, void* stream);
/* positional conv (HF:329-356): weight_norm(dim=2) w = g*v/||v||_(0,1) ; v [H][H/groups][Kw], gain [Kw];
 * w_fwd [groups][Cg][Kw*Cg] (forward), w_dgrad [groups][Cg][Kw*Cg] (flipped taps, in/out swapped; may be null);
 * norm_ws fp32 [257 * Kw]: the first Kw entries receive ||v|| per tap (needed by the weight-norm backward), the rest is
 * scratch for the two-stage reduction; Kw must divide 256. */
int aptai_posconv_weight(const float* v, const float* gain, float* norm_ws, void* w_fwd, void* w_dgrad, int64_t H,
                         int64_t groups, int64_t Kw, void* stream);
/* Backward of that weight-norm parametrisation: from dw_fwd fp32 [groups][Cg][Kw*Cg] (the weight gradient in the forward layout)
 * to dv [H][Cg][Kw] and dgain [Kw] (HF parametrizations.weight.original1 / original0).  workspace fp32 [(H + 1) * Kw]. */
int aptai_posconv_weight_bwd(const float* dw_fwd, const float* v, const float* gain, const float* norm, float* dv, float* dgain,
                             float* workspace, int64_t H, int64_t groups, int64_t Kw, void* stream);
/* The grouped convolution itself for 48 channels per group and 128 taps (wav2vec2-base), replacing the batched implicit GEMM:
 * out[b*Tp+t][grp*48+n] = residual + act(bias + sum_{kw,c} xg[grp][b][first_row + t + kw][c] * w[grp][n][kw*48+c]);
 * xg as produced by aptai_posconv_pack ([groups][B][pad+Tp+pad][48], gap rows zero), w = w_fwd (forward, first_row 0,
 * bias + GELU, out_pre = pre-activation) or w_dgrad (data gradient, first_row 1); Tp % 128 == 0. */
int aptai_posconv_gemm(const void* xg, int64_t first_row, const void* w, const float* bias, const void* residual, void* out,
                       void* out_pre, int64_t B, int64_t Tp, int64_t H, int64_t groups, int64_t Kw, int64_t pad, int gelu,
                       void* stream);
/* Weight gradient of the same convolution: dw[grp][co][kw*48+ci] (fp32, overwritten) = sum_f du_g[grp][pad + f][co] *
 * x_g[grp][f + kw][ci] over the frame axis of the packed copies (gap rows are zero), replacing the batched TN GEMM. */
int aptai_posconv_wgrad(const void* du_g, const void* x_g, float* dw, int64_t B, int64_t Tp, int64_t H, int64_t groups, int64_t Kw,
                        int64_t pad, void* stream);
/* SpecAugment time mask sampled on the device (HF:101-217 `_compute_mask_indices` as called at HF:1292-1304): mask_u8 [B][T]
 * (overwritten), frame_lens int32 [B] valid frames per utterance.  Same span-count rule and without-replacement span starts as
 * the reference; counter-hash random stream keyed by `seed` (and the device seed salt). */
int aptai_spec_augment_mask(const int32_t* frame_lens, void* mask_u8, int64_t B, int64_t T, float mask_prob, int64_t mask_length,
                            int64_t min_masks, uint64_t seed, void* stream);
/* x [B*Tp][H] bf16 -> group-major, zero-gapped xg [groups][B][pad+Tp+pad][Cg] (gap rows must be pre-zeroed once);
 * with u != null the packed value is x*gelu'(u) (backward of HF:362) and rowmajor_out also receives it. */
int aptai_posconv_pack(const void* x, const void* u, void* xg, void* rowmajor_out, int64_t B, int64_t Tp, int64_t H,
                       int64_t groups, int64_t pad, void* stream);

/* ------------------------------------------------------------------------------------------------ frame masking
 * In place on h [B*Tp][H] bf16: frames t >= min(lens[b], T) -> 0 (HF:678-681); frames with spec_mask[b][t] != 0
 * (uint8 [B][T], sampled on the host exactly like HF:101-217) -> masked_spec_embed (HF:1292-1295). */
int aptai_frame_mask_fwd(void* h, const int32_t* lens, const uint8_t* spec_mask, const float* embed, int64_t B, int64_t Tp,
                         int64_t T, int64_t H, void* stream);
int aptai_frame_mask_bwd(void* dy, const int32_t* lens, const uint8_t* spec_mask, float* dembed, void* workspace, int64_t B,
                         int64_t Tp, int64_t T, int64_t H, void* stream);
int64_t aptai_frame_mask_bwd_workspace_bytes(int64_t B, int64_t Tp, int64_t H);

/* out[n] (+)= sum_rows x[row][n]  — bias gradients */
int aptai_colsum_bf16(const void* x, int64_t ld, float* out, void* workspace, int64_t rows, int64_t N, int accumulate,
                      void* stream);
int64_t aptai_colsum_workspace_bytes(int64_t rows, int64_t N);

/* y = keep ? x/(1-p) : 0 with the counter mask of (seed, element index) — standalone nn.Dropout forward/backward */
int aptai_dropout_bf16(const void* x, void* y, int64_t n, float p, uint64_t seed, void* stream);
/* out = dy * gelu'(u) — backward of a standalone GELU (top of the conv stack, HF:267-272) */
int aptai_dgelu_bf16(const void* dy, const void* u, void* out, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------------ conv layer 0
 * Conv1d(1,512,k=10,s=5) on the raw waveform fused with GroupNorm+GELU (mode 0, HF:302-323) or
 * bias+LayerNorm+GELU (mode 1, HF:275-299).  audio fp32 [B][S]; out bf16 [B][T_alloc][512], frames >= T_real zeroed. */
int aptai_conv0_fwd(const float* audio, int64_t B, int64_t S, const float* weight, const float* bias, const float* gamma,
                    const float* beta, int mode, float eps, void* out, int64_t T_real, int64_t T_alloc, int64_t C,
                    int64_t Kw, int64_t stride, void* workspace, float* stats_out, void* stream);
int64_t aptai_conv0_workspace_bytes(int64_t B, int64_t T_real);
/* backward of the same fused layer: recomputes the conv from the waveform, folds GELU' and the GroupNorm / LayerNorm
 * backward, and reduces dweight [512][10] (+ dbias, dgamma, dbeta) deterministically.  dy bf16 [B][T_alloc][512];
 * fwd_stats fp32 [B][2][512] = (mean, rstd) written by the forward (`stats_out`), group mode only. */
int aptai_conv0_bwd(const float* audio, int64_t B, int64_t S, const float* weight, const float* bias, const float* gamma,
                    const float* beta, int mode, float eps, const void* dy, int64_t T_real, int64_t T_alloc,
                    const float* fwd_stats, float* dweight, float* dbias, float* dgamma, float* dbeta, void* workspace,
                    void* stream);
int64_t aptai_conv0_bwd_workspace_bytes(int64_t B, int64_t T_real);

/* ------------------------------------------------------------------------------------------------ APTAI heads
 * a_tv = tanh(dropout(h)), a_ph = leaky_relu(dropout(h)) — the activations in front of the two head Linears
 * (models/aptai.py:43-55); the Linears themselves run on aptai_gemm_bf16 with N padded to 64. */
int aptai_head_act_fwd(const void* h, void* a_tv, void* a_ph, int64_t n, float p_tv, float p_ph, uint64_t seed, void* stream);
int aptai_head_act_bwd(const void* h, const void* d_tv, const void* d_ph, void* dh, int64_t n, float p_tv, float p_ph,
                       uint64_t seed, void* stream);
/* LowPassFilterLayer (models/modules.py:46-61): 51-tap 'same' FIR along time, fp64 accumulate, per channel.
 * x fp32 rows (b*rows_per_b_in + t) stride ldx; y fp32 or bf16 rows (b*rows_per_b_out + t) stride ldy; frames in
 * [T, T_out) and channels in [C, C_out) of y are zero-filled.  Symmetric taps: the same call is its own backward. */
int aptai_lowpass_fir(const float* x, int64_t ldx, int64_t rows_per_b_in, const double* taps, int64_t ntaps, void* y,
                      int64_t ldy, int64_t rows_per_b_out, int out_bf16, int64_t B, int64_t T, int64_t T_out, int64_t C,
                      int64_t C_out, void* stream);
/* loss = w_mse * MSE(tv_pred[mask], tv_tgt[mask]) + w_ce * CE(logits[phn!=0], phn) and argmax (models/aptai.py:89-106;
 * models/force_aptai.py:137-141 with w_ce = 0).  scalars fp32[5] = loss, mse, ce, #valid tv elements, #valid frames. */
int aptai_aptai_loss_fwd(const float* tv_pred, const float* tv_tgt, const float* logits, int64_t ldl, int64_t rows_per_b,
                         const int64_t* phn_tgt, int64_t B, int64_t T, int64_t n_tv, int64_t n_phn, float w_mse, float w_ce,
                         float* scalars, int64_t* pred, void* workspace, void* stream);
int aptai_aptai_loss_bwd(const float* tv_pred, const float* tv_tgt, const float* logits, int64_t ldl, int64_t rows_per_b,
                         const int64_t* phn_tgt, int64_t B, int64_t T, int64_t n_tv, int64_t n_phn, float w_mse, float w_ce,
                         const float* scalars, const float* grad_out, float* d_tv, void* d_logits_bf16, int64_t ldd,
                         void* stream);
int64_t aptai_aptai_loss_workspace_bytes(void);

/* ------------------------------------------------------------------------------------------------ MX block-scaled FP8 GEMM
 * BASELINE configs[4] ("fp8 MFMA weights"): the frozen, inference-only encoder of Force_APTAI with OCP MXFP8 operands - E4M3
 * elements, one E8M0 scale per 32 consecutive k - on v_mfma_scale_f32_32x32x64_f8f6f4 (twice the bf16 MFMA rate on gfx950; no
 * reference counterpart: the reference is fp32).  aptai_mx_quantize_bf16: x bf16 [rows][K] -> q (1 byte / element,
 * row stride ldq) + scales (1 byte per 32 k, row stride lds); scale = 2^(floor(log2 amax) - 8), elements round-to-nearest-even,
 * saturated to +-448.  aptai_gemm_mxfp8: C bf16 [M][N] = dequant(A) . dequant(B)^T + bias, optional GELU, + residual;
 * replaces nn.Linear of the encoder layers (HF:495-498,546,556-572) when the model is switched to "mxfp8" inference. */
int aptai_mx_quantize_bf16(const void* x, int64_t ldx, void* q, int64_t ldq, void* scales, int64_t lds, int64_t rows, int64_t K,
                           void* stream);
int aptai_gemm_mxfp8(const void* A, const void* A_scales, int64_t lda, int64_t ldas, const void* B, const void* B_scales, int64_t ldb,
                     int64_t ldbs, void* C, int64_t ldc, const float* bias, int gelu, const void* residual, int64_t ldr, int64_t M,
                     int64_t N, int64_t K, void* stream);
/* The same product with an MXFP8 RESULT (elements Cq [M][ldcq], scales C_scales [M][ldcs], N % 32 == 0): bit-identical to aptai_gemm_mxfp8
 * followed by aptai_mx_quantize_bf16 of its bf16 output, without that tensor's trip through HBM - the FFN1 -> FFN2 hand-over of the
 * encoder layer (HF:556-572: intermediate_dense + GELU feeding output_dense). */
/* nn.LayerNorm (HF:500,556 of the pre-LN layer) whose result leaves as MXFP8 (q [rows][ldq], scales [rows][lds], one per 32 columns) and,
 * if y_bf16 is not null, as bf16 too: bit-identical to aptai_layernorm_fwd followed by aptai_mx_quantize_bf16. */
int aptai_layernorm_fwd_mx(const void* x, const float* gamma, const float* beta, void* y_bf16, void* q, int64_t ldq, void* scales,
                           int64_t lds, int64_t rows, int64_t cols, float eps, void* stream);
int aptai_gemm_mxfp8_mxout(const void* A, const void* A_scales, int64_t lda, int64_t ldas, const void* B, const void* B_scales, int64_t ldb,
                           int64_t ldbs, void* Cq, int64_t ldcq, void* C_scales, int64_t ldcs, const float* bias, int gelu, int64_t M,
                           int64_t N, int64_t K, void* stream);

/* ------------------------------------------------------------------------------------------------ CTC
 * log_softmax + CTC negative log-likelihood (alpha recursion) and its gradient w.r.t. the LOGITS (beta recursion),
 * replacing nn.functional.log_softmax + F.ctc_loss at models/w2v2_pr.py:59,73-81 and the per-sample nn.CTCLoss loop
 * of ForwardSumLoss (models/modules.py:99-116; vocab_sizes[b] = N_b + 1 classes, targets 1..N_b).
 * logits fp32 rows (b*rows_per_b + t) stride ldl; targets int32 [B][ldt] (any padding beyond target_lens[b]);
 * reduction 0 none / 1 mean (mean_b nll_b / max(len_b,1)) / 2 sum; nll fp32 [B]; loss fp32 scalar;
 * log_probs_out fp32 (T,B,V) or null; workspace (aptai_ctc_workspace_bytes) holds alpha | beta | per-state log-probs and is
 * handed unchanged to aptai_ctc_bwd.  want_beta != 0 runs the beta recursion beside the alpha recursion in the same launch
 * (both are sequential over T and independent of each other): the backward is then one parallel pass. */
int aptai_ctc_fwd(const float* logits, int64_t ldl, int64_t rows_per_b, const int32_t* targets, int64_t ldt,
                  const int32_t* input_lens, const int32_t* target_lens, const int32_t* vocab_sizes, int64_t B, int64_t T,
                  int64_t V, int blank, int reduction, int zero_infinity, float* log_probs_out, float* workspace, float* nll,
                  float* loss, int want_beta, void* stream);
/* dlogits rows (b*rows_per_b + t) stride ldd (V <= ldd <= 256), bf16 or fp32, zero outside t < input_lens[b] and v < V;
 * gradient = grad_out[0] (device scalar, may be null = 1) * extra_scale * d loss / d logits.  beta_ready = the forward call
 * was made with want_beta (otherwise the beta recursion runs here first). */
int aptai_ctc_bwd(const float* logits, int64_t ldl, int64_t rows_per_b, const int32_t* targets, int64_t ldt,
                  const int32_t* input_lens, const int32_t* target_lens, const int32_t* vocab_sizes, int64_t B, int64_t T,
                  int64_t V, int blank, int reduction, int zero_infinity, const float* workspace, const float* nll,
                  const float* grad_out, float extra_scale, void* dlogits, int64_t ldd, int out_bf16, int beta_ready, void* stream);
int64_t aptai_ctc_workspace_bytes(int64_t B, int64_t T, int64_t ldt);
/* Best-path CTC decode on the device (stand-in for the torchaudio beam decoder the reference calls at models/w2v2_pr.py:143-159,
 * which is absent here: parity unpinned): per utterance, frame argmax (first maximum) over ALL T rows, repeats collapsed, blank
 * dropped.  ids_out int32 [B][max_n] zero-padded (the aligner's phoneme slots, models/force_aptai.py:109-115), n_out int32 [B] =
 * decoded length, which the caller compares with max_n (models/force_aptai.py:111). */
int aptai_ctc_greedy_decode(const float* logits, int64_t ldl, int64_t rows_per_b, int64_t B, int64_t T, int64_t V, int blank,
                            int32_t* ids_out, int64_t max_n, int32_t* n_out, void* stream);

/* ------------------------------------------------------------------------------------------------ Force_APTAI heads (fp32)
 * Small layers behind the frozen encoder of models/force_aptai.py:108-161.  fp32 because they feed an argmax whose
 * indices must match the reference. */
/* C[m][n] (+)= alpha*sum_k A(m,k)*B(k,n) + bias[n];  A(m,k)=A[m*sam+k*sak] (fp32, or bf16 when a_bf16),
 * B(k,n)=B[k*sbk+n*sbn]; `batch` problems at element strides bsa/bsb/bsc.  Replaces nn.Linear (force_aptai.py:122,
 * modules.py:140-141,195-201) and torch.bmm (modules.py:144,149). */
/* Runs on the f32-input matrix instruction (v_mfma_f32_32x32x2_f32): exact fp32 products and accumulation.  split_k > 1 cuts K
 * into slabs (partials in `workspace`, aptai_sgemm_workspace_bytes, summed in slab order: deterministic) for gradient-shaped
 * problems whose M x N alone cannot fill the chip. */
int aptai_sgemm_f32(const void* A, int a_bf16, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn, float* C,
                    int64_t ldc, const float* bias, float alpha, int accumulate, int64_t M, int64_t N, int64_t K, int64_t batch,
                    int64_t bsa, int64_t bsb, int64_t bsc, int64_t split_k, float* workspace, void* stream);
int64_t aptai_sgemm_workspace_bytes(int64_t M, int64_t N, int64_t batch, int64_t split_k);
/* nn.Embedding(padding_idx=0) + PositionalEncoding (force_aptai.py:118-119, modules.py:217-235): out[r] = emb[ids[r]] + pe[r % N] */
int aptai_embed_pe_fwd(const int32_t* ids, const float* emb, const float* pe, float* out, int64_t rows, int64_t N, int64_t D,
                       float dropout_p, uint64_t seed, void* stream);
int aptai_embed_bwd(const int32_t* ids, const float* dout, float* demb_zeroed, int64_t rows, int64_t D, float dropout_p,
                    uint64_t seed, void* stream);
/* CrossAttention softmax + the log-softmax alignment of force_aptai.py:128-130 and its argmax read-out (:148):
 * energy = raw + mask, att = softmax(energy), att_log = log_softmax(energy + mask), mask = -1000 where phn_ids == 0;
 * align int64 [B][T] = argmax_n att_log (first maximum), bit-exact integer output.  fs_rows (optional, [B*T][64]): the rows
 * ForwardSumLoss feeds its CTC (models/modules.py:90-98) - blank log-prob -1 | att_log | zeros.  bwd: d_attlog rows have
 * stride ld_dattlog (0 = N), so the CTC gradient of those 64-float rows is read in place (pointer + 1). */
int aptai_xattn_softmax_fwd(const float* raw, const int32_t* phn_ids, float* energy, float* att, float* att_log, int64_t* align,
                            float* fs_rows, int64_t B, int64_t T, int64_t N, void* stream);
int aptai_xattn_softmax_bwd(const float* att, const float* att_log, const float* d_att, const float* d_attlog, int64_t ld_dattlog,
                            float* d_raw, int64_t rows, int64_t N, void* stream);
int aptai_layernorm_f32_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, int64_t rows,
                            int64_t cols, float eps, void* stream);
int aptai_layernorm_f32_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, float* dx,
                            float* dgamma, float* dbeta, float* workspace, int64_t rows, int64_t cols, void* stream);
int64_t aptai_layernorm_f32_bwd_workspace_bytes(int64_t cols);
/* nn.LSTM(256,256,bidirectional) over packed sequences (modules.py:195,204-206): xproj [B*Tp][2][1024] = x W_ih^T + b_ih + b_hh,
 * whh [2][1024][256] (weight_hh_l0, weight_hh_l0_reverse as stored), lens int32 [B]; hout [B*Tp][512] (zeros beyond lens);
 * gates (post-activation i,f,g,o) [B*Tp][2][1024] and cstate [B*Tp][2][256] are saved for the backward (both null in
 * inference).  GATE LAYOUT of xproj, gates and dgates (round 4): column dir * 1024 + unit * 4 + gate - gate-INTERLEAVED, i.e. the rows of
 * W_ih and of the bias sum permuted from torch's gate-major order by the caller (row unit * 4 + gate of the operand = row gate * 256 + unit
 * of weight_ih_l0); whh keeps torch's order.  The *_serial cross-check kernels keep the gate-major layout.  16 cooperating workgroups per (16 utterances, direction) keep W_hh in registers and exchange h through
 * `workspace` (aptai_lstm_workspace_bytes; zero-initialised ONCE by the caller, its first 256 bytes are a status word that
 * turns non-zero if a bounded wait ever timed out).  bwd: dgates [B*Tp][2][1024] = gradients w.r.t. the gate pre-activations
 * (zeros beyond lens). */
int aptai_lstm_fwd(const float* xproj, const float* whh, const int32_t* lens, float* hout, float* gates, float* cstate,
                   void* workspace, int64_t B, int64_t Tp, int64_t T, int64_t hidden, void* stream);
int aptai_lstm_bwd(const float* dhout, const float* whh, const int32_t* lens, const float* gates, const float* cstate, float* dgates,
                   void* workspace, int64_t B, int64_t Tp, int64_t T, int64_t hidden, void* stream);
int64_t aptai_lstm_workspace_bytes(int64_t B);
/* the same recurrences, one block per (utterance, direction), W_hh streamed from L2 on every frame (whhT [2][256][1024] for the
 * forward): the on-device cross-check of the kernels above, not on the hot path */
int aptai_lstm_fwd_serial(const float* xproj, const float* whhT, const int32_t* lens, float* hout, float* gates, float* cstate,
                          int64_t B, int64_t Tp, int64_t T, int64_t hidden, void* stream);
int aptai_lstm_bwd_serial(const float* dhout, const float* whh, const int32_t* lens, const float* gates, const float* cstate,
                          float* dgates, int64_t B, int64_t Tp, int64_t T, int64_t hidden, void* stream);
/* pred_frame_phns (force_aptai.py:152-161): out[b][t] = phn_table[b][align[b][t]] for t < lens[b], else -1 (int64) */
int aptai_gather_alignment(const int32_t* phn_table, const int64_t* align, const int32_t* lens, int64_t* out, int64_t B, int64_t T,
                           int64_t N, void* stream);
int aptai_tanh_dropout_f32(const float* x, const float* y_or_null, const float* dy_or_null, float* out, int64_t n, float dropout_p,
                           uint64_t seed, void* stream);
int aptai_dropout_f32(const float* x, float* y, int64_t n, float dropout_p, uint64_t seed, void* stream);
/* out[n] = sum_r x[r*ld + n] (bias gradients of the fp32 heads), two deterministic stages through `workspace` */
int aptai_colsum_f32(const float* x, int64_t ld, float* out, float* workspace, int64_t rows, int64_t N, void* stream);
int64_t aptai_colsum_f32_workspace_bytes(int64_t N);

#ifdef __cplusplus
}
#endif
#endif /* APTAI_HIP_H */
