/* savit - C ABI of the MI355X-native (gfx950) ViT / CaiT training hot path.
 *
 * This is the drop-in boundary.  The reference
 * (NZ99/self-attention-experiments-vision) exposes a Python API only - there is no C ABI or FFI
 * in it - so every entry point below replaces an XLA-compiled op of the Flax graph and cites the
 * reference call site (paths relative to /root/reference) it stands in for.  INTEGRATION.md shows
 * the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (kernels never allocate);
 *  - `stream` is a hipStream_t passed as void*; calls are asynchronous on it and graph-capturable
 *    (no allocation, no synchronisation inside);
 *  - return value: 0 = launched, SAVIT_EINVAL (1001) = argument contract violated (nothing was
 *    launched), any other value = hipError_t of the launch;
 *  - no global mutable state: thread-safe, one process per GPU;
 *  - "bf16" = raw bfloat16 bits (uint16_t); matrices are row-major with the leading dimension in
 *    ELEMENTS; Flax kernels are [in, out].
 */
#ifndef SAVIT_H_
#define SAVIT_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAVIT_ABI_VERSION 1
int savit_abi_version(void);

/* ---- LayerNorm: flax nn.LayerNorm(dtype), eps 1e-6 (models/vit.py:19,26,57; models/cait.py:30,42,99,111,176)
 * x fp32 [rows, d] with row stride x_stride -> y bf16 [rows, d] (compact); mean/rstd fp32 [rows] (nullable).
 * round_params_bf16 != 0: scale/bias are rounded to bf16 before use, as the reference's dtype=bf16 graph does. */
int savit_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y_bf16, float* mean, float* rstd,
                        int rows, int d, long x_stride, float eps, int round_params_bf16, void* stream);

/* Backward of the above (reference: jax.value_and_grad at train.py:94-95).
 * dy bf16 [rows,d]; dres_in fp32 (nullable) is the cotangent arriving on the residual skip, added to the result;
 * dx fp32 and dx_bf16 (nullable) use row stride out_stride; dgamma/dbeta/dcolsum fp32 [d] are ACCUMULATED
 * (atomics; caller zeroes); dcolsum (nullable) = column sums of dx (the bias gradient of the Dense that fed
 * this residual add: ff.py:31). */
int savit_layernorm_bwd(const void* dy_bf16, const float* x, const float* gamma, const float* mean, const float* rstd,
                        const float* dres_in, float* dx, void* dx_bf16, float* dgamma, float* dbeta, float* dcolsum,
                        int rows, int d, long x_stride, long out_stride, int round_params_bf16, void* stream);

/* ---- Dense / DenseGeneral GEMMs with fused epilogues (bf16 MFMA, fp32 accumulate).
 * C[M,N] = epilogue( A[M,K] . Bt[N,K]^T ).  A and Bt are bf16, K contiguous ("TN"); K % 64 == 0, N % 4 == 0.
 * Replaces nn.Dense / nn.DenseGeneral at attention.py:29-37,60-63, ff.py:26-31, patch_embed.py:23-25,
 * vit.py:96-98 and, with the weight in its other layout, their input-gradient products. */
enum savit_epilogue {
  SAVIT_EPI_BF16 = 0,      /* C bf16 = acc (* alpha for columns < alpha_cols: query/sqrt(hd), attention.py:39) (+bias) */
  SAVIT_EPI_BIAS_GELU = 1, /* u = bf16(acc+bias) -> C ; gelu_tanh(u) -> C2   (ff.py:26-28) */
  SAVIT_EPI_RESID = 2,     /* C fp32 = aux_f32 + rowscale[m/rows_per_sample] * colscale[n] * bf16(acc+bias)
                              (vit.py:24,31; cait.py:36-40,47-52: LayerScale, StochasticDepth, +residual) */
  SAVIT_EPI_DGELU = 3,     /* C bf16 = acc * gelu_tanh'(aux_bf16[m,n]); colsum[n] += column sums  (backward of ff.py:27) */
  SAVIT_EPI_F32 = 4,       /* C fp32 = acc + bias (rounded through bf16 if round_out_bf16)   (vit.py:95-98) */
  SAVIT_EPI_PATCH = 5      /* A is gathered from NHWC bf16 images (patch_embed.py:19-22); C fp32 row
                              b*tokens + token_offset + p = bf16(acc) + pos[token_offset+p]  (vit.py:85, position_embed.py:56) */
};

typedef struct savit_gemm_args {
  const void* A;          /* bf16 [M, lda]   (SAVIT_EPI_PATCH: images bf16 [B, img, img, 3]) */
  const void* Bt;         /* bf16 [N, ldb] */
  void* C;                /* see epilogue */
  void* C2;               /* SAVIT_EPI_BIAS_GELU only */
  const float* bias;      /* fp32 [N] or NULL */
  const void* aux;        /* RESID: fp32 [M, ldaux]; DGELU: bf16 [M, ldaux]; PATCH: fp32 pos [tokens, N] */
  const float* colscale;  /* fp32 [N] or NULL  (LayerScale) */
  const float* rowscale;  /* fp32 [M / rows_per_sample] or NULL  (stochastic-depth mask/keep_prob per sample) */
  float* colsum;          /* fp32 [N] or NULL, accumulated with atomics */
  int M, N, K;
  int lda, ldb, ldc, ldaux;
  int epilogue;           /* enum savit_epilogue */
  float alpha;            /* scale for columns [0, alpha_cols) */
  int alpha_cols;
  int rows_per_sample;    /* tokens per image for rowscale (>=1) */
  int round_out_bf16;     /* EPI_F32: round the result through bf16 (reference logits are bf16) */
  int round_bias_bf16;    /* round bias to bf16 before adding (reference casts params to dtype) */
  /* SAVIT_EPI_PATCH geometry */
  int img_size, patch, tokens, token_offset;
  int tile;               /* 0 = auto, 1 = 128x128, 2 = 256x256, 3 = 256x128 (benchmarks / tests) */
} savit_gemm_args;

int savit_gemm_bf16_tn(const savit_gemm_args* args, void* stream);

/* Weight-gradient GEMM: dW[Kin, Nout] += X[M, Kin]^T . dY[M, Nout]  (fp32 atomics into dW; caller zeroes).
 * X, dY bf16 row-major; reduction over M is split over `splits` workgroup groups (0 = auto).
 * patch != 0: X rows are gathered from NHWC images as in SAVIT_EPI_PATCH and dY row for (b,p) is
 * b*tokens + token_offset + p of a fp32->bf16 cotangent buffer [B*tokens, lddy]. */
int savit_gemm_bf16_wgrad(const void* X, const void* dY, float* dW, int M, int Kin, int Nout, int ldx, int lddy, int lddw,
                          int splits, int patch, int img_size, int tokens, int token_offset, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SAVIT_H_ */
