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

#define SAVIT_ABI_VERSION 2 /* 2 (round 5): savit_gemm_args gained the trailing cu_budget; TN tile ids 1-5, 7-11, 14, 15, 30 left the product library */
int savit_abi_version(void);

/* ---- LayerNorm: flax nn.LayerNorm(dtype), eps 1e-6 (models/vit.py:19,26,57; models/cait.py:30,42,99,111,176)
 * x fp32 [rows, d] with row stride x_stride -> y bf16 [rows, d] (compact); mean/rstd fp32 [rows] (nullable).
 * round_params_bf16 != 0: scale/bias are rounded to bf16 before use, as the reference's dtype=bf16 graph does. */
int savit_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y_bf16, float* mean, float* rstd,
                        int rows, int d, long x_stride, float eps, int round_params_bf16, void* stream);

/* Backward of the above (reference: jax.value_and_grad at train.py:94-95).
 * dy bf16 [rows,d]; dres_in fp32 (nullable) is the cotangent arriving on the residual skip, added to the result;
 * dx fp32 and dx_bf16 (nullable) use row stride out_stride; dgamma/dbeta/dcolsum fp32 [d] (each nullable) are
 * ACCUMULATED (caller zeroes); dcolsum (nullable) = column sums of dx (the bias gradient of the Dense that fed
 * this residual add: ff.py:31). */
int savit_layernorm_bwd(const void* dy_bf16, const float* x, const float* gamma, const float* mean, const float* rstd,
                        const float* dres_in, float* dx, void* dx_bf16, float* dgamma, float* dbeta, float* dcolsum,
                        int rows, int d, long x_stride, long out_stride, int round_params_bf16, void* workspace,
                        long workspace_bytes, void* stream);
/* The same (wide rows: d > 64), and its finalize launch ALSO adds the column sums of an extra slab to extra_out[extra_n]
 * (extra_out[c] += sum_r extra_slab[r][c], r < extra_rows, one adder per column, fixed order): the bias-gradient partials the GELU'
 * GEMM wrote (savit_gemm_colsum_rows) ride along instead of their own savit_colsum_finalize launch. */
int savit_layernorm_bwd_ex(const void* dy_bf16, const float* x, const float* gamma, const float* mean, const float* rstd,
                           const float* dres_in, float* dx, void* dx_bf16, float* dgamma, float* dbeta, float* dcolsum,
                           int rows, int d, long x_stride, long out_stride, int round_params_bf16, void* workspace,
                           long workspace_bytes, const float* extra_slab, int extra_rows, int extra_n, float* extra_out, void* stream);
/* The same (wide rows) for a row subset with a SPARSE residual gradient (round 5: the backward of a ViT's LAST encoder layer - only the
 * cls rows reach the head (vit.py:57,95), so behind the final LayerNorm the residual gradient is exactly zero on every other row and
 * the layer's MLP branch is differentiated on the B cls rows alone):
 *   stat_stride: mean / rstd of row r are read at index r * stat_stride (the cls rows inside the per-token statistics: stride N);
 *   res_mod = 0: dres_in as in savit_layernorm_bwd.  res_mod > 0: only rows r with r % res_mod == 0 have a residual gradient, stored
 *   compactly at dres_in + (r / res_mod) * res_stride - the first DENSE LayerNorm backward above the cls-only part merges the compact
 *   [B, d] residual gradient of the cls rows this way, so no [B*N, d] buffer is zero-filled to carry B rows.
 * Every output pointer NULL: deferred column sums, as in savit_layernorm_bwd.  extra_*: as in savit_layernorm_bwd_ex (nullable). */
int savit_layernorm_bwd_sparse(const void* dy_bf16, const float* x, const float* gamma, const float* mean, const float* rstd, int stat_stride,
                               const float* dres_in, int res_mod, long res_stride, float* dx, void* dx_bf16, float* dgamma, float* dbeta,
                               float* dcolsum, int rows, int d, long x_stride, long out_stride, int round_params_bf16, void* workspace,
                               long workspace_bytes, const float* extra_slab, int extra_rows, int extra_n, float* extra_out, void* stream);
/* Scratch the call above needs (per-block partial column sums; 16-B aligned, contents undefined afterwards). */
long savit_layernorm_bwd_workspace_bytes(int rows, int d);

/* Deferred column sums (round 5).  Nothing inside a backward pass reads dgamma / dbeta / the bias gradients (they are the optimizer's
 * inputs: /root/reference/train.py:94-100), so a caller may pass NULL for EVERY output pointer of savit_layernorm_bwd,
 * savit_layernorm_bwd_mapped (wide rows, d > 64), savit_layerscale_bwd or savit_layernorm_bwd_ls: the call then only leaves its partial
 * slab [nblk = savit_layernorm_bwd_grid(rows)][nf][d] in `workspace` (one workspace per deferred call), and ONE
 * savit_layernorm_bwd_finalize_jobs launch reduces the slabs of many calls where the gradients must be final (a data-parallel bucket
 * trigger, the end of backward).  Same arithmetic as the per-call finalize: out[k][c] += sum over blocks of partial[.][k][c].
 *   nf = 3: out = {dgamma, dbeta, dcolsum} (savit_layernorm_bwd*), {d_layerscale, dbias, NULL} (savit_layerscale_bwd);
 *   nf = 4: out = {dgamma, dbeta, d_layerscale, dbias} (savit_layernorm_bwd_ls).  extra_*: as in savit_layernorm_bwd_ex. */
typedef struct savit_colsum_job {
  const float* partial;
  int nblk, d, nf;
  float* out[4]; /* each nullable */
  const float* extra_slab;
  int extra_rows, extra_n;
  float* extra_out;
} savit_colsum_job;
int savit_layernorm_bwd_grid(int rows);
int savit_layernorm_bwd_finalize_jobs(const savit_colsum_job* jobs, int count, void* stream);

/* Row-mapped variants for the LayerNorm over concat([cls, x]) of CaiT's token-only layers (cait.py:98-99): logical row r of x is
 * written to (read from, for dy) row (r / grp) * grp_stride + grp_off + r % grp of the bf16 matrix, so the cls rows and the patch rows
 * are normalised by two calls into ONE [B*(N+1), d] operand without materialising the fp32 concatenation. */
int savit_layernorm_fwd_mapped(const float* x, const float* gamma, const float* beta, void* y_bf16, float* mean, float* rstd, int rows, int d,
                               long x_stride, float eps, int round_params_bf16, int grp, int grp_stride, int grp_off, void* stream);
int savit_layernorm_bwd_mapped(const void* dy_bf16, const float* x, const float* gamma, const float* mean, const float* rstd,
                               const float* dres_in, float* dx, void* dx_bf16, float* dgamma, float* dbeta, float* dcolsum, int rows, int d,
                               long x_stride, long out_stride, int round_params_bf16, int dy_grp, int dy_grp_stride, int dy_grp_off,
                               void* workspace, long workspace_bytes, void* stream);

/* ---- LayerScale + StochasticDepth backward (layerscale.py:18-23, stochastic_depth.py:16-27): for out = res + rs*ls*branch
 *   dbranch = bf16(dres * rs * ls) ; d_layerscale += sum_rows dres*rs*branch ; dbias (nullable) += column sums of dbranch.
 * branch is the bf16 tensor SAVIT_EPI_RESID stores through C2.  workspace: savit_layernorm_bwd_workspace_bytes(rows, d). */
int savit_layerscale_bwd(const float* dres, const void* branch_bf16, const float* layerscale, const float* rowscale, int rows_per_sample,
                         void* dbranch_bf16, float* d_layerscale, float* dbias, int rows, int d, long dres_stride, void* workspace,
                         long workspace_bytes, void* stream);
/* savit_layernorm_bwd (wide rows, d > 64; no bf16 copy, no column sums of dx) FOLLOWED BY savit_layerscale_bwd on the dx it produces, in
 * one pass over the residual gradient: dx = LN-VJP(dy) + dres_in; dbranch = bf16(dx * rs * ls); d_layerscale, dbias as above.  In the
 * reverse of CaiT's EncoderBlock (cait.py:28-60) every LayerNorm backward is followed by the LayerScale backward of the sub-block
 * before it; element by element the results are those of the two calls.  extra_slab (nullable): as in savit_layernorm_bwd_ex. */
int savit_layernorm_bwd_ls(const void* dy_bf16, const float* x, const float* gamma, const float* mean, const float* rstd,
                           const float* dres_in, float* dx, float* dgamma, float* dbeta, int rows, int d, long x_stride, long out_stride,
                           int round_params_bf16, const void* branch_bf16, const float* layerscale, const float* rowscale,
                           int rows_per_sample, void* dbranch_bf16, float* d_layerscale, float* dbias, void* workspace,
                           long workspace_bytes, const float* extra_slab, int extra_rows, int extra_n, float* extra_out, void* stream);

/* ---- Class attention (cait.py:10-15): one query per image (cls) against Nk keys.  q bf16 [B, ldq] pre-scaled; kv bf16 [B*Nk, ldkv]
 * with keys at column h*hd and values at column d + h*hd; o bf16 [B, d]; probs fp32 [B,H,Nk] (saved for backward). */
int savit_class_attention_fwd(const void* q, long ldq, const void* kv, int ldkv, void* o, float* probs, int B, int Nk, int H, int head_dim,
                              void* stream);
int savit_class_attention_bwd(const void* q, long ldq, const void* kv, int ldkv, const float* probs, const void* d_o, void* dq, long lddq,
                              void* dkv, int B, int Nk, int H, int head_dim, float dq_scale, void* stream);
/* The cls query of a ViT's LAST encoder layer against all Nk <= 640 keys (round 5: only row 0 of every image reaches the head, vit.py:57,95,
 * so of that layer's attention only the cls query's output is used and only its cotangent row is non-zero).  Same tensor conventions as
 * savit_class_attention_fwd / _bwd, but the rounding points of savit_attention_fwd / _bwd (fp32 scores, bf16 P operand with the un-rounded row
 * sum, LSE [B, H] saved instead of the probabilities; backward recomputes P, dS passes through bf16) - a model differentiated this way
 * agrees with the dense attention kernels to fp32 summation order.  o, d_o: bf16 [B, d]; dq row b at dq + b * lddq; dkv rows as kv. */
int savit_cls_query_attention_fwd(const void* q, long ldq, const void* kv, int ldkv, void* o, float* lse, int B, int Nk, int H, int head_dim,
                                  void* stream);
int savit_cls_query_attention_bwd(const void* q, long ldq, const void* kv, int ldkv, const void* o, const float* lse, const void* d_o, void* dq,
                                  long lddq, void* dkv, int B, int Nk, int H, int head_dim, float dq_scale, void* stream);

/* ---- Dense / DenseGeneral GEMMs with fused epilogues (bf16 MFMA, fp32 accumulate).
 * C[M,N] = epilogue( A[M,K] . Bt[N,K]^T ).  A and Bt are bf16, K contiguous ("TN"); K % 64 == 0, N % 4 == 0.
 * Replaces nn.Dense / nn.DenseGeneral at attention.py:29-37,60-63, ff.py:26-31, patch_embed.py:23-25,
 * vit.py:96-98 and, with the weight in its other layout, their input-gradient products. */
enum savit_epilogue {
  SAVIT_EPI_BF16 = 0,      /* C bf16 = acc (* alpha for columns < alpha_cols: query/sqrt(hd), attention.py:39) (+bias) */
  SAVIT_EPI_BIAS_GELU = 1, /* u = bf16(acc+bias) -> C ; gelu_tanh(u) -> C2   (ff.py:26-28) */
  SAVIT_EPI_RESID = 2,     /* C fp32 = aux_f32 + rowscale[m/rows_per_sample] * colscale[n] * bf16(acc+bias); C2 (nullable) = that bf16 branch
                              (vit.py:24,31; cait.py:36-40,47-52: LayerScale, StochasticDepth, +residual).  round_out_bf16: the sum
                              is rounded through bf16 (mlp_mixer.py:24,30: a residual stream that stays in the module dtype) */
  SAVIT_EPI_DGELU = 3,     /* C bf16 = acc * gelu_tanh'(aux_bf16[m,n]); colsum[n] += column sums  (backward of ff.py:27) */
  SAVIT_EPI_F32 = 4,       /* C fp32 = acc + bias (rounded through bf16 if round_out_bf16)   (vit.py:95-98) */
  SAVIT_EPI_PATCH = 5      /* A is gathered from NHWC bf16 images (patch_embed.py:19-22); C fp32 row
                              b*tokens + token_offset + p = bf16(acc (+bias)) + pos[token_offset+p]  (vit.py:85, position_embed.py:56);
                              aux (pos) may be NULL (mlp_mixer.py:46-49: use_bias=True, no position embedding) */
};

typedef struct savit_gemm_args {
  const void* A;          /* bf16 [M, lda]   (SAVIT_EPI_PATCH: images bf16 [B, img, img, 3]).  lda < K is allowed for operands whose
                             true width w <= lda is not a multiple of the 32-deep K-step: columns [lda, K) of a row then alias the next
                             row (zeros past the end), and Bt must hold ZEROS in columns [w, K) */
  const void* Bt;         /* bf16 [N, ldb] */
  void* C;                /* see epilogue */
  void* C2;               /* SAVIT_EPI_BIAS_GELU: gelu output; SAVIT_EPI_RESID: optional bf16 branch (for LayerScale backward) */
  const float* bias;      /* fp32 [N] or NULL */
  const void* aux;        /* RESID: fp32 [M, ldaux]; DGELU: bf16 [M, ldaux]; PATCH: fp32 pos [tokens, N] */
  const float* colscale;  /* fp32 [N] or NULL  (LayerScale) */
  const float* rowscale;  /* fp32 [M / rows_per_sample] or NULL  (stochastic-depth mask/keep_prob per sample) */
  float* colsum;          /* fp32 or NULL: [N] accumulated with atomics (colsum_rows == 0), or a [colsum_rows, N] slab of per-
                             row-tile partial sums written with plain stores (deterministic; reduce with savit_colsum_finalize) */
  int M, N, K;
  int lda, ldb, ldc, ldaux;
  int epilogue;           /* enum savit_epilogue */
  float alpha;            /* scale for columns [0, alpha_cols) */
  int alpha_cols;
  int rows_per_sample;    /* tokens per image for rowscale (>=1) */
  int round_out_bf16;     /* EPI_F32 / EPI_RESID: round the result through bf16 (reference logits are bf16) */
  int round_bias_bf16;    /* round bias to bf16 before adding (reference casts params to dtype) */
  /* SAVIT_EPI_PATCH geometry */
  int img_size, patch, tokens, token_offset;
  int tile;               /* 0 = auto; explicit ids select a kernel variant (benchmarks / tests, see gemm_tn.hip) */
  int colsum_rows;        /* 0, or the slab height savit_gemm_colsum_rows() gives for this shape and tile */
  int cu_budget;          /* 0 = the whole device; else the CUs the auto heuristic (tile == 0) may count on: a data-parallel rank leaves
                             some to the resident RCCL all-reduce (train.py:96), and a grid sized for ALL CUs would then run a second,
                             nearly empty round */
} savit_gemm_args;

int savit_gemm_bf16_tn(const savit_gemm_args* args, void* stream);
/* Bias-gradient column sums without atomics: ~200 row tiles adding into the same N addresses serialise at the memory side
 * (16 us of a 180 us launch on the fc2 input-gradient GEMM).  savit_gemm_colsum_rows: partial-sum rows the GEMM writes for
 * (M, N, K, tile) - one per (row tile, wave row); savit_colsum_finalize: out[n] (+)= sum_r slab[r][n] (fixed summation order
 * below 512 rows; taller slabs in accumulate mode are reduced in row chunks that add with fp32 atomics). */
int savit_gemm_colsum_rows(int M, int N, int K, int tile);
int savit_gemm_colsum_rows_cus(int M, int N, int K, int tile, int cu_budget); /* tile == 0: the auto choice for cu_budget CUs */
int savit_colsum_finalize(const float* slab, int rows, int N, float* out, int accumulate, void* stream);
/* Tile the auto heuristic (tile == 0) picks for a shape and epilogue.  K % 64 == 0: paired-stage kernels 17 = 192x128 (4 waves, two
 * workgroups per CU; the default), 13 = 256x256 (8 waves; GELU-forward on large grids), 12 = 128x128 (small / ragged problems);
 * otherwise 6 = 128x128 on the 32-deep ring; 18 = 17 with the last partial round in 128-row tiles; 20 / 21 / 22 = the 256x256 / 320x256 /
 * persistent 320x256 ping-pong kernels (wide outputs, K >= 768; round 2 / 3 / 5); 24 (round 5) = M <= 256 rows, every epilogue but GELU' and
 * the patch gather: no LDS, one wave per 16x16 or 32x32 output tile over all of K (the cls rows of a ViT's last layer and head, CaiT's
 * class-attention layers).  Every tile gives bit-identical results (one accumulator per output element walks K in ascending 32-steps).
 * savit_gemm_tn_auto_tile = the SAVIT_EPI_BF16 choice. */
int savit_gemm_tn_auto_tile(int M, int N, int K);
int savit_gemm_tn_auto_tile_epi(int M, int N, int K, int epilogue);
int savit_gemm_tn_auto_tile_cus(int M, int N, int K, int epilogue, int cu_budget); /* the same for cu_budget CUs (0 = all) */

/* Weight-gradient GEMM: dW[Kin, Nout] += X[M, Kin]^T . dY[M, Nout]  (fp32 atomics into dW; caller zeroes).
 * X, dY bf16 row-major; reduction over M is split over `splits` workgroup groups (0 = auto).
 * patch != 0: X rows are gathered from NHWC images as in SAVIT_EPI_PATCH and dY row for (b,p) is
 * b*tokens + token_offset + p of the bf16 cotangent buffer [B*tokens, lddy]. */
int savit_gemm_bf16_wgrad(const void* X, const void* dY, float* dW, int M, int Kin, int Nout, int ldx, int lddy, int lddw,
                          int splits, int patch, int img_size, int tokens, int token_offset, void* stream);
/* The same product without atomics: every split of the reduction stores its partial [Kin, Nout] tile set to `workspace` with plain
 * stores and a second launch adds the splits to dW in index order - bitwise reproducible, and the partial traffic moves at the
 * plain-store rate instead of the chip-wide float-atomic rate.  workspace: 16-B aligned, at least
 * savit_gemm_wgrad_workspace_bytes(M, Kin, Nout, splits, patch) bytes (the split count the launch will use x Kin x Nout x 4; 0 for the
 * shapes served by the small 2-stage kernel, which keeps atomics); NULL or too small a workspace falls back to the atomic form. */
int savit_gemm_bf16_wgrad_ws(const void* X, const void* dY, float* dW, int M, int Kin, int Nout, int ldx, int lddy, int lddw,
                             int splits, int patch, int img_size, int tokens, int token_offset, void* workspace, long workspace_bytes,
                             void* stream);
long savit_gemm_wgrad_workspace_bytes(int M, int Kin, int Nout, int splits, int patch);
/* The two launches of savit_gemm_bf16_wgrad_ws as separate calls (so a profiler or a launch plan sees one kernel per entry):
 * _partial stores the split partials to the workspace (SAVIT_EINVAL when the workspace is missing / too small or the shape is served by
 * the atomic 2-stage kernel), _reduce adds the first `splits` slabs to dW in index order; savit_gemm_wgrad_split_count = the split
 * count the launch uses for (M, Kin, Nout, splits hint, patch) - 0 for shapes without a slab form. */
int savit_gemm_bf16_wgrad_partial(const void* X, const void* dY, int M, int Kin, int Nout, int ldx, int lddy, int splits, int patch,
                                  int img_size, int tokens, int token_offset, void* workspace, long workspace_bytes, void* stream);
int savit_gemm_wgrad_reduce(const void* workspace, int splits, int Kin, int Nout, float* dW, int lddw, void* stream);
int savit_gemm_wgrad_split_count(int M, int Kin, int Nout, int splits, int patch);
/* Kernel the wgrad heuristic picks: 1 = 128x128 ring (4 waves), 3 = 256x256 ring (8 waves). */
int savit_gemm_wgrad_auto_variant(int Kin, int Nout, int patch);

/* Grouped weight gradients: dW_i += X_i^T dY_i for up to 64 independent Dense kernels in ONE launch - the reverse-mode products of
 * attention.py:29-37,60-63 and ff.py:26-31 for one or more encoder layers (train.py:94-95).  One workgroup per `tile` x `tile`
 * (256 or 128) output tile reduces over ALL M tokens: no token split, no partial slabs, no second launch; every dW element is one
 * fixed-order sum added to its old value (bitwise reproducible).  A weight gradient has no consumer before the optimizer step, so a
 * caller may group across layers until the tiles fill the CUs (savit_gemm_wgrad_group_tiles = tiles of one kernel).
 * Kin / Nout need not be multiples of the tile: edge tiles compute the full tile and store the part inside dW.
 * X_i bf16 [M_i, ldx], dY_i bf16 [M_i, lddy], dW_i fp32 [Kin_i, lddw]; Kin, Nout, ldx, lddy multiples of 8; M_i * pitch * 2 < 4 GiB. */
typedef struct savit_wgrad_problem {
  const void* X;
  const void* dY;
  float* dW;
  int M, Kin, Nout, ldx, lddy, lddw;
  int tile_begin, tile_count; /* a range of the weight's output tiles (row-major over tile x tile blocks of dW); 0, 0 = all of them.
                                 Lets a caller cut launches at exact multiples of the CU count: a weight may span two launches. */
  int overwrite;              /* 0: dW += X^T dY (the contract of every weight-gradient entry point).  1: dW = X^T dY - each output tile is
                                 written by exactly one workgroup, so when this is the only contribution of the step (no gradient
                                 accumulation) the caller need not zero dW beforehand and the kernel does not read it (round 5: removes
                                 the 346 MB memset of DeiT-B's gradient buffer and 4 B per element of the read-modify-write). */
} savit_wgrad_problem;
int savit_gemm_bf16_wgrad_grouped(const savit_wgrad_problem* problems, int count, int tile, void* stream);
/* The same; sumsq32 (nullable): 32 fp32 accumulators (caller zeroes) that receive the sum of squares of every element the launch stored,
 * spread over the 32 by workgroup - the weight gradients' share of optax.clip_by_global_norm's norm (train.py:25) without a pass over
 * them; savit_sumsq_ranges adds the accumulators to the rest.  Only meaningful when what is stored is final (overwrite entries). */
int savit_gemm_bf16_wgrad_grouped_ex(const savit_wgrad_problem* problems, int count, int tile, float* sumsq32, void* stream);
int savit_gemm_wgrad_group_tiles(int Kin, int Nout, int tile);


/* ---- Fused multi-head self-attention (attention.py:39-58): per (batch, head)
 *   S = (q/sqrt(hd)) k^T ; P = softmax_k(S) ; O = P v.      head_dim 16, 32, 48 or 64; any N <= 65 536
 *   (N <= 608: one head's K / V images resident in LDS; longer: K / V - in backward also Q / dO - stream through LDS in 256-row segments).
 * qkv bf16 [B*N, ld_qkv]: columns [0,d) = queries ALREADY scaled by 1/sqrt(hd) (SAVIT_EPI_BF16 alpha), [d,2d) keys,
 * [2d,3d) values, head-major (h*64+e) inside each (the DenseGeneral (H,hd) feature order, attention.py:29-33).
 * o bf16 [B*N, d]; lse fp32 [B,H,N] = log-sum-exp of each score row (saved for backward; nullable). */
int savit_attention_fwd(const void* qkv, void* o, float* lse, int B, int N, int H, int head_dim, int ld_qkv, void* stream);
/* Backward: dqkv bf16 [B*N, ld_qkv] receives dQ*dq_scale | dK | dV (dq_scale = 1/sqrt(hd) undoes the folded scale
 * so that the QKV input-gradient GEMM is a plain product).  P is recomputed from lse; no atomics. */
int savit_attention_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv, int B, int N, int H,
                        int head_dim, int ld_qkv, float dq_scale, void* stream);

/* Very short sequences (TNT's inner transformer, tnt.py:68-76: 16 pixel tokens, 4 heads padded to 16 columns): one wave per
 * sequence, scores / softmax / outputs in registers.  Same tensor layout and scaling conventions as savit_attention_fwd/bwd with
 * B = nseq, N = 16, H = 4, head_dim = 16, ld_qkv = 192 - the only geometry accepted (SAVIT_EINVAL otherwise; the tiled kernels
 * cover the rest).  Backward recomputes P: no lse. */
int savit_seq16_attention_fwd(const void* qkv, void* o, long nseq, int tokens, int heads, int head_dim_padded, int ld_qkv, void* stream);
int savit_seq16_attention_bwd(const void* qkv, const void* d_o, void* dqkv, long nseq, int tokens, int heads, int head_dim_padded, int ld_qkv,
                              float dq_scale, void* stream);

/* ---- Talking-heads attention (CaiT SA layers: attention.py:41-58 with talking_heads=True, talking_heads.py:9-14)
 *   S_h = q_h k_h^T ; S'_i = sum_h T1[h,i] S_h ; P_i = softmax_k(S'_i) ; P'_i = sum_h T2[h,i] P_h ; O_i = P'_i v_i
 * T1/T2 fp32 [H,H] ('h i, b h ... -> b i ...').  H in {2,4,6,8,16}, head_dim 48 or 64, N <= 256, Np = row pitch of the
 * score buffers (multiple of 8, >= N).  s_buf / p_buf: bf16 [B,H,N,Np] outputs of forward that backward consumes (S is kept,
 * p_buf is overwritten with dS); ds_buf: bf16 scratch of the same size; dT1/dT2 fp32 [H,H] are accumulated.
 * dqkv receives dQ*dq_scale | dK | dV like savit_attention_bwd. */
int savit_th_attention_fwd(const void* qkv, const float* T1, const float* T2, void* s_buf, void* p_buf, void* o, int B, int N, int H,
                           int head_dim, int ld_qkv, int Np, void* stream);
int savit_th_attention_bwd(const void* qkv, const float* T1, const float* T2, const void* s_buf, void* p_buf, const void* d_o, void* ds_buf,
                           void* dqkv, float* dT1, float* dT2, int B, int N, int H, int head_dim, int ld_qkv, int Np, float dq_scale,
                           void* workspace, long workspace_bytes, void* stream);
long savit_th_attention_bwd_workspace_bytes(int B, int N, int H);

/* ---- Fused talking-heads attention (csrc/th_fused.hip): the same function with S and P' kept in LDS.  Forward saves nothing for
 * backward; backward recomputes S / P from QKV, materialises only dS and P' (bf16 [B,H,N,Np] scratch shared by all layers) and
 * finishes with savit_th_attention_bwd_products (dV = P'^T dO, dQ = dS K * dq_scale, dK = dS^T Q).  Covered: H in {2,4,6,8},
 * head_dim 48 / 64, N <= 208 (savit_th_fused_supported); other geometries use the entries above.  savit_th_fused_preferred says
 * which path a caller that has both should take (measured: the materialising kernels are faster on MI355X, so it returns 0; the
 * library itself reads no environment variable - the Python engine takes CaiTEngine(th_fused=True) or SAVIT_TH_FUSED=1 - and the
 * fused kernels keep no per-layer S / P').
 * Workspaces: forward = V^T scratch, backward = dT1/dT2 partial slab (sizes from the *_workspace_bytes functions). */
int savit_th_fused_supported(int N, int H, int head_dim);
int savit_th_fused_preferred(int N, int H, int head_dim);
long savit_th_fused_fwd_workspace_bytes(int B, int N, int H, int head_dim);
long savit_th_fused_bwd_workspace_bytes(int B, int N, int H, int head_dim);
int savit_th_fused_attention_fwd(const void* qkv, const float* T1, const float* T2, void* o, int B, int N, int H, int head_dim, int ld_qkv,
                                 void* workspace, long workspace_bytes, void* stream);
int savit_th_fused_attention_bwd(const void* qkv, const float* T1, const float* T2, const void* d_o, void* p_buf, void* ds_buf, void* dqkv,
                                 float* dT1, float* dT2, int B, int N, int H, int head_dim, int ld_qkv, int Np, float dq_scale,
                                 void* workspace, long workspace_bytes, void* stream);
int savit_th_attention_bwd_products(const void* qkv, const void* p_buf, const void* ds_buf, const void* d_o, void* dqkv, int B, int N, int H,
                                    int head_dim, int ld_qkv, int Np, float dq_scale, void* stream);

/* ---- token assembly (vit.py:81-85, position_embed.py:52-57): x0[b,0,:] = cls + pos[0,:] (patch rows are written by
 * SAVIT_EPI_PATCH); and the backward of both adds: dpos[t,:] += sum_b dx0[b,t,:], dcls += sum_b dx0[b,0,:]. */
int savit_cls_pos_rows(const float* cls, const float* pos, float* x0, int B, long row_stride, int d, void* stream);
int savit_pos_cls_grad(const float* dx0, float* dpos, float* dcls, int B, int N, int d, int has_cls, void* stream);

/* ---- MLP-Mixer glue (mlp_mixer.py:17-31,61-63).
 * savit_transpose_bf16: B matrices src[b] = bf16 [R, Cc] (row pitch ld_src, batch stride src_batch_stride elements) are transposed:
 *   dst_bf16[b][c][r] = src[b][r][c]   (nullable; row pitch ld_dst >= R, batch stride dst_batch_stride; columns >= R are not written)
 *   out_f32[b][c][r]  = resid[b][c][r] + src[b][r][c], rounded through bf16 if round_out_bf16   (nullable pair; fp32 [B, Cc, ld_dst])
 *   rowsum_slab[k][r] = partial sums over column tiles: sum_k rowsum_slab[k][r] = sum over b, c of src[b][r][c]   (nullable; fp32
 *                       [savit_transpose_rowsum_rows(B, Cc), rowsum_ld >= R], plain stores; reduce with savit_colsum_finalize)
 * i.e. rearrange '... l d -> ... d l' (:19), the way back fused with `x = x + inputs` (:23-24), and the bias gradient of
 * the second token Dense.  Pitches and strides are multiples of 8 elements, pointers 16-B aligned.
 * savit_token_mean_fwd: z[b, :] = bf16(mean over the L tokens of h[b, :, :])  (jnp.mean(x, axis=1), :62); _bwd: dh[b, l, :] = bf16(dz[b, :] / L). */
int savit_transpose_bf16(const void* src, long src_batch_stride, int ld_src, void* dst_bf16, long dst_batch_stride, int ld_dst, int B, int R,
                         int Cc, const float* resid, float* out_f32, int round_out_bf16, float* rowsum_slab, int rowsum_ld, void* stream);
/* Several plain transposes (no residual, no row sums) in ONE launch - at most 8 jobs; job j = `batch` matrices bf16 [rows, cols] with the
 * pitches and strides of savit_transpose_bf16 (round 5: the [out, in] operand refresh of a ViT after the optimizer step, five launches -> one). */
typedef struct savit_transpose_job {
  const void* src;
  void* dst;
  long src_batch_stride, dst_batch_stride;
  int ld_src, ld_dst, batch, rows, cols;
} savit_transpose_job;
int savit_transpose_bf16_jobs(const savit_transpose_job* jobs, int count, void* stream);
int savit_transpose_rowsum_rows(int B, int Cc);
int savit_token_mean_fwd(const void* h_bf16, void* z_bf16, int B, int L, int d, void* stream);
int savit_token_mean_bwd(const void* dz_bf16, void* dh_bf16, int B, int L, int d, void* stream);

/* ---- TNT glue (tnt.py:17-33,40-51,166-193).
 * savit_tnt_pixel_gather: PixelEmbedBlock's rearranges - out bf16 [B*(S/P)^2*(P/t)^2, ld_out], row = (image, patch, pixel token),
 *   columns c*t*t + t1*t + t2 (channel slowest) of images bf16 NHWC [B, S, S, C]; columns >= C*t*t are left untouched.
 * savit_add_rows_periodic: x[r, :] += pos[r mod period, :] (AddAbsPosEmbed on the pixel stream).
 * savit_tnt_inner2outer_add: out[b, 0] = patch[b, 0]; out[b, 1+p] = patch[b, 1+p] + y[b*(N-1)+p]  (jnp.pad + add, tnt.py:49-50).
 * savit_tnt_inner2outer_split: its backward - dres += dt on every row; dy[b*(N-1)+p] = bf16(dt[b, 1+p]); dbias (nullable) +=
 *   column sums of dy (the bias gradient of the Inner2Outer Dense).  d <= 1024.
 * savit_cast_colsum: dst_bf16 (nullable) = bf16(src) for fp32 [rows, d]; colsum (nullable) += column sums of src.  d <= 1024.
 *   (A residual-stream cotangent that was completed by something other than a LayerNorm backward - which would have produced both.)
 * savit_gather_rows_bf16 / savit_scatter_rows: the head reads row 0 of every image without a LayerNorm (tnt.py:187):
 *   dst[b] = bf16(src[b*row_stride ...]);  backward writes the bf16 cotangent rows back as fp32 (and optionally bf16) rows. */
int savit_tnt_pixel_gather(const void* images_bf16, void* out_bf16, int B, int img_size, int patch, int t, int C, int ld_out, void* stream);
int savit_add_rows_periodic(float* x, const float* pos, long rows, int period, int d, void* stream);
int savit_tnt_inner2outer_add(const float* patch, const void* y_bf16, float* out, int B, int N, int d, void* stream);
int savit_tnt_inner2outer_split(const float* dt, float* dres, void* dy_bf16, float* dbias, int B, int N, int d, void* stream);
int savit_cast_colsum(const float* src, void* dst_bf16, float* colsum, long rows, int d, void* stream);
int savit_gather_rows_bf16(const float* src, long row_stride, void* dst_bf16, int B, int d, void* stream);
int savit_scatter_rows(const void* src_bf16, float* dst, void* dst_bf16, long row_stride, int B, int d, void* stream);

/* ---- loss (train.py:83-90): one_hot -> optional mix (ratio*y + (1-ratio)*y1) -> optax.smooth_labels ->
 * optax.softmax_cross_entropy -> mean.  logits fp32 [B, ld]; labels int32 [B].  Outputs (each nullable):
 * loss_rows [B]; loss_mean[0] += mean; dlogits bf16 [B, ld_dlogits] = (softmax - y)*grad_scale, columns >= C zeroed;
 * dbias[C] += column sums of dlogits; top1/top5 [B] = 1.0 when the label is among the k largest logits (utils.py:20-31). */
int savit_softmax_xent(const float* logits, int ld_logits, const int* labels, const int* mix_labels, const float* ratio,
                       float label_smoothing, float grad_scale, float* loss_rows, float* loss_mean, void* dlogits_bf16,
                       int ld_dlogits, float* dbias, float* top1, float* top5, int B, int C, void* stream);

/* ---- optimizer (train.py:25-27,100; simple_train.py:25-27): out[0] += sum(g^2); then one fused pass
 *   g' = g*grad_scale*clip ; m,v Adam moments ; p -= lr * ( mhat/(sqrt(vhat)+eps) + weight_decay*p )
 * clip = 1 if ||g*grad_scale|| < max_norm else max_norm/||.|| (optax.clip_by_global_norm), read from grad_sumsq ON DEVICE
 * (nullable / max_norm <= 0: no clipping).  step is 1-based.  n % 4 == 0. */
int savit_sumsq(const float* g, long n, float* out, void* stream);
/* The same over `count` ranges of one buffer, ranges = (offset, length) pairs in floats (both multiples of 4, base 16-byte aligned);
 * slots (nullable): `nslots` <= 256 further partial sums added in (the accumulators of savit_gemm_bf16_wgrad_grouped_ex).  And the
 * matching clear of ranges.  With the grouped weight gradients storing by first touch, these two replace the memset of the whole
 * gradient buffer and the sum of squares over it by passes over the ranges NOTHING overwrites (biases, LayerNorm parameters, embeddings). */
int savit_sumsq_ranges(const float* base, const long* ranges, int count, const float* slots, int nslots, float* out, void* stream);
int savit_zero_ranges(float* base, const long* ranges, int count, void* stream);
int savit_adamw_step(float* params, const float* grads, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                     float weight_decay, int step, const float* grad_sumsq, float max_norm, float grad_scale, void* stream);
/* The same update; params_bf16 (nullable) additionally receives the updated parameters rounded to bf16 in the same flat layout: the
 * [in, out] bf16 operands of the input-gradient GEMMs are views into it (no cast pass over the fp32 parameters). */
int savit_adamw_step_mirror(float* params, const float* grads, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                            float weight_decay, int step, const float* grad_sumsq, float max_norm, float grad_scale, void* params_bf16,
                            void* stream);

/* ---- operand preparation: fp32 master weights [batch][R][C] -> bf16 [batch][R][C] (dst_n) and/or transposed
 * [batch][C][R] (dst_t); fp32 -> bf16 elementwise; input batches [H,W,C,N] fp32 -> [N,H,W,C] bf16 (train.py:80-81). */
int savit_cast_transpose_bf16(const float* src, long src_batch_stride, int batch, int R, int C, void* dst_n,
                              long dst_n_batch_stride, int ld_n, void* dst_t, long dst_t_batch_stride, int ld_t, void* stream);
int savit_cast_bf16(const float* src, void* dst, long n, void* stream);
int savit_hwcn_to_nhwc_bf16(const float* src, void* dst, int H, int W, int C, int N, void* stream);

/* ---- GPU-side input path: what the reference's TF host pipeline does around train.py:80-81.
 * normalise: dst[n,h,w,c] = bf16( (src*scale - mean[c]) / std[c] )   (data/preprocess/preprocess.py:176-179; mean/std are HOST
 *   arrays of C floats, e.g. data/constants.py:7-8; scale = 1/255 for u8 pixels, 1 for [0,1] floats); src_format picks the
 *   source layout: the loader's [H,W,C,N] fp32 (train.py:80) or [N,H,W,C] fp32 / u8.  C <= 4.
 * mixup:  out[b] = x[b]*weight[b] + x[index[b]]*(1-weight[b])                    (augment_ops.py:144-181)
 * cutmix: out[b,y,x,:] = (y0<=y<y1 && x0<=x<x1) ? x[b,y,x,:] : x[index[b],y,x,:]   (augment_ops.py:98-141; box[b] = y0,y1,x0,x1;
 *   the reference pairs b with B-1-b).  x / out: [B,H,W,C] bf16, out != x; weight / index / box: DEVICE arrays (the random
 *   draws are the caller's: TF's stateless RNG stream is not reproducible).  elems_per_image % 8 == 0. */
enum savit_src_format { SAVIT_SRC_HWCN_F32 = 0, SAVIT_SRC_NHWC_F32 = 1, SAVIT_SRC_NHWC_U8 = 2 };
int savit_normalize_to_nhwc_bf16(const void* src, int src_format, void* dst, int H, int W, int C, int N, float scale,
                                 const float* mean, const float* std, void* stream);
int savit_batch_mixup_bf16(const void* x, void* out, const float* weight, const int* index, int B, long elems_per_image, void* stream);
int savit_batch_cutmix_bf16(const void* x, void* out, const int* box, const int* index, int B, int H, int W, int C, void* stream);

/* ---- fp32 arithmetic mode, forward + loss (create_model's default dtype=float32: models/create_model.py:6-8; BASELINE config 1,
 * ViT-Tiny/16 fp32).  Exact fp32 products on v_mfma_f32_32x32x2_f32; weights are the fp32 Flax [in, out] kernels in place.
 * savit_gemm_f32: C[M,N] = aux (nullable) + act( alpha_on_first_alpha_cols * (A[M,K] . W[K,N]) + bias ), act = tanh-GELU if gelu
 *   (nn.Dense / DenseGeneral: attention.py:29-37,60-63 with q / sqrt(hd) :39, ff.py:26-31, patch_embed.py:23-25, vit.py:96-98;
 *   residual adds vit.py:24,31).  K % 4 == 0, N % 4 == 0.
 * savit_layernorm_fwd_f32: nn.LayerNorm(dtype=float32), eps as given (vit.py:19,26,57).
 * savit_attention_fwd_f32: attention.py:41-57 per (batch, head) on a packed fp32 [B*N, ld] q|k|v buffer (q pre-scaled); N <= 256,
 *   head_dim <= 64.
 * savit_patchify_f32: patch_embed.py:19-22 rearrange on NHWC fp32 images -> [B*n, patch*patch*3].
 * savit_assemble_tokens_f32: x0 = concat([cls, tok]) + pos (vit.py:81-85, position_embed.py:56). */
int savit_gemm_f32(const float* A, const float* W, float* C, const float* bias, const float* aux, int M, int N, int K, int lda, int ldw,
                   int ldc, int ldaux, float alpha, int alpha_cols, int gelu, void* stream);
int savit_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, int rows, int d, long x_stride, long y_stride,
                            float eps, void* stream);
int savit_attention_fwd_f32(const float* qkv, float* o, int B, int N, int H, int head_dim, int ld_qkv, void* stream);

/* ---- fp32 arithmetic mode, general form (round 3): what the reference computes when create_model's dtype stays at its float32
 * default - which is ALWAYS the case for CaiT (models/create_model.py:50-213 do not forward dtype; cait.py:147-154) - and what
 * simple_train.py:72-90 differentiates.  Exact fp32 products (v_mfma_f32_32x32x2_f32), written for exactness and generality.
 * savit_gemm_f32_ex: for every batch z = outer * inner + in (pointer offsets outer * s?o + in * s?i elements; aux / C2 / U follow C):
 *     v = alpha_on_first_alpha_cols * (A . W) + bias ;  C2 = v if C2 ;  act 1: v = gelu_tanh(v), act 2: v = v * gelu_tanh'(U)
 *     C = (accumulate ? C : 0) + colscale[n] * rowscale[m / rows_per_sample] * v + aux
 *   A is [M, K] (transA: stored [K, M]), W is [K, N] (transW: stored [N, K]).  Serves Dense forward (attention.py:29-37,60-63,
 *   ff.py:26-31), its input-gradient (transW) and weight-gradient (transA, accumulate) products, Q K^T / P V and their VJPs per
 *   (image, head), LayerScale (layerscale.py:23) and stochastic depth (stochastic_depth.py:16-27) as colscale / rowscale.
 *   A plain accumulating product (accumulate with no other epilogue term: the weight gradients) with a long K and few output tiles
 *   is split over K ranges that add their partials with fp32 atomics: such results differ from run to run by summation order (1e-7).
 * savit_softmax_rows_f32 / _bwd_f32: nn.softmax over the last axis (attention.py:48) and dS = P (dP - sum_k dP P).
 * savit_head_mix_f32: TalkingHeadsBlock (talking_heads.py:13) y[b,i,e] = sum_h T[h,i] x[b,h,e], e over the N x N score positions.
 * savit_layernorm_bwd_f32: VJP of nn.LayerNorm(dtype=float32); dx (+ add, the residual cotangent), dgamma / dbeta accumulated.
 * savit_colsum_f32: out[n] += sum_m x[m,n] (Dense bias gradients).  savit_softmax_xent_grad_f32: d(mean label-smoothed CE)/dlogits
 *   in fp32 (train.py:83-90), scaled by grad_scale (1 / batch).
 * savit_layerscale_bwd_f32 (round 6): VJP of LayerScaleBlock x StochasticDepthBlock as cait.py:36-52 composes them in fp32
 *   (layerscale.py:18-23, stochastic_depth.py:16-27): dbranch[m,n] = dres[m,n] * ls[n] * rs[m / rows_per_sample],
 *   dls[n] += sum_m dres[m,n] * rs * branch[m,n] (rowscale nullable = 1; fp32 atomics on dls, one per column and 32-row block). */
typedef struct savit_gemm_f32_args {
  const float* A; const float* W; float* C; const float* bias; const float* aux; const float* colscale; const float* rowscale;
  float* C2; const float* U;
  int M, N, K, lda, ldw, ldc, ldaux;
  int transA, transW, batch, inner;
  long sAo, sAi, sWo, sWi, sCo, sCi;
  float alpha; int alpha_cols; int act; int accumulate; int rows_per_sample;
  int aux_row_mod; /* > 0: aux is a table [aux_row_mod, ldaux] shared by all batches, row m % aux_row_mod (position embeddings) */
  const float* rowbias; /* nullable: v += rowbias[m] before the activation (a Dense bias of a product computed transposed) */
} savit_gemm_f32_args;
int savit_gemm_f32_ex(const savit_gemm_f32_args* args, void* stream);
int savit_softmax_rows_f32(const float* x, float* y, long rows, int N, int ld, void* stream);
int savit_softmax_rows_bwd_f32(const float* p, const float* dp, float* ds, long rows, int N, int ld, void* stream);
int savit_head_mix_f32(const float* T, const float* x, float* y, int B, int H, long elems, void* stream);
int savit_layernorm_bwd_f32(const float* dy, const float* x, const float* gamma, const float* add, float* dx, float* dgamma, float* dbeta, int rows,
                            int d, long x_stride, long dy_stride, float eps, void* stream);
int savit_colsum_f32(const float* x, float* out, int M, int N, int ld, void* stream);
int savit_layerscale_bwd_f32(const float* dres, const float* branch, const float* ls, const float* rowscale, int rows_per_sample, float* dbranch,
                             float* dls, int M, int d, void* stream);
int savit_softmax_xent_grad_f32(const float* logits, const int* labels, float label_smoothing, float grad_scale, float* dlogits, int B, int C,
                                void* stream);
int savit_patchify_f32(const float* images, float* patches, int B, int img_size, int patch, void* stream);
/* The same rearrange on NHWC bf16 images (patch % 8 == 0; 16-byte accesses on both sides) into patches_bf16 [B * tokens, patch*patch*3]:
 * patch p of image b at row b * tokens + token_offset + p - the row of its token in the [B * tokens, d] activations; the other rows (the
 * cls slot) are not written (zero-fill once).  The dense operand of the patch-embed weight gradient when it runs as tiles of
 * savit_gemm_bf16_wgrad_grouped: X rows and cotangent rows line up, the cls rows contribute zeros (round 5). */
int savit_patchify_bf16(const void* images_bf16, void* patches_bf16, int B, int img_size, int patch, int tokens, int token_offset, void* stream);
int savit_assemble_tokens_f32(const float* tok, const float* cls, const float* pos, float* x0, int B, int N, int d, void* stream);

/* ---- measurement (bench.py, timing.py).  Not part of the training path: brackets for it.
 * savit_timer_*: a pool of HIP timing events created with hipEventDisableSystemFence - a default-flag event performs a system-scope
 *   release (cache writeback + invalidate) when recorded, which perturbs the launch being timed (round 3's bench line).  record() is
 *   asynchronous on `stream`; elapsed_ms() needs both events completed (synchronise the stream first).
 * savit_spin: one wave that waits `microseconds` on the 100 MHz constant clock (<= 200 000): the gate in front of an instrumented
 *   step, so the host enqueues the whole step behind it and no event pair contains host time.
 * savit_hold_cus: `cus` one-wave workgroups holding 96 KB of LDS each for `microseconds` - stands in for the CUs a resident RCCL
 *   all-reduce occupies during backward (row a16 / e: tools/cu_thief_probe.py).
 * savit_zero_bytes: hipMemsetAsync(dst, 0, bytes) as a plan entry (gradient / loss accumulators). */
int savit_timer_create(int n_events, void** handle);
int savit_timer_record(void* handle, int idx, void* stream);
int savit_timer_elapsed_ms(void* handle, int first, int second, float* ms);
int savit_timer_destroy(void* handle);
int savit_spin(long microseconds, void* stream);
int savit_hold_cus(int cus, long microseconds, void* stream);
/* CUs the persistent kernels (attention forward / backward: one workgroup per CU walking over (image, head) items) size their grids
 * for; 0 (default) = every CU of the device.  A data-parallel rank sets its launch plan's budget (device CUs - reserved_cus) so that no
 * persistent workgroup has to wait for a CU that RCCL's channels hold.  Process-wide; results do not depend on it. */
int savit_set_cu_budget(int cus);
int savit_zero_bytes(void* dst, long bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SAVIT_H_ */
