// fp32 arithmetic mode of the forward path (BASELINE config 1: ViT-Tiny/16, fp32; the reference's create_model() defaults to
// dtype=float32, /root/reference/models/create_model.py:6-8).  Forward + loss only: training runs in bf16 (the MFMA path).
//
// gfx950 has an exact fp32-input MFMA (v_mfma_f32_32x32x2_f32: a k-ordered fmaf chain, 1/16 of the bf16 rate) and no
// reduced-precision fp32 form, so the Dense products run on it: no bf16 splitting, results within fp32 summation order of the
// oracle.  Flax kernels are [in, out], which is exactly the B-operand layout of that instruction (lane = output column, one k per
// lane half): the fp32 master weights are used in place, no transposed or cast copy exists.
//   * savit_gemm_f32        C[M,N] = epi( A[M,K] . W[K,N] ): 64x64x32 tiles through LDS, 4 waves of one 32x32 accumulator each;
//                           epilogue: alpha on the first alpha_cols columns (q / sqrt(hd), attention.py:39), + bias, tanh-GELU,
//                           + residual (vit.py:24,31).
//   * savit_layernorm_fwd_f32   flax nn.LayerNorm with dtype float32 (fp32 statistics AND fp32 output, no parameter rounding).
//   * savit_attention_fwd_f32   attention.py:41-57 per (batch, head): K and V of the head in LDS as fp32, one wave per query row
//                           (scores on the lanes, softmax by wave reductions, P.V with the head dimension on the lanes).  N <= 256.
//   * savit_patchify_f32 / savit_assemble_tokens_f32   patch_embed.py:19-22 and vit.py:81-85 + position_embed.py:56 around the GEMM.
// Sizes here are small (ViT-Ti at batch 8 is 20 GFLOP per forward): these kernels are written for exactness and clarity first.
#include "common.h"
#include "savit.h"

namespace {

constexpr int GB = 64, GK = 32;        // block tile, K step
constexpr int A_LD = GK + 1;           // padded: lane i reads As[i][k] - distinct banks
constexpr int B_LD = GB + 4;           // 16-B aligned rows

struct GemmF32Params {
  const float* A; const float* W; float* C; const float* bias; const float* aux;
  int M, N, K, lda, ldw, ldc, ldaux;
  float alpha; int alpha_cols; int gelu;
};

__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmF32Params p) {
  __shared__ float As[GB * A_LD];
  __shared__ __attribute__((aligned(16))) float Bs[GK * B_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int row0 = blockIdx.y * GB, col0 = blockIdx.x * GB;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int li = lane & 31, lk = lane >> 5;
  for (int k0 = 0; k0 < p.K; k0 += GK) {
    // A tile: 64 rows x 32 k; thread -> (row = tid/8 + 32 r, 4 consecutive k)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int row = (tid >> 3) + 32 * r, kc = (tid & 7) * 4;
      const int m = row0 + row, k = k0 + kc;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < p.M && k < p.K) v = *reinterpret_cast<const float4*>(p.A + (size_t)m * p.lda + k);  // K % 4 == 0
      float* d = As + row * A_LD + kc;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    // W tile: 32 k x 64 columns; thread -> (k = tid/16 + 16 r, 4 consecutive columns)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int kr = (tid >> 4) + 16 * r, nc = (tid & 15) * 4;
      const int k = k0 + kr, n = col0 + nc;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k < p.K && n < p.N) v = *reinterpret_cast<const float4*>(p.W + (size_t)k * p.ldw + n);  // N % 4 == 0
      *reinterpret_cast<float4*>(Bs + kr * B_LD + nc) = v;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < GK / 2; ++s) {
      const float a = As[(wm * 32 + li) * A_LD + 2 * s + lk];
      const float b = Bs[(2 * s + lk) * B_LD + wn * 32 + li];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  const int n = col0 + wn * 32 + li;
  if (n >= p.N) return;
  const float bn = p.bias ? p.bias[n] : 0.f;
  const float sc = n < p.alpha_cols ? p.alpha : 1.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = row0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
    if (m >= p.M) continue;
    float v = acc[r] * sc + bn;
    if (p.gelu) {  // jax.nn.gelu(approximate=True) (ff.py:28), full-precision tanh
      const float z = 0.7978845608028654f * (v + 0.044715f * v * v * v);
      v = 0.5f * v * (1.0f + tanhf(z));
    }
    if (p.aux) v += p.aux[(size_t)m * p.ldaux + n];
    p.C[(size_t)m * p.ldc + n] = v;
  }
}

// one wave per row, lane i holds elements i, i + 64, ...  (d <= 4096)
__global__ __launch_bounds__(256) void ln_fwd_f32_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ y, int rows, int d, long x_stride, long y_stride, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (size_t)row * x_stride;
  float s = 0.f, s2 = 0.f;
  for (int c = lane; c < d; c += 64) {
    const float v = xr[c];
    s += v;
    s2 += v * v;
  }
  s = wave_sum(s);
  s2 = wave_sum(s2);
  const float mean = s / (float)d;
  const float var = s2 / (float)d - mean * mean;  // flax: E[x^2] - E[x]^2
  const float rstd = 1.0f / sqrtf(var + eps);
  float* yr = y + (size_t)row * y_stride;
  for (int c = lane; c < d; c += 64) yr[c] = (xr[c] - mean) * (rstd * gamma[c]) + beta[c];
}

// qkv fp32 [B*N, ld]: q (pre-scaled) | k | v, head-major inside each.  One workgroup per (batch, head); K and V rows of the head in
// LDS; a wave owns query rows wave, wave + NW, ...: scores with keys on the lanes (<= 4 per lane), output with the head dimension
// on the lanes (hd <= 64).
constexpr int AF_KPL = 4;
__global__ __launch_bounds__(256) void attn_fwd_f32_kernel(const float* __restrict__ qkv, float* __restrict__ o, int B, int N, int H, int hd, int ld) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* Ks = reinterpret_cast<float*>(smem_raw);       // [N][hd + 1]
  float* Vs = Ks + (size_t)N * (hd + 1);                // [N][hd]
  float* Ps = Vs + (size_t)N * hd;                      // [4 waves][AF_KPL * 64]
  const int b = blockIdx.x / H, h = blockIdx.x - b * H, d = H * hd;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < N * hd; i += 256) {
    const int t = i / hd, e = i - t * hd;
    const float* row = qkv + (size_t)(b * N + t) * ld;
    Ks[t * (hd + 1) + e] = row[d + h * hd + e];
    Vs[t * hd + e] = row[2 * d + h * hd + e];
  }
  __syncthreads();
  float* pw = Ps + wave * (AF_KPL * 64);
  for (int q = wave; q < N; q += 4) {
    const float* qrow = qkv + (size_t)(b * N + q) * ld + h * hd;
    float s[AF_KPL], m = -INFINITY;
#pragma unroll
    for (int kk = 0; kk < AF_KPL; ++kk) {
      const int key = lane + 64 * kk;
      s[kk] = -INFINITY;
      if (key < N) {
        float a = 0.f;
        for (int e = 0; e < hd; ++e) a = fmaf(qrow[e], Ks[key * (hd + 1) + e], a);
        s[kk] = a;
      }
      m = fmaxf(m, s[kk]);
    }
    m = wave_max(m);
    float l = 0.f;
#pragma unroll
    for (int kk = 0; kk < AF_KPL; ++kk) {
      s[kk] = (lane + 64 * kk < N) ? expf(s[kk] - m) : 0.f;
      l += s[kk];
    }
    l = wave_sum(l);
    const float inv = 1.0f / l;
#pragma unroll
    for (int kk = 0; kk < AF_KPL; ++kk) pw[lane + 64 * kk] = s[kk] * inv;
    __builtin_amdgcn_wave_barrier();  // a wave's LDS accesses complete in order: its reads below see these writes
    if (lane < hd) {
      float acc = 0.f;
      for (int key = 0; key < N; ++key) acc = fmaf(pw[key], Vs[key * hd + lane], acc);
      o[(size_t)(b * N + q) * d + h * hd + lane] = acc;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// einops 'b (h ph) (w pw) c -> b (h w) (ph pw c)' on fp32 NHWC images (patch_embed.py:19-22)
__global__ __launch_bounds__(256) void patchify_f32_kernel(const float* __restrict__ img, float* __restrict__ out, int B, int S, int P) {
  const int g = S / P, pd = P * P * 3;
  const long total = (long)B * g * g * pd;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int e = (int)(i % pd);
    const long pi = i / pd;
    const int pw_ = (int)(pi % g), ph_ = (int)((pi / g) % g), b = (int)(pi / ((long)g * g));
    const int c = e % 3, x = (e / 3) % P, y = e / (3 * P);
    out[i] = img[(((size_t)b * S + ph_ * P + y) * S + pw_ * P + x) * 3 + c];
  }
}

// x0[b, 0] = cls + pos[0];  x0[b, 1 + p] = tok[b, p] + pos[1 + p]   (vit.py:81-85, position_embed.py:56)
__global__ __launch_bounds__(256) void assemble_tokens_f32_kernel(const float* __restrict__ tok, const float* __restrict__ cls, const float* __restrict__ pos,
                                                                   float* __restrict__ x0, int B, int N, int d) {
  const long total = (long)B * N * d;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % d);
    const int t = (int)((i / d) % N);
    const int b = (int)(i / ((long)d * N));
    const float v = t == 0 ? cls[c] : tok[((size_t)b * (N - 1) + (t - 1)) * d + c];
    x0[i] = v + pos[(size_t)t * d + c];
  }
}


// ---------------------------------------------------------------------------------------------------------------- round 3
// The general form of the fp32 product, for everything the forward-only kernels above do not cover: batched (grid.z, a batch index
// split into an outer and an inner part with their own strides: (image, head) for attention), either operand transposed in place
// (the input-gradient products dY.W^T, the weight-gradient products X^T.dY, Q.K^T, dS^T.Q, P^T.dO), epilogue
//   v = alpha_on_first_alpha_cols * (A.W) + bias ; [C2 = v (pre-activation saved)] ; act 1: v = gelu_tanh(v) ; act 2: v = v * gelu'(U) ;
//   v = colscale[n] * rowscale[m / rows_per_sample] * v ; v += aux ; C = (accumulate ? C : 0) + v
// (ff.py:26-33 and its VJP, LayerScale layerscale.py:23, stochastic depth stochastic_depth.py:16-27, residual adds vit.py:24,31).
// Exact fp32 on v_mfma_f32_32x32x2_f32 like gemm_f32_kernel; written for exactness and generality, not speed.
struct GemmF32ExParams {
  const float* A; const float* W; float* C; const float* bias; const float* aux; const float* colscale; const float* rowscale;
  float* C2; const float* U;
  int M, N, K, lda, ldw, ldc, ldaux;
  int transA, transW;                 // A stored [K, M] / W stored [N, K]
  int inner;                          // batch z = outer * inner + in
  long sAo, sAi, sWo, sWi, sCo, sCi;  // element strides of the batch parts (aux, C2, U follow C)
  float alpha; int alpha_cols; int act; int accumulate; int rows_per_sample;
  int aux_row_mod;  // > 0: aux is a [aux_row_mod, ldaux] table shared by all batches, row m % aux_row_mod (position embeddings)
  const float* rowbias;  // nullable: + rowbias[m] (a Dense bias when the product is computed transposed: MLP-Mixer token mixing)
  int ksplit;            // K ranges per output tile (chosen by the launcher)
};

__device__ __forceinline__ float gelu_tanh_exact(float v) {
  const float z = 0.7978845608028654f * (v + 0.044715f * v * v * v);
  return 0.5f * v * (1.0f + tanhf(z));
}
__device__ __forceinline__ float gelu_tanh_grad_exact(float v) {
  const float z = 0.7978845608028654f * (v + 0.044715f * v * v * v);
  const float t = tanhf(z);
  return 0.5f * (1.0f + t) + 0.5f * v * (1.0f - t * t) * 0.7978845608028654f * (1.0f + 3.0f * 0.044715f * v * v);
}

__global__ __launch_bounds__(256) void gemm_f32_ex_kernel(const GemmF32ExParams p) {
  __shared__ float As[GB * A_LD];
  __shared__ float Bs[GK * B_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int row0 = blockIdx.y * GB, col0 = blockIdx.x * GB;
  // grid.z = batch x K-splits (splits > 1 only for plain accumulating products: each split adds its partial with an atomic)
  const int zb = blockIdx.z / p.ksplit, ks = blockIdx.z - zb * p.ksplit;
  const int zo = zb / p.inner, zi = zb - zo * p.inner;
  const float* A = p.A + zo * p.sAo + zi * p.sAi;
  const float* W = p.W + zo * p.sWo + zi * p.sWi;
  const long coff = zo * p.sCo + zi * p.sCi;
  const int kchunk = ((p.K + p.ksplit - 1) / p.ksplit + GK - 1) / GK * GK;
  const int kbeg = ks * kchunk, kend = min(p.K, kbeg + kchunk);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if (kbeg >= kend) return;
  const int li = lane & 31, lk = lane >> 5;
  // the next K-step's operand elements are fetched into registers under this K-step's MFMAs (the launches of the fp32 engines are
  // small: a workgroup's loop is a chain of global-memory latencies otherwise)
  float ra[8], rw[8];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {  // A tile 64 (m) x 32 (k), walking the operand's contiguous index fastest
      const int e = tid + 256 * r;
      int row, kc;
      if (p.transA) { row = e & 63; kc = e >> 6; } else { kc = e & 31; row = e >> 5; }
      const int m = row0 + row, k = k0 + kc;
      ra[r] = 0.f;
      if (m < p.M && k < kend) ra[r] = p.transA ? A[(size_t)k * p.lda + m] : A[(size_t)m * p.lda + k];
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {  // W tile 32 (k) x 64 (n)
      const int e = tid + 256 * r;
      int kr, nc;
      if (p.transW) { kr = e & 31; nc = e >> 5; } else { nc = e & 63; kr = e >> 6; }
      const int k = k0 + kr, n = col0 + nc;
      rw[r] = 0.f;
      if (k < kend && n < p.N) rw[r] = p.transW ? W[(size_t)n * p.ldw + k] : W[(size_t)k * p.ldw + n];
    }
  };
  fetch(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += GK) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int e = tid + 256 * r;
      int row, kc;
      if (p.transA) { row = e & 63; kc = e >> 6; } else { kc = e & 31; row = e >> 5; }
      As[row * A_LD + kc] = ra[r];
      int kr, nc;
      if (p.transW) { kr = e & 31; nc = e >> 5; } else { nc = e & 63; kr = e >> 6; }
      Bs[kr * B_LD + nc] = rw[r];
    }
    __syncthreads();
    if (k0 + GK < kend) fetch(k0 + GK);
#pragma unroll
    for (int s = 0; s < GK / 2; ++s) {
      const float a = As[(wm * 32 + li) * A_LD + 2 * s + lk];
      const float b = Bs[(2 * s + lk) * B_LD + wn * 32 + li];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  const int n = col0 + wn * 32 + li;
  if (n >= p.N) return;
  if (p.ksplit > 1) {  // C += partial (plain accumulating product: the launcher allows no other epilogue term with a split)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = row0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
      if (m < p.M) atomicAdd(p.C + (size_t)coff + (size_t)m * p.ldc + n, acc[r]);
    }
    return;
  }
  const float bn = p.bias ? p.bias[n] : 0.f;
  const float sc = n < p.alpha_cols ? p.alpha : 1.0f;
  const float cs = p.colscale ? p.colscale[n] : 1.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = row0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
    if (m >= p.M) continue;
    const size_t o = (size_t)coff + (size_t)m * p.ldc + n;
    float v = acc[r] * sc + bn;
    if (p.rowbias) v += p.rowbias[m];
    if (p.C2) p.C2[o] = v;
    if (p.act == 1) v = gelu_tanh_exact(v);
    if (p.act == 2) v *= gelu_tanh_grad_exact(p.U[o]);
    v *= cs;
    if (p.rowscale) v *= p.rowscale[m / p.rows_per_sample];
    if (p.aux) v += p.aux_row_mod > 0 ? p.aux[(size_t)(m % p.aux_row_mod) * p.ldaux + n] : p.aux[(size_t)coff + (size_t)m * p.ldaux + n];
    p.C[o] = p.accumulate ? p.C[o] + v : v;
  }
}

// softmax over the last axis (attention.py:48) of `rows` rows of N values, and its VJP dS = P (dP - sum_k dP P); one wave per row
// x and y may be the same buffer (the engines normalise in place: each lane reads its elements before it writes them) - no __restrict__
__global__ __launch_bounds__(256) void softmax_rows_f32_kernel(const float* x, float* y, long rows, int N, int ld) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * ld;
  float* yr = y + row * ld;
  float m = -INFINITY;
  for (int c = lane; c < N; c += 64) m = fmaxf(m, xr[c]);
  m = wave_max(m);
  float l = 0.f;
  for (int c = lane; c < N; c += 64) l += expf(xr[c] - m);
  l = wave_sum(l);
  const float inv = 1.0f / l;
  for (int c = lane; c < N; c += 64) yr[c] = expf(xr[c] - m) * inv;
}
// dp and ds may be the same buffer (in-place use by the engines) - no __restrict__ on that pair
__global__ __launch_bounds__(256) void softmax_rows_bwd_f32_kernel(const float* __restrict__ pr, const float* dp, float* ds, long rows,
                                                                    int N, int ld) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* p = pr + row * ld;
  const float* d = dp + row * ld;
  float s = 0.f;
  for (int c = lane; c < N; c += 64) s += p[c] * d[c];
  s = wave_sum(s);
  for (int c = lane; c < N; c += 64) ds[row * ld + c] = p[c] * (d[c] - s);
}

// talking heads (talking_heads.py:13): y[b, i, e] = sum_h T[h, i] x[b, h, e] over e = the N x N positions; H <= 16
__global__ __launch_bounds__(256) void head_mix_f32_kernel(const float* __restrict__ T, const float* __restrict__ x, float* __restrict__ y, int B, int H, long E) {
  const long total = (long)B * E;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long b = i / E, e = i - b * E;
    float xv[16];
    for (int h = 0; h < H; ++h) xv[h] = x[((size_t)b * H + h) * E + e];
    for (int o = 0; o < H; ++o) {
      float a = 0.f;
      for (int h = 0; h < H; ++h) a = fmaf(T[h * H + o], xv[h], a);
      y[((size_t)b * H + o) * E + e] = a;
    }
  }
}

// nn.LayerNorm(dtype=float32) VJP: dx = rstd (g - mean(g) - xhat mean(g xhat)) [+ add], g = gamma dy; dgamma += sum dy xhat, dbeta += sum dy.
// One wave per row, LNB_ROWS rows per wave; a lane keeps the dgamma / dbeta partials of its columns (lane + 64 j) in registers, the four
// waves fold them in LDS, and the block issues ONE atomic per column (d <= 64 LNB_COLS; wider rows take the per-row atomics).
// add and dx may be the same buffer (the residual cotangent updated in place): neither is __restrict__.
constexpr int LNB_ROWS = 8, LNB_COLS = 16;
template <bool WIDE>
__global__ __launch_bounds__(256) void ln_bwd_f32_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* add, float* dx, float* __restrict__ dgamma, float* __restrict__ dbeta, int rows, int d,
                                                          long x_stride, long dy_stride, float eps) {
  __shared__ float sg[WIDE ? 1 : 64 * LNB_COLS], sb[WIDE ? 1 : 64 * LNB_COLS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float pg[LNB_COLS], pb[LNB_COLS];
#pragma unroll
  for (int j = 0; j < LNB_COLS; ++j) pg[j] = pb[j] = 0.f;
  if constexpr (!WIDE) {
    for (int c = threadIdx.x; c < d; c += 256) sg[c] = sb[c] = 0.f;
    __syncthreads();
  }
  for (int rr = 0; rr < LNB_ROWS; ++rr) {
    const int row = (blockIdx.x * 4 + wave) * LNB_ROWS + rr;
    if (row >= rows) break;
    const float* xr = x + (size_t)row * x_stride;
    const float* gr = dy + (size_t)row * dy_stride;
    float s = 0.f, s2 = 0.f;
    for (int c = lane; c < d; c += 64) {
      const float v = xr[c];
      s += v;
      s2 += v * v;
    }
    s = wave_sum(s);
    s2 = wave_sum(s2);
    const float mean = s / (float)d;
    const float rstd = 1.0f / sqrtf(s2 / (float)d - mean * mean + eps);
    float a = 0.f, b = 0.f;
    for (int c = lane; c < d; c += 64) {
      const float g = gamma[c] * gr[c], xh = (xr[c] - mean) * rstd;
      a += g;
      b += g * xh;
    }
    a = wave_sum(a) / (float)d;
    b = wave_sum(b) / (float)d;
    if constexpr (WIDE) {
      for (int c = lane; c < d; c += 64) {
        const float xh = (xr[c] - mean) * rstd;
        float v = rstd * (gamma[c] * gr[c] - a - xh * b);
        if (add) v += add[(size_t)row * x_stride + c];
        dx[(size_t)row * x_stride + c] = v;
        atomicAdd(dgamma + c, gr[c] * xh);
        atomicAdd(dbeta + c, gr[c]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < LNB_COLS; ++j) {
        const int c = lane + 64 * j;
        if (c < d) {
          const float xh = (xr[c] - mean) * rstd;
          float v = rstd * (gamma[c] * gr[c] - a - xh * b);
          if (add) v += add[(size_t)row * x_stride + c];
          dx[(size_t)row * x_stride + c] = v;
          pg[j] += gr[c] * xh;
          pb[j] += gr[c];
        }
      }
    }
  }
  if constexpr (!WIDE) {
#pragma unroll
    for (int j = 0; j < LNB_COLS; ++j) {
      const int c = lane + 64 * j;
      if (c < d) {
        atomicAdd(&sg[c], pg[j]);
        atomicAdd(&sb[c], pb[j]);
      }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += 256) {
      atomicAdd(dgamma + c, sg[c]);
      atomicAdd(dbeta + c, sb[c]);
    }
  }
}

// LayerScale x stochastic-depth VJP in fp32 (layerscale.py:18-23, stochastic_depth.py:16-27 as cait.py:36-52 composes them: the residual
// stream received ls[n] * rs[m / rows_per_sample] * branch[m, n]).  Given the residual cotangent dres:
//   dbranch[m, n] = dres[m, n] * ls[n] * rs ;  dls[n] += sum_m dres[m, n] * rs * branch[m, n]
// A block takes LSB_ROWS rows, a thread the columns t, t + 256, ... (coalesced rows), one atomic per column and block.
constexpr int LSB_ROWS = 32;
__global__ __launch_bounds__(256) void layerscale_bwd_f32_kernel(const float* __restrict__ dres, const float* __restrict__ branch,
                                                                  const float* __restrict__ ls, const float* __restrict__ rowscale, int rows_per_sample,
                                                                  float* __restrict__ dbranch, float* __restrict__ dls, int M, int d) {
  const int m0 = blockIdx.x * LSB_ROWS, m1 = min(M, m0 + LSB_ROWS);
  for (int c = threadIdx.x; c < d; c += 256) {
    const float l = ls[c];
    float s = 0.f;
    for (int m = m0; m < m1; ++m) {
      const float rs = rowscale ? rowscale[m / rows_per_sample] : 1.0f;
      const float g = dres[(size_t)m * d + c] * rs;
      dbranch[(size_t)m * d + c] = g * l;
      s = fmaf(g, branch[(size_t)m * d + c], s);
    }
    atomicAdd(dls + c, s);
  }
}

// out[n] += sum_m x[m, n] (bias gradients); out[i] = a[i] + b[i]; dlogits of train.py:83-90 in fp32
__global__ __launch_bounds__(256) void colsum_f32_kernel(const float* __restrict__ x, float* __restrict__ out, int M, int N, int ld) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const int chunk = (M + gridDim.y - 1) / gridDim.y;
  const int m0 = blockIdx.y * chunk, m1 = min(M, m0 + chunk);
  float s = 0.f;
  for (int m = m0; m < m1; ++m) s += x[(size_t)m * ld + n];
  atomicAdd(out + n, s);
}
__global__ __launch_bounds__(256) void xent_grad_f32_kernel(const float* __restrict__ logits, const int* __restrict__ labels, float alpha, float scale,
                                                             float* __restrict__ dz, int B, int C) {
  __shared__ float red[4];
  const int row = blockIdx.x;
  const float* z = logits + (size_t)row * C;
  float m = -INFINITY;
  for (int c = threadIdx.x; c < C; c += 256) m = fmaxf(m, z[c]);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) s += expf(z[c] - m);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  s = red[0] + red[1] + red[2] + red[3];
  const int l = labels[row];
  for (int c = threadIdx.x; c < C; c += 256) {
    const float y = (1.0f - alpha) * (c == l ? 1.f : 0.f) + alpha / (float)C;
    dz[(size_t)row * C + c] = (expf(z[c] - m) / s - y) * scale;
  }
}

}  // namespace

extern "C" int savit_gemm_f32(const float* A, const float* W, float* C, const float* bias, const float* aux, int M, int N, int K, int lda, int ldw,
                              int ldc, int ldaux, float alpha, int alpha_cols, int gelu, void* stream) {
  SAVIT_CHECK_ARG(A && W && C && M >= 0 && N > 0 && K > 0 && K % 4 == 0 && N % 4 == 0 && lda >= K && ldw >= N && ldc >= N);
  SAVIT_CHECK_ARG(lda % 4 == 0 && ldw % 4 == 0 && ((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && (aux == nullptr || ldaux >= N));
  if (M == 0) return SAVIT_OK;
  GemmF32Params p{A, W, C, bias, aux, M, N, K, lda, ldw, ldc, ldaux, alpha, alpha_cols, gelu};
  hipLaunchKernelGGL(gemm_f32_kernel, dim3((N + GB - 1) / GB, (M + GB - 1) / GB), dim3(256), 0, (hipStream_t)stream, p);
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, int rows, int d, long x_stride,
                                       long y_stride, float eps, void* stream) {
  SAVIT_CHECK_ARG(x && gamma && beta && y && rows >= 0 && d > 0 && x_stride >= d && y_stride >= d);
  if (rows == 0) return SAVIT_OK;
  hipLaunchKernelGGL(ln_fwd_f32_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, rows, d, x_stride, y_stride, eps);
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_attention_fwd_f32(const float* qkv, float* o, int B, int N, int H, int head_dim, int ld_qkv, void* stream) {
  SAVIT_CHECK_ARG(qkv && o && B >= 0 && N > 0 && N <= 64 * AF_KPL && H > 0 && head_dim > 0 && head_dim <= 64 && ld_qkv >= 3 * H * head_dim);
  if (B == 0) return SAVIT_OK;
  const size_t lds = ((size_t)N * (head_dim + 1) + (size_t)N * head_dim + 4 * AF_KPL * 64) * sizeof(float);
  SAVIT_CHECK_ARG(lds <= 160 * 1024);
  SAVIT_LDS_ONCE(attn_fwd_f32_kernel);
  hipLaunchKernelGGL(attn_fwd_f32_kernel, dim3(B * H), dim3(256), lds, (hipStream_t)stream, qkv, o, B, N, H, head_dim, ld_qkv);
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_patchify_f32(const float* images, float* patches, int B, int img_size, int patch, void* stream) {
  SAVIT_CHECK_ARG(images && patches && B >= 0 && patch > 0 && img_size > 0 && img_size % patch == 0);
  if (B == 0) return SAVIT_OK;
  const long total = (long)B * img_size * img_size * 3;
  long blocks = (total + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(patchify_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, images, patches, B, img_size, patch);
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_assemble_tokens_f32(const float* tok, const float* cls, const float* pos, float* x0, int B, int N, int d, void* stream) {
  SAVIT_CHECK_ARG(tok && cls && pos && x0 && B >= 0 && N >= 2 && d > 0);
  if (B == 0) return SAVIT_OK;
  const long total = (long)B * N * d;
  long blocks = (total + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(assemble_tokens_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, tok, cls, pos, x0, B, N, d);
  SAVIT_LAUNCH_RET();
}

// ---- round 3: the general fp32 entries (header: include/savit.h)
extern "C" int savit_gemm_f32_ex(const savit_gemm_f32_args* g, void* stream) {
  SAVIT_CHECK_ARG(g && g->A && g->W && g->C && g->M >= 0 && g->N > 0 && g->K > 0 && g->batch >= 1 && g->inner >= 1 && g->batch % g->inner == 0);
  SAVIT_CHECK_ARG(g->lda >= (g->transA ? g->M : g->K) && g->ldw >= (g->transW ? g->K : g->N) && g->ldc >= g->N && (g->aux == nullptr || g->ldaux >= g->N));
  SAVIT_CHECK_ARG((g->act != 2 || g->U != nullptr) && (g->rowscale == nullptr || g->rows_per_sample >= 1) && g->act >= 0 && g->act <= 2 && g->aux_row_mod >= 0);
  if (g->M == 0) return SAVIT_OK;
  // A long reduction into few output tiles (weight gradients: K = tokens) is split over K ranges that add their partials with fp32
  // atomics - only for the plain accumulating form, where the order of the additions is the only thing a split changes.
  int ksplit = 1;
  const long tiles = (long)((g->N + GB - 1) / GB) * ((g->M + GB - 1) / GB) * g->batch;
  const bool plain_acc = g->accumulate && !g->bias && !g->aux && !g->colscale && !g->rowscale && !g->C2 && !g->rowbias && g->act == 0 &&
                         g->alpha_cols == 0;
  if (plain_acc && g->K >= 8 * GK && tiles < 1024) {
    ksplit = (int)((1024 + tiles - 1) / tiles);
    if (ksplit > g->K / (2 * GK)) ksplit = g->K / (2 * GK);
    if (ksplit < 1) ksplit = 1;
    if ((long)g->batch * ksplit > 65535) ksplit = 1;
  }
  GemmF32ExParams p{g->A, g->W, g->C, g->bias, g->aux, g->colscale, g->rowscale, g->C2, g->U, g->M, g->N, g->K, g->lda, g->ldw, g->ldc, g->ldaux,
                    g->transA, g->transW, g->inner, g->sAo, g->sAi, g->sWo, g->sWi, g->sCo, g->sCi, g->alpha, g->alpha_cols, g->act, g->accumulate,
                    g->rows_per_sample > 0 ? g->rows_per_sample : 1, g->aux_row_mod, g->rowbias, ksplit};
  hipLaunchKernelGGL(gemm_f32_ex_kernel, dim3((g->N + GB - 1) / GB, (g->M + GB - 1) / GB, g->batch * ksplit), dim3(256), 0, (hipStream_t)stream, p);
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_softmax_rows_f32(const float* x, float* y, long rows, int N, int ld, void* stream) {
  SAVIT_CHECK_ARG(x && y && rows >= 0 && N > 0 && ld >= N);
  if (rows == 0) return SAVIT_OK;
  hipLaunchKernelGGL(softmax_rows_f32_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, y, rows, N, ld);
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_softmax_rows_bwd_f32(const float* p, const float* dp, float* ds, long rows, int N, int ld, void* stream) {
  SAVIT_CHECK_ARG(p && dp && ds && rows >= 0 && N > 0 && ld >= N);
  if (rows == 0) return SAVIT_OK;
  hipLaunchKernelGGL(softmax_rows_bwd_f32_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p, dp, ds, rows, N, ld);
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_head_mix_f32(const float* T, const float* x, float* y, int B, int H, long elems, void* stream) {
  SAVIT_CHECK_ARG(T && x && y && x != y && B >= 0 && H >= 1 && H <= 16 && elems > 0);
  if (B == 0) return SAVIT_OK;
  long blocks = ((long)B * elems + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(head_mix_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, T, x, y, B, H, elems);
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_layernorm_bwd_f32(const float* dy, const float* x, const float* gamma, const float* add, float* dx, float* dgamma, float* dbeta,
                                       int rows, int d, long x_stride, long dy_stride, float eps, void* stream) {
  SAVIT_CHECK_ARG(dy && x && gamma && dx && dgamma && dbeta && rows >= 0 && d > 0 && x_stride >= d && dy_stride >= d);
  if (rows == 0) return SAVIT_OK;
  const dim3 grid((rows + 4 * LNB_ROWS - 1) / (4 * LNB_ROWS));
  if (d <= 64 * LNB_COLS)
    hipLaunchKernelGGL(ln_bwd_f32_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, dy, x, gamma, add, dx, dgamma, dbeta, rows, d, x_stride, dy_stride, eps);
  else
    hipLaunchKernelGGL(ln_bwd_f32_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, dy, x, gamma, add, dx, dgamma, dbeta, rows, d, x_stride, dy_stride, eps);
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_layerscale_bwd_f32(const float* dres, const float* branch, const float* ls, const float* rowscale, int rows_per_sample,
                                       float* dbranch, float* dls, int M, int d, void* stream) {
  SAVIT_CHECK_ARG(dres && branch && ls && dbranch && dls && dbranch != dres && M >= 0 && d > 0 && (rowscale == nullptr || rows_per_sample >= 1));
  if (M == 0) return SAVIT_OK;
  hipLaunchKernelGGL(layerscale_bwd_f32_kernel, dim3((M + LSB_ROWS - 1) / LSB_ROWS), dim3(256), 0, (hipStream_t)stream, dres, branch, ls, rowscale,
                     rows_per_sample > 0 ? rows_per_sample : 1, dbranch, dls, M, d);
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_colsum_f32(const float* x, float* out, int M, int N, int ld, void* stream) {
  SAVIT_CHECK_ARG(x && out && M >= 0 && N > 0 && ld >= N);
  if (M == 0) return SAVIT_OK;
  const int split = M >= 4096 ? 32 : (M >= 256 ? 8 : 1);
  hipLaunchKernelGGL(colsum_f32_kernel, dim3((N + 255) / 256, split), dim3(256), 0, (hipStream_t)stream, x, out, M, N, ld);
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_softmax_xent_grad_f32(const float* logits, const int* labels, float label_smoothing, float grad_scale, float* dlogits, int B, int C,
                                           void* stream) {
  SAVIT_CHECK_ARG(logits && labels && dlogits && B >= 0 && C > 0);
  if (B == 0) return SAVIT_OK;
  hipLaunchKernelGGL(xent_grad_f32_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, labels, label_smoothing, grad_scale, dlogits, B, C);
  SAVIT_LAUNCH_RET();
}
