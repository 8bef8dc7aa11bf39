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
