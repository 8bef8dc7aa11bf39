// GPU-side input path: the steps either side of train_step's `'H W C N -> N H W C'` + bf16 cast
// (/root/reference/train.py:80-81) that the reference runs in its TF host pipeline:
//   * mean / std normalisation            data/preprocess/preprocess.py:176-179, data/constants.py:7-10
//   * batch mixup                         data/preprocess/augment_ops.py:144-181   xmix = x*mix + x[index]*(1-mix)
//   * batch cutmix                        data/preprocess/augment_ops.py:98-141    where(box, x, x[::-1])
// All are one pass over the batch (HBM-bound: 4 or 1 B read + 2 B written per element for normalise, 2x2 B read + 2 B written
// for the mixes), fp32 arithmetic, ONE rounding to bf16 at the end - the cast train.py:81 applies to the pipeline's fp32 images.
// The random draws (mix weights, permutation, boxes) are INPUTS: TF's stateless RNG stream cannot be reproduced, so the host
// samples them (savit_amd/augment.py) and the kernels stay deterministic and testable bit for bit.
#include "common.h"
#include "savit.h"

namespace {

// src [N,H,W,C] fp32 or u8 -> dst [N,H,W,C] bf16:  (src*scale - mean[c]) * inv_std[c];  C == 3 or 1..4; 4 pixels... one
// thread handles 8 consecutive elements of the flat image (the channel of element e is e % C).
template <typename T>
__global__ __launch_bounds__(256) void normalize_nhwc_kernel(const T* __restrict__ src, bf16_t* __restrict__ dst, long n, int C, float scale,
                                                              float m0, float m1, float m2, float m3, float s0, float s1, float s2, float s3) {
  const float mean[4] = {m0, m1, m2, m3}, istd[4] = {s0, s1, s2, s3};
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += (long)gridDim.x * blockDim.x * 8) {
    float v[8];
    const int lim = (n - i) < 8 ? (int)(n - i) : 8;
    int c = (int)(i % C);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      v[k] = 0.f;
      if (k < lim) v[k] = ((float)src[i + k] * scale - mean[c]) * istd[c];
      c = (c + 1 == C) ? 0 : c + 1;
    }
    if (lim == 8 && (((uintptr_t)(dst + i)) & 15) == 0) {
      *reinterpret_cast<uint4*>(dst + i) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
    } else {
      for (int k = 0; k < lim; ++k) dst[i + k] = f32_to_bf16(v[k]);
    }
  }
}

// loader layout [H,W,C,N] fp32 -> [N,H,W,C] bf16 with the same normalisation (64x64 LDS transpose, as hwcn_to_nhwc_bf16)
__global__ __launch_bounds__(256) void normalize_hwcn_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long hwc, int n, int C,
                                                              float scale, float m0, float m1, float m2, float m3, float s0, float s1,
                                                              float s2, float s3) {
  __shared__ float tile[64][65];
  const float mean[4] = {m0, m1, m2, m3}, istd[4] = {s0, s1, s2, s3};
  const long p0 = (long)blockIdx.x * 64;
  const int n0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int pp = ty; pp < 64; pp += 4) {
    const long pos = p0 + pp;
    const int nn = n0 + tx;
    float v = 0.f;
    if (pos < hwc && nn < n) {
      const int c = (int)(pos % C);
      v = (src[pos * n + nn] * scale - mean[c]) * istd[c];
    }
    tile[pp][tx] = v;
  }
  __syncthreads();
  for (int nn = ty; nn < 64; nn += 4) {
    const long pos = p0 + tx;
    const int ni = n0 + nn;
    if (pos < hwc && ni < n) dst[(size_t)ni * hwc + pos] = f32_to_bf16(tile[tx][nn]);
  }
}

// out[b] = x[b]*w[b] + x[index[b]]*(1-w[b])   (bf16 in, fp32 math, bf16 out); per = elements per image, % 8 == 0
__global__ __launch_bounds__(256) void mixup_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, const float* __restrict__ w,
                                                     const int* __restrict__ index, long per) {
  const int b = blockIdx.y;
  const float wb = w[b], wo = 1.0f - wb;
  const bf16_t* xa = x + (size_t)b * per;
  const bf16_t* xb = x + (size_t)index[b] * per;
  bf16_t* o = out + (size_t)b * per;
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < per; i += (long)gridDim.x * blockDim.x * 8) {
    const uint4 a = *reinterpret_cast<const uint4*>(xa + i);
    const uint4 c = *reinterpret_cast<const uint4*>(xb + i);
    const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, cw[4] = {c.x, c.y, c.z, c.w};
    uint32_t r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x2 va = unpack_bf16x2(aw[k]), vc = unpack_bf16x2(cw[k]);
      r[k] = pack_bf16x2(va.x * wb + vc.x * wo, va.y * wb + vc.y * wo);
    }
    *reinterpret_cast<uint4*>(o + i) = make_uint4(r[0], r[1], r[2], r[3]);
  }
}

// out[b,y,x,:] = inside box[b] ? x[b,y,x,:] : x[index[b],y,x,:]   box = (y0, y1, x0, x1), half-open; one thread per pixel group
__global__ __launch_bounds__(256) void cutmix_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, const int* __restrict__ box,
                                                      const int* __restrict__ index, int H, int W, int C) {
  const int b = blockIdx.y;
  const int y0 = box[4 * b], y1 = box[4 * b + 1], x0 = box[4 * b + 2], x1 = box[4 * b + 3];
  const size_t per = (size_t)H * W * C;
  const bf16_t* xa = x + (size_t)b * per;
  const bf16_t* xb = x + (size_t)index[b] * per;
  bf16_t* o = out + (size_t)b * per;
  const long npix = (long)H * W;
  for (long pix = (long)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (long)gridDim.x * blockDim.x) {
    const int yy = (int)(pix / W), xx = (int)(pix - (long)yy * W);
    const bf16_t* s = (yy >= y0 && yy < y1 && xx >= x0 && xx < x1) ? xa : xb;
    for (int c = 0; c < C; ++c) o[pix * C + c] = s[pix * C + c];
  }
}

// patches[b * g * g + ph * g + pw][ (y * P + x) * 3 + c ] = images[b][ph * P + y][pw * P + x][c]   (patch_embed.py:19-22 on NHWC bf16 images):
// the dense [B * n, P * P * 3] operand of the patch-embed WEIGHT gradient, so that it can join the tile FIFO of the grouped launches
// (round 5; the forward product and its input-gradient-free backward keep gathering from the images).  A thread moves one 16-byte
// chunk: a patch row is P * 3 contiguous elements in the image (P % 8 == 0: whole chunks), the patch matrix is written contiguously.
// Patch p of image b goes to row b * tokens + token_offset + p (the token's row in the [B * tokens, d] activations, so the matrix lines up
// row by row with a cotangent of the token stream; rows of other tokens - the cls slot - are never written: the caller zero-fills once).
__global__ __launch_bounds__(256) void patchify_bf16_kernel(const bf16_t* __restrict__ img, bf16_t* __restrict__ out, int B, int S, int P, int tokens,
                                                            int token_offset) {
  const int g = S / P, cpr = P * 3 / 8;       // 16-byte chunks per patch row
  const long chunks = (long)B * g * g * P * cpr;
  const long cpp = (long)P * cpr;              // chunks per patch
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < chunks; i += (long)gridDim.x * 256) {
    const int ck = (int)(i % cpr);
    long r = i / cpr;
    const int y = (int)(r % P);
    r /= P;
    const int pw = (int)(r % g), ph = (int)((r / g) % g), b = (int)(r / ((long)g * g));
    const size_t src = (((size_t)b * S + (size_t)ph * P + y) * S + (size_t)pw * P) * 3 + (size_t)ck * 8;
    const long orow = (long)b * tokens + token_offset + (long)ph * g + pw;
    reinterpret_cast<uint4*>(out)[orow * cpp + (long)y * cpr + ck] = *reinterpret_cast<const uint4*>(img + src);
  }
}

}  // namespace

extern "C" int savit_normalize_to_nhwc_bf16(const void* src, int src_format, void* dst, int H, int W, int C, int N, float scale,
                                            const float* mean, const float* std, void* stream) {
  SAVIT_CHECK_ARG(src && dst && mean && std && H > 0 && W > 0 && C > 0 && C <= 4 && N >= 0);
  SAVIT_CHECK_ARG(src_format == SAVIT_SRC_HWCN_F32 || src_format == SAVIT_SRC_NHWC_F32 || src_format == SAVIT_SRC_NHWC_U8);
  if (N == 0) return SAVIT_OK;
  float m[4] = {0, 0, 0, 0}, is[4] = {1, 1, 1, 1};
  for (int c = 0; c < C; ++c) {
    SAVIT_CHECK_ARG(std[c] > 0.f);
    m[c] = mean[c];
    is[c] = 1.0f / std[c];
  }
  const long hwc = (long)H * W * C, n = hwc * N;
  hipStream_t s = (hipStream_t)stream;
  if (src_format == SAVIT_SRC_HWCN_F32) {
    hipLaunchKernelGGL(normalize_hwcn_kernel, dim3((unsigned)((hwc + 63) / 64), (N + 63) / 64), dim3(256), 0, s, (const float*)src, (bf16_t*)dst,
                       hwc, N, C, scale, m[0], m[1], m[2], m[3], is[0], is[1], is[2], is[3]);
  } else {
    long blocks = (n / 8 + 255) / 256;
    if (blocks > 65535L * 16) blocks = 65535L * 16;
    if (blocks < 1) blocks = 1;
    if (src_format == SAVIT_SRC_NHWC_F32)
      hipLaunchKernelGGL(normalize_nhwc_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)src, (bf16_t*)dst, n, C, scale, m[0],
                         m[1], m[2], m[3], is[0], is[1], is[2], is[3]);
    else
      hipLaunchKernelGGL(normalize_nhwc_kernel<unsigned char>, dim3((unsigned)blocks), dim3(256), 0, s, (const unsigned char*)src, (bf16_t*)dst, n,
                         C, scale, m[0], m[1], m[2], m[3], is[0], is[1], is[2], is[3]);
  }
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_batch_mixup_bf16(const void* x, void* out, const float* weight, const int* index, int B, long elems_per_image,
                                      void* stream) {
  SAVIT_CHECK_ARG(x && out && weight && index && x != out && B >= 0 && elems_per_image > 0 && elems_per_image % 8 == 0);
  SAVIT_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)out % 16) == 0);
  if (B == 0) return SAVIT_OK;
  long bx = (elems_per_image / 8 + 255) / 256;
  if (bx > 4096) bx = 4096;
  hipLaunchKernelGGL(mixup_kernel, dim3((unsigned)bx, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)out, weight, index,
                     elems_per_image);
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_batch_cutmix_bf16(const void* x, void* out, const int* box, const int* index, int B, int H, int W, int C, void* stream) {
  SAVIT_CHECK_ARG(x && out && box && index && x != out && B >= 0 && H > 0 && W > 0 && C > 0);
  if (B == 0) return SAVIT_OK;
  long bx = ((long)H * W + 255) / 256;
  if (bx > 4096) bx = 4096;
  hipLaunchKernelGGL(cutmix_kernel, dim3((unsigned)bx, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)out, box, index, H, W, C);
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_patchify_bf16(const void* images, void* patches, int B, int img_size, int patch, int tokens, int token_offset, void* stream) {
  SAVIT_CHECK_ARG(images && patches && B >= 0 && patch > 0 && patch % 8 == 0 && img_size > 0 && img_size % patch == 0);
  SAVIT_CHECK_ARG(token_offset >= 0 && tokens >= token_offset + (img_size / patch) * (img_size / patch));
  SAVIT_CHECK_ARG(((uintptr_t)images % 16) == 0 && ((uintptr_t)patches % 16) == 0);
  if (B == 0) return SAVIT_OK;
  const long chunks = (long)B * img_size * img_size * 3 / 8;
  long blocks = (chunks + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(patchify_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)images, (bf16_t*)patches, B,
                     img_size, patch, tokens, token_offset);
  SAVIT_LAUNCH_RET();
}
