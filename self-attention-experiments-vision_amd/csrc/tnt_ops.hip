// TNT glue around the shared GEMM / LayerNorm / attention kernels (/root/reference/models/tnt.py).
//   * PixelEmbedBlock's two rearranges (tnt.py:21-29): every 16x16 patch becomes a sequence of 4x4 = 16 "pixel tokens" of
//     c*t1*t2 = 48 features (channel slowest) - a gather from the NHWC image into the row-major bf16 operand of the pixel Dense;
//   * AddAbsPosEmbed on the pixel stream (tnt.py:170): x[r, :] += pos[r mod 16, :];
//   * Inner2OuterBlock's tail (tnt.py:48-50): the projected pixel sequences get a zero row in front (the cls slot) and are added
//     to the patch stream; backward splits the LayerNorm-input cotangent into the patch-stream part and the bf16 rows of the
//     projection's cotangent;
//   * the head reads the cls row without a LayerNorm (tnt.py:187-193): row gather to bf16 / scatter of its cotangent.
// All HBM-bound elementwise / gather passes, fp32 arithmetic, one rounding where the reference's bf16 graph materialises a tensor.
#include "common.h"
#include "savit.h"

namespace {

// out[(b*g*g + ph*g + pw)*s*s + p1*s + p2][c*t*t + t1*t + t2] = img[b][ph*P + p1*t + t1][pw*P + p2*t + t2][c];  out row pitch ld_out
__global__ __launch_bounds__(256) void pixel_gather_kernel(const bf16_t* __restrict__ img, bf16_t* __restrict__ out, int B, int S, int P, int t,
                                                            int C, int ld_out, long total) {
  const int g = S / P, s = P / t, F = C * t * t;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int f = (int)(i % F);
    long row = i / F;
    const int p2 = (int)(row % s); row /= s;
    const int p1 = (int)(row % s); row /= s;
    const int pw = (int)(row % g); row /= g;
    const int ph = (int)(row % g);
    const int b = (int)(row / g);
    const int c = f / (t * t), t1 = (f / t) % t, t2 = f % t;
    const int y = ph * P + p1 * t + t1, x = pw * P + p2 * t + t2;
    out[(i / F) * ld_out + f] = img[(((size_t)b * S + y) * S + x) * C + c];
  }
}

__global__ __launch_bounds__(256) void add_rows_periodic_kernel(float* __restrict__ x, const float* __restrict__ pos, long rows, int period, int d) {
  const int q = d / 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < rows * q; i += (long)gridDim.x * blockDim.x) {
    const long r = i / q;
    const int c = (int)(i - r * q);
    float4 v = reinterpret_cast<float4*>(x + r * d)[c];
    const float4 pp = reinterpret_cast<const float4*>(pos + (size_t)(r % period) * d)[c];
    v.x += pp.x; v.y += pp.y; v.z += pp.z; v.w += pp.w;
    reinterpret_cast<float4*>(x + r * d)[c] = v;
  }
}

// out[b, 0, :] = patch[b, 0, :];  out[b, 1 + p, :] = patch[b, 1 + p, :] + y[b*(N-1) + p, :]
__global__ __launch_bounds__(256) void inner2outer_add_kernel(const float* __restrict__ patch, const bf16_t* __restrict__ y, float* __restrict__ out,
                                                               int B, int N, int d) {
  const int q = d / 4;
  const long total = (long)B * N * q;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / q;
    const int c = (int)(i - r * q);
    const int b = (int)(r / N), tkn = (int)(r - (long)b * N);
    float4 v = reinterpret_cast<const float4*>(patch + r * d)[c];
    if (tkn > 0) {
      const uint2 w = reinterpret_cast<const uint2*>(y + ((size_t)b * (N - 1) + tkn - 1) * d)[c];
      v.x += __uint_as_float(w.x << 16); v.y += __uint_as_float(w.x & 0xffff0000u);
      v.z += __uint_as_float(w.y << 16); v.w += __uint_as_float(w.y & 0xffff0000u);
    }
    reinterpret_cast<float4*>(out + r * d)[c] = v;
  }
}

// Column sums ride along in registers: blockDim.x is a multiple of q = d/4, so are the grid strides, hence a thread keeps ONE
// group of 4 columns for all its rows; partials meet in LDS, then one atomic per (block, column).
__device__ __forceinline__ void block_colsum_flush(float4 acc, int c, int q, float* lds, float* __restrict__ out) {
  for (int i = threadIdx.x; i < 4 * q; i += blockDim.x) lds[i] = 0.f;
  __syncthreads();
  atomicAdd(lds + 4 * c, acc.x);
  atomicAdd(lds + 4 * c + 1, acc.y);
  atomicAdd(lds + 4 * c + 2, acc.z);
  atomicAdd(lds + 4 * c + 3, acc.w);
  __syncthreads();
  for (int i = threadIdx.x; i < 4 * q; i += blockDim.x) atomicAdd(out + i, lds[i]);
}

// dres += dt (every row);  dy[b*(N-1) + p, :] = bf16(dt[b, 1 + p, :]);  dbias[:] += sum over b, p of those bf16 rows
__global__ __launch_bounds__(256) void inner2outer_split_kernel(const float* __restrict__ dt, float* __restrict__ dres, bf16_t* __restrict__ dy,
                                                                 float* __restrict__ dbias, int B, int N, int d) {
  __shared__ float lds[1024];
  const int q = d / 4;
  const long total = (long)B * N * q;
  const int c = threadIdx.x % q;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / q;
    const int b = (int)(r / N), tkn = (int)(r - (long)b * N);
    const float4 v = reinterpret_cast<const float4*>(dt + r * d)[c];
    float4 a = reinterpret_cast<float4*>(dres + r * d)[c];
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    reinterpret_cast<float4*>(dres + r * d)[c] = a;
    if (tkn > 0) {
      const uint2 w = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
      reinterpret_cast<uint2*>(dy + ((size_t)b * (N - 1) + tkn - 1) * d)[c] = w;
      acc.x += __uint_as_float(w.x << 16); acc.y += __uint_as_float(w.x & 0xffff0000u);
      acc.z += __uint_as_float(w.y << 16); acc.w += __uint_as_float(w.y & 0xffff0000u);
    }
  }
  if (dbias != nullptr) block_colsum_flush(acc, c, q, lds, dbias);
}

// dst = bf16(src) row for row;  colsum[:] += sum over rows of src (fp32, before rounding)
__global__ __launch_bounds__(256) void cast_colsum_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, float* __restrict__ colsum, long rows,
                                                           int d) {
  __shared__ float lds[1024];
  const int q = d / 4;
  const long total = rows * q;
  const int c = threadIdx.x % q;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(src)[i];
    if (dst != nullptr) reinterpret_cast<uint2*>(dst)[i] = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  if (colsum != nullptr) block_colsum_flush(acc, c, q, lds, colsum);
}

// dst[b, :] = bf16(src[b * row_stride + :])
__global__ __launch_bounds__(256) void gather_rows_bf16_kernel(const float* __restrict__ src, long row_stride, bf16_t* __restrict__ dst, int B, int d) {
  const int q = d / 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)B * q; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / q), c = (int)(i - (long)b * q);
    const float4 v = reinterpret_cast<const float4*>(src + (size_t)b * row_stride)[c];
    reinterpret_cast<uint2*>(dst + (size_t)b * d)[c] = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
  }
}

// dst_f32[b * row_stride + :] = src_bf16[b, :] (optional bf16 copy with the same row mapping)
__global__ __launch_bounds__(256) void scatter_rows_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, bf16_t* __restrict__ dst_b,
                                                            long row_stride, int B, int d) {
  const int q = d / 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)B * q; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / q), c = (int)(i - (long)b * q);
    const uint2 w = reinterpret_cast<const uint2*>(src + (size_t)b * d)[c];
    reinterpret_cast<float4*>(dst + (size_t)b * row_stride)[c] =
        make_float4(__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16), __uint_as_float(w.y & 0xffff0000u));
    if (dst_b != nullptr) reinterpret_cast<uint2*>(dst_b + (size_t)b * row_stride)[c] = w;
  }
}

inline int grid_for(long work) {
  long g = (work + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

}  // namespace

extern "C" int savit_tnt_pixel_gather(const void* images_bf16, void* out_bf16, int B, int img_size, int patch, int t, int C, int ld_out,
                                      void* stream) {
  SAVIT_CHECK_ARG(images_bf16 && out_bf16 && B >= 0 && img_size > 0 && patch > 0 && t > 0 && C > 0 && img_size % patch == 0 && patch % t == 0 &&
                  ld_out >= C * t * t);
  if (B == 0) return SAVIT_OK;
  const long total = (long)B * img_size * img_size * C;
  hipLaunchKernelGGL(pixel_gather_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)images_bf16, (bf16_t*)out_bf16, B,
                     img_size, patch, t, C, ld_out, total);
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_add_rows_periodic(float* x, const float* pos, long rows, int period, int d, void* stream) {
  SAVIT_CHECK_ARG(x && pos && rows >= 0 && period > 0 && d > 0 && (d % 4) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)pos % 16) == 0);
  if (rows == 0) return SAVIT_OK;
  hipLaunchKernelGGL(add_rows_periodic_kernel, dim3(grid_for(rows * (d / 4))), dim3(256), 0, (hipStream_t)stream, x, pos, rows, period, d);
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_tnt_inner2outer_add(const float* patch, const void* y_bf16, float* out, int B, int N, int d, void* stream) {
  SAVIT_CHECK_ARG(patch && y_bf16 && out && B >= 0 && N > 1 && d > 0 && (d % 4) == 0 && ((uintptr_t)patch % 16) == 0 && ((uintptr_t)y_bf16 % 8) == 0 &&
                  ((uintptr_t)out % 16) == 0);
  if (B == 0) return SAVIT_OK;
  hipLaunchKernelGGL(inner2outer_add_kernel, dim3(grid_for((long)B * N * (d / 4))), dim3(256), 0, (hipStream_t)stream, patch, (const bf16_t*)y_bf16, out,
                     B, N, d);
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_tnt_inner2outer_split(const float* dt, float* dres, void* dy_bf16, float* dbias, int B, int N, int d, void* stream) {
  SAVIT_CHECK_ARG(dt && dres && dy_bf16 && B >= 0 && N > 1 && d > 0 && (d % 4) == 0 && d <= 1024 && ((uintptr_t)dt % 16) == 0 &&
                  ((uintptr_t)dres % 16) == 0 && ((uintptr_t)dy_bf16 % 8) == 0);
  if (B == 0) return SAVIT_OK;
  const int q = d / 4, threads = (256 / q) * q;  // a multiple of q: every thread keeps one column group (see block_colsum_flush)
  long blocks = ((long)B * N * q + threads - 1) / threads;
  if (blocks > 512) blocks = 512;  // each block ends with d atomics onto the same d addresses: keep them few
  hipLaunchKernelGGL(inner2outer_split_kernel, dim3((unsigned)blocks), dim3(threads), 0, (hipStream_t)stream, dt, dres, (bf16_t*)dy_bf16, dbias, B, N,
                     d);
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_cast_colsum(const float* src, void* dst_bf16, float* colsum, long rows, int d, void* stream) {
  SAVIT_CHECK_ARG(src && (dst_bf16 || colsum) && rows >= 0 && d > 0 && (d % 4) == 0 && d <= 1024 && ((uintptr_t)src % 16) == 0 &&
                  ((uintptr_t)dst_bf16 % 8) == 0);
  if (rows == 0) return SAVIT_OK;
  const int q = d / 4, threads = (256 / q) * q;
  long blocks = (rows * q + threads - 1) / threads;
  if (blocks > 512) blocks = 512;  // each block ends with d atomics onto the same d addresses: keep them few
  hipLaunchKernelGGL(cast_colsum_kernel, dim3((unsigned)blocks), dim3(threads), 0, (hipStream_t)stream, src, (bf16_t*)dst_bf16, colsum, rows, d);
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_gather_rows_bf16(const float* src, long row_stride, void* dst_bf16, int B, int d, void* stream) {
  SAVIT_CHECK_ARG(src && dst_bf16 && B >= 0 && d > 0 && (d % 4) == 0 && row_stride >= d && (row_stride % 4) == 0 && ((uintptr_t)src % 16) == 0 &&
                  ((uintptr_t)dst_bf16 % 8) == 0);
  if (B == 0) return SAVIT_OK;
  hipLaunchKernelGGL(gather_rows_bf16_kernel, dim3(grid_for((long)B * (d / 4))), dim3(256), 0, (hipStream_t)stream, src, row_stride, (bf16_t*)dst_bf16, B,
                     d);
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_scatter_rows(const void* src_bf16, float* dst, void* dst_bf16, long row_stride, int B, int d, void* stream) {
  SAVIT_CHECK_ARG(src_bf16 && dst && B >= 0 && d > 0 && (d % 4) == 0 && row_stride >= d && (row_stride % 4) == 0 && ((uintptr_t)dst % 16) == 0 &&
                  ((uintptr_t)src_bf16 % 8) == 0 && ((uintptr_t)dst_bf16 % 8) == 0);
  if (B == 0) return SAVIT_OK;
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid_for((long)B * (d / 4))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src_bf16, dst,
                     (bf16_t*)dst_bf16, row_stride, B, d);
  SAVIT_LAUNCH_RET();
}
