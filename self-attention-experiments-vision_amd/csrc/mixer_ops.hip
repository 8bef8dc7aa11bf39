// MLP-Mixer glue around the shared GEMM / LayerNorm kernels (/root/reference/models/mlp_mixer.py:17-31,61-63).
//   * token mixing runs FFBlock over the TOKEN axis: rearrange '... l d -> ... d l' before, '... d l -> ... l d' after
//     (mlp_mixer.py:19,23).  The GEMMs want that axis contiguous, so the activation is transposed per image through LDS:
//     one pass, 2 B read + 2 B written per element, whole 128-B lines on both sides.  The same kernel carries the epilogue
//     of the block (`x = x + inputs`, mlp_mixer.py:24) when it transposes back, and the row sums backward needs for the
//     bias of the second token Dense (d bias[l] = sum over images and channels of the cotangent).
//   * `jnp.mean(x, axis=1)` over tokens before the head (mlp_mixer.py:62) and its backward (a broadcast of dz / L).
// All HBM-bound; fp32 arithmetic, one rounding to bf16 where the reference's bf16 graph materialises a tensor.
#include "common.h"
#include "savit.h"

namespace {

constexpr int TT = 64;        // tile side
constexpr int TPITCH = TT + 2;  // bf16 elements per LDS row: 33 dwords, so the column reads below spread over all banks

// src: B matrices [R, Cc] bf16 (row pitch ld_src) -> their transposes [Cc, R] (row pitch ld_dst):
//   dst_bf16[b][c][r] = src[b][r][c]                                   (if dst_bf16)
//   out_f32[b][c][r]  = (round_bf16)(resid[b][c][r] + src[b][r][c])    (if out_f32; resid / out_f32 fp32 with pitch ld_dst)
//   rowsum[b * gridDim.x + tile_c][r] = sum over the tile's 64 columns of src[b][r][c]   (if rowsum: a slab of partial sums with
//     pitch rowsum_ld, reduced by savit_colsum_finalize; atomics on the R result addresses cost 60 us per launch at B*d/64 = 1536
//     adds per address)
__device__ __forceinline__ void transpose_tile(bf16_t (*tile)[TPITCH], const bf16_t* __restrict__ src, long src_bs, int ld_src,
                                               bf16_t* __restrict__ dst, long dst_bs, int ld_dst, int R, int Cc,
                                               const float* __restrict__ resid, float* __restrict__ out_f32, int round_out,
                                               float* __restrict__ rowsum, int rowsum_ld, int b, int ty, int tx, int ntx) {
  const int r0 = ty * TT, c0 = tx * TT;
  const bf16_t* s = src + (size_t)b * src_bs;
  const int t = threadIdx.x;
  // ---- load: 8 lanes x 16 B per row, 32 rows per pass
  {
    const int cu = (t & 7) * 8;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int rr = pass * 32 + (t >> 3);
      const int r = r0 + rr, c = c0 + cu;
      uint32_t w[4] = {0u, 0u, 0u, 0u};
      if (r < R) {
        if (c + 8 <= Cc) {
          const uint4 v = *reinterpret_cast<const uint4*>(s + (size_t)r * ld_src + c);
          w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        } else {
          for (int k = 0; k < 8; ++k)
            if (c + k < Cc) w[k >> 1] |= (uint32_t)s[(size_t)r * ld_src + c + k] << ((k & 1) * 16);
        }
      }
      uint32_t* trow = reinterpret_cast<uint32_t*>(&tile[rr][cu]);  // 4-B aligned: TPITCH and cu are even
      trow[0] = w[0]; trow[1] = w[1]; trow[2] = w[2]; trow[3] = w[3];
      if (rowsum != nullptr) {
        float ps = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) ps += __uint_as_float(w[k] << 16) + __uint_as_float(w[k] & 0xffff0000u);
        // the 8 lanes of a row are consecutive: fold them with quad + half-row DPP steps
        ps += dpp_mov<0xB1>(ps);
        ps += dpp_mov<0x4E>(ps);
        ps += dpp_mov<0x141>(ps);
        if ((t & 7) == 0 && r < R) rowsum[((size_t)b * ntx + tx) * rowsum_ld + r] = ps;
      }
    }
  }
  __syncthreads();
  // ---- store: destination row c (a source column), 8 consecutive r per lane
  {
    const int ru = (t & 7) * 8;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int cc = pass * 32 + (t >> 3);
      const int c = c0 + cc, r = r0 + ru;
      if (c >= Cc || r >= R) continue;
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = bf16_to_f32(tile[ru + k][cc]);
      const int lim = (R - r) < 8 ? (R - r) : 8;
      if (dst != nullptr) {
        bf16_t* d = dst + (size_t)b * dst_bs + (size_t)c * ld_dst + r;
        if (lim == 8) {
          *reinterpret_cast<uint4*>(d) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
        } else {
          for (int k = 0; k < lim; ++k) d[k] = f32_to_bf16(v[k]);
        }
      }
      if (out_f32 != nullptr) {
        const size_t o = ((size_t)b * Cc + c) * ld_dst + r;  // fp32 rows are packed per image: batch stride = Cc * ld_dst
        if (lim == 8) {
          const float4 a0 = *reinterpret_cast<const float4*>(resid + o), a1 = *reinterpret_cast<const float4*>(resid + o + 4);
          float w[8] = {a0.x + v[0], a0.y + v[1], a0.z + v[2], a0.w + v[3], a1.x + v[4], a1.y + v[5], a1.z + v[6], a1.w + v[7]};
          if (round_out) {
#pragma unroll
            for (int k = 0; k < 8; ++k) w[k] = round_bf16(w[k]);
          }
          *reinterpret_cast<float4*>(out_f32 + o) = make_float4(w[0], w[1], w[2], w[3]);
          *reinterpret_cast<float4*>(out_f32 + o + 4) = make_float4(w[4], w[5], w[6], w[7]);
        } else {
          for (int k = 0; k < lim; ++k) {
            const float w = resid[o + k] + v[k];
            out_f32[o + k] = round_out ? round_bf16(w) : w;
          }
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ src, long src_bs, int ld_src, bf16_t* __restrict__ dst,
                                                              long dst_bs, int ld_dst, int R, int Cc, const float* __restrict__ resid,
                                                              float* __restrict__ out_f32, int round_out, float* __restrict__ rowsum,
                                                              int rowsum_ld) {
  __shared__ bf16_t tile[TT][TPITCH];
  transpose_tile(tile, src, src_bs, ld_src, dst, dst_bs, ld_dst, R, Cc, resid, out_f32, round_out, rowsum, rowsum_ld, blockIdx.z, blockIdx.y,
                 blockIdx.x, gridDim.x);
}

// several transposes in ONE launch (round 5: the [out, in] operand forms of a ViT's four weight families and its patch embedding are
// refreshed from the bf16 mirror after every optimizer step - five launches of 5-26 us each with a kernel boundary between them)
constexpr int TJ_MAX = 8;
struct TransposeJobs {
  const bf16_t* src[TJ_MAX];
  bf16_t* dst[TJ_MAX];
  long src_bs[TJ_MAX], dst_bs[TJ_MAX];
  int ld_src[TJ_MAX], ld_dst[TJ_MAX], R[TJ_MAX], Cc[TJ_MAX], ntx[TJ_MAX], nty[TJ_MAX];
  unsigned first[TJ_MAX + 1];  // first workgroup of job j; first[count] = grid size
  int count;
};
__global__ __launch_bounds__(256) void transpose_bf16_jobs_kernel(const TransposeJobs J) {
  __shared__ bf16_t tile[TT][TPITCH];
  int j = 0;
#pragma unroll
  for (int k = 1; k < TJ_MAX; ++k)
    if (k < J.count && blockIdx.x >= J.first[k]) j = k;
  unsigned w = blockIdx.x - J.first[j];
  const int tx = (int)(w % (unsigned)J.ntx[j]);
  w /= (unsigned)J.ntx[j];
  const int ty = (int)(w % (unsigned)J.nty[j]);
  const int b = (int)(w / (unsigned)J.nty[j]);
  transpose_tile(tile, J.src[j], J.src_bs[j], J.ld_src[j], J.dst[j], J.dst_bs[j], J.ld_dst[j], J.R[j], J.Cc[j], nullptr, nullptr, 0, nullptr, 0, b,
                 ty, tx, J.ntx[j]);
}

// z[b][c] = bf16( (1/L) sum_l h[b][l][c] ): one workgroup per (image, 128 channels); the 4 waves split the tokens
__global__ __launch_bounds__(256) void token_mean_fwd_kernel(const bf16_t* __restrict__ h, bf16_t* __restrict__ z, int L, int d, float inv) {
  __shared__ float part[4][128];
  const int b = blockIdx.y, c = blockIdx.x * 128 + (threadIdx.x & 63) * 2, w = threadIdx.x >> 6;
  float s0 = 0.f, s1 = 0.f;
  if (c < d) {
    const bf16_t* p = h + (size_t)b * L * d + c;
    for (int l = w; l < L; l += 4) {
      const uint32_t v = *reinterpret_cast<const uint32_t*>(p + (size_t)l * d);
      s0 += __uint_as_float(v << 16);
      s1 += __uint_as_float(v & 0xffff0000u);
    }
  }
  part[w][(threadIdx.x & 63) * 2] = s0;
  part[w][(threadIdx.x & 63) * 2 + 1] = s1;
  __syncthreads();
  if (w == 0 && c < d) {
    const int i = (threadIdx.x & 63) * 2;
    const float a0 = part[0][i] + part[1][i] + part[2][i] + part[3][i];
    const float a1 = part[0][i + 1] + part[1][i + 1] + part[2][i + 1] + part[3][i + 1];
    *reinterpret_cast<uint32_t*>(z + (size_t)b * d + c) = pack_bf16x2(a0 * inv, a1 * inv);
  }
}

// dh[b][l][c] = bf16(dz[b][c] / L) for every token l
__global__ __launch_bounds__(256) void token_mean_bwd_kernel(const bf16_t* __restrict__ dz, bf16_t* __restrict__ dh, int L, int d, float inv,
                                                              long units) {
  const int upr = d / 8;  // 16-B units per row
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < units; i += (long)gridDim.x * blockDim.x) {
    const long row = i / upr;
    const int u = (int)(i - row * upr);
    const int b = (int)(row / L);
    const uint4 v = *reinterpret_cast<const uint4*>(dz + (size_t)b * d + u * 8);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = pack_bf16x2(__uint_as_float(w[k] << 16) * inv, __uint_as_float(w[k] & 0xffff0000u) * inv);
    *reinterpret_cast<uint4*>(dh + (size_t)row * d + u * 8) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

}  // namespace

extern "C" int savit_transpose_bf16(const void* src, long src_batch_stride, int ld_src, void* dst_bf16, long dst_batch_stride, int ld_dst,
                                    int B, int R, int Cc, const float* resid, float* out_f32, int round_out_bf16, float* rowsum_slab,
                                    int rowsum_ld, void* stream) {
  SAVIT_CHECK_ARG(src && (dst_bf16 || out_f32) && B >= 0 && R > 0 && Cc > 0 && ld_src >= Cc && ld_dst >= R && (ld_src % 8) == 0 &&
                  (ld_dst % 8) == 0 && (src_batch_stride % 8) == 0 && (dst_batch_stride % 8) == 0 && B <= 65535);
  SAVIT_CHECK_ARG(((uintptr_t)src % 16) == 0 && ((uintptr_t)dst_bf16 % 16) == 0 && ((uintptr_t)resid % 16) == 0 && ((uintptr_t)out_f32 % 16) == 0);
  SAVIT_CHECK_ARG((out_f32 == nullptr) == (resid == nullptr));
  SAVIT_CHECK_ARG(rowsum_slab == nullptr || rowsum_ld >= R);
  if (B == 0) return SAVIT_OK;
  const dim3 grid((Cc + TT - 1) / TT, (R + TT - 1) / TT, B);
  SAVIT_CHECK_ARG(grid.y <= 65535);
  hipLaunchKernelGGL(transpose_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, src_batch_stride, ld_src,
                     (bf16_t*)dst_bf16, dst_batch_stride, ld_dst, R, Cc, resid, out_f32, round_out_bf16, rowsum_slab, rowsum_ld);
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_transpose_bf16_jobs(const savit_transpose_job* jobs, int count, void* stream) {
  SAVIT_CHECK_ARG(count >= 0 && count <= TJ_MAX && (jobs || count == 0));
  if (count == 0) return SAVIT_OK;
  TransposeJobs J;
  unsigned long total = 0;
  int n = 0;
  for (int i = 0; i < count; ++i) {
    const savit_transpose_job& q = jobs[i];
    SAVIT_CHECK_ARG(q.src && q.dst && q.batch >= 0 && q.rows > 0 && q.cols > 0 && q.ld_src >= q.cols && q.ld_dst >= q.rows && (q.ld_src % 8) == 0 &&
                    (q.ld_dst % 8) == 0 && (q.src_batch_stride % 8) == 0 && (q.dst_batch_stride % 8) == 0);
    SAVIT_CHECK_ARG(((uintptr_t)q.src % 16) == 0 && ((uintptr_t)q.dst % 16) == 0);
    if (q.batch == 0) continue;
    J.src[n] = (const bf16_t*)q.src; J.dst[n] = (bf16_t*)q.dst;
    J.src_bs[n] = q.src_batch_stride; J.dst_bs[n] = q.dst_batch_stride;
    J.ld_src[n] = q.ld_src; J.ld_dst[n] = q.ld_dst; J.R[n] = q.rows; J.Cc[n] = q.cols;
    J.ntx[n] = (q.cols + TT - 1) / TT; J.nty[n] = (q.rows + TT - 1) / TT;
    J.first[n] = (unsigned)total;
    total += (unsigned long)J.ntx[n] * J.nty[n] * q.batch;
    SAVIT_CHECK_ARG(total < (1ul << 31));
    ++n;
  }
  if (n == 0) return SAVIT_OK;
  for (int k = n; k <= TJ_MAX; ++k) J.first[k] = (unsigned)total;
  for (int k = n; k < TJ_MAX; ++k) { J.src[k] = nullptr; J.dst[k] = nullptr; J.src_bs[k] = J.dst_bs[k] = 0; J.ld_src[k] = J.ld_dst[k] = J.R[k] = J.Cc[k] = 0; J.ntx[k] = J.nty[k] = 1; }
  J.count = n;
  hipLaunchKernelGGL(transpose_bf16_jobs_kernel, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, J);
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_transpose_rowsum_rows(int B, int Cc) { return (B <= 0 || Cc <= 0) ? 0 : B * ((Cc + TT - 1) / TT); }

extern "C" int savit_token_mean_fwd(const void* h, void* z, int B, int L, int d, void* stream) {
  SAVIT_CHECK_ARG(h && z && B >= 0 && L > 0 && d > 0 && (d % 8) == 0 && B <= 65535);
  if (B == 0) return SAVIT_OK;
  hipLaunchKernelGGL(token_mean_fwd_kernel, dim3((d + 127) / 128, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)h, (bf16_t*)z, L, d,
                     1.0f / (float)L);
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_token_mean_bwd(const void* dz, void* dh, int B, int L, int d, void* stream) {
  SAVIT_CHECK_ARG(dz && dh && B >= 0 && L > 0 && d > 0 && (d % 8) == 0 && ((uintptr_t)dz % 16) == 0 && ((uintptr_t)dh % 16) == 0);
  if (B == 0) return SAVIT_OK;
  const long units = (long)B * L * (d / 8);
  const int grid = (int)((units + 255) / 256 < 8192 ? (units + 255) / 256 : 8192);
  hipLaunchKernelGGL(token_mean_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dz, (bf16_t*)dh, L, d,
                     1.0f / (float)L, units);
  SAVIT_LAUNCH_RET();
}
