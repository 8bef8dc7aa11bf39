// LayerNorm forward for the fp32 residual stream (HBM-bound).
// Semantics: flax nn.LayerNorm(dtype=bf16) as used at /root/reference models/vit.py:19,26,57 and
// models/cait.py:30,42,99,111,176 - fp32 statistics, biased variance E[x^2]-E[x]^2, eps 1e-6,
// scale/bias rounded to bf16 before use, output bf16.
//
// Layout: x fp32 [rows, d] row-major, y bf16 [rows, d].  One wave (64 lanes) owns one row; a lane
// holds d/64 elements as float4 chunks (lane i takes chunks i, i+64, ...), so every wave-level load
// is a coalesced 1 KiB and reductions stay inside the wave.  d % 4 == 0, d <= 4096.
// Algorithmic bytes: (4+2)*rows*d + 8*rows.
//
// Built with -fno-slp-vectorize (csrc/Makefile), and tests/test_abi.py checks that this object holds no v_pk_*_f32 instruction.
// Why: with the SLP vectoriser the normalisation tail becomes an IN-PLACE `v_pk_add_f32 x, x, -mean op_sel:[0,1]` per float4
// half.  When two PROCESSES time-slice one GPU (never with one process per GPU, the deployment model), about one launch in 200
// returned rows in which, for the 16 lanes of one DPP row, the LOW half of exactly that instruction had not been applied: the
// logged wrong values equal (x - 0) * rstd * g + b to bf16 rounding in all 13 recorded events (tools/ln_forensics.py recomputes
// them from the probe's seeded inputs; profiles/r02_ln_forensics.log), in the wide kernel and in the narrow one - which has no
// v_permlane*_swap at all, so the hand-timed swaps round 1 suspected are not involved (they are compiler builtins now anyway:
// common.h).  Statistics, the high half and every other instruction were right.  The mechanism is below the ISA contract (a lost
// VOP3P pass around a wave context switch); the kernels simply do not need packed math (HBM-bound, 18.8 us either way).
#include "layernorm_common.h"

namespace {

template <int CH>  // CH = ceil(d/4/64) chunks per lane
__global__ __launch_bounds__(LN_THREADS) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                             float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                             int rows, int d, long x_stride, float eps, int round_params, int grp,
                                                             int grp_stride, int grp_off) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int nchunk = d >> 2;
  float4 g[CH], b[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int ci = lane + 64 * c;
    if (ci < nchunk) {
      g[c] = reinterpret_cast<const float4*>(gamma)[ci];
      b[c] = reinterpret_cast<const float4*>(beta)[ci];
      if (round_params) {
        g[c] = make_float4(round_bf16(g[c].x), round_bf16(g[c].y), round_bf16(g[c].z), round_bf16(g[c].w));
        b[c] = make_float4(round_bf16(b[c].x), round_bf16(b[c].y), round_bf16(b[c].z), round_bf16(b[c].w));
      }
    } else {
      g[c] = make_float4(0, 0, 0, 0);
      b[c] = make_float4(0, 0, 0, 0);
    }
  }
  const float inv_d = 1.0f / (float)d;
  for (int row = blockIdx.x * LN_WAVES + wave; row < rows; row += gridDim.x * LN_WAVES) {
    const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * x_stride);
    float4 v[CH];
    float s = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int ci = lane + 64 * c;
      v[c] = (ci < nchunk) ? xr[ci] : make_float4(0, 0, 0, 0);
      s += (v[c].x + v[c].y) + (v[c].z + v[c].w);
      s2 += (v[c].x * v[c].x + v[c].y * v[c].y) + (v[c].z * v[c].z + v[c].w * v[c].w);
    }
    s = wave_sum(s);
    s2 = wave_sum(s2);
    const float mean = s * inv_d;
    const float var = fmaxf(s2 * inv_d - mean * mean, 0.f);
    const float rstd = rsqrtf(var + eps);
    if (lane == 0) {
      if (mean_out) mean_out[row] = mean;
      if (rstd_out) rstd_out[row] = rstd;
    }
    const size_t yrow = grp > 0 ? (size_t)(row / grp) * grp_stride + grp_off + (row % grp) : (size_t)row;
    uint2* yr = reinterpret_cast<uint2*>(y + yrow * d);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int ci = lane + 64 * c;
      if (ci < nchunk) {
        float o0 = (v[c].x - mean) * (rstd * g[c].x) + b[c].x;
        float o1 = (v[c].y - mean) * (rstd * g[c].y) + b[c].y;
        float o2 = (v[c].z - mean) * (rstd * g[c].z) + b[c].z;
        float o3 = (v[c].w - mean) * (rstd * g[c].w) + b[c].w;
        yr[ci] = make_uint2(pack_bf16x2(o0, o1), pack_bf16x2(o2, o3));
      }
    }
  }
}

__global__ __launch_bounds__(LN_THREADS) void ln_fwd_narrow_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                                    float* __restrict__ mean_out, float* __restrict__ rstd_out, int rows, int d,
                                                                    long x_stride, float eps, int round_params) {
  const int sub = threadIdx.x >> 4, ci = threadIdx.x & 15;  // 16 row slots per workgroup
  const bool on = ci < (d >> 2);
  float4 g = make_float4(0, 0, 0, 0), b = make_float4(0, 0, 0, 0);
  if (on) {
    g = reinterpret_cast<const float4*>(gamma)[ci];
    b = reinterpret_cast<const float4*>(beta)[ci];
    if (round_params) {
      g = make_float4(round_bf16(g.x), round_bf16(g.y), round_bf16(g.z), round_bf16(g.w));
      b = make_float4(round_bf16(b.x), round_bf16(b.y), round_bf16(b.z), round_bf16(b.w));
    }
  }
  const float inv_d = 1.0f / (float)d;
  const int step = gridDim.x * 16;
  for (int row0 = blockIdx.x * 16; row0 < rows; row0 += step) {  // uniform trip count: the DPP steps need every lane
    const int row = row0 + sub;
    const bool live = on && row < rows;
    float4 v = make_float4(0, 0, 0, 0);
    if (live) v = reinterpret_cast<const float4*>(x + (size_t)row * x_stride)[ci];
    const float s = row16_sum((v.x + v.y) + (v.z + v.w));
    const float s2 = row16_sum((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w));
    const float mean = s * inv_d;
    const float rstd = rsqrtf(fmaxf(s2 * inv_d - mean * mean, 0.f) + eps);
    if (ci == 0 && row < rows) {
      if (mean_out) mean_out[row] = mean;
      if (rstd_out) rstd_out[row] = rstd;
    }
    if (live) {
      const float o0 = (v.x - mean) * (rstd * g.x) + b.x, o1 = (v.y - mean) * (rstd * g.y) + b.y;
      const float o2 = (v.z - mean) * (rstd * g.z) + b.z, o3 = (v.w - mean) * (rstd * g.w) + b.w;
      reinterpret_cast<uint2*>(y + (size_t)row * d)[ci] = make_uint2(pack_bf16x2(o0, o1), pack_bf16x2(o2, o3));
    }
  }
}


}  // namespace

extern "C" int savit_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                   int rows, int d, long x_stride, float eps, int round_params_bf16, void* stream) {
  SAVIT_CHECK_ARG(x && gamma && beta && y && x_stride >= d && (x_stride % 4) == 0 && rows >= 0 && d > 0 && (d % 4) == 0 && d <= 64 * 4 * LN_MAX_CHUNKS);
  if (rows == 0) return SAVIT_OK;
  hipStream_t s = (hipStream_t)stream;
  if (d <= 64) {  // narrow rows: 16 lanes per row, 16 rows per workgroup
    const int g16 = (rows + 15) / 16;
    hipLaunchKernelGGL(ln_fwd_narrow_kernel, dim3(g16 < 256 * 16 ? g16 : 256 * 16), dim3(LN_THREADS), 0, s, x, gamma, beta, (bf16_t*)y, mean, rstd,
                       rows, d, x_stride, eps, round_params_bf16);
    SAVIT_LAUNCH_RET();
  }
  const int ch = (d / 4 + 63) / 64;
  const int grid = ln_grid(rows, 256 * 16);
  LN_DISPATCH(ch, ln_fwd_kernel, grid, x, gamma, beta, (bf16_t*)y, mean, rstd, rows, d, x_stride, eps, round_params_bf16, 0, 0, 0);
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_layernorm_fwd_mapped(const float* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int rows,
                                          int d, long x_stride, float eps, int round_params_bf16, int grp, int grp_stride, int grp_off,
                                          void* stream) {
  SAVIT_CHECK_ARG(x && gamma && beta && y && x_stride >= d && (x_stride % 4) == 0 && rows >= 0 && d > 0 && (d % 4) == 0 && d <= 64 * 4 * LN_MAX_CHUNKS);
  SAVIT_CHECK_ARG(grp > 0 && grp_stride >= grp && grp_off >= 0 && grp_off + grp <= grp_stride);
  if (rows == 0) return SAVIT_OK;
  hipStream_t s = (hipStream_t)stream;
  const int ch = (d / 4 + 63) / 64;
  const int grid = ln_grid(rows, 256 * 16);
  LN_DISPATCH(ch, ln_fwd_kernel, grid, x, gamma, beta, (bf16_t*)y, mean, rstd, rows, d, x_stride, eps, round_params_bf16, grp, grp_stride, grp_off);
  SAVIT_LAUNCH_RET();
}
