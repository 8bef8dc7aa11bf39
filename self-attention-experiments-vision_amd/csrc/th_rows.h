// Row operations of talking-heads attention shared by the materialising kernels (attention.hip) and the fused ones (th_fused.hip):
// head mix -> softmax over the keys of one (image, query) row held by ONE wave (lane = 4 consecutive keys, every head), and the
// 64-partial reduce-scatter the backward uses for the dT1 / dT2 sums.  attention.py:44-52, talking_heads.py:13 of the reference.
#pragma once
#include "common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;

// ---- head mixing + softmax rows: one wave per (b, q) row, all H heads, keys strided over the lanes (<= 4 per lane)
constexpr int TH_KPL = 4;  // keys per lane: Np <= 256
template <int H>
__device__ __forceinline__ void th_row_forward(const float (&s)[H][TH_KPL], const float* T1, int N, int lane, float (&pr)[H][TH_KPL]) {
#pragma unroll
  for (int i = 0; i < H; ++i) {
    float sp[TH_KPL];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < TH_KPL; ++k) {
      float a = 0.f;
#pragma unroll
      for (int h = 0; h < H; ++h) a += T1[h * H + i] * s[h][k];
      sp[k] = (4 * lane + k < N) ? a : -INFINITY;
      m = fmaxf(m, sp[k]);
    }
    m = wave_max(m);
    float l = 0.f;
#pragma unroll
    for (int k = 0; k < TH_KPL; ++k) {
      sp[k] = __builtin_amdgcn_exp2f((sp[k] - m) * LOG2E);
      l += sp[k];
    }
    l = wave_sum(l);
    const float inv = 1.0f / l;
#pragma unroll
    for (int k = 0; k < TH_KPL; ++k) pr[i][k] = sp[k] * inv;
  }
}

// Eight wave-wide reductions at once (one value per head), as a transposing butterfly: three halving exchanges (lanes 32, 16, 8 apart)
// leave each lane with ONE head's partial - head (lane >> 3) & 7 - three DPP steps finish it inside the 8-lane group, and eight
// v_readlane broadcast the results as scalars.  26 instructions instead of 8 x 12-14 for eight separate wave reductions.
// MAX = false: sums; MAX = true: maxima.  out[h] is wave-uniform.
template <bool MAX>
__device__ __forceinline__ void wave_reduce8(const float (&v)[8], float (&out)[8], int lane) {
  auto op = [](float a, float b) { return MAX ? fmaxf(a, b) : a + b; };
  float w[4], u[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) {  // lanes 0-31: head j, lanes 32-63: head j + 4
    float a = v[j], b = v[j + 4];
    permlane32_swap(a, b);
    w[j] = op(a, b);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {  // 16-lane row r: head j + 2 (r & 1) + 4 (r >> 1)
    float a = w[j], b = w[j + 2];
    permlane16_swap(a, b);
    u[j] = op(a, b);
  }
  const bool b3 = (lane & 8) != 0;
  const float keep = b3 ? u[1] : u[0], send = b3 ? u[0] : u[1];
  float t = op(keep, dpp_mov<0x128>(send));  // row_ror:8 - the partner 8 lanes away inside the row
  t = op(t, dpp_mov<0xB1>(t));               // quad_perm [1,0,3,2]
  t = op(t, dpp_mov<0x4E>(t));               // quad_perm [2,3,0,1]
  t = op(t, dpp_mov<0x141>(t));              // row_half_mirror: the other quad of the 8-lane group
#pragma unroll
  for (int h = 0; h < 8; ++h) out[h] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t), 8 * h));
}

__device__ __forceinline__ float th_unpack(uint32_t w, int odd) { return odd ? __uint_as_float(w & 0xffff0000u) : __uint_as_float(w << 16); }

// Reduce-scatter of 64 per-lane partials over the 64 lanes: after 6 halving exchanges lane l holds sum_lanes g[l].
// 63 shuffles instead of 64 full wave reductions, and the H x H dT partials need not persist in registers across rows.
// The two widest exchanges (lanes 32 and 16 apart: 48 of the 63) are gfx950's v_permlane32_swap / v_permlane16_swap: swapping
// the upper half (odd rows) of g[j] with the lower half (even rows) of g[j + n2] leaves exactly "kept + received" in the two
// registers, so one swap + one add replaces two selects, a ds_bpermute and an add (126 bpermutes per row kept the LDS pipe busy
// for a third of this kernel).  The swaps are the builtins of common.h (hazards placed by hipcc).
__device__ __forceinline__ float reduce_scatter64(float (&g)[64], int lane) {
#pragma unroll
  for (int j = 0; j < 32; j += 4) {
#pragma unroll
    for (int k = 0; k < 4; ++k) permlane32_swap(g[j + k], g[j + k + 32]);
#pragma unroll
    for (int k = 0; k < 4; ++k) g[j + k] += g[j + k + 32];
  }
#pragma unroll
  for (int j = 0; j < 16; j += 4) {
#pragma unroll
    for (int k = 0; k < 4; ++k) permlane16_swap(g[j + k], g[j + k + 16]);
#pragma unroll
    for (int k = 0; k < 4; ++k) g[j + k] += g[j + k + 16];
  }
#pragma unroll
  for (int st = 2; st < 6; ++st) {
    const int mask = 32 >> st, n2 = 32 >> st;
    const bool upper = (lane & mask) != 0;
#pragma unroll
    for (int j = 0; j < n2; ++j) {
      const float lo = g[j], hi = g[j + n2];
      const float send = upper ? lo : hi, keep = upper ? hi : lo;
      g[j] = keep + __shfl_xor(send, mask, 64);
    }
  }
  return g[0];
}


// one wave per output element: rows of the slab strided over the lanes, wave reduction, one add
__global__ __launch_bounds__(256) void th_dT_finalize_kernel(const float* __restrict__ slab, int nblk, int hh2, float* __restrict__ dT1,
                                                              float* __restrict__ dT2) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= 2 * hh2) return;
  const int lane = threadIdx.x & 63;
  float a = 0.f;
  for (int r = lane; r < nblk; r += 64) a += slab[(size_t)r * 2 * hh2 + i];
  a = wave_sum(a);
  if (lane == 0) {
    if (i < hh2) dT1[i] += a; else dT2[i - hh2] += a;
  }
}

}  // namespace
