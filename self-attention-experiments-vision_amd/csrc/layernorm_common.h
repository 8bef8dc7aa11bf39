// Shared pieces of the LayerNorm translation units (layernorm_fwd.hip, layernorm.hip).
#pragma once
#include "common.h"
#include "savit.h"

namespace {

constexpr int LN_THREADS = 256;
constexpr int LN_WAVES = LN_THREADS / 64;
constexpr int LN_MAX_CHUNKS = 16;  // 16 float4 * 64 lanes = 4096 columns

// ------------------------------------------------------------------------------------------------------------
// Narrow rows (d <= 64: TNT's pixel stream, 24 / 40 channels).  One 64-lane wave per row would keep 6 - 10 lanes busy; here a
// row is owned by one DPP row of 16 lanes (a lane holds one float4), i.e. 4 rows per wave and 16 per workgroup, and the row
// reductions are the four DPP steps that stay inside 16 lanes.  Same arithmetic, statistics and outputs as the kernels above.
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);  // row_half_mirror
  v += dpp_mov<0x140>(v);  // row_mirror
  return v;
}

inline int ln_grid(int rows, int cap) {
  int g = (rows + LN_WAVES - 1) / LN_WAVES;
  return g < cap ? (g < 1 ? 1 : g) : cap;
}


}  // namespace

#define LN_DISPATCH(CHV, KERNEL, GRID, ...)                                                        \
  switch (CHV) {                                                                                    \
    case 1: hipLaunchKernelGGL(KERNEL<1>, dim3(GRID), dim3(LN_THREADS), 0, s, __VA_ARGS__); break;  \
    case 2: hipLaunchKernelGGL(KERNEL<2>, dim3(GRID), dim3(LN_THREADS), 0, s, __VA_ARGS__); break;  \
    case 3: hipLaunchKernelGGL(KERNEL<3>, dim3(GRID), dim3(LN_THREADS), 0, s, __VA_ARGS__); break;  \
    case 4: hipLaunchKernelGGL(KERNEL<4>, dim3(GRID), dim3(LN_THREADS), 0, s, __VA_ARGS__); break;  \
    case 5: case 6: case 7: case 8:                                                                 \
      hipLaunchKernelGGL(KERNEL<8>, dim3(GRID), dim3(LN_THREADS), 0, s, __VA_ARGS__); break;        \
    default: hipLaunchKernelGGL(KERNEL<16>, dim3(GRID), dim3(LN_THREADS), 0, s, __VA_ARGS__); break; \
  }
