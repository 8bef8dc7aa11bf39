// Two ways to read the result of __builtin_amdgcn_permlane{16,32}_swap (ROCm 7.2, hipcc -O3 --offload-arch=gfx950 -S):
// kernel k   reads it as __builtin_bit_cast(float, r[i]): the front end loads element 0 for both i (the ISA adds v1 + v1);
// kernel k_ok copies r[0], r[1] to scalars first: correct (v_permlane*_swap a, b ; v_add a, b), hazard nops placed by hipcc.
#include <hip/hip_runtime.h>
__device__ __forceinline__ float half_sum_b(float v) {
  auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
  return __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
}
__device__ __forceinline__ float row_sum_b(float v) {
  auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
  return __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
}
__global__ void k(const float* x, float* y) {
  float v = x[threadIdx.x];
  float w = x[threadIdx.x + 64] * 3.f;
  v = row_sum_b(v); w = row_sum_b(w);
  v = half_sum_b(v); w = half_sum_b(w);
  y[threadIdx.x] = v * w;
}
__device__ __forceinline__ float half_sum_ok(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const unsigned r0 = r[0], r1 = r[1];
  return __uint_as_float(r0) + __uint_as_float(r1);
}
__device__ __forceinline__ float row_sum_ok(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const unsigned r0 = r[0], r1 = r[1];
  return __uint_as_float(r0) + __uint_as_float(r1);
}
__global__ void k_ok(const float* x, float* y) {
  float v = x[threadIdx.x];
  float w = x[threadIdx.x + 64] * 3.f;
  v = row_sum_ok(v); w = row_sum_b(w);
  v = half_sum_ok(v); w = half_sum_b(w);
  y[threadIdx.x] = v * w;
}
