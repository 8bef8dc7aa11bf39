#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
#define SWAP16(a, b) asm volatile("s_nop 3\n v_permlane16_swap_b32 %0, %1\n s_nop 3" : "+v"(a), "+v"(b))
#define SWAP32(a, b) asm volatile("s_nop 3\n v_permlane32_swap_b32 %0, %1\n s_nop 3" : "+v"(a), "+v"(b))
__device__ __forceinline__ float wave_sum2(float v) {
  v += dpp_mov<0xB1>(v); v += dpp_mov<0x4E>(v); v += dpp_mov<0x141>(v); v += dpp_mov<0x140>(v);
  float a = v, b = v; SWAP16(a, b); v = a + b;
  a = v; b = v; SWAP32(a, b);
  return a + b;
}
__global__ void k(float* out) {
  const int l = threadIdx.x;
  float a = (float)l, b = (float)(100 + l);
  SWAP16(a, b);
  out[0 * 64 + l] = a; out[1 * 64 + l] = b;
  out[2 * 64 + l] = wave_sum2((float)l);
  out[3 * 64 + l] = wave_sum2((float)(l * l % 7));
}
int main() {
  float* d; hipMalloc(&d, 4 * 64 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[4 * 64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[4] = {"swap16 a", "swap16 b", "sum(l)=2016", "sum(l*l%7)"};
  int ref = 0; for (int l = 0; l < 64; ++l) ref += l * l % 7;
  for (int r = 0; r < 4; ++r) { printf("%-12s", names[r]); for (int l = 0; l < 64; ++l) printf(" %3d", (int)h[r * 64 + l]); printf("\n"); }
  printf("ref %d\n", ref);
  return 0;
}
