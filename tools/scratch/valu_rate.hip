// VALU issue-rate probe: N dependent-free v_fma_f32 per wave, W waves per SIMD.  hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a) {
  float x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = threadIdx.x + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) x[i] = __builtin_fmaf(x[i], a, 1.0f);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main(int argc, char** argv) {
  int wgs_per_cu = argc > 1 ? atoi(argv[1]) : 1;  // 256-thread WGs per CU = waves per SIMD
  int iters = 20000;
  float* out;
  hipMalloc(&out, 256 * 8 * 256 * sizeof(float) * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256 * wgs_per_cu), dim3(256), 0, 0, out, iters, 0.999f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double instr_per_wave = (double)iters * 128;
    double cyc = ms * 1e-3 * 2.4e9;
    printf("waves/SIMD %d: %.3f ms, %.2f cycles (at 2.4 GHz) per wave-instruction per SIMD, %.1f TFLOP/s fp32\n", wgs_per_cu, ms,
           cyc / (instr_per_wave * wgs_per_cu), 2.0 * 64 * instr_per_wave * 4 * wgs_per_cu * 256 / (ms * 1e-3) / 1e12);
  }
  return 0;
}
