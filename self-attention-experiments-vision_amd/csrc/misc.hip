#include "common.h"
#include "savit.h"
extern "C" int savit_abi_version(void) { return SAVIT_ABI_VERSION; }

// ---- measurement utilities (include/savit.h "measurement")
// Timing events are created with hipEventDisableSystemFence.  A default-flag HIP event performs a system-scope release when it is
// recorded (L2 writeback + invalidate: hip_runtime_api.h says so itself, "the performance impact of those actions on the execution
// of following work"), so a pair of them around a launch changes what it measures; round 3's bench line carried exactly that
// (one kernel class 6x over its rocprofv3 time on some boxes, the timed region unaffected).
namespace {
struct Timer {
  int n;
  hipEvent_t* ev;
};

// Spins until `ticks` of the 100 MHz constant clock have passed: the gate in front of an instrumented step, so that the host
// enqueues the whole step while the GPU is still busy and no event pair can contain host time.  One wave; always terminates.
__global__ void spin_kernel(unsigned long long ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

// `cus` workgroups of 64 threads that each keep a whole CU's LDS (so nothing with a large LDS tile is co-scheduled there) for
// `ticks`: stands in for the CUs a resident RCCL all-reduce holds during backward (tools/cu_thief_probe.py).
__global__ void hold_cus_kernel(unsigned long long ticks) {
  extern __shared__ char hold_lds[];
  if (threadIdx.x == 0) hold_lds[0] = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
}  // namespace

extern "C" int savit_timer_create(int n_events, void** handle) {
  SAVIT_CHECK_ARG(handle != nullptr && n_events > 0 && n_events <= (1 << 20));
  Timer* t = new Timer{n_events, new hipEvent_t[n_events]};
  for (int i = 0; i < n_events; ++i) {
    const hipError_t e = hipEventCreateWithFlags(&t->ev[i], hipEventDisableSystemFence);
    if (e != hipSuccess) {
      for (int j = 0; j < i; ++j) (void)hipEventDestroy(t->ev[j]);
      delete[] t->ev;
      delete t;
      return (int)e;
    }
  }
  *handle = t;
  return SAVIT_OK;
}

extern "C" int savit_timer_record(void* handle, int idx, void* stream) {
  Timer* t = reinterpret_cast<Timer*>(handle);
  SAVIT_CHECK_ARG(t != nullptr && idx >= 0 && idx < t->n);
  return (int)hipEventRecord(t->ev[idx], (hipStream_t)stream);
}

extern "C" int savit_timer_elapsed_ms(void* handle, int first, int second, float* ms) {
  Timer* t = reinterpret_cast<Timer*>(handle);
  SAVIT_CHECK_ARG(t != nullptr && ms != nullptr && first >= 0 && first < t->n && second >= 0 && second < t->n);
  return (int)hipEventElapsedTime(ms, t->ev[first], t->ev[second]);
}

extern "C" int savit_timer_destroy(void* handle) {
  Timer* t = reinterpret_cast<Timer*>(handle);
  SAVIT_CHECK_ARG(t != nullptr);
  for (int i = 0; i < t->n; ++i) (void)hipEventDestroy(t->ev[i]);
  delete[] t->ev;
  delete t;
  return SAVIT_OK;
}

extern "C" int savit_spin(long microseconds, void* stream) {
  SAVIT_CHECK_ARG(microseconds >= 0 && microseconds <= 200000);  // bounded: a gate, not a hang
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long)microseconds * 100ull);
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_hold_cus(int cus, long microseconds, void* stream) {
  SAVIT_CHECK_ARG(cus > 0 && cus <= 256 && microseconds >= 0 && microseconds <= 200000);
  auto kfn = hold_cus_kernel;
  SAVIT_LDS_ONCE(kfn);
  // 96 KB of dynamic LDS: more than half of a CU's 160 KB, so two holders never share a CU and each holder takes one
  hipLaunchKernelGGL(kfn, dim3(cus), dim3(64), 96 * 1024, (hipStream_t)stream, (unsigned long long)microseconds * 100ull);
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_zero_bytes(void* dst, long bytes, void* stream) {
  SAVIT_CHECK_ARG(dst != nullptr && bytes >= 0);
  if (bytes == 0) return SAVIT_OK;
  return (int)hipMemsetAsync(dst, 0, (size_t)bytes, (hipStream_t)stream);
}

extern "C" int savit_set_cu_budget(int cus) {
  SAVIT_CHECK_ARG(cus >= 0);
  savit_cu_budget_.store(cus, std::memory_order_relaxed);
  return SAVIT_OK;
}
