// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the ViT/CaiT training path.
// Wave = 64 lanes everywhere; no other target is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) int int32x4_t;  // a buffer descriptor as an inline-asm SGPR operand

// Non-temporal ("nt") accesses for data with no reuse in the near future: a tensor saved for backward, or the last read of one.
// They keep the 256 MB memory-side cache for the tensors the NEXT kernel reads (measured: storing the pre-GELU activation nt
// makes the following fc2 GEMM 14 us faster per layer).  Never for operands several workgroups share: nt lines are evicted first.
__device__ __forceinline__ float4 nt_load_f4(const float* p) {
  const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ uint4 nt_load_u4(const void* p) {
  const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
  return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ uint2 nt_load_u2(const void* p) {
  const u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(p));
  return make_uint2(v[0], v[1]);
}
__device__ __forceinline__ void nt_store_u4(void* p, uint4 v) {
  __builtin_nontemporal_store(u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4*>(p));
}

#define SAVIT_OK 0
#define SAVIT_EINVAL 1001  // shape / alignment contract violated (host-side check)

#define SAVIT_CHECK_ARG(cond) \
  do {                        \
    if (!(cond)) return SAVIT_EINVAL; \
  } while (0)

#define SAVIT_LAUNCH_RET() return (int)hipGetLastError()

// CUs the persistent kernels (one workgroup per CU walking over work items: attention forward / backward) may count on; 0 = all.
// Set through savit_set_cu_budget by a host that knows other kernels hold CUs beside its own (a data-parallel rank: RCCL's channels);
// a persistent grid of one workgroup per CU of the WHOLE chip would leave the workgroups that find no free CU to start when the
// first ones have finished all their items - twice the kernel's time.  One variable for the library (C++17 inline).
inline std::atomic<int> savit_cu_budget_{0};

// Kernels that use more than the default 64 KB of dynamic LDS need their limit raised.  That is done ONCE per kernel symbol (a
// function-local static: initialised thread-safely, C++11) to the CU's whole 160 KB - a limit, not an allocation: each launch still
// passes the bytes it uses - instead of a driver call in front of every launch.
inline hipError_t savit_raise_lds_limit(const void* kfn) {
  hipFuncAttributes at{};
  const hipError_t e = hipFuncGetAttributes(&at, kfn);
  if (e != hipSuccess) return e;
  // the limit covers dynamic LDS only: a kernel's static __shared__ arrays come out of the same 160 KB
  return hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - (int)at.sharedSizeBytes);
}
// The attribute applies to the CURRENT device: the once-flag is per (kernel symbol, device) - a process that later launches on another
// GPU raises the limit there too (first launch per device; at most 16 devices per process, beyond that the call is made every time).
#define SAVIT_LDS_ONCE(kfn)                                                                  \
  do {                                                                                       \
    static std::atomic<int> lds_done_[16];                                                   \
    int dev_ = 0;                                                                            \
    (void)hipGetDevice(&dev_);                                                               \
    if (dev_ < 0 || dev_ >= 16 || lds_done_[dev_].load(std::memory_order_acquire) == 0) {    \
      const hipError_t e_ = savit_raise_lds_limit((const void*)(kfn));                       \
      if (e_ != hipSuccess) return (int)e_;                                                  \
      if (dev_ >= 0 && dev_ < 16) lds_done_[dev_].store(1, std::memory_order_release);       \
    }                                                                                        \
  } while (0)

// Development switches (ablation tiles that compute wrong results on purpose, SAVIT_* environment overrides of the tile / split
// heuristics, phases compiled out for timing) exist only in builds that define SAVIT_EXPERIMENTS - tools/build_variant.sh does, the
// Makefile of the product library does not: libsavit.so reads no environment variable and has no run-time ablation path.
#ifdef SAVIT_EXPERIMENTS
#include <stdlib.h>
#define SAVIT_EXP_ENV_INT(name, dflt) ([] { const char* e_ = getenv(name); return e_ ? atoi(e_) : (dflt); }())
#else
#define SAVIT_EXP_ENV_INT(name, dflt) (dflt)
#endif

__device__ __forceinline__ float bf16_to_f32(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// round-to-nearest-even; a plain cast keeps NaN a NaN (v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;
  return *reinterpret_cast<bf16_t*>(&b);
}

// both halves in ONE v_cvt_pk_bf16_f32 (two scalar casts cost two converts plus a shift/or to merge them)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) float f2_t;
  typedef __attribute__((ext_vector_type(2))) __bf16 b2_t;
  const b2_t r = __builtin_convertvector(f2_t{lo, hi}, b2_t);
  return __builtin_bit_cast(uint32_t, r);
}

__device__ __forceinline__ float round_bf16(float f) { return bf16_to_f32(f32_to_bf16(f)); }

// Wave-wide all-reduce without the LDS crossbar: __shfl_xor lowers to ds_bpermute_b32 (six dependent ~100-cycle hops per
// reduction).  Here lanes 1, 2 apart exchange by DPP quad_perm, 4 and 8 apart by row_half_mirror / row_mirror (after the quad steps
// every quad is uniform, so mirroring reaches "the other quad / the other half row"), 16 and 32 apart by gfx950's
// v_permlane16_swap / v_permlane32_swap on two copies of the value (a' + b' then holds both partners everywhere).
// The swaps are the compiler's builtins, so hipcc's hazard recogniser places the wait states (2 between a VALU write of an operand
// and the swap, none after it) - round 1 used inline asm with hand-counted s_nops because the builtin seemed to "return its first
// result twice".  That was a front-end bug in how the result was READ: __builtin_bit_cast(float, r[i]) on an element of the
// returned vector loads element 0 for every i (clang 19 / ROCm 7.2, visible in the -O0 IR); copying the elements to scalars first
// is compiled correctly (tools/scratch/permlane_builtin.hip holds the two forms).
__device__ __forceinline__ void permlane32_swap(float& a, float& b) {  // lanes 32-63 of a <-> lanes 0-31 of b
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  const unsigned r0 = r[0], r1 = r[1];
  a = __uint_as_float(r0);
  b = __uint_as_float(r1);
}
__device__ __forceinline__ void permlane16_swap(float& a, float& b) {  // odd rows (16 lanes) of a <-> even rows of b
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  const unsigned r0 = r[0], r1 = r[1];
  a = __uint_as_float(r0);
  b = __uint_as_float(r1);
}
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// v(l) (+ / max) v(l ^ 32): the two halves of a wave, e.g. the two key halves of a 32x32 MFMA accumulator column
__device__ __forceinline__ float half_sum(float v) {
  float a = v, b = v;
  permlane32_swap(a, b);
  return a + b;
}
__device__ __forceinline__ float half_max(float v) {
  float a = v, b = v;
  permlane32_swap(a, b);
  return fmaxf(a, b);
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);  // row_half_mirror
  v += dpp_mov<0x140>(v);  // row_mirror
  float a = v, b = v;
  permlane16_swap(a, b);
  v = a + b;
  a = v;
  b = v;
  permlane32_swap(a, b);
  return a + b;
}
// v_permlane32_swap on raw dwords, and the store-widening exchange built on it: a lane (row, half) of a swapped-layout MFMA tile
// holds columns 8g + 4*half .. +3 of its row as one 8-byte piece per g.  Given the pieces of g0 (a) and g0 + 1 (b), the two lanes
// of a row trade one piece each, after which lane half = 0 holds columns 8*g0 .. 8*g0+7 and lane half = 1 columns 8*(g0+1) .. :
// one 16-byte store per lane at column 8*(g0 + half) instead of two 8-byte ones (store issue, not bandwidth, bounds these tails).
__device__ __forceinline__ void permlane32_swap_u32(uint32_t& a, uint32_t& b) {
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  const unsigned r0 = r[0], r1 = r[1];
  a = r0;
  b = r1;
}
__device__ __forceinline__ uint4 merge_row_halves(uint2 a, uint2 b) {
  permlane32_swap_u32(a.x, b.x);
  permlane32_swap_u32(a.y, b.y);
  return make_uint4(a.x, a.y, b.x, b.y);
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  v = fmaxf(v, dpp_mov<0x140>(v));
  float a = v, b = v;
  permlane16_swap(a, b);
  v = fmaxf(a, b);
  a = v;
  b = v;
  permlane32_swap(a, b);
  return fmaxf(a, b);
}

// jax.nn.gelu(approximate=True): 0.5 x (1 + tanh(z)), z = sqrt(2/pi) (x + 0.044715 x^3).
// Algebraically 0.5 (1 + tanh z) = sigmoid(2z) = 1 / (1 + exp2(-2 z log2 e)), which costs 5 VALU + 2 transcendental
// instructions per element (v_exp_f32, v_rcp_f32) instead of a tanh expansion; exact at +-inf, abs error ~1e-7.
__device__ __forceinline__ float gelu_sigmoid_arg(float x) {
  // -2 * log2(e) * sqrt(2/pi) * (x + 0.044715 x^3) = x * (k1 + k3 x^2)
  const float k1 = -2.302208198f;    // -2 * 1.4426950409 * 0.7978845608
  const float k3 = -0.1029432397f;   // k1 * 0.044715
  return x * (k1 + k3 * x * x);
}
__device__ __forceinline__ float gelu_tanh_f(float x) {
  const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(gelu_sigmoid_arg(x)));
  return x * s;
}
// d/dx gelu_tanh(x) = s + x s (1 - s) * 2 sqrt(2/pi) (1 + 3*0.044715 x^2),  s = sigmoid(2z)
// Two elements at a time: the FMAs / multiplies / adds become v_pk_*_f32 (two lanes' worth per issue slot); only v_exp_f32 and
// v_rcp_f32 stay scalar.  The fused GELU epilogues are VALU-bound (about 18 VALU + 2 transcendental instructions per element
// over a 256x256 tile per CU), so this is where their time goes.  Same formulas as the scalar forms above.
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 unpack_bf16x2(uint32_t w) { return f32x2{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)}; }
__device__ __forceinline__ f32x2 gelu_sigmoid2(f32x2 x) {
  const f32x2 t = x * (-2.302208198f + -0.1029432397f * x * x);
  const f32x2 e = f32x2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + 1.0f;
  return f32x2{__builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y)};
}
__device__ __forceinline__ f32x2 gelu_tanh2(f32x2 x) { return x * gelu_sigmoid2(x); }
__device__ __forceinline__ f32x2 gelu_tanh_grad2(f32x2 x) {
  const f32x2 s = gelu_sigmoid2(x);
  const f32x2 w = 1.5957691216f + 0.2140610297f * x * x;
  return s + x * s * (1.0f - s) * w;
}
__device__ __forceinline__ float gelu_tanh_grad_f(float x) {
  const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(gelu_sigmoid_arg(x)));
  const float w = 1.5957691216f + 0.2140610297f * x * x;  // 2c (1 + 0.134145 x^2)
  return s + x * s * (1.0f - s) * w;
}

// Bijective XCD-aware remap of a linear workgroup id: blocks b and b+8 share an XCD (speed only,
// never correctness), so give each XCD a contiguous chunk of the tile list.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (orig >> 3);
}
