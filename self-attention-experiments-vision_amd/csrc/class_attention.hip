// Class attention (CaiT token-only layers): ClassSelfAttentionBlock at /root/reference/models/cait.py:10-15 -
// AttentionBlock (attention.py:21-67) with ONE query (the cls token, row 0 of the normalised [cls; x] sequence) against all
// Nk = N+1 keys.  Per (batch, head) this is a 1 x Nk softmax and two GEMVs: latency-bound, no MFMA.  One 64-lane wave per
// (batch, head): keys are strided over the lanes (<= 4 per lane), the query / output-cotangent row is broadcast-loaded into
// every lane, row max / sum / dot products are wave shuffles.  The K and V projections over all tokens - where the FLOPs of a
// token-only layer are - are ordinary savit_gemm_bf16_tn calls on a fused [d, 2d] weight.
#include "common.h"
#include "savit.h"

namespace {

constexpr int CA_KPL = 4;  // keys per lane: Nk <= 256

template <int HDV>
__device__ __forceinline__ void load_row_f32(const bf16_t* p, float (&out)[HDV]) {
#pragma unroll
  for (int c = 0; c < HDV / 8; ++c) {
    const uint4 v = reinterpret_cast<const uint4*>(p)[c];
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      out[8 * c + 2 * k] = __uint_as_float(w[k] << 16);
      out[8 * c + 2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u);
    }
  }
}

template <int HDV>
__device__ __forceinline__ void store_row_bf16(bf16_t* p, const float (&v)[HDV]) {
#pragma unroll
  for (int c = 0; c < HDV / 8; ++c)
    reinterpret_cast<uint4*>(p)[c] = make_uint4(pack_bf16x2(v[8 * c], v[8 * c + 1]), pack_bf16x2(v[8 * c + 2], v[8 * c + 3]),
                                                pack_bf16x2(v[8 * c + 4], v[8 * c + 5]), pack_bf16x2(v[8 * c + 6], v[8 * c + 7]));
}

// q [B, ldq] (row b, columns h*hd..: already scaled), kv [B*Nk, ldkv]: k at column h*hd, v at column d + h*hd
template <int HDV>
__global__ __launch_bounds__(256) void class_attn_fwd_kernel(const bf16_t* __restrict__ q, long ldq, const bf16_t* __restrict__ kv, int ldkv,
                                                              bf16_t* __restrict__ o, float* __restrict__ probs, int B, int Nk, int H) {
  const int lane = threadIdx.x & 63;
  const int bh = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (bh >= B * H) return;
  const int b = bh / H, h = bh - b * H, d = H * HDV;
  float qv[HDV];
  load_row_f32<HDV>(q + (size_t)b * ldq + h * HDV, qv);
  float s[CA_KPL], m = -INFINITY;
#pragma unroll
  for (int kk = 0; kk < CA_KPL; ++kk) {
    const int key = lane + 64 * kk;
    s[kk] = -INFINITY;
    if (key < Nk) {
      float kr[HDV];
      load_row_f32<HDV>(kv + ((size_t)b * Nk + key) * ldkv + h * HDV, kr);
      float a = 0.f;
#pragma unroll
      for (int e = 0; e < HDV; ++e) a += qv[e] * kr[e];
      s[kk] = round_bf16(a);  // the reference's score tensor is bf16
    }
    m = fmaxf(m, s[kk]);
  }
  m = wave_max(m);
  float l = 0.f;
#pragma unroll
  for (int kk = 0; kk < CA_KPL; ++kk) {
    s[kk] = __expf(s[kk] - m);
    l += s[kk];
  }
  l = wave_sum(l);
  const float inv = 1.0f / l;
  float acc[HDV];
#pragma unroll
  for (int e = 0; e < HDV; ++e) acc[e] = 0.f;
#pragma unroll
  for (int kk = 0; kk < CA_KPL; ++kk) {
    const int key = lane + 64 * kk;
    const float pk = s[kk] * inv;
    if (key < Nk) {
      probs[((size_t)b * H + h) * Nk + key] = pk;
      float vr[HDV];
      load_row_f32<HDV>(kv + ((size_t)b * Nk + key) * ldkv + d + h * HDV, vr);
#pragma unroll
      for (int e = 0; e < HDV; ++e) acc[e] += pk * vr[e];
    }
  }
#pragma unroll
  for (int e = 0; e < HDV; ++e) acc[e] = wave_sum(acc[e]);
  if (lane == 0) store_row_bf16<HDV>(o + (size_t)b * d + h * HDV, acc);
}

template <int HDV>
__global__ __launch_bounds__(256) void class_attn_bwd_kernel(const bf16_t* __restrict__ q, long ldq, const bf16_t* __restrict__ kv, int ldkv,
                                                              const float* __restrict__ probs, const bf16_t* __restrict__ d_o,
                                                              bf16_t* __restrict__ dq, long lddq, bf16_t* __restrict__ dkv, int B, int Nk, int H,
                                                              float dq_scale) {
  const int lane = threadIdx.x & 63;
  const int bh = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (bh >= B * H) return;
  const int b = bh / H, h = bh - b * H, d = H * HDV;
  float qv[HDV], dov[HDV];
  load_row_f32<HDV>(q + (size_t)b * ldq + h * HDV, qv);
  load_row_f32<HDV>(d_o + (size_t)b * d + h * HDV, dov);
  float pk[CA_KPL], dp[CA_KPL], del = 0.f;
#pragma unroll
  for (int kk = 0; kk < CA_KPL; ++kk) {
    const int key = lane + 64 * kk;
    pk[kk] = 0.f;
    dp[kk] = 0.f;
    if (key < Nk) {
      pk[kk] = probs[((size_t)b * H + h) * Nk + key];
      float vr[HDV];
      load_row_f32<HDV>(kv + ((size_t)b * Nk + key) * ldkv + d + h * HDV, vr);
      float a = 0.f;
#pragma unroll
      for (int e = 0; e < HDV; ++e) a += dov[e] * vr[e];
      dp[kk] = a;
      del += pk[kk] * a;
    }
  }
  del = wave_sum(del);
  float dqa[HDV];
#pragma unroll
  for (int e = 0; e < HDV; ++e) dqa[e] = 0.f;
#pragma unroll
  for (int kk = 0; kk < CA_KPL; ++kk) {
    const int key = lane + 64 * kk;
    if (key < Nk) {
      const float ds = pk[kk] * (dp[kk] - del);
      float kr[HDV], dkr[HDV], dvr[HDV];
      load_row_f32<HDV>(kv + ((size_t)b * Nk + key) * ldkv + h * HDV, kr);
#pragma unroll
      for (int e = 0; e < HDV; ++e) {
        dqa[e] += ds * kr[e];
        dkr[e] = ds * qv[e];
        dvr[e] = pk[kk] * dov[e];
      }
      store_row_bf16<HDV>(dkv + ((size_t)b * Nk + key) * ldkv + h * HDV, dkr);
      store_row_bf16<HDV>(dkv + ((size_t)b * Nk + key) * ldkv + d + h * HDV, dvr);
    }
  }
#pragma unroll
  for (int e = 0; e < HDV; ++e) dqa[e] = wave_sum(dqa[e]) * dq_scale;
  if (lane == 0) store_row_bf16<HDV>(dq + (size_t)b * lddq + h * HDV, dqa);
}


// ------------------------------------------------------------------------------------------------------------
// The cls query of a ViT's LAST encoder layer (round 5).  Only row 0 of every image reaches the head (vit.py:57,95), so of the last
// layer's attention only the cls query's output is ever used and only its row of the cotangent is non-zero: one query per image
// against all N keys - the shape of the kernels above - but with the ROUNDING POINTS OF THE MFMA ATTENTION KERNELS (csrc/attention.hip),
// so that a model differentiated this way agrees with the dense plan to fp32 summation order: fp32 scores (the query is pre-scaled),
// P operand = bf16(exp2((s - max) log2 e)) un-normalised with the row sum taken from the un-rounded values, O = sum(P v) / l, LSE saved;
// backward: P = exp2(s log2 e - LSE log2 e) recomputed, dV = bf16(P) dO, dP = dO . v, delta = rowsum(dO * O), dS = P (dP - delta) in
// fp32, dK = bf16(dS) q, dQ = dq_scale * sum bf16(dS) k.  One wave per (image, head).
constexpr float CQ_LOG2E = 1.4426950408889634f;

// Backward: lane = (key slot ks = lane / 8, 16-byte piece pc = lane % 8): one wave instruction moves 8 WHOLE key rows of 128 bytes (head
// width 48: six of the eight pieces).  (First form: a lane per key read and wrote whole rows - 64 different cache lines per instruction,
// eight instructions per row set, 64 wave reductions for dQ: 55 us per launch at DeiT-B's layer against an HBM floor of 28; this form: 32.)  Dot products: 8 multiply-adds per lane + a 3-step fold over the row's 8 lanes; sums over the keys: a 3-step fold over
// the 8 key slots at the end.
__device__ __forceinline__ void cq_unpack8(const uint4 w, float (&out)[8]) {
  out[0] = __uint_as_float(w.x << 16); out[1] = __uint_as_float(w.x & 0xffff0000u);
  out[2] = __uint_as_float(w.y << 16); out[3] = __uint_as_float(w.y & 0xffff0000u);
  out[4] = __uint_as_float(w.z << 16); out[5] = __uint_as_float(w.z & 0xffff0000u);
  out[6] = __uint_as_float(w.w << 16); out[7] = __uint_as_float(w.w & 0xffff0000u);
}
__device__ __forceinline__ void cq_load8(const bf16_t* p, bool ok, float (&out)[8]) {
  uint4 w = make_uint4(0u, 0u, 0u, 0u);
  if (ok) w = *reinterpret_cast<const uint4*>(p);
  cq_unpack8(w, out);
}
__device__ __forceinline__ void cq_store8(bf16_t* p, const float (&v)[8]) {
  *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
}
__device__ __forceinline__ float cq_row_sum(float x) {  // over the 8 lanes of a key row
  x += __shfl_xor(x, 1, 64);
  x += __shfl_xor(x, 2, 64);
  x += __shfl_xor(x, 4, 64);
  return x;
}
__device__ __forceinline__ float cq_slot_sum(float x) {  // over the 8 key slots (lanes with equal pc)
  x += __shfl_xor(x, 8, 64);
  x += __shfl_xor(x, 16, 64);
  x += __shfl_xor(x, 32, 64);
  return x;
}
__device__ __forceinline__ float cq_dot8(const float (&a)[8], const float (&b)[8]) {
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) s += a[e] * b[e];
  return s;
}

// Forward: a lane per key (KPL keys per lane), the whole 128-byte row per lane.  The 8-lanes-per-row form of the backward below was
// built for the forward too and is SLOWER there (21.4-24.6 us against 20.0 per launch at DeiT-B's layer, with 4 or 8 row loads in flight
// per lane, with and without the value rows held in registers): two dependent passes over the keys (scores -> row maximum -> P V) of
// 25 short iterations each leave a wave waiting on its own shuffles; the row-per-lane form does 64-term dot products without any.
template <int HDV, int KPL>
__global__ __launch_bounds__(256) void cls_query_attn_fwd_kernel(const bf16_t* __restrict__ q, long ldq, const bf16_t* __restrict__ kv, int ldkv,
                                                                  bf16_t* __restrict__ o, float* __restrict__ lse, int B, int Nk, int H) {
  const int lane = threadIdx.x & 63;
  const int bh = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (bh >= B * H) return;
  const int b = bh / H, h = bh - b * H, d = H * HDV;
  float qv[HDV];
  load_row_f32<HDV>(q + (size_t)b * ldq + h * HDV, qv);
  float s[KPL], m = -INFINITY;
#pragma unroll
  for (int kk = 0; kk < KPL; ++kk) {
    const int key = lane + 64 * kk;
    s[kk] = -INFINITY;
    if (key < Nk) {
      float kr[HDV];
      load_row_f32<HDV>(kv + ((size_t)b * Nk + key) * ldkv + h * HDV, kr);
      float a = 0.f;
#pragma unroll
      for (int e = 0; e < HDV; ++e) a += qv[e] * kr[e];
      s[kk] = a;
    }
    m = fmaxf(m, s[kk]);
  }
  m = wave_max(m);
  const float mb = m * CQ_LOG2E;
  float l = 0.f;
  float acc[HDV];
#pragma unroll
  for (int e = 0; e < HDV; ++e) acc[e] = 0.f;
#pragma unroll
  for (int kk = 0; kk < KPL; ++kk) {
    const int key = lane + 64 * kk;
    const float pe = __builtin_amdgcn_exp2f(s[kk] * CQ_LOG2E - mb);  // keys >= Nk: exp2(-inf) = 0
    l += pe;
    if (key < Nk) {
      const float pb = round_bf16(pe);
      float vr[HDV];
      load_row_f32<HDV>(kv + ((size_t)b * Nk + key) * ldkv + d + h * HDV, vr);
#pragma unroll
      for (int e = 0; e < HDV; ++e) acc[e] += pb * vr[e];
    }
  }
  l = wave_sum(l);
  const float inv = 1.0f / l;
#pragma unroll
  for (int e = 0; e < HDV; ++e) acc[e] = wave_sum(acc[e]) * inv;
  if (lane == 0) {
    store_row_bf16<HDV>(o + (size_t)b * d + h * HDV, acc);
    lse[(size_t)b * H + h] = m + __logf(l);
  }
}

template <int HDV, int NIT>
__global__ __launch_bounds__(256) void cls_query_attn_bwd_kernel(const bf16_t* __restrict__ q, long ldq, const bf16_t* __restrict__ kv, int ldkv,
                                                                  const bf16_t* __restrict__ o, const float* __restrict__ lse,
                                                                  const bf16_t* __restrict__ d_o, bf16_t* __restrict__ dq, long lddq,
                                                                  bf16_t* __restrict__ dkv, int B, int Nk, int H, float dq_scale) {
  const int lane = threadIdx.x & 63;
  const int bh = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (bh >= B * H) return;
  const int b = bh / H, h = bh - b * H, d = H * HDV;
  const int ks = lane >> 3, pc = lane & 7;
  const bool pc_ok = pc * 8 < HDV;
  float qv[8], dov[8], del;
  cq_load8(q + (size_t)b * ldq + h * HDV + pc * 8, pc_ok, qv);
  cq_load8(d_o + (size_t)b * d + h * HDV + pc * 8, pc_ok, dov);
  {
    float ov[8];
    cq_load8(o + (size_t)b * d + h * HDV + pc * 8, pc_ok, ov);
    del = cq_row_sum(cq_dot8(ov, dov));
  }
  const float nl2 = -CQ_LOG2E * lse[(size_t)b * H + h];
  const size_t row0 = (size_t)b * Nk * ldkv + h * HDV + pc * 8;
  float dqa[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) dqa[e] = 0.f;
  const int nit = (Nk + 7) / 8;
  constexpr int G = 4;  // key iterations requested together: 8 row loads in flight per lane
  for (int it0 = 0; it0 < nit; it0 += G) {
    float kr[G][8], vr[G][8];
    bool ok[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int key = (it0 + g) * 8 + ks;
      ok[g] = pc_ok && key < Nk;
      cq_load8(kv + row0 + (size_t)key * ldkv, ok[g], kr[g]);
      cq_load8(kv + row0 + (size_t)key * ldkv + d, ok[g], vr[g]);
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int key = (it0 + g) * 8 + ks;
      const float sc = cq_row_sum(cq_dot8(qv, kr[g])), dp = cq_row_sum(cq_dot8(dov, vr[g]));
      const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(sc, CQ_LOG2E, nl2));
      const float pb = round_bf16(pr), dsb = round_bf16(pr * (dp - del));
      float dkr[8], dvr[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        dqa[e] += ok[g] ? dsb * kr[g][e] : 0.f;
        dkr[e] = dsb * qv[e];
        dvr[e] = pb * dov[e];
      }
      if (ok[g]) {
        cq_store8(dkv + row0 + (size_t)key * ldkv, dkr);
        cq_store8(dkv + row0 + (size_t)key * ldkv + d, dvr);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) dqa[e] = cq_slot_sum(dqa[e]) * dq_scale;
  if (ks == 0 && pc_ok) cq_store8(dq + (size_t)b * lddq + h * HDV + pc * 8, dqa);
}

}  // namespace

extern "C" int savit_class_attention_fwd(const void* q, long ldq, const void* kv, int ldkv, void* o, float* probs, int B, int Nk, int H,
                                         int head_dim, void* stream) {
  SAVIT_CHECK_ARG(q && kv && o && probs && B >= 0 && Nk > 0 && Nk <= 64 * CA_KPL && H > 0 && (head_dim == 48 || head_dim == 64));
  SAVIT_CHECK_ARG(ldq >= H * head_dim && ldq % 8 == 0 && ldkv >= 2 * H * head_dim && ldkv % 8 == 0);
  SAVIT_CHECK_ARG(((uintptr_t)q % 16) == 0 && ((uintptr_t)kv % 16) == 0 && ((uintptr_t)o % 16) == 0);
  if (B == 0) return SAVIT_OK;
  const dim3 grid((B * H + 3) / 4), block(256);
  if (head_dim == 48)
    hipLaunchKernelGGL(class_attn_fwd_kernel<48>, grid, block, 0, (hipStream_t)stream, (const bf16_t*)q, ldq, (const bf16_t*)kv, ldkv, (bf16_t*)o,
                       probs, B, Nk, H);
  else
    hipLaunchKernelGGL(class_attn_fwd_kernel<64>, grid, block, 0, (hipStream_t)stream, (const bf16_t*)q, ldq, (const bf16_t*)kv, ldkv, (bf16_t*)o,
                       probs, B, Nk, H);
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_class_attention_bwd(const void* q, long ldq, const void* kv, int ldkv, const float* probs, const void* d_o, void* dq,
                                         long lddq, void* dkv, int B, int Nk, int H, int head_dim, float dq_scale, void* stream) {
  SAVIT_CHECK_ARG(lddq >= H * head_dim && lddq % 8 == 0);
  SAVIT_CHECK_ARG(q && kv && probs && d_o && dq && dkv && B >= 0 && Nk > 0 && Nk <= 64 * CA_KPL && H > 0 && (head_dim == 48 || head_dim == 64));
  SAVIT_CHECK_ARG(ldq >= H * head_dim && ldq % 8 == 0 && ldkv >= 2 * H * head_dim && ldkv % 8 == 0);
  SAVIT_CHECK_ARG(((uintptr_t)q % 16) == 0 && ((uintptr_t)kv % 16) == 0 && ((uintptr_t)d_o % 16) == 0 && ((uintptr_t)dq % 16) == 0 &&
                  ((uintptr_t)dkv % 16) == 0);
  if (B == 0) return SAVIT_OK;
  const dim3 grid((B * H + 3) / 4), block(256);
  if (head_dim == 48)
    hipLaunchKernelGGL(class_attn_bwd_kernel<48>, grid, block, 0, (hipStream_t)stream, (const bf16_t*)q, ldq, (const bf16_t*)kv, ldkv, probs,
                       (const bf16_t*)d_o, (bf16_t*)dq, lddq, (bf16_t*)dkv, B, Nk, H, dq_scale);
  else
    hipLaunchKernelGGL(class_attn_bwd_kernel<64>, grid, block, 0, (hipStream_t)stream, (const bf16_t*)q, ldq, (const bf16_t*)kv, ldkv, probs,
                       (const bf16_t*)d_o, (bf16_t*)dq, lddq, (bf16_t*)dkv, B, Nk, H, dq_scale);
  SAVIT_LAUNCH_RET();
}

// ---- the cls query against all keys with the MFMA kernels' rounding points (see cls_query_attn_fwd_kernel)
#define CQ_DISPATCH(KERNEL, SMALL, LARGE, ...)                                                                         \
  do {                                                                                                                 \
    const dim3 grid((B * H + 3) / 4), block(256);                                                                      \
    if (head_dim == 48) {                                                                                              \
      if (Nk <= 256) hipLaunchKernelGGL((KERNEL<48, SMALL>), grid, block, 0, (hipStream_t)stream, __VA_ARGS__);         \
      else hipLaunchKernelGGL((KERNEL<48, LARGE>), grid, block, 0, (hipStream_t)stream, __VA_ARGS__);                  \
    } else {                                                                                                           \
      if (Nk <= 256) hipLaunchKernelGGL((KERNEL<64, SMALL>), grid, block, 0, (hipStream_t)stream, __VA_ARGS__);         \
      else hipLaunchKernelGGL((KERNEL<64, LARGE>), grid, block, 0, (hipStream_t)stream, __VA_ARGS__);                  \
    }                                                                                                                  \
  } while (0)

extern "C" int savit_cls_query_attention_fwd(const void* q, long ldq, const void* kv, int ldkv, void* o, float* lse, int B, int Nk, int H,
                                             int head_dim, void* stream) {
  SAVIT_CHECK_ARG(q && kv && o && lse && B >= 0 && Nk > 0 && Nk <= 640 && H > 0 && (head_dim == 48 || head_dim == 64));
  SAVIT_CHECK_ARG(ldq >= H * head_dim && ldq % 8 == 0 && ldkv >= 2 * H * head_dim && ldkv % 8 == 0);
  SAVIT_CHECK_ARG(((uintptr_t)q % 16) == 0 && ((uintptr_t)kv % 16) == 0 && ((uintptr_t)o % 16) == 0);
  if (B == 0) return SAVIT_OK;
  CQ_DISPATCH(cls_query_attn_fwd_kernel, 4, 10, (const bf16_t*)q, ldq, (const bf16_t*)kv, ldkv, (bf16_t*)o, lse, B, Nk, H);
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_cls_query_attention_bwd(const void* q, long ldq, const void* kv, int ldkv, const void* o, const float* lse, const void* d_o,
                                             void* dq, long lddq, void* dkv, int B, int Nk, int H, int head_dim, float dq_scale, void* stream) {
  SAVIT_CHECK_ARG(q && kv && o && lse && d_o && dq && dkv && B >= 0 && Nk > 0 && Nk <= 640 && H > 0 && (head_dim == 48 || head_dim == 64));
  SAVIT_CHECK_ARG(ldq >= H * head_dim && ldq % 8 == 0 && ldkv >= 2 * H * head_dim && ldkv % 8 == 0 && lddq >= H * head_dim && lddq % 8 == 0);
  SAVIT_CHECK_ARG(((uintptr_t)q % 16) == 0 && ((uintptr_t)kv % 16) == 0 && ((uintptr_t)o % 16) == 0 && ((uintptr_t)d_o % 16) == 0 &&
                  ((uintptr_t)dq % 16) == 0 && ((uintptr_t)dkv % 16) == 0);
  if (B == 0) return SAVIT_OK;
  CQ_DISPATCH(cls_query_attn_bwd_kernel, 32, 80, (const bf16_t*)q, ldq, (const bf16_t*)kv, ldkv, (const bf16_t*)o, lse, (const bf16_t*)d_o, (bf16_t*)dq,
              lddq, (bf16_t*)dkv, B, Nk, H, dq_scale);
  SAVIT_LAUNCH_RET();
}
