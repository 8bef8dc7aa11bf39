// Self-attention over very short sequences: TNT's inner transformer (/root/reference/models/tnt.py:68-76 -> attention.py:21-67)
// attends over the 16 pixel tokens of one patch with 4 heads that are 6 or 10 wide - B*196 = 25 088 independent sequences per
// 128-image batch.  The tiled MFMA kernels (attention.hip) spend one 64-thread workgroup and two LDS image stagings per
// (sequence, head): 100 k workgroups, 115 us forward / 300 us backward per layer.  Here one WAVE owns one sequence:
//   lane = (head h = lane >> 4, query q = lane & 15); the sequence's 16 x 192 bf16 q|k|v rows (6 KB, contiguous in HBM) are parked
//   in LDS once; a lane computes its 16 scores, the softmax and its output row entirely in registers (no cross-lane reduction:
//   a row of S lives in one lane), reading k / v rows as LDS broadcasts.  Backward recomputes P (16 exps per lane), forms dQ in
//   the same mapping, exchanges P and dS through LDS (stored [h][key][q], so the second pass - lane = (h, key) - reads its column
//   as 16 contiguous floats) and forms dK, dV.  VALU fp32 math: 2 * 16 * 16 FMAs per lane forward - far below the HBM time
//   (205 MB forward, 360 MB backward at the TNT-B shapes), which is what bounds these kernels.
// (The LDS regions are private to a wave, yet replacing the workgroup barriers by wave-level hand-offs measured 45 % SLOWER: with
// the barriers the four waves issue their tile loads together.)
// Register note: with the key loops fully unrolled hipcc's SLP vectoriser pairs iterations across keys and keeps every k / v row of
// the sequence live (256 VGPRs, one wave per SIMD; a register cap only made it spill 100-500 VGPRs).  This file is therefore
// compiled with -fno-slp-vectorize (csrc/Makefile) and backward re-reads its rows after a compiler memory barrier between
// phases instead of keeping them: 69 VGPRs forward, 132 backward, no spills.
// Layout as for the tiled kernels: qkv bf16 [S*16, ld] = q (pre-scaled) | k | v, head-major, heads padded to 16 columns (zeros);
// o bf16 [S*16, 64]; dqkv receives dQ * dq_scale | dK | dV.  P is rounded to bf16 before P.V like an MFMA operand would be.
#include "common.h"
#include "savit.h"

namespace {

constexpr int T = 16, H = 4, HP = 16, W = H * HP;  // tokens, heads, padded head width, attention width (64)
constexpr float LOG2E_F = 1.4426950408889634f;

__device__ __forceinline__ void unpack16(const bf16_t* p, float (&v)[16]) {
  const uint4 a = reinterpret_cast<const uint4*>(p)[0], b = reinterpret_cast<const uint4*>(p)[1];
  const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    v[2 * i] = __uint_as_float(w[i] << 16);
    v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ void store16(bf16_t* p, const float (&v)[16], float scale) {
  uint32_t w[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) w[i] = pack_bf16x2(v[2 * i] * scale, v[2 * i + 1] * scale);
  reinterpret_cast<uint4*>(p)[0] = make_uint4(w[0], w[1], w[2], w[3]);
  reinterpret_cast<uint4*>(p)[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

// scores of this lane's query against the 16 keys of its head, softmax in registers; p = fp32 probabilities
__device__ __forceinline__ void row_softmax(const bf16_t* tile, int h, const float (&qv)[16], float (&p)[16]) {
  float m = -INFINITY;
#pragma unroll
  for (int key = 0; key < T; ++key) {
    float kv[16];
    unpack16(tile + key * 3 * W + W + h * HP, kv);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) s += qv[e] * kv[e];
    p[key] = s;
    m = fmaxf(m, s);
  }
  float l = 0.f;
#pragma unroll
  for (int key = 0; key < T; ++key) {
    p[key] = __builtin_amdgcn_exp2f((p[key] - m) * LOG2E_F);
    l += p[key];
  }
  const float inv = 1.0f / l;
#pragma unroll
  for (int key = 0; key < T; ++key) p[key] *= inv;
}

__global__ __launch_bounds__(256, 4) void seq16_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o, long nseq) {
  __shared__ __attribute__((aligned(16))) bf16_t tiles[4][T * 3 * W];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 4, q = lane & 15;
  bf16_t* tile = tiles[wave];
  for (long s0 = (long)blockIdx.x * 4; s0 < nseq; s0 += (long)gridDim.x * 4) {  // uniform trip count: barriers inside
    const long seq = s0 + wave;
    if (seq < nseq) {
      const uint4* src = reinterpret_cast<const uint4*>(qkv + (size_t)seq * T * 3 * W);
#pragma unroll
      for (int i = 0; i < 6; ++i) reinterpret_cast<uint4*>(tile)[lane + 64 * i] = src[lane + 64 * i];
    }
    __syncthreads();
    if (seq < nseq) {
      float qv[16], p[16], acc[16];
      unpack16(tile + q * 3 * W + h * HP, qv);
      row_softmax(tile, h, qv, p);
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int key = 0; key < T; ++key) {
        float vv[16];
        unpack16(tile + key * 3 * W + 2 * W + h * HP, vv);
        const float pb = round_bf16(p[key]);
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] += pb * vv[e];
      }
      store16(o + ((size_t)seq * T + q) * W + h * HP, acc, 1.0f);
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256, 2) void seq16_bwd_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ d_o, bf16_t* __restrict__ dqkv,
                                                         long nseq, float dq_scale) {
  __shared__ __attribute__((aligned(16))) bf16_t tiles[4][T * 3 * W];
  __shared__ __attribute__((aligned(16))) bf16_t dots[4][T * W];
  __shared__ __attribute__((aligned(16))) bf16_t pm[4][H * T * T], dsm[4][H * T * T];  // [h][key][q]; bf16-rounded values
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 4, q = lane & 15;
  bf16_t* tile = tiles[wave];
  bf16_t* dot = dots[wave];
  for (long s0 = (long)blockIdx.x * 4; s0 < nseq; s0 += (long)gridDim.x * 4) {
    const long seq = s0 + wave;
    const bool live = seq < nseq;
    if (live) {
      const uint4* src = reinterpret_cast<const uint4*>(qkv + (size_t)seq * T * 3 * W);
#pragma unroll
      for (int i = 0; i < 6; ++i) reinterpret_cast<uint4*>(tile)[lane + 64 * i] = src[lane + 64 * i];
      const uint4* dsrc = reinterpret_cast<const uint4*>(d_o + (size_t)seq * T * W);
#pragma unroll
      for (int i = 0; i < 2; ++i) reinterpret_cast<uint4*>(dot)[lane + 64 * i] = dsrc[lane + 64 * i];
    }
    __syncthreads();
    if (live) {  // pass 1: lane = (h, query)
      float qv[16], p[16], dov[16], dq[16], dp[16];
      unpack16(tile + q * 3 * W + h * HP, qv);
      row_softmax(tile, h, qv, p);
      asm volatile("" ::: "memory");
      unpack16(dot + q * W + h * HP, dov);
      float delta = 0.f;
#pragma unroll
      for (int key = 0; key < T; ++key) {
        float vv[16];
        unpack16(tile + key * 3 * W + 2 * W + h * HP, vv);
        float a = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) a += dov[e] * vv[e];
        dp[key] = a;
        delta += p[key] * a;
      }
      asm volatile("" ::: "memory");
#pragma unroll
      for (int e = 0; e < 16; ++e) dq[e] = 0.f;
#pragma unroll
      for (int key = 0; key < T; ++key) {
        const float ds = round_bf16(p[key] * (dp[key] - delta));  // dS and P are bf16 MFMA operands in the tiled kernels
        float kv[16];
        unpack16(tile + key * 3 * W + W + h * HP, kv);
#pragma unroll
        for (int e = 0; e < 16; ++e) dq[e] += ds * kv[e];
        pm[wave][(h * T + key) * T + q] = f32_to_bf16(p[key]);
        dsm[wave][(h * T + key) * T + q] = f32_to_bf16(ds);
      }
      store16(dqkv + ((size_t)seq * T + q) * 3 * W + h * HP, dq, dq_scale);
    }
    __syncthreads();
    if (live) {  // pass 2: lane = (h, key)
      const int key = q;
      float pc[16], dc[16], dk[16], dv[16];
      unpack16(pm[wave] + (h * T + key) * T, pc);
      unpack16(dsm[wave] + (h * T + key) * T, dc);
#pragma unroll
      for (int e = 0; e < 16; ++e) { dk[e] = 0.f; dv[e] = 0.f; }
#pragma unroll
      for (int qq = 0; qq < T; ++qq) {
        float qr[16], dr[16];
        unpack16(tile + qq * 3 * W + h * HP, qr);
        unpack16(dot + qq * W + h * HP, dr);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          dk[e] += dc[qq] * qr[e];
          dv[e] += pc[qq] * dr[e];
        }
      }
      bf16_t* row = dqkv + ((size_t)seq * T + key) * 3 * W;
      store16(row + W + h * HP, dk, 1.0f);
      store16(row + 2 * W + h * HP, dv, 1.0f);
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" int savit_seq16_attention_fwd(const void* qkv, void* o, long nseq, int tokens, int heads, int head_dim_padded, int ld_qkv, void* stream) {
  SAVIT_CHECK_ARG(qkv && o && nseq >= 0 && tokens == T && heads == H && head_dim_padded == HP && ld_qkv == 3 * W);
  SAVIT_CHECK_ARG(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)o % 16) == 0);
  if (nseq == 0) return SAVIT_OK;
  long blocks = (nseq + 3) / 4;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(seq16_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, (bf16_t*)o, nseq);
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_seq16_attention_bwd(const void* qkv, const void* d_o, void* dqkv, long nseq, int tokens, int heads, int head_dim_padded,
                                         int ld_qkv, float dq_scale, void* stream) {
  SAVIT_CHECK_ARG(qkv && d_o && dqkv && nseq >= 0 && tokens == T && heads == H && head_dim_padded == HP && ld_qkv == 3 * W);
  SAVIT_CHECK_ARG(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)d_o % 16) == 0 && ((uintptr_t)dqkv % 16) == 0);
  if (nseq == 0) return SAVIT_OK;
  long blocks = (nseq + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(seq16_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, (const bf16_t*)d_o,
                     (bf16_t*)dqkv, nseq, dq_scale);
  SAVIT_LAUNCH_RET();
}
