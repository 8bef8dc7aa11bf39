// Fused multi-head self-attention, forward and backward, head_dim 64, sequence <= 256 tokens
// (ViT/DeiT 224^2: N = 197; CaiT SA layers: N = 196).
//
// Replaces, per (batch, head), attention.py:41-58 of /root/reference/models/layers/attentions/:
//   S = einsum('qhd,khd->hqk', q/sqrt(hd), k);  P = softmax(S);  O = einsum('hqk,khd->qhd', P, v)
// and its reverse-mode derivative.  S and P never touch HBM.
//
// gfx950 design (one workgroup per (batch, head); one 64-lane wave per 32-row block):
//   * K, V (and Q, dO in backward) tiles of the head go HBM -> LDS once by LDS-DMA straight out of the packed
//     [M, 3d] QKV buffer; rows >= N are zero-filled through the buffer descriptor's range check;
//   * every LDS image is [token][64 x bf16] (128-B rows) with ONE swizzle, chunk ^= rot(row) where
//     rot(row) = ((row>>1)&1)<<2 | ((row>>2)&3): conflict-free for ds_read_b128 row fragments AND for
//     ds_read_b64_tr_b16 transposed fragments, so the same image feeds QK^T-type and PV-type products;
//   * "swapped" score tile S^T = K.Q^T (v_mfma_f32_32x32x16_bf16): the query index lives on the lane, so the
//     softmax row max / row sum are in-register reductions plus ONE cross-half shuffle, and the score
//     accumulators are directly the B operand of the next product (O^T = V^T.P^T) - no LDS round trip for P;
//   * softmax statistics in fp32 (exp2 with log2e folded in), P rounded to bf16 only as an MFMA operand;
//   * forward saves LSE = max + log(sum); backward recomputes P = exp(S - LSE) (no N x N tensor is stored);
//   * backward runs two passes inside one launch over the same LDS images: pass A (wave owns 32 queries)
//     produces dQ and delta = rowsum(dO * O); pass B (wave owns 32 keys) produces dK and dV - no atomics, no
//     cross-workgroup reduction, bitwise reproducible.
#include "common.h"
#include "savit.h"
#include <stdlib.h>

namespace {

constexpr int HD = 64;              // head dim
constexpr int ROWB = HD * 2;        // LDS row bytes
constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ int rot3(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }

// byte offset of 16-B chunk c of `row` inside an image
__device__ __forceinline__ int img_off(int row, int c) { return row * ROWB + ((c ^ rot3(row)) << 4); }

__device__ __forceinline__ bf16x8 lds_row_frag(const char* img, int row, int c) {
  return *reinterpret_cast<const bf16x8*>(img + img_off(row, c));
}

// transposed fragment for a 32x32x16 A operand: rows row0..row0+3 and row0+8..row0+11 (this lane addresses
// row0 + (t>>2)), columns col0 + 4*(t&3) .. +3 where t = lane & 15
__device__ __forceinline__ bf16x8 lds_tr_frag(const char* img, int row0, int col) {
  const int r = row0;
  const int o0 = r * ROWB + (((col >> 3) ^ rot3(r)) << 4) + ((col & 7) << 1);
  const int r8 = r + 8;
  const int o1 = r8 * ROWB + (((col >> 3) ^ rot3(r8)) << 4) + ((col & 7) << 1);
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(img + o0));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(img + o1));
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// registers 8s..8s+7 of a 32x32 fp32 accumulator -> bf16 B-operand fragment of k-step s
__device__ __forceinline__ bf16x8 acc_to_frag(const f32x16& a, int s) {
  union { uint32_t u[4]; bf16x8 v; } r;
#pragma unroll
  for (int j = 0; j < 4; ++j) r.u[j] = pack_bf16x2(a[8 * s + 2 * j], a[8 * s + 2 * j + 1]);
  return r.v;
}

// Stage `rows` (multiple of 8*NW... handled by caller loop) token rows of one [token][64] slice into an LDS image.
// src column offset `col0` (elements) inside rows of length ld; token t maps to global row row_base + t; t >= N -> zeros.
template <int NW, int NT>
__device__ __forceinline__ void stage_image(char* img, __amdgpu_buffer_rsrc_t srd, long row_base, int N, int ld, int col0, int wave,
                                            int lane) {
  constexpr int INSTR = NT * 32 / 8;  // wave-instructions for the whole image (8 rows each)
  const int lrow = lane >> 3, pc = lane & 7;
  for (int inst = wave; inst < INSTR; inst += NW) {
    const int t = inst * 8 + lrow;
    const int c = pc ^ rot3(t);
    uint32_t voff = 0xfffffff0u;
    if (t < N) voff = (uint32_t)(((size_t)(row_base + t) * ld + col0 + c * 8) * 2);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)(img + inst * 1024), 16, voff, 0, 0, 0);
  }
}

struct AttnParams {
  const bf16_t* qkv;  // [B*N, ld] : q | k | v, each d = H*64 wide, head-major
  bf16_t* o;          // [B*N, d]
  float* lse;         // [B, H, N]
  const bf16_t* d_o;  // backward: [B*N, d]
  bf16_t* dqkv;       // backward: [B*N, ld]
  int B, N, H, ld, d;
  float dq_scale;     // backward: dQ is multiplied by this (the 1/sqrt(hd) folded into the QKV epilogue)
};

// ------------------------------------------------------------------------------------------ forward
template <int NT>
__global__ __launch_bounds__(64 * NT) void attn_fwd_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int IMG = NT * 32 * ROWB;
  char* imgK = smem;
  char* imgV = smem + IMG;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x / p.H, hh = blockIdx.x - b * p.H;
  const long row_base = (long)b * p.N;
  size_t bytes = (size_t)p.B * p.N * p.ld * 2;
  if (bytes > 0xffffffe0ull) bytes = 0xffffffe0ull;
  const auto srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.qkv), 0, (uint32_t)bytes, 0x00020000);
  stage_image<NT, NT>(imgK, srd, row_base, p.N, p.ld, p.d + hh * HD, wave, lane);
  stage_image<NT, NT>(imgV, srd, row_base, p.N, p.ld, 2 * p.d + hh * HD, wave, lane);

  // Q fragments of this wave's 32 queries straight from HBM (B operand: lane = (q, half), 8 consecutive e)
  const int ql = lane & 31, half = lane >> 5;
  const int q = wave * 32 + ql;
  bf16x8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    qf[ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    if (q < p.N) qf[ks] = *reinterpret_cast<const bf16x8*>(p.qkv + (size_t)(row_base + q) * p.ld + hh * HD + 16 * ks + 8 * half);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // S^T tiles: rows = keys (registers), column = query (lane)
  f32x16 s[NT];
#pragma unroll
  for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 kf = lds_row_frag(imgK, kt * 32 + ql, 2 * ks + half);
      s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[kt], 0, 0, 0);
    }
  }
  // mask keys >= N, row max
  float m = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (key >= p.N) s[kt][r] = -INFINITY;
      m = fmaxf(m, s[kt][r]);
    }
  }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  const float mb = m * LOG2E;
  float l = 0.f;
#pragma unroll
  for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = __builtin_amdgcn_exp2f(s[kt][r] * LOG2E - mb);
      s[kt][r] = e;
      l += e;
    }
  }
  l += __shfl_xor(l, 32, 64);

  // O^T[e][q] = sum_key V^T[e][key] P^T[key][q]
  f32x16 oacc[2];
#pragma unroll
  for (int eb = 0; eb < 2; ++eb)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[eb][r] = 0.f;
  const int g = lane >> 4, t = lane & 15;
  const int trow = 4 * (g >> 1) + (t >> 2);        // + 32*kt + 16*s2
  const int tcol = 16 * (g & 1) + 4 * (t & 3);      // + 32*eb
#pragma unroll
  for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const bf16x8 pf = acc_to_frag(s[kt], s2);
#pragma unroll
      for (int eb = 0; eb < 2; ++eb) {
        const bf16x8 vf = lds_tr_frag(imgV, kt * 32 + 16 * s2 + trow, 32 * eb + tcol);
        oacc[eb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[eb], 0, 0, 0);
      }
    }
  }
  if (q < p.N) {
    const float inv = 1.0f / l;
    bf16_t* orow = p.o + (size_t)(row_base + q) * p.d + hh * HD;
#pragma unroll
    for (int eb = 0; eb < 2; ++eb)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int e = 32 * eb + 8 * g4 + 4 * half;
        *reinterpret_cast<uint2*>(orow + e) = make_uint2(pack_bf16x2(oacc[eb][4 * g4] * inv, oacc[eb][4 * g4 + 1] * inv),
                                                         pack_bf16x2(oacc[eb][4 * g4 + 2] * inv, oacc[eb][4 * g4 + 3] * inv));
      }
    if (half == 0 && p.lse != nullptr) p.lse[((size_t)b * p.H + hh) * p.N + q] = m + __logf(l);
  }
}

// ------------------------------------------------------------------------------------------ backward
template <int NT>
__global__ __launch_bounds__(64 * NT) void attn_bwd_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int IMG = NT * 32 * ROWB;
  char* imgK = smem;
  char* imgV = smem + IMG;
  char* imgQ = smem + 2 * IMG;
  char* imgD = smem + 3 * IMG;  // dO
  float* lse_s = reinterpret_cast<float*>(smem + 4 * IMG);
  float* del_s = lse_s + NT * 32;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x / p.H, hh = blockIdx.x - b * p.H;
  const long row_base = (long)b * p.N;
  size_t bytes = (size_t)p.B * p.N * p.ld * 2;
  if (bytes > 0xffffffe0ull) bytes = 0xffffffe0ull;
  size_t bytes_o = (size_t)p.B * p.N * p.d * 2;
  if (bytes_o > 0xffffffe0ull) bytes_o = 0xffffffe0ull;
  const auto srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.qkv), 0, (uint32_t)bytes, 0x00020000);
  const auto srdD = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.d_o), 0, (uint32_t)bytes_o, 0x00020000);
  stage_image<NT, NT>(imgQ, srd, row_base, p.N, p.ld, hh * HD, wave, lane);
  stage_image<NT, NT>(imgK, srd, row_base, p.N, p.ld, p.d + hh * HD, wave, lane);
  stage_image<NT, NT>(imgV, srd, row_base, p.N, p.ld, 2 * p.d + hh * HD, wave, lane);
  stage_image<NT, NT>(imgD, srdD, row_base, p.N, p.d, hh * HD, wave, lane);
  for (int i = threadIdx.x; i < NT * 32; i += 64 * NT)
    lse_s[i] = (i < p.N) ? p.lse[((size_t)b * p.H + hh) * p.N + i] : INFINITY;

  const int ql = lane & 31, half = lane >> 5;
  const int g = lane >> 4, t = lane & 15;
  const int trow = 4 * (g >> 1) + (t >> 2);
  const int tcol = 16 * (g & 1) + 4 * (t & 3);

  // ---- pass A prologue: delta_q = sum_e dO[q][e] * O[q][e] for this wave's queries (O from HBM)
  const int q = wave * 32 + ql;
  float delta = 0.f;
  if (q < p.N) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const size_t off = (size_t)(row_base + q) * p.d + hh * HD + 16 * ks + 8 * half;
      const bf16x8 ov = *reinterpret_cast<const bf16x8*>(p.o + off);
      const bf16x8 dv = *reinterpret_cast<const bf16x8*>(p.d_o + off);
#pragma unroll
      for (int j = 0; j < 8; ++j) delta += bf16_to_f32((bf16_t)ov[j]) * bf16_to_f32((bf16_t)dv[j]);
    }
  }
  delta += __shfl_xor(delta, 32, 64);
  if (half == 0) del_s[q] = delta;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- pass A: this wave owns queries (lane = query).  dQ^T[e][q] = sum_key K^T[e][key] dS^T[key][q]
  {
    bf16x8 qf[4], df[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[ks] = lds_row_frag(imgQ, wave * 32 + ql, 2 * ks + half);
      df[ks] = lds_row_frag(imgD, wave * 32 + ql, 2 * ks + half);
    }
    const float nlse = -lse_s[q];  // +inf rows (q >= N) give p = 0
    f32x16 dq[2];
#pragma unroll
    for (int eb = 0; eb < 2; ++eb)
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[eb][r] = 0.f;
#pragma unroll 1
    for (int kt = 0; kt < NT; ++kt) {
      f32x16 sa, da;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        sa[r] = nlse;
        da[r] = -delta;
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 kf = lds_row_frag(imgK, kt * 32 + ql, 2 * ks + half);
        const bf16x8 vf = lds_row_frag(imgV, kt * 32 + ql, 2 * ks + half);
        sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sa, 0, 0, 0);
        da = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, df[ks], da, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        float pr = __builtin_amdgcn_exp2f(sa[r] * LOG2E);
        if (key >= p.N) pr = 0.f;
        sa[r] = pr * da[r];  // dS^T
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 dsf = acc_to_frag(sa, s2);
#pragma unroll
        for (int eb = 0; eb < 2; ++eb) {
          const bf16x8 ktf = lds_tr_frag(imgK, kt * 32 + 16 * s2 + trow, 32 * eb + tcol);
          dq[eb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf, dsf, dq[eb], 0, 0, 0);
        }
      }
    }
    if (q < p.N) {
      bf16_t* drow = p.dqkv + (size_t)(row_base + q) * p.ld + hh * HD;
#pragma unroll
      for (int eb = 0; eb < 2; ++eb)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int e = 32 * eb + 8 * g4 + 4 * half;
          *reinterpret_cast<uint2*>(drow + e) =
              make_uint2(pack_bf16x2(dq[eb][4 * g4] * p.dq_scale, dq[eb][4 * g4 + 1] * p.dq_scale),
                         pack_bf16x2(dq[eb][4 * g4 + 2] * p.dq_scale, dq[eb][4 * g4 + 3] * p.dq_scale));
        }
    }
  }
  __syncthreads();  // every wave's delta is in del_s

  // ---- pass B: this wave owns keys (lane = key).  S[q][key] non-swapped: rows = queries (registers).
  {
    const int key = wave * 32 + ql;
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kf[ks] = lds_row_frag(imgK, key, 2 * ks + half);
      vf[ks] = lds_row_frag(imgV, key, 2 * ks + half);
    }
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int eb = 0; eb < 2; ++eb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        dk[eb][r] = 0.f;
        dv[eb][r] = 0.f;
      }
#pragma unroll 1
    for (int qt = 0; qt < NT; ++qt) {
      f32x16 sa, da;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qq = qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        sa[r] = -lse_s[qq];
        da[r] = -del_s[qq];
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 qfr = lds_row_frag(imgQ, qt * 32 + ql, 2 * ks + half);
        const bf16x8 dfr = lds_row_frag(imgD, qt * 32 + ql, 2 * ks + half);
        sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr, kf[ks], sa, 0, 0, 0);
        da = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dfr, vf[ks], da, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pr = __builtin_amdgcn_exp2f(sa[r] * LOG2E);  // rows q >= N: lse = +inf -> 0
        sa[r] = pr;
        da[r] = pr * da[r];
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pf = acc_to_frag(sa, s2);
        const bf16x8 dsf = acc_to_frag(da, s2);
#pragma unroll
        for (int eb = 0; eb < 2; ++eb) {
          const bf16x8 dtf = lds_tr_frag(imgD, qt * 32 + 16 * s2 + trow, 32 * eb + tcol);
          const bf16x8 qtf = lds_tr_frag(imgQ, qt * 32 + 16 * s2 + trow, 32 * eb + tcol);
          dv[eb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dtf, pf, dv[eb], 0, 0, 0);
          dk[eb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf, dsf, dk[eb], 0, 0, 0);
        }
      }
    }
    if (key < p.N) {
      bf16_t* krow = p.dqkv + (size_t)(row_base + key) * p.ld + p.d + hh * HD;
      bf16_t* vrow = krow + p.d;
#pragma unroll
      for (int eb = 0; eb < 2; ++eb)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int e = 32 * eb + 8 * g4 + 4 * half;
          *reinterpret_cast<uint2*>(krow + e) = make_uint2(pack_bf16x2(dk[eb][4 * g4], dk[eb][4 * g4 + 1]),
                                                           pack_bf16x2(dk[eb][4 * g4 + 2], dk[eb][4 * g4 + 3]));
          *reinterpret_cast<uint2*>(vrow + e) = make_uint2(pack_bf16x2(dv[eb][4 * g4], dv[eb][4 * g4 + 1]),
                                                           pack_bf16x2(dv[eb][4 * g4 + 2], dv[eb][4 * g4 + 3]));
        }
    }
  }
}



// ------------------------------------------------------------------------------------------------------------
// General kernels: any N <= 608 (19 key tiles: ViT-L/16 at 384^2 has N = 577), head_dim 48 or 64 (48 = every CaiT size;
// rows are zero-padded to 64 inside LDS, the 16 padding columns cost MFMA issue slots only in the P.V products).
// Workgroup = min(8, tiles) waves per (batch, head); q-blocks / key-blocks are dealt round-robin to the waves.
//   forward : K and V images resident (2 x tiles x 4 KB), ONLINE softmax over chunks of 4 key tiles (running max / sum,
//             O rescaled per chunk) so the score registers stay at 64 regardless of N;
//   backward: pass A with K,V resident (dQ, delta), barrier, the SAME LDS is re-staged with Q,dO for pass B (dK, dV) -
//             half the LDS of the N<=256 kernel above, so two workgroups fit a CU at N = 197.
template <int NTV>  // stage with run-time tile / wave counts; chunks >= hd/8 and rows >= N are zero-filled
__device__ __forceinline__ void stage_image_rt(char* img, __amdgpu_buffer_rsrc_t srd, long row_base, int N, int ld, int col0, int wave,
                                               int nwv, int lane, int nt, int hd) {
  const int lrow = lane >> 3, pc = lane & 7;
  for (int inst = wave; inst < nt * 4; inst += nwv) {
    const int t = inst * 8 + lrow;
    const int c = pc ^ rot3(t);
    uint32_t voff = 0xfffffff0u;
    if (t < N && c * 8 < hd) voff = (uint32_t)(((size_t)(row_base + t) * ld + col0 + c * 8) * 2);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)(img + inst * 1024), 16, voff, 0, 0, 0);
  }
}

struct AttnParams2 {
  AttnParams a;
  int nt;   // ceil(N / 32)
  int hd;   // 48 or 64
};

__device__ __forceinline__ bf16x8 load_row_frag_global(const bf16_t* base, size_t row, int ld, int col, bool valid) {
  bf16x8 z = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  if (valid) z = *reinterpret_cast<const bf16x8*>(base + row * ld + col);
  return z;
}

__global__ __launch_bounds__(512) void attn_fwd2_kernel(const AttnParams2 pp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const AttnParams& p = pp.a;
  const int NT = pp.nt, hd = pp.hd;
  const int IMG = NT * 32 * ROWB;
  char* imgK = smem;
  char* imgV = smem + IMG;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwv = blockDim.x >> 6;
  const int b = blockIdx.x / p.H, hh = blockIdx.x - b * p.H;
  const long row_base = (long)b * p.N;
  size_t bytes = (size_t)p.B * p.N * p.ld * 2;
  if (bytes > 0xffffffe0ull) bytes = 0xffffffe0ull;
  const auto srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.qkv), 0, (uint32_t)bytes, 0x00020000);
  stage_image_rt<0>(imgK, srd, row_base, p.N, p.ld, p.d + hh * hd, wave, nwv, lane, NT, hd);
  stage_image_rt<0>(imgV, srd, row_base, p.N, p.ld, 2 * p.d + hh * hd, wave, nwv, lane, NT, hd);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int ql = lane & 31, half = lane >> 5;
  const int g = lane >> 4, t = lane & 15;
  const int trow = 4 * (g >> 1) + (t >> 2);
  const int tcol = 16 * (g & 1) + 4 * (t & 3);
  constexpr int KC = 4;
  for (int qb = wave; qb < NT; qb += nwv) {
    const int q = qb * 32 + ql;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      qf[ks] = load_row_frag_global(p.qkv, (size_t)(row_base + q), p.ld, hh * hd + 16 * ks + 8 * half, q < p.N && 16 * ks < hd);
    float m = -INFINITY, l = 0.f;
    f32x16 oacc[2];
#pragma unroll
    for (int eb = 0; eb < 2; ++eb)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[eb][r] = 0.f;
    for (int c0 = 0; c0 < NT; c0 += KC) {
      f32x16 s[KC];
      float cmax = -INFINITY;
#pragma unroll
      for (int j = 0; j < KC; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[j][r] = -INFINITY;
        const int kt = c0 + j;
        if (kt < NT) {
#pragma unroll
          for (int r = 0; r < 16; ++r) s[j][r] = 0.f;
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            if (16 * ks < hd) {
              const bf16x8 kf = lds_row_frag(imgK, kt * 32 + ql, 2 * ks + half);
              s[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[j], 0, 0, 0);
            }
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (key >= p.N) s[j][r] = -INFINITY;
            cmax = fmaxf(cmax, s[j][r]);
          }
        }
      }
      cmax = fmaxf(cmax, __shfl_xor(cmax, 32, 64));
      const float m_new = fmaxf(m, cmax);  // finite: every chunk holds at least one valid key
      const float alpha = __builtin_amdgcn_exp2f((m - m_new) * LOG2E);
      const float mb = m_new * LOG2E;
      l *= alpha;
#pragma unroll
      for (int eb = 0; eb < 2; ++eb)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[eb][r] *= alpha;
#pragma unroll
      for (int j = 0; j < KC; ++j) {
        const int kt = c0 + j;
        if (kt < NT) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float e = __builtin_amdgcn_exp2f(s[j][r] * LOG2E - mb);
            s[j][r] = e;
            l += e;
          }
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 pf = acc_to_frag(s[j], s2);
#pragma unroll
            for (int eb = 0; eb < 2; ++eb) {
              const bf16x8 vf = lds_tr_frag(imgV, kt * 32 + 16 * s2 + trow, 32 * eb + tcol);
              oacc[eb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[eb], 0, 0, 0);
            }
          }
        }
      }
      m = m_new;
    }
    l += __shfl_xor(l, 32, 64);
    if (q < p.N) {
      const float inv = 1.0f / l;
      bf16_t* orow = p.o + (size_t)(row_base + q) * p.d + hh * hd;
#pragma unroll
      for (int eb = 0; eb < 2; ++eb)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int e = 32 * eb + 8 * g4 + 4 * half;
          if (e < hd)
            *reinterpret_cast<uint2*>(orow + e) = make_uint2(pack_bf16x2(oacc[eb][4 * g4] * inv, oacc[eb][4 * g4 + 1] * inv),
                                                             pack_bf16x2(oacc[eb][4 * g4 + 2] * inv, oacc[eb][4 * g4 + 3] * inv));
        }
      if (half == 0 && p.lse != nullptr) p.lse[((size_t)b * p.H + hh) * p.N + q] = m + __logf(l);
    }
  }
}

__global__ __launch_bounds__(512) void attn_bwd2_kernel(const AttnParams2 pp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const AttnParams& p = pp.a;
  const int NT = pp.nt, hd = pp.hd;
  const int IMG = NT * 32 * ROWB;
  char* img0 = smem;        // pass A: K      pass B: Q
  char* img1 = smem + IMG;  // pass A: V      pass B: dO
  float* lse_s = reinterpret_cast<float*>(smem + 2 * IMG);
  float* del_s = lse_s + NT * 32;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwv = blockDim.x >> 6;
  const int b = blockIdx.x / p.H, hh = blockIdx.x - b * p.H;
  const long row_base = (long)b * p.N;
  size_t bytes = (size_t)p.B * p.N * p.ld * 2;
  if (bytes > 0xffffffe0ull) bytes = 0xffffffe0ull;
  size_t bytes_o = (size_t)p.B * p.N * p.d * 2;
  if (bytes_o > 0xffffffe0ull) bytes_o = 0xffffffe0ull;
  const auto srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.qkv), 0, (uint32_t)bytes, 0x00020000);
  const auto srdD = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.d_o), 0, (uint32_t)bytes_o, 0x00020000);
  stage_image_rt<0>(img0, srd, row_base, p.N, p.ld, p.d + hh * hd, wave, nwv, lane, NT, hd);
  stage_image_rt<0>(img1, srd, row_base, p.N, p.ld, 2 * p.d + hh * hd, wave, nwv, lane, NT, hd);
  for (int i = threadIdx.x; i < NT * 32; i += blockDim.x)
    lse_s[i] = (i < p.N) ? p.lse[((size_t)b * p.H + hh) * p.N + i] : INFINITY;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int ql = lane & 31, half = lane >> 5;
  const int g = lane >> 4, t = lane & 15;
  const int trow = 4 * (g >> 1) + (t >> 2);
  const int tcol = 16 * (g & 1) + 4 * (t & 3);

  // ---- pass A: queries on the lane; K, V resident
  for (int qb = wave; qb < NT; qb += nwv) {
    const int q = qb * 32 + ql;
    bf16x8 qf[4], df[4];
    float delta = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bool ok = q < p.N && 16 * ks < hd;
      qf[ks] = load_row_frag_global(p.qkv, (size_t)(row_base + q), p.ld, hh * hd + 16 * ks + 8 * half, ok);
      df[ks] = load_row_frag_global(p.d_o, (size_t)(row_base + q), p.d, hh * hd + 16 * ks + 8 * half, ok);
      const bf16x8 ov = load_row_frag_global(p.o, (size_t)(row_base + q), p.d, hh * hd + 16 * ks + 8 * half, ok);
#pragma unroll
      for (int j = 0; j < 8; ++j) delta += bf16_to_f32((bf16_t)ov[j]) * bf16_to_f32((bf16_t)df[ks][j]);
    }
    delta += __shfl_xor(delta, 32, 64);
    if (half == 0) del_s[q] = delta;
    const float nlse = -lse_s[q];
    f32x16 dq[2];
#pragma unroll
    for (int eb = 0; eb < 2; ++eb)
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[eb][r] = 0.f;
#pragma unroll 1
    for (int kt = 0; kt < NT; ++kt) {
      f32x16 sa, da;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        sa[r] = nlse;
        da[r] = -delta;
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (16 * ks < hd) {
          const bf16x8 kf = lds_row_frag(img0, kt * 32 + ql, 2 * ks + half);
          const bf16x8 vf = lds_row_frag(img1, kt * 32 + ql, 2 * ks + half);
          sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sa, 0, 0, 0);
          da = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, df[ks], da, 0, 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        float pr = __builtin_amdgcn_exp2f(sa[r] * LOG2E);
        if (key >= p.N) pr = 0.f;
        sa[r] = pr * da[r];
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 dsf = acc_to_frag(sa, s2);
#pragma unroll
        for (int eb = 0; eb < 2; ++eb) {
          const bf16x8 ktf = lds_tr_frag(img0, kt * 32 + 16 * s2 + trow, 32 * eb + tcol);
          dq[eb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf, dsf, dq[eb], 0, 0, 0);
        }
      }
    }
    if (q < p.N) {
      bf16_t* drow = p.dqkv + (size_t)(row_base + q) * p.ld + hh * hd;
#pragma unroll
      for (int eb = 0; eb < 2; ++eb)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int e = 32 * eb + 8 * g4 + 4 * half;
          if (e < hd)
            *reinterpret_cast<uint2*>(drow + e) =
                make_uint2(pack_bf16x2(dq[eb][4 * g4] * p.dq_scale, dq[eb][4 * g4 + 1] * p.dq_scale),
                           pack_bf16x2(dq[eb][4 * g4 + 2] * p.dq_scale, dq[eb][4 * g4 + 3] * p.dq_scale));
        }
    }
  }
  __syncthreads();  // pass A reads of K,V are done everywhere; delta is complete
  stage_image_rt<0>(img0, srd, row_base, p.N, p.ld, hh * hd, wave, nwv, lane, NT, hd);
  stage_image_rt<0>(img1, srdD, row_base, p.N, p.d, hh * hd, wave, nwv, lane, NT, hd);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- pass B: keys on the lane; Q, dO resident; this wave's K, V fragments from HBM
  for (int kb = wave; kb < NT; kb += nwv) {
    const int key = kb * 32 + ql;
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bool ok = key < p.N && 16 * ks < hd;
      kf[ks] = load_row_frag_global(p.qkv, (size_t)(row_base + key), p.ld, p.d + hh * hd + 16 * ks + 8 * half, ok);
      vf[ks] = load_row_frag_global(p.qkv, (size_t)(row_base + key), p.ld, 2 * p.d + hh * hd + 16 * ks + 8 * half, ok);
    }
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int eb = 0; eb < 2; ++eb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        dk[eb][r] = 0.f;
        dv[eb][r] = 0.f;
      }
#pragma unroll 1
    for (int qt = 0; qt < NT; ++qt) {
      f32x16 sa, da;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qq = qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        sa[r] = -lse_s[qq];
        da[r] = -del_s[qq];
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (16 * ks < hd) {
          const bf16x8 qfr = lds_row_frag(img0, qt * 32 + ql, 2 * ks + half);
          const bf16x8 dfr = lds_row_frag(img1, qt * 32 + ql, 2 * ks + half);
          sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr, kf[ks], sa, 0, 0, 0);
          da = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dfr, vf[ks], da, 0, 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pr = __builtin_amdgcn_exp2f(sa[r] * LOG2E);
        sa[r] = pr;
        da[r] = pr * da[r];
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pf = acc_to_frag(sa, s2);
        const bf16x8 dsf = acc_to_frag(da, s2);
#pragma unroll
        for (int eb = 0; eb < 2; ++eb) {
          const bf16x8 dtf = lds_tr_frag(img1, qt * 32 + 16 * s2 + trow, 32 * eb + tcol);
          const bf16x8 qtf = lds_tr_frag(img0, qt * 32 + 16 * s2 + trow, 32 * eb + tcol);
          dv[eb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dtf, pf, dv[eb], 0, 0, 0);
          dk[eb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf, dsf, dk[eb], 0, 0, 0);
        }
      }
    }
    if (key < p.N) {
      bf16_t* krow = p.dqkv + (size_t)(row_base + key) * p.ld + p.d + hh * hd;
      bf16_t* vrow = krow + p.d;
#pragma unroll
      for (int eb = 0; eb < 2; ++eb)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int e = 32 * eb + 8 * g4 + 4 * half;
          if (e < hd) {
            *reinterpret_cast<uint2*>(krow + e) = make_uint2(pack_bf16x2(dk[eb][4 * g4], dk[eb][4 * g4 + 1]),
                                                             pack_bf16x2(dk[eb][4 * g4 + 2], dk[eb][4 * g4 + 3]));
            *reinterpret_cast<uint2*>(vrow + e) = make_uint2(pack_bf16x2(dv[eb][4 * g4], dv[eb][4 * g4 + 1]),
                                                             pack_bf16x2(dv[eb][4 * g4 + 2], dv[eb][4 * g4 + 3]));
          }
        }
    }
  }
}

}  // namespace

static bool attn_force_general() {
  static const bool f = [] { const char* e = getenv("SAVIT_ATTN_GENERAL"); return e && atoi(e) != 0; }();
  return f;
}

#define ATTN_DISPATCH(KERNEL, LDS_EXPR)                                                                        \
  switch (nt) {                                                                                                \
    case 1: ATTN_CASE(KERNEL, 1, LDS_EXPR)                                                                     \
    case 2: ATTN_CASE(KERNEL, 2, LDS_EXPR)                                                                     \
    case 3: ATTN_CASE(KERNEL, 3, LDS_EXPR)                                                                     \
    case 4: ATTN_CASE(KERNEL, 4, LDS_EXPR)                                                                     \
    case 5: ATTN_CASE(KERNEL, 5, LDS_EXPR)                                                                     \
    case 6: ATTN_CASE(KERNEL, 6, LDS_EXPR)                                                                     \
    case 7: ATTN_CASE(KERNEL, 7, LDS_EXPR)                                                                     \
    case 8: ATTN_CASE(KERNEL, 8, LDS_EXPR)                                                                     \
    default: return SAVIT_EINVAL;                                                                              \
  }
#define ATTN_CASE(KERNEL, NTV, LDS_EXPR)                                                                       \
  {                                                                                                            \
    constexpr int NT = NTV;                                                                                    \
    const size_t lds = (LDS_EXPR);                                                                             \
    auto kfn = KERNEL<NTV>;                                                                                    \
    if (lds > 48 * 1024) {                                                                                     \
      hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      if (e != hipSuccess) return (int)e;                                                                      \
    }                                                                                                          \
    hipLaunchKernelGGL(kfn, dim3(B * H), dim3(64 * NTV), lds, (hipStream_t)stream, p);                         \
  } break;

extern "C" int savit_attention_fwd(const void* qkv, void* o, float* lse, int B, int N, int H, int head_dim, int ld_qkv,
                                   void* stream) {
  SAVIT_CHECK_ARG(qkv && o && B >= 0 && N > 0 && H > 0);
  SAVIT_CHECK_ARG((head_dim == 64 || head_dim == 48) && N <= 608 && ld_qkv >= 3 * H * head_dim && ld_qkv % 8 == 0);
  SAVIT_CHECK_ARG(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)o % 16) == 0);
  if (B == 0) return SAVIT_OK;
  AttnParams p{};
  p.qkv = (const bf16_t*)qkv; p.o = (bf16_t*)o; p.lse = lse;
  p.B = B; p.N = N; p.H = H; p.ld = ld_qkv; p.d = H * head_dim;
  const int nt = (N + 31) / 32;
  if (head_dim != HD || nt > 8 || attn_force_general()) {
    AttnParams2 pp{p, nt, head_dim};
    const size_t lds = (size_t)2 * nt * 32 * ROWB;
    hipError_t e = hipFuncSetAttribute((const void*)attn_fwd2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(attn_fwd2_kernel, dim3(B * H), dim3(64 * (nt < 8 ? nt : 8)), lds, (hipStream_t)stream, pp);
    SAVIT_LAUNCH_RET();
  }
  ATTN_DISPATCH(attn_fwd_kernel, (size_t)2 * NT * 32 * ROWB)
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_attention_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv, int B, int N,
                                   int H, int head_dim, int ld_qkv, float dq_scale, void* stream) {
  SAVIT_CHECK_ARG(qkv && o && d_o && lse && dqkv && B >= 0 && N > 0 && H > 0);
  SAVIT_CHECK_ARG((head_dim == 64 || head_dim == 48) && N <= 608 && ld_qkv >= 3 * H * head_dim && ld_qkv % 8 == 0);
  SAVIT_CHECK_ARG(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)o % 16) == 0 && ((uintptr_t)d_o % 16) == 0 && ((uintptr_t)dqkv % 16) == 0);
  if (B == 0) return SAVIT_OK;
  AttnParams p{};
  p.qkv = (const bf16_t*)qkv; p.o = (bf16_t*)const_cast<void*>(o); p.lse = const_cast<float*>(lse);
  p.d_o = (const bf16_t*)d_o; p.dqkv = (bf16_t*)dqkv;
  p.B = B; p.N = N; p.H = H; p.ld = ld_qkv; p.d = H * head_dim; p.dq_scale = dq_scale;
  const int nt = (N + 31) / 32;
  if (head_dim != HD || nt > 8 || attn_force_general()) {
    AttnParams2 pp{p, nt, head_dim};
    const size_t lds = (size_t)2 * nt * 32 * ROWB + (size_t)2 * nt * 32 * sizeof(float);
    hipError_t e = hipFuncSetAttribute((const void*)attn_bwd2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(attn_bwd2_kernel, dim3(B * H), dim3(64 * (nt < 8 ? nt : 8)), lds, (hipStream_t)stream, pp);
    SAVIT_LAUNCH_RET();
  }
  ATTN_DISPATCH(attn_bwd_kernel, (size_t)4 * NT * 32 * ROWB + (size_t)2 * NT * 32 * sizeof(float))
  SAVIT_LAUNCH_RET();
}
