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
#include "th_rows.h"

namespace {

// MFMA groups of the backward passes issue at raised wave priority (the SIMD partner's LDS requests otherwise interleave with them:
// 140.1 -> 137.4 us at DeiT-B's layer, same box)
#define ATTN_PRIO(x) __builtin_amdgcn_s_setprio(x)

constexpr int HD = 64;              // head dim
constexpr int ROWB = HD * 2;        // LDS row bytes

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

// The same transposed fragment for kernels that keep an LDS-DMA in flight across their passes: hipcc (ROCm 7.2) puts
// `s_waitcnt vmcnt(0)` in front of the ds_read_tr16 BUILTIN whenever an LDS-DMA may be outstanding (it cannot prove that the two do not
// alias), which makes the prefetch of the next item land before the pass starts - no overlap (the persistent backward of round 2 and
// the first one of round 4 both ran at the one-item kernel's speed for this reason).  The asm form is invisible to that pass;
// completion is waited for by lds_tr_wait (an `s_waitcnt lgkmcnt(0)` that names every destination "+v", so no consumer or copy can be
// scheduled above it; the compiler's own counted lgkmcnt waits stay correct because LDS returns in order - they can only over-wait).
struct TrFrag {
  bf16x4 lo, hi;
};
__device__ __forceinline__ uint32_t lds_addr32(const char* p) { return (uint32_t)(uintptr_t)((__attribute__((address_space(3))) const char*)p); }
// a_lo / a_hi: LDS byte addresses of rows row0 / row0 + 8 (tr_lane_off below + image + 32-row tile); OFF: compile-time byte offset
// on top (another image, the second k-step), so that one pair of address registers serves every fragment of a tile column
template <int OFF>
__device__ __forceinline__ void lds_tr_issue(TrFrag& f, uint32_t a_lo, uint32_t a_hi) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.lo) : "v"(a_lo), "n"(OFF));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.hi) : "v"(a_hi), "n"(OFF));
}
// byte offset inside an image of this lane's piece of the transposed fragment at rows trow (+ 8 if hi), columns col..col+3; adding
// multiples of 16 rows does not change the swizzle (rot3 looks at row bits 1..3)
__device__ __forceinline__ uint32_t tr_lane_off(int trow, int col, bool hi) {
  const int r = trow + (hi ? 8 : 0);
  return (uint32_t)(r * ROWB + (((col >> 3) ^ rot3(r)) << 4) + ((col & 7) << 1));
}
__device__ __forceinline__ void lds_tr_wait(TrFrag& a, TrFrag& b, TrFrag& c, TrFrag& d) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a.lo), "+v"(a.hi), "+v"(b.lo), "+v"(b.hi), "+v"(c.lo), "+v"(c.hi), "+v"(d.lo), "+v"(d.hi));
}
// 16-byte LDS reads in the same style (OFF: compile-time byte offset)
__device__ __forceinline__ void lds_read_f4(f32x4& v, uint32_t addr, int off) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(off));
}
__device__ __forceinline__ void lds_f4_wait(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ void lds_read_u4(uint4& v, uint32_t addr) {
  u32x4 t;
  asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(addr));
  v = make_uint4(t[0], t[1], t[2], t[3]);
}
__device__ __forceinline__ bf16x8 tr_join(const TrFrag& f) { return __builtin_shufflevector(f.lo, f.hi, 0, 1, 2, 3, 4, 5, 6, 7); }

// registers 8s..8s+7 of a 32x32 fp32 accumulator -> bf16 B-operand fragment of k-step s
__device__ __forceinline__ bf16x8 acc_to_frag(const f32x16& a, int s) {
  union { uint32_t u[4]; bf16x8 v; } r;
#pragma unroll
  for (int j = 0; j < 4; ++j) r.u[j] = pack_bf16x2(a[8 * s + 2 * j], a[8 * s + 2 * j + 1]);
  return r.v;
}

// One row of a swapped-layout output tile pair (lane = (row, half); acc[eb] registers 4g .. 4g+3 = columns 32*eb + 8*g + 4*half ..)
// scaled, rounded to bf16 and stored as 16-byte pieces: the two lanes of a row trade 8-byte pieces first (merge_row_halves,
// common.h; 8-byte stores made the epilogues store-issue-bound: attention backward 146 -> 139 us at DeiT-B's layer).  Columns
// >= hd are skipped (hd is a multiple of 16).  Both lanes of a row must be active.
__device__ __forceinline__ void store_row_tile(bf16_t* row, const f32x16 (&acc)[2], int half, int hd, float sc) {
#pragma unroll
  for (int eb = 0; eb < 2; ++eb)
#pragma unroll
    for (int g4 = 0; g4 < 4; g4 += 2) {
      const int e0 = 32 * eb + 8 * g4;
      if (e0 < hd)
        *reinterpret_cast<uint4*>(row + e0 + 8 * half) = merge_row_halves(
            make_uint2(pack_bf16x2(acc[eb][4 * g4] * sc, acc[eb][4 * g4 + 1] * sc), pack_bf16x2(acc[eb][4 * g4 + 2] * sc, acc[eb][4 * g4 + 3] * sc)),
            make_uint2(pack_bf16x2(acc[eb][4 * g4 + 4] * sc, acc[eb][4 * g4 + 5] * sc), pack_bf16x2(acc[eb][4 * g4 + 6] * sc, acc[eb][4 * g4 + 7] * sc)));
    }
}

// The same tile through a 4 KB LDS image private to the wave (128-byte rows, the images' swizzle), read back as WHOLE rows - lane =
// (row 8j + lane/8, 16-byte chunk lane%8) - and stored 8 rows x 128 B per instruction.  store_row_tile's instructions touch 32 cache
// lines each (32 B per row); the L1's tag pipe takes a few cycles per line, and in the persistent backward - which issues its
// stores while the partner waves compute - that was 4.3 us per pass for the second wave of every SIMD (round 4, s_memrealtime stamps).
// rows_valid: rows of the tile inside the matrix.  The reads are inline asm (see TrFrag); LDS serves a wave's operations in order.
__device__ __forceinline__ void store_tile_rows(char* scr, bf16_t* tile0, size_t ld, int rows_valid, const f32x16 (&acc)[2], int ql, int half,
                                                int lane, float sc) {
#pragma unroll
  for (int eb = 0; eb < 2; ++eb)
#pragma unroll
    for (int g4 = 0; g4 < 4; g4 += 2)
      *reinterpret_cast<uint4*>(scr + img_off(ql, 4 * eb + g4 + half)) = merge_row_halves(
          make_uint2(pack_bf16x2(acc[eb][4 * g4] * sc, acc[eb][4 * g4 + 1] * sc), pack_bf16x2(acc[eb][4 * g4 + 2] * sc, acc[eb][4 * g4 + 3] * sc)),
          make_uint2(pack_bf16x2(acc[eb][4 * g4 + 4] * sc, acc[eb][4 * g4 + 5] * sc), pack_bf16x2(acc[eb][4 * g4 + 6] * sc, acc[eb][4 * g4 + 7] * sc)));
  uint4 v[4];
  const uint32_t a = lds_addr32(scr);
#pragma unroll
  for (int j = 0; j < 4; ++j) lds_read_u4(v[j], a + (uint32_t)img_off(8 * j + (lane >> 3), lane & 7));
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = 8 * j + (lane >> 3);
    if (row < rows_valid) *reinterpret_cast<uint4*>(tile0 + (size_t)row * ld + 8 * (lane & 7)) = v[j];
  }
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

// The same with a compile-time instruction count per wave (NW waves, NT * 4 instructions, NW divides them): straight-line code, so
// hipcc can COUNT the LDS-DMA instructions behind an earlier global load and wait for that load alone (after a loop of unknown
// trip count its vmcnt waits drain the LDS-DMA as well)
template <int NW, int NT>
__device__ __forceinline__ void stage_image_u(char* img, __amdgpu_buffer_rsrc_t srd, long row_base, int N, int ld, int col0, int wave,
                                              int lane) {
  constexpr int INSTR = NT * 32 / 8;
  static_assert(INSTR % NW == 0, "whole instructions per wave");
  const int lrow = lane >> 3, pc = lane & 7;
#pragma unroll
  for (int k = 0; k < INSTR / NW; ++k) {
    const int inst = wave * (INSTR / NW) + k;  // a wave stages its OWN 32-row block: after its own vmcnt(0) it may read those rows
    const int t = inst * 8 + lrow;
    const int c = pc ^ rot3(t);
    uint32_t voff = 0xfffffff0u;
    if (t < N) voff = (uint32_t)(((size_t)(row_base + t) * ld + col0 + c * 8) * 2);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)(img + inst * 1024), 16, voff, 0, 0, 0);
  }
}

__device__ __forceinline__ bf16x8 load_row_frag_global(const bf16_t* base, size_t row, int ld, int col, bool valid) {
  bf16x8 z = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  if (valid) z = *reinterpret_cast<const bf16x8*>(base + row * ld + col);
  return z;
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
// Persistent: each workgroup walks over (batch, head) items blockIdx.x, += gridDim.x with TWO K/V image pairs in LDS, so the
// LDS-DMA of the next item runs under the MFMA/softmax work of the current one.  (Measured before: loads + stores alone 34 us,
// compute alone 54 us, together 63 us per launch - every workgroup of a launch is in the same phase, so co-resident workgroups
// do not cover each other; two 4-wave workgroups per CU ran in the same 63 us.)  Round 4: O leaves as whole 128-byte rows through a
// fifth LDS image (store_tile_rows; 43.8 -> 41.7 us per launch at DeiT-B's layer - 32-byte row pieces cost the L1 32 lines per store).
template <int NT>
__global__ __launch_bounds__(64 * NT) void attn_fwd_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int IMG = NT * 32 * ROWB;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nitems = p.B * p.H;
  size_t bytes = (size_t)p.B * p.N * p.ld * 2;
  if (bytes > 0xffffffe0ull) bytes = 0xffffffe0ull;
  const auto srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.qkv), 0, (uint32_t)bytes, 0x00020000);
  const int ql = lane & 31, half = lane >> 5;
  const int q = wave * 32 + ql;
  const int g = lane >> 4, t = lane & 15;
  const int trow = 4 * (g >> 1) + (t >> 2);        // + 32*kt + 16*s2
  const int tcol = 16 * (g & 1) + 4 * (t & 3);      // + 32*eb

  auto stage_item = [&](int item, char* buf) {
    const int b = item / p.H, hh = item - b * p.H;
    const long row_base = (long)b * p.N;
    stage_image<NT, NT>(buf, srd, row_base, p.N, p.ld, p.d + hh * HD, wave, lane);
    stage_image<NT, NT>(buf + IMG, srd, row_base, p.N, p.ld, 2 * p.d + hh * HD, wave, lane);
  };
  // Q fragments of this wave's 32 queries straight from HBM (B operand: lane = (q, half), 8 consecutive e)
  bf16x8 qf[4];
  auto load_q = [&](int item) {
    const int b = item / p.H, hh = item - b * p.H;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      if (q < p.N) qf[ks] = *reinterpret_cast<const bf16x8*>(p.qkv + (size_t)((long)b * p.N + q) * p.ld + hh * HD + 16 * ks + 8 * half);
    }
  };

  int item = blockIdx.x;
  if (item >= nitems) return;
  stage_item(item, smem);
  load_q(item);
  int cur = 0;
#pragma unroll 1
  for (; item < nitems; item += gridDim.x) {
    // this item's images have landed (every wave's share) and every wave is done reading the other pair
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int nxt = item + gridDim.x;
    char* imgK = smem + cur * 2 * IMG;
    char* imgV = imgK + IMG;
    if (nxt < nitems) stage_item(nxt, smem + (cur ^ 1) * 2 * IMG);
    const int b = item / p.H, hh = item - b * p.H;
    const long row_base = (long)b * p.N;

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
    if (nxt < nitems) load_q(nxt);  // qf is dead: the next item's queries arrive under the softmax
    // mask keys >= N, row max
    // only the last key tile can hold keys >= N (NT = ceil(N / 32))
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = (NT - 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (key >= p.N) s[NT - 1][r] = -INFINITY;
    }
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, s[kt][r]);
    }
    m = half_max(m);
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
    l = half_sum(l);

    // O^T[e][q] = sum_key V^T[e][key] P^T[key][q]
    f32x16 oacc[2];
#pragma unroll
    for (int eb = 0; eb < 2; ++eb)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[eb][r] = 0.f;
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
    {
      const float inv = 1.0f / l;  // rows q >= N: finite (every key of the last tile masked alike), never stored
      store_tile_rows(smem + 4 * IMG + wave * (32 * ROWB), p.o + (size_t)(row_base + wave * 32) * p.d + hh * HD, p.d, p.N - wave * 32, oacc, ql,
                      half, lane, inv);
      if (q < p.N && half == 0 && p.lse != nullptr) p.lse[((size_t)b * p.H + hh) * p.N + q] = m + __logf(l);
    }
    cur ^= 1;
  }
}

// Round 6: the same kernel with a LOADER wave (NT <= 7: eight waves, two per SIMD).  The K / V images of the next item are requested
// by one extra wave that does nothing else (56 LDS-DMA instructions per item: no registers), so the compute waves no longer queue them
// in front of their score MFMAs - a CU takes one vector-memory instruction per ~50 cycles (see attn_bwd_persl_kernel).  One bare barrier
// per item: the loader reaches it behind its own vmcnt(0) (this item's images landed), the compute waves behind their last read of the
// other image pair.  Same arithmetic: bitwise the results of attn_fwd_kernel.
template <int NT>
__global__ __launch_bounds__(64 * (NT + 1)) void attn_fwdl_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int IMG = NT * 32 * ROWB;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nitems = p.B * p.H;
  size_t bytes = (size_t)p.B * p.N * p.ld * 2;
  if (bytes > 0xffffffe0ull) bytes = 0xffffffe0ull;
  const auto srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.qkv), 0, (uint32_t)bytes, 0x00020000);
  const int ql = lane & 31, half = lane >> 5;
  const int q = wave * 32 + ql;
  const int g = lane >> 4, t = lane & 15;
  const int trow = 4 * (g >> 1) + (t >> 2);        // + 32*kt + 16*s2
  const int tcol = 16 * (g & 1) + 4 * (t & 3);      // + 32*eb

  const bool loader = wave == NT;
  auto stage_item = [&](int item, char* buf) {  // (loader) both images, every instruction
    const int b = item / p.H, hh = item - b * p.H;
    const long row_base = (long)b * p.N;
    stage_image<1, NT>(buf, srd, row_base, p.N, p.ld, p.d + hh * HD, 0, lane);
    stage_image<1, NT>(buf + IMG, srd, row_base, p.N, p.ld, 2 * p.d + hh * HD, 0, lane);
  };
  // Q fragments of this wave's 32 queries straight from HBM (B operand: lane = (q, half), 8 consecutive e)
  bf16x8 qf[4];
  auto load_q = [&](int item) {
    const int b = item / p.H, hh = item - b * p.H;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      if (q < p.N) qf[ks] = *reinterpret_cast<const bf16x8*>(p.qkv + (size_t)((long)b * p.N + q) * p.ld + hh * HD + 16 * ks + 8 * half);
    }
  };

  int item = blockIdx.x;
  if (item >= nitems) return;
  int cur = 0;
  if (loader) {
    stage_item(item, smem);
#pragma unroll 1
    for (; item < nitems; item += gridDim.x) {
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");  // this item's images landed; the other pair is free
      const int nxt = item + gridDim.x;
      if (nxt < nitems) stage_item(nxt, smem + (cur ^ 1) * 2 * IMG);
      cur ^= 1;
    }
    return;
  }
  load_q(item);
#pragma unroll 1
  for (; item < nitems; item += gridDim.x) {
    // this item's images have landed (the loader waited for them) and every wave is done reading the other pair.  Bare: the wait for
    // this wave's own Q rows is the compiler's, in front of their first use - the barrier does not wait for the previous item's stores
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const int nxt = item + gridDim.x;
    char* imgK = smem + cur * 2 * IMG;
    char* imgV = imgK + IMG;
    const int b = item / p.H, hh = item - b * p.H;
    const long row_base = (long)b * p.N;

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
    if (nxt < nitems) load_q(nxt);  // qf is dead: the next item's queries arrive under the softmax
    // mask keys >= N, row max
    // only the last key tile can hold keys >= N (NT = ceil(N / 32))
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = (NT - 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (key >= p.N) s[NT - 1][r] = -INFINITY;
    }
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, s[kt][r]);
    }
    m = half_max(m);
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
    l = half_sum(l);

    // O^T[e][q] = sum_key V^T[e][key] P^T[key][q]
    f32x16 oacc[2];
#pragma unroll
    for (int eb = 0; eb < 2; ++eb)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[eb][r] = 0.f;
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
    {
      const float inv = 1.0f / l;  // rows q >= N: finite (every key of the last tile masked alike), never stored
      store_tile_rows(smem + 4 * IMG + wave * (32 * ROWB), p.o + (size_t)(row_base + wave * 32) * p.d + hh * HD, p.d, p.N - wave * 32, oacc, ql,
                      half, lane, inv);
      if (q < p.N && half == 0 && p.lse != nullptr) p.lse[((size_t)b * p.H + hh) * p.N + q] = m + __logf(l);
    }
    cur ^= 1;
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
  {
  stage_image<NT, NT>(imgQ, srd, row_base, p.N, p.ld, hh * HD, wave, lane);
  stage_image<NT, NT>(imgK, srd, row_base, p.N, p.ld, p.d + hh * HD, wave, lane);
  stage_image<NT, NT>(imgV, srd, row_base, p.N, p.ld, 2 * p.d + hh * HD, wave, lane);
  stage_image<NT, NT>(imgD, srdD, row_base, p.N, p.d, hh * HD, wave, lane);
  }
  // lse_s holds -LSE * log2(e), so P = exp2(fma(S, log2 e, lse_s)); rows q >= N hold -inf (P = 0)
  for (int i = threadIdx.x; i < NT * 32; i += 64 * NT)
    lse_s[i] = (i < p.N) ? -LOG2E * p.lse[((size_t)b * p.H + hh) * p.N + i] : -INFINITY;

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
  delta = half_sum(delta);
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
    const float nlse2 = lse_s[q];  // -LSE * log2 e; -inf for rows q >= N (P = 0)
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 dq[2];
#pragma unroll
    for (int eb = 0; eb < 2; ++eb)
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[eb][r] = 0.f;
#pragma unroll 1
    for (int kt = 0; kt < NT; ++kt) {
      // accumulators start from the MFMA's inline-constant zero C operand (no per-tile register initialisation); the LSE /
      // delta offsets fold into the exp argument and the dS product: this loop is VALU-bound, every instruction per score counts
      f32x16 sa = zero16, da = zero16;
      ATTN_PRIO(1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 kf = lds_row_frag(imgK, kt * 32 + ql, 2 * ks + half);
        const bf16x8 vf = lds_row_frag(imgV, kt * 32 + ql, 2 * ks + half);
        sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sa, 0, 0, 0);
        da = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, df[ks], da, 0, 0, 0);
      }
      ATTN_PRIO(0);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(sa[r], LOG2E, nlse2));
        sa[r] = pr * (da[r] - delta);  // dS^T
      }
      if (kt == NT - 1) {  // only the last key tile can hold keys >= N
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (key >= p.N) sa[r] = 0.f;
        }
      }
      ATTN_PRIO(1);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 dsf = acc_to_frag(sa, s2);
#pragma unroll
        for (int eb = 0; eb < 2; ++eb) {
          const bf16x8 ktf = lds_tr_frag(imgK, kt * 32 + 16 * s2 + trow, 32 * eb + tcol);
          dq[eb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf, dsf, dq[eb], 0, 0, 0);
        }
      }
      ATTN_PRIO(0);
    }
    if (q < p.N) {
      bf16_t* drow = p.dqkv + (size_t)(row_base + q) * p.ld + hh * HD;
      store_row_tile(drow, dq, half, HD, p.dq_scale);
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
    const f32x16 zero16b = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
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
      f32x16 sa = zero16b, da = zero16b;
      // -LSE*log2e and delta of this tile's queries: register r holds query qt*32 + 8*(r>>2) + 4*half + (r&3): four 16-byte reads each
      float4 nl4[4], dl4[4];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        nl4[g4] = *reinterpret_cast<const float4*>(lse_s + qt * 32 + 8 * g4 + 4 * half);
        dl4[g4] = *reinterpret_cast<const float4*>(del_s + qt * 32 + 8 * g4 + 4 * half);
      }
      ATTN_PRIO(1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 qfr = lds_row_frag(imgQ, qt * 32 + ql, 2 * ks + half);
        const bf16x8 dfr = lds_row_frag(imgD, qt * 32 + ql, 2 * ks + half);
        sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr, kf[ks], sa, 0, 0, 0);
        da = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dfr, vf[ks], da, 0, 0, 0);
      }
      ATTN_PRIO(0);
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const float nl[4] = {nl4[g4].x, nl4[g4].y, nl4[g4].z, nl4[g4].w};
        const float dl[4] = {dl4[g4].x, dl4[g4].y, dl4[g4].z, dl4[g4].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = 4 * g4 + j;
          const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(sa[r], LOG2E, nl[j]));  // rows q >= N: -inf -> 0
          sa[r] = pr;
          da[r] = pr * (da[r] - dl[j]);
        }
      }
      ATTN_PRIO(1);
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
      ATTN_PRIO(0);
    }
    if (key < p.N) {
      bf16_t* krow = p.dqkv + (size_t)(row_base + key) * p.ld + p.d + hh * HD;
      bf16_t* vrow = krow + p.d;
      store_row_tile(krow, dk, half, HD, 1.0f);
      store_row_tile(vrow, dv, half, HD, 1.0f);
    }
  }
}


// Persistent form of the backward (round 4; N <= 224): a workgroup walks over (batch, head) items with the SAME four images, but only
// two are live per pass, so the other two are refilled under it.  Pass B (dK, dV) needs Q, dO whole and this wave's 32 rows of K, V;
// pass A (dQ) needs K, V whole and this wave's rows of Q, dO plus delta = rowsum(dO * O).  Per item, pass B first:
//   [barrier a]  LDS-DMA K,V(i) | pass B(i) | this wave's Q,dO rows out of the images -> registers
//   [barrier b]  dK,dV stores | requests for item i+1: LDS-DMA of this wave's O rows, its K,V rows, LDS-DMA Q,dO | pass A(i) |
//                own vmcnt(0) | delta, LSE of item i+1 -> LDS; K,V rows -> pass-B fragments | dQ stores
// What it took (each item measured with s_memrealtime stamps, tools/attn_stamps.py; profiles/r04_attn_bwd_pers.log):
//  * every LDS read that hipcc can attribute to the LDS address space gets `s_waitcnt vmcnt(0)` in front while an LDS-DMA is in
//    flight (the ds_read_tr16 builtin, plain float4 reads of the statistics): the transposed fragments and the statistics are read
//    by inline asm (TrFrag, lds_read_f4) - without that nothing overlaps (round 2's persistent form, and this one's first build);
//  * a spilled register is reloaded by a VMEM operation that waits for everything issued before it, the LDS-DMA included: the rows
//    requested from HBM are live only during pass A (the pass with registers to spare), the O rows go by LDS-DMA into a 4 KB image
//    private to the wave, fragments and statistics are requested in halves; 251 VGPRs, no scratch;
//  * the LDS-DMA is issued by straight-line code (stage_image_u) so that the waits hipcc places are COUNTED ones, and each wave
//    stages its own 32-row block, so delta / the K, V fragments of the next item need only the wave's own vmcnt(0), before barrier a;
//  * the CU accepts a vector-memory instruction every ~50 cycles, slower when it touches 32 cache lines: all row traffic is whole
//    128-byte rows - stores through the wave's image (store_tile_rows: the second wave of every SIMD spent 4.3 us issuing its dK, dV
//    stores as 32-byte pieces), K, V rows read whole and turned into fragments through the image, barrier a bare (not waiting for
//    the dQ stores).
// DeiT-B's layer: 138 -> 129 us per launch (117 in the step); a workgroup's item 20.7 -> 17 us with the passes at 10.7 us: what is
// left between the passes is the issue of ~45 vector-memory instructions per wave and two barriers.
template <int NT>
__global__ __launch_bounds__(64 * NT) void attn_bwd_pers_kernel(const AttnParams p) {
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
  char* scr = smem + 4 * IMG + 2 * NT * 32 * (int)sizeof(float) + wave * (32 * ROWB);  // this wave's 4 KB transposition image
  const int nitems = p.B * p.H;
  size_t bytes = (size_t)p.B * p.N * p.ld * 2;
  if (bytes > 0xffffffe0ull) bytes = 0xffffffe0ull;
  size_t bytes_o = (size_t)p.B * p.N * p.d * 2;
  if (bytes_o > 0xffffffe0ull) bytes_o = 0xffffffe0ull;
  const auto srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.qkv), 0, (uint32_t)bytes, 0x00020000);
  const auto srdD = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.d_o), 0, (uint32_t)bytes_o, 0x00020000);
  const int ql = lane & 31, half = lane >> 5;
  const int g = lane >> 4, t = lane & 15;
  const int trow = 4 * (g >> 1) + (t >> 2);
  const int tcol = 16 * (g & 1) + 4 * (t & 3);
  const int q = wave * 32 + ql;  // this wave's row in both passes (key in B, query in A)
  uint32_t trc[2][2];  // [eb][lo / hi]: transposed-fragment addresses of tile 0 in the first image
#pragma unroll
  for (int eb = 0; eb < 2; ++eb) {
    trc[eb][0] = lds_addr32(smem) + tr_lane_off(trow, 32 * eb + tcol, false);
    trc[eb][1] = lds_addr32(smem) + tr_lane_off(trow, 32 * eb + tcol, true);
  }
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  auto stage_kv = [&](int item) {
    const int b = item / p.H, hh = item - b * p.H;
    stage_image_u<NT, NT>(imgK, srd, (long)b * p.N, p.N, p.ld, p.d + hh * HD, wave, lane);
    stage_image_u<NT, NT>(imgV, srd, (long)b * p.N, p.N, p.ld, 2 * p.d + hh * HD, wave, lane);
  };
  auto stage_qd = [&](int item) {
    const int b = item / p.H, hh = item - b * p.H;
    stage_image_u<NT, NT>(imgQ, srd, (long)b * p.N, p.N, p.ld, hh * HD, wave, lane);
    stage_image_u<NT, NT>(imgD, srdD, (long)b * p.N, p.N, p.d, hh * HD, wave, lane);
  };
  // this wave's 32 rows of O (for delta = rowsum(dO * O); dO comes out of its LDS image) go by LDS-DMA into the wave's transposition
  // image (no registers), its LSE values and its rows of K, V (B operands of pass B) into registers; all as whole 128-byte rows.
  bf16x8 kf[4], vf[4];
  float lse_q = 0.f;
  const auto srdO = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.o), 0, (uint32_t)bytes_o, 0x00020000);
  auto request_o = [&](int item) {
    const int b = item / p.H, hh = item - b * p.H;
    lse_q = INFINITY;  // rows q >= N: -LSE * log2 e = -inf, P = 0
    if (q < p.N) lse_q = p.lse[((size_t)b * p.H + hh) * p.N + q];
    const int lrow = lane >> 3, pc = lane & 7;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int t = 8 * j + lrow;
      const int c = pc ^ rot3(t);
      uint32_t voff = 0xfffffff0u;
      if (wave * 32 + t < p.N) voff = (uint32_t)(((size_t)((long)b * p.N + wave * 32 + t) * p.d + hh * HD + c * 8) * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srdO, (__attribute__((address_space(3))) void*)(scr + j * 1024), 16, voff, 0, 0, 0);
    }
  };
  auto request_rows = [&](int item, bf16x8 (&f)[4], int col0) {  // whole rows again (kf / vf hold them raw until rows_to_frags)
    const int b = item / p.H, hh = item - b * p.H;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = wave * 32 + 8 * j + (lane >> 3);
      f[j] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      if (row < p.N) f[j] = *reinterpret_cast<const bf16x8*>(p.qkv + (size_t)((long)b * p.N + row) * p.ld + col0 + hh * HD + 8 * (lane & 7));
    }
  };
  auto request_kv = [&](int item) {
    request_rows(item, kf, p.d);
    request_rows(item, vf, 2 * p.d);
  };
  // raw rows -> B-operand fragments (lane = (row, half), 16-byte chunks 2 ks + half) through the transposition image
  auto rows_to_frags = [&](bf16x8 (&f)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<bf16x8*>(scr + img_off(8 * j + (lane >> 3), lane & 7)) = f[j];
    uint4 v[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) lds_read_u4(v[ks], lds_addr32(scr) + (uint32_t)img_off(ql, 2 * ks + half));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) f[ks] = __builtin_bit_cast(bf16x8, v[ks]);
  };
  // delta and -LSE * log2 e of this wave's queries, once ITS rows of the item's dO image and of O have landed (its own LDS-DMA: no
  // barrier needed): LDS for everybody's pass B (published by barrier a), registers for this wave's pass A.
  float delta = 0.f, nlse2 = 0.f;
  auto finish_delta = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = wave * 32 + 8 * j + (lane >> 3);
      const bf16x8 dv = lds_row_frag(imgD, row, lane & 7);  // rows >= N are zero in both
      const bf16x8 ov = lds_row_frag(scr, 8 * j + (lane >> 3), lane & 7);
      float part = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) part += bf16_to_f32((bf16_t)ov[e]) * bf16_to_f32((bf16_t)dv[e]);
      part += dpp_mov<0xB1>(part);   // quad_perm [1,0,3,2]
      part += dpp_mov<0x4E>(part);   // quad_perm [2,3,0,1]
      part += dpp_mov<0x141>(part);  // row_half_mirror: the other quad of the 8 lanes of this row
      if ((lane & 7) == 0) del_s[row] = part;
    }
    nlse2 = -LOG2E * lse_q;
    if (half == 0) lse_s[q] = nlse2;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(delta) : "v"(lds_addr32(smem) + 4 * IMG + (uint32_t)((NT * 32 + q) * 4)) : "memory");
  };

  int item = blockIdx.x;
  if (item >= nitems) return;
  stage_qd(item);
  request_o(item);
  request_kv(item);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  finish_delta();
  rows_to_frags(kf);
  rows_to_frags(vf);
  __syncthreads();  // barrier a of the first item
#pragma unroll 1
  for (;;) {
    const int nxt = item + gridDim.x;
    const bool has_next = nxt < nitems;
    const int b = item / p.H, hh = item - b * p.H;
    const long row_base = (long)b * p.N;
    stage_kv(item);
    // ---- pass B: this wave owns keys (lane = key).  S[q][key] non-swapped: rows = queries (registers).
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int eb = 0; eb < 2; ++eb) {
      dk[eb] = zero16;
      dv[eb] = zero16;
    }
#pragma unroll 1
    for (int qt = 0; qt < NT; ++qt) {
      f32x16 sa = zero16, da = zero16;
      // -LSE*log2e and delta of this tile's queries: register r holds query qt*32 + 8*(r>>2) + 4*half + (r&3): four 16-byte reads
      // each, as inline asm in two halves (any LDS load the compiler knows the address space of gets `s_waitcnt vmcnt(0)` in front
      // while an LDS-DMA is in flight - these did, and the K, V prefetch landed before the pass went on)
      f32x4 nl4[4], dl4[4];
      const uint32_t sta = lds_addr32(smem) + 4 * IMG + (uint32_t)((qt * 32 + 4 * half) * 4);
#pragma unroll
      for (int g4 = 0; g4 < 2; ++g4) {
        lds_read_f4(nl4[g4], sta, 32 * g4);
        lds_read_f4(dl4[g4], sta, NT * 32 * 4 + 32 * g4);
      }
      ATTN_PRIO(1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 qfr = lds_row_frag(imgQ, qt * 32 + ql, 2 * ks + half);
        const bf16x8 dfr = lds_row_frag(imgD, qt * 32 + ql, 2 * ks + half);
        sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr, kf[ks], sa, 0, 0, 0);
        da = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dfr, vf[ks], da, 0, 0, 0);
        if (ks == 1) __builtin_amdgcn_sched_barrier(0);  // the second half's fragments are requested under these MFMAs: 16 registers less
      }
      ATTN_PRIO(0);
      TrFrag dtf[2][2], qtf[2][2];  // k-step 1's are requested under k-step 0's MFMAs
      const uint32_t tb = (uint32_t)(qt * 32 * ROWB);
#ifdef ATTN_TR_EARLY
#pragma unroll
      for (int eb = 0; eb < 2; ++eb) {
        lds_tr_issue<3 * IMG - 2 * IMG>(dtf[0][eb], trc[eb][0] + tb + 2 * IMG, trc[eb][1] + tb + 2 * IMG);
        lds_tr_issue<0>(qtf[0][eb], trc[eb][0] + tb + 2 * IMG, trc[eb][1] + tb + 2 * IMG);
      }
#endif
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        if (g4 == 0) {
          lds_f4_wait(nl4[0], dl4[0], nl4[1], dl4[1]);
#pragma unroll
          for (int h4 = 2; h4 < 4; ++h4) {  // the second half arrives under the first half's arithmetic
            lds_read_f4(nl4[h4], sta, 32 * h4);
            lds_read_f4(dl4[h4], sta, NT * 32 * 4 + 32 * h4);
          }
        }
        if (g4 == 2) lds_f4_wait(nl4[2], dl4[2], nl4[3], dl4[3]);
        const float nl[4] = {nl4[g4][0], nl4[g4][1], nl4[g4][2], nl4[g4][3]};
        const float dl[4] = {dl4[g4][0], dl4[g4][1], dl4[g4][2], dl4[g4][3]};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = 4 * g4 + j;
          const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(sa[r], LOG2E, nl[j]));  // rows q >= N: -inf -> 0
          sa[r] = pr;
          da[r] = pr * (da[r] - dl[j]);
        }
      }
#ifndef ATTN_TR_EARLY
#pragma unroll
      for (int eb = 0; eb < 2; ++eb) {
        lds_tr_issue<3 * IMG - 2 * IMG>(dtf[0][eb], trc[eb][0] + tb + 2 * IMG, trc[eb][1] + tb + 2 * IMG);
        lds_tr_issue<0>(qtf[0][eb], trc[eb][0] + tb + 2 * IMG, trc[eb][1] + tb + 2 * IMG);
      }
#endif
      lds_tr_wait(dtf[0][0], dtf[0][1], qtf[0][0], qtf[0][1]);
#pragma unroll
      for (int eb = 0; eb < 2; ++eb) {
        lds_tr_issue<IMG + 16 * ROWB>(dtf[1][eb], trc[eb][0] + tb + 2 * IMG, trc[eb][1] + tb + 2 * IMG);
        lds_tr_issue<16 * ROWB>(qtf[1][eb], trc[eb][0] + tb + 2 * IMG, trc[eb][1] + tb + 2 * IMG);
      }
      ATTN_PRIO(1);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pf = acc_to_frag(sa, s2);
        const bf16x8 dsf = acc_to_frag(da, s2);
        if (s2 == 1) lds_tr_wait(dtf[1][0], dtf[1][1], qtf[1][0], qtf[1][1]);
#pragma unroll
        for (int eb = 0; eb < 2; ++eb) {
          dv[eb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(dtf[s2][eb]), pf, dv[eb], 0, 0, 0);
          dk[eb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(qtf[s2][eb]), dsf, dk[eb], 0, 0, 0);
        }
      }
      ATTN_PRIO(0);
    }
    // this wave's Q, dO rows for pass A, before the images are released
    bf16x8 qf[4], df[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[ks] = lds_row_frag(imgQ, q, 2 * ks + half);
      df[ks] = lds_row_frag(imgD, q, 2 * ks + half);
    }
    const float delta_a = delta, nlse2_a = nlse2;
    // barrier b must order every wave's K / V LDS-DMA (stage_kv) before pass A reads rows other waves staged: the wait is explicit
    // (a workgroup-scope fence does not oblige the compiler to drain vmcnt on gfx9; tests/test_abi.py checks the ISA for it)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // barrier b: K, V landed; Q, dO images and the statistics free
    {
      bf16_t* ktile = p.dqkv + (size_t)(row_base + wave * 32) * p.ld + p.d + hh * HD;
      store_tile_rows(scr, ktile, p.ld, p.N - wave * 32, dk, ql, half, lane, 1.0f);
      store_tile_rows(scr, ktile + p.d, p.ld, p.N - wave * 32, dv, ql, half, lane, 1.0f);
    }
    if (has_next) {
      // (issued a few per key tile inside pass A instead - unrolled tile loop - these spill registers, and a spill reload is a VMEM
      // operation that waits for everything before it: measured slower)
      request_o(nxt);
      request_kv(nxt);
      stage_qd(nxt);
    }
    // ---- pass A: this wave owns queries (lane = query).  dQ^T[e][q] = sum_key K^T[e][key] dS^T[key][q]
    f32x16 dq[2];
#pragma unroll
    for (int eb = 0; eb < 2; ++eb) dq[eb] = zero16;
#pragma unroll 1
    for (int kt = 0; kt < NT; ++kt) {
      // accumulators start from the MFMA's inline-constant zero C operand (no per-tile register initialisation); the LSE /
      // delta offsets fold into the exp argument and the dS product: this loop is VALU-bound, every instruction per score counts
      f32x16 sa = zero16, da = zero16;
      ATTN_PRIO(1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 kfr = lds_row_frag(imgK, kt * 32 + ql, 2 * ks + half);
        const bf16x8 vfr = lds_row_frag(imgV, kt * 32 + ql, 2 * ks + half);
        sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr, qf[ks], sa, 0, 0, 0);
        da = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr, df[ks], da, 0, 0, 0);
        if (ks == 1) __builtin_amdgcn_sched_barrier(0);
      }
      ATTN_PRIO(0);
      TrFrag ktf[2][2];
      const uint32_t tb = (uint32_t)(kt * 32 * ROWB);
#ifdef ATTN_TR_EARLY
#pragma unroll
      for (int eb = 0; eb < 2; ++eb) {
        lds_tr_issue<0>(ktf[0][eb], trc[eb][0] + tb, trc[eb][1] + tb);
        lds_tr_issue<16 * ROWB>(ktf[1][eb], trc[eb][0] + tb, trc[eb][1] + tb);
      }
#endif
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(sa[r], LOG2E, nlse2_a));
        sa[r] = pr * (da[r] - delta_a);  // dS^T
      }
      if (kt == NT - 1) {  // only the last key tile can hold keys >= N
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (key >= p.N) sa[r] = 0.f;
        }
      }
#ifndef ATTN_TR_EARLY
#pragma unroll
      for (int eb = 0; eb < 2; ++eb) {
        lds_tr_issue<0>(ktf[0][eb], trc[eb][0] + tb, trc[eb][1] + tb);
        lds_tr_issue<16 * ROWB>(ktf[1][eb], trc[eb][0] + tb, trc[eb][1] + tb);
      }
#endif
      lds_tr_wait(ktf[0][0], ktf[0][1], ktf[1][0], ktf[1][1]);
      ATTN_PRIO(1);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 dsf = acc_to_frag(sa, s2);
#pragma unroll
        for (int eb = 0; eb < 2; ++eb) dq[eb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(ktf[s2][eb]), dsf, dq[eb], 0, 0, 0);
      }
      ATTN_PRIO(0);
    }
    if (has_next) {  // this wave's rows of the next item's dO, O, K, V are here after ITS vmcnt(0): statistics, B operands of pass B
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      finish_delta();
      rows_to_frags(kf);
      rows_to_frags(vf);
    }
    store_tile_rows(scr, p.dqkv + (size_t)(row_base + wave * 32) * p.ld + hh * HD, p.ld, p.N - wave * 32, dq, ql, half, lane, p.dq_scale);
    // barrier a of the next item: its Q, dO landed (every wave waited for its share above), its statistics written; K, V images
    // free.  Bare: a __syncthreads() would wait for the dQ stores as well.
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (!has_next) break;
    item = nxt;
  }
}

// Round 6: the persistent backward with a LOADER wave (N <= 224; NT compute waves + 1).  Stamps of the kernel above put 4.3 + 2.2 us of
// an item's 17.1 between and behind its passes, where all seven waves queue ~45 vector-memory instructions each and the CU takes one
// every ~50 cycles while nothing computes.  Two thirds of them are LDS-DMA requests (the four images, the O rows), which need no
// registers: here ONE extra wave - the second wave of the SIMD that held a single compute wave - issues every LDS-DMA of the
// workgroup (140 per item) beside the passes, and the compute waves keep their K / V row loads, LSE loads and result stores.
//   compute wave:  [a] pass B | own Q, dO rows -> registers | K / V rows + LSE of item i+1 requested | dK, dV stores  [b] pass A
//                  [c] own vmcnt(0) | delta, LSE -> LDS; K, V rows -> fragments | dQ stores  [a] ...
//   loader wave:   [a] LDS-DMA K, V(i); vmcnt(0)  [b] LDS-DMA O rows(i+1) -> every wave's transposition image, Q, dO(i+1); vmcnt(0)  [c] [a]
// Barriers are bare (s_waitcnt lgkmcnt(0); s_barrier): no wave waits for another's stores.  WAR: the loader overwrites the K / V
// images behind barrier a (last read: pass A, before c), the Q / dO images and the transposition images behind b (last read: pass B,
// the own-row fragments and the dK / dV transposition, all before b); RAW: its vmcnt(0) precedes the barrier that publishes the data.
// Same arithmetic per element as attn_bwd_pers_kernel: bitwise the same results.  Measured (profiles/r06_attn_loader_ab.log, DeiT-B's
// layer, same box, A / B / A / B): 129-130 -> 119-120 us alone, 120 -> 109.5 us per dense launch in the step.  Tried on top and dropped:
// the loader at raised priority (+1 us), the SIMD partners (waves 4-6) started half a tile late in both passes (+3...6 us).
template <int NT>
__global__ __launch_bounds__(64 * (NT + 1)) void attn_bwd_persl_kernel(const AttnParams p) {
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
  const bool loader = wave == NT;  // the extra wave: issues every LDS-DMA of the workgroup (see above the kernel)
  char* scr0 = smem + 4 * IMG + 2 * NT * 32 * (int)sizeof(float);
  char* scr = scr0 + (loader ? 0 : wave) * (32 * ROWB);  // this wave's 4 KB transposition image
  const int nitems = p.B * p.H;
  size_t bytes = (size_t)p.B * p.N * p.ld * 2;
  if (bytes > 0xffffffe0ull) bytes = 0xffffffe0ull;
  size_t bytes_o = (size_t)p.B * p.N * p.d * 2;
  if (bytes_o > 0xffffffe0ull) bytes_o = 0xffffffe0ull;
  const auto srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.qkv), 0, (uint32_t)bytes, 0x00020000);
  const auto srdD = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.d_o), 0, (uint32_t)bytes_o, 0x00020000);
  const int ql = lane & 31, half = lane >> 5;
  const int g = lane >> 4, t = lane & 15;
  const int trow = 4 * (g >> 1) + (t >> 2);
  const int tcol = 16 * (g & 1) + 4 * (t & 3);
  const int q = wave * 32 + ql;  // this wave's row in both passes (key in B, query in A)
  uint32_t trc[2][2];  // [eb][lo / hi]: transposed-fragment addresses of tile 0 in the first image
#pragma unroll
  for (int eb = 0; eb < 2; ++eb) {
    trc[eb][0] = lds_addr32(smem) + tr_lane_off(trow, 32 * eb + tcol, false);
    trc[eb][1] = lds_addr32(smem) + tr_lane_off(trow, 32 * eb + tcol, true);
  }
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // (loader) every 32-row block of an image pair: NT x 4 wave-instructions per image, straight-line
  auto stage_kv = [&](int item) {
    const int b = item / p.H, hh = item - b * p.H;
#pragma unroll
    for (int w = 0; w < NT; ++w) {
      stage_image_u<NT, NT>(imgK, srd, (long)b * p.N, p.N, p.ld, p.d + hh * HD, w, lane);
      stage_image_u<NT, NT>(imgV, srd, (long)b * p.N, p.N, p.ld, 2 * p.d + hh * HD, w, lane);
    }
  };
  auto stage_qd = [&](int item) {
    const int b = item / p.H, hh = item - b * p.H;
#pragma unroll
    for (int w = 0; w < NT; ++w) {
      stage_image_u<NT, NT>(imgQ, srd, (long)b * p.N, p.N, p.ld, hh * HD, w, lane);
      stage_image_u<NT, NT>(imgD, srdD, (long)b * p.N, p.N, p.d, hh * HD, w, lane);
    }
  };
  // this wave's 32 rows of O (for delta = rowsum(dO * O); dO comes out of its LDS image) go by LDS-DMA into the wave's transposition
  // image (no registers), its LSE values and its rows of K, V (B operands of pass B) into registers; all as whole 128-byte rows.
  bf16x8 kf[4], vf[4];
  float lse_q = 0.f;
  const auto srdO = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.o), 0, (uint32_t)bytes_o, 0x00020000);
  auto request_lse = [&](int item) {  // (compute wave) LSE of its 32 queries
    const int b = item / p.H, hh = item - b * p.H;
    lse_q = INFINITY;  // rows q >= N: -LSE * log2 e = -inf, P = 0
    if (q < p.N) lse_q = p.lse[((size_t)b * p.H + hh) * p.N + q];
  };
  auto request_o = [&](int item) {    // (loader) every wave's 32 rows of O into that wave's transposition image
    const int b = item / p.H, hh = item - b * p.H;
    const int lrow = lane >> 3, pc = lane & 7;
#pragma unroll
    for (int w = 0; w < NT; ++w)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int t = 8 * j + lrow;
        const int c = pc ^ rot3(t);
        uint32_t voff = 0xfffffff0u;
        if (w * 32 + t < p.N) voff = (uint32_t)(((size_t)((long)b * p.N + w * 32 + t) * p.d + hh * HD + c * 8) * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srdO, (__attribute__((address_space(3))) void*)(scr0 + w * (32 * ROWB) + j * 1024), 16, voff, 0, 0, 0);
      }
  };
  auto request_rows = [&](int item, bf16x8 (&f)[4], int col0) {  // whole rows again (kf / vf hold them raw until rows_to_frags)
    const int b = item / p.H, hh = item - b * p.H;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = wave * 32 + 8 * j + (lane >> 3);
      f[j] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      if (row < p.N) f[j] = *reinterpret_cast<const bf16x8*>(p.qkv + (size_t)((long)b * p.N + row) * p.ld + col0 + hh * HD + 8 * (lane & 7));
    }
  };
  auto request_kv = [&](int item) {
    request_rows(item, kf, p.d);
    request_rows(item, vf, 2 * p.d);
  };
  // raw rows -> B-operand fragments (lane = (row, half), 16-byte chunks 2 ks + half) through the transposition image
  auto rows_to_frags = [&](bf16x8 (&f)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<bf16x8*>(scr + img_off(8 * j + (lane >> 3), lane & 7)) = f[j];
    uint4 v[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) lds_read_u4(v[ks], lds_addr32(scr) + (uint32_t)img_off(ql, 2 * ks + half));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) f[ks] = __builtin_bit_cast(bf16x8, v[ks]);
  };
  // delta and -LSE * log2 e of this wave's queries, once ITS rows of the item's dO image and of O have landed (its own LDS-DMA: no
  // barrier needed): LDS for everybody's pass B (published by barrier a), registers for this wave's pass A.
  float delta = 0.f, nlse2 = 0.f;
  auto finish_delta = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = wave * 32 + 8 * j + (lane >> 3);
      const bf16x8 dv = lds_row_frag(imgD, row, lane & 7);  // rows >= N are zero in both
      const bf16x8 ov = lds_row_frag(scr, 8 * j + (lane >> 3), lane & 7);
      float part = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) part += bf16_to_f32((bf16_t)ov[e]) * bf16_to_f32((bf16_t)dv[e]);
      part += dpp_mov<0xB1>(part);   // quad_perm [1,0,3,2]
      part += dpp_mov<0x4E>(part);   // quad_perm [2,3,0,1]
      part += dpp_mov<0x141>(part);  // row_half_mirror: the other quad of the 8 lanes of this row
      if ((lane & 7) == 0) del_s[row] = part;
    }
    nlse2 = -LOG2E * lse_q;
    if (half == 0) lse_s[q] = nlse2;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(delta) : "v"(lds_addr32(smem) + 4 * IMG + (uint32_t)((NT * 32 + q) * 4)) : "memory");
  };

  int item = blockIdx.x;
  if (item >= nitems) return;
#define ATTN_BARE_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
  if (loader) {
    // ---- the loader wave: nothing but LDS-DMA requests, its own vmcnt(0) and the workgroup's barriers (p, a, then b, c, a per item)
    stage_qd(item);
    request_o(item);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ATTN_BARE_BARRIER();  // p: Q, dO images and O rows of the first item landed
    ATTN_BARE_BARRIER();  // a
#pragma unroll 1
    for (;;) {
      const int nxt = item + gridDim.x;
      const bool has_next = nxt < nitems;
      stage_kv(item);  // K, V images (pass A) land under pass B
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      ATTN_BARE_BARRIER();  // b: K, V landed; Q, dO images, statistics and the transposition images are free
      if (has_next) {
        request_o(nxt);
        stage_qd(nxt);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      ATTN_BARE_BARRIER();  // c: the next item's Q, dO images and O rows landed (under pass A)
      ATTN_BARE_BARRIER();  // a
      if (!has_next) break;
      item = nxt;
    }
    return;
  }
  request_kv(item);
  request_lse(item);
  ATTN_BARE_BARRIER();  // p
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  finish_delta();
  rows_to_frags(kf);
  rows_to_frags(vf);
  ATTN_BARE_BARRIER();  // barrier a of the first item
#pragma unroll 1
  for (;;) {
    const int nxt = item + gridDim.x;
    const bool has_next = nxt < nitems;
    const int b = item / p.H, hh = item - b * p.H;
    const long row_base = (long)b * p.N;
    // ---- pass B: this wave owns keys (lane = key).  S[q][key] non-swapped: rows = queries (registers).
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int eb = 0; eb < 2; ++eb) {
      dk[eb] = zero16;
      dv[eb] = zero16;
    }
#pragma unroll 1
    for (int qt = 0; qt < NT; ++qt) {
      f32x16 sa = zero16, da = zero16;
      // -LSE*log2e and delta of this tile's queries: register r holds query qt*32 + 8*(r>>2) + 4*half + (r&3): four 16-byte reads
      // each, as inline asm in two halves (any LDS load the compiler knows the address space of gets `s_waitcnt vmcnt(0)` in front
      // while an LDS-DMA is in flight - these did, and the K, V prefetch landed before the pass went on)
      f32x4 nl4[4], dl4[4];
      const uint32_t sta = lds_addr32(smem) + 4 * IMG + (uint32_t)((qt * 32 + 4 * half) * 4);
#pragma unroll
      for (int g4 = 0; g4 < 2; ++g4) {
        lds_read_f4(nl4[g4], sta, 32 * g4);
        lds_read_f4(dl4[g4], sta, NT * 32 * 4 + 32 * g4);
      }
      ATTN_PRIO(1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 qfr = lds_row_frag(imgQ, qt * 32 + ql, 2 * ks + half);
        const bf16x8 dfr = lds_row_frag(imgD, qt * 32 + ql, 2 * ks + half);
        sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr, kf[ks], sa, 0, 0, 0);
        da = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dfr, vf[ks], da, 0, 0, 0);
        if (ks == 1) __builtin_amdgcn_sched_barrier(0);  // the second half's fragments are requested under these MFMAs: 16 registers less
      }
      ATTN_PRIO(0);
      TrFrag dtf[2][2], qtf[2][2];  // k-step 1's are requested under k-step 0's MFMAs
      const uint32_t tb = (uint32_t)(qt * 32 * ROWB);
#ifdef ATTN_TR_EARLY
#pragma unroll
      for (int eb = 0; eb < 2; ++eb) {
        lds_tr_issue<3 * IMG - 2 * IMG>(dtf[0][eb], trc[eb][0] + tb + 2 * IMG, trc[eb][1] + tb + 2 * IMG);
        lds_tr_issue<0>(qtf[0][eb], trc[eb][0] + tb + 2 * IMG, trc[eb][1] + tb + 2 * IMG);
      }
#endif
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        if (g4 == 0) {
          lds_f4_wait(nl4[0], dl4[0], nl4[1], dl4[1]);
#pragma unroll
          for (int h4 = 2; h4 < 4; ++h4) {  // the second half arrives under the first half's arithmetic
            lds_read_f4(nl4[h4], sta, 32 * h4);
            lds_read_f4(dl4[h4], sta, NT * 32 * 4 + 32 * h4);
          }
        }
        if (g4 == 2) lds_f4_wait(nl4[2], dl4[2], nl4[3], dl4[3]);
        const float nl[4] = {nl4[g4][0], nl4[g4][1], nl4[g4][2], nl4[g4][3]};
        const float dl[4] = {dl4[g4][0], dl4[g4][1], dl4[g4][2], dl4[g4][3]};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = 4 * g4 + j;
          const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(sa[r], LOG2E, nl[j]));  // rows q >= N: -inf -> 0
          sa[r] = pr;
          da[r] = pr * (da[r] - dl[j]);
        }
      }
#ifndef ATTN_TR_EARLY
#pragma unroll
      for (int eb = 0; eb < 2; ++eb) {
        lds_tr_issue<3 * IMG - 2 * IMG>(dtf[0][eb], trc[eb][0] + tb + 2 * IMG, trc[eb][1] + tb + 2 * IMG);
        lds_tr_issue<0>(qtf[0][eb], trc[eb][0] + tb + 2 * IMG, trc[eb][1] + tb + 2 * IMG);
      }
#endif
      lds_tr_wait(dtf[0][0], dtf[0][1], qtf[0][0], qtf[0][1]);
#pragma unroll
      for (int eb = 0; eb < 2; ++eb) {
        lds_tr_issue<IMG + 16 * ROWB>(dtf[1][eb], trc[eb][0] + tb + 2 * IMG, trc[eb][1] + tb + 2 * IMG);
        lds_tr_issue<16 * ROWB>(qtf[1][eb], trc[eb][0] + tb + 2 * IMG, trc[eb][1] + tb + 2 * IMG);
      }
      ATTN_PRIO(1);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pf = acc_to_frag(sa, s2);
        const bf16x8 dsf = acc_to_frag(da, s2);
        if (s2 == 1) lds_tr_wait(dtf[1][0], dtf[1][1], qtf[1][0], qtf[1][1]);
#pragma unroll
        for (int eb = 0; eb < 2; ++eb) {
          dv[eb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(dtf[s2][eb]), pf, dv[eb], 0, 0, 0);
          dk[eb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(qtf[s2][eb]), dsf, dk[eb], 0, 0, 0);
        }
      }
      ATTN_PRIO(0);
    }
    // this wave's Q, dO rows for pass A, before the images are released
    bf16x8 qf[4], df[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[ks] = lds_row_frag(imgQ, q, 2 * ks + half);
      df[ks] = lds_row_frag(imgD, q, 2 * ks + half);
    }
    const float delta_a = delta, nlse2_a = nlse2;
    // dK, dV leave BEFORE barrier b (through the wave's transposition image, which the loader refills with the next item's O rows
    // behind that barrier); the barrier is bare - it must not wait for these stores.  The K / V images are the loader's: it waits for
    // its own vmcnt(0) in front of barrier b.
    if (has_next) {  // kf / vf are dead behind pass B: the next item's rows are requested in front of the stores (119 -> 115 us)
      request_kv(nxt);
      request_lse(nxt);
    }
    {
      bf16_t* ktile = p.dqkv + (size_t)(row_base + wave * 32) * p.ld + p.d + hh * HD;
      store_tile_rows(scr, ktile, p.ld, p.N - wave * 32, dk, ql, half, lane, 1.0f);
      store_tile_rows(scr, ktile + p.d, p.ld, p.N - wave * 32, dv, ql, half, lane, 1.0f);
    }
    ATTN_BARE_BARRIER();  // barrier b: K, V landed; Q, dO images and the statistics free
    // ---- pass A: this wave owns queries (lane = query).  dQ^T[e][q] = sum_key K^T[e][key] dS^T[key][q]
    f32x16 dq[2];
#pragma unroll
    for (int eb = 0; eb < 2; ++eb) dq[eb] = zero16;
#pragma unroll 1
    for (int kt = 0; kt < NT; ++kt) {
      // accumulators start from the MFMA's inline-constant zero C operand (no per-tile register initialisation); the LSE /
      // delta offsets fold into the exp argument and the dS product: this loop is VALU-bound, every instruction per score counts
      f32x16 sa = zero16, da = zero16;
      ATTN_PRIO(1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 kfr = lds_row_frag(imgK, kt * 32 + ql, 2 * ks + half);
        const bf16x8 vfr = lds_row_frag(imgV, kt * 32 + ql, 2 * ks + half);
        sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr, qf[ks], sa, 0, 0, 0);
        da = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr, df[ks], da, 0, 0, 0);
        if (ks == 1) __builtin_amdgcn_sched_barrier(0);
      }
      ATTN_PRIO(0);
      TrFrag ktf[2][2];
      const uint32_t tb = (uint32_t)(kt * 32 * ROWB);
#ifdef ATTN_TR_EARLY
#pragma unroll
      for (int eb = 0; eb < 2; ++eb) {
        lds_tr_issue<0>(ktf[0][eb], trc[eb][0] + tb, trc[eb][1] + tb);
        lds_tr_issue<16 * ROWB>(ktf[1][eb], trc[eb][0] + tb, trc[eb][1] + tb);
      }
#endif
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(sa[r], LOG2E, nlse2_a));
        sa[r] = pr * (da[r] - delta_a);  // dS^T
      }
      if (kt == NT - 1) {  // only the last key tile can hold keys >= N
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (key >= p.N) sa[r] = 0.f;
        }
      }
#ifndef ATTN_TR_EARLY
#pragma unroll
      for (int eb = 0; eb < 2; ++eb) {
        lds_tr_issue<0>(ktf[0][eb], trc[eb][0] + tb, trc[eb][1] + tb);
        lds_tr_issue<16 * ROWB>(ktf[1][eb], trc[eb][0] + tb, trc[eb][1] + tb);
      }
#endif
      lds_tr_wait(ktf[0][0], ktf[0][1], ktf[1][0], ktf[1][1]);
      ATTN_PRIO(1);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 dsf = acc_to_frag(sa, s2);
#pragma unroll
        for (int eb = 0; eb < 2; ++eb) dq[eb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(ktf[s2][eb]), dsf, dq[eb], 0, 0, 0);
      }
      ATTN_PRIO(0);
    }
    ATTN_BARE_BARRIER();  // barrier c: the loader's requests for the next item (Q, dO images, O rows) have landed
    if (has_next) {  // this wave's K, V rows and LSE values of the next item are here after ITS vmcnt(0): statistics, B operands of pass B
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      finish_delta();
      rows_to_frags(kf);
      rows_to_frags(vf);
    }
    store_tile_rows(scr, p.dqkv + (size_t)(row_base + wave * 32) * p.ld + hh * HD, p.ld, p.N - wave * 32, dq, ql, half, lane, p.dq_scale);
    // barrier a of the next item: its Q, dO landed (every wave waited for its share above), its statistics written; K, V images
    // free.  Bare: a __syncthreads() would wait for the dQ stores as well.
    ATTN_BARE_BARRIER();
    if (!has_next) break;
    item = nxt;
  }
}
#undef ATTN_BARE_BARRIER

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


template <int KS>  // head_dim / 16: compile-time, so that the k-step loops are branch-free (round 4: with a run-time head_dim hipcc emitted
                   // `read one fragment - wait - one MFMA - branch` per k-step, every LDS latency exposed: 870 cycles per score tile in this phase)
__global__ __launch_bounds__(512) void attn_fwd2_kernel(const AttnParams2 pp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const AttnParams& p = pp.a;
  const int NT = pp.nt;
  constexpr int hd = 16 * KS;
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
  uint32_t trv[2][2];  // [eb][rows r.. / r + 8..]: transposed-fragment addresses of V's tile 0 (TrFrag: requested under the exps)
#pragma unroll
  for (int eb = 0; eb < 2; ++eb) {
    trv[eb][0] = lds_addr32(imgV) + tr_lane_off(trow, 32 * eb + tcol, false);
    trv[eb][1] = lds_addr32(imgV) + tr_lane_off(trow, 32 * eb + tcol, true);
  }
  const f32x16 zero16f = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int qb = wave; qb < NT; qb += nwv) {
    const int q = qb * 32 + ql;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = load_row_frag_global(p.qkv, (size_t)(row_base + q), p.ld, hh * hd + 16 * ks + 8 * half, q < p.N);
    float m = -INFINITY, l = 0.f;
    f32x16 oacc[2];
#pragma unroll
    for (int eb = 0; eb < 2; ++eb)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[eb][r] = 0.f;
    // one chunk of K (<= KC) key tiles, K a compile-time count: straight-line code (fragment reads hoisted, MFMAs back to back)
    auto chunk = [&](int c0, auto kc) {
      constexpr int K = decltype(kc)::value;
      f32x16 s[K];
      float cmax = -INFINITY;
#pragma unroll
      for (int j = 0; j < K; ++j) {
        // the accumulator starts from the MFMA's inline-constant zero C operand; only the LAST key tile can hold keys >= N
        s[j] = zero16f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const bf16x8 kf = lds_row_frag(imgK, (c0 + j) * 32 + ql, 2 * ks + half);
          s[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[j], 0, 0, 0);
        }
      }
      if (c0 + K == NT) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = (NT - 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (key >= p.N) s[K - 1][r] = -INFINITY;
        }
      }
#pragma unroll
      for (int j = 0; j < K; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) cmax = fmaxf(cmax, s[j][r]);
      cmax = half_max(cmax);
      const float m_new = fmaxf(m, cmax);  // finite: every chunk holds at least one valid key
      const float alpha = __builtin_amdgcn_exp2f((m - m_new) * LOG2E);
      const float mb = m_new * LOG2E;
      l *= alpha;
#pragma unroll
      for (int eb = 0; eb < 2; ++eb)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[eb][r] *= alpha;
#pragma unroll
      for (int j = 0; j < K; ++j) {
        // this tile's V^T fragments are requested before its exps and waited for behind them (hipcc keeps the ds_read_tr16 builtin
        // next to its MFMA: one exposed LDS round trip per MFMA)
        TrFrag vf[2][2];
        const uint32_t tb = (uint32_t)((c0 + j) * 32 * ROWB);
#pragma unroll
        for (int eb = 0; eb < 2; ++eb) {
          if (32 * eb >= hd) continue;
          lds_tr_issue<0>(vf[0][eb], trv[eb][0] + tb, trv[eb][1] + tb);
          lds_tr_issue<16 * ROWB>(vf[1][eb], trv[eb][0] + tb, trv[eb][1] + tb);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float e = __builtin_amdgcn_exp2f(s[j][r] * LOG2E - mb);
          s[j][r] = e;
          l += e;
        }
        if constexpr (hd > 32) lds_tr_wait(vf[0][0], vf[0][1], vf[1][0], vf[1][1]);
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vf[0][0].lo), "+v"(vf[0][0].hi), "+v"(vf[1][0].lo), "+v"(vf[1][0].hi));
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 pf = acc_to_frag(s[j], s2);
#pragma unroll
          for (int eb = 0; eb < 2; ++eb) {
            if (32 * eb >= hd) continue;
            oacc[eb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(vf[s2][eb]), pf, oacc[eb], 0, 0, 0);
          }
        }
      }
      m = m_new;
    };
    int c0 = 0;
#pragma unroll 1
    for (; c0 + KC <= NT; c0 += KC) chunk(c0, std::integral_constant<int, KC>{});
    switch (NT - c0) {
      case 1: chunk(c0, std::integral_constant<int, 1>{}); break;
      case 2: chunk(c0, std::integral_constant<int, 2>{}); break;
      case 3: chunk(c0, std::integral_constant<int, 3>{}); break;
      default: break;
    }
    l = half_sum(l);
    if (q < p.N) {
      const float inv = 1.0f / l;
      bf16_t* orow = p.o + (size_t)(row_base + q) * p.d + hh * hd;
      store_row_tile(orow, oacc, half, hd, inv);
      if (half == 0 && p.lse != nullptr) p.lse[((size_t)b * p.H + hh) * p.N + q] = m + __logf(l);
    }
  }
}

template <int KS>  // head_dim / 16 (see attn_fwd2_kernel)
__global__ __launch_bounds__(512) void attn_bwd2_kernel(const AttnParams2 pp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const AttnParams& p = pp.a;
  const int NT = pp.nt;
  constexpr int hd = 16 * KS;
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
  // lse_s holds -LSE * log2(e), so P = exp2(fma(S, log2 e, lse_s)); rows q >= N hold -inf (P = 0)
  for (int i = threadIdx.x; i < NT * 32; i += blockDim.x)
    lse_s[i] = (i < p.N) ? -LOG2E * p.lse[((size_t)b * p.H + hh) * p.N + i] : -INFINITY;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int ql = lane & 31, half = lane >> 5;
  const int g = lane >> 4, t = lane & 15;
  const int trow = 4 * (g >> 1) + (t >> 2);
  const int tcol = 16 * (g & 1) + 4 * (t & 3);
  const f32x16 zero16f = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  uint32_t trc[2][2];  // [eb][rows r.. / r + 8..]: transposed-fragment addresses of tile 0 in image 0 (TrFrag: requested under the exps -
                       // hipcc keeps the ds_read_tr16 builtin next to its MFMA, one exposed LDS round trip each)
#pragma unroll
  for (int eb = 0; eb < 2; ++eb) {
    trc[eb][0] = lds_addr32(img0) + tr_lane_off(trow, 32 * eb + tcol, false);
    trc[eb][1] = lds_addr32(img0) + tr_lane_off(trow, 32 * eb + tcol, true);
  }

  // ---- pass A: queries on the lane; K, V resident.  (Round 4: both passes are VALU-issue-bound at N = 577, so they take the lean
  // per-score arithmetic of the resident kernel: zero-C accumulators, LSE / delta folded into the exp argument and the dS product,
  // the key mask on the last tile only, vector reads of the statistics.)
  for (int qb = wave; qb < NT; qb += nwv) {
    const int q = qb * 32 + ql;
    bf16x8 qf[4], df[4];
    float delta = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bool ok = q < p.N;
      qf[ks] = load_row_frag_global(p.qkv, (size_t)(row_base + q), p.ld, hh * hd + 16 * ks + 8 * half, ok);
      df[ks] = load_row_frag_global(p.d_o, (size_t)(row_base + q), p.d, hh * hd + 16 * ks + 8 * half, ok);
      const bf16x8 ov = load_row_frag_global(p.o, (size_t)(row_base + q), p.d, hh * hd + 16 * ks + 8 * half, ok);
#pragma unroll
      for (int j = 0; j < 8; ++j) delta += bf16_to_f32((bf16_t)ov[j]) * bf16_to_f32((bf16_t)df[ks][j]);
    }
    delta = half_sum(delta);
    if (half == 0) del_s[q] = delta;
    const float nlse2 = lse_s[q];  // -LSE * log2 e; -inf for rows q >= N
    f32x16 dq[2];
#pragma unroll
    for (int eb = 0; eb < 2; ++eb)
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[eb][r] = 0.f;
#pragma unroll 1
    for (int kt = 0; kt < NT; ++kt) {
      f32x16 sa = zero16f, da = zero16f;
      ATTN_PRIO(1);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 kf = lds_row_frag(img0, kt * 32 + ql, 2 * ks + half);
        const bf16x8 vf = lds_row_frag(img1, kt * 32 + ql, 2 * ks + half);
        sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sa, 0, 0, 0);
        da = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, df[ks], da, 0, 0, 0);
      }
      ATTN_PRIO(0);
      TrFrag ktf[2][2];
      const uint32_t tb = (uint32_t)(kt * 32 * ROWB);
#pragma unroll
      for (int eb = 0; eb < 2; ++eb) {
        if (32 * eb >= hd) continue;
        lds_tr_issue<0>(ktf[0][eb], trc[eb][0] + tb, trc[eb][1] + tb);
        lds_tr_issue<16 * ROWB>(ktf[1][eb], trc[eb][0] + tb, trc[eb][1] + tb);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(sa[r], LOG2E, nlse2));
        sa[r] = pr * (da[r] - delta);  // dS^T
      }
      if (kt == NT - 1) {  // only the last key tile can hold keys >= N
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (key >= p.N) sa[r] = 0.f;
        }
      }
      if constexpr (hd > 32) lds_tr_wait(ktf[0][0], ktf[0][1], ktf[1][0], ktf[1][1]);
      else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ktf[0][0].lo), "+v"(ktf[0][0].hi), "+v"(ktf[1][0].lo), "+v"(ktf[1][0].hi));
      ATTN_PRIO(1);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 dsf = acc_to_frag(sa, s2);
#pragma unroll
        for (int eb = 0; eb < 2; ++eb) {
          if (32 * eb >= hd) continue;
          dq[eb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(ktf[s2][eb]), dsf, dq[eb], 0, 0, 0);
        }
      }
      ATTN_PRIO(0);
    }
    if (q < p.N) {
      bf16_t* drow = p.dqkv + (size_t)(row_base + q) * p.ld + hh * hd;
      store_row_tile(drow, dq, half, hd, p.dq_scale);
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
    for (int ks = 0; ks < KS; ++ks) {
      const bool ok = key < p.N;
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
      f32x16 sa = zero16f, da = zero16f;
      // -LSE*log2e and delta of this tile's queries: register r holds query qt*32 + 8*(r>>2) + 4*half + (r&3): four 16-byte reads each
      float4 nl4[4], dl4[4];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        nl4[g4] = *reinterpret_cast<const float4*>(lse_s + qt * 32 + 8 * g4 + 4 * half);
        dl4[g4] = *reinterpret_cast<const float4*>(del_s + qt * 32 + 8 * g4 + 4 * half);
      }
      ATTN_PRIO(1);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 qfr = lds_row_frag(img0, qt * 32 + ql, 2 * ks + half);
        const bf16x8 dfr = lds_row_frag(img1, qt * 32 + ql, 2 * ks + half);
        sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr, kf[ks], sa, 0, 0, 0);
        da = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dfr, vf[ks], da, 0, 0, 0);
      }
      ATTN_PRIO(0);
      TrFrag dtf[2][2], qtf[2][2];  // k-step 0's requested under the exps, k-step 1's under k-step 0's MFMAs
      const uint32_t tb = (uint32_t)(qt * 32 * ROWB), imgd = (uint32_t)IMG;
#pragma unroll
      for (int eb = 0; eb < 2; ++eb) {
        if (32 * eb >= hd) continue;
        lds_tr_issue<0>(dtf[0][eb], trc[eb][0] + tb + imgd, trc[eb][1] + tb + imgd);
        lds_tr_issue<0>(qtf[0][eb], trc[eb][0] + tb, trc[eb][1] + tb);
      }
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const float nl[4] = {nl4[g4].x, nl4[g4].y, nl4[g4].z, nl4[g4].w};
        const float dl[4] = {dl4[g4].x, dl4[g4].y, dl4[g4].z, dl4[g4].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = 4 * g4 + j;
          const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(sa[r], LOG2E, nl[j]));  // rows q >= N: -inf -> 0
          sa[r] = pr;
          da[r] = pr * (da[r] - dl[j]);
        }
      }
      if constexpr (hd > 32) lds_tr_wait(dtf[0][0], dtf[0][1], qtf[0][0], qtf[0][1]);
      else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dtf[0][0].lo), "+v"(dtf[0][0].hi), "+v"(qtf[0][0].lo), "+v"(qtf[0][0].hi));
#pragma unroll
      for (int eb = 0; eb < 2; ++eb) {
        if (32 * eb >= hd) continue;
        lds_tr_issue<16 * ROWB>(dtf[1][eb], trc[eb][0] + tb + imgd, trc[eb][1] + tb + imgd);
        lds_tr_issue<16 * ROWB>(qtf[1][eb], trc[eb][0] + tb, trc[eb][1] + tb);
      }
      ATTN_PRIO(1);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pf = acc_to_frag(sa, s2);
        const bf16x8 dsf = acc_to_frag(da, s2);
        if (s2 == 1) {
          if constexpr (hd > 32) lds_tr_wait(dtf[1][0], dtf[1][1], qtf[1][0], qtf[1][1]);
          else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dtf[1][0].lo), "+v"(dtf[1][0].hi), "+v"(qtf[1][0].lo), "+v"(qtf[1][0].hi));
        }
#pragma unroll
        for (int eb = 0; eb < 2; ++eb) {
          if (32 * eb >= hd) continue;
          dv[eb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(dtf[s2][eb]), pf, dv[eb], 0, 0, 0);
          dk[eb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(qtf[s2][eb]), dsf, dk[eb], 0, 0, 0);
        }
      }
      ATTN_PRIO(0);
    }
    if (key < p.N) {
      bf16_t* krow = p.dqkv + (size_t)(row_base + key) * p.ld + p.d + hh * hd;
      bf16_t* vrow = krow + p.d;
      store_row_tile(krow, dk, half, hd, 1.0f);
      store_row_tile(vrow, dv, half, hd, 1.0f);
    }
  }
}


// ------------------------------------------------------------------------------------------------------------
// Streaming kernels: ANY sequence length (round 5; attention.py:41-58 and position_embed.py:52-57 take any image size: ViT-B/16 at
// 512^2 is N = 1 025).  The kernels above keep one head's K / V (or Q / dO) resident in LDS, which stops at 608 tokens.  Here a
// workgroup owns 8 row blocks of 32 (one per wave) of ONE side and the other side streams through LDS in segments of SEG_T = 8 tiles
// (256 rows: two 32 KB images), double-buffered: the LDS-DMA of segment s + 1 is issued right behind the barrier that publishes
// segment s and lands under its arithmetic (one barrier per segment).  Same per-tile arithmetic, rounding points and fragment code as
// the resident general kernels (attn_fwd2 / attn_bwd2): lean per-score math, transposed fragments by inline asm.
//   forward    : wave = 32 queries (Q fragments, running max / sum, O accumulators in registers); K, V stream; online softmax.
//   backward dQ : wave = 32 queries (Q, dO fragments, delta = rowsum(dO * O) in registers); K, V stream.
//   backward dKV: wave = 32 keys (K, V fragments in registers); Q, dO stream together with -LSE log2 e and delta of the segment's 256
//                 queries - delta is recomputed per segment from O and dO rows (L2-resident) by all 512 threads, so the two backward
//                 kernels share nothing and need no workspace (the C entry point has none).
// Every wave takes part in staging and barriers even when its own row block lies beyond N (the last round of a sequence).
constexpr int SEG_T = 8;
constexpr int SEG_ROWS = SEG_T * 32;
constexpr int SEG_IMG = SEG_ROWS * ROWB;  // 32 KB

// rows row0 .. row0 + 255 of one operand into a segment image (rows >= N and chunks >= hd / 8 zero-filled); 4 LDS-DMA per wave at 8 waves
__device__ __forceinline__ void stage_segment(char* img, __amdgpu_buffer_rsrc_t srd, long row_base, int row0, int N, int ld, int col0, int wave,
                                              int nwv, int lane, int hd) {
  const int lrow = lane >> 3, pc = lane & 7;
  for (int inst = wave; inst < SEG_T * 4; inst += nwv) {
    const int tl = inst * 8 + lrow;  // row inside the segment image
    const int t = row0 + tl;
    const int c = pc ^ rot3(tl);
    uint32_t voff = 0xfffffff0u;
    if (t < N && c * 8 < hd) voff = (uint32_t)(((size_t)(row_base + t) * ld + col0 + c * 8) * 2);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)(img + inst * 1024), 16, voff, 0, 0, 0);
  }
}

template <int KS>
__global__ __launch_bounds__(512) void attn_fwd_stream_kernel(const AttnParams2 pp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][K image | V image]
  const AttnParams& p = pp.a;
  const int NT = pp.nt;
  constexpr int hd = 16 * KS;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwv = blockDim.x >> 6;
  const int rounds = (NT + nwv - 1) / nwv;
  const int item = blockIdx.x / rounds, round = blockIdx.x - item * rounds;
  const int b = item / p.H, hh = item - b * p.H;
  const long row_base = (long)b * p.N;
  size_t bytes = (size_t)p.B * p.N * p.ld * 2;
  if (bytes > 0xffffffe0ull) bytes = 0xffffffe0ull;
  const auto srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.qkv), 0, (uint32_t)bytes, 0x00020000);
  const int nseg = (NT + SEG_T - 1) / SEG_T;
  auto stage = [&](int sg) {
    char* buf = smem + (sg & 1) * 2 * SEG_IMG;
    stage_segment(buf, srd, row_base, sg * SEG_ROWS, p.N, p.ld, p.d + hh * hd, wave, nwv, lane, hd);
    stage_segment(buf + SEG_IMG, srd, row_base, sg * SEG_ROWS, p.N, p.ld, 2 * p.d + hh * hd, wave, nwv, lane, hd);
  };
  stage(0);

  const int ql = lane & 31, half = lane >> 5;
  const int g = lane >> 4, t = lane & 15;
  const int trow = 4 * (g >> 1) + (t >> 2);
  const int tcol = 16 * (g & 1) + 4 * (t & 3);
  constexpr int KC = 4;
  uint32_t trv[2][2];  // transposed-fragment addresses of tile 0 of buffer 0's V image
#pragma unroll
  for (int eb = 0; eb < 2; ++eb) {
    trv[eb][0] = lds_addr32(smem + SEG_IMG) + tr_lane_off(trow, 32 * eb + tcol, false);
    trv[eb][1] = lds_addr32(smem + SEG_IMG) + tr_lane_off(trow, 32 * eb + tcol, true);
  }
  const f32x16 zero16f = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int qb = round * nwv + wave;
  const bool active = qb < NT;
  const int q = qb * 32 + ql;
  bf16x8 qf[4];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) qf[ks] = load_row_frag_global(p.qkv, (size_t)(row_base + q), p.ld, hh * hd + 16 * ks + 8 * half, active && q < p.N);
  float m = -INFINITY, l = 0.f;
  f32x16 oacc[2];
#pragma unroll
  for (int eb = 0; eb < 2; ++eb)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[eb][r] = 0.f;

  for (int sg = 0; sg < nseg; ++sg) {
    // segment sg (requested one segment of arithmetic ago) has landed for this wave; behind the barrier for everybody, and everybody
    // is done reading the other buffer
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (sg + 1 < nseg) stage(sg + 1);
    if (!active) continue;
    const uint32_t boff = (uint32_t)((sg & 1) * 2 * SEG_IMG);
    const char* imgK = smem + boff;
    const int t0 = sg * SEG_T;                                   // first key tile of the segment
    const int ntl = (NT - t0) < SEG_T ? (NT - t0) : SEG_T;       // its tiles
    auto chunk = [&](int c0, auto kc) {  // as in attn_fwd2_kernel; c0 = tile inside the segment
      constexpr int K = decltype(kc)::value;
      f32x16 sc[K];
      float cmax = -INFINITY;
#pragma unroll
      for (int j = 0; j < K; ++j) {
        sc[j] = zero16f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const bf16x8 kf = lds_row_frag(imgK, (c0 + j) * 32 + ql, 2 * ks + half);
          sc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sc[j], 0, 0, 0);
        }
      }
      if (t0 + c0 + K == NT) {  // only the sequence's last key tile can hold keys >= N
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = (NT - 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (key >= p.N) sc[K - 1][r] = -INFINITY;
        }
      }
#pragma unroll
      for (int j = 0; j < K; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) cmax = fmaxf(cmax, sc[j][r]);
      cmax = half_max(cmax);
      const float m_new = fmaxf(m, cmax);  // finite: every chunk holds at least one valid key
      const float alpha = __builtin_amdgcn_exp2f((m - m_new) * LOG2E);
      const float mb = m_new * LOG2E;
      l *= alpha;
#pragma unroll
      for (int eb = 0; eb < 2; ++eb)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[eb][r] *= alpha;
#pragma unroll
      for (int j = 0; j < K; ++j) {
        TrFrag vf[2][2];
        const uint32_t tb = boff + (uint32_t)((c0 + j) * 32 * ROWB);
#pragma unroll
        for (int eb = 0; eb < 2; ++eb) {
          if (32 * eb >= hd) continue;
          lds_tr_issue<0>(vf[0][eb], trv[eb][0] + tb, trv[eb][1] + tb);
          lds_tr_issue<16 * ROWB>(vf[1][eb], trv[eb][0] + tb, trv[eb][1] + tb);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float e = __builtin_amdgcn_exp2f(sc[j][r] * LOG2E - mb);
          sc[j][r] = e;
          l += e;
        }
        if constexpr (hd > 32) lds_tr_wait(vf[0][0], vf[0][1], vf[1][0], vf[1][1]);
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vf[0][0].lo), "+v"(vf[0][0].hi), "+v"(vf[1][0].lo), "+v"(vf[1][0].hi));
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 pf = acc_to_frag(sc[j], s2);
#pragma unroll
          for (int eb = 0; eb < 2; ++eb) {
            if (32 * eb >= hd) continue;
            oacc[eb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(vf[s2][eb]), pf, oacc[eb], 0, 0, 0);
          }
        }
      }
      m = m_new;
    };
    int c0 = 0;
#pragma unroll 1
    for (; c0 + KC <= ntl; c0 += KC) chunk(c0, std::integral_constant<int, KC>{});
    switch (ntl - c0) {
      case 1: chunk(c0, std::integral_constant<int, 1>{}); break;
      case 2: chunk(c0, std::integral_constant<int, 2>{}); break;
      case 3: chunk(c0, std::integral_constant<int, 3>{}); break;
      default: break;
    }
  }
  if (!active) return;
  l = half_sum(l);
  if (q < p.N) {
    const float inv = 1.0f / l;
    bf16_t* orow = p.o + (size_t)(row_base + q) * p.d + hh * hd;
    store_row_tile(orow, oacc, half, hd, inv);
    if (half == 0 && p.lse != nullptr) p.lse[((size_t)b * p.H + hh) * p.N + q] = m + __logf(l);
  }
}

template <int KS>
__global__ __launch_bounds__(512) void attn_bwd_dq_stream_kernel(const AttnParams2 pp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][K image | V image]
  const AttnParams& p = pp.a;
  const int NT = pp.nt;
  constexpr int hd = 16 * KS;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwv = blockDim.x >> 6;
  const int rounds = (NT + nwv - 1) / nwv;
  const int item = blockIdx.x / rounds, round = blockIdx.x - item * rounds;
  const int b = item / p.H, hh = item - b * p.H;
  const long row_base = (long)b * p.N;
  size_t bytes = (size_t)p.B * p.N * p.ld * 2;
  if (bytes > 0xffffffe0ull) bytes = 0xffffffe0ull;
  const auto srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.qkv), 0, (uint32_t)bytes, 0x00020000);
  const int nseg = (NT + SEG_T - 1) / SEG_T;
  auto stage = [&](int sg) {
    char* buf = smem + (sg & 1) * 2 * SEG_IMG;
    stage_segment(buf, srd, row_base, sg * SEG_ROWS, p.N, p.ld, p.d + hh * hd, wave, nwv, lane, hd);
    stage_segment(buf + SEG_IMG, srd, row_base, sg * SEG_ROWS, p.N, p.ld, 2 * p.d + hh * hd, wave, nwv, lane, hd);
  };
  stage(0);

  const int ql = lane & 31, half = lane >> 5;
  const int g = lane >> 4, t = lane & 15;
  const int trow = 4 * (g >> 1) + (t >> 2);
  const int tcol = 16 * (g & 1) + 4 * (t & 3);
  const f32x16 zero16f = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  uint32_t trc[2][2];  // transposed-fragment addresses of tile 0 of buffer 0's K image
#pragma unroll
  for (int eb = 0; eb < 2; ++eb) {
    trc[eb][0] = lds_addr32(smem) + tr_lane_off(trow, 32 * eb + tcol, false);
    trc[eb][1] = lds_addr32(smem) + tr_lane_off(trow, 32 * eb + tcol, true);
  }
  const int qb = round * nwv + wave;
  const bool active = qb < NT;
  const int q = qb * 32 + ql;
  const bool ok = active && q < p.N;
  bf16x8 qf[4], df[4];
  float delta = 0.f;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    qf[ks] = load_row_frag_global(p.qkv, (size_t)(row_base + q), p.ld, hh * hd + 16 * ks + 8 * half, ok);
    df[ks] = load_row_frag_global(p.d_o, (size_t)(row_base + q), p.d, hh * hd + 16 * ks + 8 * half, ok);
    const bf16x8 ov = load_row_frag_global(p.o, (size_t)(row_base + q), p.d, hh * hd + 16 * ks + 8 * half, ok);
#pragma unroll
    for (int j = 0; j < 8; ++j) delta += bf16_to_f32((bf16_t)ov[j]) * bf16_to_f32((bf16_t)df[ks][j]);
  }
  delta = half_sum(delta);
  const float nlse2 = ok ? -LOG2E * p.lse[((size_t)b * p.H + hh) * p.N + q] : -INFINITY;  // rows q >= N: P = 0
  f32x16 dq[2];
#pragma unroll
  for (int eb = 0; eb < 2; ++eb)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[eb][r] = 0.f;

  for (int sg = 0; sg < nseg; ++sg) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (sg + 1 < nseg) stage(sg + 1);
    if (!active) continue;
    const uint32_t boff = (uint32_t)((sg & 1) * 2 * SEG_IMG);
    const char* img0 = smem + boff;
    const char* img1 = img0 + SEG_IMG;
    const int t0 = sg * SEG_T;
    const int ntl = (NT - t0) < SEG_T ? (NT - t0) : SEG_T;
#pragma unroll 1
    for (int kt = 0; kt < ntl; ++kt) {
      f32x16 sa = zero16f, da = zero16f;
      ATTN_PRIO(1);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 kf = lds_row_frag(img0, kt * 32 + ql, 2 * ks + half);
        const bf16x8 vf = lds_row_frag(img1, kt * 32 + ql, 2 * ks + half);
        sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sa, 0, 0, 0);
        da = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, df[ks], da, 0, 0, 0);
      }
      ATTN_PRIO(0);
      TrFrag ktf[2][2];
      const uint32_t tb = boff + (uint32_t)(kt * 32 * ROWB);
#pragma unroll
      for (int eb = 0; eb < 2; ++eb) {
        if (32 * eb >= hd) continue;
        lds_tr_issue<0>(ktf[0][eb], trc[eb][0] + tb, trc[eb][1] + tb);
        lds_tr_issue<16 * ROWB>(ktf[1][eb], trc[eb][0] + tb, trc[eb][1] + tb);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(sa[r], LOG2E, nlse2));
        sa[r] = pr * (da[r] - delta);  // dS^T
      }
      if (t0 + kt == NT - 1) {  // only the sequence's last key tile can hold keys >= N
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = (NT - 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (key >= p.N) sa[r] = 0.f;
        }
      }
      if constexpr (hd > 32) lds_tr_wait(ktf[0][0], ktf[0][1], ktf[1][0], ktf[1][1]);
      else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ktf[0][0].lo), "+v"(ktf[0][0].hi), "+v"(ktf[1][0].lo), "+v"(ktf[1][0].hi));
      ATTN_PRIO(1);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 dsf = acc_to_frag(sa, s2);
#pragma unroll
        for (int eb = 0; eb < 2; ++eb) {
          if (32 * eb >= hd) continue;
          dq[eb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(ktf[s2][eb]), dsf, dq[eb], 0, 0, 0);
        }
      }
      ATTN_PRIO(0);
    }
  }
  if (ok) {
    bf16_t* drow = p.dqkv + (size_t)(row_base + q) * p.ld + hh * hd;
    store_row_tile(drow, dq, half, hd, p.dq_scale);
  }
}

template <int KS>
__global__ __launch_bounds__(512) void attn_bwd_dkv_stream_kernel(const AttnParams2 pp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][Q image | dO image], then [2 buffers][-LSE log2 e | delta][256]
  const AttnParams& p = pp.a;
  const int NT = pp.nt;
  constexpr int hd = 16 * KS;
  float* stats = reinterpret_cast<float*>(smem + 4 * SEG_IMG);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwv = blockDim.x >> 6;
  const int rounds = (NT + nwv - 1) / nwv;
  const int item = blockIdx.x / rounds, round = blockIdx.x - item * rounds;
  const int b = item / p.H, hh = item - b * p.H;
  const long row_base = (long)b * p.N;
  size_t bytes = (size_t)p.B * p.N * p.ld * 2;
  if (bytes > 0xffffffe0ull) bytes = 0xffffffe0ull;
  size_t bytes_o = (size_t)p.B * p.N * p.d * 2;
  if (bytes_o > 0xffffffe0ull) bytes_o = 0xffffffe0ull;
  const auto srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.qkv), 0, (uint32_t)bytes, 0x00020000);
  const auto srdD = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.d_o), 0, (uint32_t)bytes_o, 0x00020000);
  const int nseg = (NT + SEG_T - 1) / SEG_T;
  auto stage = [&](int sg) {
    char* buf = smem + (sg & 1) * 2 * SEG_IMG;
    stage_segment(buf, srd, row_base, sg * SEG_ROWS, p.N, p.ld, hh * hd, wave, nwv, lane, hd);
    stage_segment(buf + SEG_IMG, srdD, row_base, sg * SEG_ROWS, p.N, p.d, hh * hd, wave, nwv, lane, hd);
  };
  // -LSE log2 e and delta = rowsum(dO * O) of the segment's queries: a thread takes half a row (head_dim / 2 columns) of two rows' worth
  // of threads; rows >= N get -inf / 0 (P = 0)
  auto fill_stats = [&](int sg) {
    float* st = stats + (sg & 1) * 2 * SEG_ROWS;
    for (int i = threadIdx.x; i < 2 * SEG_ROWS; i += blockDim.x) {
      const int r = i >> 1, hf = i & 1;
      const int qg = sg * SEG_ROWS + r;
      float part = 0.f;
      if (qg < p.N) {
        const bf16_t* orow = p.o + (size_t)(row_base + qg) * p.d + hh * hd + hf * (hd / 2);
        const bf16_t* drow = p.d_o + (size_t)(row_base + qg) * p.d + hh * hd + hf * (hd / 2);
#pragma unroll
        for (int c = 0; c < hd / 16; ++c) {
          const bf16x8 ov = *reinterpret_cast<const bf16x8*>(orow + 8 * c);
          const bf16x8 dv = *reinterpret_cast<const bf16x8*>(drow + 8 * c);
#pragma unroll
          for (int e = 0; e < 8; ++e) part += bf16_to_f32((bf16_t)ov[e]) * bf16_to_f32((bf16_t)dv[e]);
        }
      }
      part += __shfl_xor(part, 1, 64);
      if (hf == 0) {
        st[r] = (qg < p.N) ? -LOG2E * p.lse[((size_t)b * p.H + hh) * p.N + qg] : -INFINITY;
        st[SEG_ROWS + r] = part;
      }
    }
  };
  stage(0);
  fill_stats(0);

  const int ql = lane & 31, half = lane >> 5;
  const int g = lane >> 4, t = lane & 15;
  const int trow = 4 * (g >> 1) + (t >> 2);
  const int tcol = 16 * (g & 1) + 4 * (t & 3);
  const f32x16 zero16f = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  uint32_t trc[2][2];  // transposed-fragment addresses of tile 0 of buffer 0's Q image (dO image: + SEG_IMG)
#pragma unroll
  for (int eb = 0; eb < 2; ++eb) {
    trc[eb][0] = lds_addr32(smem) + tr_lane_off(trow, 32 * eb + tcol, false);
    trc[eb][1] = lds_addr32(smem) + tr_lane_off(trow, 32 * eb + tcol, true);
  }
  const int kb = round * nwv + wave;
  const bool active = kb < NT;
  const int key = kb * 32 + ql;
  const bool ok = active && key < p.N;
  bf16x8 kf[4], vf[4];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
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

  for (int sg = 0; sg < nseg; ++sg) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (sg + 1 < nseg) stage(sg + 1);
    if (active) {
      const uint32_t boff = (uint32_t)((sg & 1) * 2 * SEG_IMG);
      const char* img0 = smem + boff;
      const char* img1 = img0 + SEG_IMG;
      const uint32_t sta0 = lds_addr32(smem) + 4 * SEG_IMG + (uint32_t)((sg & 1) * 2 * SEG_ROWS * 4);
      const int t0 = sg * SEG_T;
      const int ntl = (NT - t0) < SEG_T ? (NT - t0) : SEG_T;
#pragma unroll 1
      for (int qt = 0; qt < ntl; ++qt) {
        f32x16 sa = zero16f, da = zero16f;
        // register r holds query qt*32 + 8*(r>>2) + 4*half + (r&3): four 16-byte reads of each statistic (inline asm: an LDS read hipcc
        // can attribute gets `s_waitcnt vmcnt(0)` in front while the next segment's LDS-DMA is in flight)
        f32x4 nl4[4], dl4[4];
        const uint32_t sta = sta0 + (uint32_t)((qt * 32 + 4 * half) * 4);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          lds_read_f4(nl4[g4], sta, 32 * g4);
          lds_read_f4(dl4[g4], sta, SEG_ROWS * 4 + 32 * g4);
        }
        ATTN_PRIO(1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const bf16x8 qfr = lds_row_frag(img0, qt * 32 + ql, 2 * ks + half);
          const bf16x8 dfr = lds_row_frag(img1, qt * 32 + ql, 2 * ks + half);
          sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr, kf[ks], sa, 0, 0, 0);
          da = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dfr, vf[ks], da, 0, 0, 0);
        }
        ATTN_PRIO(0);
        TrFrag dtf[2][2], qtf[2][2];
        const uint32_t tb = boff + (uint32_t)(qt * 32 * ROWB), imgd = (uint32_t)SEG_IMG;
#pragma unroll
        for (int eb = 0; eb < 2; ++eb) {
          if (32 * eb >= hd) continue;
          lds_tr_issue<0>(dtf[0][eb], trc[eb][0] + tb + imgd, trc[eb][1] + tb + imgd);
          lds_tr_issue<0>(qtf[0][eb], trc[eb][0] + tb, trc[eb][1] + tb);
        }
        lds_f4_wait(nl4[0], dl4[0], nl4[1], dl4[1]);  // (in-order LDS: this also retires the transposed reads above - waited again below)
        lds_f4_wait(nl4[2], dl4[2], nl4[3], dl4[3]);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int r = 4 * g4 + j;
            const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(sa[r], LOG2E, nl4[g4][j]));  // rows q >= N: -inf -> 0
            sa[r] = pr;
            da[r] = pr * (da[r] - dl4[g4][j]);
          }
        }
        if constexpr (hd > 32) lds_tr_wait(dtf[0][0], dtf[0][1], qtf[0][0], qtf[0][1]);
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dtf[0][0].lo), "+v"(dtf[0][0].hi), "+v"(qtf[0][0].lo), "+v"(qtf[0][0].hi));
#pragma unroll
        for (int eb = 0; eb < 2; ++eb) {
          if (32 * eb >= hd) continue;
          lds_tr_issue<16 * ROWB>(dtf[1][eb], trc[eb][0] + tb + imgd, trc[eb][1] + tb + imgd);
          lds_tr_issue<16 * ROWB>(qtf[1][eb], trc[eb][0] + tb, trc[eb][1] + tb);
        }
        ATTN_PRIO(1);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 pf = acc_to_frag(sa, s2);
          const bf16x8 dsf = acc_to_frag(da, s2);
          if (s2 == 1) {
            if constexpr (hd > 32) lds_tr_wait(dtf[1][0], dtf[1][1], qtf[1][0], qtf[1][1]);
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dtf[1][0].lo), "+v"(dtf[1][0].hi), "+v"(qtf[1][0].lo), "+v"(qtf[1][0].hi));
          }
#pragma unroll
          for (int eb = 0; eb < 2; ++eb) {
            if (32 * eb >= hd) continue;
            dv[eb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(dtf[s2][eb]), pf, dv[eb], 0, 0, 0);
            dk[eb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(qtf[s2][eb]), dsf, dk[eb], 0, 0, 0);
          }
        }
        ATTN_PRIO(0);
      }
    }
    // the statistics of the next segment, behind this segment's arithmetic (its LDS-DMA was issued a whole segment ago: the waits hipcc
    // puts in front of these global loads and LDS stores cost nothing)
    if (sg + 1 < nseg) fill_stats(sg + 1);
  }
  if (ok) {
    bf16_t* krow = p.dqkv + (size_t)(row_base + key) * p.ld + p.d + hh * hd;
    bf16_t* vrow = krow + p.d;
    store_row_tile(krow, dk, half, hd, 1.0f);
    store_row_tile(vrow, dv, half, hd, 1.0f);
  }
}


// ------------------------------------------------------------------------------------------------------------
// Talking-heads attention (CaiT SA layers): attention.py:41-58 with talking_heads=True (talking_heads.py:9-14)
//   S_h = q_h k_h^T ; S'_i = sum_h T1[h,i] S_h ; P_i = softmax_k(S'_i) ; P'_i = sum_h T2[h,i] P_h ; O_i = P'_i v_i
// The two H x H mixings couple every head at every (q,k), so a per-head flash kernel cannot fuse them.  Round-1
// form: S and P' are materialised in HBM as bf16 [B,H,N,Np] (exactly the tensors the reference's XLA graph
// materialises; S is bf16 there too) and the work is split into three kernels per direction:
//   th_scores (MFMA, per (b,h))  ->  th_softmax (VALU, one wave per (b,q) row, all heads, fp32)  ->  th_pv (MFMA)
// and backward th_pv_bwd (dP', dV) -> th_softmax_bwd (dS, dT1, dT2; P recomputed from S) -> th_scores_bwd (dQ, dK).
// Tile loads/stores use the accumulator layouts of the fused kernels above, so no LDS transposition is needed.
constexpr int TH_MAX_NT = 8;  // 32-row tiles per image: N <= 255 patches (checked by the host wrappers)
struct ThParams {
  const bf16_t* qkv;   // [B*N, ld]
  bf16_t* sbuf;        // S or dS      [B,H,N,Np]
  bf16_t* pbuf;        // P' or dP'    [B,H,N,Np]
  bf16_t* o;           // O  [B*N, d]  (fwd out)      /  dO (bwd in, const)
  bf16_t* dqkv;        // [B*N, ld]
  int B, N, H, ld, d, Np, nt, hd;
  float dq_scale;
};

// S^T / dP'^T accumulator tile -> buf[q][key]: lane = (q, half); registers 4g..4g+3 = keys kt*32 + 8g + 4*half + 0..3
__device__ __forceinline__ void store_tile_T(bf16_t* rowp, const f32x16& a, int kt, int half, int Np) {
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    const int key = kt * 32 + 8 * g4 + 4 * half;
    if (key < Np)
      *reinterpret_cast<uint2*>(rowp + key) = make_uint2(pack_bf16x2(a[4 * g4], a[4 * g4 + 1]), pack_bf16x2(a[4 * g4 + 2], a[4 * g4 + 3]));
  }
}
// Two adjacent key tiles (64 keys = one 128-B row segment per query) through a 4 KB image private to the wave, so that the HBM
// stores are whole row segments: 8 lanes x 16 B per row, 8 rows per instruction, instead of 16 B per row and 32 rows per
// instruction (partial lines: the 167 MB S / dP' tensors were written at 2.3 TB/s).  `tile` uses the staged images' layout
// (128-B rows, 16-B chunk index XORed with rot3(row)); the LDS pipe is in order per wave, so no barrier is needed.
__device__ __forceinline__ void park_tile_T(char* tile, const f32x16& a, int sub, int ql, int half) {
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4)
    *reinterpret_cast<uint2*>(tile + ql * ROWB + (((sub * 4 + g4) ^ rot3(ql)) << 4) + 8 * half) =
        make_uint2(pack_bf16x2(a[4 * g4], a[4 * g4 + 1]), pack_bf16x2(a[4 * g4 + 2], a[4 * g4 + 3]));
}
__device__ __forceinline__ void drain_tile_pair(const char* tile, bf16_t* buf, int q0, int ktp, int lane, int N, int Np) {
  const int r = lane >> 3, c = lane & 7;
  const int key = ktp * 64 + 8 * c;  // Np % 8 == 0: a 16-B chunk is entirely inside or outside the row
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = r + 8 * i;
    const uint4 v = *reinterpret_cast<const uint4*>(tile + row * ROWB + ((c ^ rot3(row)) << 4));
    if (q0 + row < N && key < Np) *reinterpret_cast<uint4*>(buf + (size_t)(q0 + row) * Np + key) = v;
  }
}
// B-operand fragment of k-step s2 read from buf[q][.]: elements j <-> key = kt*32 + 16*s2 + 8*(j>>2) + 4*half + (j&3)
__device__ __forceinline__ bf16x8 load_tile_T_frag(const bf16_t* rowp, bool row_ok, int kt, int s2, int half, int Np) {
  union { uint2 u[2]; bf16x8 v; } r;
  r.u[0] = make_uint2(0u, 0u);
  r.u[1] = make_uint2(0u, 0u);
  const int k0 = kt * 32 + 16 * s2 + 4 * half;
  if (row_ok && k0 < Np) r.u[0] = *reinterpret_cast<const uint2*>(rowp + k0);
  if (row_ok && k0 + 8 < Np) r.u[1] = *reinterpret_cast<const uint2*>(rowp + k0 + 8);
  return r.v;
}
// The same fragment in NATURAL contraction order (round 6): element j <-> key = kt*32 + 16*s2 + 8*half + j - ONE 16-byte load per lane, so
// a wave instruction moves 32 rows x 32 B = 1 KB instead of 512 B (a CU accepts about one vector-memory instruction per ~50 cycles
// whatever it carries: the P' / dS reads of th_pv and of the scores backward were half-width requests).  An MFMA sums over its 16
// contraction slots whatever keys they hold, as long as the A operand uses the same order: lds_tr_frag_nat below.
__device__ __forceinline__ bf16x8 load_tile_T_frag16(const bf16_t* rowp, bool row_ok, int kt, int s2, int half, int Np) {
  bf16x8 r = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  const int k0 = kt * 32 + 16 * s2 + 8 * half;  // Np % 8 == 0: the chunk is entirely inside or outside the row
  if (row_ok && k0 < Np) r = *reinterpret_cast<const bf16x8*>(rowp + k0);
  return r;
}
// transposed A fragment whose contraction slots are rows row0 .. row0 + 7 (this lane addresses row0 + (t >> 2), row0 = tile row + 8 * half)
__device__ __forceinline__ bf16x8 lds_tr_frag_nat(const char* img, int row0, int col) {
  const int r = row0, r4 = row0 + 4;
  const int o0 = r * ROWB + (((col >> 3) ^ rot3(r)) << 4) + ((col & 7) << 1);
  const int o1 = r4 * ROWB + (((col >> 3) ^ rot3(r4)) << 4) + ((col & 7) << 1);
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(img + o0));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(img + o1));
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
#ifndef SAVIT_TH_NAT16
#define SAVIT_TH_NAT16 1
#endif
// B-operand fragment with the contraction index on the ROWS of buf (column gather): element j <-> row
// q = qt*32 + 16*s2 + 8*(j>>2) + 4*half + (j&3), fixed column `key`
__device__ __forceinline__ bf16x8 load_tile_col_frag(const bf16_t* base, int key, int qt, int s2, int half, int N, int Np) {
  bf16x8 r = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  if (key < Np) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int q = qt * 32 + 16 * s2 + 8 * (j >> 2) + 4 * half + (j & 3);
      if (q < N) r[j] = (short)base[(size_t)q * Np + key];
    }
  }
  return r;
}

__global__ __launch_bounds__(512) void th_scores_kernel(const ThParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int NT = p.nt, hd = p.hd;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwv = blockDim.x >> 6;
  const int b = blockIdx.x / p.H, hh = blockIdx.x - b * p.H;
  const long row_base = (long)b * p.N;
  size_t bytes = (size_t)p.B * p.N * p.ld * 2;
  if (bytes > 0xffffffe0ull) bytes = 0xffffffe0ull;
  const auto srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.qkv), 0, (uint32_t)bytes, 0x00020000);
  stage_image_rt<0>(smem, srd, row_base, p.N, p.ld, p.d + hh * hd, wave, nwv, lane, NT, hd);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int ql = lane & 31, half = lane >> 5;
  bf16_t* sb = p.sbuf + ((size_t)b * p.H + hh) * p.N * p.Np;
  char* tile = smem + (size_t)NT * 32 * ROWB + (size_t)wave * (32 * ROWB);
  for (int qb = wave; qb < NT; qb += nwv) {
    const int q = qb * 32 + ql;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      qf[ks] = load_row_frag_global(p.qkv, (size_t)(row_base + q), p.ld, hh * hd + 16 * ks + 8 * half, q < p.N && 16 * ks < hd);
    for (int kt = 0; kt < NT; ++kt) {
      f32x16 sa;
#pragma unroll
      for (int r = 0; r < 16; ++r) sa[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        if (16 * ks < hd) sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_row_frag(smem, kt * 32 + ql, 2 * ks + half), qf[ks], sa, 0, 0, 0);
      park_tile_T(tile, sa, kt & 1, ql, half);
      if ((kt & 1) || kt + 1 == NT) drain_tile_pair(tile, sb, qb * 32, kt >> 1, lane, p.N, p.Np);
    }
  }
}

__global__ __launch_bounds__(512) void th_pv_kernel(const ThParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int NT = p.nt, hd = p.hd;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwv = blockDim.x >> 6;
  const int b = blockIdx.x / p.H, hh = blockIdx.x - b * p.H;
  const long row_base = (long)b * p.N;
  size_t bytes = (size_t)p.B * p.N * p.ld * 2;
  if (bytes > 0xffffffe0ull) bytes = 0xffffffe0ull;
  const auto srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.qkv), 0, (uint32_t)bytes, 0x00020000);
  stage_image_rt<0>(smem, srd, row_base, p.N, p.ld, 2 * p.d + hh * hd, wave, nwv, lane, NT, hd);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int ql = lane & 31, half = lane >> 5, g = lane >> 4, t = lane & 15;
  const int trow = 4 * (g >> 1) + (t >> 2), tcol = 16 * (g & 1) + 4 * (t & 3);
  [[maybe_unused]] const int trow_n = 8 * (g >> 1) + (t >> 2);
  const bf16_t* pb = p.pbuf + ((size_t)b * p.H + hh) * p.N * p.Np;
  for (int qb = wave; qb < NT; qb += nwv) {
    const int q = qb * 32 + ql;
    f32x16 oacc[2];
#pragma unroll
    for (int eb = 0; eb < 2; ++eb)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[eb][r] = 0.f;
    // every P' fragment of this query block is requested before the first MFMA: a load -> MFMA loop exposed the HBM latency
    // once per key tile (7 times per block; these kernels were latency-, not bandwidth-bound).  NT <= TH_MAX_NT.
    bf16x8 pf[TH_MAX_NT][2];
#pragma unroll
    for (int kt = 0; kt < TH_MAX_NT; ++kt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
        pf[kt][s2] = SAVIT_TH_NAT16 ? load_tile_T_frag16(pb + (size_t)q * p.Np, q < p.N && kt < NT, kt, s2, half, p.Np)
                                    : load_tile_T_frag(pb + (size_t)q * p.Np, q < p.N && kt < NT, kt, s2, half, p.Np);
#pragma unroll
    for (int kt = 0; kt < TH_MAX_NT; ++kt) {
      if (kt >= NT) break;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int eb = 0; eb < 2; ++eb)
          oacc[eb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(SAVIT_TH_NAT16 ? lds_tr_frag_nat(smem, kt * 32 + 16 * s2 + trow_n, 32 * eb + tcol)
                                                                            : lds_tr_frag(smem, kt * 32 + 16 * s2 + trow, 32 * eb + tcol),
                                                             pf[kt][s2], oacc[eb], 0, 0, 0);
    }
    if (q < p.N) {
      bf16_t* orow = p.o + (size_t)(row_base + q) * p.d + hh * hd;
      store_row_tile(orow, oacc, half, hd, 1.0f);
    }
  }
}

// MODE 0: dP' = dO V^T (pbuf out) and dV = P'^T dO (pbuf in is P' -> sbuf carries P' here); see host wrapper.
// pass A writes out_t[q][key] = sum_e X[q][e] * IMG_A[key][e]; pass B accumulates G[key][e] = sum_q in_c[q][key] * IMG_B[q][e].
template <int MODE>  // 0: pv backward (A image = V, X = dO, out = dP';  B image = dO, in = P', G = dV)
                     // 1: scores backward (pass A: dQ^T[e][q] = sum_key K^T[e][key] dS^T[key][q]; pass B: G = dK, B image = Q, in = dS)
                     // 2: pass B of mode 0 alone (dV from P' and dO): dP' comes from the fused row kernel (th_fused.hip)
__global__ __launch_bounds__(512) void th_bwd_kernel(const ThParams p) {
  constexpr bool PV = MODE != 1;      // the dO / P' / dV flavour
  constexpr bool PASS_A = MODE != 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int NT = p.nt, hd = p.hd;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwv = blockDim.x >> 6;
  const int b = blockIdx.x / p.H, hh = blockIdx.x - b * p.H;
  const long row_base = (long)b * p.N;
  size_t bytes = (size_t)p.B * p.N * p.ld * 2;
  if (bytes > 0xffffffe0ull) bytes = 0xffffffe0ull;
  size_t bytes_o = (size_t)p.B * p.N * p.d * 2;
  if (bytes_o > 0xffffffe0ull) bytes_o = 0xffffffe0ull;
  const auto srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.qkv), 0, (uint32_t)bytes, 0x00020000);
  const auto srdO = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.o), 0, (uint32_t)bytes_o, 0x00020000);
  const int ql = lane & 31, half = lane >> 5, g = lane >> 4, t = lane & 15;
  const int trow = 4 * (g >> 1) + (t >> 2), tcol = 16 * (g & 1) + 4 * (t & 3);
  [[maybe_unused]] const int trow_n = 8 * (g >> 1) + (t >> 2);
  const size_t bh = ((size_t)b * p.H + hh) * p.N * p.Np;
  // One wave owns one 32-row block in each pass (the launch uses NT waves).  Everything a pass reads from HBM into registers
  // is requested BEFORE the wait for the staged image, so a workgroup exposes the HBM latency twice (once per pass) instead of
  // once per staging plus once per key tile.
  const int qb = wave, q = qb * 32 + ql;
  // ---- pass A
  if constexpr (PASS_A) {
  if (MODE == 0) stage_image_rt<0>(smem, srd, row_base, p.N, p.ld, 2 * p.d + hh * hd, wave, nwv, lane, NT, hd);  // V
  else           stage_image_rt<0>(smem, srd, row_base, p.N, p.ld, p.d + hh * hd, wave, nwv, lane, NT, hd);      // K
  bf16x8 df[4];                    // MODE 0: dO row fragments
  bf16x8 dsf[TH_MAX_NT][2];        // MODE 1: dS row fragments of every key tile
  if (MODE == 0) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      df[ks] = load_row_frag_global(p.o, (size_t)(row_base + q), p.d, hh * hd + 16 * ks + 8 * half, q < p.N && 16 * ks < hd);
  } else {
#pragma unroll
    for (int kt = 0; kt < TH_MAX_NT; ++kt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
        dsf[kt][s2] = SAVIT_TH_NAT16 ? load_tile_T_frag16(p.sbuf + bh + (size_t)q * p.Np, q < p.N && kt < NT, kt, s2, half, p.Np)
                                     : load_tile_T_frag(p.sbuf + bh + (size_t)q * p.Np, q < p.N && kt < NT, kt, s2, half, p.Np);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  char* ptile = smem + (size_t)NT * 32 * ROWB + (size_t)wave * (32 * ROWB);  // the wave's private tile (pass B reuses it)
  if (qb < NT) {
    if (MODE == 0) {
      for (int kt = 0; kt < NT; ++kt) {
        f32x16 da;
#pragma unroll
        for (int r = 0; r < 16; ++r) da[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          if (16 * ks < hd) da = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_row_frag(smem, kt * 32 + ql, 2 * ks + half), df[ks], da, 0, 0, 0);
        park_tile_T(ptile, da, kt & 1, ql, half);
        if ((kt & 1) || kt + 1 == NT) drain_tile_pair(ptile, p.pbuf + bh, qb * 32, kt >> 1, lane, p.N, p.Np);
      }
    } else {
      f32x16 dq[2];
#pragma unroll
      for (int eb = 0; eb < 2; ++eb)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[eb][r] = 0.f;
#pragma unroll
      for (int kt = 0; kt < TH_MAX_NT; ++kt) {
        if (kt >= NT) break;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int eb = 0; eb < 2; ++eb)
            dq[eb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(SAVIT_TH_NAT16 ? lds_tr_frag_nat(smem, kt * 32 + 16 * s2 + trow_n, 32 * eb + tcol)
                                                                            : lds_tr_frag(smem, kt * 32 + 16 * s2 + trow, 32 * eb + tcol),
                                                             dsf[kt][s2], dq[eb], 0, 0, 0);
      }
      if (q < p.N) {
        bf16_t* drow = p.dqkv + (size_t)(row_base + q) * p.ld + hh * hd;
      store_row_tile(drow, dq, half, hd, p.dq_scale);
      }
    }
  }
  }  // PASS_A
  // ---- pass B: G^T[e][key] = sum_q IMG^T[e][q] * in[q][key]
  // The contraction index q is the ROW index of in[q][key].  Gathering the column strips from HBM with two-byte loads (8 per
  // fragment) made this pass 125 of the kernel's 230 us; instead every 32 x 32 tile is read row-contiguous (two 16-B loads per
  // lane, all tiles of the strip requested up front), parked in a 4 KB image private to the wave (same 128-B rows and chunk
  // swizzle as the staged images: 32 keys fill half of each row) and read back transposed by ds_read_b64_tr_b16.  The LDS pipe
  // serves a wave's operations in order, so the next tile's stores cannot overtake this tile's reads: no barrier.
  const bf16_t* inb = p.sbuf + bh;  // MODE 0, 2: P' ; MODE 1: dS
  char* tile = smem + (size_t)NT * 32 * ROWB + (size_t)wave * (32 * ROWB);
  const int lr = lane >> 2, lc = lane & 3;
  const int kb = wave, key = kb * 32 + ql;
  const int kcol = kb * 32 + 8 * lc;  // Np % 8 == 0: a 16-B chunk is entirely inside or outside the row
  uint4 raw[TH_MAX_NT][2];
#pragma unroll
  for (int qt = 0; qt < TH_MAX_NT; ++qt)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int qq = qt * 32 + lr + 16 * i;
      raw[qt][i] = make_uint4(0u, 0u, 0u, 0u);
      if (qt < NT && qq < p.N && kcol < p.Np) raw[qt][i] = *reinterpret_cast<const uint4*>(inb + (size_t)qq * p.Np + kcol);
    }
  __syncthreads();  // every wave is done with the pass-A image
  if (PV) stage_image_rt<0>(smem, srdO, row_base, p.N, p.d, hh * hd, wave, nwv, lane, NT, hd);   // dO
  else    stage_image_rt<0>(smem, srd, row_base, p.N, p.ld, hh * hd, wave, nwv, lane, NT, hd);    // Q
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (kb < NT) {
    f32x16 acc[2];
#pragma unroll
    for (int eb = 0; eb < 2; ++eb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[eb][r] = 0.f;
#pragma unroll
    for (int qt = 0; qt < TH_MAX_NT; ++qt) {
      if (qt >= NT) break;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = lr + 16 * i;
        *reinterpret_cast<uint4*>(tile + row * ROWB + ((lc ^ rot3(row)) << 4)) = raw[qt][i];
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 cf = lds_tr_frag(tile, 16 * s2 + trow, tcol);
#pragma unroll
        for (int eb = 0; eb < 2; ++eb)
          acc[eb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_tr_frag(smem, qt * 32 + 16 * s2 + trow, 32 * eb + tcol), cf, acc[eb], 0, 0, 0);
      }
    }
    if (key < p.N) {
      bf16_t* grow = p.dqkv + (size_t)(row_base + key) * p.ld + (PV ? 2 : 1) * p.d + hh * hd;
      store_row_tile(grow, acc, half, hd, 1.0f);
    }
  }
}

// Packed form of the forward row (H = 8 and 4): a lane's four keys are two key PAIRS, the H x H head mixes run as v_pk_fma_f32 with
// the coefficient pair (T[h][2j], T[h][2j+1]) as ONE 64-bit SGPR operand whose low / high half is broadcast to both halves of the
// packed instruction through op_sel.  hipcc does not form this by itself (it splats the scalar through v_mov), and as C++ loads
// the 2 x 64 coefficients either spill SGPRs (scalar loads hoisted out of the row loop: 240 v_readlane / v_writelane per row) or
// become per-lane vector loads (32 flat loads per row: measured slower).  So one matrix at a time is fetched by inline-asm
// s_load_dwordx16 inside the row and its register pairs feed the FMAs directly: 2 x 128 packed FMAs per row instead of 2 x 256
// scalar ones, no spill traffic.  Same FMA order per accumulator: bitwise the scalar form's results (H = 2, 6, 16 keep that form).
typedef __attribute__((ext_vector_type(16))) uint32_t th_u32x16;
typedef __attribute__((ext_vector_type(8))) uint64_t th_u64x8;
template <int H>
struct ThCoef {  // the H x H matrix as H * H / 2 SGPR pairs
  th_u64x8 q[(H * H / 2 + 7) / 8];
  th_u32x16 r0, r1, r2, r3;
  // issue() requests the matrix, wait() must run before the first use.  Between the two the registers are in flight: the uses are
  // tied to wait()'s outputs, and the ISA of every kernel using this was checked for copies of r0..r3 in between (there are none).
  __device__ __forceinline__ void issue(const float* T) {
    static_assert(H == 8 || H == 4, "whole 64-byte quads");
    if constexpr (H == 8) {
      asm volatile("s_load_dwordx16 %0, %4, 0x0\n\ts_load_dwordx16 %1, %4, 0x40\n\ts_load_dwordx16 %2, %4, 0x80\n\ts_load_dwordx16 %3, %4, 0xc0"
                   : "=s"(r0), "=s"(r1), "=s"(r2), "=s"(r3)
                   : "s"(T)
                   : "memory");
    } else {
      asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(r0) : "s"(T) : "memory");
    }
  }
  __device__ __forceinline__ void wait() {
    if constexpr (H == 8) {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r0), "+s"(r1), "+s"(r2), "+s"(r3)::"memory");
      q[0] = __builtin_bit_cast(th_u64x8, r0); q[1] = __builtin_bit_cast(th_u64x8, r1);
      q[2] = __builtin_bit_cast(th_u64x8, r2); q[3] = __builtin_bit_cast(th_u64x8, r3);
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r0)::"memory");
      q[0] = __builtin_bit_cast(th_u64x8, r0);
    }
  }
  __device__ __forceinline__ void load(const float* T) {
    issue(T);
    wait();
  }
};
// lo / hi[kp] = sum_h T[h][2j] / T[h][2j+1] * x[h][kp]  (kp = key pair)
#define TH_COEF(C, H_, J_) ((C).q[((H_) * (H / 2) + (J_)) / 8][((H_) * (H / 2) + (J_)) % 8])
#define TH_MIX_PK(C, X, J, LO, HI)                                                                                             \
  do {                                                                                                                          \
    LO[0] = LO[1] = HI[0] = HI[1] = f32x2{0.f, 0.f};                                                                            \
    _Pragma("unroll") for (int h_ = 0; h_ < H; ++h_) {                                                                          \
      const uint64_t tp_ = (C).q[(h_ * (H / 2) + (J)) / 8][(h_ * (H / 2) + (J)) % 8];                                           \
      _Pragma("unroll") for (int kp_ = 0; kp_ < 2; ++kp_) {                                                                      \
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(LO[kp_]) : "s"(tp_), "v"(X[h_][kp_]));                       \
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(HI[kp_]) : "s"(tp_), "v"(X[h_][kp_]));        \
      }                                                                                                                         \
    }                                                                                                                           \
  } while (0)

template <int H>
__global__ __launch_bounds__(256) void th_softmax_fwd_kernel(const bf16_t* __restrict__ S, bf16_t* __restrict__ Pp, const float* __restrict__ T1g,
                                                              const float* __restrict__ T2g, int B, int N, int Np) {
  const int lane = threadIdx.x & 63;
  const long rows = (long)B * N;
  if constexpr (H == 8 || H == 4) {
    constexpr int HP = H / 2;
    for (long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (long)gridDim.x * 4) {
      const int b = (int)(row / N), q = (int)(row - (long)b * N);
      f32x2 s2[H][2], p2[H][2];
      const bool in_row = 4 * lane < Np;
      ThCoef<H> c1, c2;
      c1.issue(T1g);  // under the latency of the row loads
#pragma unroll
      for (int h = 0; h < H; ++h) {  // lane owns 4 CONSECUTIVE keys (one 8-byte load per head)
        uint2 v = make_uint2(0u, 0u);
        if (in_row) v = *reinterpret_cast<const uint2*>(S + (((size_t)b * H + h) * N + q) * Np + 4 * lane);
        s2[h][0] = f32x2{__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u)};
        s2[h][1] = f32x2{__uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u)};
      }
      const bool k_ok[4] = {4 * lane < N, 4 * lane + 1 < N, 4 * lane + 2 < N, 4 * lane + 3 < N};
      c1.wait();
#pragma unroll
      for (int j = 0; j < HP; ++j) {  // all mixes first: T1's registers are free for T2 before the softmax work starts
        f32x2 lo[2], hi[2];
        TH_MIX_PK(c1, s2, j, lo, hi);
        p2[2 * j][0] = lo[0]; p2[2 * j][1] = lo[1];
        p2[2 * j + 1][0] = hi[0]; p2[2 * j + 1][1] = hi[1];
      }
      c2.issue(T2g);  // under the softmax of the eight heads
      if constexpr (H == 8) {  // the eight row maxima / row sums as ONE transposing reduction each (th_rows.h)
        float lm[8], m[8], ls[8], l[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          p2[i][0] = f32x2{k_ok[0] ? p2[i][0].x : -INFINITY, k_ok[1] ? p2[i][0].y : -INFINITY};
          p2[i][1] = f32x2{k_ok[2] ? p2[i][1].x : -INFINITY, k_ok[3] ? p2[i][1].y : -INFINITY};
          lm[i] = fmaxf(fmaxf(p2[i][0].x, p2[i][0].y), fmaxf(p2[i][1].x, p2[i][1].y));
        }
        wave_reduce8<true>(lm, m, lane);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float e0 = __builtin_amdgcn_exp2f((p2[i][0].x - m[i]) * LOG2E), e1 = __builtin_amdgcn_exp2f((p2[i][0].y - m[i]) * LOG2E);
          const float e2 = __builtin_amdgcn_exp2f((p2[i][1].x - m[i]) * LOG2E), e3 = __builtin_amdgcn_exp2f((p2[i][1].y - m[i]) * LOG2E);
          p2[i][0] = f32x2{e0, e1};
          p2[i][1] = f32x2{e2, e3};
          ls[i] = (e0 + e1) + (e2 + e3);
        }
        wave_reduce8<false>(ls, l, lane);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float inv = 1.0f / l[i];
          p2[i][0] = p2[i][0] * inv;
          p2[i][1] = p2[i][1] * inv;
        }
      } else {
#pragma unroll
        for (int i = 0; i < H; ++i) {
          float sp[4] = {k_ok[0] ? p2[i][0].x : -INFINITY, k_ok[1] ? p2[i][0].y : -INFINITY, k_ok[2] ? p2[i][1].x : -INFINITY,
                         k_ok[3] ? p2[i][1].y : -INFINITY};
          const float m = wave_max(fmaxf(fmaxf(sp[0], sp[1]), fmaxf(sp[2], sp[3])));
          float l = 0.f;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            sp[k] = __builtin_amdgcn_exp2f((sp[k] - m) * LOG2E);
            l += sp[k];
          }
          const float inv = 1.0f / wave_sum(l);
          p2[i][0] = f32x2{sp[0] * inv, sp[1] * inv};
          p2[i][1] = f32x2{sp[2] * inv, sp[3] * inv};
        }
      }
      c2.wait();
#pragma unroll
      for (int j = 0; j < HP; ++j) {
        f32x2 lo[2], hi[2];
        TH_MIX_PK(c2, p2, j, lo, hi);
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          const f32x2* a = w ? hi : lo;
          const float o4[4] = {k_ok[0] ? a[0].x : 0.f, k_ok[1] ? a[0].y : 0.f, k_ok[2] ? a[1].x : 0.f, k_ok[3] ? a[1].y : 0.f};
          if (in_row)
            *reinterpret_cast<uint2*>(Pp + (((size_t)b * H + 2 * j + w) * N + q) * Np + 4 * lane) =
                make_uint2(pack_bf16x2(o4[0], o4[1]), pack_bf16x2(o4[2], o4[3]));
        }
      }
    }
    return;
  }
  const float* __restrict__ T1 = T1g;
  const float* __restrict__ T2 = T2g;
  for (long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (long)gridDim.x * 4) {
    const int b = (int)(row / N), q = (int)(row - (long)b * N);
    float s[H][TH_KPL], pr[H][TH_KPL];
    // lane owns 4 CONSECUTIVE keys (one 8-byte load per head): strided 2-byte loads made this kernel latency-bound
    const bool in_row = 4 * lane < Np;
#pragma unroll
    for (int h = 0; h < H; ++h) {
      uint2 v = make_uint2(0u, 0u);
      if (in_row) v = *reinterpret_cast<const uint2*>(S + (((size_t)b * H + h) * N + q) * Np + 4 * lane);
      s[h][0] = __uint_as_float(v.x << 16); s[h][1] = __uint_as_float(v.x & 0xffff0000u);
      s[h][2] = __uint_as_float(v.y << 16); s[h][3] = __uint_as_float(v.y & 0xffff0000u);
    }
    th_row_forward<H>(s, T1, N, lane, pr);
#pragma unroll
    for (int i = 0; i < H; ++i) {
      float o4[TH_KPL];
#pragma unroll
      for (int k = 0; k < TH_KPL; ++k) {
        float a = 0.f;
#pragma unroll
        for (int h = 0; h < H; ++h) a += T2[h * H + i] * pr[h][k];
        o4[k] = (4 * lane + k < N) ? a : 0.f;
      }
      if (in_row)
        *reinterpret_cast<uint2*>(Pp + (((size_t)b * H + i) * N + q) * Np + 4 * lane) = make_uint2(pack_bf16x2(o4[0], o4[1]), pack_bf16x2(o4[2], o4[3]));
    }
  }
}

// backward of the row op; dT1/dT2 partial sums go to slab[block][2*H*H] (summed by th_dT_finalize_kernel).
// Register plan (H = 8): S and dP' stay PACKED as bf16 pairs (16 + 16 VGPRs), P / dP are fp32 [H][4] (32 + 32), the 2 x H x H
// per-lane dT partials persist across the rows a wave processes (128); every mixing loop runs key-outer so only 8 + 8 unpacked
// temporaries are live.  The first version kept everything in fp32 and spilled 580 B/lane to scratch (1.55 ms per layer).
// Round 4 (H = 8): the two outer products run on the matrix pipe (MO below): 189 -> ~140 us per CaiT-S24 layer, the whole backward of the
// layer's attention 430 -> 379 us (tools/th_bench.py).
template <int H>
__global__ __launch_bounds__(256) void th_softmax_bwd_kernel(const bf16_t* __restrict__ S, const bf16_t* __restrict__ dPp, bf16_t* __restrict__ dS,
                                                              const float* __restrict__ T1g, const float* __restrict__ T2g,
                                                              float* __restrict__ slab, int B, int N, int Np) {
  static_assert(H <= 8 || H == 16, "dT partials are reduce-scattered as 8x8 tiles: up to 8 heads, or 16 as four tiles per matrix");
  constexpr int NB = (H + 7) / 8;  // 8x8 tiles per dimension of dT (H = 16: cait_m_*; that path spills registers - correct, not fast)
  constexpr bool PK = (H == 8 || H == 4);  // packed head mixes (the three T-mixes of a row: 3 x 128 v_pk_fma_f32 instead of 3 x 256 FMAs)
  // H = 8: the two 8 x 8 outer products dT2 += P dP'^T, dT1 += S dS'^T (contraction over the row's keys) on the MATRIX pipe.  Per row
  // they were 2 x (192 packed multiply-adds + 32 adds + a 64-value reduce-scatter of 156 instructions) = a third of this kernel's
  // vector instructions (VALU-issue-bound: ~2 300 per row).  P and dS' (hi + lo bf16 halves: fp32-exact products against the bf16
  // S, dP') go through an LDS image per wave - [16 rows = 8 heads hi | 8 heads lo][256 keys], rows 528 B apart so that the 16 rows
  // of a fragment read hit 16 different bank groups - and v_mfma_f32_16x16x32_bf16 accumulates C[hi | lo head][head] over the rows a
  // wave processes: 16 MFMAs + ~100 LDS operations + 128 conversions per row.  Other H keep the packed / scalar vector forms.
  constexpr bool MO = (H == 8);
  constexpr bool PKO = PK && !MO;  // vector form of the outer products, packed over key pairs, four heads at a time
  constexpr int MO_PITCH = 528, MO_A = 16 * MO_PITCH, MO_WAVE = MO_A + 8 * MO_PITCH;
  __shared__ __attribute__((aligned(16))) char mo_lds[MO ? 4 * MO_WAVE : 16];
  __shared__ float red[4][2 * NB * NB * 64];
  f32x4 mo_c1 = {0.f, 0.f, 0.f, 0.f}, mo_c2 = {0.f, 0.f, 0.f, 0.f};  // C1[P hi | lo head][dP' head], C2[dS' hi | lo head][S head]
  [[maybe_unused]] char* mo_a = mo_lds + (MO ? (threadIdx.x >> 6) * MO_WAVE : 0);
  [[maybe_unused]] char* mo_b = mo_a + (MO ? MO_A : 0);
  // 8 k-steps of 32 keys; lanes holding keys >= N wrote zeros
  [[maybe_unused]] auto mo_mfma = [&](f32x4 c) {
    const int lane_ = threadIdx.x & 63;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const bf16x8 af = *reinterpret_cast<const bf16x8*>(mo_a + (lane_ & 15) * MO_PITCH + ks * 64 + (lane_ >> 4) * 16);
      const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(mo_b + (lane_ & 7) * MO_PITCH + ks * 64 + (lane_ >> 4) * 16);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr, c, 0, 0, 0);
    }
    return c;
  };
  const float* __restrict__ T1 = T1g;
  const float* __restrict__ T2 = T2g;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc1[NB * NB], acc2[NB * NB];  // lane l accumulates dT1 / dT2 entry (h = 8*hb + (l>>3), i = 8*ib + (l&7)) of tile hb*NB + ib
#pragma unroll
  for (int t = 0; t < NB * NB; ++t) acc1[t] = acc2[t] = 0.f;
  const long rows = (long)B * N;
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    const int b = (int)(row / N), q = (int)(row - (long)b * N);
    uint32_t sp[H][TH_KPL / 2], dq[H][TH_KPL / 2];
    // lane owns keys 4*lane .. 4*lane+3: one 8-byte load per head and tensor; pair k2 = keys (2*k2, 2*k2+1)
    const bool in_row = 4 * lane < Np;
    [[maybe_unused]] ThCoef<PK ? H : 4> c;
    if constexpr (PK) c.issue(T1g);  // under the latency of the row loads
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const size_t off = (((size_t)b * H + h) * N + q) * Np + 4 * lane;
      uint2 sv2 = make_uint2(0u, 0u), dv2 = make_uint2(0u, 0u);
      if (in_row) {
        sv2 = *reinterpret_cast<const uint2*>(S + off);
        dv2 = *reinterpret_cast<const uint2*>(dPp + off);
      }
      sp[h][0] = sv2.x; sp[h][1] = sv2.y;
      dq[h][0] = dv2.x; dq[h][1] = dv2.y;
    }
    // ---- forward recompute: P_i = softmax_k(sum_h T1[h][i] S_h)
    float pr[H][TH_KPL];
    if constexpr (PK) {  // packed head mixes: key pairs, coefficient pairs from SGPRs (see th_softmax_fwd_kernel)
      c.wait();
#pragma unroll
      for (int kp = 0; kp < 2; ++kp) {
        f32x2 sv2[H];
#pragma unroll
        for (int h = 0; h < H; ++h) sv2[h] = unpack_bf16x2(sp[h][kp]);
#pragma unroll
        for (int j = 0; j < H / 2; ++j) {
          f32x2 lo = {0.f, 0.f}, hi = {0.f, 0.f};
#pragma unroll
          for (int h = 0; h < H; ++h) {
            const uint64_t tp = TH_COEF(c, h, j);
            asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(lo) : "s"(tp), "v"(sv2[h]));
            asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(hi) : "s"(tp), "v"(sv2[h]));
          }
          pr[2 * j][2 * kp] = (4 * lane + 2 * kp < N) ? lo.x : -INFINITY;
          pr[2 * j][2 * kp + 1] = (4 * lane + 2 * kp + 1 < N) ? lo.y : -INFINITY;
          pr[2 * j + 1][2 * kp] = (4 * lane + 2 * kp < N) ? hi.x : -INFINITY;
          pr[2 * j + 1][2 * kp + 1] = (4 * lane + 2 * kp + 1 < N) ? hi.y : -INFINITY;
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < TH_KPL; ++k) {
        float sv[H];
#pragma unroll
        for (int h = 0; h < H; ++h) sv[h] = th_unpack(sp[h][k >> 1], k & 1);
#pragma unroll
        for (int i = 0; i < H; ++i) {
          float a = 0.f;
#pragma unroll
          for (int h = 0; h < H; ++h) a += T1[h * H + i] * sv[h];
          pr[i][k] = (4 * lane + k < N) ? a : -INFINITY;
        }
      }
    }
    if constexpr (PK) c.issue(T2g);  // T1's registers are free: T2 arrives under the softmax
    if constexpr (H == 8) {  // eight maxima / sums as one transposing reduction each (th_rows.h)
      float lm[8], m[8], ls[8], l[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) lm[i] = fmaxf(fmaxf(pr[i][0], pr[i][1]), fmaxf(pr[i][2], pr[i][3]));
      wave_reduce8<true>(lm, m, lane);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int k = 0; k < TH_KPL; ++k) pr[i][k] = __builtin_amdgcn_exp2f((pr[i][k] - m[i]) * LOG2E);
        ls[i] = (pr[i][0] + pr[i][1]) + (pr[i][2] + pr[i][3]);
      }
      wave_reduce8<false>(ls, l, lane);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float inv = 1.0f / l[i];
#pragma unroll
        for (int k = 0; k < TH_KPL; ++k) pr[i][k] *= inv;
      }
    } else {
#pragma unroll
      for (int i = 0; i < H; ++i) {
        float m = fmaxf(fmaxf(pr[i][0], pr[i][1]), fmaxf(pr[i][2], pr[i][3]));
        m = wave_max(m);
        float l = 0.f;
#pragma unroll
        for (int k = 0; k < TH_KPL; ++k) {
          pr[i][k] = __builtin_amdgcn_exp2f((pr[i][k] - m) * LOG2E);
          l += pr[i][k];
        }
        const float inv = 1.0f / wave_sum(l);
#pragma unroll
        for (int k = 0; k < TH_KPL; ++k) pr[i][k] *= inv;
      }
    }
    // ---- dT2[h][i] += sum_k P_h dP'_i ; dP_h = sum_i T2[h][i] dP'_i ; delta_h = sum_k P_h dP_h
    float dp[H][TH_KPL], del[H];
    if constexpr (H <= 8) {
      float g[64];
#pragma unroll
      for (int j = 0; j < 64; ++j) g[j] = 0.f;
#pragma unroll
      for (int h = 0; h < H; ++h) del[h] = 0.f;
      if constexpr (PK) {
        c.wait();
#pragma unroll
        for (int kp = 0; kp < 2; ++kp) {
          f32x2 dv2[H];
#pragma unroll
          for (int i = 0; i < H; ++i) dv2[i] = unpack_bf16x2(dq[i][kp]);
#pragma unroll
          for (int h = 0; h < H; ++h) {
            f32x2 a = {0.f, 0.f};
#pragma unroll
            for (int j = 0; j < H / 2; ++j) {  // a += T2[h][2j] dv[2j] + T2[h][2j+1] dv[2j+1]
              const uint64_t tp = TH_COEF(c, h, j);
              asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a) : "s"(tp), "v"(dv2[2 * j]));
              asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(a) : "s"(tp), "v"(dv2[2 * j + 1]));
            }
            if constexpr (!PKO) {
#pragma unroll
              for (int i = 0; i < H; ++i) {
                g[h * 8 + i] += pr[h][2 * kp] * dv2[i].x;
                g[h * 8 + i] += pr[h][2 * kp + 1] * dv2[i].y;
              }
            }
            dp[h][2 * kp] = a.x;
            dp[h][2 * kp + 1] = a.y;
            del[h] += pr[h][2 * kp] * a.x;
            del[h] += pr[h][2 * kp + 1] * a.y;
          }
        }
        if constexpr (PKO) {  // dT2 partials g[h][i] = sum over this lane's keys of P_h dP'_i: key pairs packed, four heads at a time
#pragma unroll
          for (int hh = 0; hh < H; hh += 4) {
            f32x2 g2[4][H];
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
#pragma unroll
              for (int i = 0; i < H; ++i) {
                const f32x2 dvi = unpack_bf16x2(dq[i][kp]);
#pragma unroll
                for (int h4 = 0; h4 < 4; ++h4) {
                  const f32x2 ph = f32x2{pr[hh + h4][2 * kp], pr[hh + h4][2 * kp + 1]};
                  g2[h4][i] = kp == 0 ? ph * dvi : g2[h4][i] + ph * dvi;
                }
              }
            }
#pragma unroll
            for (int h4 = 0; h4 < 4; ++h4)
#pragma unroll
              for (int i = 0; i < H; ++i) g[(hh + h4) * 8 + i] = g2[h4][i].x + g2[h4][i].y;
          }
        }
      } else {
#pragma unroll
        for (int k = 0; k < TH_KPL; ++k) {
          float dv[H];
#pragma unroll
          for (int i = 0; i < H; ++i) dv[i] = th_unpack(dq[i][k >> 1], k & 1);
#pragma unroll
          for (int h = 0; h < H; ++h) {
            float a = 0.f;
#pragma unroll
            for (int i = 0; i < H; ++i) {
              a += T2[h * H + i] * dv[i];
              g[h * 8 + i] += pr[h][k] * dv[i];
            }
            dp[h][k] = a;
            del[h] += pr[h][k] * a;
          }
        }
      }
      if constexpr (PK) c.issue(T1g);  // for the third mix, under the reduce-scatter and the delta reductions
      if constexpr (MO) {
#pragma unroll
        for (int h = 0; h < H; ++h) {  // P (hi, lo) -> rows h, 8 + h; dP' -> B row h; this lane's 4 keys = 8 bytes
          const uint32_t h0 = pack_bf16x2(pr[h][0], pr[h][1]), h1 = pack_bf16x2(pr[h][2], pr[h][3]);
          const f32x2 f0 = unpack_bf16x2(h0), f1 = unpack_bf16x2(h1);
          *reinterpret_cast<uint2*>(mo_a + h * MO_PITCH + lane * 8) = make_uint2(h0, h1);
          *reinterpret_cast<uint2*>(mo_a + (8 + h) * MO_PITCH + lane * 8) =
              make_uint2(pack_bf16x2(pr[h][0] - f0.x, pr[h][1] - f0.y), pack_bf16x2(pr[h][2] - f1.x, pr[h][3] - f1.y));
          *reinterpret_cast<uint2*>(mo_b + h * MO_PITCH + lane * 8) = make_uint2(dq[h][0], dq[h][1]);
        }
        mo_c1 = mo_mfma(mo_c1);
      } else {
        acc2[0] += reduce_scatter64(g, lane);
      }
    } else {
#pragma unroll
      for (int h = 0; h < H; ++h) del[h] = 0.f;
#pragma unroll
      for (int k = 0; k < TH_KPL; ++k) {
#pragma unroll
        for (int h = 0; h < H; ++h) {
          float a = 0.f;
#pragma unroll
          for (int i = 0; i < H; ++i) a += T2[h * H + i] * th_unpack(dq[i][k >> 1], k & 1);
          dp[h][k] = a;
          del[h] += pr[h][k] * a;
        }
      }
#pragma unroll
      for (int hb = 0; hb < NB; ++hb)
#pragma unroll
        for (int ib = 0; ib < NB; ++ib) {
          float g[64];
#pragma unroll
          for (int j = 0; j < 64; ++j) g[j] = 0.f;
#pragma unroll
          for (int k = 0; k < TH_KPL; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const float dvi = th_unpack(dq[ib * 8 + i][k >> 1], k & 1);
#pragma unroll
              for (int h = 0; h < 8; ++h) g[h * 8 + i] += pr[hb * 8 + h][k] * dvi;
            }
          acc2[hb * NB + ib] += reduce_scatter64(g, lane);
        }
    }
    if constexpr (H == 8) {
      float dl[8];
#pragma unroll
      for (int h = 0; h < 8; ++h) dl[h] = del[h];
      wave_reduce8<false>(dl, del, lane);
    } else {
#pragma unroll
      for (int h = 0; h < H; ++h) del[h] = wave_sum(del[h]);
    }
    // ---- dS'_i = P_i (dP_i - delta_i) ; dS_h = sum_i T1[h][i] dS'_i ; dT1[h][i] += sum_k S_h dS'_i
    if constexpr (H > 8) {
#pragma unroll
      for (int h = 0; h < H; ++h) {
        float o4[TH_KPL];
#pragma unroll
        for (int k = 0; k < TH_KPL; ++k) {
          float a = 0.f;
#pragma unroll
          for (int i = 0; i < H; ++i) a += T1[h * H + i] * (pr[i][k] * (dp[i][k] - del[i]));
          o4[k] = (4 * lane + k < N) ? a : 0.f;
        }
        if (in_row)
          *reinterpret_cast<uint2*>(dS + (((size_t)b * H + h) * N + q) * Np + 4 * lane) = make_uint2(pack_bf16x2(o4[0], o4[1]), pack_bf16x2(o4[2], o4[3]));
      }
#pragma unroll
      for (int hb = 0; hb < NB; ++hb)
#pragma unroll
        for (int ib = 0; ib < NB; ++ib) {
          float g[64];
#pragma unroll
          for (int j = 0; j < 64; ++j) g[j] = 0.f;
#pragma unroll
          for (int k = 0; k < TH_KPL; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const float dspi = pr[ib * 8 + i][k] * (dp[ib * 8 + i][k] - del[ib * 8 + i]);
#pragma unroll
              for (int h = 0; h < 8; ++h) g[h * 8 + i] += th_unpack(sp[hb * 8 + h][k >> 1], k & 1) * dspi;
            }
          acc1[hb * NB + ib] += reduce_scatter64(g, lane);
        }
    } else {
      float g[64];
#pragma unroll
      for (int j = 0; j < 64; ++j) g[j] = 0.f;
      float dsv[H][TH_KPL];
      if constexpr (PK) {
        c.wait();
#pragma unroll
        for (int kp = 0; kp < 2; ++kp) {
          f32x2 dsp2[H], sv2[H];
#pragma unroll
          for (int i = 0; i < H; ++i) {
            dsp2[i] = f32x2{pr[i][2 * kp] * (dp[i][2 * kp] - del[i]), pr[i][2 * kp + 1] * (dp[i][2 * kp + 1] - del[i])};
            sv2[i] = unpack_bf16x2(sp[i][kp]);
            if constexpr (MO) {  // dS' (hi, lo) -> rows i, 8 + i (this key pair = 4 bytes); S -> B row i
              const uint32_t hi_ = pack_bf16x2(dsp2[i].x, dsp2[i].y);
              const f32x2 f_ = unpack_bf16x2(hi_);
              *reinterpret_cast<uint32_t*>(mo_a + i * MO_PITCH + lane * 8 + kp * 4) = hi_;
              *reinterpret_cast<uint32_t*>(mo_a + (8 + i) * MO_PITCH + lane * 8 + kp * 4) = pack_bf16x2(dsp2[i].x - f_.x, dsp2[i].y - f_.y);
              *reinterpret_cast<uint32_t*>(mo_b + i * MO_PITCH + lane * 8 + kp * 4) = sp[i][kp];
            }
          }
#pragma unroll
          for (int h = 0; h < H; ++h) {
            f32x2 a = {0.f, 0.f};
#pragma unroll
            for (int j = 0; j < H / 2; ++j) {
              const uint64_t tp = TH_COEF(c, h, j);
              asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a) : "s"(tp), "v"(dsp2[2 * j]));
              asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(a) : "s"(tp), "v"(dsp2[2 * j + 1]));
            }
            if constexpr (!PKO) {
#pragma unroll
              for (int i = 0; i < H; ++i) {
                g[h * 8 + i] += sv2[h].x * dsp2[i].x;
                g[h * 8 + i] += sv2[h].y * dsp2[i].y;
              }
            }
            dsv[h][2 * kp] = (4 * lane + 2 * kp < N) ? a.x : 0.f;
            dsv[h][2 * kp + 1] = (4 * lane + 2 * kp + 1 < N) ? a.y : 0.f;
          }
        }
        if constexpr (PKO) {  // dT1 partials g[h][i] = sum over this lane's keys of S_h dS'_i, packed as above
#pragma unroll
          for (int hh = 0; hh < H; hh += 4) {
            f32x2 g2[4][H];
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
#pragma unroll
              for (int i = 0; i < H; ++i) {
                const f32x2 dsi = f32x2{pr[i][2 * kp] * (dp[i][2 * kp] - del[i]), pr[i][2 * kp + 1] * (dp[i][2 * kp + 1] - del[i])};
#pragma unroll
                for (int h4 = 0; h4 < 4; ++h4) {
                  const f32x2 sh = unpack_bf16x2(sp[hh + h4][kp]);
                  g2[h4][i] = kp == 0 ? sh * dsi : g2[h4][i] + sh * dsi;
                }
              }
            }
#pragma unroll
            for (int h4 = 0; h4 < 4; ++h4)
#pragma unroll
              for (int i = 0; i < H; ++i) g[(hh + h4) * 8 + i] = g2[h4][i].x + g2[h4][i].y;
          }
        }
      } else {
#pragma unroll
        for (int k = 0; k < TH_KPL; ++k) {
          float dsp[H], sv[H];
#pragma unroll
          for (int i = 0; i < H; ++i) {
            dsp[i] = pr[i][k] * (dp[i][k] - del[i]);
            sv[i] = th_unpack(sp[i][k >> 1], k & 1);
          }
#pragma unroll
          for (int h = 0; h < H; ++h) {
            float a = 0.f;
#pragma unroll
            for (int i = 0; i < H; ++i) {
              a += T1[h * H + i] * dsp[i];
              g[h * 8 + i] += sv[h] * dsp[i];
            }
            dsv[h][k] = (4 * lane + k < N) ? a : 0.f;
          }
        }
      }
      if (in_row) {
#pragma unroll
        for (int h = 0; h < H; ++h)
          *reinterpret_cast<uint2*>(dS + (((size_t)b * H + h) * N + q) * Np + 4 * lane) =
              make_uint2(pack_bf16x2(dsv[h][0], dsv[h][1]), pack_bf16x2(dsv[h][2], dsv[h][3]));
      }
      if constexpr (MO) {
        mo_c2 = mo_mfma(mo_c2);
      } else {
        acc1[0] += reduce_scatter64(g, lane);
      }
    }
  }
  if constexpr (MO) {
    // lane (n, quad) holds C[4 quad + j][n]: rows 0-7 the hi halves, 8-15 the lo halves.  dT1[h][i] = C2[i][h] + C2[8+i][h] goes to
    // red[wave][h*8 + i], dT2[h][i] = C1[h][i] + C1[8+h][i] to red[wave][64 + h*8 + i]: the lo quads add onto the hi quads' stores
    const int n_ = lane & 15, quad = lane >> 4;
    __syncthreads();  // (every wave is past its last fragment read)
    if (n_ < 8 && quad < 2) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        red[wave][n_ * 8 + 4 * quad + j] = mo_c2[j];
        red[wave][64 + (4 * quad + j) * 8 + n_] = mo_c1[j];
      }
    }
    __syncthreads();
    if (n_ < 8 && quad >= 2) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        red[wave][n_ * 8 + 4 * (quad - 2) + j] += mo_c2[j];
        red[wave][64 + (4 * (quad - 2) + j) * 8 + n_] += mo_c1[j];
      }
    }
  } else {
#pragma unroll
    for (int t = 0; t < NB * NB; ++t) {
      red[wave][t * 64 + lane] = acc1[t];
      red[wave][(NB * NB + t) * 64 + lane] = acc2[t];
    }
  }
  __syncthreads();
  // slab row: [dT1 (H*H) | dT2 (H*H)], entry (h, i) sits in tile (h/8, i/8), reduce-scatter lane (h%8)*8 + i%8
  for (int t = threadIdx.x; t < 2 * H * H; t += blockDim.x) {
    const int which = t / (H * H), e = t - which * H * H;
    const int h = e / H, i = e % H;
    const int src = (which * NB * NB + (h >> 3) * NB + (i >> 3)) * 64 + (h & 7) * 8 + (i & 7);
    slab[(size_t)blockIdx.x * 2 * H * H + t] = red[0][src] + red[1][src] + red[2][src] + red[3][src];
  }
}

}  // namespace

// development switch (SAVIT_EXPERIMENTS builds only): run the general online-softmax kernels where the resident ones would be chosen
constexpr int ATTN_RESIDENT_MAX_NT = 19;  // 608 tokens: the longest sequence whose K and V (or Q and dO) images of one head fit the LDS
#ifdef SAVIT_EXPERIMENTS
static const bool attn_stream_forced = SAVIT_EXP_ENV_INT("SAVIT_ATTN_STREAM", 0) != 0;  // A/B runs: the streaming kernels at every length
#else
constexpr bool attn_stream_forced = false;
#endif
static bool attn_force_general() {
  static const bool f = SAVIT_EXP_ENV_INT("SAVIT_ATTN_GENERAL", 0) != 0;
  return f;
}

// workgroups that can be resident at once: CUs x min(LDS, thread) occupancy (register limits are covered by launch bounds)
static int persistent_grid(int items, size_t lds_bytes, int threads) {
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
  }();
  int per_cu = (int)((160 * 1024) / (lds_bytes ? lds_bytes : 1));
  const int by_threads = 2048 / threads;
  if (per_cu > by_threads) per_cu = by_threads;
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 2) per_cu = 2;  // the kernels hold 128-256 VGPRs
  int use = savit_cu_budget_.load(std::memory_order_relaxed);  // savit_set_cu_budget: CUs not held by other kernels (0: all)
  if (use <= 0 || use > cus) use = cus;
  const long g = (long)use * per_cu;
  return (int)(items < g ? items : g);
}
#define ATTN_DISPATCH(KERNEL, LDS_EXPR, GRID)                                                                       \
  switch (nt) {                                                                                                \
    case 1: ATTN_CASE(KERNEL, 1, LDS_EXPR, GRID)                                                                   \
    case 2: ATTN_CASE(KERNEL, 2, LDS_EXPR, GRID)                                                                   \
    case 3: ATTN_CASE(KERNEL, 3, LDS_EXPR, GRID)                                                                   \
    case 4: ATTN_CASE(KERNEL, 4, LDS_EXPR, GRID)                                                                   \
    case 5: ATTN_CASE(KERNEL, 5, LDS_EXPR, GRID)                                                                   \
    case 6: ATTN_CASE(KERNEL, 6, LDS_EXPR, GRID)                                                                   \
    case 7: ATTN_CASE(KERNEL, 7, LDS_EXPR, GRID)                                                                   \
    case 8: ATTN_CASE(KERNEL, 8, LDS_EXPR, GRID)                                                                   \
    default: return SAVIT_EINVAL;                                                                              \
  }
#define ATTN_CASE(KERNEL, NTV, LDS_EXPR, GRID)                                                                    \
  {                                                                                                            \
    constexpr int NT = NTV;                                                                                    \
    const size_t lds = (LDS_EXPR);                                                                             \
    auto kfn = KERNEL<NTV>;                                                                                    \
    SAVIT_LDS_ONCE(kfn);                                                                                \
    hipLaunchKernelGGL(kfn, dim3(GRID), dim3(64 * NTV), lds, (hipStream_t)stream, p);                          \
  } break;

extern "C" int savit_attention_fwd(const void* qkv, void* o, float* lse, int B, int N, int H, int head_dim, int ld_qkv,
                                   void* stream) {
  SAVIT_CHECK_ARG(qkv && o && B >= 0 && N > 0 && H > 0);
  SAVIT_CHECK_ARG((head_dim == 64 || head_dim == 48 || head_dim == 32 || head_dim == 16) && N <= 65536 && ld_qkv >= 3 * H * head_dim && ld_qkv % 8 == 0);
  SAVIT_CHECK_ARG(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)o % 16) == 0);
  if (B == 0) return SAVIT_OK;
  AttnParams p{};
  p.qkv = (const bf16_t*)qkv; p.o = (bf16_t*)o; p.lse = lse;
  p.B = B; p.N = N; p.H = H; p.ld = ld_qkv; p.d = H * head_dim;
  const int nt = (N + 31) / 32;
  if (nt > ATTN_RESIDENT_MAX_NT || attn_stream_forced) {  // longer than one head's K / V images fit the LDS: the streaming kernels
    AttnParams2 pp{p, nt, head_dim};
    const int rounds = (nt + 7) / 8;
    const size_t lds = (size_t)4 * SEG_IMG;
    SAVIT_CHECK_ARG((long)B * H * rounds <= 0x7fffffffL);
#define ATTNS_LAUNCH(KERNEL, KSV, LDS)                                                                                             \
  {                                                                                                                                \
    auto kfn = KERNEL<KSV>;                                                                                                        \
    SAVIT_LDS_ONCE(kfn);                                                                                                           \
    hipLaunchKernelGGL(kfn, dim3((unsigned)(B * H * rounds)), dim3(512), (LDS), (hipStream_t)stream, pp);                           \
  } break;
    switch (head_dim / 16) {
      case 1: ATTNS_LAUNCH(attn_fwd_stream_kernel, 1, lds)
      case 2: ATTNS_LAUNCH(attn_fwd_stream_kernel, 2, lds)
      case 3: ATTNS_LAUNCH(attn_fwd_stream_kernel, 3, lds)
      default: ATTNS_LAUNCH(attn_fwd_stream_kernel, 4, lds)
    }
    SAVIT_LAUNCH_RET();
  }
  if (head_dim != HD || nt > 8 || attn_force_general()) {
    AttnParams2 pp{p, nt, head_dim};
    const size_t lds = (size_t)2 * nt * 32 * ROWB;
#define ATTN2_LAUNCH(KERNEL, KSV)                                                                                                  \
  {                                                                                                                                \
    auto kfn = KERNEL<KSV>;                                                                                                        \
    SAVIT_LDS_ONCE(kfn);                                                                                                           \
    hipLaunchKernelGGL(kfn, dim3(B * H), dim3(64 * (nt < 8 ? nt : 8)), lds, (hipStream_t)stream, pp);                              \
  } break;
    switch (head_dim / 16) {
      case 1: ATTN2_LAUNCH(attn_fwd2_kernel, 1)
      case 2: ATTN2_LAUNCH(attn_fwd2_kernel, 2)
      case 3: ATTN2_LAUNCH(attn_fwd2_kernel, 3)
      default: ATTN2_LAUNCH(attn_fwd2_kernel, 4)
    }
    SAVIT_LAUNCH_RET();
  }
  // persistent workgroups: as many as fit the CUs' LDS at once (two K/V image pairs each), every one walking over items
  static const int fwd_loader = SAVIT_EXP_ENV_INT("SAVIT_ATTN_FWD_LOADER", 1);
  if (fwd_loader && nt <= 7) {  // round 6: + one loader wave (attn_fwdl_kernel; nine waves at nt = 8 would not fit two per SIMD)
#define ATTN_CASE_FL(NTV)                                                                                          \
  {                                                                                                                \
    const size_t lds = (size_t)5 * NTV * 32 * ROWB;                                                                \
    auto kfn = attn_fwdl_kernel<NTV>;                                                                              \
    SAVIT_LDS_ONCE(kfn);                                                                                           \
    hipLaunchKernelGGL(kfn, dim3(persistent_grid(B * H, lds, 64 * (NTV + 1))), dim3(64 * (NTV + 1)), lds, (hipStream_t)stream, p); \
  } break;
    switch (nt) {
      case 1: ATTN_CASE_FL(1)
      case 2: ATTN_CASE_FL(2)
      case 3: ATTN_CASE_FL(3)
      case 4: ATTN_CASE_FL(4)
      case 5: ATTN_CASE_FL(5)
      case 6: ATTN_CASE_FL(6)
      default: ATTN_CASE_FL(7)
    }
    SAVIT_LAUNCH_RET();
  }
  ATTN_DISPATCH(attn_fwd_kernel, (size_t)5 * NT * 32 * ROWB, persistent_grid(B * H, (size_t)5 * NT * 32 * ROWB, 64 * NT))
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_attention_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv, int B, int N,
                                   int H, int head_dim, int ld_qkv, float dq_scale, void* stream) {
  SAVIT_CHECK_ARG(qkv && o && d_o && lse && dqkv && B >= 0 && N > 0 && H > 0);
  SAVIT_CHECK_ARG((head_dim == 64 || head_dim == 48 || head_dim == 32 || head_dim == 16) && N <= 65536 && ld_qkv >= 3 * H * head_dim && ld_qkv % 8 == 0);
  SAVIT_CHECK_ARG(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)o % 16) == 0 && ((uintptr_t)d_o % 16) == 0 && ((uintptr_t)dqkv % 16) == 0);
  if (B == 0) return SAVIT_OK;
  AttnParams p{};
  p.qkv = (const bf16_t*)qkv; p.o = (bf16_t*)const_cast<void*>(o); p.lse = const_cast<float*>(lse);
  p.d_o = (const bf16_t*)d_o; p.dqkv = (bf16_t*)dqkv;
  p.B = B; p.N = N; p.H = H; p.ld = ld_qkv; p.d = H * head_dim; p.dq_scale = dq_scale;
  const int nt = (N + 31) / 32;
  if (nt > ATTN_RESIDENT_MAX_NT || attn_stream_forced) {  // the streaming kernels: dQ (queries own, K / V stream), then dK / dV (keys own, Q / dO stream)
    AttnParams2 pp{p, nt, head_dim};
    const int rounds = (nt + 7) / 8;
    const size_t lds_q = (size_t)4 * SEG_IMG, lds_kv = lds_q + (size_t)4 * SEG_ROWS * sizeof(float);
    SAVIT_CHECK_ARG((long)B * H * rounds <= 0x7fffffffL);
    switch (head_dim / 16) {
      case 1: ATTNS_LAUNCH(attn_bwd_dq_stream_kernel, 1, lds_q)
      case 2: ATTNS_LAUNCH(attn_bwd_dq_stream_kernel, 2, lds_q)
      case 3: ATTNS_LAUNCH(attn_bwd_dq_stream_kernel, 3, lds_q)
      default: ATTNS_LAUNCH(attn_bwd_dq_stream_kernel, 4, lds_q)
    }
    if (hipGetLastError() != hipSuccess) return SAVIT_EINVAL;
    switch (head_dim / 16) {
      case 1: ATTNS_LAUNCH(attn_bwd_dkv_stream_kernel, 1, lds_kv)
      case 2: ATTNS_LAUNCH(attn_bwd_dkv_stream_kernel, 2, lds_kv)
      case 3: ATTNS_LAUNCH(attn_bwd_dkv_stream_kernel, 3, lds_kv)
      default: ATTNS_LAUNCH(attn_bwd_dkv_stream_kernel, 4, lds_kv)
    }
    SAVIT_LAUNCH_RET();
  }
  if (head_dim != HD || nt > 8 || attn_force_general()) {
    AttnParams2 pp{p, nt, head_dim};
    const size_t lds = (size_t)2 * nt * 32 * ROWB + (size_t)2 * nt * 32 * sizeof(float);
    switch (head_dim / 16) {
      case 1: ATTN2_LAUNCH(attn_bwd2_kernel, 1)
      case 2: ATTN2_LAUNCH(attn_bwd2_kernel, 2)
      case 3: ATTN2_LAUNCH(attn_bwd2_kernel, 3)
      default: ATTN2_LAUNCH(attn_bwd2_kernel, 4)
    }
    SAVIT_LAUNCH_RET();
  }
  // N <= 224: persistent workgroups (attn_bwd_pers_kernel).  Round 2 built a persistent form as well and dropped it (148 us against
  // 139): its transposed-fragment reads were the ds_read_tr16 builtin, in front of which hipcc waits for vmcnt(0) while an LDS-DMA is
  // in flight, so nothing overlapped - see TrFrag.  N = 225 .. 256 (four images + the transposition images exceed the LDS) and
  // -DATTN_BWD_ONE_ITEM experiment builds: one (batch, head) item per workgroup.
#ifndef ATTN_BWD_ONE_ITEM
  if (nt <= 7) {  // persistent form: four images + statistics + one 4 KB transposition image per wave (N <= 224: 145 KB of LDS)
#define ATTN_PERS_LDS ((size_t)4 * NT * 32 * ROWB + (size_t)2 * NT * 32 * sizeof(float) + (size_t)NT * 32 * ROWB)
    // round 6: NT compute waves + one loader wave that issues every LDS-DMA of the workgroup (attn_bwd_persl_kernel); the form
    // without it stays for experiment builds (SAVIT_ATTN_BWD_LOADER=0)
    static const int with_loader = SAVIT_EXP_ENV_INT("SAVIT_ATTN_BWD_LOADER", 1);
#define ATTN_CASE_L(NTV)                                                                                           \
  {                                                                                                                \
    constexpr int NT = NTV;                                                                                        \
    const size_t lds = ATTN_PERS_LDS;                                                                              \
    auto kfn = attn_bwd_persl_kernel<NTV>;                                                                         \
    SAVIT_LDS_ONCE(kfn);                                                                                           \
    hipLaunchKernelGGL(kfn, dim3(persistent_grid(B * H, lds, 64 * (NTV + 1))), dim3(64 * (NTV + 1)), lds, (hipStream_t)stream, p); \
  } break;
    if (with_loader) {
      switch (nt) {
        case 1: ATTN_CASE_L(1)
        case 2: ATTN_CASE_L(2)
        case 3: ATTN_CASE_L(3)
        case 4: ATTN_CASE_L(4)
        case 5: ATTN_CASE_L(5)
        case 6: ATTN_CASE_L(6)
        default: ATTN_CASE_L(7)
      }
      SAVIT_LAUNCH_RET();
    }
    switch (nt) {
      case 1: ATTN_CASE(attn_bwd_pers_kernel, 1, ATTN_PERS_LDS, persistent_grid(B * H, lds, 64 * NT))
      case 2: ATTN_CASE(attn_bwd_pers_kernel, 2, ATTN_PERS_LDS, persistent_grid(B * H, lds, 64 * NT))
      case 3: ATTN_CASE(attn_bwd_pers_kernel, 3, ATTN_PERS_LDS, persistent_grid(B * H, lds, 64 * NT))
      case 4: ATTN_CASE(attn_bwd_pers_kernel, 4, ATTN_PERS_LDS, persistent_grid(B * H, lds, 64 * NT))
      case 5: ATTN_CASE(attn_bwd_pers_kernel, 5, ATTN_PERS_LDS, persistent_grid(B * H, lds, 64 * NT))
      case 6: ATTN_CASE(attn_bwd_pers_kernel, 6, ATTN_PERS_LDS, persistent_grid(B * H, lds, 64 * NT))
      default: ATTN_CASE(attn_bwd_pers_kernel, 7, ATTN_PERS_LDS, persistent_grid(B * H, lds, 64 * NT))
    }
    SAVIT_LAUNCH_RET();
  }
  switch (nt) { default: ATTN_CASE(attn_bwd_kernel, 8, (size_t)4 * NT * 32 * ROWB + (size_t)2 * NT * 32 * sizeof(float), B * H) }
#else
  ATTN_DISPATCH(attn_bwd_kernel, (size_t)4 * NT * 32 * ROWB + (size_t)2 * NT * 32 * sizeof(float), B * H)
#endif
  SAVIT_LAUNCH_RET();
}

// ------------------------------------------------------------------------------------------------------------ talking heads (C ABI)
static int th_fill(ThParams& p, const void* qkv, int B, int N, int H, int head_dim, int ld_qkv, int Np) {
  if (!(qkv && B >= 0 && N > 0 && H > 0 && (head_dim == 48 || head_dim == 64) && N <= 256 && Np >= N && Np % 8 == 0 && Np <= 256 &&
        ld_qkv >= 3 * H * head_dim && ld_qkv % 8 == 0 && ((uintptr_t)qkv % 16) == 0))
    return SAVIT_EINVAL;
  p.qkv = (const bf16_t*)qkv; p.B = B; p.N = N; p.H = H; p.ld = ld_qkv; p.d = H * head_dim; p.Np = Np; p.nt = (N + 31) / 32; p.hd = head_dim;
  return SAVIT_OK;
}
#define TH_H_DISPATCH(KERNEL, GRID, ...)                                                                  \
  switch (H) {                                                                                             \
    case 2: hipLaunchKernelGGL(KERNEL<2>, dim3(GRID), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); break; \
    case 4: hipLaunchKernelGGL(KERNEL<4>, dim3(GRID), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); break; \
    case 6: hipLaunchKernelGGL(KERNEL<6>, dim3(GRID), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); break; \
    case 8: hipLaunchKernelGGL(KERNEL<8>, dim3(GRID), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); break; \
    case 16: hipLaunchKernelGGL(KERNEL<16>, dim3(GRID), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); break; \
    default: return SAVIT_EINVAL;                                                                          \
  }

extern "C" long savit_th_attention_bwd_workspace_bytes(int B, int N, int H) {
  long rows = (long)B * N, blocks = (rows + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  return blocks * 2 * H * H * (long)sizeof(float);
}

extern "C" int savit_th_attention_fwd(const void* qkv, const float* T1, const float* T2, void* s_buf, void* p_buf, void* o, int B, int N, int H,
                                      int head_dim, int ld_qkv, int Np, void* stream) {
  ThParams p{};
  int rc = th_fill(p, qkv, B, N, H, head_dim, ld_qkv, Np);
  if (rc) return rc;
  SAVIT_CHECK_ARG(T1 && T2 && s_buf && p_buf && o && (H == 2 || H == 4 || H == 6 || H == 8 || H == 16));
  if (B == 0) return SAVIT_OK;
  p.sbuf = (bf16_t*)s_buf; p.pbuf = (bf16_t*)p_buf; p.o = (bf16_t*)o;
  const size_t lds = (size_t)p.nt * 32 * ROWB;
  const int threads = 64 * (p.nt < 8 ? p.nt : 8);
  const size_t lds_scores = lds + (size_t)(threads / 64) * 32 * ROWB;  // + one 32-row tile per wave for the coalesced S stores
  SAVIT_LDS_ONCE(th_scores_kernel);
  SAVIT_LDS_ONCE(th_pv_kernel);
  hipLaunchKernelGGL(th_scores_kernel, dim3(B * H), dim3(threads), lds_scores, (hipStream_t)stream, p);
  long rows = (long)B * N, blocks = (rows + 3) / 4;
  if (blocks > 16384) blocks = 16384;
  TH_H_DISPATCH(th_softmax_fwd_kernel, (unsigned)blocks, (const bf16_t*)s_buf, (bf16_t*)p_buf, T1, T2, B, N, Np)
  hipLaunchKernelGGL(th_pv_kernel, dim3(B * H), dim3(threads), lds, (hipStream_t)stream, p);
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_th_attention_bwd(const void* qkv, const float* T1, const float* T2, const void* s_buf, void* p_buf, const void* d_o,
                                      void* ds_buf, void* dqkv, float* dT1, float* dT2, int B, int N, int H, int head_dim, int ld_qkv, int Np,
                                      float dq_scale, void* workspace, long workspace_bytes, void* stream) {
  ThParams p{};
  int rc = th_fill(p, qkv, B, N, H, head_dim, ld_qkv, Np);
  if (rc) return rc;
  SAVIT_CHECK_ARG(T1 && T2 && s_buf && p_buf && d_o && ds_buf && dqkv && dT1 && dT2 && (H == 2 || H == 4 || H == 6 || H == 8 || H == 16));
  SAVIT_CHECK_ARG(workspace && workspace_bytes >= savit_th_attention_bwd_workspace_bytes(B, N, H) && ((uintptr_t)workspace % 16) == 0);
  if (B == 0) return SAVIT_OK;
  const int threads = 64 * (p.nt < 8 ? p.nt : 8);
  const size_t lds = (size_t)p.nt * 32 * ROWB + (size_t)(threads / 64) * 32 * ROWB;  // staged image + one 32-row tile per wave
  SAVIT_LDS_ONCE(th_bwd_kernel<0>);
  SAVIT_LDS_ONCE(th_bwd_kernel<1>);
  // 1) dP' (into ds_buf) and dV from P' (p_buf) and dO
  p.o = (bf16_t*)const_cast<void*>(d_o); p.dqkv = (bf16_t*)dqkv; p.dq_scale = dq_scale;
  p.sbuf = (bf16_t*)p_buf;   // pass B input: P'
  p.pbuf = (bf16_t*)ds_buf;  // pass A output: dP'
  hipLaunchKernelGGL(th_bwd_kernel<0>, dim3(B * H), dim3(threads), lds, (hipStream_t)stream, p);
  // 2) rows: dS (into p_buf: P' is dead now), dT1, dT2
  long rows = (long)B * N, blocks = (rows + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  TH_H_DISPATCH(th_softmax_bwd_kernel, (unsigned)blocks, (const bf16_t*)s_buf, (const bf16_t*)ds_buf, (bf16_t*)p_buf, T1, T2, (float*)workspace, B,
                N, Np)
  hipLaunchKernelGGL(th_dT_finalize_kernel, dim3((2 * H * H + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, (int)blocks,
                     H * H, dT1, dT2);
  // 3) dQ, dK from dS (p_buf)
  p.sbuf = (bf16_t*)p_buf;
  hipLaunchKernelGGL(th_bwd_kernel<1>, dim3(B * H), dim3(threads), lds, (hipStream_t)stream, p);
  SAVIT_LAUNCH_RET();
}

// dV from P' (p_buf) and dO, dQ and dK from dS (ds_buf): the MFMA half of the talking-heads backward on tensors the fused row kernel
// (th_fused.hip) materialised.
extern "C" int savit_th_attention_bwd_products(const void* qkv, const void* p_buf, const void* ds_buf, const void* d_o, void* dqkv, int B, int N,
                                               int H, int head_dim, int ld_qkv, int Np, float dq_scale, void* stream) {
  ThParams p{};
  int rc = th_fill(p, qkv, B, N, H, head_dim, ld_qkv, Np);
  if (rc) return rc;
  SAVIT_CHECK_ARG(p_buf && ds_buf && d_o && dqkv && ((uintptr_t)d_o % 16) == 0 && ((uintptr_t)dqkv % 16) == 0);
  if (B == 0) return SAVIT_OK;
  const int threads = 64 * (p.nt < 8 ? p.nt : 8);
  const size_t lds = (size_t)p.nt * 32 * ROWB + (size_t)(threads / 64) * 32 * ROWB;
  SAVIT_LDS_ONCE(th_bwd_kernel<2>);
  SAVIT_LDS_ONCE(th_bwd_kernel<1>);
  p.o = (bf16_t*)const_cast<void*>(d_o); p.dqkv = (bf16_t*)dqkv; p.dq_scale = dq_scale;
  p.sbuf = (bf16_t*)const_cast<void*>(p_buf);
  hipLaunchKernelGGL(th_bwd_kernel<2>, dim3(B * H), dim3(threads), lds, (hipStream_t)stream, p);
  p.sbuf = (bf16_t*)const_cast<void*>(ds_buf);
  hipLaunchKernelGGL(th_bwd_kernel<1>, dim3(B * H), dim3(threads), lds, (hipStream_t)stream, p);
  SAVIT_LAUNCH_RET();
}
