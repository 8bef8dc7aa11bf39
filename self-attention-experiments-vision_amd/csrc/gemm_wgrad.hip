// Weight-gradient GEMM  dW[Kin, Nout] += X[M, Kin]^T . dY[M, Nout]   (bf16 in, fp32 accumulate).
//
// Backward of every nn.Dense / nn.DenseGeneral kernel on the hot path (reference: jax.value_and_grad at
// /root/reference/train.py:94-95 differentiating attention.py:29-37,60-63, ff.py:26-31, patch_embed.py:23-25,
// vit.py:96-98).  The reduction runs over the token dimension M (25k-148k), both operands are row-major
// activations, i.e. the reduction index is the SLOW index of both.  gfx950 design:
//   * tiles of 64 tokens x BI (X) and 64 tokens x BJ (dY) go HBM -> LDS by LDS-DMA (buffer_load ... lds);
//     rows >= M are zero-filled by the descriptor's bounds check;
//   * MFMA fragments are read with ds_read_b64_tr_b16 (hardware transpose: 4 tokens x 16 columns per
//     16-lane group), so no transposed copy of any activation ever exists in HBM;
//   * the 16-B chunk index inside each 256-B LDS segment is XORed with (token&3)<<2 (on the DMA source
//     address and on the read address) so the 4 token rows of a transpose block hit 4 different bank groups;
//   * v_mfma_f32_32x32x16_bf16, un-swapped: one accumulator register = 32 consecutive columns of 2 rows =
//     two 128-B segments, the shape global fp32 atomics run at full rate with;
//   * split over M: grid.y groups each own a contiguous token range and add their partial tile into the
//     fp32 gradient with global_atomic_add_f32 (order-dependent in the last bits, like any atomic sum).
#include "common.h"
#include "savit.h"
#include <type_traits>

namespace {

constexpr int TK = 64;  // tokens per reduction tile

struct WgradParams {
  const bf16_t* X;
  const bf16_t* dY;
  float* dW;
  int M, Kin, Nout, ldx, lddy, lddw;
  int tiles_i, tiles_j, splits, tiles_per_split;
  int patch, img_size, tokens, token_offset, grid_side, chunks_per_prow;
  int abl;  // SAVIT_EXPERIMENTS builds only: timing ablation (SAVIT_WGRAD_ABL): 1 = skip the epilogue
  float* slab;       // non-null: every split stores its partial tile to slab + split * slab_stride (dense [Kin, Nout]) with plain
  long slab_stride;  // stores and wgrad_reduce_kernel sums the splits into dW: no atomics, bitwise reproducible
  int no_reduce;     // slab form only: leave the ordered sum to a later savit_gemm_wgrad_reduce call
  int rmw;           // grouped launches (one workgroup per output tile, no split): dW += tile with plain loads / stores
  int overwrite;     // rmw form: dW = tile (the FIRST and only touch of these elements this step: no read, and no memset of dW before it)
  float* sumsq;      // rmw form, nullable: 32 accumulators; this workgroup adds the sum of squares of what it stored to sumsq[block & 31]
                     // (the global gradient norm of optax.clip_by_global_norm, train.py:25, without a pass over the weight gradients)
};

// Epilogue of the one-workgroup-per-tile forms: dW (+)= acc for one 32 x 32 accumulator; returns the sum of squares of what was stored.
__device__ __forceinline__ float wgrad_store_tile(const WgradParams& p, const f32x16& acc, int ibase, int j, int hi5) {
  float ss = 0.f;
  if (p.overwrite) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = ibase + (r & 3) + 8 * (r >> 2) + 4 * hi5;
      if (i < p.Kin && j < p.Nout) {
        p.dW[(size_t)i * p.lddw + j] = acc[r];
        ss += acc[r] * acc[r];
      }
    }
    return ss;
  }
  float old[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = ibase + (r & 3) + 8 * (r >> 2) + 4 * hi5;
    old[r] = (i < p.Kin && j < p.Nout) ? p.dW[(size_t)i * p.lddw + j] : 0.f;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = ibase + (r & 3) + 8 * (r >> 2) + 4 * hi5;
    if (i < p.Kin && j < p.Nout) {
      const float v = old[r] + acc[r];
      p.dW[(size_t)i * p.lddw + j] = v;
      ss += v * v;
    }
  }
  return ss;
}
// one atomic per WAVE into one of 32 accumulators (a single address would serialise 2 048 adds per launch at the memory side)
__device__ __forceinline__ void wgrad_publish_sumsq(const WgradParams& p, float ss) {
  if (p.sumsq == nullptr) return;
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) atomicAdd(p.sumsq + ((blockIdx.x * 8 + (threadIdx.x >> 6)) & 31), ss);
}

__device__ __forceinline__ bf16x4 ds_read_tr16_b64(const char* p) {
  // hardware transpose read; the builtin lets hipcc count the read in lgkmcnt and fold constant offsets
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(p));
}

template <int BI, int BJ, int WGI, int WGJ, bool PATCH>
__global__ __launch_bounds__(64 * WGI * WGJ) void gemm_wgrad_kernel(const WgradParams p) {
  constexpr int NW = WGI * WGJ;
  constexpr int WTI = BI / WGI, WTJ = BJ / WGJ;
  constexpr int II = WTI / 32, JJ = WTJ / 32;
  constexpr int XROW = BI * 2, YROW = BJ * 2;           // LDS row bytes
  constexpr int X_BYTES = TK * XROW, Y_BYTES = TK * YROW;
  constexpr int STAGE_BYTES = X_BYTES + Y_BYTES;
  constexpr int X_INSTR = X_BYTES / 1024 / NW, Y_INSTR = Y_BYTES / 1024 / NW;
  static_assert(X_BYTES % (1024 * NW) == 0 && Y_BYTES % (1024 * NW) == 0, "tile/wave mismatch");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wi = wave / WGJ, wj = wave % WGJ;

  const int ntile = p.tiles_i * p.tiles_j;
  const int tid = blockIdx.x;
  const int ti = tid / p.tiles_j, tj = tid - ti * p.tiles_j;
  const int i0 = ti * BI, j0 = tj * BJ;
  const int split = blockIdx.y;
  const int kt_begin = split * p.tiles_per_split;
  const int kt_total = (p.M + TK - 1) / TK;
  int kt_end = kt_begin + p.tiles_per_split;
  if (kt_end > kt_total) kt_end = kt_total;
  if (kt_begin >= kt_end) return;
  (void)ntile;

  // descriptors: X over the whole activation (or image) buffer, dY likewise; token rows >= M must read zero,
  // so token validity is folded into the per-lane offset (0xfffffff0 is out of range by construction).
  size_t xbytes = PATCH ? (size_t)(p.M / (p.grid_side * p.grid_side)) * p.img_size * p.img_size * 6 : (size_t)p.M * p.ldx * 2;
  size_t ybytes = PATCH ? (size_t)(p.M / (p.grid_side * p.grid_side)) * p.tokens * p.lddy * 2 : (size_t)p.M * p.lddy * 2;
  if (xbytes > 0xffffffe0ull) xbytes = 0xffffffe0ull;
  if (ybytes > 0xffffffe0ull) ybytes = 0xffffffe0ull;
  const auto srdX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.X), 0, (uint32_t)xbytes, 0x00020000);
  const auto srdY = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.dY), 0, (uint32_t)ybytes, 0x00020000);

  // staging: a wave-instruction writes 1 KiB = (1024/ROW) rows; lane -> (row, physical chunk)
  auto stage = [&](int kt, int buf) {
    char* sX = smem + buf * STAGE_BYTES;
    char* sY = sX + X_BYTES;
    const int m0 = kt * TK;
    {
      constexpr int LPR = XROW / 16;  // lanes (chunks) per row
      constexpr int RPI = 64 / LPR;   // rows per instruction
#pragma unroll
      for (int i = 0; i < X_INSTR; ++i) {
        const int inst = wave * X_INSTR + i;
        const int r = inst * RPI + lane / LPR;  // token row inside the tile
        const int pc = lane % LPR;
        const int c = pc ^ ((r & 3) << 2);     // logical chunk (XOR acts inside each 256-B segment)
        const int m = m0 + r;
        uint32_t voff = 0xfffffff0u;
        if (m < p.M) {
          if (PATCH) {
            const int ppi = p.grid_side * p.grid_side;
            const int b = m / ppi, pp = m - b * ppi;
            const int pi = pp / p.grid_side, pj = pp - pi * p.grid_side;
            const int kc = i0 / 8 + c;
            const int ph = kc / p.chunks_per_prow, within = kc - ph * p.chunks_per_prow;
            const size_t pix = ((size_t)b * p.img_size + (size_t)pi * p.patch + ph) * p.img_size + (size_t)pj * p.patch;
            voff = (uint32_t)(pix * 6 + (size_t)within * 16);
          } else {
            voff = (uint32_t)((size_t)m * p.ldx * 2 + (size_t)(i0 + c * 8) * 2);
          }
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srdX, (__attribute__((address_space(3))) void*)(sX + inst * 1024), 16, voff, 0, 0, 0);
      }
    }
    {
      constexpr int LPR = YROW / 16;
      constexpr int RPI = 64 / LPR;
#pragma unroll
      for (int i = 0; i < Y_INSTR; ++i) {
        const int inst = wave * Y_INSTR + i;
        const int r = inst * RPI + lane / LPR;
        const int pc = lane % LPR;
        const int c = pc ^ ((r & 3) << 2);
        const int m = m0 + r;
        uint32_t voff = 0xfffffff0u;
        if (m < p.M) {
          size_t row = m;
          if (PATCH) {
            const int ppi = p.grid_side * p.grid_side;
            const int b = m / ppi, pp = m - b * ppi;
            row = (size_t)b * p.tokens + p.token_offset + pp;
          }
          voff = (uint32_t)(row * p.lddy * 2 + (size_t)(j0 + c * 8) * 2);
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srdY, (__attribute__((address_space(3))) void*)(sY + inst * 1024), 16, voff, 0, 0, 0);
      }
    }
  };

  // transpose-read geometry (32x32x16 operand): 16-lane group g: column sub-block (g&1), k-half h = g>>1;
  // lane t of the group addresses token row q = t>>2, columns 4*(t&3)..+3 of the 4x16 block.
  const int g = lane >> 4, t = lane & 15;
  const int q = t >> 2, h = g >> 1;
  const int colx = wi * WTI + 16 * (g & 1) + 4 * (t & 3);  // + 32*ii
  const int coly = wj * WTJ + 16 * (g & 1) + 4 * (t & 3);  // + 32*jj
  auto tr_off = [&](int col, int rowbytes, int tok) -> int {
    const int chunk = col >> 3;
    const int pch = chunk ^ ((tok & 3) << 2);
    return tok * rowbytes + pch * 16 + (col & 7) * 2;
  };

  f32x16 acc[II][JJ];
#pragma unroll
  for (int a = 0; a < II; ++a)
#pragma unroll
    for (int b = 0; b < JJ; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  stage(kt_begin, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int cur = (kt - kt_begin) & 1;
    if (kt + 1 < kt_end) stage(kt + 1, cur ^ 1);
    const char* sX = smem + cur * STAGE_BYTES;
    const char* sY = sX + X_BYTES;
#pragma unroll
    for (int ks = 0; ks < TK / 16; ++ks) {
      const int tok0 = 16 * ks + 8 * h + q;  // first transpose block; second is +4 tokens
      bf16x8 xa[II], yb[JJ];
#pragma unroll
      for (int a = 0; a < II; ++a) {
        const bf16x4 lo = ds_read_tr16_b64(sX + tr_off(colx + 32 * a, XROW, tok0));
        const bf16x4 hi = ds_read_tr16_b64(sX + tr_off(colx + 32 * a, XROW, tok0 + 4));
        xa[a] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int b = 0; b < JJ; ++b) {
        const bf16x4 lo = ds_read_tr16_b64(sY + tr_off(coly + 32 * b, YROW, tok0));
        const bf16x4 hi = ds_read_tr16_b64(sY + tr_off(coly + 32 * b, YROW, tok0 + 4));
        yb[b] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int a = 0; a < II; ++a)
#pragma unroll
        for (int b = 0; b < JJ; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[a], yb[b], acc[a][b], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // epilogue: D[i][j]: j = lane&31, i = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const int jl = lane & 31, hi5 = lane >> 5;
#pragma unroll
  for (int a = 0; a < II; ++a)
#pragma unroll
    for (int b = 0; b < JJ; ++b) {
      const int j = j0 + wj * WTJ + 32 * b + jl;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = i0 + wi * WTI + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * hi5;
        if (i < p.Kin && j < p.Nout) atomicAdd(p.dW + (size_t)i * p.lddw + j, acc[a][b][r]);
      }
    }
}


// ------------------------------------------------------------------------------------------------------------
// Ring-pipelined variant: 32-token stages in an S-slot LDS ring filled by LDS-DMA with counted vmcnt (loads stay in
// flight across the per-stage barrier) and register-prefetched transposed fragments (the fragments of k-step t+1 are
// read from LDS while the MFMAs of k-step t issue).  One barrier per stage, placed between its two k-steps: every read
// of a stage's slot is issued before that barrier, so the slot is refilled right after it.
// One output tile (ti, tj) of one token range (`split`): the body shared by the single-problem kernel and the grouped one.
template <int BI, int BJ, int WGI, int WGJ, int S, bool PATCH>
__device__ __forceinline__ void wgrad_ring_tile(const WgradParams& p, const int split, const int ti, const int tj, char* smem) {
  constexpr int NW = WGI * WGJ;
  constexpr int TS = 32;  // tokens per stage
  constexpr int WTI = BI / WGI, WTJ = BJ / WGJ;
  constexpr int II = WTI / 32, JJ = WTJ / 32;
  constexpr int XROW = BI * 2, YROW = BJ * 2;
  constexpr int X_BYTES = TS * XROW, Y_BYTES = TS * YROW, STAGE = X_BYTES + Y_BYTES;
  constexpr int X_INSTR = X_BYTES / 1024 / NW, Y_INSTR = Y_BYTES / 1024 / NW, G = X_INSTR + Y_INSTR;
  static_assert(X_BYTES % (1024 * NW) == 0 && Y_BYTES % (1024 * NW) == 0, "tile/wave mismatch");
  static_assert(G * (S - 1) <= 63, "vmcnt immediate");

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wi = wave / WGJ, wj = wave % WGJ;
  const int i0 = ti * BI, j0 = tj * BJ;
  const int st_total = (p.M + TS - 1) / TS;
  const int st_begin = split * p.tiles_per_split;  // in 32-token stages
  int st_end = st_begin + p.tiles_per_split;
  if (st_end > st_total) st_end = st_total;
  const int NS = st_end - st_begin;
  if (NS <= 0) return;

  size_t xbytes = PATCH ? (size_t)(p.M / (p.grid_side * p.grid_side)) * p.img_size * p.img_size * 6 : (size_t)p.M * p.ldx * 2;
  size_t ybytes = PATCH ? (size_t)(p.M / (p.grid_side * p.grid_side)) * p.tokens * p.lddy * 2 : (size_t)p.M * p.lddy * 2;
  if (xbytes > 0xffffffe0ull) xbytes = 0xffffffe0ull;
  if (ybytes > 0xffffffe0ull) ybytes = 0xffffffe0ull;
  const auto srdX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.X), 0, (uint32_t)xbytes, 0x00020000);
  const auto srdY = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.dY), 0, (uint32_t)ybytes, 0x00020000);

  auto stage = [&](int st, int slot) {
    char* sX = smem + slot * STAGE;
    char* sY = sX + X_BYTES;
    const int m0 = (st_begin + st) * TS;
    {
      constexpr int LPR = XROW / 16, RPI = 64 / LPR;
#pragma unroll
      for (int i = 0; i < X_INSTR; ++i) {
        const int inst = wave * X_INSTR + i;
        const int r = inst * RPI + lane / LPR;
        const int pc = lane % LPR;
        const int c = pc ^ ((r & 3) << 2);
        const int m = m0 + r;
        uint32_t voff = 0xfffffff0u;
        if (m < p.M) {
          if (PATCH) {
            const int ppi = p.grid_side * p.grid_side;
            const int b = m / ppi, pp = m - b * ppi;
            const int pi = pp / p.grid_side, pj = pp - pi * p.grid_side;
            const int kc = i0 / 8 + c;
            const int ph = kc / p.chunks_per_prow, within = kc - ph * p.chunks_per_prow;
            const size_t pix = ((size_t)b * p.img_size + (size_t)pi * p.patch + ph) * p.img_size + (size_t)pj * p.patch;
            voff = (uint32_t)(pix * 6 + (size_t)within * 16);
          } else {
            voff = (uint32_t)((size_t)m * p.ldx * 2 + (size_t)(i0 + c * 8) * 2);
          }
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srdX, (__attribute__((address_space(3))) void*)(sX + inst * 1024), 16, voff, 0, 0, 0);
      }
    }
    {
      constexpr int LPR = YROW / 16, RPI = 64 / LPR;
#pragma unroll
      for (int i = 0; i < Y_INSTR; ++i) {
        const int inst = wave * Y_INSTR + i;
        const int r = inst * RPI + lane / LPR;
        const int pc = lane % LPR;
        const int c = pc ^ ((r & 3) << 2);
        const int m = m0 + r;
        uint32_t voff = 0xfffffff0u;
        if (m < p.M) {
          size_t row = m;
          if (PATCH) {
            const int ppi = p.grid_side * p.grid_side;
            const int b = m / ppi, pp = m - b * ppi;
            row = (size_t)b * p.tokens + p.token_offset + pp;
          }
          voff = (uint32_t)(row * p.lddy * 2 + (size_t)(j0 + c * 8) * 2);
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srdY, (__attribute__((address_space(3))) void*)(sY + inst * 1024), 16, voff, 0, 0, 0);
      }
    }
  };

  // transposed-fragment addressing (see the 2-stage kernel above): lane -> token q (+8h) of a 4x16 block, columns col..col+3.
  // The reads are INLINE ASM on purpose: hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of the
  // __builtin_amdgcn_ds_read_tr16_b64 builtin whenever an LDS-DMA is in flight (it cannot prove the two do not alias),
  // which drains the ring every k-step.  The asm form is invisible to that pass; completion is waited for by
  // `frag_wait` (s_waitcnt lgkmcnt(0) that names every destination "+v", so no consumer or copy can be scheduled above it).
  const int g = lane >> 4, t = lane & 15;
  const int q = t >> 2, h = g >> 1;
  const int colx = wi * WTI + 16 * (g & 1) + 4 * (t & 3);
  const int coly = wj * WTJ + 16 * (g & 1) + 4 * (t & 3);
  const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
  uint32_t offx[II], offy[JJ];  // LDS byte address (slot 0) of token row (8h + q), k-step 0
#pragma unroll
  for (int a = 0; a < II; ++a) {
    const int col = colx + 32 * a;
    offx[a] = lds0 + (8 * h + q) * XROW + (((col >> 3) ^ (q << 2)) << 4) + ((col & 7) << 1);
  }
#pragma unroll
  for (int b = 0; b < JJ; ++b) {
    const int col = coly + 32 * b;
    offy[b] = lds0 + X_BYTES + (8 * h + q) * YROW + (((col >> 3) ^ (q << 2)) << 4) + ((col & 7) << 1);
  }

  f32x16 acc[II][JJ];
#pragma unroll
  for (int a = 0; a < II; ++a)
#pragma unroll
    for (int b = 0; b < JJ; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  struct Frags {
    bf16x4 xl[II], xh[II], yl[JJ], yh[JJ];
  };
  Frags f0, f1;
#define SAVIT_TR_READ(dst, addr, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF))
#define SAVIT_ISSUE_FRAGS(f, slot_off, KS)                                              \
  do {                                                                                  \
    _Pragma("unroll") for (int a_ = 0; a_ < II; ++a_) {                                 \
      SAVIT_TR_READ((f).xl[a_], offx[a_] + (slot_off), (KS) * 16 * XROW);               \
      SAVIT_TR_READ((f).xh[a_], offx[a_] + (slot_off), (KS) * 16 * XROW + 4 * XROW);    \
    }                                                                                   \
    _Pragma("unroll") for (int b_ = 0; b_ < JJ; ++b_) {                                 \
      SAVIT_TR_READ((f).yl[b_], offy[b_] + (slot_off), (KS) * 16 * YROW);               \
      SAVIT_TR_READ((f).yh[b_], offy[b_] + (slot_off), (KS) * 16 * YROW + 4 * YROW);    \
    }                                                                                   \
  } while (0)
  auto frag_wait = [&](Frags& f) {
#pragma unroll
    for (int a = 0; a < II; ++a) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.xl[a]), "+v"(f.xh[a]));
#pragma unroll
    for (int b = 0; b < JJ; ++b) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.yl[b]), "+v"(f.yh[b]));
  };
  auto mfmas = [&](const Frags& f) {
    bf16x8 xa[II], yb[JJ];
#pragma unroll
    for (int a = 0; a < II; ++a) xa[a] = __builtin_shufflevector(f.xl[a], f.xh[a], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
    for (int b = 0; b < JJ; ++b) yb[b] = __builtin_shufflevector(f.yl[b], f.yh[b], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
    for (int a = 0; a < II; ++a)
#pragma unroll
      for (int b = 0; b < JJ; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[a], yb[b], acc[a][b], 0, 0, 0);
  };

  const int pre = NS < S ? NS : S;
  for (int st = 0; st < pre; ++st) stage(st, st);
  if (pre == S) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G * (S - 1)) : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  SAVIT_ISSUE_FRAGS(f0, 0u, 0);

  int slot = 0;
  for (int st = 0; st < NS; ++st) {
    const uint32_t cur_off = (uint32_t)(slot * STAGE);
    const int nslot = (slot + 1 == S) ? 0 : slot + 1;
    // k-step 0 of this stage; its fragments were requested one half-step ago
    frag_wait(f0);
    SAVIT_ISSUE_FRAGS(f1, cur_off, 1);
    mfmas(f0);
    // every read of this slot must be complete before the barrier (the slot is refilled right after it), and
    // stage st+1 must have landed before its first fragments are requested below
    frag_wait(f1);
    if (st + S - 1 < NS) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G * (S - 2)) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (st + S < NS) stage(st + S, slot);
    // k-step 1, requesting k-step 0 of the next stage (after the last stage: a valid slot, values unused)
    SAVIT_ISSUE_FRAGS(f0, (uint32_t)(nslot * STAGE), 0);
    mfmas(f1);
    slot = nslot;
  }
  frag_wait(f0);  // nothing may stay in flight into the epilogue
#undef SAVIT_ISSUE_FRAGS
#undef SAVIT_TR_READ
#ifdef SAVIT_EXPERIMENTS
  if ((p.abl & 1) && p.M > 0) return;  // timing-only: the accumulators stay live (the condition is a run-time value)
#endif

  const int jl = lane & 31, hi5 = lane >> 5;
  if (p.rmw) {
    // the tile IS the whole sum over the tokens (no split): dW (+)= tile with plain accesses (one accumulator register = 2 rows x 128 B:
    // whole lines per wave-instruction); no other workgroup touches these elements
    float ss = 0.f;
#pragma unroll
    for (int a = 0; a < II; ++a)
#pragma unroll
      for (int b = 0; b < JJ; ++b) ss += wgrad_store_tile(p, acc[a][b], i0 + wi * WTI + 32 * a, j0 + wj * WTJ + 32 * b + jl, hi5);
    wgrad_publish_sumsq(p, ss);
    return;
  }
  if (p.slab != nullptr) {
    // one accumulator register = 2 rows x 128 B: whole lines per wave-instruction, the shape plain dword stores run at the HBM rate with
    float* out = p.slab + (size_t)split * p.slab_stride;
#pragma unroll
    for (int a = 0; a < II; ++a)
#pragma unroll
      for (int b = 0; b < JJ; ++b) {
        const int j = j0 + wj * WTJ + 32 * b + jl;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = i0 + wi * WTI + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * hi5;
          if (i < p.Kin && j < p.Nout) out[(size_t)i * p.Nout + j] = acc[a][b][r];
        }
      }
    return;
  }
#pragma unroll
  for (int a = 0; a < II; ++a)
#pragma unroll
    for (int b = 0; b < JJ; ++b) {
      const int j = j0 + wj * WTJ + 32 * b + jl;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = i0 + wi * WTI + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * hi5;
        if (i < p.Kin && j < p.Nout) atomicAdd(p.dW + (size_t)i * p.lddw + j, acc[a][b][r]);
      }
    }
}

template <int BI, int BJ, int WGI, int WGJ, int S, bool PATCH>
__global__ __launch_bounds__(64 * WGI * WGJ, 2) void gemm_wgrad_ring_kernel(const WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // 1-D grid, XCD-aware: consecutive remapped ids = the tiles of ONE token range (split), so the workgroups that
  // stream the same X / dY rows sit on the same XCD and share them through its L2 (measured before the remap: 653 MB
  // of L2 misses per launch for 194 MB of operands - every XCD re-fetched every token range).
  const int ntile = p.tiles_i * p.tiles_j;
  const int wid = xcd_remap(blockIdx.x, ntile * p.splits);
  const int split = wid / ntile;
  const int tid = wid - split * ntile;
  const int ti = tid / p.tiles_j;
  wgrad_ring_tile<BI, BJ, WGI, WGJ, S, PATCH>(p, split, ti, tid - ti * p.tiles_j, smem);
}

// ------------------------------------------------------------------------------------------------------------
// Ping-pong form of the ring tile for the grouped launches (one tile per workgroup over ALL tokens: no split, dW += tile).  In
// wgrad_ring_tile the eight waves move in lock step: the two waves of a SIMD want the matrix pipe at the same time and wait for
// their LDS fragments at the same time.  Here the waves of row half 0 (wi = 0: one per SIMD) and of row half 1 run ONE BARRIER
// APART through four phases per 32-token stage,
//     L0: request the fragments of k-step 0        | C0: wait for them, 8 MFMAs
//     L1: request k-step 1, refill a ring slot     | C1: wait, 8 MFMAs
// so that between two barriers one group issues MFMAs while its SIMD partners request fragments / issue LDS-DMA (the structure of
// gemm_tn_pp_kernel).  Same K order per accumulator as the ring tile: the results are bitwise the same.
//   WAR: the slot of stage st-1 is refilled (stage st-1+S) in L1(st): both groups have waited for their last fragments of stage
//        st-1 by then (group 1's C1(st-1) ended one barrier before group 0's L1(st) began, and vice versa).
//   RAW: every wave waits for ITS share of stage st+1 at the end of L1(st) (vmcnt(G (S-2)): all but the S-2 newest stages) and at
//        least one barrier follows before any wave requests fragments of stage st+1 (L0(st+1)).
template <int BI, int BJ, int WGI, int WGJ, int S>
__device__ __forceinline__ void wgrad_pp_tile(const WgradParams& p, const int ti, const int tj, char* smem) {
  static_assert(WGI == 2, "two row halves = the two ping-pong groups");
  constexpr int NW = WGI * WGJ;
  constexpr int TS = 32;
  constexpr int WTI = BI / WGI, WTJ = BJ / WGJ;
  constexpr int II = WTI / 32, JJ = WTJ / 32;
  constexpr int XROW = BI * 2, YROW = BJ * 2;
  constexpr int X_BYTES = TS * XROW, Y_BYTES = TS * YROW, STAGE = X_BYTES + Y_BYTES;
  constexpr int X_INSTR = X_BYTES / 1024 / NW, Y_INSTR = Y_BYTES / 1024 / NW, G = X_INSTR + Y_INSTR;
  static_assert(X_BYTES % (1024 * NW) == 0 && Y_BYTES % (1024 * NW) == 0, "tile/wave mismatch");
  static_assert(S >= 3 && G * (S - 1) <= 63, "ring depth / vmcnt immediate");
  static_assert(XROW % 256 == 0 && YROW % 256 == 0 && WTI % 32 == 0 && WTJ % 32 == 0, "rows are whole 256-B swizzle segments");

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wi = wave / WGJ, wj = wave % WGJ;
  const int i0 = ti * BI, j0 = tj * BJ;
  const int NS = (p.M + TS - 1) / TS;
  if (NS <= 0) return;

  size_t xbytes = (size_t)p.M * p.ldx * 2, ybytes = (size_t)p.M * p.lddy * 2;
  if (xbytes > 0xffffffe0ull) xbytes = 0xffffffe0ull;
  if (ybytes > 0xffffffe0ull) ybytes = 0xffffffe0ull;
  const auto srdX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.X), 0, (uint32_t)xbytes, 0x00020000);
  const auto srdY = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.dY), 0, (uint32_t)ybytes, 0x00020000);

  auto stage = [&](int st, int slot) {  // as in wgrad_ring_tile (no patch gather on this path)
    char* sX = smem + slot * STAGE;
    char* sY = sX + X_BYTES;
    const int m0 = st * TS;
    {
      constexpr int LPR = XROW / 16;  // 16-B chunks per row; a 1-KB piece of the linear image may span two rows (768-B rows: BI = 384)
#pragma unroll
      for (int i = 0; i < X_INSTR; ++i) {
        const int inst = wave * X_INSTR + i;
        const int L = inst * 64 + lane;
        const int r = L / LPR;
        const int c = (L - r * LPR) ^ ((r & 3) << 2);
        const int m = m0 + r;
        uint32_t voff = 0xfffffff0u;
        if (m < p.M) voff = (uint32_t)((size_t)m * p.ldx * 2 + (size_t)(i0 + c * 8) * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srdX, (__attribute__((address_space(3))) void*)(sX + inst * 1024), 16, voff, 0, 0, 0);
      }
    }
    {
      constexpr int LPR = YROW / 16;
#pragma unroll
      for (int i = 0; i < Y_INSTR; ++i) {
        const int inst = wave * Y_INSTR + i;
        const int L = inst * 64 + lane;
        const int r = L / LPR;
        const int c = (L - r * LPR) ^ ((r & 3) << 2);
        const int m = m0 + r;
        uint32_t voff = 0xfffffff0u;
        if (m < p.M) voff = (uint32_t)((size_t)m * p.lddy * 2 + (size_t)(j0 + c * 8) * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srdY, (__attribute__((address_space(3))) void*)(sY + inst * 1024), 16, voff, 0, 0, 0);
      }
    }
  };

  const int g = lane >> 4, t = lane & 15;
  const int q = t >> 2, h = g >> 1;
  const int colx = wi * WTI + 16 * (g & 1) + 4 * (t & 3);
  const int coly = wj * WTJ + 16 * (g & 1) + 4 * (t & 3);
  const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
  uint32_t offx[II], offy[JJ];
#pragma unroll
  for (int a = 0; a < II; ++a) {
    const int col = colx + 32 * a;
    offx[a] = lds0 + (8 * h + q) * XROW + (((col >> 3) ^ (q << 2)) << 4) + ((col & 7) << 1);
  }
#pragma unroll
  for (int b = 0; b < JJ; ++b) {
    const int col = coly + 32 * b;
    offy[b] = lds0 + X_BYTES + (8 * h + q) * YROW + (((col >> 3) ^ (q << 2)) << 4) + ((col & 7) << 1);
  }

  f32x16 acc[II][JJ];
#pragma unroll
  for (int a = 0; a < II; ++a)
#pragma unroll
    for (int b = 0; b < JJ; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  bf16x4 xl[II], xh[II], yl[JJ], yh[JJ];
#define SAVIT_TR_READ(dst, addr, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF))
#define SAVIT_PP_REQUEST(slot_off, KS)                                                  \
  do {                                                                                  \
    _Pragma("unroll") for (int a_ = 0; a_ < II; ++a_) {                                 \
      SAVIT_TR_READ(xl[a_], offx[a_] + (slot_off), (KS) * 16 * XROW);                   \
      SAVIT_TR_READ(xh[a_], offx[a_] + (slot_off), (KS) * 16 * XROW + 4 * XROW);        \
    }                                                                                   \
    _Pragma("unroll") for (int b_ = 0; b_ < JJ; ++b_) {                                 \
      SAVIT_TR_READ(yl[b_], offy[b_] + (slot_off), (KS) * 16 * YROW);                   \
      SAVIT_TR_READ(yh[b_], offy[b_] + (slot_off), (KS) * 16 * YROW + 4 * YROW);        \
    }                                                                                   \
  } while (0)
  auto compute = [&]() {  // between two barriers: wait for this wave's fragments, then its 8 MFMAs at raised priority
#pragma unroll
    for (int a = 0; a < II; ++a) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xl[a]), "+v"(xh[a]));
#pragma unroll
    for (int b = 0; b < JJ; ++b) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(yl[b]), "+v"(yh[b]));
    bf16x8 xa[II], yb[JJ];
#pragma unroll
    for (int a = 0; a < II; ++a) xa[a] = __builtin_shufflevector(xl[a], xh[a], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
    for (int b = 0; b < JJ; ++b) yb[b] = __builtin_shufflevector(yl[b], yh[b], 0, 1, 2, 3, 4, 5, 6, 7);
    __builtin_amdgcn_s_setprio(1);  // without it the same kernel is 13 % slower: the partner wave's requests interleave with the MFMA issue
#pragma unroll
    for (int a = 0; a < II; ++a)
#pragma unroll
      for (int b = 0; b < JJ; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[a], yb[b], acc[a][b], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  const int pre = NS < S ? NS : S;
  for (int st = 0; st < pre; ++st) stage(st, st);
  if (pre == S) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G * (S - 1)) : "memory");  // stage 0
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (wi == 1) __builtin_amdgcn_s_barrier();  // row half 1 runs one barrier behind row half 0

  int slot = 0;
  for (int st = 0; st < NS; ++st) {
    const uint32_t cur_off = (uint32_t)(slot * STAGE);
    // ---- L0: fragments of k-step 0
    SAVIT_PP_REQUEST(cur_off, 0);
    __builtin_amdgcn_s_barrier();
    // ---- C0
    compute();
    __builtin_amdgcn_s_barrier();
    // ---- L1: fragments of k-step 1; refill the slot of stage st-1; this wave's share of stage st+1 must have landed
    SAVIT_PP_REQUEST(cur_off, 1);
    if (st >= 1 && st - 1 + S < NS) stage(st - 1 + S, slot == 0 ? S - 1 : slot - 1);
    if (st - 1 + S < NS) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G * (S - 2)) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    // ---- C1
    compute();
    __builtin_amdgcn_s_barrier();
    slot = (slot + 1 == S) ? 0 : slot + 1;
  }
  if (wi == 0) __builtin_amdgcn_s_barrier();  // balance the barrier count
#undef SAVIT_PP_REQUEST
#undef SAVIT_TR_READ

  // dW (+)= tile (the tile IS the whole sum over the tokens): plain accesses, whole 128-B row segments per instruction
  const int jl = lane & 31, hi5 = lane >> 5;
  float ss = 0.f;
#pragma unroll
  for (int a = 0; a < II; ++a)
#pragma unroll
    for (int b = 0; b < JJ; ++b) ss += wgrad_store_tile(p, acc[a][b], i0 + wi * WTI + 32 * a, j0 + wj * WTJ + 32 * b + jl, hi5);
  wgrad_publish_sumsq(p, ss);
}

// ------------------------------------------------------------------------------------------------------------
// The same ping-pong tile on v_mfma_f32_16x16x32_bf16 (round 4).  Under this kernel the chip holds its clock down (1.95 GHz: DESIGN
// 6.3), and on MI355X a bf16 MFMA loop on the 16x16x32 shape holds a higher clock than the same flops on 32x32x16
// (MI355X_MICROARCH.md, DVFS give-back item 7) at the same LDS bytes per flop: one MFMA spans a whole 32-token stage, a fragment is
// 16 columns x 32 tokens (lane = (column, token block of 8): the four 16-lane groups of a transposing read take four token blocks of
// the SAME 16 columns), two ds_read_b64_tr_b16 per fragment as before.  The swizzle of the stage image therefore separates rows
// 8 apart as well: chunk ^= (row & 3) << 2 ^ ((row >> 3) & 1) << 1 (the 8 rows x 32 B a half-wave reads cover all 64 banks once).
// Phases per stage as in wgrad_pp_tile:  L0: request the dY fragments + the first half of the X fragments | C0: II/2 x JJ MFMAs |
// L1: request the second half of the X fragments, refill a ring slot | C1: II/2 x JJ MFMAs (the dY fragments stay in registers).
// Rows of the stage image may be 768 B (BJ = 384: the d = 384 models, whose matrices are multiples of 128 and 384, not of 256):
// the LDS-DMA pieces are cut from the linear image, a piece may span two rows.
template <int BI, int BJ, int WGI, int WGJ, int S>
__device__ __forceinline__ void wgrad_pp16_tile(const WgradParams& p, const int ti, const int tj, char* smem) {
  static_assert(WGI == 2, "two row halves = the two ping-pong groups");
  constexpr int NW = WGI * WGJ;
  constexpr int TS = 32;
  constexpr int WTI = BI / WGI, WTJ = BJ / WGJ;
  constexpr int II = WTI / 16, JJ = WTJ / 16, IH = II / 2;
  static_assert(II % 2 == 0 && WTI % 16 == 0 && WTJ % 16 == 0, "wave tile");
  constexpr int XROW = BI * 2, YROW = BJ * 2;
  constexpr int X_BYTES = TS * XROW, Y_BYTES = TS * YROW, STAGE = X_BYTES + Y_BYTES;
  constexpr int X_INSTR = X_BYTES / 1024 / NW, Y_INSTR = Y_BYTES / 1024 / NW, G = X_INSTR + Y_INSTR;
  static_assert(X_BYTES % (1024 * NW) == 0 && Y_BYTES % (1024 * NW) == 0, "tile/wave mismatch");
  static_assert(XROW % 256 == 0 && YROW % 256 == 0, "rows are whole 256-B swizzle segments");
  static_assert(S >= 3 && G * (S - 1) <= 63, "ring depth / vmcnt immediate");

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wi = wave / WGJ, wj = wave % WGJ;
  const int i0 = ti * BI, j0 = tj * BJ;
  const int NS = (p.M + TS - 1) / TS;
  if (NS <= 0) return;

  size_t xbytes = (size_t)p.M * p.ldx * 2, ybytes = (size_t)p.M * p.lddy * 2;
  if (xbytes > 0xffffffe0ull) xbytes = 0xffffffe0ull;
  if (ybytes > 0xffffffe0ull) ybytes = 0xffffffe0ull;
  const auto srdX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.X), 0, (uint32_t)xbytes, 0x00020000);
  const auto srdY = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.dY), 0, (uint32_t)ybytes, 0x00020000);

  auto swz = [](int r) { return ((r & 3) << 2) ^ (((r >> 3) & 1) << 1); };
  auto stage = [&](int st, int slot) {
    char* sX = smem + slot * STAGE;
    char* sY = sX + X_BYTES;
    const int m0 = st * TS;
    {
      constexpr int LPR = XROW / 16;  // 16-B chunks per row
#pragma unroll
      for (int i = 0; i < X_INSTR; ++i) {
        const int inst = wave * X_INSTR + i;
        const int L = inst * 64 + lane;
        const int r = L / LPR;
        const int c = (L - r * LPR) ^ swz(r);
        const int m = m0 + r;
        uint32_t voff = 0xfffffff0u;
        if (m < p.M) voff = (uint32_t)((size_t)m * p.ldx * 2 + (size_t)(i0 + c * 8) * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srdX, (__attribute__((address_space(3))) void*)(sX + inst * 1024), 16, voff, 0, 0, 0);
      }
    }
    {
      constexpr int LPR = YROW / 16;
#pragma unroll
      for (int i = 0; i < Y_INSTR; ++i) {
        const int inst = wave * Y_INSTR + i;
        const int L = inst * 64 + lane;
        const int r = L / LPR;
        const int c = (L - r * LPR) ^ swz(r);
        const int m = m0 + r;
        uint32_t voff = 0xfffffff0u;
        if (m < p.M) voff = (uint32_t)((size_t)m * p.lddy * 2 + (size_t)(j0 + c * 8) * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srdY, (__attribute__((address_space(3))) void*)(sY + inst * 1024), 16, voff, 0, 0, 0);
      }
    }
  };

  // transposing reads: lane (g = lane >> 4, t = lane & 15) addresses token row 8 g + (t >> 2) (+ 4 for the second read) and the
  // 8-byte piece t & 3 of the fragment's 16 columns; it receives column t, tokens 8 g .. 8 g + 7
  const int g = lane >> 4, t = lane & 15;
  const int row = 8 * g + (t >> 2);
  const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
  uint32_t offx[II], offy[JJ], offx4[II], offy4[JJ];
#pragma unroll
  for (int a = 0; a < II; ++a) {
    const int col = wi * WTI + 16 * a + 4 * (t & 3);
    offx[a] = lds0 + row * XROW + (((col >> 3) ^ swz(row)) << 4) + ((col & 7) << 1);
    offx4[a] = lds0 + (row + 4) * XROW + (((col >> 3) ^ swz(row + 4)) << 4) + ((col & 7) << 1);
  }
#pragma unroll
  for (int b = 0; b < JJ; ++b) {
    const int col = wj * WTJ + 16 * b + 4 * (t & 3);
    offy[b] = lds0 + X_BYTES + row * YROW + (((col >> 3) ^ swz(row)) << 4) + ((col & 7) << 1);
    offy4[b] = lds0 + X_BYTES + (row + 4) * YROW + (((col >> 3) ^ swz(row + 4)) << 4) + ((col & 7) << 1);
  }

  f32x4 acc[II][JJ];
#pragma unroll
  for (int a = 0; a < II; ++a)
#pragma unroll
    for (int b = 0; b < JJ; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  bf16x4 xl[IH], xh[IH], yl[JJ], yh[JJ];
#define SAVIT_TR_READ(dst, addr) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(dst) : "v"(addr))
  auto compute = [&](auto half_c) {  // between two barriers: wait for this wave's fragments, then its IH x JJ MFMAs at raised priority
    constexpr int HALF = decltype(half_c)::value;
#pragma unroll
    for (int a = 0; a < IH; ++a) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xl[a]), "+v"(xh[a]));
#pragma unroll
    for (int b = 0; b < JJ; ++b) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(yl[b]), "+v"(yh[b]));
    bf16x8 xa[IH], yb[JJ];
#pragma unroll
    for (int a = 0; a < IH; ++a) xa[a] = __builtin_shufflevector(xl[a], xh[a], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
    for (int b = 0; b < JJ; ++b) yb[b] = __builtin_shufflevector(yl[b], yh[b], 0, 1, 2, 3, 4, 5, 6, 7);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int a = 0; a < IH; ++a)
#pragma unroll
      for (int b = 0; b < JJ; ++b)
        acc[HALF * IH + a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[a], yb[b], acc[HALF * IH + a][b], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  const int pre = NS < S ? NS : S;
  for (int st = 0; st < pre; ++st) stage(st, st);
  if (pre == S) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G * (S - 1)) : "memory");  // stage 0
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (wi == 1) __builtin_amdgcn_s_barrier();  // row half 1 runs one barrier behind row half 0

  int slot = 0;
  for (int st = 0; st < NS; ++st) {
    const uint32_t cur_off = (uint32_t)(slot * STAGE);
    // ---- L0: the dY fragments and the first half of the X fragments
#pragma unroll
    for (int b = 0; b < JJ; ++b) {
      SAVIT_TR_READ(yl[b], offy[b] + cur_off);
      SAVIT_TR_READ(yh[b], offy4[b] + cur_off);
    }
#pragma unroll
    for (int a = 0; a < IH; ++a) {
      SAVIT_TR_READ(xl[a], offx[a] + cur_off);
      SAVIT_TR_READ(xh[a], offx4[a] + cur_off);
    }
    __builtin_amdgcn_s_barrier();
    // ---- C0
    compute(std::integral_constant<int, 0>{});
    __builtin_amdgcn_s_barrier();
    // ---- L1: the second half of the X fragments; refill the slot of stage st-1; this wave's share of stage st+1 must have landed
#pragma unroll
    for (int a = 0; a < IH; ++a) {
      SAVIT_TR_READ(xl[a], offx[IH + a] + cur_off);
      SAVIT_TR_READ(xh[a], offx4[IH + a] + cur_off);
    }
    if (st >= 1 && st - 1 + S < NS) stage(st - 1 + S, slot == 0 ? S - 1 : slot - 1);
    if (st - 1 + S < NS) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G * (S - 2)) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    // ---- C1
    compute(std::integral_constant<int, 1>{});
    __builtin_amdgcn_s_barrier();
    slot = (slot + 1 == S) ? 0 : slot + 1;
  }
  if (wi == 0) __builtin_amdgcn_s_barrier();  // balance the barrier count
#undef SAVIT_TR_READ

  // dW += tile (the tile IS the whole sum over the tokens).  Accumulator (a, b): lane = (column j = lane & 15, rows 4 (lane >> 4) ..
  // + 3): one register = 16 lanes x 4 B = a 64-B row segment; once per tile over hundreds of stages
  const int jl = lane & 15, i4 = 4 * (lane >> 4);
  float ss16 = 0.f;
#pragma unroll
  for (int a = 0; a < II; ++a) {
    float old[JJ][4];
#pragma unroll
    for (int b = 0; b < JJ; ++b) {
      const int j = j0 + wj * WTJ + 16 * b + jl;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + wi * WTI + 16 * a + i4 + r;
        old[b][r] = (!p.overwrite && i < p.Kin && j < p.Nout) ? p.dW[(size_t)i * p.lddw + j] : 0.f;
      }
    }
#pragma unroll
    for (int b = 0; b < JJ; ++b) {
      const int j = j0 + wj * WTJ + 16 * b + jl;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + wi * WTI + 16 * a + i4 + r;
        if (i < p.Kin && j < p.Nout) {
          const float v = old[b][r] + acc[a][b][r];
          p.dW[(size_t)i * p.lddw + j] = v;
          ss16 += v * v;
        }
      }
    }
  }
  wgrad_publish_sumsq(p, ss16);
}

// ------------------------------------------------------------------------------------------------------------
// Grouped launch: the weight gradients of SEVERAL Dense kernels (the four of an encoder layer, of one or more layers) in ONE grid,
// one workgroup per 256 x 256 (or 128 x 128) output tile, each reducing over ALL tokens.  A single weight gradient has too few
// output tiles for the chip (DeiT-B's W1: 36 tiles of 256 x 256 on 256 CUs), which is why the single-problem launch splits the
// token range 7 ways - and then pays for it: 7 partial slabs of the whole matrix written and read again (65 MB per launch, 1.75 x
// the algorithmic traffic) plus an ordered-sum launch.  Together the four gradients of two DeiT-B layers are 216 tiles: one tile
// per workgroup fills 84 % of the CUs with NO split, no slab, no second launch, the long reduction (788 stages) amortises the
// prologue, and the result is still a fixed-order sum (bitwise reproducible).  The weight gradients have no consumer before the
// optimizer step, so the engine is free to compute them a layer or two after their operands are produced.
constexpr int WGRAD_GROUP_MAX = 64;  // entries per launch (the kernel-argument block stays under 4 KB)
struct WgradGroupParams {
  int n;
  int tile_end[WGRAD_GROUP_MAX];  // tiles of problems 0 .. i (prefix sums)
  struct {
    const bf16_t* X;
    const bf16_t* dY;
    float* dW;
    int M, Kin, Nout, ldx, lddy, lddw;
    int tile_begin;  // first output tile of this entry (an entry may cover a RANGE of a weight's tiles: the rest runs in another launch)
    int overwrite;   // dW = tile instead of dW += tile (savit_wgrad_problem.overwrite)
  } pr[WGRAD_GROUP_MAX];
  float* sumsq;      // nullable: 32 accumulators for the sum of squares of the stored gradients (savit_gemm_bf16_wgrad_grouped_ex)
};
static_assert(sizeof(WgradGroupParams) <= 4096, "kernel-argument block");

// Tile 640: 256 x 384 or 384 x 256, whichever covers the weight with fewer tiles (d = 384 models: Wqkv 384 x 1152 and W1 384 x 1536 take
// 384 x 256 - 5 and 6 tiles -, Wo 384 x 384 and W2 1536 x 384 take 256 x 384 - 2 and 6: 19 tiles per layer, 95 % of a launch inside a
// matrix, 12 MFMAs per 14 transposing reads per wave and k-step instead of 6 per 10 of the 128 x 384 tile).
__host__ __device__ inline bool wgrad_mixed_tall(int Kin, int Nout) {  // true: 384 (Kin) x 256 (Nout)
  const int a = ((Kin + 255) / 256) * ((Nout + 383) / 384), b = ((Kin + 383) / 384) * ((Nout + 255) / 256);
  return b < a;
}

template <int S>
__global__ __launch_bounds__(512, 2) void gemm_wgrad_group_mixed_kernel(const WgradGroupParams g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wid = xcd_remap(blockIdx.x, g.tile_end[g.n - 1]);
  int pi = 0;
  while (wid >= g.tile_end[pi]) ++pi;
  const int t = wid - (pi ? g.tile_end[pi - 1] : 0) + g.pr[pi].tile_begin;
  WgradParams p{};
  p.X = g.pr[pi].X; p.dY = g.pr[pi].dY; p.dW = g.pr[pi].dW;
  p.M = g.pr[pi].M; p.Kin = g.pr[pi].Kin; p.Nout = g.pr[pi].Nout;
  p.ldx = g.pr[pi].ldx; p.lddy = g.pr[pi].lddy; p.lddw = g.pr[pi].lddw;
  p.splits = 1;
  p.tiles_per_split = (p.M + 31) / 32;
  p.rmw = 1;
  p.overwrite = g.pr[pi].overwrite;
  p.sumsq = g.sumsq;
  if (wgrad_mixed_tall(p.Kin, p.Nout)) {
    p.tiles_i = (p.Kin + 383) / 384;
    p.tiles_j = (p.Nout + 255) / 256;
    const int ti = t / p.tiles_j;
    wgrad_pp_tile<384, 256, 2, 4, S>(p, ti, t - ti * p.tiles_j, smem);
  } else {
    p.tiles_i = (p.Kin + 255) / 256;
    p.tiles_j = (p.Nout + 383) / 384;
    const int ti = t / p.tiles_j;
    wgrad_pp_tile<256, 384, 2, 4, S>(p, ti, t - ti * p.tiles_j, smem);
  }
}

template <int BI, int BJ, int WGI, int WGJ, int S, int MF = 32>  // MF: MFMA shape of the tile (16 = 16x16x32 ping-pong, 32 = 32x32x16)
__global__ __launch_bounds__(64 * WGI * WGJ, 2) void gemm_wgrad_group_kernel(const WgradGroupParams g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // XCD-aware: each XCD takes a contiguous run of the tile list (problem-major, then row tile, then column tile), i.e. tiles that
  // stream the same X columns / dY columns of the same problem share them through that XCD's L2
  const int wid = xcd_remap(blockIdx.x, g.tile_end[g.n - 1]);
  int pi = 0;
  while (wid >= g.tile_end[pi]) ++pi;  // uniform: scalar loop over at most WGRAD_GROUP_MAX entries
  const int t = wid - (pi ? g.tile_end[pi - 1] : 0) + g.pr[pi].tile_begin;
  WgradParams p{};
  p.X = g.pr[pi].X; p.dY = g.pr[pi].dY; p.dW = g.pr[pi].dW;
  p.M = g.pr[pi].M; p.Kin = g.pr[pi].Kin; p.Nout = g.pr[pi].Nout;
  p.ldx = g.pr[pi].ldx; p.lddy = g.pr[pi].lddy; p.lddw = g.pr[pi].lddw;
  p.tiles_i = (p.Kin + BI - 1) / BI;
  p.tiles_j = (p.Nout + BJ - 1) / BJ;
  p.splits = 1;
  p.tiles_per_split = (p.M + 31) / 32;
  p.rmw = 1;
  p.overwrite = g.pr[pi].overwrite;
  p.sumsq = g.sumsq;
  const int ti = t / p.tiles_j;
  if constexpr (MF == 16)
    wgrad_pp16_tile<BI, BJ, WGI, WGJ, S>(p, ti, t - ti * p.tiles_j, smem);
  else if constexpr (BI == 256 || BJ == 384)
    wgrad_pp_tile<BI, BJ, WGI, WGJ, S>(p, ti, t - ti * p.tiles_j, smem);
  else
    wgrad_ring_tile<BI, BJ, WGI, WGJ, S, false>(p, 0, ti, t - ti * p.tiles_j, smem);
}

// dW[i, j] += sum over splits of slab[s][i][j], splits added in index order (fixed order: the result is bitwise reproducible).
// HBM / memory-side-cache bound: (splits + 2) * 4 B per element; a lane owns 4 consecutive columns (16-B accesses).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, int splits, long stride, float* __restrict__ dW,
                                                            int Kin, int Nout, int lddw) {
  const int n4 = Nout >> 2;
  const long total = (long)Kin * n4;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int i = (int)(e / n4), j4 = (int)(e - (long)i * n4);
    const float* src = slab + (size_t)i * Nout + 4 * j4;
    float4 acc = nt_load_f4(src);
    for (int sp = 1; sp < splits; ++sp) {
      const float4 v = nt_load_f4(src + (size_t)sp * stride);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    float4* dst = reinterpret_cast<float4*>(dW + (size_t)i * lddw + 4 * j4);
    const float4 old = *dst;
    *dst = make_float4(old.x + acc.x, old.y + acc.y, old.z + acc.z, old.w + acc.w);
  }
}

// Split count of the ring kernels (32-token stages): the caller's hint, or the smallest count whose grid fills whole rounds of
// `slots` resident workgroups (>= 92 %); every split owns at least one stage.
inline int ring_split_count(int M, int Kin, int Nout, int BI, int BJ, int slots, int hint, int* tiles_per_split) {
  const int st_total = (M + 31) / 32;
  int splits = hint;
  if (splits <= 0) {
    const int tiles = ((Kin + BI - 1) / BI) * ((Nout + BJ - 1) / BJ);
    splits = 1;
    float best = 0.f;
    for (int sp = 1; sp <= 24; ++sp) {
      const int wgs = tiles * sp;
      const float eff = (float)wgs / (float)(((wgs + slots - 1) / slots) * slots);
      if (eff > best + 0.02f) { best = eff; splits = sp; }
      if (eff >= 0.92f) { splits = sp; break; }
    }
  }
  if (splits > st_total) splits = st_total;
  if (splits < 1) splits = 1;
  const int tps = (st_total + splits - 1) / splits;
  if (tiles_per_split) *tiles_per_split = tps;
  return tps > 0 ? (st_total + tps - 1) / tps : 1;
}

template <int BI, int BJ, int WGI, int WGJ, int S>
int launch_wgrad_ring(WgradParams p, hipStream_t s, int slots, long ws_bytes) {
  constexpr int TS = 32;
  p.tiles_i = (p.Kin + BI - 1) / BI;
  p.tiles_j = (p.Nout + BJ - 1) / BJ;
  p.splits = ring_split_count(p.M, p.Kin, p.Nout, BI, BJ, slots, p.splits, &p.tiles_per_split);
  // partial slabs instead of atomics when the caller's workspace holds them (a single split needs neither: its tile IS the sum, but
  // dW accumulates, so it goes through the same reduce)
  p.slab_stride = (long)p.Kin * p.Nout;
  const bool use_slab = p.slab != nullptr && ws_bytes >= (long)p.splits * p.slab_stride * 4 && p.lddw % 4 == 0 && p.Nout % 4 == 0 &&
                        ((uintptr_t)p.dW % 16) == 0;
  if (!use_slab) p.slab = nullptr;
  const dim3 grid(p.tiles_i * p.tiles_j * p.splits), block(64 * WGI * WGJ);
  const size_t lds = (size_t)S * TS * (BI + BJ) * 2;
  if (p.patch) {
    auto kfn = gemm_wgrad_ring_kernel<BI, BJ, WGI, WGJ, S, true>;
    SAVIT_LDS_ONCE(kfn);
    hipLaunchKernelGGL(kfn, grid, block, lds, s, p);
  } else {
    auto kfn = gemm_wgrad_ring_kernel<BI, BJ, WGI, WGJ, S, false>;
    SAVIT_LDS_ONCE(kfn);
    hipLaunchKernelGGL(kfn, grid, block, lds, s, p);
  }
  if (p.no_reduce && !use_slab) return SAVIT_EINVAL;  // the caller asked for partials only and gave no (or too small a) workspace
  if (use_slab && !p.no_reduce) {
    const long total4 = (long)p.Kin * (p.Nout / 4);
    long blocks = (total4 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p.slab, p.splits, p.slab_stride, p.dW, p.Kin, p.Nout, p.lddw);
  }
  SAVIT_LAUNCH_RET();
}

template <int BI, int BJ, int WGI, int WGJ>
int launch_wgrad(WgradParams p, hipStream_t s) {
  p.tiles_i = (p.Kin + BI - 1) / BI;
  p.tiles_j = (p.Nout + BJ - 1) / BJ;
  const int kt_total = (p.M + TK - 1) / TK;
  int splits = p.splits;
  if (splits <= 0) {
    // 64 KB of LDS per workgroup -> 2 resident workgroups per CU = 512 slots.  Pick the smallest split count whose
    // grid fills whole rounds of slots (>= 92 %): a 576-workgroup grid on 512 slots idles the chip for 44 % of its run.
    const int tiles = p.tiles_i * p.tiles_j;
    const int slots = 512;
    splits = 1;
    float best = 0.f;
    for (int sp = 1; sp <= 16; ++sp) {
      const int wgs = tiles * sp;
      const float eff = (float)wgs / (float)(((wgs + slots - 1) / slots) * slots);
      if (eff > best + 0.02f) { best = eff; splits = sp; }
      if (eff >= 0.92f) { splits = sp; break; }
    }
  }
  if (splits > kt_total) splits = kt_total;
  p.tiles_per_split = (kt_total + splits - 1) / splits;
  p.splits = (kt_total + p.tiles_per_split - 1) / p.tiles_per_split;
  const dim3 grid(p.tiles_i * p.tiles_j, p.splits), block(64 * WGI * WGJ);
  const size_t lds = 2 * TK * (BI + BJ) * 2;
  if (p.patch) {
    auto kfn = gemm_wgrad_kernel<BI, BJ, WGI, WGJ, true>;
    SAVIT_LDS_ONCE(kfn);
    hipLaunchKernelGGL(kfn, grid, block, lds, s, p);
  } else {
    auto kfn = gemm_wgrad_kernel<BI, BJ, WGI, WGJ, false>;
    SAVIT_LDS_ONCE(kfn);
    hipLaunchKernelGGL(kfn, grid, block, lds, s, p);
  }
  SAVIT_LAUNCH_RET();
}

}  // namespace

extern "C" int savit_gemm_wgrad_auto_variant(int Kin, int Nout, int patch) {
  // measured on MI355X (tools/bench_wgrad.py, tools/bench_wgrad_small.py, cold caches): big weights run best on 256x256 tiles with
  // one 8-wave workgroup per CU (fewest L2->LDS bytes per flop and fewest partial-slab bytes: 4*Kin*Nout*splits); so do the MLP
  // weights of the d = 384 models (384 x 1536: 113 -> 98 us although the second 256-row tile is half empty); small or ragged ones
  // on 128x128 tiles, also sized for ONE workgroup per CU so the split count stays low
  if (patch) return 1;
  const long sz = (long)Kin * Nout;
  if (Kin % 256 == 0 && Nout % 256 == 0 && sz >= (1 << 20)) return 3;
  if (Kin % 128 == 0 && Nout % 128 == 0 && sz >= (1 << 19) && (Kin >= 1024 || Nout >= 1024)) return 3;
  return 1;
}

static int wgrad_variant(int Kin, int Nout, int patch) {
  // SAVIT_WGRAD_VARIANT (SAVIT_EXPERIMENTS builds only): 0 = auto, 1..4 = ring kernels, 9 = the 2-stage kernel
  static const int variant = SAVIT_EXP_ENV_INT("SAVIT_WGRAD_VARIANT", 0);
  return variant ? variant : savit_gemm_wgrad_auto_variant(Kin, Nout, patch);
}

static bool ring_geometry(int v, int* bi, int* bj, int* slots) {
  switch (v) {
    case 1: *bi = 128; *bj = 128; *slots = 256; return true;
    case 2: *bi = 256; *bj = 128; *slots = 512; return true;
    case 3: *bi = 256; *bj = 256; *slots = 256; return true;
    case 4: *bi = 128; *bj = 128; *slots = 768; return true;
    case 5: *bi = 128; *bj = 256; *slots = 512; return true;
    default: return false;
  }
}

extern "C" long savit_gemm_wgrad_workspace_bytes(int M, int Kin, int Nout, int splits, int patch) {
  if (M <= 0 || Kin <= 0 || Nout <= 0) return 0;
  int bi, bj, slots;
  if (!ring_geometry(wgrad_variant(Kin, Nout, patch), &bi, &bj, &slots)) return 0;  // the 2-stage kernel adds with atomics
  return (long)ring_split_count(M, Kin, Nout, bi, bj, slots, splits, nullptr) * Kin * Nout * 4;
}

static int wgrad_dispatch(const void* X, const void* dY, float* dW, int M, int Kin, int Nout, int ldx, int lddy, int lddw, int splits, int patch,
                          int img_size, int tokens, int token_offset, void* workspace, long workspace_bytes, int no_reduce, void* stream);

extern "C" int savit_gemm_bf16_wgrad(const void* X, const void* dY, float* dW, int M, int Kin, int Nout, int ldx, int lddy,
                                     int lddw, int splits, int patch, int img_size, int tokens, int token_offset, void* stream) {
  return wgrad_dispatch(X, dY, dW, M, Kin, Nout, ldx, lddy, lddw, splits, patch, img_size, tokens, token_offset, nullptr, 0, 0, stream);
}

extern "C" int savit_gemm_bf16_wgrad_ws(const void* X, const void* dY, float* dW, int M, int Kin, int Nout, int ldx, int lddy, int lddw,
                                        int splits, int patch, int img_size, int tokens, int token_offset, void* workspace,
                                        long workspace_bytes, void* stream) {
  SAVIT_CHECK_ARG(workspace == nullptr || (((uintptr_t)workspace % 16) == 0 && workspace_bytes >= 0));
  return wgrad_dispatch(X, dY, dW, M, Kin, Nout, ldx, lddy, lddw, splits, patch, img_size, tokens, token_offset, workspace, workspace_bytes,
                        0, stream);
}

extern "C" int savit_gemm_bf16_wgrad_partial(const void* X, const void* dY, int M, int Kin, int Nout, int ldx, int lddy, int splits, int patch,
                                             int img_size, int tokens, int token_offset, void* workspace, long workspace_bytes, void* stream) {
  SAVIT_CHECK_ARG(workspace != nullptr && ((uintptr_t)workspace % 16) == 0 && Nout % 4 == 0);
  // dW is not touched by the partial launch; the dispatcher only checks its alignment and pitch
  return wgrad_dispatch(X, dY, (float*)workspace, M, Kin, Nout, ldx, lddy, Nout, splits, patch, img_size, tokens, token_offset, workspace,
                        workspace_bytes, 1, stream);
}

extern "C" int savit_gemm_wgrad_reduce(const void* workspace, int splits, int Kin, int Nout, float* dW, int lddw, void* stream) {
  SAVIT_CHECK_ARG(workspace && dW && splits >= 1 && Kin > 0 && Nout > 0 && Nout % 4 == 0 && lddw >= Nout && lddw % 4 == 0);
  SAVIT_CHECK_ARG(((uintptr_t)workspace % 16) == 0 && ((uintptr_t)dW % 16) == 0);
  const long total4 = (long)Kin * (Nout / 4);
  long blocks = (total4 + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, splits, (long)Kin * Nout, dW,
                     Kin, Nout, lddw);
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_gemm_wgrad_split_count(int M, int Kin, int Nout, int splits, int patch) {
  if (M <= 0 || Kin <= 0 || Nout <= 0) return 0;
  int bi, bj, slots;
  if (!ring_geometry(wgrad_variant(Kin, Nout, patch), &bi, &bj, &slots)) return 0;
  return ring_split_count(M, Kin, Nout, bi, bj, slots, splits, nullptr);
}

static int wgrad_dispatch(const void* X, const void* dY, float* dW, int M, int Kin, int Nout, int ldx, int lddy, int lddw, int splits, int patch,
                          int img_size, int tokens, int token_offset, void* workspace, long workspace_bytes, int no_reduce, void* stream) {
  SAVIT_CHECK_ARG(X && dY && dW && M >= 0 && Kin > 0 && Nout > 0 && lddw >= Nout);
  SAVIT_CHECK_ARG(Kin % 8 == 0 && Nout % 8 == 0 && lddy % 8 == 0 && lddy >= Nout);
  SAVIT_CHECK_ARG(((uintptr_t)X % 16) == 0 && ((uintptr_t)dY % 16) == 0);
  WgradParams p{};
  p.X = (const bf16_t*)X; p.dY = (const bf16_t*)dY; p.dW = dW;
  p.M = M; p.Kin = Kin; p.Nout = Nout; p.ldx = ldx; p.lddy = lddy; p.lddw = lddw; p.splits = splits;
  p.patch = patch; p.img_size = img_size; p.tokens = tokens; p.token_offset = token_offset;
  p.abl = SAVIT_EXP_ENV_INT("SAVIT_WGRAD_ABL", 0);
  p.slab = (float*)workspace;
  p.no_reduce = no_reduce;
  if (patch) {
    SAVIT_CHECK_ARG(patch % 8 == 0 && img_size % patch == 0 && Kin == patch * patch * 3 && tokens > 0 && token_offset >= 0);
    p.grid_side = img_size / patch;
    p.chunks_per_prow = patch * 3 / 8;
    SAVIT_CHECK_ARG(M % (p.grid_side * p.grid_side) == 0 && token_offset + p.grid_side * p.grid_side <= tokens);
  } else {
    SAVIT_CHECK_ARG(ldx % 8 == 0 && ldx >= Kin);
  }
  if (M == 0) return SAVIT_OK;
  const int v = wgrad_variant(Kin, Nout, patch);
  switch (v) {
    case 1: return launch_wgrad_ring<128, 128, 2, 2, 4>(p, (hipStream_t)stream, 256, workspace_bytes);
    case 2: return launch_wgrad_ring<256, 128, 2, 2, 3>(p, (hipStream_t)stream, 512, workspace_bytes);
    case 3: return launch_wgrad_ring<256, 256, 2, 4, 4>(p, (hipStream_t)stream, 256, workspace_bytes);
    case 4: return launch_wgrad_ring<128, 128, 2, 2, 3>(p, (hipStream_t)stream, 768, workspace_bytes);
    case 5: return launch_wgrad_ring<128, 256, 2, 2, 3>(p, (hipStream_t)stream, 512, workspace_bytes);
    default:
      if (no_reduce) return SAVIT_EINVAL;
      p.slab = nullptr;
      return launch_wgrad<128, 128, 2, 2>(p, (hipStream_t)stream);
  }
}

// ---- grouped weight gradients (see gemm_wgrad_group_kernel)
// tile codes of the grouped launch: 256 = 256 x 256, 128 = 128 x 128, 384 = 128 (Kin) x 384 (Nout) - the d = 384 models, whose matrices
// (384 x 1152, 384 x 384, 384 x 1536, 1536 x 384) it covers exactly (256 x 256 tiles: 71 % of a launch inside a matrix)
static bool group_tile_shape(int tile, int* bi, int* bj) {
  switch (tile) {
    case 128: *bi = 128; *bj = 128; return true;
    case 256: *bi = 256; *bj = 256; return true;
    case 384: *bi = 128; *bj = 384; return true;
    case 640: *bi = 256; *bj = 384; return true;  // or 384 x 256 per weight: wgrad_mixed_tall
    default: return false;
  }
}

extern "C" int savit_gemm_wgrad_group_tiles(int Kin, int Nout, int tile) {
  int bi, bj;
  if (Kin <= 0 || Nout <= 0 || !group_tile_shape(tile, &bi, &bj)) return 0;
  if (tile == 640 && wgrad_mixed_tall(Kin, Nout)) return ((Kin + 383) / 384) * ((Nout + 255) / 256);
  return ((Kin + bi - 1) / bi) * ((Nout + bj - 1) / bj);
}

extern "C" int savit_gemm_bf16_wgrad_grouped(const savit_wgrad_problem* problems, int count, int tile, void* stream) {
  return savit_gemm_bf16_wgrad_grouped_ex(problems, count, tile, nullptr, stream);
}

extern "C" int savit_gemm_bf16_wgrad_grouped_ex(const savit_wgrad_problem* problems, int count, int tile, float* sumsq32, void* stream) {
  int tile_bi, tile_bj;
  SAVIT_CHECK_ARG(problems != nullptr && count >= 1 && count <= WGRAD_GROUP_MAX && group_tile_shape(tile, &tile_bi, &tile_bj));
  WgradGroupParams g{};
  int tiles = 0, n = 0;
  for (int i = 0; i < count; ++i) {
    const savit_wgrad_problem& q = problems[i];
    SAVIT_CHECK_ARG(q.X && q.dY && q.dW && q.M >= 0 && q.Kin > 0 && q.Nout > 0 && q.lddw >= q.Nout);
    SAVIT_CHECK_ARG(q.Kin % 8 == 0 && q.Nout % 8 == 0 && q.ldx % 8 == 0 && q.ldx >= q.Kin && q.lddy % 8 == 0 && q.lddy >= q.Nout);
    SAVIT_CHECK_ARG(((uintptr_t)q.X % 16) == 0 && ((uintptr_t)q.dY % 16) == 0 && ((uintptr_t)q.dW % 4) == 0);
    SAVIT_CHECK_ARG((size_t)q.M * q.ldx * 2 <= 0xffffffe0ull && (size_t)q.M * q.lddy * 2 <= 0xffffffe0ull);  // 32-bit buffer offsets
    SAVIT_CHECK_ARG(q.overwrite == 0 || q.overwrite == 1);
    if (q.M == 0 && !q.overwrite) continue;  // nothing to add (an overwriting entry still stores its zeros)
    const int all = savit_gemm_wgrad_group_tiles(q.Kin, q.Nout, tile);
    const int cnt = q.tile_count > 0 ? q.tile_count : all - q.tile_begin;
    SAVIT_CHECK_ARG(q.tile_begin >= 0 && cnt >= 1 && q.tile_begin + cnt <= all);
    tiles += cnt;
    g.tile_end[n] = tiles;
    g.pr[n].tile_begin = q.tile_begin;
    g.pr[n].X = (const bf16_t*)q.X; g.pr[n].dY = (const bf16_t*)q.dY; g.pr[n].dW = q.dW;
    g.pr[n].M = q.M; g.pr[n].Kin = q.Kin; g.pr[n].Nout = q.Nout; g.pr[n].ldx = q.ldx; g.pr[n].lddy = q.lddy; g.pr[n].lddw = q.lddw;
    g.pr[n].overwrite = q.overwrite;
    ++n;
  }
  if (n == 0) return SAVIT_OK;
  g.n = n;
  g.sumsq = sumsq32;
  for (int i = n; i < WGRAD_GROUP_MAX; ++i) g.tile_end[i] = tiles;
  if (tile == 256) {
#ifndef WGRAD_GROUP_S
#define WGRAD_GROUP_S 3  // ring slots of the 256 x 256 grouped tile (32 KB each).  3, not 4: the stages 32 workgroups of an XCD keep in flight
                         // fill its 4 MB L2 (4 x 32 KB x 32 = 4 MB), and panels shared between workgroups are evicted before their second
                         // reader arrives: FETCH_SIZE 1.96 -> 1.71 GB per 216-tile launch, 0.787 -> 0.764 ms per launch in the step
#endif
#ifndef WGRAD_GROUP_MF
#define WGRAD_GROUP_MF 32  // MFMA shape of the grouped tiles (16: the 16x16x32 form, measured 11 % slower - A/B builds: tools/build_variant.sh)
#endif
    auto kfn = gemm_wgrad_group_kernel<256, 256, 2, 4, WGRAD_GROUP_S, WGRAD_GROUP_MF>;
    SAVIT_LDS_ONCE(kfn);
    hipLaunchKernelGGL(kfn, dim3(tiles), dim3(512), (size_t)WGRAD_GROUP_S * 32 * (256 + 256) * 2, (hipStream_t)stream, g);
  } else if (tile == 640) {
    auto kfn = gemm_wgrad_group_mixed_kernel<3>;
    SAVIT_LDS_ONCE(kfn);
    hipLaunchKernelGGL(kfn, dim3(tiles), dim3(512), (size_t)3 * 32 * (256 + 384) * 2, (hipStream_t)stream, g);
  } else if (tile == 384) {
    auto kfn = gemm_wgrad_group_kernel<128, 384, 2, 4, 3, WGRAD_GROUP_MF>;
    SAVIT_LDS_ONCE(kfn);
    hipLaunchKernelGGL(kfn, dim3(tiles), dim3(512), (size_t)3 * 32 * (128 + 384) * 2, (hipStream_t)stream, g);
  } else {
    auto kfn = gemm_wgrad_group_kernel<128, 128, 2, 2, 4>;
    SAVIT_LDS_ONCE(kfn);
    hipLaunchKernelGGL(kfn, dim3(tiles), dim3(256), (size_t)4 * 32 * (128 + 128) * 2, (hipStream_t)stream, g);
  }
  SAVIT_LAUNCH_RET();
}
