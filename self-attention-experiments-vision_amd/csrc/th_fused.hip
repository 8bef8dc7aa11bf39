// Fused talking-heads attention (CaiT self-attention layers): the score tensors S and P' never touch HBM in forward, and in
// backward only dS and P' do (once each, for the dK / dV products that contract over the queries of a whole image).
//
// Replaces, per image, attention.py:44-52 + talking_heads.py:13 of /root/reference/models/layers/attentions/:
//   S_h = Q_h K_h^T (q pre-scaled);  S'_i = sum_h T1[h][i] S_h;  P_i = softmax_k(S'_i);  P'_i = sum_h T2[h][i] P_h;  O_i = P'_i V_i
// Same roundings as the materialising kernels of attention.hip (S, P', dP', dS pass through bf16; mixes and softmax in fp32).
//
// gfx950 design.  The head mix couples all H heads of one (query, key) pair, so the unit of work is (image, tile of queries, ALL
// heads): H x QT x 224 scores live in LDS as bf16 (464-byte rows: 16 lanes reading 16 rows with ds_read_b128 hit 16 distinct
// bank quads), 118 KB for H = 8, QT = 32.
//   forward (QT = 32, one workgroup of 8 waves per tile):
//     1. scores: wave <-> (head, 16-query block).  K_h and Q_h fragments come straight from the packed QKV rows (the contraction
//        index e is contiguous in both: no LDS staging), v_mfma_f32_16x16x32_bf16 with keys on the register index, so a lane
//        holds 4 consecutive keys of one query = one 8-byte LDS store;
//     2. rows: wave <-> query row, lane <-> 4 consecutive keys of EVERY head (th_rows.h): both 8x8 mixes and the softmax in
//        registers, T1 / T2 from scalar registers; P' overwrites S in place;
//     3. P'V: wave <-> (head, 16-query block).  The V operand needs the key index contiguous per lane: V^T [B,H,hd,224] is
//        written once per layer by th_vt_kernel (19 MB) and read like K; P' fragments are ds_read_b128 rows of the score buffer.
//   backward rows (QT = 16: S and dP' buffers side by side): recompute S, dP' = dO V^T (same fragment scheme, e contiguous),
//     the row backward of attention.hip on LDS operands (dS over S, P' over dP'), dT1 / dT2 partials -> slab -> finalize;
//     dS and P' leave as whole 16-byte row chunks for savit_th_attention_bwd_products (dQ, dK, dV: MFMA passes over an image).
// Limits: H <= 8 (even), N <= 208, head_dim 48 or 64: every CaiT XXS / XS / S size at 224^2; others use the materialising path.
#include "common.h"
#include "savit.h"
#include "th_rows.h"

namespace {

constexpr int THF_PIT = 232;   // bf16 elements per (head, query) row of an LDS score buffer (464 B)
constexpr int THF_KEYS = 224;  // key columns the P'V product walks (7 k-steps of 32); columns >= N hold zeros
constexpr int THF_KT = 13;     // 16-key tiles a score buffer is filled with: N <= 208
constexpr int THF_KW = 16 * THF_KT;

struct ThFusedParams {
  const bf16_t* qkv;   // [B*N, ld]
  const bf16_t* d_o;   // backward: [B*N, d]
  bf16_t* vt;          // V^T [B, H, hd, THF_KEYS]
  bf16_t* o;           // forward out [B*N, d]
  bf16_t* dsbuf;       // backward out: dS  [B, H, N, Np]
  bf16_t* pbuf;        // backward out: P'  [B, H, N, Np]
  const float* T1;
  const float* T2;
  float* slab;         // backward: [workgroups][2*H*H]
  int B, N, H, ld, d, hd, Np, qtiles;
  int debug;  // SAVIT_EXPERIMENTS builds only (SAVIT_THF_DEBUG): bit mask of phases to skip: 1 = rows, 2 = scores, 4 = P'V (forward) / row copy-out (backward)
};

#ifdef SAVIT_EXPERIMENTS
#define THF_SKIP(p, bit) (((p).debug & (bit)) != 0)
#else
#define THF_SKIP(p, bit) false
#endif

__device__ __forceinline__ uint2 pack4(const f32x4& a) { return make_uint2(pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3])); }

// ---- V^T: one workgroup per (image, head); V_h [N][hd] through LDS (odd dword pitch: conflict-free column reads)
__global__ __launch_bounds__(256) void th_vt_kernel(const ThFusedParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint32_t* img = reinterpret_cast<uint32_t*>(smem);
  const int hd = p.hd, pitch = hd / 2 + 1;  // dwords per key row
  const int b = blockIdx.x / p.H, h = blockIdx.x - b * p.H;
  const int cpr = hd / 8;  // 16-byte chunks per row
  for (int i = threadIdx.x; i < THF_KEYS * cpr; i += blockDim.x) {
    const int key = i / cpr, c = i - key * cpr;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (key < p.N) v = *reinterpret_cast<const uint4*>(p.qkv + (size_t)((long)b * p.N + key) * p.ld + 2 * p.d + h * hd + 8 * c);
    uint32_t* dst = img + key * pitch + 4 * c;
    dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
  }
  __syncthreads();
  const bf16_t* img16 = reinterpret_cast<const bf16_t*>(smem);
  bf16_t* out = p.vt + (size_t)(b * p.H + h) * hd * THF_KEYS;
  for (int i = threadIdx.x; i < hd * (THF_KEYS / 8); i += blockDim.x) {
    const int kc = i / hd, e = i - kc * hd;  // e fastest: lanes read one LDS row, 2 bytes apart
    uint32_t w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t lo = img16[(size_t)(8 * kc + 2 * j) * (2 * pitch) + e];
      const uint32_t hi = img16[(size_t)(8 * kc + 2 * j + 1) * (2 * pitch) + e];
      w[j] = lo | (hi << 16);
    }
    *reinterpret_cast<uint4*>(out + (size_t)e * THF_KEYS + 8 * kc) = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

// ---- scores of one (head, 16-query block) item: out[q][key] = sum_e X[q][e] Y[key][e] for keys 0 .. 16*THF_KT-1, X / Y rows of
// hd contiguous elements.  Lane (fr, kg): query fr, keys 4*kg .. 4*kg+3 of every 16-key tile.  Operands are buffer loads (rows past
// the image's N and the e >= hd half of the second k-step come back as zeros from the range check: no branches), requested seven
// tiles at a time - with predicated global loads hipcc waited for every tile's pair before its MFMAs: 26 exposed L2 round trips
// per item made the first version of these kernels 4x slower than the materialising path.
constexpr uint32_t THF_OOB = 0x7ffffff0u;
__device__ __forceinline__ bf16x8 buf_frag(__amdgpu_buffer_rsrc_t srd, uint32_t byte_off) {
  union { u32x4 u; bf16x8 v; } r;
  r.u = __builtin_amdgcn_raw_buffer_load_b128(srd, byte_off, 0, 0);
  return r.v;
}
// xoff: byte offset of X[q][8*kg] in srdX (THF_OOB for a query past N); yoff0: byte offset of Y[fr][8*kg] in srdY, ypitch: bytes
// between 16-row tiles of Y
template <int HDV>
__device__ __forceinline__ void th_score_item(__amdgpu_buffer_rsrc_t srdX, uint32_t xoff, __amdgpu_buffer_rsrc_t srdY, uint32_t yoff0,
                                              uint32_t ypitch, bf16_t* out_row, int kg) {
  const bool hi_ok = 32 + 8 * kg < HDV;
  const bf16x8 x0 = buf_frag(srdX, xoff);
  const bf16x8 x1 = buf_frag(srdX, hi_ok ? xoff + 64 : THF_OOB);
  constexpr int TB = 7;
#pragma unroll
  for (int tb = 0; tb < THF_KT; tb += TB) {
    bf16x8 y0[TB], y1[TB];
#pragma unroll
    for (int j = 0; j < TB; ++j) {
      if (tb + j < THF_KT) {
        const uint32_t off = yoff0 + (uint32_t)(tb + j) * ypitch;
        y0[j] = buf_frag(srdY, off);
        y1[j] = buf_frag(srdY, hi_ok ? off + 64 : THF_OOB);
      }
    }
#pragma unroll
    for (int j = 0; j < TB; ++j) {
      if (tb + j < THF_KT) {
        // Two independent accumulators added on the VALU, not acc = mfma(y1, x1, mfma(y0, x0, 0)): hipcc gave the chained pair
        // different result registers (vdst != srcC) with three scalar instructions between them and no wait states, and the
        // second MFMA then read a stale srcC now and then (H < 8 instances of the backward kernel: S off at scattered keys,
        // run-to-run different, while dP' - other registers, other spacing - was exact).  MFMA -> VALU dependencies get their s_nops.
        const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4 acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y0[j], x0, zero, 0, 0, 0);
        const f32x4 acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y1[j], x1, zero, 0, 0, 0);
        const f32x4 acc = acc0 + acc1;
        *reinterpret_cast<uint2*>(out_row + 16 * (tb + j) + 4 * kg) = pack4(acc);
      }
    }
  }
}

// buffer over the rows of one image (N rows of `ld` elements): offsets past the last row read zeros
__device__ __forceinline__ __amdgpu_buffer_rsrc_t image_rsrc(const bf16_t* base, size_t row_base, int ld, int N) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(base + row_base * ld), 0, (uint32_t)((size_t)N * ld * 2), 0x00020000);
}

template <int H, int HDV>
__global__ __launch_bounds__(512) void th_fused_fwd_kernel(const ThFusedParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int QT = 32, HS = QT * THF_PIT;
  bf16_t* SB = reinterpret_cast<bf16_t*>(smem);  // [H][QT][THF_PIT]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwv = blockDim.x >> 6;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);  // the query tiles of an image run on one XCD: K / V^T re-reads are L2 hits
  const int b = wg / p.qtiles, q0 = (wg - b * p.qtiles) * QT;
  const size_t row_base = (size_t)b * p.N;
  const int fr = lane & 15, kg = lane >> 4;

  // ---- 1. scores
  const auto srdQ = image_rsrc(p.qkv, row_base, p.ld, p.N);
  for (int item = wave; item < (THF_SKIP(p, 2) ? 0 : 2 * H); item += nwv) {
    const int h = item >> 1, qb = item & 1;
    const int q = q0 + 16 * qb + fr;
    const uint32_t xoff = q < p.N ? (uint32_t)(((size_t)q * p.ld + h * HDV + 8 * kg) * 2) : THF_OOB;
    const uint32_t yoff0 = (uint32_t)(((size_t)fr * p.ld + p.d + h * HDV + 8 * kg) * 2);
    th_score_item<HDV>(srdQ, xoff, srdQ, yoff0, (uint32_t)(16 * p.ld * 2), SB + h * HS + (16 * qb + fr) * THF_PIT, kg);
  }
  __syncthreads();

  // ---- 2. rows: mix, softmax, mix; P' over S
  for (int r = wave; r < QT; r += nwv) {
    if (q0 + r >= p.N || THF_SKIP(p, 1)) continue;
    bf16_t* row = SB + r * THF_PIT + 4 * lane;
    float s[H][TH_KPL], pr[H][TH_KPL];
#pragma unroll
    for (int h = 0; h < H; ++h) {
      uint2 v = make_uint2(0u, 0u);
      if (4 * lane < THF_KW) v = *reinterpret_cast<const uint2*>(row + h * HS);
      s[h][0] = __uint_as_float(v.x << 16); s[h][1] = __uint_as_float(v.x & 0xffff0000u);
      s[h][2] = __uint_as_float(v.y << 16); s[h][3] = __uint_as_float(v.y & 0xffff0000u);
    }
    th_row_forward<H>(s, p.T1, p.N, lane, pr);
#pragma unroll
    for (int i = 0; i < H; ++i) {
      float o4[TH_KPL];
#pragma unroll
      for (int k = 0; k < TH_KPL; ++k) {
        float a = 0.f;
#pragma unroll
        for (int h = 0; h < H; ++h) a += p.T2[h * H + i] * pr[h][k];
        o4[k] = (4 * lane + k < p.N) ? a : 0.f;
      }
      if (4 * lane < THF_KEYS) *reinterpret_cast<uint2*>(row + i * HS) = make_uint2(pack_bf16x2(o4[0], o4[1]), pack_bf16x2(o4[2], o4[3]));
    }
  }
  __syncthreads();

  // ---- 3. O_g^T[e][q] = sum_key V_g^T[e][key] P'_g[q][key]
  constexpr int NEB = HDV / 16;
  for (int item = wave; item < (THF_SKIP(p, 4) ? 0 : 2 * H); item += nwv) {
    const int g = item >> 1, qb = item & 1;
    const int q = q0 + 16 * qb + fr;
    const bf16_t* prow = SB + g * HS + (16 * qb + fr) * THF_PIT + 8 * kg;
    const bf16_t* vrow = p.vt + ((size_t)(b * H + g) * HDV + fr) * THF_KEYS + 8 * kg;
    f32x4 acc[NEB];
#pragma unroll
    for (int eb = 0; eb < NEB; ++eb) acc[eb] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int KS = THF_KEYS / 32, KB = 4;  // V^T fragments are requested four k-steps at a time
#pragma unroll
    for (int k0 = 0; k0 < KS; k0 += KB) {
      bf16x8 vf[KB][NEB];
#pragma unroll
      for (int j = 0; j < KB; ++j)
#pragma unroll
        for (int eb = 0; eb < NEB; ++eb)
          if (k0 + j < KS) vf[j][eb] = *reinterpret_cast<const bf16x8*>(vrow + (size_t)16 * eb * THF_KEYS + 32 * (k0 + j));
#pragma unroll
      for (int j = 0; j < KB; ++j) {
        if (k0 + j < KS) {
          const bf16x8 pf = *reinterpret_cast<const bf16x8*>(prow + 32 * (k0 + j));
#pragma unroll
          for (int eb = 0; eb < NEB; ++eb) acc[eb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[j][eb], pf, acc[eb], 0, 0, 0);
        }
      }
    }
    if (q < p.N) {
      bf16_t* orow = p.o + (row_base + q) * p.d + g * HDV + 4 * kg;
#pragma unroll
      for (int eb = 0; eb < NEB; ++eb) *reinterpret_cast<uint2*>(orow + 16 * eb) = pack4(acc[eb]);
    }
  }
}

// ---- backward rows.  Register plan as th_softmax_bwd_kernel (attention.hip): S and dP' stay packed, P / dP fp32, the 64 dT
// partials of a row are reduce-scattered at once.
template <int H, int HDV>
__global__ __launch_bounds__(512) void th_fused_bwd_rows_kernel(const ThFusedParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int QT = 16, HS = QT * THF_PIT;
  bf16_t* SB = reinterpret_cast<bf16_t*>(smem);  // S, then dS   [H][QT][THF_PIT]
  bf16_t* DB = SB + H * HS;                      // dP', then P'
  __shared__ float red[8][128];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwv = blockDim.x >> 6;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int b = wg / p.qtiles, q0 = (wg - b * p.qtiles) * QT;
  const size_t row_base = (size_t)b * p.N;
  const int fr = lane & 15, kg = lane >> 4;
  const float* __restrict__ T1 = p.T1;
  const float* __restrict__ T2 = p.T2;

  // ---- 1. S = Q K^T and dP' = dO V^T
  const auto srdQ = image_rsrc(p.qkv, row_base, p.ld, p.N);
  const auto srdD = image_rsrc(p.d_o, row_base, p.d, p.N);
  for (int item = wave; item < (THF_SKIP(p, 2) ? 0 : 2 * H); item += nwv) {
    const int h = item >> 1;
    const int q = q0 + fr;
    const uint32_t tile_pitch = (uint32_t)(16 * p.ld * 2);
    if (item & 1) {
      const uint32_t xoff = q < p.N ? (uint32_t)(((size_t)q * p.d + h * HDV + 8 * kg) * 2) : THF_OOB;
      const uint32_t yoff0 = (uint32_t)(((size_t)fr * p.ld + 2 * p.d + h * HDV + 8 * kg) * 2);
      th_score_item<HDV>(srdD, xoff, srdQ, yoff0, tile_pitch, DB + h * HS + fr * THF_PIT, kg);
    } else {
      const uint32_t xoff = q < p.N ? (uint32_t)(((size_t)q * p.ld + h * HDV + 8 * kg) * 2) : THF_OOB;
      const uint32_t yoff0 = (uint32_t)(((size_t)fr * p.ld + p.d + h * HDV + 8 * kg) * 2);
      th_score_item<HDV>(srdQ, xoff, srdQ, yoff0, tile_pitch, SB + h * HS + fr * THF_PIT, kg);
    }
  }
  __syncthreads();

  // ---- 2. rows
  float acc1 = 0.f, acc2 = 0.f;  // lane l accumulates dT1 / dT2 entry (h = l >> 3, i = l & 7)
  for (int r = wave; r < QT; r += nwv) {
    if (q0 + r >= p.N || THF_SKIP(p, 1)) continue;
    bf16_t* srow = SB + r * THF_PIT + 4 * lane;
    bf16_t* drow = DB + r * THF_PIT + 4 * lane;
    const bool in_row = 4 * lane < THF_KW;
    uint32_t sp[H][TH_KPL / 2], dq[H][TH_KPL / 2];
#pragma unroll
    for (int h = 0; h < H; ++h) {
      uint2 sv2 = make_uint2(0u, 0u), dv2 = make_uint2(0u, 0u);
      if (in_row) {
        sv2 = *reinterpret_cast<const uint2*>(srow + h * HS);
        dv2 = *reinterpret_cast<const uint2*>(drow + h * HS);
      }
      sp[h][0] = sv2.x; sp[h][1] = sv2.y;
      dq[h][0] = dv2.x; dq[h][1] = dv2.y;
    }
    // forward recompute: P_i = softmax_k(sum_h T1[h][i] S_h)
    float pr[H][TH_KPL];
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
        pr[i][k] = (4 * lane + k < p.N) ? a : -INFINITY;
      }
    }
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
    // P'_i = sum_h T2[h][i] P_h over dP' (every dP' of this lane is in registers)
#pragma unroll
    for (int i = 0; i < H; ++i) {
      float o4[TH_KPL];
#pragma unroll
      for (int k = 0; k < TH_KPL; ++k) {
        float a = 0.f;
#pragma unroll
        for (int h = 0; h < H; ++h) a += T2[h * H + i] * pr[h][k];
        o4[k] = a;  // P_h = 0 for keys >= N
      }
      if (in_row) *reinterpret_cast<uint2*>(drow + i * HS) = make_uint2(pack_bf16x2(o4[0], o4[1]), pack_bf16x2(o4[2], o4[3]));
    }
    // dT2[h][i] += sum_k P_h dP'_i ; dP_h = sum_i T2[h][i] dP'_i ; delta_h = sum_k P_h dP_h
    float dp[H][TH_KPL], del[H];
    {
      float g[64];
#pragma unroll
      for (int j = 0; j < 64; ++j) g[j] = 0.f;
#pragma unroll
      for (int h = 0; h < H; ++h) del[h] = 0.f;
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
      acc2 += reduce_scatter64(g, lane);
    }
#pragma unroll
    for (int h = 0; h < H; ++h) del[h] = wave_sum(del[h]);
    // dS'_i = P_i (dP_i - delta_i) ; dS_h = sum_i T1[h][i] dS'_i ; dT1[h][i] += sum_k S_h dS'_i
    {
      float g[64];
#pragma unroll
      for (int j = 0; j < 64; ++j) g[j] = 0.f;
      float dsv[H][TH_KPL];
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
          dsv[h][k] = (4 * lane + k < p.N) ? a : 0.f;
        }
      }
      if (in_row) {
#pragma unroll
        for (int h = 0; h < H; ++h)
          *reinterpret_cast<uint2*>(srow + h * HS) = make_uint2(pack_bf16x2(dsv[h][0], dsv[h][1]), pack_bf16x2(dsv[h][2], dsv[h][3]));
      }
      acc1 += reduce_scatter64(g, lane);
    }
  }
  red[wave][lane] = acc1;
  red[wave][64 + lane] = acc2;
  __syncthreads();
  // slab row: [dT1 (H*H) | dT2 (H*H)], entry (h, i) at reduce-scatter lane h*8 + i
  for (int t = threadIdx.x; t < 2 * H * H; t += blockDim.x) {
    const int which = t / (H * H), e = t - which * H * H;
    const int src = which * 64 + (e / H) * 8 + (e % H);
    float a = 0.f;
    for (int w = 0; w < nwv; ++w) a += red[w][src];
    p.slab[(size_t)wg * 2 * H * H + t] = a;
  }

  // ---- 3. dS and P' rows -> HBM in 16-byte chunks
  const int npc = p.Np / 8;
  for (int i = threadIdx.x; i < (THF_SKIP(p, 4) ? 0 : 2 * H * QT * npc); i += blockDim.x) {
    const int c = i % npc, rowi = i / npc;
    const int which = rowi / (H * QT), hr = rowi - which * (H * QT);
    const int h = hr / QT, r = hr - h * QT;
    const int q = q0 + r;
    if (q >= p.N) continue;
    const uint4 v = *reinterpret_cast<const uint4*>((which ? DB : SB) + h * HS + r * THF_PIT + 8 * c);
    *reinterpret_cast<uint4*>((which ? p.pbuf : p.dsbuf) + (((size_t)b * H + h) * p.N + q) * p.Np + 8 * c) = v;
  }
}

}  // namespace

static bool thf_ok(int N, int H, int head_dim) {
  return N > 0 && N <= THF_KW && (H == 2 || H == 4 || H == 6 || H == 8) && (head_dim == 48 || head_dim == 64);
}

// 1 when the fused kernels cover this geometry (else only savit_th_attention_fwd / _bwd apply).
extern "C" int savit_th_fused_supported(int N, int H, int head_dim) { return thf_ok(N, H, head_dim) ? 1 : 0; }

// 1 when a caller with both paths available should take the fused one.  Measured on MI355X at the CaiT-S24 layer (256 images, 8 heads
// of 48, N = 196; tools/thf_bench.py): forward 240 us fused vs 253 us materialising, backward 660 vs 500 us - the 160 KB of LDS cap
// the query tile at 32 (forward) / 16 (backward) rows of all heads, so K / V are re-read from L2 7 / 13 times per image (0.8 / 1.1 GB
// per layer: 72 / 142 us), and the row phase is the same VALU work as the materialising row kernels.  Default: off; the engine's `th_fused=True` (or SAVIT_TH_FUSED=1 read by cait_engine.py)
// selects the fused kernels (they keep no S / P' per layer: 7.5 GB less HBM at CaiT-S24, 256 images).
extern "C" int savit_th_fused_preferred(int N, int H, int head_dim) {
  (void)N; (void)H; (void)head_dim;
  return 0;  // no geometry measured so far runs faster fused; callers that want the fused kernels ask for them (CaiTEngine(th_fused=True))
}

// bytes of the V^T scratch the forward needs (shared by all layers)
extern "C" long savit_th_fused_fwd_workspace_bytes(int B, int N, int H, int head_dim) {
  if (!thf_ok(N, H, head_dim)) return 0;
  return (long)B * H * head_dim * THF_KEYS * 2;
}

// bytes of the dT slab the backward needs
extern "C" long savit_th_fused_bwd_workspace_bytes(int B, int N, int H, int head_dim) {
  if (!thf_ok(N, H, head_dim)) return 0;
  return (long)B * ((N + 15) / 16) * 2 * H * H * (long)sizeof(float);
}

#define THF_DISPATCH(KERNEL, GRID, LDS)                                                                                   \
  {                                                                                                                       \
    const void* kfn = nullptr;                                                                                            \
    switch (H * 100 + head_dim) {                                                                                         \
      case 248: { kfn = (const void*)KERNEL<2, 48>; SAVIT_LDS_ONCE(kfn); } break;                                       \
      case 264: { kfn = (const void*)KERNEL<2, 64>; SAVIT_LDS_ONCE(kfn); } break;                                       \
      case 448: { kfn = (const void*)KERNEL<4, 48>; SAVIT_LDS_ONCE(kfn); } break;                                       \
      case 464: { kfn = (const void*)KERNEL<4, 64>; SAVIT_LDS_ONCE(kfn); } break;                                       \
      case 648: { kfn = (const void*)KERNEL<6, 48>; SAVIT_LDS_ONCE(kfn); } break;                                       \
      case 664: { kfn = (const void*)KERNEL<6, 64>; SAVIT_LDS_ONCE(kfn); } break;                                       \
      case 848: { kfn = (const void*)KERNEL<8, 48>; SAVIT_LDS_ONCE(kfn); } break;                                       \
      case 864: { kfn = (const void*)KERNEL<8, 64>; SAVIT_LDS_ONCE(kfn); } break;                                       \
      default: return SAVIT_EINVAL;                                                                                       \
    }                                                                                                                     \
    void* args_[] = {(void*)&p};                                                                                          \
    hipError_t e_ = hipLaunchKernel(kfn, dim3(GRID), dim3(512), args_, (LDS), (hipStream_t)stream);                                  \
    if (e_ != hipSuccess) return (int)e_;                                                                                 \
  }

static int thf_fill(ThFusedParams& p, const void* qkv, const float* T1, const float* T2, int B, int N, int H, int head_dim, int ld_qkv) {
  if (!(qkv && T1 && T2 && B >= 0 && thf_ok(N, H, head_dim) && ld_qkv >= 3 * H * head_dim && ld_qkv % 8 == 0 && ((uintptr_t)qkv % 16) == 0))
    return SAVIT_EINVAL;
  p.qkv = (const bf16_t*)qkv; p.T1 = T1; p.T2 = T2;
  p.B = B; p.N = N; p.H = H; p.ld = ld_qkv; p.d = H * head_dim; p.hd = head_dim;
  return SAVIT_OK;
}

// Forward: O = talking-heads attention of the packed QKV rows; `workspace` >= savit_th_fused_fwd_workspace_bytes.  Nothing is saved
// for backward (it recomputes S and P from QKV).
extern "C" int savit_th_fused_attention_fwd(const void* qkv, const float* T1, const float* T2, void* o, int B, int N, int H, int head_dim,
                                            int ld_qkv, void* workspace, long workspace_bytes, void* stream) {
  ThFusedParams p{};
  int rc = thf_fill(p, qkv, T1, T2, B, N, H, head_dim, ld_qkv);
  if (rc) return rc;
  SAVIT_CHECK_ARG(o && ((uintptr_t)o % 16) == 0 && workspace && ((uintptr_t)workspace % 16) == 0 &&
                  workspace_bytes >= savit_th_fused_fwd_workspace_bytes(B, N, H, head_dim));
  if (B == 0) return SAVIT_OK;
  p.o = (bf16_t*)o; p.vt = (bf16_t*)workspace;
  p.debug = SAVIT_EXP_ENV_INT("SAVIT_THF_DEBUG", 0);
  p.qtiles = (N + 31) / 32;
  const size_t lds_vt = (size_t)THF_KEYS * (head_dim / 2 + 1) * 4;
  hipLaunchKernelGGL(th_vt_kernel, dim3(B * H), dim3(256), lds_vt, (hipStream_t)stream, p);
  const size_t lds = (size_t)H * 32 * THF_PIT * 2;
  THF_DISPATCH(th_fused_fwd_kernel, B * p.qtiles, lds)
  SAVIT_LAUNCH_RET();
}

// Backward: dqkv (bf16 [B*N, ld_qkv]) and dT1 / dT2 (+=) from QKV, T1, T2 and dO.  p_buf / ds_buf: bf16 [B,H,N,Np] scratch (Np = N
// rounded up to 8) shared by all layers; `workspace` >= savit_th_fused_bwd_workspace_bytes.
extern "C" int savit_th_fused_attention_bwd(const void* qkv, const float* T1, const float* T2, const void* d_o, void* p_buf, void* ds_buf,
                                            void* dqkv, float* dT1, float* dT2, int B, int N, int H, int head_dim, int ld_qkv, int Np,
                                            float dq_scale, void* workspace, long workspace_bytes, void* stream) {
  ThFusedParams p{};
  int rc = thf_fill(p, qkv, T1, T2, B, N, H, head_dim, ld_qkv);
  if (rc) return rc;
  SAVIT_CHECK_ARG(d_o && p_buf && ds_buf && dqkv && dT1 && dT2 && Np >= N && Np % 8 == 0 && Np <= THF_KW);
  SAVIT_CHECK_ARG(((uintptr_t)d_o % 16) == 0 && ((uintptr_t)p_buf % 16) == 0 && ((uintptr_t)ds_buf % 16) == 0);
  SAVIT_CHECK_ARG(workspace && ((uintptr_t)workspace % 16) == 0 && workspace_bytes >= savit_th_fused_bwd_workspace_bytes(B, N, H, head_dim));
  if (B == 0) return SAVIT_OK;
  p.d_o = (const bf16_t*)d_o; p.pbuf = (bf16_t*)p_buf; p.dsbuf = (bf16_t*)ds_buf; p.slab = (float*)workspace; p.Np = Np;
  p.qtiles = (N + 15) / 16;
  const size_t lds = (size_t)2 * H * 16 * THF_PIT * 2;
  p.debug = SAVIT_EXP_ENV_INT("SAVIT_THF_DEBUG", 0);
  const int nblk = B * p.qtiles;
  THF_DISPATCH(th_fused_bwd_rows_kernel, nblk, lds)
  hipLaunchKernelGGL(th_dT_finalize_kernel, dim3((2 * H * H + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, nblk, H * H, dT1,
                     dT2);
  return savit_th_attention_bwd_products(qkv, p_buf, ds_buf, d_o, dqkv, B, N, H, head_dim, ld_qkv, Np, dq_scale, stream);
}
