// bf16 MFMA GEMM  C[M,N] = epi( A[M,K] . Bt[N,K]^T ), fp32 accumulate, fused epilogues.
//
// Stands in for every nn.Dense / nn.DenseGeneral product of the hot path and for their input-gradient
// products (reference call sites listed in include/savit.h).  gfx950 design:
//   * tiles BMxBNx64, 64-lane waves in a WGM x WGN grid, v_mfma_f32_16x16x32_bf16;
//   * operands go HBM -> LDS directly with `buffer_load_dwordx4 ... lds` (LDS-DMA, 16 B/lane, no VGPR
//     staging).  The buffer descriptor's bounds check zero-fills rows >= M (>= N for Bt), so ragged M/N
//     need no host padding;
//   * LDS rows are 128 B (64 bf16); the 16-B chunk index is XOR-swizzled with (row>>1)&7.  Because the
//     LDS-DMA destination is lane-linear, the swizzle is applied to the per-lane SOURCE address and
//     again on the ds_read_b128 side (same involution) -> conflict-free fragment reads;
//   * 2-stage pipeline: tile k+1 is in flight while tile k feeds the MFMAs, one barrier per K-tile;
//   * MFMA operands are swapped (Bt fragment as A-operand) so each lane owns 4 CONSECUTIVE columns of
//     one output row: epilogue loads/stores are 8-16 B vectors;
//   * workgroup ids are remapped so that each XCD (private L2) works on a contiguous run of tiles that
//     share A row-panels.
#include "common.h"
#include "savit.h"
#include <type_traits>

static int device_cus();
extern "C" int savit_gemm_tn_auto_tile_cus(int M, int N, int K, int epilogue, int cu_budget);

namespace {

constexpr int BK = 64;           // bf16 elements per K-tile  (128 B per LDS row)
constexpr int ROW_BYTES = 128;

struct GemmParams {
  savit_gemm_args a;
  int tiles_m, tiles_n;
  int chunks_per_prow;  // PATCH: 16-B chunks per (patch row) = patch*3/8
  int grid_side;        // PATCH: patches per image side
  int desync_phases;    // ping-pong kernel: workgroups start in this many phase groups ...
  int desync_sleep;     // ... each delayed by (group index) x this many s_sleep(127) units (0 = all start together)
  int row_group;        // ping-pong kernel: tiles are visited in groups of this many row panels, column by column inside a group
  int big_tiles, small_tiles, big_rows;  // tail-split pair kernel: tile counts of the two heights, rows covered by the tall tiles
  int pf_start, pf_policy;  // 320 x 256 kernel, RESID / DGELU: first K-tile of the fused operand's cache prefetch (< 0: off), its cache policy bits
};

// Output stores of the coalesced epilogues.  SAVIT_EPI_SC1 (experiment builds): write-through `sc1` stores - nothing of the output stays
// dirty in the XCD's L2, so the release at the end of the kernel has nothing to write back (the dependent-launch boundary grows with the
// bytes a kernel leaves dirty: MI355X_MICROARCH.md, price list row 'boundary').  Measured in the DeiT-B step (profiles/r06_sc1_stores_ab.log,
// A / B / A / B on one box): bf16 outputs 1-2 us faster per launch (proj input gradient 31.1 -> 30.0, GELU' 153 -> 151), the fp32
// residual-stream outputs 4-6 us SLOWER (proj + residual 50 -> 56, fc2 + residual 109 -> 112.7: the next LayerNorm reads them from L2
// otherwise), the step 16.25 -> 16.36 ms: not taken.
__device__ __forceinline__ void epi_store_u4(void* p, uint4 v) {
#ifdef SAVIT_EPI_SC1
  const u32x4 t = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(t) : "memory");
#else
  *reinterpret_cast<uint4*>(p) = v;
#endif
}
__device__ __forceinline__ void epi_store_f4(void* p, float4 v) {
#ifdef SAVIT_EPI_SC1
  const f32x4 t = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(t) : "memory");
#else
  *reinterpret_cast<float4*>(p) = v;
#endif
}

template <int EPI>
__device__ __forceinline__ void epilogue_store(const GemmParams& p, int m, int n, f32x4 acc, float (&csum)[4]) {
  const savit_gemm_args& a = p.a;
  if (m >= a.M || n >= a.N) return;
  float v[4] = {acc[0], acc[1], acc[2], acc[3]};
  if (EPI == SAVIT_EPI_BF16) {
    if (n < a.alpha_cols) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] *= a.alpha;
    }
  }
  if (EPI != SAVIT_EPI_DGELU && a.bias != nullptr) {
    const float4 b = *reinterpret_cast<const float4*>(a.bias + n);
    float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] += a.round_bias_bf16 ? round_bf16(bb[i]) : bb[i];
  }
  if (EPI == SAVIT_EPI_BF16) {
    bf16_t* c = reinterpret_cast<bf16_t*>(a.C) + (size_t)m * a.ldc + n;
    *reinterpret_cast<uint2*>(c) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
  } else if (EPI == SAVIT_EPI_BIAS_GELU) {
    float g[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[i] = round_bf16(v[i]);
      g[i] = gelu_tanh_f(v[i]);
    }
    bf16_t* c = reinterpret_cast<bf16_t*>(a.C) + (size_t)m * a.ldc + n;
    bf16_t* c2 = reinterpret_cast<bf16_t*>(a.C2) + (size_t)m * a.ldc + n;
    *reinterpret_cast<uint2*>(c) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
    *reinterpret_cast<uint2*>(c2) = make_uint2(pack_bf16x2(g[0], g[1]), pack_bf16x2(g[2], g[3]));
  } else if (EPI == SAVIT_EPI_RESID) {
    const float4 r = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a.aux) + (size_t)m * a.ldaux + n);
    float rs = 1.0f;
    if (a.rowscale != nullptr) rs = a.rowscale[m / a.rows_per_sample];
    float cs[4] = {1.f, 1.f, 1.f, 1.f};
    if (a.colscale != nullptr) {
      const float4 c4 = *reinterpret_cast<const float4*>(a.colscale + n);
      cs[0] = c4.x; cs[1] = c4.y; cs[2] = c4.z; cs[3] = c4.w;
    }
    if (a.C2 != nullptr)  // the bf16 branch value, needed by the LayerScale gradient (layerscale.py:23) - as epilogue_lds stores it
      *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(a.C2) + (size_t)m * a.ldc + n) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
    float4 o;
    o.x = r.x + rs * cs[0] * round_bf16(v[0]);
    o.y = r.y + rs * cs[1] * round_bf16(v[1]);
    o.z = r.z + rs * cs[2] * round_bf16(v[2]);
    o.w = r.w + rs * cs[3] * round_bf16(v[3]);
    if (a.round_out_bf16) o = make_float4(round_bf16(o.x), round_bf16(o.y), round_bf16(o.z), round_bf16(o.w));
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.C) + (size_t)m * a.ldc + n) = o;
  } else if (EPI == SAVIT_EPI_DGELU) {
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(a.aux) + (size_t)m * a.ldaux + n);
    const float uu[4] = {__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                         __uint_as_float(u.y & 0xffff0000u)};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[i] = round_bf16(v[i] * gelu_tanh_grad_f(uu[i]));
      csum[i] += v[i];
    }
    bf16_t* c = reinterpret_cast<bf16_t*>(a.C) + (size_t)m * a.ldc + n;
    *reinterpret_cast<uint2*>(c) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
  } else if (EPI == SAVIT_EPI_F32) {
    if (a.round_out_bf16) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = round_bf16(v[i]);
    }
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.C) + (size_t)m * a.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
  } else if (EPI == SAVIT_EPI_PATCH) {
    const int ppi = p.grid_side * p.grid_side;
    const int b = m / ppi, pp = m - b * ppi;
    const int tok = a.token_offset + pp;
    float4 pos = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.aux != nullptr) pos = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a.aux) + (size_t)tok * a.ldaux + n);
    float4 o = make_float4(round_bf16(v[0]) + pos.x, round_bf16(v[1]) + pos.y, round_bf16(v[2]) + pos.z, round_bf16(v[3]) + pos.w);
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.C) + ((size_t)b * a.tokens + tok) * a.ldc + n) = o;
  }
}


// ------------------------------------------------------------------------------------------------------------
// Coalesced epilogue for the bf16-valued epilogues (BF16, BIAS_GELU, RESID, DGELU).
// The MFMA accumulator layout gives a lane 4 consecutive columns of 16 different rows: stored directly, every wave
// store instruction touches 16 rows x 32 B - partial cache lines, measured at 1.8-3.9 TB/s effective on the 155 MB
// outputs of the MLP GEMMs (more than the GELU math).  Instead each wave parks its tile, already rounded to bf16
// exactly where the reference's bf16 graph rounds (after bias / alpha), in its own slice of the now idle staging LDS
// (ds_write_b64, 16-B unit XOR-swizzled by row: conflict-free), reads it back row-contiguous (ds_read_b128) and issues
// 16-B-per-lane loads/stores: one wave instruction = 4 rows x 256 B (or 8 x 128 B) of whole cache lines, for the output
// AND for the fused operands (fp32 residual, bf16 pre-activation).
template <int EPI, int WTM, int WTN, int MI, int NI>
__device__ __forceinline__ void epilogue_lds(const GemmParams& p, f32x4 (&acc)[MI][NI], char* wsm, int row0w, int col0w, int lane,
                                             int slab_row = 0) {
  const savit_gemm_args& a = p.a;
  constexpr int ROWB = WTN * 2;        // bytes per staged row
  constexpr int UPR = WTN / 8;         // 16-B units per row (16 or 8)
  constexpr int RPI = 64 / UPR;        // rows per read instruction
  const int fr = lane & 15, fq = lane >> 4;
  // ---- park: lane holds rows 16i+fr, columns 16j + 4fq .. +3
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int n = col0w + j * 16 + fq * 4;
    float bb[4] = {0.f, 0.f, 0.f, 0.f};
    if (EPI != SAVIT_EPI_DGELU && a.bias != nullptr && n < a.N) {
      const float4 b = *reinterpret_cast<const float4*>(a.bias + n);
      bb[0] = b.x; bb[1] = b.y; bb[2] = b.z; bb[3] = b.w;
      if (a.round_bias_bf16) {
#pragma unroll
        for (int k = 0; k < 4; ++k) bb[k] = round_bf16(bb[k]);
      }
    }
    const float sc = (EPI == SAVIT_EPI_BF16 && n < a.alpha_cols) ? a.alpha : 1.0f;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = i * 16 + fr;
      const int unit = (2 * j + (fq >> 1)) ^ (m & (UPR - 1));
      const f32x4 v = acc[i][j];
      float w0 = v[0], w1 = v[1], w2 = v[2], w3 = v[3];
      if (EPI == SAVIT_EPI_BF16) { w0 *= sc; w1 *= sc; w2 *= sc; w3 *= sc; }
      *reinterpret_cast<uint2*>(wsm + m * ROWB + unit * 16 + (fq & 1) * 8) =
          make_uint2(pack_bf16x2(w0 + bb[0], w1 + bb[1]), pack_bf16x2(w2 + bb[2], w3 + bb[3]));
    }
  }
  // ---- drain: row-contiguous, 8 columns (16 B) per lane
  const int urow = lane / UPR, ucol = lane % UPR;
  const int n = col0w + ucol * 8;
  float cs8[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) cs8[k] = 0.f;
  // The fused operands (fp32 residual rows / bf16 pre-activation rows) come from HBM at ~2 us latency: ALL of this wave's
  // loads are issued back to back BEFORE the first use (the accumulators are dead after the park, so the registers are
  // free); a load-use-load-use loop exposed that latency once per group of rows and made these epilogues latency-bound.
  constexpr int NIT = WTM / RPI;
  const bool ncol_ok = n < a.N;
  [[maybe_unused]] uint4 uaux[EPI == SAVIT_EPI_DGELU ? NIT : 1];
  if constexpr (EPI == SAVIT_EPI_RESID) {
    // fp32 rows: 4 columns (one float4) per lane, so a wave instruction moves whole 128-B lines (RPR rows x WTN*4 B); the
    // 8-column mapping of the bf16 epilogues would split every fp32 line between two instructions.
    constexpr int LPR = WTN / 4;       // lanes per row (16 or 8)
    constexpr int RPR = 64 / LPR;      // rows per instruction
    constexpr int NR = WTM / RPR;
    const int rrow = lane / LPR, rcol = lane % LPR;
    const int nn = col0w + rcol * 4;
    const bool nn_ok = nn < a.N;
    float cscale[4] = {1.f, 1.f, 1.f, 1.f};
    if (a.colscale != nullptr && nn_ok) {
      const float4 c0 = *reinterpret_cast<const float4*>(a.colscale + nn);
      cscale[0] = c0.x; cscale[1] = c0.y; cscale[2] = c0.z; cscale[3] = c0.w;
    }
    float4 res[NR];
#pragma unroll
    for (int it = 0; it < NR; ++it) {
      const int m = row0w + it * RPR + rrow;
      res[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < a.M && nn_ok)
        res[it] = nt_load_f4(reinterpret_cast<const float*>(a.aux) + (size_t)m * a.ldaux + nn);  // not read again before backward
    }
#pragma unroll
    for (int it = 0; it < NR; ++it) {
      const int ml = it * RPR + rrow;
      const int m = row0w + ml;
      const uint2 raw = *reinterpret_cast<const uint2*>(wsm + ml * ROWB + (((rcol >> 1) ^ (ml & (UPR - 1))) << 4) + (rcol & 1) * 8);
      if (m >= a.M || !nn_ok) continue;
      float rs = 1.0f;
      if (a.rowscale != nullptr) rs = a.rowscale[m / a.rows_per_sample];
      if (a.C2 != nullptr)  // the bf16 branch value, needed by the LayerScale gradient (layerscale.py:23)
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(a.C2) + (size_t)m * a.ldc + nn) = raw;
      const float4 r = res[it];
      float4 o = make_float4(r.x + rs * cscale[0] * __uint_as_float(raw.x << 16), r.y + rs * cscale[1] * __uint_as_float(raw.x & 0xffff0000u),
                             r.z + rs * cscale[2] * __uint_as_float(raw.y << 16), r.w + rs * cscale[3] * __uint_as_float(raw.y & 0xffff0000u));
      if (a.round_out_bf16) o = make_float4(round_bf16(o.x), round_bf16(o.y), round_bf16(o.z), round_bf16(o.w));  // bf16 residual stream
      epi_store_f4(reinterpret_cast<float*>(a.C) + (size_t)m * a.ldc + nn, o);
    }
    return;
  }
  if (EPI == SAVIT_EPI_DGELU) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int m = row0w + it * RPI + urow;
      uaux[it] = make_uint4(0u, 0u, 0u, 0u);
      if (m < a.M && ncol_ok) {
        const bf16_t* up = reinterpret_cast<const bf16_t*>(a.aux) + (size_t)m * a.ldaux + n;
        uaux[it] = nt_load_u4(up);  // last use of the saved pre-activation
      }
    }
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int ml = it * RPI + urow;
    const int m = row0w + ml;
    const uint4 raw = *reinterpret_cast<const uint4*>(wsm + ml * ROWB + ((ucol ^ (ml & (UPR - 1))) << 4));
    if (m >= a.M || !ncol_ok) continue;
    const uint32_t rw[4] = {raw.x, raw.y, raw.z, raw.w};
    if (EPI == SAVIT_EPI_BF16) {
      epi_store_u4(reinterpret_cast<bf16_t*>(a.C) + (size_t)m * a.ldc + n, raw);
    } else if (EPI == SAVIT_EPI_BIAS_GELU) {
      uint32_t g[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x2 gv = gelu_tanh2(unpack_bf16x2(rw[k]));
        g[k] = pack_bf16x2(gv.x, gv.y);
      }
      // the pre-activation is only read again in backward: non-temporal, so it does not push the activation out of the cache
      nt_store_u4(reinterpret_cast<bf16_t*>(a.C) + (size_t)m * a.ldc + n, raw);
      epi_store_u4(reinterpret_cast<bf16_t*>(a.C2) + (size_t)m * a.ldc + n, make_uint4(g[0], g[1], g[2], g[3]));
    } else if (EPI == SAVIT_EPI_DGELU) {
      const uint4 uraw = uaux[it];
      const uint32_t uw[4] = {uraw.x, uraw.y, uraw.z, uraw.w};
      uint32_t o[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x2 dv = unpack_bf16x2(rw[k]) * gelu_tanh_grad2(unpack_bf16x2(uw[k]));
        o[k] = pack_bf16x2(dv.x, dv.y);
        const f32x2 r = unpack_bf16x2(o[k]);  // the column sum adds the bf16-rounded values, as the stored tensor holds them
        cs8[2 * k] += r.x;
        cs8[2 * k + 1] += r.y;
      }
      epi_store_u4(reinterpret_cast<bf16_t*>(a.C) + (size_t)m * a.ldc + n, make_uint4(o[0], o[1], o[2], o[3]));
    }
  }
  if (EPI == SAVIT_EPI_DGELU && a.colsum != nullptr) {
    // lanes with equal (lane % UPR) own the same 8 columns: fold the RPI row groups, then one atomic per column
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float sdu = cs8[k];
      if (UPR <= 32) sdu += __shfl_xor(sdu, 32, 64);
      if (UPR <= 16) sdu += __shfl_xor(sdu, 16, 64);
      if (UPR <= 8) sdu += __shfl_xor(sdu, 8, 64);
      if (lane < UPR && n + k < a.N) {
        if (a.colsum_rows > 0)
          a.colsum[(size_t)slab_row * a.N + n + k] = sdu;  // this (row tile, wave row)'s partial: no atomics, deterministic
        else
          atomicAdd(a.colsum + n + k, sdu);
      }
    }
  }
}

template <int EPI>
constexpr bool epi_uses_lds() { return EPI == SAVIT_EPI_BF16 || EPI == SAVIT_EPI_BIAS_GELU || EPI == SAVIT_EPI_RESID || EPI == SAVIT_EPI_DGELU; }

template <int BM, int BN, int WGM, int WGN, int EPI>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_tn_kernel(const GemmParams p) {
  constexpr int NW = WGM * WGN;
  constexpr int WTM = BM / WGM, WTN = BN / WGN;  // wave tile
  constexpr int MI = WTM / 16, NI = WTN / 16;
  constexpr int A_BYTES = BM * ROW_BYTES, B_BYTES = BN * ROW_BYTES;
  constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr bool PATCH = (EPI == SAVIT_EPI_PATCH);
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const savit_gemm_args& a = p.a;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WGN, wn = wave % WGN;

  const int nwg = p.tiles_m * p.tiles_n;
  const int tid = xcd_remap(blockIdx.x, nwg);
  const int tm = tid / p.tiles_n, tn = tid - tm * p.tiles_n;
  const int row0 = tm * BM, col0 = tn * BN;

  // buffer descriptors rebased to this tile's first row: OOB rows read as zero.
  const bf16_t* Abase = reinterpret_cast<const bf16_t*>(a.A);
  const bf16_t* Bbase = reinterpret_cast<const bf16_t*>(a.Bt) + (size_t)col0 * a.ldb;
  uint32_t a_bytes, b_bytes;
  if (PATCH) {
    const size_t tot = (size_t)(a.M / (p.grid_side * p.grid_side)) * a.img_size * a.img_size * 3 * 2;
    a_bytes = (uint32_t)(tot > 0xfffffff0ull ? 0xfffffff0ull : tot);
  } else {
    Abase += (size_t)row0 * a.lda;
    const size_t tot = (size_t)(a.M - row0) * a.lda * 2;
    a_bytes = (uint32_t)(tot > 0xfffffff0ull ? 0xfffffff0ull : tot);
  }
  {
    const size_t tot = (size_t)(a.N - col0) * a.ldb * 2;
    b_bytes = (uint32_t)(tot > 0xfffffff0ull ? 0xfffffff0ull : tot);
  }
  const auto srdA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Abase), 0, a_bytes, 0x00020000);
  const auto srdB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Bbase), 0, b_bytes, 0x00020000);

  // ---- staging geometry: one wave-instruction fills 8 rows x 128 B; lane -> (row = lane>>3, phys chunk = lane&7)
  constexpr int A_INSTR = BM / 8 / NW, B_INSTR = BN / 8 / NW;  // per wave
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile/wave mismatch");
  const int lrow = lane >> 3, pch = lane & 7;

  auto stage = [&](int kt, int buf) {
    char* sA = smem + buf * STAGE_BYTES;
    char* sB = sA + A_BYTES;
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
      const int inst = wave * A_INSTR + i;  // uniform
      const int r = inst * 8 + lrow;        // tile-local row
      const int c = pch ^ ((r >> 1) & 7);   // logical 16-B chunk held at this physical slot
      uint32_t voff;
      if (PATCH) {
        const int m = row0 + r;
        const int ppi = p.grid_side * p.grid_side;
        const int b = m / ppi, pp = m - b * ppi;
        const int pi = pp / p.grid_side, pj = pp - pi * p.grid_side;
        const int kc = kt * 8 + c;  // global 16-B chunk index along K
        const int ph = kc / p.chunks_per_prow, within = kc - ph * p.chunks_per_prow;
        const size_t pix = ((size_t)b * a.img_size + (size_t)pi * a.patch + ph) * a.img_size + (size_t)pj * a.patch;
        voff = (m < a.M) ? (uint32_t)(pix * 6 + (size_t)within * 16) : 0xfffffff0u;
      } else {
        voff = (uint32_t)r * (uint32_t)(a.lda * 2) + (uint32_t)(kt * ROW_BYTES + c * 16);
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srdA, (__attribute__((address_space(3))) void*)(sA + inst * 1024), 16, voff, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) {
      const int inst = wave * B_INSTR + i;
      const int r = inst * 8 + lrow;
      const int c = pch ^ ((r >> 1) & 7);
      const uint32_t voff = (uint32_t)r * (uint32_t)(a.ldb * 2) + (uint32_t)(kt * ROW_BYTES + c * 16);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srdB, (__attribute__((address_space(3))) void*)(sB + inst * 1024), 16, voff, 0, 0, 0);
    }
  };

  // ---- fragment read offsets (bytes within a stage).  Row r = base16 + (lane&15); chunk c = 4*ks + (lane>>4);
  //      physical chunk = c ^ ((r>>1)&7) and (r>>1)&7 == ((lane&15)>>1) because base16 % 16 == 0.
  const int fr = lane & 15, fq = lane >> 4;
  const int swz = fr >> 1;
  const int off_k0 = fr * ROW_BYTES + ((fq ^ swz) << 4);        // ks = 0: logical chunk fq
  const int off_k1 = fr * ROW_BYTES + (((4 + fq) ^ swz) << 4);  // ks = 1: logical chunk 4+fq
  const int a_frag_base = (wm * WTM) * ROW_BYTES;
  const int b_frag_base = A_BYTES + (wn * WTN) * ROW_BYTES;

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int KT = a.K / BK;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < KT) stage(kt + 1, cur ^ 1);
    const char* sbase = smem + cur * STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int off = ks == 0 ? off_k0 : off_k1;
      bf16x8 af[MI], bfr[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sbase + a_frag_base + i * 16 * ROW_BYTES + off);
#pragma unroll
      for (int j = 0; j < NI; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(sbase + b_frag_base + j * 16 * ROW_BYTES + off);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue
  if constexpr (epi_uses_lds<EPI>()) {
    // the staging LDS is idle now (the last K-tile's barrier has passed): each wave parks its tile in its own slice
    epilogue_lds<EPI, WTM, WTN, MI, NI>(p, acc, smem + wave * (WTM * WTN * 2), row0 + wm * WTM, col0 + wn * WTN, lane, tm * WGM + wm);
  } else {
    // direct stores: lane holds C[m = .. + (lane&15)][n = .. + 4*(lane>>4) + 0..3]
    const int mrow = row0 + wm * WTM + fr;
    const int ncol = col0 + wn * WTN + fq * 4;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      float csum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < MI; ++i) epilogue_store<EPI>(p, mrow + i * 16, ncol + j * 16, acc[i][j], csum);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Ring-pipelined variant: K-stages of 32 (64-B LDS rows), S-slot LDS ring filled by LDS-DMA with COUNTED vmcnt
// (loads stay in flight across the per-stage barrier), and register-prefetched fragments: the fragments of stage
// kt+1 are read from LDS while the MFMAs of stage kt issue, so MFMAs resume right after each barrier.
//   slot bytes = (BM + BN) * 64 ; per stage a wave issues G = (BM + BN)/16/NW LDS-DMA instructions.
//   16-B chunk swizzle for 64-B rows: phys = chunk ^ ((-(row>>2)) & 3)  (conflict-free ds_read_b128, checked per
//   16-lane read group), applied on the DMA source address and on the fragment read.

template <int BM, int BN, int WGM, int WGN, int S, int EPI, bool LATE = false>
__global__ __launch_bounds__(64 * WGM * WGN, 2) void gemm_tn_ring_kernel(const GemmParams p) {
  constexpr int NW = WGM * WGN;
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int MI = WTM / 16, NI = WTN / 16;
  constexpr int RB = 64;
  constexpr int A_BYTES = BM * RB, B_BYTES = BN * RB, STAGE = A_BYTES + B_BYTES;
  constexpr int A_INSTR = BM / 16 / NW, B_INSTR = BN / 16 / NW, G = A_INSTR + B_INSTR;
  constexpr bool PATCH = (EPI == SAVIT_EPI_PATCH);
  static_assert(BM % (16 * NW) == 0 && BN % (16 * NW) == 0, "tile/wave mismatch");
  static_assert(G * (S - 2) <= 63, "vmcnt immediate");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const savit_gemm_args& a = p.a;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int nwg = p.tiles_m * p.tiles_n;
  const int tid = xcd_remap(blockIdx.x, nwg);
  const int tm = tid / p.tiles_n, tn = tid - tm * p.tiles_n;
  const int row0 = tm * BM, col0 = tn * BN;

  const bf16_t* Abase = reinterpret_cast<const bf16_t*>(a.A);
  const bf16_t* Bbase = reinterpret_cast<const bf16_t*>(a.Bt) + (size_t)col0 * a.ldb;
  uint32_t a_bytes, b_bytes;
  if (PATCH) {
    const size_t tot = (size_t)(a.M / (p.grid_side * p.grid_side)) * a.img_size * a.img_size * 3 * 2;
    a_bytes = (uint32_t)(tot > 0xfffffff0ull ? 0xfffffff0ull : tot);
  } else {
    Abase += (size_t)row0 * a.lda;
    const size_t tot = (size_t)(a.M - row0) * a.lda * 2;
    a_bytes = (uint32_t)(tot > 0xfffffff0ull ? 0xfffffff0ull : tot);
  }
  {
    const size_t tot = (size_t)(a.N - col0) * a.ldb * 2;
    b_bytes = (uint32_t)(tot > 0xfffffff0ull ? 0xfffffff0ull : tot);
  }
  const auto srdA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Abase), 0, a_bytes, 0x00020000);
  const auto srdB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Bbase), 0, b_bytes, 0x00020000);

  // DMA geometry: one wave-instruction = 16 rows x 64 B; lane -> (row = lane>>2, physical chunk = lane&3)
  const int lrow = lane >> 2, pch = lane & 3;
  // one LDS-DMA wave-instruction of stage kt: piece g < A_INSTR loads A rows, the rest B rows
  auto stage_piece = [&](int kt, int slot, int g) {
    char* sA = smem + slot * STAGE;
    char* sB = sA + A_BYTES;
    if (g < A_INSTR) {
      const int inst = wave * A_INSTR + g;
      const int r = inst * 16 + lrow;
      const int c = pch ^ ((-(r >> 2)) & 3);  // logical 16-B chunk held at this physical slot
      uint32_t voff;
      if (PATCH) {
        const int m = row0 + r;
        const int ppi = p.grid_side * p.grid_side;
        const int b = m / ppi, pp = m - b * ppi;
        const int pi = pp / p.grid_side, pj = pp - pi * p.grid_side;
        const int kc = kt * 4 + c;
        const int ph = kc / p.chunks_per_prow, within = kc - ph * p.chunks_per_prow;
        const size_t pix = ((size_t)b * a.img_size + (size_t)pi * a.patch + ph) * a.img_size + (size_t)pj * a.patch;
        voff = (m < a.M) ? (uint32_t)(pix * 6 + (size_t)within * 16) : 0xfffffff0u;
      } else {
        voff = (uint32_t)r * (uint32_t)(a.lda * 2) + (uint32_t)(kt * RB + c * 16);
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srdA, (__attribute__((address_space(3))) void*)(sA + inst * 1024), 16, voff, 0, 0, 0);
    } else {
      const int inst = wave * B_INSTR + (g - A_INSTR);
      const int r = inst * 16 + lrow;
      const int c = pch ^ ((-(r >> 2)) & 3);
      uint32_t voffb = (uint32_t)r * (uint32_t)(a.ldb * 2) + (uint32_t)(kt * RB + c * 16);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srdB, (__attribute__((address_space(3))) void*)(sB + inst * 1024), 16, voffb, 0, 0, 0);
    }
  };
  auto stage = [&](int kt, int slot) {
#pragma unroll
    for (int g = 0; g < G; ++g) stage_piece(kt, slot, g);
  };

  // fragment read offsets inside a slot: row = base16 + (lane&15), logical chunk = lane>>4
  const int fr = lane & 15, fq = lane >> 4;
  const int foff = fr * RB + ((fq ^ ((-(fr >> 2)) & 3)) << 4);
  const int a_frag = (wm * WTM) * RB + foff;
  const int b_frag = A_BYTES + (wn * WTN) * RB + foff;

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int KT = a.K / 32;
  int slot_issue = 0;
  // LATE: stage kt+S is issued AFTER the barrier of step kt, interleaved with that step's MFMAs (the slot of stage kt is free
  // then: its fragments were read during step kt-1).  Otherwise stage kt+S-1 is issued before the barrier.
  constexpr int PRE = LATE ? S : S - 1;
  const int pre = (PRE < KT) ? PRE : KT;
  for (int s = 0; s < pre; ++s) {
    stage(s, slot_issue);
    slot_issue = (slot_issue + 1 == S) ? 0 : slot_issue + 1;
  }
  if (pre == PRE) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G * (PRE - 1)) : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();

  bf16x8 af0[MI], af1[MI], bfr[NI];
#pragma unroll
  for (int i = 0; i < MI; ++i) af0[i] = *reinterpret_cast<const bf16x8*>(smem + a_frag + i * 16 * RB);
#pragma unroll
  for (int j = 0; j < NI; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(smem + b_frag + j * 16 * RB);
  int slot_next = (S > 1) ? 1 : 0;  // slot holding stage kt+1

  auto body = [&](bf16x8(&ac)[MI], bf16x8(&an)[MI], int kt) {
    bool issue;
    if constexpr (LATE) {
      issue = (kt + S < KT);
      // stages kt+2 .. kt+S-1 may stay in flight; in the tail (nothing left to issue) drain instead of counting
      if (kt + S - 1 < KT) {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(G * (S - 2)) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
    } else {
      issue = (kt + S - 1 < KT);
      if (issue) {
        stage(kt + S - 1, slot_issue);
        slot_issue = (slot_issue + 1 == S) ? 0 : slot_issue + 1;
      }
      if (issue) {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(G * (S - 2)) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
    }
    __builtin_amdgcn_s_barrier();
    const char* nb = smem + slot_next * STAGE;
    constexpr int APJ = (MI + NI - 1) / NI;  // A refills per n-tile step
    constexpr int GPJ = (G + NI - 1) / NI;   // LDS-DMA pieces per n-tile step (LATE)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
#pragma unroll
      for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], ac[i], acc[i][j], 0, 0, 0);
      // refill in place for stage kt+1 (after the last stage this reads a valid but unused slot: branch-free)
      bfr[j] = *reinterpret_cast<const bf16x8*>(nb + b_frag + j * 16 * RB);
#pragma unroll
      for (int q = 0; q < APJ; ++q) {
        const int i = j * APJ + q;
        if (i < MI) an[i] = *reinterpret_cast<const bf16x8*>(nb + a_frag + i * 16 * RB);
      }
      if constexpr (LATE) {
        if (issue) {
#pragma unroll
          for (int q = 0; q < GPJ; ++q)
            if (j * GPJ + q < G) stage_piece(kt + S, slot_issue, j * GPJ + q);
        }
      }
    }
    if constexpr (LATE) {
      if (issue) slot_issue = (slot_issue + 1 == S) ? 0 : slot_issue + 1;
    }
    slot_next = (slot_next + 1 == S) ? 0 : slot_next + 1;
  };

  int kt = 0;
  for (; kt + 1 < KT; kt += 2) {
    body(af0, af1, kt);
    body(af1, af0, kt + 1);
  }
  if (kt < KT) body(af0, af1, kt);

  if constexpr (epi_uses_lds<EPI>()) {
    // all waves must be done reading the ring (the last body's refill reads included) before it is reused
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    epilogue_lds<EPI, WTM, WTN, MI, NI>(p, acc, smem + wave * (WTM * WTN * 2), row0 + wm * WTM, col0 + wn * WTN, lane, tm * WGM + wm);
  } else {
    const int mrow = row0 + wm * WTM + fr;
    const int ncol = col0 + wn * WTN + fq * 4;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      float csum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < MI; ++i) epilogue_store<EPI>(p, mrow + i * 16, ncol + j * 16, acc[i][j], csum);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Paired-stage variant.  The ring kernel above stages 64-B row pieces (K = 32), so every LDS-DMA wave-instruction touches
// 16 HALF cache lines and each 128-B line of A / W is requested twice, one K-step apart: measured on MI355X the L2 -> LDS
// feed, not the MFMA pipe, bounds that kernel (removing the main-loop DMA: 0.84 -> 1.3 PFLOP/s on the 128x256 tile).
// Here one wave-instruction fetches 8 rows x 128 B = whole lines, and a ring slot holds a PAIR of K-steps:
//   slot bytes = (BM + BN) * 128 ; row r at r * 128 ; logical 16-B chunk c (0-3: even K-step, 4-7: odd K-step) stored at
//   physical chunk c ^ ((r >> 1) & 7)   (conflict-free for ds_read_b128's 16-lane groups, applied on the DMA source side).
// Schedule per pair p (fragments are register-prefetched one K-step ahead as above):
//   even step : MFMAs of K-step 2p   | read fragments of 2p+1 (same slot, no barrier: nothing was overwritten)
//   s_waitcnt vmcnt (slot p+1 landed) ; s_barrier
//   odd step  : MFMAs of K-step 2p+1 | read fragments of 2p+2 (slot p+1) | issue the DMA of pair p+ND into slot p (free now)
// i.e. ONE barrier per two K-steps, and the DMA issue is spread between the MFMAs of the odd step.
template <int BM, int BN, int WGM, int WGN, int ND, int EPI>
__device__ __forceinline__ void pair_tile(const GemmParams& p, char* smem, const int tm, const int tn, const int row_base) {
  constexpr int NW = WGM * WGN;
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int MI = WTM / 16, NI = WTN / 16;
  constexpr int RB = 128;
  constexpr int A_BYTES = BM * RB, B_BYTES = BN * RB, SLOT = A_BYTES + B_BYTES;
  constexpr int A_INSTR = BM / 8 / NW, B_INSTR = BN / 8 / NW, G = A_INSTR + B_INSTR;
  constexpr bool PATCH = (EPI == SAVIT_EPI_PATCH);
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile/wave mismatch");
  static_assert(ND >= 2 && G * (ND - 2) <= 63, "ring depth / vmcnt immediate");
  const savit_gemm_args& a = p.a;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int row0 = row_base + tm * BM, col0 = tn * BN;

  const bf16_t* Abase = reinterpret_cast<const bf16_t*>(a.A);
  const bf16_t* Bbase = reinterpret_cast<const bf16_t*>(a.Bt) + (size_t)col0 * a.ldb;
  uint32_t a_bytes, b_bytes;
  if (PATCH) {
    const size_t tot = (size_t)(a.M / (p.grid_side * p.grid_side)) * a.img_size * a.img_size * 3 * 2;
    a_bytes = (uint32_t)(tot > 0xfffffff0ull ? 0xfffffff0ull : tot);
  } else {
    Abase += (size_t)row0 * a.lda;
    const size_t tot = (size_t)(a.M - row0) * a.lda * 2;
    a_bytes = (uint32_t)(tot > 0xfffffff0ull ? 0xfffffff0ull : tot);
  }
  {
    const size_t tot = (size_t)(a.N - col0) * a.ldb * 2;
    b_bytes = (uint32_t)(tot > 0xfffffff0ull ? 0xfffffff0ull : tot);
  }
  const auto srdA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Abase), 0, a_bytes, 0x00020000);
  const auto srdB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Bbase), 0, b_bytes, 0x00020000);

  // DMA geometry: one wave-instruction = 8 rows x 128 B; lane -> (row = lane>>3, physical chunk = lane&7)
  const int lrow = lane >> 3, pch = lane & 7;
  auto pair_piece = [&](int pp, int slot, int g) {
    char* sA = smem + slot * SLOT;
    char* sB = sA + A_BYTES;
    if (g < A_INSTR) {
      const int inst = wave * A_INSTR + g;
      const int r = inst * 8 + lrow;
      const int c = pch ^ ((r >> 1) & 7);  // logical 16-B chunk held at this physical position
      uint32_t voff;
      if (PATCH) {
        const int m = row0 + r;
        const int ppi = p.grid_side * p.grid_side;
        const int b = m / ppi, pq = m - b * ppi;
        const int pi = pq / p.grid_side, pj = pq - pi * p.grid_side;
        const int kc = pp * 8 + c;
        const int ph = kc / p.chunks_per_prow, within = kc - ph * p.chunks_per_prow;
        const size_t pix = ((size_t)b * a.img_size + (size_t)pi * a.patch + ph) * a.img_size + (size_t)pj * a.patch;
        voff = (m < a.M) ? (uint32_t)(pix * 6 + (size_t)within * 16) : 0xfffffff0u;
      } else {
        voff = (uint32_t)r * (uint32_t)(a.lda * 2) + (uint32_t)(pp * RB + c * 16);
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srdA, (__attribute__((address_space(3))) void*)(sA + inst * 1024), 16, voff, 0, 0, 0);
    } else {
      const int inst = wave * B_INSTR + (g - A_INSTR);
      const int r = inst * 8 + lrow;
      const int c = pch ^ ((r >> 1) & 7);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srdB, (__attribute__((address_space(3))) void*)(sB + inst * 1024), 16,
                                               (uint32_t)r * (uint32_t)(a.ldb * 2) + (uint32_t)(pp * RB + c * 16), 0, 0, 0);
    }
  };

  // fragment read offsets inside a slot: row = base16 + (lane&15); logical chunk = (lane>>4) + 4 * (K-step parity)
  const int fr = lane & 15, fq = lane >> 4;
  const int sw = (fr >> 1) & 7;
  const int foff0 = fr * RB + ((fq ^ sw) << 4);
  const int foff1 = fr * RB + (((fq + 4) ^ sw) << 4);
  const int a_row = (wm * WTM) * RB, b_row = A_BYTES + (wn * WTN) * RB;

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int KP = a.K / 64;
  const int pre = (ND < KP) ? ND : KP;
  for (int s = 0; s < pre; ++s) {
#pragma unroll
    for (int g = 0; g < G; ++g) pair_piece(s, s, g);
  }
  if (pre == ND) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G * (ND - 1)) : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();

  bf16x8 af0[MI], af1[MI], bfr[NI];
#pragma unroll
  for (int i = 0; i < MI; ++i) af0[i] = *reinterpret_cast<const bf16x8*>(smem + a_row + foff0 + i * 16 * RB);
#pragma unroll
  for (int j = 0; j < NI; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(smem + b_row + foff0 + j * 16 * RB);

  constexpr int APJ = (MI + NI - 1) / NI;  // A refills per n-tile step
  constexpr int GPJ = (G + NI - 1) / NI;   // LDS-DMA pieces per n-tile step (odd steps)
  int slot = 0;                            // slot of pair pp
  for (int pp = 0; pp < KP; ++pp) {
    const int slot_nx = (slot + 1 == ND) ? 0 : slot + 1;
    // ---- even step: K-step 2pp from af0/bfr; prefetch K-step 2pp+1 (same slot, chunks 4-7) into af1/bfr
    {
      const char* nb = smem + slot * SLOT;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af0[i], acc[i][j], 0, 0, 0);
        bfr[j] = *reinterpret_cast<const bf16x8*>(nb + b_row + foff1 + j * 16 * RB);
#pragma unroll
        for (int q = 0; q < APJ; ++q) {
          const int i = j * APJ + q;
          if (i < MI) af1[i] = *reinterpret_cast<const bf16x8*>(nb + a_row + foff1 + i * 16 * RB);
        }
      }
    }
    // pair pp+1 must have landed (pairs pp+2 .. pp+ND-1 may stay in flight); all waves are done reading slot `slot`
    if (pp + ND - 1 < KP) {
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(G * (ND - 2)) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    // ---- odd step: K-step 2pp+1 from af1/bfr; prefetch K-step 2pp+2 (next slot, chunks 0-3); refill slot `slot` with pair pp+ND
    {
      const bool issue = (pp + ND < KP);
      const char* nb = smem + slot_nx * SLOT;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af1[i], acc[i][j], 0, 0, 0);
        // after the last pair this reads a valid but unused slot: branch-free
        bfr[j] = *reinterpret_cast<const bf16x8*>(nb + b_row + foff0 + j * 16 * RB);
#pragma unroll
        for (int q = 0; q < APJ; ++q) {
          const int i = j * APJ + q;
          if (i < MI) af0[i] = *reinterpret_cast<const bf16x8*>(nb + a_row + foff0 + i * 16 * RB);
        }
        if (issue) {
#pragma unroll
          for (int q = 0; q < GPJ; ++q)
            if (j * GPJ + q < G) pair_piece(pp + ND, slot, j * GPJ + q);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    slot = slot_nx;
  }

  if constexpr (epi_uses_lds<EPI>()) {
    // all waves must be done reading the ring (the last step's refill reads included) before it is reused
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    epilogue_lds<EPI, WTM, WTN, MI, NI>(p, acc, smem + wave * (WTM * WTN * 2), row0 + wm * WTM, col0 + wn * WTN, lane, tm * WGM + wm);
  } else {
    const int mrow = row0 + wm * WTM + fr;
    const int ncol = col0 + wn * WTN + fq * 4;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      float csum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < MI; ++i) epilogue_store<EPI>(p, mrow + i * 16, ncol + j * 16, acc[i][j], csum);
    }
  }
}

template <int BM, int BN, int WGM, int WGN, int ND, int EPI>
__global__ __launch_bounds__(64 * WGM * WGN, 2) void gemm_tn_pair_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
  const int tm = tid / p.tiles_n;
  pair_tile<BM, BN, WGM, WGN, ND, EPI>(p, smem, tm, tid - tm * p.tiles_n, 0);
}

// Mixed tile heights in ONE launch ("tail split").  792 tiles of 192 x 128 on 512 workgroup slots are 1.55 rounds: the second
// round runs at 55 % occupancy and the launch takes 2 tile times.  Here the first `big_tiles` workgroups (whole rounds of slots)
// take BM-row tiles over the first big_rows rows and the remaining rows are cut into shorter BM2-row tiles, which start as the
// big ones drain: about 1 + BM2/BM tile times.  Same arithmetic per output element (same K order), so results are bitwise those of
// the plain kernel.  Epilogues without column sums only (the slab index is per row tile).
template <int BM, int BM2, int BN, int WGM, int WGN, int ND, int EPI>
__global__ __launch_bounds__(64 * WGM * WGN, 2) void gemm_tn_pair_tail_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.x;
  if (b < p.big_tiles) {
    const int tid = xcd_remap(b, p.big_tiles);
    const int tm = tid / p.tiles_n;
    pair_tile<BM, BN, WGM, WGN, ND, EPI>(p, smem, tm, tid - tm * p.tiles_n, 0);
  } else {
    const int tid = xcd_remap(b - p.big_tiles, p.small_tiles);
    const int tm = tid / p.tiles_n;
    pair_tile<BM2, BN, WGM, WGN, ND, EPI>(p, smem, tm, tid - tm * p.tiles_n, p.big_rows);
  }
}

// ------------------------------------------------------------------------------------------------------------
// Ping-pong variant (256 x 256 x 64, 8 waves = 2 (M) x 4 (N), one workgroup per CU).  In the kernels above all waves of a workgroup
// do the same thing at the same time: the two waves that share a SIMD both want the matrix pipe, then both wait at the barrier.
// Here the waves of M-half 0 (waves 0-3, one per SIMD) and of M-half 1 (waves 4-7) run the SAME program one barrier apart: between
// two consecutive barriers one group issues the 16 MFMAs of a 64 x 32 output quadrant (K = 64) while its SIMD partners read their
// next fragments from LDS and issue the LDS-DMA of a later K-tile, then the roles swap - the matrix pipe of every SIMD always has a
// wave in its MFMA segment (cdna_hip_programming.md "256^2 8-phase template"; MI355X_MICROARCH.md "Two waves per SIMD").
//   LDS: 2 buffers x 64 KB (one K-tile: 256 A rows + 256 Bt rows of 128 B), same XOR swizzle as the pair kernel (conflict-free
//   ds_read_b128, applied on the LDS-DMA source side).  A buffer is cut into 8 UNITS of 64 rows: A(h, qm) = unit 2h + qm (rows
//   of M-half h, 64-row quadrant qm), B(wc) = unit 4 + wc (the 64 Bt rows = output columns of the waves in N-quarter wc).  One unit
//   = one 1-KB LDS-DMA instruction from each of the 8 waves.
//   K-tile t, phase p = 0..3 computes quadrant (qm, qn) = (0,0) (0,1) (1,1) (1,0) of the wave's 128 x 64 tile.  Load segment
//   L(t,p) (what a wave does between its previous MFMA segment and the barrier that opens this one):
//     p0: read A(wr,0) [8 x ds_read_b128] and the qn = 0 half of B(wc) [4]   | DMA units 6, 7 of K-tile t+1
//     p1: read the qn = 1 half of B(wc) [4]                                    | DMA units 1, 3 of K-tile t+1
//     p2: read A(wr,1) [8]                                                     | DMA units 0, 2 of K-tile t+2
//     p3: -                                                                    | DMA units 4, 5 of K-tile t+2 ; s_waitcnt vmcnt(4)
//   Hazards.  WAR: a unit is overwritten two phases after its last read or later (A(h,0): read p0, written from p2 on; B: read
//   p0/p1, written p3 / next p0; A(h,1): read p2, written next p1) - with the groups one barrier apart that is at least one
//   barrier after the last reader's lgkmcnt(0).  RAW: K-tile t+1 is complete in LDS once every wave has passed the vmcnt(4) of
//   L(t,p3) (all but the four DMAs of K-tile t+2 have landed) and the barrier behind it; its first read is in L(t+1,p0), which
//   for both groups lies behind a barrier that every wave reaches after that wait.
template <int EPI, int ABL = 0>  // ABL != 0: timing-only ablations, instantiated in SAVIT_EXPERIMENTS builds only: 1 no main-loop DMA, 2 no LDS reads, 4 no barriers, 8 no MFMA
__global__ __launch_bounds__(512) void gemm_tn_pp_kernel(const GemmParams p) {
  constexpr int BM = 256, BN = 256, RB = 128;
  constexpr int UNIT = 64 * RB, BUF = (BM + BN) * RB;
  constexpr int WTM = 128, WTN = 64, MI = 8, NI = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const savit_gemm_args& a = p.a;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int nwg = p.tiles_m * p.tiles_n;
  const int tid = xcd_remap(blockIdx.x, nwg);
  // Tile order inside the XCD's contiguous chunk: row panels in groups of G, and inside a group column by column (all G rows of a
  // column, then the next column).  The 32 tiles an XCD runs at a time then touch about G A-panels and 32/G W-panels instead of
  // 2-3 A-panels and EVERY W-panel (N = 3072: 12 panels = 4.7 MB, more than the 4 MB L2, so each wave of tiles re-fetched all of W:
  // 242 MB of L2 misses per fc1 launch for 43 MB of operands), and the group's A-panels stay in L2 across its columns.
  int tm, tn;
  {
    const int G = p.row_group, per = G * p.tiles_n;
    const int g = tid / per, rem = tid - g * per;
    const int rows_g = (p.tiles_m - g * G) < G ? (p.tiles_m - g * G) : G;
    tn = rem / rows_g;
    tm = g * G + (rem - tn * rows_g);
  }
  const int row0 = tm * BM, col0 = tn * BN;

  const bf16_t* Abase = reinterpret_cast<const bf16_t*>(a.A) + (size_t)row0 * a.lda;
  const bf16_t* Bbase = reinterpret_cast<const bf16_t*>(a.Bt) + (size_t)col0 * a.ldb;
  uint32_t a_bytes, b_bytes;
  {
    const size_t ta = (size_t)(a.M - row0) * a.lda * 2, tb = (size_t)(a.N - col0) * a.ldb * 2;
    a_bytes = (uint32_t)(ta > 0xfffffff0ull ? 0xfffffff0ull : ta);
    b_bytes = (uint32_t)(tb > 0xfffffff0ull ? 0xfffffff0ull : tb);
  }
  const auto srdA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Abase), 0, a_bytes, 0x00020000);
  const auto srdB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Bbase), 0, b_bytes, 0x00020000);

  // LDS-DMA geometry: one wave-instruction = 8 rows x 128 B; this wave owns rows wave*8 .. +7 of every unit
  const int lrow = lane >> 3, pch = lane & 7;
  const int ur = wave * 8 + lrow;                               // row inside a unit (0..63)
  const int uc = (pch ^ ((ur >> 1) & 7)) * 16;                  // byte offset of the logical chunk stored at this lane's position
  const uint32_t a_voff = (uint32_t)ur * (uint32_t)(a.lda * 2) + (uint32_t)uc;
  const uint32_t b_voff = (uint32_t)ur * (uint32_t)(a.ldb * 2) + (uint32_t)uc;
  const uint32_t a_unit = 64u * (uint32_t)(a.lda * 2), b_unit = 64u * (uint32_t)(a.ldb * 2);
  const int KT = a.K / 64;
  auto dma = [&](int kt, int u) {  // unit u (compile-time) of K-tile kt into buffer kt & 1
    if (kt >= KT) return;
    if ((ABL & 1) && kt >= 2) return;
    char* dst = smem + (kt & 1) * BUF + u * UNIT + wave * 1024;
    if (u < 4)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srdA, (__attribute__((address_space(3))) void*)dst, 16, a_voff + (uint32_t)u * a_unit, kt * RB, 0, 0);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srdB, (__attribute__((address_space(3))) void*)dst, 16, b_voff + (uint32_t)(u - 4) * b_unit, kt * RB, 0, 0);
  };

  // fragment read offsets (bytes inside a unit): row = 16 i + (lane & 15), logical chunk = 4 ks + (lane >> 4)
  const int fr = lane & 15, fq = lane >> 4;
  const int sw = (fr >> 1) & 7;
  const int foff0 = fr * RB + ((fq ^ sw) << 4);
  const int foff1 = fr * RB + (((fq + 4) ^ sw) << 4);
  const int a_base = (2 * wr) * UNIT, b_base = (4 + wc) * UNIT;

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 af[4][2], bf[4][2];

#ifdef SAVIT_EXPERIMENTS
  if (p.desync_sleep > 0 && blockIdx.x < 256u) {  // the first round only: later workgroups inherit the phase of the CU they land on
    // spread the workgroups of a round over phase groups: with every CU in its epilogue at the same moment the output stores of
    // a round queue behind one another at the memory side while the matrix pipes idle (and the reverse during the main loops)
    const int n = (int)(blockIdx.x % (unsigned)p.desync_phases) * p.desync_sleep;
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(32);  // ~1 us per unit
  }
#endif
  // prologue: all of K-tile 0 and the four units of K-tile 1 that L(-1,p2), L(-1,p3) would have issued
#pragma unroll
  for (int u = 0; u < 8; ++u) dma(0, u);
  dma(1, 0); dma(1, 2); dma(1, 4); dma(1, 5);
  if (KT > 1) {
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();  // M-half 1 runs one barrier behind M-half 0

#define PP_READ_A(QM)                                                                                          \
  if (!(ABL & 2) || kt == 0) _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                              \
    af[i][0] = *reinterpret_cast<const bf16x8*>(cur + a_base + (QM) * UNIT + i * 16 * RB + foff0);             \
    af[i][1] = *reinterpret_cast<const bf16x8*>(cur + a_base + (QM) * UNIT + i * 16 * RB + foff1);             \
  }
#define PP_READ_B(QN)                                                                                          \
  if (!(ABL & 2) || kt == 0) _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                              \
    bf[2 * (QN) + j][0] = *reinterpret_cast<const bf16x8*>(cur + b_base + (2 * (QN) + j) * 16 * RB + foff0);  \
    bf[2 * (QN) + j][1] = *reinterpret_cast<const bf16x8*>(cur + b_base + (2 * (QN) + j) * 16 * RB + foff1);  \
  }
#define PP_COMPUTE(QM, QN)                                                                                     \
  if (!(ABL & 4)) __builtin_amdgcn_s_barrier();                                                                \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                           \
  __builtin_amdgcn_sched_barrier(0);                                                                           \
  __builtin_amdgcn_s_setprio(1);                                                                               \
  if (!(ABL & 8)) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                             \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                              \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                            \
        acc[4 * (QM) + i][2 * (QN) + j] =                                                                      \
            __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[2 * (QN) + j][ks], af[i][ks], acc[4 * (QM) + i][2 * (QN) + j], 0, 0, 0); \
  __builtin_amdgcn_s_setprio(0);                                                                               \
  __builtin_amdgcn_sched_barrier(0);                                                                           \
  if (!(ABL & 4)) __builtin_amdgcn_s_barrier();

  for (int kt = 0; kt < KT; ++kt) {
    const char* cur = smem + (kt & 1) * BUF;
    // ---- phase 0: quadrant (0,0)
    PP_READ_B(0)
    __builtin_amdgcn_sched_barrier(0);
    PP_READ_A(0)
    dma(kt + 1, 6); dma(kt + 1, 7);
    PP_COMPUTE(0, 0)
    // ---- phase 1: quadrant (0,1)
    PP_READ_B(1)
    dma(kt + 1, 1); dma(kt + 1, 3);
    PP_COMPUTE(0, 1)
    // ---- phase 2: quadrant (1,1)
    PP_READ_A(1)
    dma(kt + 2, 0); dma(kt + 2, 2);
    PP_COMPUTE(1, 1)
    // ---- phase 3: quadrant (1,0); K-tile kt+1 must be complete behind this phase's first barrier
    dma(kt + 2, 4); dma(kt + 2, 5);
    if (kt + 2 < KT) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    PP_COMPUTE(1, 0)
  }
#undef PP_READ_A
#undef PP_READ_B
#undef PP_COMPUTE
  if (wr == 0) __builtin_amdgcn_s_barrier();  // balance the barrier count: every wave is past its last LDS read after this one
  if (ABL & 8) {  // keep the fragments alive
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        asm volatile("" ::"v"(af[i][ks]));
        asm volatile("" ::"v"(bf[i][ks]));
      }
  }

  if constexpr (epi_uses_lds<EPI>()) {
    epilogue_lds<EPI, WTM, WTN, MI, NI>(p, acc, smem + wave * (WTM * WTN * 2), row0 + wr * WTM, col0 + wc * WTN, lane, tm * 2 + wr);
  } else {
    const int mrow = row0 + wr * WTM + fr;
    const int ncol = col0 + wc * WTN + fq * 4;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      float csum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < MI; ++i) epilogue_store<EPI>(p, mrow + i * 16, ncol + j * 16, acc[i][j], csum);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Streaming variant (tile 30): PERSISTENT workgroups, 256 x 128 tiles, and the epilogue of tile t runs UNDER the MFMAs of tile t+1.
// Why: with K = 768 a tile is 12 K-tiles of matrix work and then a burst of HBM traffic (fc1 + GELU writes 310 MB: ~60 us at the
// HBM rate next to ~75 us of MFMA time).  Every kernel above runs the two back to back - all CUs compute, then all CUs store - and a
// launch costs their SUM (141 us).  Overlap inside a workgroup needs the finished tile to live somewhere while the next one
// accumulates: a 256 x 256 tile cannot (128 accumulator registers per wave, 128 KB as bf16), a 256 x 128 tile can - its 64 fp32
// accumulators per wave pack into 32 registers of bf16 at the tile boundary, and those are drained 8 rows at a time through a 2-KB
// per-wave LDS park (transposition to whole 128-B lines, as in epilogue_lds) while the next tile's MFMAs issue.
//   * one workgroup (8 waves = 4 (M) x 2 (N), wave tile 64 x 64) per CU walks over its tiles; the K-pipeline never drains at a tile
//     boundary: a ring of 3 K-tile slots (48 KB each: 256 A rows + 128 Bt rows of 128 B, pair_tile's swizzle), the LDS-DMA of K-tile
//     t+2 - possibly the NEXT tile's operands - is issued during K-tile t;
//   * main loop = the ping-pong schedule of gemm_tn_pp_kernel: waves 0-3 (M-half 0) and their SIMD partners 4-7 (M-half 1) run the
//     same program ONE BARRIER apart; a K-tile is two phases (column halves of the wave tile), each [load segment | barrier |
//     16 MFMAs | barrier], so one group's load segment - fragment reads, LDS-DMA issue, and the epilogue work below - runs beside
//     the other group's MFMA segment.  K-tile t, per wave:
//       L(t,0): read the B fragments of columns 0-31 and all A fragments of slot t%3 | epilogue stage 2 and stage 0
//       L(t,1): read the B fragments of columns 32-63 | issue the 6 DMA pieces of K-tile t+2 into slot (t+2)%3 | epilogue stage 1 |
//               s_waitcnt vmcnt: K-tile t+1 has landed
//     Hazards (group B = one barrier behind group A; interval n = between barriers n and n+1; A runs L(t,0) in interval 4t):
//       WAR: slot (t+2)%3 held K-tile t-1, last read by group B in its L(t-1,1) (interval 4t-1), complete behind the lgkmcnt(0) that
//            follows barrier 4t; the DMA is issued in L(t,1): interval 4t+2 (A) / 4t+3 (B) - at least one barrier later.
//       RAW: every wave waits for its own pieces of K-tile t+1 in L(t,1) and then passes barrier 4t+3 (A) / 4t+4 (B); the first read
//            of K-tile t+1 is group A's L(t+1,0) in interval 4t+4, behind a barrier every wave reached after that wait.
//   * the previous tile's epilogue is cut into 8 chunks (8 rows x 64 columns per wave) of 3 stages, one stage per load segment:
//       stage 0: park a 16-row slice of the packed accumulators (every second chunk; inline-asm ds_write_b64) and request the
//                chunk's rows back row-contiguous (ds_read_b128; + the GELU' pre-activation rows that an LDS-DMA left in the park);
//       stage 1: first half of the math; LDS-DMA of the NEXT chunk's pre-activation rows (or, behind the last chunk, of the bias
//                of the tile being accumulated) into the half of the park this chunk is done with;
//       stage 2: second half of the math, 16-byte buffer stores of whole 128-B lines (+ the column-sum slab row of GELU').
//     Load segments are not interleaved with MFMAs by the compiler (the MFMA segments are fenced), so they may branch: chunk and
//     stage are run-time state, only the slice index of the parked registers is a compile-time switch;
//   * every vector-memory operation is an LDS-DMA or a store, so hipcc has no load result to wait for and inserts no vmcnt wait (for
//     a register-destination load it drains the whole ring; an inline-asm load left its destination registers open to compiler
//     copies while the data was in flight: measured, wrong rows now and then).  The counted wait of L(t,1) accounts for what the
//     epilogue issued since the pieces it waits for: stores of this K-tile's stage 2, the DMA of its stage 1, the 6 pieces of t+2;
//   * rows >= M and the "no previous tile yet" case are handled by the buffer descriptors' bounds checks (loads return 0, stores are
//     dropped): no branches on them.
// Same K order per output element as every other TN kernel: results are bitwise those of tiles 13 / 17 / 20.
template <int EPI>
__global__ __launch_bounds__(512) void gemm_tn_stream_kernel(const GemmParams p) {
  constexpr int BM = 256, BN = 128, RB = 128, ND = 3, G = 6;
  constexpr int A_BYTES = BM * RB, B_BYTES = BN * RB, SLOT = A_BYTES + B_BYTES;
  constexpr int PARK = ND * SLOT;  // 8 x 2 KB behind the ring
  constexpr bool GELU = (EPI == SAVIT_EPI_BIAS_GELU), DG = (EPI == SAVIT_EPI_DGELU);
  static_assert(EPI == SAVIT_EPI_BF16 || GELU || DG, "epilogues of the streaming kernel");
  constexpr int ES = GELU ? 2 : 1;  // stores of a chunk's stage 2 (the last chunk of GELU' adds the two slab stores)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const savit_gemm_args& a = p.a;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;  // waves 0-3: rows 0-127 (group A), waves 4-7: rows 128-255 (group B, their SIMD partners)
  const int grp = wave >> 2;

  // ---- this workgroup's tiles: XCD x owns a contiguous run of the tile list; its `per` workgroups take consecutive tiles, round by round
  const int ntiles = p.tiles_m * p.tiles_n;
  const int xcd = blockIdx.x & 7, wslot = blockIdx.x >> 3, per = gridDim.x >> 3;
  const int q8 = ntiles >> 3, r8 = ntiles & 7;
  const int xbase = (xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int xlen = q8 + (xcd < r8 ? 1 : 0);
  const int NT = wslot < xlen ? (xlen - wslot + per - 1) / per : 0;
  if (NT == 0) return;
  auto tile_rc = [&](int k, int& row0, int& col0) {
    const int tid = xbase + wslot + k * per;
    const int Gr = p.row_group, perg = Gr * p.tiles_n;
    const int g = tid / perg, rem = tid - g * perg;
    const int rows_g = (p.tiles_m - g * Gr) < Gr ? (p.tiles_m - g * Gr) : Gr;
    const int tn = rem / rows_g;
    row0 = (g * Gr + (rem - tn * rows_g)) * BM;
    col0 = tn * BN;
  };
  const int KP = a.K / 64;    // K-tiles per tile (>= 9: stream_ok)
  const int total = NT * KP;  // K-tiles this workgroup computes

  // ---- LDS-DMA side (runs two K-tiles ahead of the MFMAs)
  const int lrow = lane >> 3, pch = lane & 7;
  uint32_t a_voff[4], b_voff[2];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int r = (wave * 4 + g) * 8 + lrow;
    a_voff[g] = (uint32_t)r * (uint32_t)(a.lda * 2) + (uint32_t)((pch ^ ((r >> 1) & 7)) << 4);
  }
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int r = (wave * 2 + g) * 8 + lrow;
    b_voff[g] = (uint32_t)r * (uint32_t)(a.ldb * 2) + (uint32_t)((pch ^ ((r >> 1) & 7)) << 4);
  }
  int d_k = 0, d_pp = 0, d_slot = 0;
  __amdgpu_buffer_rsrc_t srdA, srdB;
  auto dma_set_tile = [&](int k) {
    int r0, c0;
    tile_rc(k, r0, c0);
    const size_t ta = (size_t)(a.M - r0) * a.lda * 2, tb = (size_t)(a.N - c0) * a.ldb * 2;
    srdA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(reinterpret_cast<const bf16_t*>(a.A) + (size_t)r0 * a.lda), 0,
                                             (uint32_t)(ta > 0xfffffff0ull ? 0xfffffff0ull : ta), 0x00020000);
    srdB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(reinterpret_cast<const bf16_t*>(a.Bt) + (size_t)c0 * a.ldb), 0,
                                             (uint32_t)(tb > 0xfffffff0ull ? 0xfffffff0ull : tb), 0x00020000);
  };
  auto dma_ktile = [&]() {  // the 6 pieces of K-tile (d_k, d_pp) into slot d_slot, then advance
    char* base = smem + d_slot * SLOT;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srdA, (__attribute__((address_space(3))) void*)(base + (wave * 4 + g) * 1024), 16, a_voff[g], d_pp * RB, 0, 0);
#pragma unroll
    for (int g = 0; g < 2; ++g)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srdB, (__attribute__((address_space(3))) void*)(base + A_BYTES + (wave * 2 + g) * 1024), 16, b_voff[g],
                                               d_pp * RB, 0, 0);
    d_slot = (d_slot + 1 == ND) ? 0 : d_slot + 1;
    if (++d_pp == KP) {
      d_pp = 0;
      if (++d_k < NT) dma_set_tile(d_k);
    }
  };

  // ---- fragment addressing (as pair_tile)
  const int fr = lane & 15, fq = lane >> 4;
  const int sw = (fr >> 1) & 7;
  const int foff0 = fr * RB + ((fq ^ sw) << 4), foff1 = fr * RB + (((fq + 4) ^ sw) << 4);
  const int a_row = (wm * 64) * RB, b_row = A_BYTES + (wn * 64) * RB;

  // ---- epilogue side
  const __amdgpu_buffer_rsrc_t srdC = __builtin_amdgcn_make_buffer_rsrc(a.C, 0, (uint32_t)((size_t)a.M * a.ldc * 2), 0x00020000);
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t srdC2 =
      __builtin_amdgcn_make_buffer_rsrc(GELU ? a.C2 : a.C, 0, (uint32_t)((size_t)a.M * a.ldc * 2), 0x00020000);
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t srdSlab = __builtin_amdgcn_make_buffer_rsrc(
      DG && a.colsum ? (void*)a.colsum : a.C, 0, (uint32_t)(DG && a.colsum ? (size_t)a.colsum_rows * a.N * 4 : 0), 0x00020000);
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t srdAux = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(DG ? a.aux : (const void*)a.C), 0, (uint32_t)(DG ? (size_t)a.M * a.ldaux * 2 : 0), 0x00020000);
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t srdBias = __builtin_amdgcn_make_buffer_rsrc(
      GELU ? (void*)const_cast<float*>(a.bias) : a.C, 0, (uint32_t)(GELU ? (size_t)a.N * 4 : 0), 0x00020000);
  const int rrow = lane >> 3, rcol = lane & 7;  // drain mapping: 8 rows x 8 lanes x 16 B
  // The 2-KB park of a wave is two 1-KB halves (rows 0-7 / rows 8-15 of a 16-row slice).  Once a chunk has read its half, that half
  // is free until the next slice is parked: the LDS-DMA of the NEXT chunk's pre-activation rows (GELU') lands there - lane L's 16
  // bytes at L * 16, read back by the same lane - and, behind a tile's last chunk, the 64 bias values of the tile being accumulated
  // (GELU), read at the tile boundary.  The park is written and read with INLINE-ASM LDS instructions: for a ds_write / ds_read it
  // can see, hipcc puts `s_waitcnt vmcnt(0)` in front whenever an LDS-DMA is in flight (it cannot prove they do not alias), which
  // drained the ring at every chunk.  LDS operations of one wave execute in order.
  const uint32_t park0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem) + (uint32_t)(PARK + wave * 2048);
  const uint32_t park_w = park0 + (uint32_t)(fr * 128 + (fq & 1) * 8);                      // + swizzled 16-B unit of column block j
  const uint32_t park_r = park0 + (uint32_t)(rrow * 128 + ((rcol ^ (rrow & 7)) << 4));      // rows 0-7 (+ 1024: rows 8-15, same swizzle)
  [[maybe_unused]] char* const park_ptr = smem + PARK + wave * 2048;
  int c_row0, c_col0;                            // tile being accumulated
  int o_row0 = a.M, o_col0 = 0;                  // tile being drained (none yet: every row out of range)
  tile_rc(0, c_row0, c_col0);

  f32x4 acc[4][4];
  uint2 old[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      old[i][j] = make_uint2(0u, 0u);
    }
  u32x4 rawv = {0u, 0u, 0u, 0u};                        // the chunk's 8 bf16 values per lane (stage 0 -> 2)
  [[maybe_unused]] u32x4 auxv = {0u, 0u, 0u, 0u};       // GELU': the pre-activation at the same positions
  [[maybe_unused]] uint32_t half_out[2] = {0u, 0u};     // stage 1's packed results (second output of GELU / output of GELU')
  [[maybe_unused]] float cs8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // ---- epilogue stages of chunk c (rows 16 (c >> 1) + 8 (c & 1) .. + 7 of this wave's 64 x 64 tile of the tile in `old`)
  auto stage0 = [&](const int c, const bool tail) {
    const int h = c & 1;
    if constexpr (DG) {
      // the pre-activation rows the previous chunk's stage 1 requested: behind that LDS-DMA only the 6 pieces of one K-tile and
      // the previous chunk's stores
      if (tail) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 + ES) : "memory");
      }
      // an even chunk's rows sit in the upper half (read BEFORE the slice overwrites it), an odd chunk's in the lower one
      asm volatile("ds_read_b128 %0, %1" : "=v"(auxv) : "v"(park0 + (uint32_t)((h ? 0 : 1024) + lane * 16)) : "memory");
    }
    if (h == 0) {
      // the parked slice is a compile-time register index: each case holds its own (volatile asm) stores, so the switch cannot be
      // turned into a run-time indexed array (which hipcc would move to scratch memory)
      auto park_slice = [&](const uint2(&sl)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {  // lane: row fr, columns 16 j + 4 fq .. + 3 -> 8 bytes at unit (2 j + (fq >> 1)) ^ (row & 7)
          const u32x2 v = {sl[j].x, sl[j].y};
          asm volatile("ds_write_b64 %0, %1" ::"v"(park_w + (uint32_t)((((2 * j + (fq >> 1)) ^ (fr & 7)) << 4))), "v"(v) : "memory");
        }
      };
      switch (c >> 1) {
        case 0: park_slice(old[0]); break;
        case 1: park_slice(old[1]); break;
        case 2: park_slice(old[2]); break;
        default: park_slice(old[3]); break;
      }
    }
    asm volatile("ds_read_b128 %0, %1" : "=v"(rawv) : "v"(park_r + (uint32_t)(h * 1024)) : "memory");
    // waited for HERE, inside the branch: a destination register of an inline-asm read must not cross a control-flow join while the
    // data is in flight (the compiler may copy it there).  ~one LDS round trip per chunk, before this segment's fragment reads.
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rawv), "+v"(auxv)::"memory");
  };
  auto stage1 = [&](const int c) {
    if constexpr (GELU) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const f32x2 gv = gelu_tanh2(unpack_bf16x2(rawv[k]));
        half_out[k] = pack_bf16x2(gv.x, gv.y);
      }
      if (c == 7 && lane < 16)
        // the bias of the tile being ACCUMULATED (64 columns of this wave = 16 lanes x 16 B) into the lower half, which nobody reads
        // again before the next tile's first slice is parked; consumed at the tile boundary
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srdBias, (__attribute__((address_space(3))) void*)park_ptr, 16,
                                                 (uint32_t)(c_col0 + wn * 64 + 4 * lane) * 4u, 0, 0, 0);
    } else if constexpr (DG) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const f32x2 dv = unpack_bf16x2(rawv[k]) * gelu_tanh_grad2(unpack_bf16x2(auxv[k]));
        half_out[k] = pack_bf16x2(dv.x, dv.y);
      }
      // the NEXT chunk's pre-activation rows (chunk 0 of the tile being accumulated behind this tile's last chunk) into the half
      // this chunk has finished with (its reads completed behind the last barrier)
      const int cn = (c + 1) & 7;
      const int nrow = (c == 7 ? c_row0 : o_row0) + wm * 64 + 16 * (cn >> 1) + 8 * (cn & 1) + rrow;
      const int ncol = (c == 7 ? c_col0 : o_col0) + wn * 64 + 8 * rcol;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srdAux, (__attribute__((address_space(3))) void*)(park_ptr + ((c & 1) ? 1024 : 0)), 16,
                                               (uint32_t)nrow * (uint32_t)(a.ldaux * 2) + (uint32_t)(ncol * 2), 0, 0, 0);
    }
  };
  auto stage2 = [&](const int c) {
    const int pr = 8 * (c & 1) + rrow;
    const int row = o_row0 + wm * 64 + 16 * (c >> 1) + pr, col = o_col0 + wn * 64 + 8 * rcol;
    const uint32_t voff = (uint32_t)row * (uint32_t)(a.ldc * 2) + (uint32_t)(col * 2);
    if constexpr (EPI == SAVIT_EPI_BF16) {
      __builtin_amdgcn_raw_buffer_store_b128(rawv, srdC, voff, 0, 0);
    } else if constexpr (GELU) {
      uint32_t gw[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const f32x2 gv = gelu_tanh2(unpack_bf16x2(rawv[2 + k]));
        gw[k] = pack_bf16x2(gv.x, gv.y);
      }
      __builtin_amdgcn_raw_buffer_store_b128(rawv, srdC, voff, 0, 2);  // nt: read again in backward only
      __builtin_amdgcn_raw_buffer_store_b128(u32x4{half_out[0], half_out[1], gw[0], gw[1]}, srdC2, voff, 0, 0);
    } else {
      uint32_t ow[4] = {half_out[0], half_out[1], 0u, 0u};
#pragma unroll
      for (int k = 2; k < 4; ++k) {
        const f32x2 dv = unpack_bf16x2(rawv[k]) * gelu_tanh_grad2(unpack_bf16x2(auxv[k]));
        ow[k] = pack_bf16x2(dv.x, dv.y);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x2 r = unpack_bf16x2(ow[k]);  // the column sum adds the bf16-rounded values, as the stored tensor holds them
        cs8[2 * k] += r.x;
        cs8[2 * k + 1] += r.y;
      }
      __builtin_amdgcn_raw_buffer_store_b128(u32x4{ow[0], ow[1], ow[2], ow[3]}, srdC, voff, 0, 0);
      if (c == 7) {
        // fold the 8 row groups (lanes with equal lane & 7 own the same 8 columns), then lanes 0-7 write this (row tile, wave row)'s
        // partial (other lanes and the "no tile yet" placeholder store out of range: dropped; the offsets must not wrap to 0)
        float s8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float v = cs8[k];
          v += __shfl_xor(v, 32, 64);
          v += __shfl_xor(v, 16, 64);
          v += __shfl_xor(v, 8, 64);
          s8[k] = v;
          cs8[k] = 0.f;
        }
        const bool live = lane < 8 && o_row0 < a.M;
        const uint32_t soff = live ? ((uint32_t)((o_row0 / BM) * 4 + wm) * (uint32_t)a.N + (uint32_t)col) * 4u : 0xffffff00u;
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(s8[0]), __float_as_uint(s8[1]), __float_as_uint(s8[2]), __float_as_uint(s8[3])},
                                               srdSlab, soff, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(s8[4]), __float_as_uint(s8[5]), __float_as_uint(s8[6]), __float_as_uint(s8[7])},
                                               srdSlab, soff + 16u, 0, 0);
      }
    }
  };

  // ---- prologue: K-tiles 0 and 1 in flight, K-tile 0 landed
  dma_set_tile(0);
  dma_ktile();
  dma_ktile();
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();  // group B runs one barrier behind group A

  bf16x8 af[4][2], bq[2][2];
#define STREAM_READ_A()                                                                              \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                  \
    af[i][0] = *reinterpret_cast<const bf16x8*>(cur + a_row + i * 16 * RB + foff0);                \
    af[i][1] = *reinterpret_cast<const bf16x8*>(cur + a_row + i * 16 * RB + foff1);                \
  }
#define STREAM_READ_B(Q)                                                                             \
  _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) {                                               \
    bq[jj][0] = *reinterpret_cast<const bf16x8*>(cur + b_row + (2 * (Q) + jj) * 16 * RB + foff0);  \
    bq[jj][1] = *reinterpret_cast<const bf16x8*>(cur + b_row + (2 * (Q) + jj) * 16 * RB + foff1);  \
  }
#define STREAM_COMPUTE(Q)                                                                            \
  __builtin_amdgcn_s_barrier();                                                                      \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                \
  __builtin_amdgcn_sched_barrier(0);                                                                 \
  __builtin_amdgcn_s_setprio(1);                                                                     \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                 \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                  \
      _Pragma("unroll") for (int jj = 0; jj < 2; ++jj)                                             \
        acc[i][2 * (Q) + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[jj][ks], af[i][ks], acc[i][2 * (Q) + jj], 0, 0, 0); \
  __builtin_amdgcn_s_setprio(0);                                                                     \
  __builtin_amdgcn_sched_barrier(0);                                                                 \
  __builtin_amdgcn_s_barrier();

  int slot = 0;  // ring slot of the K-tile being computed
  int t = 0;     // its index among this workgroup's K-tiles
  for (int k = 0; k < NT; ++k) {
    int c_next = 0;      // next chunk to start (stage 0) ...
    int pp_next = 0;     // ... in the L(.,0) of this K-tile of the tile
    int c_run = -1;      // chunk between stage 0 and stage 2
    for (int pp = 0; pp < KP; ++pp) {
      const char* cur = smem + slot * SLOT;
      const bool tail = t + 2 >= total;  // no K-tile left to request: the counts below do not apply, wait for everything
      // ---- L(t,0)
      int n_st = 0;  // stores issued in this segment
      if (c_run >= 0) {  // stage 2 of the chunk whose stages 0 and 1 ran in the previous K-tile
        stage2(c_run);
        n_st = (DG && c_run == 7) ? ES + 2 : ES;
        c_run = -1;
      }
      const bool start = (c_next < 8 && pp == pp_next);
      if (start) stage0(c_next, tail);
      STREAM_READ_B(0)
      __builtin_amdgcn_sched_barrier(0);
      STREAM_READ_A()
      STREAM_COMPUTE(0)
      // ---- L(t,1)
      STREAM_READ_B(1)
      int n_x = 0;  // the epilogue's own LDS-DMA in this segment
      if (start) {
        stage1(c_next);
        n_x = (DG || (GELU && c_next == 7)) ? 1 : 0;
        c_run = c_next;
        ++c_next;
        pp_next = (c_next * KP) >> 3;
      }
      if (!tail) dma_ktile();
      // K-tile t+1 must have landed; in flight may stay: this segment's 6 pieces and epilogue DMA, the stores of L(t,0)
      if (tail) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        switch (n_st + n_x) {
          case 0: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
          case 1: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
          case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
          case 3: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
          default: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        }
      }
      STREAM_COMPUTE(1)
      slot = (slot + 1 == ND) ? 0 : slot + 1;
      ++t;
    }
    // ---- tile boundary (every chunk of the previous tile has finished: its last stage 2 ran in L(.,0) of a K-tile of this tile):
    // scale / bias, ONE rounding to bf16 (where the reference's graph rounds), accumulators restart from zero
    [[maybe_unused]] u32x4 biasv[4];
    if constexpr (GELU) {
      // the bias the last chunk's stage 1 requested (behind it at least one K-tile's 6 pieces and that chunk's two stores)
      if (k + 1 == NT) {  // last tile: the ring's requests have run out, the count below does not apply
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 + ES) : "memory");
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(biasv[j]) : "v"(park0 + (uint32_t)((16 * j + 4 * fq) * 4)) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(biasv[0]), "+v"(biasv[1]), "+v"(biasv[2]), "+v"(biasv[3])::"memory");
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      [[maybe_unused]] float bb[4] = {0.f, 0.f, 0.f, 0.f};
      if constexpr (GELU) {
#pragma unroll
        for (int q = 0; q < 4; ++q) bb[q] = a.round_bias_bf16 ? round_bf16(__uint_as_float(biasv[j][q])) : __uint_as_float(biasv[j][q]);
      }
      const float sc = (EPI == SAVIT_EPI_BF16 && c_col0 + wn * 64 + 16 * j + 4 * fq < a.alpha_cols) ? a.alpha : 1.0f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4 v = acc[i][j];
        if constexpr (EPI == SAVIT_EPI_BF16) v = v * sc;
        old[i][j] = make_uint2(pack_bf16x2(v[0] + bb[0], v[1] + bb[1]), pack_bf16x2(v[2] + bb[2], v[3] + bb[3]));
        acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    o_row0 = c_row0;
    o_col0 = c_col0;
    if (k + 1 < NT) {
      tile_rc(k + 1, c_row0, c_col0);
    } else {
      c_row0 = a.M;  // nothing follows: the last chunk's look-ahead request falls out of range
    }
  }
#undef STREAM_READ_A
#undef STREAM_READ_B
#undef STREAM_COMPUTE
  if (grp == 0) __builtin_amdgcn_s_barrier();  // balance the barrier count
  // ---- drain the last tile (nothing left to overlap with)
  for (int c = 0; c < 8; ++c) {
    stage0(c, true);
    stage1(c);
    stage2(c);
  }
}

#ifdef SAVIT_EXPERIMENTS  // tile 30 is never selected (round 3: built, bitwise-correct, slower): experiment builds only
int launch_stream(const GemmParams& p0, hipStream_t s) {
  GemmParams p = p0;
  p.tiles_m = (p.a.M + 255) / 256;
  p.tiles_n = p.a.N / 128;
  p.row_group = p.tiles_m < 4 ? p.tiles_m : 4;  // 32 concurrent tiles per XCD = 4 A-panels x 8 W-panels (K = 768: 1.5 MB + 1.5 MB of L2)
  const int cus = device_cus();
  const int ntiles = p.tiles_m * p.tiles_n;
  int per = (cus + 7) / 8;  // workgroups per XCD: one per CU
  if (per > (ntiles + 7) / 8) per = (ntiles + 7) / 8;
  const dim3 grid(8 * per);
  const size_t lds = 3 * (256 + 128) * 128 + 8 * 2048;
#define SAVIT_LAUNCH_EPI(E)                                   \
  case E: {                                                   \
    auto kfn = gemm_tn_stream_kernel<E>;                      \
    SAVIT_LDS_ONCE(kfn);                                      \
    hipLaunchKernelGGL(kfn, grid, dim3(512), lds, s, p);      \
  } break;
  switch (p.a.epilogue) {
    SAVIT_LAUNCH_EPI(SAVIT_EPI_BF16)
    SAVIT_LAUNCH_EPI(SAVIT_EPI_BIAS_GELU)
    SAVIT_LAUNCH_EPI(SAVIT_EPI_DGELU)
    default: return SAVIT_EINVAL;
  }
#undef SAVIT_LAUNCH_EPI
  SAVIT_LAUNCH_RET();
}

// what the streaming kernel takes: its three epilogues in their hot-path forms, whole 128-column tiles, at least 9 K-tiles per tile
inline bool stream_ok(const savit_gemm_args& a) {
  if (a.K % 64 != 0 || a.K < 576 || a.N % 128 != 0 || a.lda < a.K || a.M < 1) return false;
  // 32-bit buffer offsets, also for the rows of a ragged last tile and of the "no tile yet" placeholder (row index < M + 512)
  if ((size_t)(a.M + 512) * a.ldc * 2 > 0xffffffe0ull || a.ldc % 8 != 0 || a.rowscale != nullptr || a.colscale != nullptr) return false;
  if (a.epilogue == SAVIT_EPI_BF16) return a.bias == nullptr;
  if (a.epilogue == SAVIT_EPI_BIAS_GELU) return a.bias != nullptr && a.C2 != nullptr;
  if (a.epilogue == SAVIT_EPI_DGELU) return a.aux != nullptr && a.ldaux % 8 == 0 && (size_t)(a.M + 512) * a.ldaux * 2 <= 0xffffffe0ull &&
                                            (a.colsum == nullptr || a.colsum_rows > 0);
  return false;
}
#endif  // SAVIT_EXPERIMENTS

// ------------------------------------------------------------------------------------------------------------
// Ping-pong variant with 320-row tiles (tile 21: 320 x 256 x 64, 8 waves = 2 (M) x 4 (N), wave tile 160 x 64).  Why 320: the N = 768
// products of DeiT-B (proj, fc2 and the three input-gradient GEMMs that end in d columns: 5 of the 8 TN GEMMs of a layer) have
// M = 25 216 rows: 256-row tiles are 99 x 3 = 297 workgroups - 1.16 rounds of the 256 CUs, so the 192 x 128 tiles ran instead
// (792 + tail) and paid for it in operand feed: (192 + 128) x 128 B per 4-wave K-tile step = 52 B per MFMA-pipe cycle and CU against
// 31 for 256 x 256 - and the L2 -> LDS path of a CU (about 70 GB/s), not the matrix pipe, is what bounds these kernels.
// 320-row tiles are 79 x 3 = 237 workgroups: ONE round at 93 % fill, no tail, and 72 KB per 2 560 pipe cycles = 28 B per cycle.
// Same structure and proofs as gemm_tn_pp_kernel; what changes:
//   LDS: 2 buffers x 72 KB; a buffer = 9 units of 64 rows: A0..A4 (rows 0-319) and B0..B3 (B(wc) = the Bt rows of N-quarter wc).
//   M-half h = rows 160 h .. +159, quadrant rows 80 (5 row blocks of 16): 20 MFMAs per phase.  Unit reads by phase: A0 p0 (rows 0-63:
//   M-half 0, first quadrant); A1 p0 + p2 (rows 64-79 / 80-127); A2 p2 (M-half 0) + p0 (rows 160-191: M-half 1); A3 p0 + p2; A4 p2;
//   B units p0 (columns 0-31 of the wave) + p1.
//   Load segments:  p0: read A(qm 0) [10 x ds_read_b128] + B(qn 0) [4]      | DMA A1, A2 of K-tile t+1
//                   p1: read B(qn 1) [4]                                      | DMA A3, A4 of K-tile t+1
//                   p2: read A(qm 1) [10]                                     | DMA A0 of K-tile t+2
//                   p3: -                                                     | DMA B0..B3 of K-tile t+2 ; s_waitcnt vmcnt(5)
//   WAR (a unit is overwritten two phases after its last read or later): A1..A4 of K-tile t+1 replace K-tile t-1's, last read in
//   (t-1, p2), written from (t, p0) on; A0 of t+2 replaces K-tile t's A0, read in (t, p0) only, written in (t, p2); B of t+2 replaces
//   K-tile t's, last read (t, p1), written (t, p3).  RAW: K-tile t+1 is complete behind the vmcnt(5) of (t, p3) - all but A0 and the
//   four B units of K-tile t+2 have landed - and the barrier after it; its first read is in (t+1, p0).
//   Epilogue: the wave's 160 x 64 tile is parked in the staging LDS in two 80-row halves (10 KB per wave each).
template <int EPI>
__global__ __launch_bounds__(512) void gemm_tn_pp320_kernel(const GemmParams p) {
  constexpr int BM = 320, BN = 256, RB = 128;
  constexpr int UNIT = 64 * RB, NUA = 5, BUF = 9 * UNIT;
  constexpr int WTM = 160, WTN = 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const savit_gemm_args& a = p.a;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int nwg = p.tiles_m * p.tiles_n;
  const int tid = xcd_remap(blockIdx.x, nwg);
  int tm, tn;
  {
    const int G = p.row_group, per = G * p.tiles_n;
    const int g = tid / per, rem = tid - g * per;
    const int rows_g = (p.tiles_m - g * G) < G ? (p.tiles_m - g * G) : G;
    tn = rem / rows_g;
    tm = g * G + (rem - tn * rows_g);
  }
  const int row0 = tm * BM, col0 = tn * BN;

  const bf16_t* Abase = reinterpret_cast<const bf16_t*>(a.A) + (size_t)row0 * a.lda;
  const bf16_t* Bbase = reinterpret_cast<const bf16_t*>(a.Bt) + (size_t)col0 * a.ldb;
  uint32_t a_bytes, b_bytes;
  {
    const size_t ta = (size_t)(a.M - row0) * a.lda * 2, tb = (size_t)(a.N - col0) * a.ldb * 2;
    a_bytes = (uint32_t)(ta > 0xfffffff0ull ? 0xfffffff0ull : ta);
    b_bytes = (uint32_t)(tb > 0xfffffff0ull ? 0xfffffff0ull : tb);
  }
  const auto srdA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Abase), 0, a_bytes, 0x00020000);
  const auto srdB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Bbase), 0, b_bytes, 0x00020000);

  // LDS-DMA geometry: one wave-instruction = 8 rows x 128 B; this wave owns rows wave*8 .. +7 of every unit
  const int lrow = lane >> 3, pch = lane & 7;
  const int ur = wave * 8 + lrow;
  const int uc = (pch ^ ((ur >> 1) & 7)) * 16;
  const uint32_t a_voff = (uint32_t)ur * (uint32_t)(a.lda * 2) + (uint32_t)uc;
  const uint32_t b_voff = (uint32_t)ur * (uint32_t)(a.ldb * 2) + (uint32_t)uc;
  const uint32_t a_unit = 64u * (uint32_t)(a.lda * 2), b_unit = 64u * (uint32_t)(a.ldb * 2);
  const int KT = a.K / 64;
  auto dma = [&](int kt, int u) {  // unit u (compile-time; 0-4 = A0..A4, 5-8 = B0..B3) of K-tile kt into buffer kt & 1
    if (kt >= KT) return;
    char* dst = smem + (kt & 1) * BUF + u * UNIT + wave * 1024;
    if (u < NUA)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srdA, (__attribute__((address_space(3))) void*)dst, 16, a_voff + (uint32_t)u * a_unit, kt * RB, 0, 0);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srdB, (__attribute__((address_space(3))) void*)dst, 16, b_voff + (uint32_t)(u - NUA) * b_unit, kt * RB, 0, 0);
  };

  // fragment read offsets: row = 16 i + (lane & 15) from a base that is a multiple of 16, logical chunk = 4 ks + (lane >> 4)
  const int fr = lane & 15, fq = lane >> 4;
  const int sw = (fr >> 1) & 7;
  const int foff0 = fr * RB + ((fq ^ sw) << 4);
  const int foff1 = fr * RB + (((fq + 4) ^ sw) << 4);
  const int a_base = (WTM * wr) * RB, b_base = NUA * UNIT + (WTN * wc) * RB;

  f32x4 acc[10][4];
#pragma unroll
  for (int i = 0; i < 10; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 af[5][2], bf[4][2];

  // ---- cache prefetch of the epilogue's fused operand (round 6).  The saved pre-activation tile of the GELU' epilogue (DGELU: 160 KB
  // per workgroup, written a whole forward + half a backward ago: an HBM read) is requested by the epilogue while every CU of a round
  // sits in its epilogue too - and HBM idles under the main loops (operands come out of L2 / the Infinity Cache).  Ten of the K-tiles
  // therefore carry ONE extra request per wave: a 4-byte LDS-DMA per lane into a 256-byte scratch slot nobody reads, lane l touching
  // cache line l of 16 lines of this wave's own epilogue footprint, so that the lines are in L2 / the Infinity Cache when the epilogue
  // asks for them.  The request is the YOUNGEST one in front of a K-tile's counted wait (vmcnt(6) instead of (5)): it is waited for one
  // K-tile later.  No arithmetic changes: results are bitwise the same.  Measured in the DeiT-B step, same box, A / B / A / B
  // (profiles/r06_aux_prefetch_ab.log): GELU' 156.3 / 157.0 -> 151.8 / 151.7 us.  The same for the fp32 residual tile of the RESID
  // epilogue (320 KB per workgroup, two lines per row) was built and is SLOWER (proj + residual 51.0 -> 52.5 us, fc2 + residual 108.5 ->
  // 110.5; with the non-temporal policy 54.9 / 114.0): that epilogue already moves its 154 MB at the HBM rate, and 10 MB of prefetched
  // lines per XCD push operand panels out of the 4 MB L2.  It stays in the code behind the launcher's switch (experiment builds).
  [[maybe_unused]] uint32_t pf_voff = 0x7ffffff0u, pf_step = 0;
  [[maybe_unused]] auto srdP = srdA;
  constexpr bool PF = (EPI == SAVIT_EPI_RESID || EPI == SAVIT_EPI_DGELU);
  constexpr int PF_N = 10;  // requests per wave: 160 rows = 10 x 16
  if constexpr (PF) {
    const int esz = (EPI == SAVIT_EPI_RESID) ? 4 : 2;
    const int prow = row0 + wr * WTM, pcol = col0 + wc * WTN;
    const char* pb = reinterpret_cast<const char*>(a.aux) + ((size_t)prow * a.ldaux + pcol) * esz;
    const long rows_left = (long)a.M - prow;
    const size_t tb = rows_left > 0 ? (size_t)rows_left * a.ldaux * esz : 0;
    srdP = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(pb), 0, (uint32_t)(tb > 0xfffffff0ull ? 0xfffffff0ull : tb), 0x00020000);
    if (EPI == SAVIT_EPI_RESID) {  // 64 fp32 columns = two 128-byte lines per row: lane -> (row lane >> 1 of 16, line lane & 1), lanes 0-31
      if (lane < 32 && pcol < a.N) pf_voff = (uint32_t)(lane >> 1) * (uint32_t)(a.ldaux * 4) + (uint32_t)(lane & 1) * 128u;
    } else {                       // 64 bf16 columns = one line per row: lane -> row lane of 16, lanes 0-15
      if (lane < 16 && pcol < a.N) pf_voff = (uint32_t)lane * (uint32_t)(a.ldaux * 2);
    }
    pf_step = 16u * (uint32_t)(a.ldaux * esz);
  }
  auto prefetch = [&](int kt) {
    if constexpr (PF) {
      const int j = kt - p.pf_start;
      const uint32_t off = (p.pf_start >= 0 && j >= 0 && j < PF_N) ? pf_voff + (uint32_t)j * pf_step : 0x7ffffff0u;
      char* dst = smem + 2 * BUF + wave * 256;
      if (p.pf_policy == 2)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srdP, (__attribute__((address_space(3))) void*)dst, 4, off, 0, 0, 2);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srdP, (__attribute__((address_space(3))) void*)dst, 4, off, 0, 0, 0);
    }
  };

  // prologue: all of K-tile 0 and the five units of K-tile 1 that L(-1,p2), L(-1,p3) would have issued
#pragma unroll
  for (int u = 0; u < 9; ++u) dma(0, u);
  dma(1, 0); dma(1, 5); dma(1, 6); dma(1, 7); dma(1, 8);
  if (KT > 1) {
    asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();  // M-half 1 runs one barrier behind M-half 0

#define PP5_READ_A(QM)                                                                                         \
  _Pragma("unroll") for (int i = 0; i < 5; ++i) {                                                              \
    af[i][0] = *reinterpret_cast<const bf16x8*>(cur + a_base + (80 * (QM) + 16 * i) * RB + foff0);             \
    af[i][1] = *reinterpret_cast<const bf16x8*>(cur + a_base + (80 * (QM) + 16 * i) * RB + foff1);             \
  }
#define PP5_READ_B(QN)                                                                                         \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                              \
    bf[2 * (QN) + j][0] = *reinterpret_cast<const bf16x8*>(cur + b_base + (2 * (QN) + j) * 16 * RB + foff0);  \
    bf[2 * (QN) + j][1] = *reinterpret_cast<const bf16x8*>(cur + b_base + (2 * (QN) + j) * 16 * RB + foff1);  \
  }
#define PP5_COMPUTE(QM, QN)                                                                                    \
  __builtin_amdgcn_s_barrier();                                                                                \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                           \
  __builtin_amdgcn_sched_barrier(0);                                                                           \
  __builtin_amdgcn_s_setprio(1);                                                                               \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                             \
    _Pragma("unroll") for (int i = 0; i < 5; ++i)                                                              \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                            \
        acc[5 * (QM) + i][2 * (QN) + j] =                                                                      \
            __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[2 * (QN) + j][ks], af[i][ks], acc[5 * (QM) + i][2 * (QN) + j], 0, 0, 0); \
  __builtin_amdgcn_s_setprio(0);                                                                               \
  __builtin_amdgcn_sched_barrier(0);                                                                           \
  __builtin_amdgcn_s_barrier();

  for (int kt = 0; kt < KT; ++kt) {
    const char* cur = smem + (kt & 1) * BUF;
    // ---- phase 0: quadrant (0,0)
    PP5_READ_B(0)
    __builtin_amdgcn_sched_barrier(0);
    PP5_READ_A(0)
    dma(kt + 1, 1); dma(kt + 1, 2);
    PP5_COMPUTE(0, 0)
    // ---- phase 1: quadrant (0,1)
    PP5_READ_B(1)
    dma(kt + 1, 3); dma(kt + 1, 4);
    PP5_COMPUTE(0, 1)
    // ---- phase 2: quadrant (1,1)
    PP5_READ_A(1)
    dma(kt + 2, 0);
    PP5_COMPUTE(1, 1)
    // ---- phase 3: quadrant (1,0); K-tile kt+1 must be complete behind this phase's first barrier
    dma(kt + 2, 5); dma(kt + 2, 6); dma(kt + 2, 7); dma(kt + 2, 8);
    prefetch(kt);
    if (kt + 2 < KT) {
      if constexpr (PF) {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      }
    } else {
      if constexpr (PF) {  // (kt + 2 >= KT: the five dma() above were no-ops; only the prefetch request may stay in flight)
        asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    PP5_COMPUTE(1, 0)
  }
#undef PP5_READ_A
#undef PP5_READ_B
#undef PP5_COMPUTE
  if (wr == 0) __builtin_amdgcn_s_barrier();  // balance the barrier count: every wave is past its last LDS read after this one

  if constexpr (epi_uses_lds<EPI>()) {
    // two 80-row halves through this wave's 10-KB slice of the (now idle) staging LDS
    epilogue_lds<EPI, 80, WTN, 5, 4>(p, reinterpret_cast<f32x4(&)[5][4]>(acc[0]), smem + wave * (80 * WTN * 2), row0 + wr * WTM, col0 + wc * WTN, lane,
                                     (tm * 2 + wr) * 2);
    epilogue_lds<EPI, 80, WTN, 5, 4>(p, reinterpret_cast<f32x4(&)[5][4]>(acc[5]), smem + wave * (80 * WTN * 2), row0 + wr * WTM + 80, col0 + wc * WTN,
                                     lane, (tm * 2 + wr) * 2 + 1);
  } else {
    const int mrow = row0 + wr * WTM + fr;
    const int ncol = col0 + wc * WTN + fq * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float csum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 10; ++i) epilogue_store<EPI>(p, mrow + i * 16, ncol + j * 16, acc[i][j], csum);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// PERSISTENT form of the 320 x 256 ping-pong tile (tile 22, round 5) for launches of more than one round of tiles: one workgroup per
// CU walks tiles b, b + G, b + 2 G ... and the K-tile stream does not stop at a tile boundary - the next tile's K-tile 0 is requested
// by the LAST K-tile's load segments (the same (kt + 1) / (kt + 2) requests the steady state issues, with the next tile's buffer
// descriptors), so it lands under the last MFMA phases and the epilogue's stores.  In-kernel stamps of the one-tile kernel
// (profiles/r04_tile_stamps.log) put a tile's prologue at 2.5 us - all nine units of K-tile 0 requested at once by a CU that has
// nothing else to do - next to 1.8 us per K-tile: on DeiT-B's three-to-four-round launches (qkv 711 tiles, fc1 / GELU' 948) that is
// 5-7 us per launch, on ViT-L's 7..29-round launches 2-7 % of every product.
//   LDS: buffer 0 at [0, 72 KB), 16 KB spare, buffer 1 at [88 KB, 160 KB): the whole CU.  K-tile number v of the workgroup's stream
//   (v counts across tiles) lives in buffer v & 1, so an odd K-tile count per tile alternates by itself.
//   Boundary: the last K-tile of a tile (buffer L) issues in p0 / p1 the units A1..A4 of the NEXT tile's K-tile 0 (buffer O = 1 - L:
//   the same WAR argument as inside a tile: O's previous contents were last read one K-tile earlier), its p2 / p3 requests (A0 and B of
//   K-tile 1, which would go to L) are DEFERRED behind the epilogue, because the epilogue parks the accumulators in L: 80 KB = buffer L
//   and the 8 KB of spare next to it.  Nothing that is in flight during the epilogue targets that region (K-tile 0 of the next tile
//   is complete in O or landing there).  Behind the epilogue: barrier (every wave is done with the park), the five deferred requests,
//   vmcnt(5) (everything older - all of K-tile 0 - has landed), barrier, and the half-barrier stagger of the two M-halves as at a
//   kernel start.  Same K order per accumulator, same epilogue code: results are bitwise those of tile 21.
template <int EPI>
__global__ __launch_bounds__(512) void gemm_tn_pp320p_kernel(const GemmParams p) {
  constexpr int BM = 320, BN = 256, RB = 128;
  constexpr int UNIT = 64 * RB, NUA = 5, BUF = 9 * UNIT, BUF1 = BUF + 16 * 1024;  // offset of buffer 1
  constexpr int WTM = 160, WTN = 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const savit_gemm_args& a = p.a;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int nwg = p.tiles_m * p.tiles_n;
  const int G = gridDim.x;

  const uint32_t a_unit = 64u * (uint32_t)(a.lda * 2), b_unit = 64u * (uint32_t)(a.ldb * 2);
  const int KT = a.K / 64;

  // tile number `it` of this workgroup -> (row tile, column tile).  As in the one-tile kernel's grid (xcd_remap) each XCD - the
  // workgroups b, b + 8, ... - owns ONE contiguous run of the row-grouped tile order for the whole launch and its workgroups walk
  // that run together (workgroup j of the XCD takes its tiles j, j + nx, j + 2 nx, ...), so the A row panels and the W column panels
  // an XCD's L2 holds keep being reused from round to round.  (First build: every round re-dealt all tiles over the XCDs - each XCD
  // jumped to another row group per round and the persistent launches ran 5-10 us SLOWER than one tile per workgroup.)
  const int nxcd = G < 8 ? G : 8;                           // (fewer than 8 workgroups: a test grid)
  const int xcd = blockIdx.x % nxcd, xj = blockIdx.x / nxcd;
  const int nx = (G - xcd + nxcd - 1) / nxcd;               // workgroups of this XCD
  const int xq = nwg / nxcd, xr = nwg % nxcd;
  const int xbase = xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq;
  const int xcount = xq + (xcd < xr ? 1 : 0);               // tiles of this XCD
  auto tile_of = [&](int it, int& tm, int& tn) -> bool {
    const int k = it * nx + xj;
    if (k >= xcount) return false;
    const int tid = xbase + k;
    const int Gr = p.row_group, per = Gr * p.tiles_n;
    const int g = tid / per, rem = tid - g * per;
    const int rows_g = (p.tiles_m - g * Gr) < Gr ? (p.tiles_m - g * Gr) : Gr;
    tn = rem / rows_g;
    tm = g * Gr + (rem - tn * rows_g);
    return true;
  };
#define PP5_MAKE_SRD(TM, TN, SA, SB)                                                                                              \
  do {                                                                                                                            \
    const int row0_ = (TM) * BM, col0_ = (TN) * BN;                                                                               \
    const bf16_t* Ab_ = reinterpret_cast<const bf16_t*>(a.A) + (size_t)row0_ * a.lda;                                            \
    const bf16_t* Bb_ = reinterpret_cast<const bf16_t*>(a.Bt) + (size_t)col0_ * a.ldb;                                           \
    const size_t ta_ = (size_t)(a.M - row0_) * a.lda * 2, tb_ = (size_t)(a.N - col0_) * a.ldb * 2;                               \
    SA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Ab_), 0, (uint32_t)(ta_ > 0xfffffff0ull ? 0xfffffff0ull : ta_), 0x00020000); \
    SB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Bb_), 0, (uint32_t)(tb_ > 0xfffffff0ull ? 0xfffffff0ull : tb_), 0x00020000); \
  } while (0)

  int tm, tn;
  if (!tile_of(0, tm, tn)) return;
  auto srdA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(reinterpret_cast<const bf16_t*>(a.A)), 0, 0, 0x00020000);
  auto srdB = srdA, nsrdA = srdA, nsrdB = srdA;
  PP5_MAKE_SRD(tm, tn, srdA, srdB);

  f32x4 acc[10][4];
  bf16x8 af[5][2], bf[4][2];
  int v = 0;  // K-tiles this workgroup has consumed: K-tile kt of the current tile sits in buffer (v + kt) & 1

#define PP5_READ_A(QM)                                                                                         \
  _Pragma("unroll") for (int i = 0; i < 5; ++i) {                                                              \
    af[i][0] = *reinterpret_cast<const bf16x8*>(cur + a_base + (80 * (QM) + 16 * i) * RB + foff0);             \
    af[i][1] = *reinterpret_cast<const bf16x8*>(cur + a_base + (80 * (QM) + 16 * i) * RB + foff1);             \
  }
#define PP5_READ_B(QN)                                                                                         \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                              \
    bf[2 * (QN) + j][0] = *reinterpret_cast<const bf16x8*>(cur + b_base + (2 * (QN) + j) * 16 * RB + foff0);  \
    bf[2 * (QN) + j][1] = *reinterpret_cast<const bf16x8*>(cur + b_base + (2 * (QN) + j) * 16 * RB + foff1);  \
  }
#define PP5_COMPUTE(QM, QN)                                                                                    \
  __builtin_amdgcn_s_barrier();                                                                                \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                           \
  __builtin_amdgcn_sched_barrier(0);                                                                           \
  __builtin_amdgcn_s_setprio(1);                                                                               \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                             \
    _Pragma("unroll") for (int i = 0; i < 5; ++i)                                                              \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                            \
        acc[5 * (QM) + i][2 * (QN) + j] =                                                                      \
            __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[2 * (QN) + j][ks], af[i][ks], acc[5 * (QM) + i][2 * (QN) + j], 0, 0, 0); \
  __builtin_amdgcn_s_setprio(0);                                                                               \
  __builtin_amdgcn_sched_barrier(0);                                                                           \
  __builtin_amdgcn_s_barrier();
  // One K-tile (four phases).  MODE 0: inside the tile (requests for K-tiles kt + 1 and kt + 2 of this tile).  MODE 1: the second to
  // last K-tile (kt + 2 is the NEXT tile's K-tile 0).  MODE 2: the last K-tile (kt + 1 is the next tile's K-tile 0; the kt + 2 requests
  // are deferred behind the epilogue; no wait: the next K-tile is waited for there).  The boundary K-tiles are branch-free: a
  // workgroup's LAST tile "prefetches" its own K-tile 0 again (valid addresses, a free buffer, 72 KB nobody reads) - with a uniform
  // branch around the requests hipcc renamed accumulators across the merge and spilled one, and the reload of a spill waits for
  // every LDS-DMA issued before it (vmcnt counts in order): the prefetch would land before it could overlap anything.
#define PP5_KTILE(MODE)                                                                                        \
  {                                                                                                            \
    const char* cur = smem + (((v + kt) & 1) ? BUF1 : 0);                                                      \
    const int pn = (v + kt + 1) & 1;                                                                           \
    PP5_READ_B(0)                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    PP5_READ_A(0)                                                                                              \
    if (MODE < 2) { dma_to(false, pn, kt + 1, 1); dma_to(false, pn, kt + 1, 2); }                              \
    else { dma_to(true, pn, 0, 1); dma_to(true, pn, 0, 2); }                                     \
    PP5_COMPUTE(0, 0)                                                                                          \
    PP5_READ_B(1)                                                                                              \
    if (MODE < 2) { dma_to(false, pn, kt + 1, 3); dma_to(false, pn, kt + 1, 4); }                              \
    else { dma_to(true, pn, 0, 3); dma_to(true, pn, 0, 4); }                                     \
    PP5_COMPUTE(0, 1)                                                                                          \
    PP5_READ_A(1)                                                                                              \
    if (MODE == 0) dma_to(false, pn ^ 1, kt + 2, 0);                                                           \
    else if (MODE == 1) dma_to(true, pn ^ 1, 0, 0);                                                \
    PP5_COMPUTE(1, 1)                                                                                          \
    if (MODE == 0) {                                                                                           \
      dma_to(false, pn ^ 1, kt + 2, 5); dma_to(false, pn ^ 1, kt + 2, 6); dma_to(false, pn ^ 1, kt + 2, 7);    \
      dma_to(false, pn ^ 1, kt + 2, 8);                                                                        \
      asm volatile("s_waitcnt vmcnt(5)" ::: "memory");                                                         \
    } else if (MODE == 1) {                                                                                    \
      dma_to(true, pn ^ 1, 0, 5); dma_to(true, pn ^ 1, 0, 6); dma_to(true, pn ^ 1, 0, 7); dma_to(true, pn ^ 1, 0, 8); \
      asm volatile("s_waitcnt vmcnt(5)" ::: "memory");                                                         \
    }                                                                                                          \
    PP5_COMPUTE(1, 0)                                                                                          \
  }

  for (int it = 0;; ++it) {
    int ntm = 0, ntn = 0;
    const bool has_next = tile_of(it + 1, ntm, ntn);
    if (!has_next) { ntm = tm; ntn = tn; }
    PP5_MAKE_SRD(ntm, ntn, nsrdA, nsrdB);
    // Everything derived from the lane id is recomputed per tile behind an opaque copy: hoisted out of this loop it would stay live
    // across the epilogue, whose fused-operand loads (80 registers of fp32 residual rows) already fill the register file - the
    // one-tile kernel gets those registers back for free because its loop state is dead by then (first build: 73 spilled VGPRs).
    int ln = lane;
    asm volatile("" : "+v"(ln));
    // LDS-DMA geometry: one wave-instruction = 8 rows x 128 B; this wave owns rows wave*8 .. +7 of every unit
    const int lrow = ln >> 3, pch = ln & 7;
    const int ur = wave * 8 + lrow;
    const int uc = (pch ^ ((ur >> 1) & 7)) * 16;
    const uint32_t a_voff = (uint32_t)ur * (uint32_t)(a.lda * 2) + (uint32_t)uc;
    const uint32_t b_voff = (uint32_t)ur * (uint32_t)(a.ldb * 2) + (uint32_t)uc;
    auto dma_to = [&](bool next, int par, int kt, int u) {  // unit u (0-4 = A0..A4, 5-8 = B0..B3) of K-tile kt into buffer `par`; next: the NEXT tile's descriptors
      char* dst = smem + (par ? BUF1 : 0) + u * UNIT + wave * 1024;
      if (u < NUA) {
        if (next)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(nsrdA, (__attribute__((address_space(3))) void*)dst, 16, a_voff + (uint32_t)u * a_unit, kt * RB, 0, 0);
        else
          __builtin_amdgcn_raw_ptr_buffer_load_lds(srdA, (__attribute__((address_space(3))) void*)dst, 16, a_voff + (uint32_t)u * a_unit, kt * RB, 0, 0);
      } else {
        if (next)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(nsrdB, (__attribute__((address_space(3))) void*)dst, 16, b_voff + (uint32_t)(u - NUA) * b_unit, kt * RB, 0, 0);
        else
          __builtin_amdgcn_raw_ptr_buffer_load_lds(srdB, (__attribute__((address_space(3))) void*)dst, 16, b_voff + (uint32_t)(u - NUA) * b_unit, kt * RB, 0, 0);
      }
    };
    const int fr = ln & 15, fq = ln >> 4;
    const int sw = (fr >> 1) & 7;
    const int foff0 = fr * RB + ((fq ^ sw) << 4);
    const int foff1 = fr * RB + (((fq + 4) ^ sw) << 4);
    const int a_base = (WTM * wr) * RB, b_base = NUA * UNIT + (WTN * wc) * RB;
    if (it == 0) {
      // prologue of the FIRST tile: all of its K-tile 0 and the five units of K-tile 1 that (-1, p2), (-1, p3) would have issued
#pragma unroll
      for (int u = 0; u < 9; ++u) dma_to(false, 0, 0, u);
      dma_to(false, 1, 1, 0);
      dma_to(false, 1, 1, 5); dma_to(false, 1, 1, 6); dma_to(false, 1, 1, 7); dma_to(false, 1, 1, 8);
      asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (wr == 1) __builtin_amdgcn_s_barrier();  // M-half 1 runs one barrier behind M-half 0
    }
#pragma unroll
    for (int i = 0; i < 10; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    {
      int kt = 0;
      for (; kt < KT - 2; ++kt) PP5_KTILE(0)
      PP5_KTILE(1)
      ++kt;
      PP5_KTILE(2)
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();  // balance the barrier count: every wave is past its last LDS read of this tile
    // The next tile's K-tile 0 (requested one to two phases ago) is waited for HERE, in front of the epilogue's stores: vmcnt counts
    // in order, so a wait placed behind the epilogue - next to the deferred requests - would also wait for every store of this tile
    // to be acknowledged before the next tile may start (first build: each persistent launch 4-7 us SLOWER than one tile per workgroup).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // the accumulators are parked in the buffer that held the last K-tile (+ 8 KB of the spare next to it): 80 KB
    const int last = (v + KT - 1) & 1;
    char* park = smem + (last ? (BUF1 - 8 * 1024) : 0) + wave * (80 * WTN * 2);
    const int row0 = tm * BM, col0 = tn * BN;
    int le = lane;  // (opaque again: the epilogue's per-lane address arithmetic must not be hoisted out of the tile loop either)
    asm volatile("" : "+v"(le));
    if constexpr (epi_uses_lds<EPI>()) {
      epilogue_lds<EPI, 80, WTN, 5, 4>(p, reinterpret_cast<f32x4(&)[5][4]>(acc[0]), park, row0 + wr * WTM, col0 + wc * WTN, le, (tm * 2 + wr) * 2);
      epilogue_lds<EPI, 80, WTN, 5, 4>(p, reinterpret_cast<f32x4(&)[5][4]>(acc[5]), park, row0 + wr * WTM + 80, col0 + wc * WTN, le,
                                       (tm * 2 + wr) * 2 + 1);
    } else {
      const int mrow = row0 + wr * WTM + (le & 15);
      const int ncol = col0 + wc * WTN + (le >> 4) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float csum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 10; ++i) epilogue_store<EPI>(p, mrow + i * 16, ncol + j * 16, acc[i][j], csum);
      }
    }
    if (!has_next) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the dummy prefetch of the last tile)
      break;
    }
    // ---- next tile: its K-tile 0 is in (or on its way to) the other buffer; the deferred requests of K-tile 1 go where the park was
    v += KT;
    tm = ntm; tn = ntn;
    srdA = nsrdA; srdB = nsrdB;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave has read its park back; every wave's share of the next K-tile 0 had landed before its epilogue
    dma_to(false, (v + 1) & 1, 1, 0);  // (K >= 128: there is a K-tile 1)
    dma_to(false, (v + 1) & 1, 1, 5); dma_to(false, (v + 1) & 1, 1, 6); dma_to(false, (v + 1) & 1, 1, 7);
    dma_to(false, (v + 1) & 1, 1, 8);
    // no wait: K-tile 0 is complete (above), these five are what the steady state leaves in flight behind its vmcnt(5) - and the
    // epilogue's stores drain under the next tile's first K-tile
    if (wr == 1) __builtin_amdgcn_s_barrier();
  }
#undef PP5_READ_A
#undef PP5_READ_B
#undef PP5_COMPUTE
#undef PP5_KTILE
#undef PP5_MAKE_SRD
}

// ------------------------------------------------------------------------------------------------------------
// Few-row products (tile 24, round 5): M of a few hundred rows at most - the B cls rows a ViT's LAST encoder layer and its head work on
// (vit.py:57,95), CaiT's class-attention layers (cait.py:96-122) - where a 128-row LDS tile is 1-6 workgroups, each walking all of K
// behind one barrier per K-tile (DeiT-B's fc2 on 128 rows, K = 3 072: 53 us for 0.6 GFLOP).  These products are latency problems: what
// counts is how many loads the chip has in flight.  No LDS, no barrier: ONE WAVE = one 16 x 16 output tile over all of K, both
// fragments straight from global memory in the MFMA operand layout (both operands are K-contiguous: a lane's 8 consecutive k are ONE
// 16-byte buffer load; rows >= M, columns >= N and k-steps >= K / 32 come back as zeros from the range check / an out-of-range
// offset: ragged shapes and the aliased-row operand form need no branches), 16 k-steps (32 loads) per stage, the next stage requested in
// front of the current stage's MFMAs: 64 loads in flight per wave, 384 waves for DeiT-B's 128 x 768 products.  One accumulator per output
// tile walks K in ascending 32-steps with the operand order of the LDS tiles: BITWISE their results (a first form - eight waves of a
// workgroup splitting K and meeting in LDS - was as fast and gave up that invariant: the data-parallel tests compare a half batch on
// this kernel with a whole batch on the LDS tiles).  grid = (N / 16, M / 16) ONE-WAVE workgroups: 2 x 2 waves per workgroup share rows in
// the L1 but use 96 of the 256 CUs (fc2 26.6 us against 15.0).  Measured in the DeiT-B step, 128 rows (tools/profile_step.py with
// SAVIT_PROFILE_LAYER=11; LDS tile 12 -> this kernel): fc2 (K = 3 072) 52.4 -> 15.3 us, fc1 input gradient 49.3 -> 15.0, proj 16.2 -> 6.8,
// fc1 + GELU (N = 3 072: 32 x 32 per wave) 16.9 -> 10.2, proj input gradient 14.9 -> 6.7, head 16.9 -> 9.1, head input gradient 18.2 -> 7.6.
template <int EPI, int T>  // T x T sub-tiles of 16 x 16 per wave: T = 1 where the grid would otherwise be short of waves, T = 2 for wide outputs
__global__ __launch_bounds__(64) void gemm_tn_rows_kernel(const GemmParams p) {
  const savit_gemm_args& a = p.a;
  const int lane = threadIdx.x & 63;
  const int fr = lane & 15, fq = lane >> 4;
  const int row0 = blockIdx.y * (16 * T), col0 = blockIdx.x * (16 * T);
  constexpr uint32_t OOB = 0x7ffffff0u;
  if (row0 >= a.M || col0 >= a.N) return;
  // descriptors start at THIS wave's first row / column and end with the matrix (capped below 2 GB): any pitch works - the cls rows of a
  // dense [B * N, F] activation are B rows 6-8 MB apart - and the aliased-row form (lda < K) still reads zeros behind the last row
  const size_t ta = (size_t)(a.M - row0) * a.lda * 2, tb = (size_t)(a.N - col0) * a.ldb * 2;
  const auto srdA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(reinterpret_cast<const bf16_t*>(a.A) + (size_t)row0 * a.lda), 0,
                                                      (uint32_t)(ta > OOB ? OOB : ta), 0x00020000);
  const auto srdB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(reinterpret_cast<const bf16_t*>(a.Bt) + (size_t)col0 * a.ldb), 0,
                                                      (uint32_t)(tb > OOB ? OOB : tb), 0x00020000);
  uint32_t aoff[T], boff[T];
#pragma unroll
  for (int i = 0; i < T; ++i) {
    const int r = 16 * i + fr;
    aoff[i] = row0 + r < a.M ? (uint32_t)r * (uint32_t)(a.lda * 2) + (uint32_t)fq * 16u : OOB;
    boff[i] = col0 + r < a.N ? (uint32_t)r * (uint32_t)(a.ldb * 2) + (uint32_t)fq * 16u : OOB;
  }
  const int steps = a.K / 32;  // K % 32 == 0
  auto frag = [&](__amdgpu_buffer_rsrc_t srd, uint32_t off, int step) -> bf16x8 {
    union { u32x4 u; bf16x8 v; } x;
    x.u = __builtin_amdgcn_raw_buffer_load_b128(srd, step < steps ? off + (uint32_t)step * 64u : OOB, 0, 0);
    return x.v;
  };
  f32x4 acc[T][T];
#pragma unroll
  for (int i = 0; i < T; ++i)
#pragma unroll
    for (int j = 0; j < T; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int U = 16 / T;  // k-steps per stage: 32 loads
  bf16x8 af[2][U][T], bf[2][U][T];
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int i = 0; i < T; ++i) {
      af[0][u][i] = frag(srdA, aoff[i], u);
      bf[0][u][i] = frag(srdB, boff[i], u);
    }
  for (int s0 = 0; s0 < steps; s0 += 2 * U) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {  // stage `half` computes, stage `half ^ 1` receives the steps behind it
      const int base = s0 + half * U;
      if (base >= steps) break;
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int i = 0; i < T; ++i) {
          af[half ^ 1][u][i] = frag(srdA, aoff[i], base + U + u);
          bf[half ^ 1][u][i] = frag(srdB, boff[i], base + U + u);
        }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int i = 0; i < T; ++i)
#pragma unroll
          for (int j = 0; j < T; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[half][u][j], af[half][u][i], acc[i][j], 0, 0, 0);
    }
  }
  float csum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < T; ++i)
#pragma unroll
    for (int j = 0; j < T; ++j) epilogue_store<EPI>(p, row0 + 16 * i + fr, col0 + 16 * j + fq * 4, acc[i][j], csum);
}

int launch_rows(const GemmParams& p0, hipStream_t s) {
  GemmParams p = p0;
  if ((size_t)31 * p.a.lda * 2 + (size_t)p.a.K * 2 >= 0x7ffffff0ull || (size_t)31 * p.a.ldb * 2 + (size_t)p.a.K * 2 >= 0x7ffffff0ull) return SAVIT_EINVAL;  // (a pitch of 34 M elements)
  // 32 x 32 per wave (half the L2 -> CU bytes per flop) where that still leaves waves for every CU; else 16 x 16
  const long w16 = (long)((p.a.M + 15) / 16) * ((p.a.N + 15) / 16);
  const int T = w16 >= 1536 ? 2 : 1;
  p.tiles_m = (p.a.M + 16 * T - 1) / (16 * T);
  p.tiles_n = (p.a.N + 16 * T - 1) / (16 * T);
  if (p.tiles_m > 65535) return SAVIT_EINVAL;
  const dim3 grid(p.tiles_n, p.tiles_m);
#define SAVIT_LAUNCH_EPI(E)                                                                             \
  case E:                                                                                               \
    if (T == 2) hipLaunchKernelGGL((gemm_tn_rows_kernel<E, 2>), grid, dim3(64), 0, s, p);               \
    else hipLaunchKernelGGL((gemm_tn_rows_kernel<E, 1>), grid, dim3(64), 0, s, p);                      \
    break;
  switch (p.a.epilogue) {
    SAVIT_LAUNCH_EPI(SAVIT_EPI_BF16)
    SAVIT_LAUNCH_EPI(SAVIT_EPI_BIAS_GELU)
    SAVIT_LAUNCH_EPI(SAVIT_EPI_RESID)
    SAVIT_LAUNCH_EPI(SAVIT_EPI_F32)
    default: return SAVIT_EINVAL;  // (GELU' with its column-sum slab and the patch gather stay with the LDS tiles)
  }
#undef SAVIT_LAUNCH_EPI
  SAVIT_LAUNCH_RET();
}

int launch_pp320(const GemmParams& p0, hipStream_t s, bool pers) {
  GemmParams p = p0;
  p.tiles_m = (p.a.M + 319) / 320;
  p.tiles_n = (p.a.N + 255) / 256;
  // (row panels per group: as for the 256-row kernel; see pp_row_group below)
  int g = (int)((3l << 20) / ((long)320 * p.a.K * 2));
  if (g < 1) g = 1;
  if (g > 8) g = 8;
  if (p.a.epilogue == SAVIT_EPI_DGELU && g > 4) g = 4;
  {
    static const int force_g = SAVIT_EXP_ENV_INT("SAVIT_PP320_ROW_GROUP", 0);  // experiment builds: the tile order's row-panel group
    if (force_g > 0) g = force_g;
  }
  if (g > p.tiles_m) g = p.tiles_m;
  p.row_group = g;
  const int tiles = p.tiles_m * p.tiles_n;
  const int cus = (p.a.cu_budget > 0 && p.a.cu_budget < device_cus()) ? p.a.cu_budget : device_cus();
  // more than one round of tiles: the persistent form (tile 22), one workgroup per CU; a single round: one tile per workgroup (tile 21)
  const bool persistent = pers && tiles > cus && p.a.K >= 128;  // (the kernel peels the last two K-tiles of a tile)
  const dim3 grid(persistent ? cus : tiles);
  const bool pf_epi = p.a.epilogue == SAVIT_EPI_RESID || p.a.epilogue == SAVIT_EPI_DGELU;
  const size_t lds = persistent ? 160 * 1024 : 2 * 9 * 64 * 128 + (pf_epi ? 8 * 256 : 0);
  {
    // fused-operand cache prefetch (one-tile kernel): ten K-tiles ending two K-tiles before the last; off for short products
    static const int pf_mode = SAVIT_EXP_ENV_INT("SAVIT_PP_PREFETCH", 1), pf_lead = SAVIT_EXP_ENV_INT("SAVIT_PP_PREFETCH_LEAD", 2),
                     pf_pol = SAVIT_EXP_ENV_INT("SAVIT_PP_PREFETCH_POLICY", 0);
    const int KT = p.a.K / 64;
    const bool pf_on = pf_mode == 2 ? pf_epi : (pf_mode == 1 && p.a.epilogue == SAVIT_EPI_DGELU);  // (2: the RESID epilogue too - measured slower)
    p.pf_start = (pf_on && p.a.aux != nullptr && KT >= 6) ? (KT - 10 - pf_lead > 0 ? KT - 10 - pf_lead : 0) : -1;
    p.pf_policy = pf_pol;
  }
#define SAVIT_LAUNCH_EPI(E)                                                                            \
  case E: {                                                                                            \
    if (persistent) {                                                                                  \
      auto kfn = gemm_tn_pp320p_kernel<E>;                                                             \
      SAVIT_LDS_ONCE(kfn);                                                                             \
      hipLaunchKernelGGL(kfn, grid, dim3(512), lds, s, p);                                             \
    } else {                                                                                           \
      auto kfn = gemm_tn_pp320_kernel<E>;                                                              \
      SAVIT_LDS_ONCE(kfn);                                                                             \
      hipLaunchKernelGGL(kfn, grid, dim3(512), lds, s, p);                                             \
    }                                                                                                  \
  } break;
  switch (p.a.epilogue) {
    SAVIT_LAUNCH_EPI(SAVIT_EPI_BF16)
    SAVIT_LAUNCH_EPI(SAVIT_EPI_BIAS_GELU)
    SAVIT_LAUNCH_EPI(SAVIT_EPI_RESID)
    SAVIT_LAUNCH_EPI(SAVIT_EPI_DGELU)
    SAVIT_LAUNCH_EPI(SAVIT_EPI_F32)
    default: return SAVIT_EINVAL;
  }
#undef SAVIT_LAUNCH_EPI
  SAVIT_LAUNCH_RET();
}

// row panels per group of the ping-pong kernel's tile order: as many 256-row A-panels (256 x K x 2 B) as fit in about 3 MB of the
// XCD's 4 MB L2, at most 8 (K = 768: 8, K = 1024: 6, K >= 3072: 1 = plain row-major order).  Measured on DeiT-B with cold operands
// (tools/gemm_epi_bench.py, G = 1 -> 8): qkv 104 -> 97 us, fc1+GELU 156 -> 145 us; the GELU' epilogue, which also READS a [M, N]
// tensor, is best at 4 (177 -> 172 us; 186 us at 8: fewer concurrent tiles per row of that tensor).
inline int pp_row_group(int K, int tiles_m, int epilogue) {
  static const int force = SAVIT_EXP_ENV_INT("SAVIT_PP_ROW_GROUP", 0);  // SAVIT_EXPERIMENTS builds only
  int g = force > 0 ? force : (int)((3l << 20) / ((long)256 * K * 2));
  if (g < 1) g = 1;
  if (g > 8) g = 8;
  if (force <= 0 && epilogue == SAVIT_EPI_DGELU && g > 4) g = 4;
  if (g > tiles_m) g = tiles_m;
  return g;
}

#ifdef SAVIT_EXPERIMENTS
template <int ABL>
int launch_pp_ablation(const GemmParams& p0, hipStream_t s) {
  GemmParams p = p0;
  p.tiles_m = (p.a.M + 255) / 256;
  p.tiles_n = (p.a.N + 255) / 256;
  p.row_group = pp_row_group(p.a.K, p.tiles_m, p.a.epilogue);
  auto kfn = gemm_tn_pp_kernel<SAVIT_EPI_BF16, ABL>;
  SAVIT_LDS_ONCE(kfn);
  hipLaunchKernelGGL(kfn, dim3(p.tiles_m * p.tiles_n), dim3(512), 2 * 512 * 128, s, p);
  SAVIT_LAUNCH_RET();
}
#endif

int launch_pp(const GemmParams& p0, hipStream_t s) {
  GemmParams p = p0;
  p.tiles_m = (p.a.M + 255) / 256;
  p.tiles_n = (p.a.N + 255) / 256;
#ifdef SAVIT_EXPERIMENTS
  {  // staggered-start experiment (measured neutral to -10 %: DESIGN.md 6.2); read once
    static const int ph = SAVIT_EXP_ENV_INT("SAVIT_PP_PHASES", 1), sl = SAVIT_EXP_ENV_INT("SAVIT_PP_SLEEP", 0);
    p.desync_phases = ph < 1 ? 1 : ph;
    p.desync_sleep = sl;
  }
#endif
  p.row_group = pp_row_group(p.a.K, p.tiles_m, p.a.epilogue);
  const dim3 grid(p.tiles_m * p.tiles_n);
  const size_t lds = 2 * 512 * 128;
#define SAVIT_LAUNCH_EPI(E)                                                                            \
  case E: {                                                                                            \
    auto kfn = gemm_tn_pp_kernel<E>;                                                                   \
    SAVIT_LDS_ONCE(kfn);                                                                                \
    hipLaunchKernelGGL(kfn, grid, dim3(512), lds, s, p);                                               \
  } break;
  switch (p.a.epilogue) {
    SAVIT_LAUNCH_EPI(SAVIT_EPI_BF16)
    SAVIT_LAUNCH_EPI(SAVIT_EPI_BIAS_GELU)
    SAVIT_LAUNCH_EPI(SAVIT_EPI_RESID)
    SAVIT_LAUNCH_EPI(SAVIT_EPI_DGELU)
    SAVIT_LAUNCH_EPI(SAVIT_EPI_F32)
    default: return SAVIT_EINVAL;
  }
#undef SAVIT_LAUNCH_EPI
  SAVIT_LAUNCH_RET();
}

template <int BM, int BN, int WGM, int WGN, int ND>
int launch_pair(const GemmParams& p0, hipStream_t s) {
  GemmParams p = p0;
  p.tiles_m = (p.a.M + BM - 1) / BM;
  p.tiles_n = (p.a.N + BN - 1) / BN;
  const dim3 grid(p.tiles_m * p.tiles_n);
  const size_t lds = (size_t)ND * (BM + BN) * 128;
#define SAVIT_LAUNCH_EPI(E)                                                                            \
  case E: {                                                                                            \
    auto kfn = gemm_tn_pair_kernel<BM, BN, WGM, WGN, ND, E>;                                           \
    SAVIT_LDS_ONCE(kfn);                                                                                \
    hipLaunchKernelGGL(kfn, grid, dim3(64 * WGM * WGN), lds, s, p);                                   \
  } break;
  switch (p.a.epilogue) {
    SAVIT_LAUNCH_EPI(SAVIT_EPI_BF16)
    SAVIT_LAUNCH_EPI(SAVIT_EPI_BIAS_GELU)
    SAVIT_LAUNCH_EPI(SAVIT_EPI_RESID)
    SAVIT_LAUNCH_EPI(SAVIT_EPI_DGELU)
    SAVIT_LAUNCH_EPI(SAVIT_EPI_F32)
    SAVIT_LAUNCH_EPI(SAVIT_EPI_PATCH)
    default: return SAVIT_EINVAL;
  }
#undef SAVIT_LAUNCH_EPI
  SAVIT_LAUNCH_RET();
}

// rounds of `slots` resident workgroups a tail-split launch saves: plan = (tall row panels, short row panels), or 0 panels if the
// plain kernel is at least as good (less than one full round of tall tiles, or the remainder fills its round anyway)
inline bool tail_split_plan(int M, int N, int BM, int BM2, int BN, int slots, int* big_panels, int* small_panels) {
  const int tiles_n = (N + BN - 1) / BN;
  const int panels = (M + BM - 1) / BM;
  const long all = (long)panels * tiles_n;
  const long rounds_plain = (all + slots - 1) / slots;
  const int bp = (int)(((all / slots) * slots) / tiles_n);  // whole rounds of slots, rounded down to whole row panels
  if (bp <= 0 || bp >= panels) return false;
  const int rem_rows = M - bp * BM;
  const int sp = (rem_rows + BM2 - 1) / BM2;
  const long small_rounds = ((long)sp * tiles_n + slots - 1) / slots;
  const double t_tail = (double)(((long)bp * tiles_n + slots - 1) / slots) + (double)small_rounds * BM2 / BM;  // in tall-tile times
  if (t_tail > 0.93 * (double)rounds_plain) return false;
  *big_panels = bp;
  *small_panels = sp;
  return true;
}

template <int BM, int BM2, int BN, int WGM, int WGN, int ND>
int launch_pair_tail(const GemmParams& p0, hipStream_t s) {
  GemmParams p = p0;
  int bp = 0, sp = 0;
  const int slots = 2 * ((p.a.cu_budget > 0 && p.a.cu_budget < device_cus()) ? p.a.cu_budget : device_cus());
  if (!tail_split_plan(p.a.M, p.a.N, BM, BM2, BN, slots, &bp, &sp)) return launch_pair<BM, BN, WGM, WGN, ND>(p0, s);
  p.tiles_n = (p.a.N + BN - 1) / BN;
  p.tiles_m = bp;
  p.big_tiles = bp * p.tiles_n;
  p.small_tiles = sp * p.tiles_n;
  p.big_rows = bp * BM;
  const dim3 grid(p.big_tiles + p.small_tiles);
  const size_t lds = (size_t)ND * (BM + BN) * 128;
#define SAVIT_LAUNCH_EPI(E)                                                                            \
  case E: {                                                                                            \
    auto kfn = gemm_tn_pair_tail_kernel<BM, BM2, BN, WGM, WGN, ND, E>;                                 \
    SAVIT_LDS_ONCE(kfn);                                                                                \
    hipLaunchKernelGGL(kfn, grid, dim3(64 * WGM * WGN), lds, s, p);                                   \
  } break;
  switch (p.a.epilogue) {
    SAVIT_LAUNCH_EPI(SAVIT_EPI_BF16)
    SAVIT_LAUNCH_EPI(SAVIT_EPI_BIAS_GELU)
    SAVIT_LAUNCH_EPI(SAVIT_EPI_RESID)
    SAVIT_LAUNCH_EPI(SAVIT_EPI_F32)
    default: return launch_pair<BM, BN, WGM, WGN, ND>(p0, s);  // column-sum slabs are indexed per row tile: plain kernel
  }
#undef SAVIT_LAUNCH_EPI
  SAVIT_LAUNCH_RET();
}

template <int BM, int BN, int WGM, int WGN, int S, bool LATE = false>
int launch_ring(const GemmParams& p0, hipStream_t s) {
  GemmParams p = p0;
  p.tiles_m = (p.a.M + BM - 1) / BM;
  p.tiles_n = (p.a.N + BN - 1) / BN;
  const dim3 grid(p.tiles_m * p.tiles_n), block(64 * WGM * WGN);
  const size_t lds = (size_t)S * (BM + BN) * 64;
#define SAVIT_LAUNCH_EPI(E)                                                                            \
  case E: {                                                                                            \
    auto kfn = gemm_tn_ring_kernel<BM, BN, WGM, WGN, S, E, LATE>;                                            \
    SAVIT_LDS_ONCE(kfn);                                                                                \
    hipLaunchKernelGGL(kfn, grid, block, lds, s, p);                                                   \
  } break;
  switch (p.a.epilogue) {
    SAVIT_LAUNCH_EPI(SAVIT_EPI_BF16)
    SAVIT_LAUNCH_EPI(SAVIT_EPI_BIAS_GELU)
    SAVIT_LAUNCH_EPI(SAVIT_EPI_RESID)
    SAVIT_LAUNCH_EPI(SAVIT_EPI_DGELU)
    SAVIT_LAUNCH_EPI(SAVIT_EPI_F32)
    SAVIT_LAUNCH_EPI(SAVIT_EPI_PATCH)
    default: return SAVIT_EINVAL;
  }
#undef SAVIT_LAUNCH_EPI
  SAVIT_LAUNCH_RET();
}

template <int BM, int BN, int WGM, int WGN>
int launch_tile(const GemmParams& p0, hipStream_t s) {
  GemmParams p = p0;
  p.tiles_m = (p.a.M + BM - 1) / BM;
  p.tiles_n = (p.a.N + BN - 1) / BN;
  const dim3 grid(p.tiles_m * p.tiles_n), block(64 * WGM * WGN);
  const size_t lds = 2 * (BM + BN) * ROW_BYTES;
#define SAVIT_LAUNCH_EPI(E)                                                                            \
  case E: {                                                                                            \
    auto kfn = gemm_tn_kernel<BM, BN, WGM, WGN, E>;                                                    \
    SAVIT_LDS_ONCE(kfn);                                                                                \
    hipLaunchKernelGGL(kfn, grid, block, lds, s, p);                                                   \
  } break;
  switch (p.a.epilogue) {
    SAVIT_LAUNCH_EPI(SAVIT_EPI_BF16)
    SAVIT_LAUNCH_EPI(SAVIT_EPI_BIAS_GELU)
    SAVIT_LAUNCH_EPI(SAVIT_EPI_RESID)
    SAVIT_LAUNCH_EPI(SAVIT_EPI_DGELU)
    SAVIT_LAUNCH_EPI(SAVIT_EPI_F32)
    SAVIT_LAUNCH_EPI(SAVIT_EPI_PATCH)
    default: return SAVIT_EINVAL;
  }
#undef SAVIT_LAUNCH_EPI
  SAVIT_LAUNCH_RET();
}

}  // namespace

static int device_cus() {
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
  }();
  return cus;
}

extern "C" int savit_gemm_tn_auto_tile_epi(int M, int N, int K, int epilogue) { return savit_gemm_tn_auto_tile_cus(M, N, K, epilogue, 0); }

// cu_budget: the CUs this launch may count on (0 = all of the device's).  A data-parallel rank passes less than the device has while an
// all-reduce is resident (engine.reserved_cus): the round model below then prices the tiles for THAT many CUs - a 237-tile grid that is one
// round on 256 CUs is two rounds, the second nearly empty, on 224.
extern "C" int savit_gemm_tn_auto_tile_cus(int M, int N, int K, int epilogue, int cu_budget) {
  // measured on MI355X in the training pipeline (tools/profile_step.py, tools/gemm_probe.py; DeiT-B, DeiT-S, ViT-L/16-384 shapes).
  // K % 64 == 0 selects the paired-stage kernels (whole-cache-line LDS-DMA), other K the 64-B-row ring.
  //  * 192x128, two 4-wave workgroups per CU (80 KB of LDS each): the default.  Against 128x128 it needs 19 % fewer L2->LDS bytes per
  //    flop (the feed bounds these kernels) and keeps the co-resident workgroup that covers prologue and epilogue: fc2 167 -> 148 us,
  //    fc1/qkv input gradients -8..-10 %.
  //  * 256x256, one 8-wave workgroup per CU (half the feed again, nothing to overlap with): where its per-tile fixed costs are
  //    amortised - grids of at least two rounds with K >= 1024 (ViT-L: every product 5-20 % faster than 192x128), or the
  //    GELU-forward epilogue (two outputs).
  //  * 128x128: small or ragged problems (few rows, N not a multiple of 128).
  static const int force = SAVIT_EXP_ENV_INT("SAVIT_GEMM_TILE", 0);  // SAVIT_EXPERIMENTS builds only
  if (force > 0) return force;
  const long t256 = (long)((M + 255) / 256) * ((N + 255) / 256);
  const bool big = (t256 >= 512 && N % 128 == 0);
  //  * tile 24 (round 5): few rows - no LDS, no barriers, one wave per 32 x 32 outputs.  Measured in the DeiT-B step (128 cls rows of the last
  //    layer; tools/profile_step.py with SAVIT_PROFILE_LAYER=11): see the table at the kernel.
  static const int rows_max = SAVIT_EXP_ENV_INT("SAVIT_ROWS_TILE_MAX_M", 256);  // SAVIT_EXPERIMENTS builds only (A/B runs)
  if (M <= rows_max && (epilogue == SAVIT_EPI_BF16 || epilogue == SAVIT_EPI_BIAS_GELU || epilogue == SAVIT_EPI_RESID || epilogue == SAVIT_EPI_F32))
    return 24;
  if (K % 64 != 0) return 6;  // TNT's pixel stream (K = 32 / 96 / 160): narrow outputs, the 128x128 ring (a 256x256 ring tile existed through round 3; no shape of a supported model reached it)
  //  * 256x256 ping-pong (tile 20: the two M-halves of the 8-wave workgroup run one barrier apart, one in its MFMA segment while
  //    the other reads fragments and issues LDS-DMA): wide outputs on large grids with K >= 768 - measured against the next best
  //    tile with cold operands (tools/gemm_epi_bench.py): fc1+GELU 172 -> 155 us, fc2 input-gradient+GELU' 179 -> 171 us, qkv 117 -> 111 us,
  //    ViT-L shapes 4-6 % over the pair kernel.  Narrow outputs (N = 768) keep 192x128: 297 tiles of 256x256 are 1.16 rounds.
  //    K = 384 (DeiT-S, CaiT-S at 256 images) stays with the pair kernels: with cold operands the ping-pong tile is 7-11 % faster on
  //    those shapes too (SAVIT_EPI_SHAPE=50432,384,1536 tools/gemm_epi_bench.py), inside the training step - warm operands, same box,
  //    alternating runs - DeiT-S lost 1.4 % (17.63 -> 17.38 k img/s) and CaiT-S24, TNT, Mixer did not move.
  static const int pp_min_k = SAVIT_EXP_ENV_INT("SAVIT_PP_MIN_K", 768);  // SAVIT_EXPERIMENTS builds only (A/B runs)
  // (round 5: also K = 384 with a wide output - DeiT-S / CaiT-S fc1 and fc2 input gradient, N = 1536: in the training step, same box,
  //  alternating runs: DeiT-S 19 458 / 19 515 -> 19 686 / 19 717 img/s, CaiT-S24 6 524 -> 6 552; with cold operands fc1 + GELU 121 -> 103 us.
  //  NOT K = 512 / 640: MLP-Mixer-S/16 (d = 512, N = 2048) lost 3.2 % with it (23 403 -> 22 657 img/s), TNT-B's outer stream 0.3 %.)
  // (round 6, measured and NOT taken: K = 384 with N = 1152 - the qkv projection of DeiT-S / CaiT-S / TNT-S's outer stream - on the ping-pong
  //  tiles with the fifth column tile half empty (rows of W past N read as zeros, the stores are masked).  Cold operands: 77.6 (192 x 128)
  //  -> 72.4 (256 x 256) / 70.2 us (320 x 256); inside the training step, same box: qkv 65.4 -> 66.7 / 69.0 us, DeiT-S 20.44-20.52 k
  //  img/s either way, CaiT-S24 -0.3..-0.9 % (profiles/r06_ragged_n_ab.log) - warm operands again decide for the pair kernel.)
  static const int ragged_n = SAVIT_EXP_ENV_INT("SAVIT_PP_RAGGED_N", 0);  // SAVIT_EXPERIMENTS builds only: 1 = the round model's tile, 2 = 320 x 256
  const bool n_ok = N % 256 == 0 || (ragged_n && K == 384 && N % 256 == 128 && N >= 1024);
  if ((K >= pp_min_k || (K == 384 && N >= 1024 && pp_min_k == 768)) && n_ok && M >= 4096 && epilogue != SAVIT_EPI_PATCH) {
    //  * 320x256 ping-pong (tile 21, round 3) against 256x256 (tile 20) and 192x128 (17 / 18): what a launch costs is (rounds of
    //    workgroups over the CUs) x (rows of a tile), weighted by what the tile's operand feed costs - the L2 -> LDS path of a CU, not
    //    the matrix pipe, bounds these kernels, and a 192x128 tile moves 1.7x the bytes per flop of a 256-wide one (measured: the same
    //    product 1.35x slower per flop at equal fill).  DeiT-B (M = 25 216): N = 768 is 237 tiles of 320 rows = ONE round (297 of 256
    //    rows = two; 792 + tail of 192x128): fc1 input gradient 142 -> 98 us, fc2 + residual 166 -> 121, qkv input gradient 112 -> 80,
    //    proj + residual 69 -> 58; N = 2304: 3 rounds of 320 rows against 4 of 256: qkv 108 -> 96 us; N = 3072: 4 x 320 = 5 x 256, the
    //    320-row tile moves 10 % fewer bytes: fc1 + GELU 147 -> 143, GELU' 167 -> 162 (tools/gemm_epi_bench.py, cold operands).
    const int cus = (cu_budget > 0 && cu_budget < device_cus()) ? cu_budget : device_cus();
    const long tn = (N + 255) / 256;
    const long r256 = (((long)(M + 255) / 256 * tn + cus - 1) / cus) * 256;
    const long r320 = (((long)(M + 319) / 320 * tn + cus - 1) / cus) * 320;
    // 192x128, two workgroups per CU, the last partial round cut into 128-row tiles: rounds in thirds of a tile time
    const long t192 = (long)((M + 191) / 192) * (N / 128);
    const long c192 = ((t192 * 3 + 2L * cus - 1) / (2L * cus)) * 64 * 135 / 100;
    const long best = r320 <= r256 ? r320 : r256;
    //  * tile 22 = the 320 x 256 tile as a PERSISTENT grid (one workgroup per CU walking its tiles, the next tile's first K-tile
    //    requested under the current tile's last phases and epilogue; round 5).  Measured (tools/gemm_epi_bench.py 21 22, cold operands,
    //    profiles/r05_gemm_persistent.log): it does NOT pay on DeiT-B's 3-4-round launches (qkv 92 -> 98 us, fc1 + GELU 146 -> 145,
    //    GELU' 160 -> 163): vmcnt counts loads and stores in ONE in-order queue, so the first counted wait of the next tile also waits
    //    for the previous tile's 20-40 stores per wave to be acknowledged - while every CU of the chip is draining its 160 KB tile at
    //    the same moment - which a fresh workgroup of the one-tile grid never does.  On ViT-L's launches of 7-29 rounds with K >= 1024
    //    (a tile's main loop is 30-120 us, the store drain a small part of it) it is neutral to 8 % faster (qkv 860 -> 813 us): taken
    //    there only.
    if (best <= c192 || (big && N >= 1024)) {
      const long t320 = (long)(M + 319) / 320 * tn;
      if (ragged_n == 2 && N % 256 != 0) return 21;
      return r320 <= r256 ? ((K >= 1024 && t320 >= 6L * cus) ? 22 : 21) : 20;
    }
  }
  if (big && N >= 1024 && K >= 768 && epilogue != SAVIT_EPI_PATCH) return 20;
  if (big && (epilogue == SAVIT_EPI_BIAS_GELU || K >= 1024)) return 13;
  if (N % 128 == 0 && M >= 1536 && epilogue != SAVIT_EPI_PATCH) {
    //  * 192x128 with the LAST partial round cut into 128-row tiles (tile 18) when that saves at least 7 % of a round count:
    //    DeiT-B's N = 768 products are 792 tiles on 512 slots (2 rounds, the second at 55 %) -> 510 tall + 420 short tiles.
    int bp, sp;
    static const bool no_tail = SAVIT_EXP_ENV_INT("SAVIT_NO_TAIL_SPLIT", 0) != 0;  // SAVIT_EXPERIMENTS builds only (A/B runs)
    const int slots = 2 * ((cu_budget > 0 && cu_budget < device_cus()) ? cu_budget : device_cus());
    if (!no_tail && epilogue != SAVIT_EPI_DGELU && tail_split_plan(M, N, 192, 128, 128, slots, &bp, &sp)) return 18;
    return 17;
  }
  return big ? 13 : 12;
}

extern "C" int savit_gemm_tn_auto_tile(int M, int N, int K) { return savit_gemm_tn_auto_tile_epi(M, N, K, SAVIT_EPI_BF16); }

namespace {
// out[n] (+)= sum_r slab[r][n]: a block owns 64 columns (16 lanes x float4) and splits the rows over 16 groups (independent
// loads, ~rows/16 deep), then an LDS tree; fixed summation order, so the result is reproducible.  Tall slabs (the MLP-Mixer
// token GEMMs: >= 1024 rows of 64..256 columns, 28 us on 4 workgroups) are cut into gridDim.y row chunks that add their partial
// with fp32 atomics - launched only in accumulate mode.
__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* __restrict__ slab, int rows, int N, float* __restrict__ out,
                                                              int accumulate) {
  __shared__ float4 part[16][16];
  const int cx = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + cx * 4;
  const int chunk = (rows + gridDim.y - 1) / gridDim.y;
  const int rbeg = blockIdx.y * chunk;
  rows = min(rows, rbeg + chunk);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < N) {  // N % 4 == 0
    for (int r = rbeg + g; r < rows; r += 16) {
      const float4 v = *reinterpret_cast<const float4*>(slab + (size_t)r * N + c);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  part[g][cx] = s;
  __syncthreads();
  if (g == 0 && c < N) {
    float4 t = part[0][cx];
#pragma unroll
    for (int k = 1; k < 16; ++k) {
      const float4 v = part[k][cx];
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    if (gridDim.y > 1) {
      atomicAdd(out + c, t.x);
      atomicAdd(out + c + 1, t.y);
      atomicAdd(out + c + 2, t.z);
      atomicAdd(out + c + 3, t.w);
      return;
    }
    float4* o = reinterpret_cast<float4*>(out + c);
    if (accumulate) {
      const float4 old = *o;
      t.x += old.x; t.y += old.y; t.z += old.z; t.w += old.w;
    }
    *o = t;
  }
}

// rows of A per workgroup tile and wave rows per workgroup, by tile id
inline bool tile_geometry(int tile, int* bm, int* wgm) {
  switch (tile) {
    case 1: case 4: case 6: case 8: case 9: case 11: case 12: case 14: *bm = 128; *wgm = 2; return true;
    case 2: case 5: case 7: case 10: case 13: case 15: *bm = 256; *wgm = 2; return true;
    case 3: *bm = 256; *wgm = 4; return true;
    case 17: case 18: *bm = 192; *wgm = 2; return true;
    case 20: *bm = 256; *wgm = 2; return true;
    case 21: case 22: *bm = 320; *wgm = 4; return true;
    case 24: *bm = 16; *wgm = 1; return true;
    case 30: *bm = 256; *wgm = 4; return true;
    default: return false;
  }
}
}  // namespace

extern "C" int savit_gemm_colsum_rows(int M, int N, int K, int tile) { return savit_gemm_colsum_rows_cus(M, N, K, tile, 0); }

extern "C" int savit_gemm_colsum_rows_cus(int M, int N, int K, int tile, int cu_budget) {
  if (tile == 0) tile = savit_gemm_tn_auto_tile_cus(M, N, K, SAVIT_EPI_DGELU, cu_budget);  // the only epilogue with column sums
  int bm = 0, wgm = 0;
  if (!tile_geometry(tile, &bm, &wgm) || M < 0) return -1;
  return ((M + bm - 1) / bm) * wgm;
}

extern "C" int savit_colsum_finalize(const float* slab, int rows, int N, float* out, int accumulate, void* stream) {
  SAVIT_CHECK_ARG(slab && out && rows >= 0 && N > 0 && N % 4 == 0 && ((uintptr_t)slab % 16) == 0 && ((uintptr_t)out % 16) == 0);
  const int split = (accumulate && rows >= 512) ? (rows / 256 < 32 ? rows / 256 : 32) : 1;  // <= 263 rows on the ViT shapes: one chunk
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3((N + 63) / 64, split), dim3(256), 0, (hipStream_t)stream, slab, rows, N, out, accumulate);
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_gemm_bf16_tn(const savit_gemm_args* args, void* stream) {
  SAVIT_CHECK_ARG(args != nullptr);
  const savit_gemm_args& a = *args;
  SAVIT_CHECK_ARG(a.A && a.Bt && a.C && a.M >= 0 && a.N > 0 && a.K > 0);
  SAVIT_CHECK_ARG(a.K % 32 == 0 && (a.K % 64 == 0 || a.tile == 0 || a.tile >= 4) && a.N % 4 == 0 && (a.epilogue > SAVIT_EPI_DGELU || a.N % 8 == 0) && a.ldb >= a.K && a.ldb % 8 == 0 && a.ldc % 4 == 0 && a.ldc >= a.N);
  SAVIT_CHECK_ARG(((uintptr_t)a.A % 16) == 0 && ((uintptr_t)a.Bt % 16) == 0 && ((uintptr_t)a.C % 16) == 0);
  GemmParams p{};
  p.a = a;
  if (a.epilogue == SAVIT_EPI_PATCH) {
    SAVIT_CHECK_ARG(a.patch > 0 && a.patch % 8 == 0 && a.img_size % a.patch == 0 && a.K == a.patch * a.patch * 3);
    SAVIT_CHECK_ARG(a.tokens > 0 && a.token_offset >= 0 && (a.aux == nullptr || (a.ldaux >= a.N && a.ldaux % 4 == 0)));
    p.grid_side = a.img_size / a.patch;
    p.chunks_per_prow = a.patch * 3 / 8;
    SAVIT_CHECK_ARG(a.M % (p.grid_side * p.grid_side) == 0 && a.token_offset + p.grid_side * p.grid_side <= a.tokens);
  } else {
    // lda < K is allowed: columns [lda, K) of row m then alias the head of row m + 1 (zero beyond the last row: buffer descriptor),
    // for operands whose true width is not a multiple of the 32-deep K-step - Bt must hold zeros in those K columns
    SAVIT_CHECK_ARG(a.lda > 0 && a.lda % 8 == 0);
  }
  if (a.epilogue == SAVIT_EPI_BIAS_GELU) SAVIT_CHECK_ARG(a.C2 != nullptr && a.bias != nullptr);
  if (a.epilogue == SAVIT_EPI_RESID || a.epilogue == SAVIT_EPI_DGELU)
    SAVIT_CHECK_ARG(a.aux != nullptr && a.ldaux >= a.N && a.ldaux % 4 == 0);
  if (a.rowscale != nullptr) SAVIT_CHECK_ARG(a.rows_per_sample >= 1);
  if (a.M == 0) return SAVIT_OK;
  hipStream_t s = (hipStream_t)stream;
  int tile = a.tile;
  if (tile == 0) {
    tile = savit_gemm_tn_auto_tile_cus(a.M, a.N, a.K, a.epilogue, a.cu_budget);
    if ((tile == 20 || tile == 21 || tile == 22) && a.lda < a.K) tile = 13;  // the ping-pong kernels do not take the aliased-row operand form
  }
  if (a.colsum != nullptr && a.colsum_rows != 0) {
    int bm_ = 0, wgm_ = 0;
    SAVIT_CHECK_ARG(tile_geometry(tile, &bm_, &wgm_) && a.colsum_rows == ((a.M + bm_ - 1) / bm_) * wgm_);
  }
  // The product library holds exactly the tiles savit_gemm_tn_auto_tile_cus can return (tests/test_abi.py checks that every one of
  // them is reached by a shape of a supported model); the tiles it never picks - earlier rounds' forms, kept for A/B runs - exist
  // in SAVIT_EXPERIMENTS builds only (tools/build_variant.sh).
  switch (tile) {
    case 6: return launch_ring<128, 128, 2, 2, 4>(p, s);
    case 12: return a.K % 64 ? SAVIT_EINVAL : launch_pair<128, 128, 2, 2, 2>(p, s);
    case 13: return a.K % 64 ? SAVIT_EINVAL : launch_pair<256, 256, 2, 4, 2>(p, s);
    case 17: return a.K % 64 ? SAVIT_EINVAL : launch_pair<192, 128, 2, 2, 2>(p, s);
    case 18: return a.K % 64 ? SAVIT_EINVAL : launch_pair_tail<192, 128, 128, 2, 2, 2>(p, s);  // 17 with 128-row tiles for the last partial round
    case 20: return (a.K % 64 || a.lda < a.K || a.epilogue == SAVIT_EPI_PATCH) ? SAVIT_EINVAL : launch_pp(p, s);
    case 21: return (a.K % 64 || a.lda < a.K || a.epilogue == SAVIT_EPI_PATCH) ? SAVIT_EINVAL : launch_pp320(p, s, false);
    case 22: return (a.K % 64 || a.lda < a.K || a.epilogue == SAVIT_EPI_PATCH) ? SAVIT_EINVAL : launch_pp320(p, s, true);  // persistent over >1 round
    case 24: return launch_rows(p, s);  // few rows: no LDS, fragments straight from global memory
#ifdef SAVIT_EXPERIMENTS
    case 7: return launch_ring<256, 256, 2, 4, 4>(p, s);
    case 1: return launch_tile<128, 128, 2, 2>(p, s);
    case 2: return launch_tile<256, 256, 2, 4>(p, s);
    case 3: return launch_tile<256, 128, 4, 2>(p, s);
    case 4: return launch_ring<128, 256, 2, 2, 3>(p, s);
    case 5: return launch_ring<256, 128, 2, 2, 3>(p, s);
    case 8: return launch_ring<128, 128, 2, 2, 2>(p, s);
    case 9: return launch_ring<128, 128, 2, 2, 3>(p, s);
    case 10: return launch_ring<256, 256, 2, 4, 4, true>(p, s);
    case 11: return launch_ring<128, 128, 2, 2, 4, true>(p, s);
    case 14: return a.K % 64 ? SAVIT_EINVAL : launch_pair<128, 256, 2, 2, 2>(p, s);
    case 15: return a.K % 64 ? SAVIT_EINVAL : launch_pair<256, 128, 2, 2, 2>(p, s);
    case 30: return stream_ok(a) ? launch_stream(p, s) : SAVIT_EINVAL;  // persistent streaming 256x128 (round 3: built, bitwise-correct, slower)
    // timing-only ablations of tile 20 (wrong results by construction; SAVIT_EPI_BF16 only) - never in the product library
    case 101: return launch_pp_ablation<1>(p, s);
    case 102: return launch_pp_ablation<2>(p, s);
    case 103: return launch_pp_ablation<3>(p, s);
    case 104: return launch_pp_ablation<4>(p, s);
    case 107: return launch_pp_ablation<7>(p, s);
    case 108: return launch_pp_ablation<8>(p, s);
    case 109: return launch_pp_ablation<9>(p, s);
    case 110: return launch_pp_ablation<10>(p, s);
#endif
    default: return SAVIT_EINVAL;
  }
}
