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

namespace {

constexpr int BK = 64;           // bf16 elements per K-tile  (128 B per LDS row)
constexpr int ROW_BYTES = 128;

struct GemmParams {
  savit_gemm_args a;
  int tiles_m, tiles_n;
  int chunks_per_prow;  // PATCH: 16-B chunks per (patch row) = patch*3/8
  int grid_side;        // PATCH: patches per image side
};

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
    float4 o;
    o.x = r.x + rs * cs[0] * round_bf16(v[0]);
    o.y = r.y + rs * cs[1] * round_bf16(v[1]);
    o.z = r.z + rs * cs[2] * round_bf16(v[2]);
    o.w = r.w + rs * cs[3] * round_bf16(v[3]);
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
    const float4 pos = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a.aux) + (size_t)tok * a.ldaux + n);
    float4 o = make_float4(round_bf16(v[0]) + pos.x, round_bf16(v[1]) + pos.y, round_bf16(v[2]) + pos.z, round_bf16(v[3]) + pos.w);
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.C) + ((size_t)b * a.tokens + tok) * a.ldc + n) = o;
  }
}

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

  // ---- epilogue: lane holds C[m = .. + (lane&15)][n = .. + 4*(lane>>4) + 0..3]
  const int mrow = row0 + wm * WTM + fr;
  const int ncol = col0 + wn * WTN + fq * 4;
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    float csum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < MI; ++i) epilogue_store<EPI>(p, mrow + i * 16, ncol + j * 16, acc[i][j], csum);
    if (EPI == SAVIT_EPI_DGELU && a.colsum != nullptr) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float s = csum[c];
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64);
        s += __shfl_xor(s, 8, 64);
        const int n = ncol + j * 16 + c;
        if (fr == 0 && n < a.N) atomicAdd(a.colsum + n, s);
      }
    }
  }
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
    if (lds > 48 * 1024) {                                                                             \
      hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      if (e != hipSuccess) return (int)e;                                                              \
    }                                                                                                  \
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

extern "C" int savit_gemm_bf16_tn(const savit_gemm_args* args, void* stream) {
  SAVIT_CHECK_ARG(args != nullptr);
  const savit_gemm_args& a = *args;
  SAVIT_CHECK_ARG(a.A && a.Bt && a.C && a.M >= 0 && a.N > 0 && a.K > 0);
  SAVIT_CHECK_ARG(a.K % BK == 0 && a.N % 4 == 0 && a.ldb >= a.K && a.ldb % 8 == 0 && a.ldc % 4 == 0 && a.ldc >= a.N);
  SAVIT_CHECK_ARG(((uintptr_t)a.A % 16) == 0 && ((uintptr_t)a.Bt % 16) == 0 && ((uintptr_t)a.C % 16) == 0);
  GemmParams p{};
  p.a = a;
  if (a.epilogue == SAVIT_EPI_PATCH) {
    SAVIT_CHECK_ARG(a.patch > 0 && a.patch % 8 == 0 && a.img_size % a.patch == 0 && a.K == a.patch * a.patch * 3);
    SAVIT_CHECK_ARG(a.aux != nullptr && a.tokens > 0 && a.token_offset >= 0 && a.ldaux >= a.N && a.ldaux % 4 == 0);
    p.grid_side = a.img_size / a.patch;
    p.chunks_per_prow = a.patch * 3 / 8;
    SAVIT_CHECK_ARG(a.M % (p.grid_side * p.grid_side) == 0 && a.token_offset + p.grid_side * p.grid_side <= a.tokens);
  } else {
    SAVIT_CHECK_ARG(a.lda >= a.K && a.lda % 8 == 0);
  }
  if (a.epilogue == SAVIT_EPI_BIAS_GELU) SAVIT_CHECK_ARG(a.C2 != nullptr && a.bias != nullptr);
  if (a.epilogue == SAVIT_EPI_RESID || a.epilogue == SAVIT_EPI_DGELU)
    SAVIT_CHECK_ARG(a.aux != nullptr && a.ldaux >= a.N && a.ldaux % 4 == 0);
  if (a.rowscale != nullptr) SAVIT_CHECK_ARG(a.rows_per_sample >= 1);
  if (a.M == 0) return SAVIT_OK;
  hipStream_t s = (hipStream_t)stream;
  int tile = a.tile;
  if (tile == 0) {
    // heuristic: big tiles only when they still fill the chip (>= ~2 waves of 256 CUs)
    const long t256 = (long)((a.M + 255) / 256) * ((a.N + 255) / 256);
    tile = (t256 >= 512 && a.N % 256 == 0) ? 2 : 1;
  }
  switch (tile) {
    case 1: return launch_tile<128, 128, 2, 2>(p, s);
    case 2: return launch_tile<256, 256, 2, 4>(p, s);
    case 3: return launch_tile<256, 128, 4, 2>(p, s);
    default: return SAVIT_EINVAL;
  }
}
