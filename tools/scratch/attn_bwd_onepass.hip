// The one-pass attention backward of round 3 (attn_bwd1_kernel), moved out of csrc/attention.hip in round 4: built, bitwise-correct,
// and slower than the two-pass kernel (160.7 us against 141.5 us at DeiT-B's layer; DESIGN.md section 4 has the ablation table that the
// ABL_* switches below produced).  Not compiled into anything: to revive it, paste it back behind attn_bwd_kernel in attention.hip (it uses
// that file's helpers) and add its dispatch to savit_attention_bwd.
// ------------------------------------------------------------------------------------------ backward, one pass (N <= 224)
// EXPERIMENT BUILDS ONLY (tools/build_variant.sh; SAVIT_ATTN_ONEPASS=1): built and measured in round 3, bitwise-correct (it passes
// every attention-backward test), and SLOWER than the two-pass kernel above - 160.7 us against 141.5 us at DeiT-B's layer in
// tools/attn_bench.py.  The cross-wave dQ exchange (LDS read-add-write + one barrier per step: 34 us) costs more than recomputing S
// and dP in pass A (~20 us); LDS fp32 atomics serialise over the lanes (873 us).  DESIGN.md section 4 has the ablation table.
#ifdef SAVIT_EXPERIMENTS
// Every score is computed ONCE: a wave owns 32 keys (K, V row fragments and K^T fragments in registers for the whole item) and walks
// over the query tiles; per tile S = Q K^T and dP = dO V^T (8 MFMAs), P and dS in registers, dV^T += dO^T P and dK^T += Q^T dS
// (8 MFMAs), and its share of dQ^T = K^T dS^T (4 MFMAs) - 20 products per tile against 28 of the two-pass kernel above, and half
// the exponentials.  dS^T is the one operand whose contraction index (the key) sits on the lanes: it goes through a 4 KB LDS image
// of the wave's own (its rows of the K image, dead once K^T is in registers) and comes back with ds_read_b64_tr_b16.  The dQ shares
// of the key-owning waves meet in an fp32 LDS tile [query][68]: at step i wave w works on query tile (w + i) mod NT, so no two
// waves add into the same rows within a step, and a barrier between the steps fixes the order of the additions - bitwise
// reproducible like the two-pass kernel.  LDS: Q, dO, K images + the dQ tile + LSE / delta = 146 KB at NT = 7 (NT = 8 does not fit:
// N in 225..256 keeps the two-pass kernel).
constexpr int DQ_LD = 68;  // fp32 words per query row of the dQ tile: 16-byte aligned rows, consecutive queries 4 banks apart (b128 conflict-free)
template <int NT>
__global__ __launch_bounds__(64 * NT) void attn_bwd1_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int IMG = NT * 32 * ROWB;
  char* imgQ = smem;
  char* imgD = smem + IMG;      // dO
  char* imgK = smem + 2 * IMG;  // K; rows 32 w .. 32 w + 31 become wave w's dS^T image
  float* dq_s = reinterpret_cast<float*>(smem + 3 * IMG);
  float* lse_s = dq_s + NT * 32 * DQ_LD;
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
  stage_image<NT, NT>(imgK, srd, row_base, p.N, p.ld, p.d + hh * HD, wave, lane);
  stage_image<NT, NT>(imgQ, srd, row_base, p.N, p.ld, hh * HD, wave, lane);
  stage_image<NT, NT>(imgD, srdD, row_base, p.N, p.d, hh * HD, wave, lane);
  for (int i = threadIdx.x; i < NT * 32; i += 64 * NT)
    lse_s[i] = (i < p.N) ? -LOG2E * p.lse[((size_t)b * p.H + hh) * p.N + i] : -INFINITY;
  for (int i = threadIdx.x; i < NT * 32 * DQ_LD; i += 64 * NT) dq_s[i] = 0.f;

  const int ql = lane & 31, half = lane >> 5;
  const int g = lane >> 4, t = lane & 15;
  const int trow = 4 * (g >> 1) + (t >> 2);
  const int tcol = 16 * (g & 1) + 4 * (t & 3);
  const int key = wave * 32 + ql;  // this lane's key; as a query index in the prologue / epilogue: this lane's query

  // delta_q = sum_e dO[q][e] O[q][e] for the queries 32 w + ql (O from HBM), and this lane's K / V rows as B-operand fragments
  float delta = 0.f;
  bf16x8 kf[4], vf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    kf[ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    vf[ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    if (key < p.N) {
      const size_t off = (size_t)(row_base + key) * p.d + hh * HD + 16 * ks + 8 * half;
      const bf16x8 o8 = *reinterpret_cast<const bf16x8*>(p.o + off);
      const bf16x8 d8 = *reinterpret_cast<const bf16x8*>(p.d_o + off);
#pragma unroll
      for (int j = 0; j < 8; ++j) delta += bf16_to_f32((bf16_t)o8[j]) * bf16_to_f32((bf16_t)d8[j]);
      const size_t qoff = (size_t)(row_base + key) * p.ld + hh * HD + 16 * ks + 8 * half;
      kf[ks] = *reinterpret_cast<const bf16x8*>(p.qkv + qoff + p.d);
      vf[ks] = *reinterpret_cast<const bf16x8*>(p.qkv + qoff + 2 * p.d);
    }
  }
  delta = half_sum(delta);
  if (half == 0) del_s[key] = delta;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // K^T fragments of this wave's keys (A operand [e][key] of the dQ^T product); after this the wave's K rows are its dS^T image
  bf16x8 ktf[2][2];
#pragma unroll
  for (int eb = 0; eb < 2; ++eb)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) ktf[eb][s2] = lds_tr_frag(imgK, wave * 32 + 16 * s2 + trow, 32 * eb + tcol);

  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int eb = 0; eb < 2; ++eb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      dk[eb][r] = 0.f;
      dv[eb][r] = 0.f;
    }
  int qt = wave;
#pragma unroll 1
#ifdef ABL_STEPS
  for (int step = 0; step < ABL_STEPS; ++step) {
#else
  for (int step = 0; step < NT; ++step) {
#endif
    f32x16 sa = zero16, da = zero16;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#ifdef ABL_NO_ROWFRAG
      const bf16x8 qfr = kf[(ks + 1) & 3], dfr = vf[(ks + 1) & 3];
#else
      const bf16x8 qfr = lds_row_frag(imgQ, qt * 32 + ql, 2 * ks + half);
      const bf16x8 dfr = lds_row_frag(imgD, qt * 32 + ql, 2 * ks + half);
#endif
      sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr, kf[ks], sa, 0, 0, 0);  // S[q][key]: rows = queries (registers), lane = key
      da = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dfr, vf[ks], da, 0, 0, 0);  // dP[q][key]
    }
    // register r holds query qt*32 + 8*(r>>2) + 4*half + (r&3): -LSE*log2e and delta as four 16-byte reads each
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const float4 nl4 = *reinterpret_cast<const float4*>(lse_s + qt * 32 + 8 * g4 + 4 * half);
      const float4 dl4 = *reinterpret_cast<const float4*>(del_s + qt * 32 + 8 * g4 + 4 * half);
      const float nl[4] = {nl4.x, nl4.y, nl4.z, nl4.w};
      const float dl[4] = {dl4.x, dl4.y, dl4.z, dl4.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = 4 * g4 + j;
#ifdef ABL_NO_EXP
        const float pr = sa[r] + nl[j];
#else
        const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(sa[r], LOG2E, nl[j]));  // rows q >= N: -inf -> 0
#endif
        sa[r] = pr;
        da[r] = pr * (da[r] - dl[j]);
      }
    }
    bf16x8 pf[2], dsf[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      pf[s2] = acc_to_frag(sa, s2);
      dsf[s2] = acc_to_frag(da, s2);
    }
    // dS^T image of the wave: row = key, 4 consecutive queries (registers 4 g4 .. 4 g4 + 3) = 8 bytes at column 8 g4 + 4 half
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      union { bf16x8 v; uint2 u[2]; } w;
      w.v = dsf[g4 >> 1];
      *reinterpret_cast<uint2*>(imgK + img_off(key, g4) + 8 * half) = w.u[g4 & 1];
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
      for (int eb = 0; eb < 2; ++eb) {
#ifdef ABL_NO_TR
        const bf16x8 dtf = kf[eb + s2], qtf = vf[eb + s2];
#else
        const bf16x8 dtf = lds_tr_frag(imgD, qt * 32 + 16 * s2 + trow, 32 * eb + tcol);
        const bf16x8 qtf = lds_tr_frag(imgQ, qt * 32 + 16 * s2 + trow, 32 * eb + tcol);
#endif
#ifdef ABL_NO_MFMA2
        dv[eb][0] += bf16_to_f32((bf16_t)dtf[0]) * bf16_to_f32((bf16_t)pf[s2][0]);
        dk[eb][0] += bf16_to_f32((bf16_t)qtf[0]) * bf16_to_f32((bf16_t)dsf[s2][0]);
#else
        dv[eb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dtf, pf[s2], dv[eb], 0, 0, 0);
        dk[eb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf, dsf[s2], dk[eb], 0, 0, 0);
#endif
      }
    }
    // this wave's share of dQ^T[e][q] = sum over its keys of K^T[e][key] dS^T[key][q] (lane = query, registers = e)
#ifndef ABL_NO_DQ
    {
      const bf16x8 st0 = lds_tr_frag(imgK, wave * 32 + trow, tcol);
      const bf16x8 st1 = lds_tr_frag(imgK, wave * 32 + 16 + trow, tcol);
      float* qrow = dq_s + (qt * 32 + ql) * DQ_LD;
#pragma unroll
      for (int eb = 0; eb < 2; ++eb) {
        f32x16 dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf[eb][0], st0, zero16, 0, 0, 0);
        dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf[eb][1], st1, dq, 0, 0, 0);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {  // plain read-add-write: LDS fp32 atomics serialise over the lanes (measured 7x the whole kernel)
          float4* cell = reinterpret_cast<float4*>(qrow + 32 * eb + 8 * g4 + 4 * half);
          float4 v = *cell;
          v.x += dq[4 * g4];
          v.y += dq[4 * g4 + 1];
          v.z += dq[4 * g4 + 2];
          v.w += dq[4 * g4 + 3];
          *cell = v;
        }
      }
    }
#endif
#ifndef ABL_NO_BARRIER
    __syncthreads();  // orders the additions of consecutive steps into one query tile (and this wave's next dS^T image behind its reads)
#endif
    qt = (qt + 1 == NT) ? 0 : qt + 1;
  }
#ifdef ABL_NO_STORE
  if (key < p.N && dk[0][0] == 123.f) {
#else
  if (key < p.N) {
#endif
    bf16_t* krow = p.dqkv + (size_t)(row_base + key) * p.ld + p.d + hh * HD;
    bf16_t* vrow = krow + p.d;
    store_row_tile(krow, dk, half, HD, 1.0f);
    store_row_tile(vrow, dv, half, HD, 1.0f);
    // dQ of query 32 w + ql: this lane converts columns 32 half .. 32 half + 31
    const float* qrow = dq_s + key * DQ_LD + 32 * half;
    bf16_t* drow = p.dqkv + (size_t)(row_base + key) * p.ld + hh * HD + 32 * half;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float4 lo = *reinterpret_cast<const float4*>(qrow + 8 * c), hi = *reinterpret_cast<const float4*>(qrow + 8 * c + 4);
      const uint32_t w[4] = {pack_bf16x2(lo.x * p.dq_scale, lo.y * p.dq_scale), pack_bf16x2(lo.z * p.dq_scale, lo.w * p.dq_scale),
                             pack_bf16x2(hi.x * p.dq_scale, hi.y * p.dq_scale), pack_bf16x2(hi.z * p.dq_scale, hi.w * p.dq_scale)};
      *reinterpret_cast<uint4*>(drow + 8 * c) = make_uint4(w[0], w[1], w[2], w[3]);
    }
  }
}

#endif  // SAVIT_EXPERIMENTS
