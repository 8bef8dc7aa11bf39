// Small HBM-/latency-bound kernels around the encoder blocks: cls/pos rows, loss, optimizer, weight casts.
// Reference call sites are cited per kernel (paths relative to /root/reference).
#include "common.h"
#include "savit.h"

namespace {

// ---- x0[b, 0, :] = cls + pos[0]   (models/vit.py:81-85 concat of the tiled cls token; position_embed.py:56)
__global__ void cls_rows_kernel(const float* __restrict__ cls, const float* __restrict__ pos, float* __restrict__ x0, int B,
                                long row_stride, int d) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // float4 index
  const int per = d >> 2;
  if (i >= B * per) return;
  const int b = i / per, c = i - b * per;
  const float4 a = reinterpret_cast<const float4*>(cls)[c];
  const float4 p = reinterpret_cast<const float4*>(pos)[c];
  reinterpret_cast<float4*>(x0 + (size_t)b * row_stride)[c] = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
}

// ---- dpos[t,:] += sum_b dx0[b,t,:] ; dcls += sum_b dx0[b,0,:]   (backward of the two adds above)
__global__ void pos_cls_grad_kernel(const float* __restrict__ dx0, float* __restrict__ dpos, float* __restrict__ dcls, int B,
                                    int N, int d, int has_cls) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // float4 index over [N, d]
  const int per = d >> 2;
  if (i >= N * per) return;
  float4 s = make_float4(0, 0, 0, 0);
  const size_t stride = (size_t)N * per;
  const float4* src = reinterpret_cast<const float4*>(dx0) + i;
#pragma unroll 4
  for (int b = 0; b < B; ++b) {
    const float4 v = src[(size_t)b * stride];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  float4* o = reinterpret_cast<float4*>(dpos) + i;
  float4 cur = *o;
  *o = make_float4(cur.x + s.x, cur.y + s.y, cur.z + s.z, cur.w + s.w);
  if (has_cls && i < per && dcls != nullptr) {
    float4* oc = reinterpret_cast<float4*>(dcls) + i;
    float4 c2 = *oc;
    *oc = make_float4(c2.x + s.x, c2.y + s.y, c2.z + s.z, c2.w + s.w);
  }
}

// ---- label-smoothed softmax cross-entropy, forward + gradient  (train.py:83-90; optax.smooth_labels,
// optax.softmax_cross_entropy).  One 256-thread block per row.
//   y = one_hot(label) [mixed with one_hot(label2) by ratio] ; y = (1-a) y + a/C ; loss_row = -sum y log_softmax(z)
//   dz = (softmax(z) - y) * grad_scale     (grad_scale = 1/B for the batch mean)
constexpr int CE_THREADS = 256;
__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
  v = is_max ? wave_max(v) : wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float r = red[0];
#pragma unroll
  for (int w = 1; w < CE_THREADS / 64; ++w) r = is_max ? fmaxf(r, red[w]) : r + red[w];
  return r;
}

__global__ __launch_bounds__(CE_THREADS) void xent_kernel(const float* __restrict__ logits, int ld, const int* __restrict__ labels,
                                                           const int* __restrict__ labels2, const float* __restrict__ ratio,
                                                           float alpha, float grad_scale, float* __restrict__ loss_rows,
                                                           float* __restrict__ loss_mean, bf16_t* __restrict__ dz_bf16, int ld_dz,
                                                           float* __restrict__ dbias, float* __restrict__ top1,
                                                           float* __restrict__ top5, int B, int C) {
  __shared__ float red[CE_THREADS / 64];
  const int row = blockIdx.x;
  const float* z = logits + (size_t)row * ld;
  float m = -INFINITY;
  for (int c = threadIdx.x; c < C; c += CE_THREADS) m = fmaxf(m, z[c]);
  m = block_reduce(m, red, true);
  float s = 0.f;
  for (int c = threadIdx.x; c < C; c += CE_THREADS) s += __expf(z[c] - m);
  s = block_reduce(s, red, false);
  const float lse = m + __logf(s);
  const int l1 = labels[row];
  const int l2 = labels2 ? labels2[row] : l1;
  const float r1 = labels2 ? ratio[row] : 1.0f;
  const float zl = z[l1];
  float lsum = 0.f, above = 0.f;
  for (int c = threadIdx.x; c < C; c += CE_THREADS) {
    const float zc = z[c];
    float y = (c == l1 ? r1 : 0.f) + (c == l2 ? (1.0f - r1) : 0.f);
    y = (1.0f - alpha) * y + alpha / (float)C;
    const float logp = zc - lse;
    lsum -= y * logp;
    const float dz = (__expf(logp) - y) * grad_scale;
    if (dz_bf16) dz_bf16[(size_t)row * ld_dz + c] = f32_to_bf16(dz);
    if (dbias) atomicAdd(dbias + c, round_bf16(dz));
    above += (zc > zl) ? 1.f : 0.f;
  }
  if (dz_bf16)
    for (int c = C + threadIdx.x; c < ld_dz; c += CE_THREADS) dz_bf16[(size_t)row * ld_dz + c] = 0;
  lsum = block_reduce(lsum, red, false);
  above = block_reduce(above, red, false);
  if (threadIdx.x == 0) {
    if (loss_rows) loss_rows[row] = lsum;
    if (loss_mean) atomicAdd(loss_mean, lsum / (float)B);
    if (top1) top1[row] = above < 0.5f ? 1.f : 0.f;   // utils.py:20-31: label among the k largest logits
    if (top5) top5[row] = above < 4.5f ? 1.f : 0.f;
  }
}

// ---- global gradient norm, stage 1: out[0] += sum g^2   (optax.clip_by_global_norm, train.py:25)
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(g)[i];
    s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const float v = g[(n4 << 2) + threadIdx.x];
    s += v * v;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

// ---- the same over a LIST of ranges of one buffer, and the matching clear (round 5).  The grouped weight-gradient launches store every
// element of the matrices they cover (first touch: savit_wgrad_problem.overwrite) and add the sum of squares of what they stored to 32
// accumulators (savit_gemm_bf16_wgrad_grouped_ex), so what is left of "zero the gradient buffer" / "sum of squares of the gradient
// buffer" are the few ranges nothing overwrites: LayerNorm scales / biases, Dense biases, embeddings, head (6 MB of DeiT-B's 346 MB).
// A block takes one 2 048-float chunk of one range; ranges are 16-byte aligned and multiples of 4 floats.
constexpr int RANGE_MAX = 120, RANGE_CHUNK = 2048;
struct RangeList {
  int n;
  int chunk_end[RANGE_MAX];  // prefix sums of the chunks per range
  long off[RANGE_MAX], len[RANGE_MAX];
};
__device__ __forceinline__ bool range_chunk(const RangeList& r, long& begin, long& end, int b = blockIdx.x) {
  int i = 0;
  while (i < r.n && b >= r.chunk_end[i]) ++i;  // uniform scalar scan
  if (i >= r.n) return false;
  const long c = b - (i ? r.chunk_end[i - 1] : 0);
  begin = r.off[i] + c * RANGE_CHUNK;
  end = r.off[i] + r.len[i];
  if (end > begin + RANGE_CHUNK) end = begin + RANGE_CHUNK;
  return true;
}
__global__ __launch_bounds__(256) void zero_ranges_kernel(float* __restrict__ base, const RangeList r) {
  long b0, b1;
  if (!range_chunk(r, b0, b1)) return;
  for (long i = b0 + 4 * threadIdx.x; i < b1; i += 4 * 256) *reinterpret_cast<float4*>(base + i) = make_float4(0.f, 0.f, 0.f, 0.f);
}
// out[0] += sum over the ranges of g^2; block 0 also adds the `nslots` accumulators of the grouped weight-gradient launches
__global__ __launch_bounds__(256) void sumsq_ranges_kernel(const float* __restrict__ base, const RangeList r, const float* __restrict__ slots,
                                                            int nslots, float* __restrict__ out, int chunks) {
  __shared__ float red[4];
  float s = 0.f;
  // a block walks chunks blockIdx.x, + gridDim.x, ...: at most SUMSQ_BLOCKS adds meet on the one result word (one block per chunk
  // was 3 400 atomics on one address for DeiT-B's 6.9 M leftover gradients: 21 us of a 25 us launch)
  for (int c = blockIdx.x; c < chunks; c += gridDim.x) {
    long b0, b1;
    if (range_chunk(r, b0, b1, c))
      for (long i = b0 + 4 * threadIdx.x; i < b1; i += 4 * 256) {
        const float4 v = *reinterpret_cast<const float4*>(base + i);
        s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
      }
  }
  if (blockIdx.x == 0 && slots != nullptr && (int)threadIdx.x < nslots) s += slots[threadIdx.x];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}
constexpr int SUMSQ_BLOCKS = 512;

// ---- fused AdamW over the flat parameter buffer  (optax.chain(clip_by_global_norm, scale_by_adam,
// additive_weight_decay, scale(-lr)) + apply_updates: train.py:25-27,100 with the descent sign of
// simple_train.py:27).  28 B/param of HBM traffic, one launch for the whole model.
// mirror (nullable): the updated parameters are also stored rounded to bf16, in the same flat layout - the [in, out] MFMA operands
// of the input-gradient GEMMs are views into that mirror, so no separate cast pass re-reads the fp32 parameters (30 B/param).
template <bool MIRROR>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                     float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                                     float wd, float bc1, float bc2, const float* __restrict__ sumsq, float max_norm,
                                                     float grad_scale, bf16_t* __restrict__ mirror) {
  float gs = grad_scale;
  if (sumsq != nullptr && max_norm > 0.f) {
    const float norm = sqrtf(*sumsq) * grad_scale;
    if (!(norm < max_norm)) gs *= max_norm / norm;
  }
  // a thread owns 8 consecutive parameters (two float4 of each stream), so the bf16 mirror is written as one 16-byte piece
  const long n8 = n >> 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    float4 pv[2], mv[2], vv[2], gv[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      pv[h] = reinterpret_cast<float4*>(p)[2 * i + h];
      gv[h] = reinterpret_cast<const float4*>(g)[2 * i + h];
      mv[h] = reinterpret_cast<float4*>(m)[2 * i + h];
      vv[h] = reinterpret_cast<float4*>(v)[2 * i + h];
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float* pp = &pv[h].x; const float* gp = &gv[h].x; float* mp = &mv[h].x; float* vp = &vv[h].x;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float gg = gp[k] * gs;
        mp[k] = b1 * mp[k] + (1.0f - b1) * gg;
        vp[k] = b2 * vp[k] + (1.0f - b2) * gg * gg;
        const float u = (mp[k] / bc1) / (sqrtf(vp[k] / bc2) + eps) + wd * pp[k];
        pp[k] -= lr * u;
      }
      reinterpret_cast<float4*>(p)[2 * i + h] = pv[h];
      reinterpret_cast<float4*>(m)[2 * i + h] = mv[h];
      reinterpret_cast<float4*>(v)[2 * i + h] = vv[h];
    }
    if (MIRROR)
      reinterpret_cast<uint4*>(mirror)[i] = make_uint4(pack_bf16x2(pv[0].x, pv[0].y), pack_bf16x2(pv[0].z, pv[0].w),
                                                       pack_bf16x2(pv[1].x, pv[1].y), pack_bf16x2(pv[1].z, pv[1].w));
  }
  // n % 8 == 4: the last float4 (n % 4 == 0 is the entry point's contract)
  if ((n & 4) && blockIdx.x == 0 && threadIdx.x == 0) {
    const long i4 = (n >> 2) - 1;
    float4 pv = reinterpret_cast<float4*>(p)[i4], mv = reinterpret_cast<float4*>(m)[i4], vv = reinterpret_cast<float4*>(v)[i4];
    const float4 gv = reinterpret_cast<const float4*>(g)[i4];
    float* pp = &pv.x; const float* gp = &gv.x; float* mp = &mv.x; float* vp = &vv.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gg = gp[k] * gs;
      mp[k] = b1 * mp[k] + (1.0f - b1) * gg;
      vp[k] = b2 * vp[k] + (1.0f - b2) * gg * gg;
      const float u = (mp[k] / bc1) / (sqrtf(vp[k] / bc2) + eps) + wd * pp[k];
      pp[k] -= lr * u;
    }
    reinterpret_cast<float4*>(p)[i4] = pv;
    reinterpret_cast<float4*>(m)[i4] = mv;
    reinterpret_cast<float4*>(v)[i4] = vv;
    if (MIRROR) reinterpret_cast<uint2*>(mirror)[i4] = make_uint2(pack_bf16x2(pv.x, pv.y), pack_bf16x2(pv.z, pv.w));
  }
}

// ---- fp32 master weights -> bf16 MFMA operands in both layouts (batched over layers)
// src [batch][R][C] fp32 (batch stride src_bs elements) -> dst_n [batch][R][C] bf16 and dst_t [batch][C][R] bf16.
// 64x64 tile through LDS so both reads and both writes are coalesced.
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ src, long src_bs, int R, int C,
                                                              bf16_t* __restrict__ dst_n, long dn_bs, int ldn,
                                                              bf16_t* __restrict__ dst_t, long dt_bs, int ldt) {
  __shared__ float tile[64][65];
  const int bz = blockIdx.z;
  const float* s = src + (size_t)bz * src_bs;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int rr = ty; rr < 64; rr += 4) {
    const int r = r0 + rr, c = c0 + tx;
    float v = 0.f;
    if (r < R && c < C) v = s[(size_t)r * C + c];
    tile[rr][tx] = v;
    if (dst_n != nullptr && r < R && c < C) dst_n[(size_t)bz * dn_bs + (size_t)r * ldn + c] = f32_to_bf16(v);
  }
  __syncthreads();
  if (dst_t != nullptr) {
    for (int cc = ty; cc < 64; cc += 4) {
      const int c = c0 + cc, r = r0 + tx;
      if (r < R && c < C) dst_t[(size_t)bz * dt_bs + (size_t)c * ldt + r] = f32_to_bf16(tile[tx][cc]);
    }
  }
}

// ---- fp32 -> bf16 elementwise (images, small buffers)
__global__ void cast_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long n) {
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(src)[i];
    reinterpret_cast<uint2*>(dst)[i] = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst[(n4 << 2) + threadIdx.x] = f32_to_bf16(src[(n4 << 2) + threadIdx.x]);
}

// ---- batch layout change of the input pipeline: [H, W, C, N] fp32 -> [N, H, W, C] bf16  (train.py:80-81)
__global__ __launch_bounds__(256) void hwcn_to_nhwc_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long hwc,
                                                                 int n) {
  __shared__ float tile[64][65];
  const long p0 = (long)blockIdx.x * 64;  // position in hwc
  const int n0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int pp = ty; pp < 64; pp += 4) {
    const long pos = p0 + pp;
    const int nn = n0 + tx;
    tile[pp][tx] = (pos < hwc && nn < n) ? src[pos * n + nn] : 0.f;
  }
  __syncthreads();
  for (int nn = ty; nn < 64; nn += 4) {
    const long pos = p0 + tx;
    const int ni = n0 + nn;
    if (pos < hwc && ni < n) dst[(size_t)ni * hwc + pos] = f32_to_bf16(tile[tx][nn]);
  }
}

}  // namespace

extern "C" int savit_cls_pos_rows(const float* cls, const float* pos, float* x0, int B, long row_stride, int d, void* stream) {
  SAVIT_CHECK_ARG(cls && pos && x0 && B >= 0 && d > 0 && d % 4 == 0 && row_stride >= d && row_stride % 4 == 0);
  if (B == 0) return SAVIT_OK;
  const int n = B * (d / 4);
  hipLaunchKernelGGL(cls_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, cls, pos, x0, B, row_stride, d);
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_pos_cls_grad(const float* dx0, float* dpos, float* dcls, int B, int N, int d, int has_cls, void* stream) {
  SAVIT_CHECK_ARG(dx0 && dpos && B >= 0 && N > 0 && d > 0 && d % 4 == 0 && (!has_cls || dcls));
  if (B == 0) return SAVIT_OK;
  const int n = N * (d / 4);
  hipLaunchKernelGGL(pos_cls_grad_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, dx0, dpos, dcls, B, N, d, has_cls);
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_softmax_xent(const float* logits, int ld_logits, const int* labels, const int* mix_labels, const float* ratio,
                                  float label_smoothing, float grad_scale, float* loss_rows, float* loss_mean, void* dlogits_bf16,
                                  int ld_dlogits, float* dbias, float* top1, float* top5, int B, int C, void* stream) {
  SAVIT_CHECK_ARG(logits && labels && B >= 0 && C > 0 && ld_logits >= C);
  SAVIT_CHECK_ARG((mix_labels == nullptr) == (ratio == nullptr));
  SAVIT_CHECK_ARG(dlogits_bf16 == nullptr || ld_dlogits >= C);
  if (B == 0) return SAVIT_OK;
  hipLaunchKernelGGL(xent_kernel, dim3(B), dim3(CE_THREADS), 0, (hipStream_t)stream, logits, ld_logits, labels, mix_labels, ratio,
                     label_smoothing, grad_scale, loss_rows, loss_mean, (bf16_t*)dlogits_bf16, ld_dlogits, dbias, top1, top5, B, C);
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_sumsq(const float* g, long n, float* out, void* stream) {
  SAVIT_CHECK_ARG(g && out && n >= 0 && ((uintptr_t)g % 16) == 0);
  if (n == 0) return SAVIT_OK;
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, n, out);
  SAVIT_LAUNCH_RET();
}

static int fill_ranges(RangeList& r, const long* ranges, int first, int count) {  // -> chunks; ranges = (offset, length) pairs in floats
  r.n = count;
  int chunks = 0;
  for (int i = 0; i < count; ++i) {
    r.off[i] = ranges[2 * (first + i)];
    r.len[i] = ranges[2 * (first + i) + 1];
    chunks += (int)((r.len[i] + RANGE_CHUNK - 1) / RANGE_CHUNK);
    r.chunk_end[i] = chunks;
  }
  for (int i = count; i < RANGE_MAX; ++i) r.chunk_end[i] = chunks;
  return chunks;
}
static bool ranges_ok(const void* base, const long* ranges, int count) {
  if (!base || ((uintptr_t)base % 16) != 0 || (count > 0 && !ranges) || count < 0) return false;
  for (int i = 0; i < count; ++i)
    if (ranges[2 * i] < 0 || ranges[2 * i + 1] < 0 || (ranges[2 * i] % 4) != 0 || (ranges[2 * i + 1] % 4) != 0 ||
        ranges[2 * i + 1] > (long)RANGE_CHUNK * 0x3fffff) return false;
  return true;
}

extern "C" int savit_zero_ranges(float* base, const long* ranges, int count, void* stream) {
  SAVIT_CHECK_ARG(ranges_ok(base, ranges, count));
  for (int first = 0; first < count; first += RANGE_MAX) {
    RangeList r{};
    const int chunks = fill_ranges(r, ranges, first, count - first < RANGE_MAX ? count - first : RANGE_MAX);
    if (chunks == 0) continue;
    hipLaunchKernelGGL(zero_ranges_kernel, dim3((unsigned)chunks), dim3(256), 0, (hipStream_t)stream, base, r);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  }
  return SAVIT_OK;
}

extern "C" int savit_sumsq_ranges(const float* base, const long* ranges, int count, const float* slots, int nslots, float* out, void* stream) {
  SAVIT_CHECK_ARG(ranges_ok(base, ranges, count) && out && nslots >= 0 && nslots <= 256 && (nslots == 0 || slots));
  bool slots_done = (nslots == 0);
  for (int first = 0; first < count || !slots_done; first += RANGE_MAX) {
    RangeList r{};
    const int n = count - first < RANGE_MAX ? (count - first > 0 ? count - first : 0) : RANGE_MAX;
    int chunks = fill_ranges(r, ranges, first, n);
    if (chunks == 0 && slots_done) continue;
    const int grid = chunks == 0 ? 1 : (chunks < SUMSQ_BLOCKS ? chunks : SUMSQ_BLOCKS);  // (no chunk: block 0 still adds the accumulators)
    hipLaunchKernelGGL(sumsq_ranges_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, base, r, slots_done ? nullptr : slots,
                       slots_done ? 0 : nslots, out, chunks);
    slots_done = true;
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  }
  return SAVIT_OK;
}

extern "C" int savit_adamw_step(float* params, const float* grads, float* m, float* v, long n, float lr, float b1, float b2,
                                float eps, float weight_decay, int step, const float* grad_sumsq, float max_norm, float grad_scale,
                                void* stream) {
  return savit_adamw_step_mirror(params, grads, m, v, n, lr, b1, b2, eps, weight_decay, step, grad_sumsq, max_norm, grad_scale, nullptr, stream);
}

extern "C" int savit_adamw_step_mirror(float* params, const float* grads, float* m, float* v, long n, float lr, float b1, float b2,
                                       float eps, float weight_decay, int step, const float* grad_sumsq, float max_norm, float grad_scale,
                                       void* params_bf16, void* stream) {
  SAVIT_CHECK_ARG(params && grads && m && v && n >= 0 && n % 4 == 0 && step >= 1 && ((uintptr_t)params_bf16 % 8) == 0);
  SAVIT_CHECK_ARG(((uintptr_t)params % 16) == 0 && ((uintptr_t)grads % 16) == 0 && ((uintptr_t)m % 16) == 0 && ((uintptr_t)v % 16) == 0);
  if (n == 0) return SAVIT_OK;
  const float bc1 = 1.0f - powf(b1, (float)step), bc2 = 1.0f - powf(b2, (float)step);
  long blocks = (n / 8 + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks < 1) blocks = 1;
  if (params_bf16 != nullptr) {
    SAVIT_CHECK_ARG(((uintptr_t)params_bf16 % 16) == 0);
    hipLaunchKernelGGL(adamw_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, params, grads, m, v, n, lr, b1, b2, eps,
                       weight_decay, bc1, bc2, grad_sumsq, max_norm, grad_scale, (bf16_t*)params_bf16);
  } else {
    hipLaunchKernelGGL(adamw_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, params, grads, m, v, n, lr, b1, b2, eps,
                       weight_decay, bc1, bc2, grad_sumsq, max_norm, grad_scale, (bf16_t*)nullptr);
  }
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_cast_transpose_bf16(const float* src, long src_batch_stride, int batch, int R, int C, void* dst_n,
                                         long dst_n_batch_stride, int ld_n, void* dst_t, long dst_t_batch_stride, int ld_t,
                                         void* stream) {
  SAVIT_CHECK_ARG(src && batch >= 0 && R > 0 && C > 0 && (dst_n || dst_t));
  SAVIT_CHECK_ARG((dst_n == nullptr || ld_n >= C) && (dst_t == nullptr || ld_t >= R));
  if (batch == 0) return SAVIT_OK;
  hipLaunchKernelGGL(cast_transpose_kernel, dim3((C + 63) / 64, (R + 63) / 64, batch), dim3(256), 0, (hipStream_t)stream, src,
                     src_batch_stride, R, C, (bf16_t*)dst_n, dst_n_batch_stride, ld_n, (bf16_t*)dst_t, dst_t_batch_stride, ld_t);
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_cast_bf16(const float* src, void* dst, long n, void* stream) {
  SAVIT_CHECK_ARG(src && dst && n >= 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 8) == 0);
  if (n == 0) return SAVIT_OK;
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n);
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_hwcn_to_nhwc_bf16(const float* src, void* dst, int H, int W, int C, int N, void* stream) {
  SAVIT_CHECK_ARG(src && dst && H > 0 && W > 0 && C > 0 && N >= 0);
  if (N == 0) return SAVIT_OK;
  const long hwc = (long)H * W * C;
  hipLaunchKernelGGL(hwcn_to_nhwc_bf16_kernel, dim3((unsigned)((hwc + 63) / 64), (N + 63) / 64), dim3(256), 0, (hipStream_t)stream, src,
                     (bf16_t*)dst, hwc, N);
  SAVIT_LAUNCH_RET();
}
