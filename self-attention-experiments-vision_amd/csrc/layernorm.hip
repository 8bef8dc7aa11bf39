// LayerNorm backward, LayerScale backward and their finalize kernels for the fp32 residual stream (HBM-bound).
// Semantics: the VJP of flax nn.LayerNorm(dtype=bf16) as used at /root/reference models/vit.py:19,26,57 and
// models/cait.py:30,42,99,111,176 (forward: layernorm_fwd.hip).  Algorithmic bytes: (2+4+4+4+2)*rows*d.
#include "layernorm_common.h"

namespace {


// Backward.  dy bf16 [rows,d] (cotangent of the bf16 LN output), x fp32, mean/rstd from forward.
//   xhat = (x-mean)*rstd ; g = dy*gamma ; dx = rstd*(g - mean_d(g) - xhat*mean_d(g*xhat))
//   out  = dx (+ dres_in)           -> fp32 dx_out (gradient of the residual stream) and optional bf16 copy
//   dgamma += sum_rows dy*xhat ; dbeta += sum_rows dy ; dcolsum += sum_rows out   (per-lane registers ->
//   LDS cross-wave -> per-block partial slab -> finalize kernel)
// dres_in and dx_out carry no __restrict__: the engines pass the SAME buffer for both (the residual gradient is updated in place; a
// row is read - one pipeline stage ahead - before it is written, by the wave that owns it).
template <int CH>
__global__ __launch_bounds__(LN_THREADS) void ln_bwd_kernel(const bf16_t* __restrict__ dy, const float* __restrict__ x,
                                                             const float* __restrict__ gamma, const float* __restrict__ mean_in,
                                                             const float* __restrict__ rstd_in, const float* dres_in,
                                                             float* dx_out, bf16_t* __restrict__ dx_bf16,
                                                             float* __restrict__ partial, int rows, int d, long x_stride,
                                                             long out_stride, int round_params, int grp, int grp_stride, int grp_off,
                                                             int res_mod = 0, long res_stride = 0, int stat_stride = 1) {
  // res_mod > 0: the residual-gradient input is SPARSE - only rows r with r % res_mod == 0 have one, stored compactly at
  // dres_in + (r / res_mod) * res_stride (the cls rows of a ViT's last layer: engine.py, round 5); stat_stride: mean / rstd of row r
  // sit at index r * stat_stride (the cls rows' statistics inside the per-token arrays of the forward pass).
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int nchunk = d >> 2;
  float4 g[CH], dg[CH], db[CH], dc[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int ci = lane + 64 * c;
    g[c] = (ci < nchunk) ? reinterpret_cast<const float4*>(gamma)[ci] : make_float4(0, 0, 0, 0);
    if (round_params) g[c] = make_float4(round_bf16(g[c].x), round_bf16(g[c].y), round_bf16(g[c].z), round_bf16(g[c].w));
    dg[c] = make_float4(0, 0, 0, 0);
    db[c] = make_float4(0, 0, 0, 0);
    dc[c] = make_float4(0, 0, 0, 0);
  }
  const float inv_d = 1.0f / (float)d;
  // two-deep software pipeline over this wave's rows: the loads of the NEXT row (x, dy, residual gradient, statistics) are in
  // flight while this row is reduced and written, so a wave always has a row's worth of HBM requests outstanding.  (The residual
  // gradient used to be loaded after the two row reductions: a second, exposed, HBM round trip per row.)
  float4 xv_n[CH], rs_n[CH];
  uint2 dv_n[CH];
  float mean_n = 0.f, rstd_n = 0.f;
  auto load_row = [&](int row) {
    const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * x_stride);
    const size_t dyrow = grp > 0 ? (size_t)(row / grp) * grp_stride + grp_off + (row % grp) : (size_t)row;
    const uint2* dyr = reinterpret_cast<const uint2*>(dy + dyrow * d);
    mean_n = mean_in[(size_t)row * stat_stride];
    rstd_n = rstd_in[(size_t)row * stat_stride];
    const float* rsrc = nullptr;
    if (dres_in != nullptr) {
      if (res_mod <= 0) rsrc = dres_in + (size_t)row * out_stride;
      else if (row % res_mod == 0) rsrc = dres_in + (size_t)(row / res_mod) * res_stride;
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int ci = lane + 64 * c;
      xv_n[c] = make_float4(0, 0, 0, 0);
      rs_n[c] = make_float4(0, 0, 0, 0);
      dv_n[c] = make_uint2(0u, 0u);
      if (ci < nchunk) {
        xv_n[c] = xr[ci];
        dv_n[c] = dyr[ci];
        if (rsrc) rs_n[c] = reinterpret_cast<const float4*>(rsrc)[ci];
      }
    }
  };
  const int row_step = gridDim.x * LN_WAVES;
  int row = blockIdx.x * LN_WAVES + wave;
  if (row < rows) load_row(row);
  for (; row < rows; row += row_step) {
    const float mean = mean_n, rstd = rstd_n;
    float4 xh[CH], gy[CH], rs[CH];
    uint2 dvc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      xh[c] = xv_n[c];
      rs[c] = rs_n[c];
      dvc[c] = dv_n[c];
    }
    if (row + row_step < rows) load_row(row + row_step);
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int ci = lane + 64 * c;
      if (ci < nchunk) {
        const float4 xv = xh[c];
        const uint2 dv = dvc[c];
        const float d0 = __uint_as_float(dv.x << 16), d1 = __uint_as_float(dv.x & 0xffff0000u);
        const float d2 = __uint_as_float(dv.y << 16), d3 = __uint_as_float(dv.y & 0xffff0000u);
        xh[c] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
        gy[c] = make_float4(d0 * g[c].x, d1 * g[c].y, d2 * g[c].z, d3 * g[c].w);
        dg[c].x += d0 * xh[c].x; dg[c].y += d1 * xh[c].y; dg[c].z += d2 * xh[c].z; dg[c].w += d3 * xh[c].w;
        db[c].x += d0; db[c].y += d1; db[c].z += d2; db[c].w += d3;
        c1 += (gy[c].x + gy[c].y) + (gy[c].z + gy[c].w);
        c2 += (gy[c].x * xh[c].x + gy[c].y * xh[c].y) + (gy[c].z * xh[c].z + gy[c].w * xh[c].w);
      } else {
        xh[c] = make_float4(0, 0, 0, 0);
        gy[c] = make_float4(0, 0, 0, 0);
      }
    }
    c1 = wave_sum(c1) * inv_d;
    c2 = wave_sum(c2) * inv_d;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int ci = lane + 64 * c;
      if (ci < nchunk) {
        float4 o = make_float4(rstd * (gy[c].x - c1 - xh[c].x * c2), rstd * (gy[c].y - c1 - xh[c].y * c2),
                               rstd * (gy[c].z - c1 - xh[c].z * c2), rstd * (gy[c].w - c1 - xh[c].w * c2));
        o.x += rs[c].x; o.y += rs[c].y; o.z += rs[c].z; o.w += rs[c].w;
        reinterpret_cast<float4*>(dx_out + (size_t)row * out_stride)[ci] = o;
        if (dx_bf16)
          reinterpret_cast<uint2*>(dx_bf16 + (size_t)row * out_stride)[ci] = make_uint2(pack_bf16x2(o.x, o.y), pack_bf16x2(o.z, o.w));
        dc[c].x += o.x; dc[c].y += o.y; dc[c].z += o.z; dc[c].w += o.w;
      }
    }
  }
  // cross-wave reduction through LDS, then ONE plain store per (block, column) into the partial slab
  // partial[block][which][d]; ln_bwd_finalize_kernel sums the slab (all blocks adding straight into the
  // same 3*d addresses would serialise: same-address fp32 atomics run ~14x below the chip rate).
  __shared__ float4 red[LN_WAVES][64];
#pragma unroll
  for (int which = 0; which < 3; ++which) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const float4 v = which == 0 ? dg[c] : (which == 1 ? db[c] : dc[c]);
      __syncthreads();
      red[wave][lane] = v;
      __syncthreads();
      if (wave == 0) {
        float4 t = red[0][lane];
#pragma unroll
        for (int w = 1; w < LN_WAVES; ++w) {
          t.x += red[w][lane].x; t.y += red[w][lane].y; t.z += red[w][lane].z; t.w += red[w][lane].w;
        }
        const int ci = lane + 64 * c;
        if (ci < nchunk) reinterpret_cast<float4*>(partial + ((size_t)blockIdx.x * 3 + which) * d)[ci] = t;
      }
    }
  }
}


__global__ __launch_bounds__(LN_THREADS) void ln_bwd_narrow_kernel(const bf16_t* __restrict__ dy, const float* __restrict__ x,
                                                                    const float* __restrict__ gamma, const float* __restrict__ mean_in,
                                                                    const float* __restrict__ rstd_in, const float* dres_in,
                                                                    float* dx_out, bf16_t* __restrict__ dx_bf16,
                                                                    float* __restrict__ partial, int rows, int d, long x_stride, long out_stride,
                                                                    int round_params) {
  const int sub = threadIdx.x >> 4, ci = threadIdx.x & 15;
  const bool on = ci < (d >> 2);
  float4 g = make_float4(0, 0, 0, 0);
  if (on) {
    g = reinterpret_cast<const float4*>(gamma)[ci];
    if (round_params) g = make_float4(round_bf16(g.x), round_bf16(g.y), round_bf16(g.z), round_bf16(g.w));
  }
  float4 dg = make_float4(0, 0, 0, 0), db = make_float4(0, 0, 0, 0), dc = make_float4(0, 0, 0, 0);
  const float inv_d = 1.0f / (float)d;
  const int step = gridDim.x * 16;
  for (int row0 = blockIdx.x * 16; row0 < rows; row0 += step) {
    const int row = row0 + sub;
    const bool live = on && row < rows;
    float4 xv = make_float4(0, 0, 0, 0), rs = make_float4(0, 0, 0, 0);
    uint2 dv = make_uint2(0u, 0u);
    float mean = 0.f, rstd = 0.f;
    if (live) {
      xv = reinterpret_cast<const float4*>(x + (size_t)row * x_stride)[ci];
      dv = reinterpret_cast<const uint2*>(dy + (size_t)row * d)[ci];
      if (dres_in) rs = reinterpret_cast<const float4*>(dres_in + (size_t)row * out_stride)[ci];
      mean = mean_in[row];
      rstd = rstd_in[row];
    }
    const float d0 = __uint_as_float(dv.x << 16), d1 = __uint_as_float(dv.x & 0xffff0000u);
    const float d2 = __uint_as_float(dv.y << 16), d3 = __uint_as_float(dv.y & 0xffff0000u);
    const float4 xh = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
    const float4 gy = make_float4(d0 * g.x, d1 * g.y, d2 * g.z, d3 * g.w);
    float c1 = (gy.x + gy.y) + (gy.z + gy.w);
    float c2 = (gy.x * xh.x + gy.y * xh.y) + (gy.z * xh.z + gy.w * xh.w);
    c1 = row16_sum(live ? c1 : 0.f) * inv_d;
    c2 = row16_sum(live ? c2 : 0.f) * inv_d;
    if (live) {
      dg.x += d0 * xh.x; dg.y += d1 * xh.y; dg.z += d2 * xh.z; dg.w += d3 * xh.w;
      db.x += d0; db.y += d1; db.z += d2; db.w += d3;
      float4 o = make_float4(rstd * (gy.x - c1 - xh.x * c2), rstd * (gy.y - c1 - xh.y * c2), rstd * (gy.z - c1 - xh.z * c2),
                             rstd * (gy.w - c1 - xh.w * c2));
      o.x += rs.x; o.y += rs.y; o.z += rs.z; o.w += rs.w;
      reinterpret_cast<float4*>(dx_out + (size_t)row * out_stride)[ci] = o;
      if (dx_bf16) reinterpret_cast<uint2*>(dx_bf16 + (size_t)row * out_stride)[ci] = make_uint2(pack_bf16x2(o.x, o.y), pack_bf16x2(o.z, o.w));
      dc.x += o.x; dc.y += o.y; dc.z += o.z; dc.w += o.w;
    }
  }
  // the 16 row slots of the workgroup fold through LDS; one plain store per (block, column) into partial[block][which][d]
  __shared__ float4 red[16][16];
#pragma unroll
  for (int which = 0; which < 3; ++which) {
    const float4 v = which == 0 ? dg : (which == 1 ? db : dc);
    __syncthreads();
    red[sub][ci] = v;
    __syncthreads();
    if (sub == 0 && on) {
      float4 t = red[0][ci];
#pragma unroll
      for (int w = 1; w < 16; ++w) {
        t.x += red[w][ci].x; t.y += red[w][ci].y; t.z += red[w][ci].z; t.w += red[w][ci].w;
      }
      reinterpret_cast<float4*>(partial + ((size_t)blockIdx.x * 3 + which) * d)[ci] = t;
    }
  }
}

// Sums the partial slab [nblk][3][d] over blocks: grid (ceil(3d/64), FIN_SPLIT); a block handles 64 columns x
// its share of slab rows with 4 row-lanes per column, reduces in LDS and issues one atomic per column
// (FIN_SPLIT adders per address).
constexpr int FIN_SPLIT = 8;
// Blocks behind the first ceil(3d/64) columns of the grid (y = 0 only) reduce an EXTRA slab [xrows][xn] into xout += column sums, in a
// fixed order (one adder per column): the bias-gradient partials of the GELU' GEMM ride along with a LayerNorm finalize of the same
// layer instead of paying for a launch of their own (12 per DeiT-B step).
__device__ __forceinline__ void ln_finalize_body(const int bx, const int by, const float* __restrict__ partial, int nblk, int d,
                                                 float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dcolsum,
                                                 const float* __restrict__ xslab, int xrows, int xn, float* __restrict__ xout, int nf,
                                                 float* __restrict__ out3) {
  const int nb0 = (nf * d + 63) / 64;  // nf families of column sums in a slab row (3; 4 from ln_bwd_ls_kernel)
  if (bx >= nb0) {
    if (by != 0) return;
    // 64 columns = 16 lanes x float4, the rows over 16 groups (independent loads, ~rows/16 deep), then an LDS tree: fixed order
    __shared__ float4 xpart[16][16];
    const int cx = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int c = (bx - nb0) * 64 + cx * 4;
    float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < xn) {  // xn % 4 == 0
      for (int r = grp; r < xrows; r += 16) {
        const float4 v = *reinterpret_cast<const float4*>(xslab + (size_t)r * xn + c);
        s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
      }
    }
    xpart[grp][cx] = s4;
    __syncthreads();
    if (grp == 0 && c < xn) {
      float4 t4 = xpart[0][cx];
#pragma unroll
      for (int k = 1; k < 16; ++k) {
        const float4 v = xpart[k][cx];
        t4.x += v.x; t4.y += v.y; t4.z += v.z; t4.w += v.w;
      }
      float4* o = reinterpret_cast<float4*>(xout + c);
      const float4 old = *o;
      *o = make_float4(old.x + t4.x, old.y + t4.y, old.z + t4.z, old.w + t4.w);
    }
    return;
  }
  const int col = bx * 64 + (threadIdx.x & 63);  // in [0, nf * d)
  const int rl = threadIdx.x >> 6;                        // row lane 0..3
  const int per = (nblk + FIN_SPLIT - 1) / FIN_SPLIT;
  const int r0 = by * per;
  const int r1 = min(nblk, r0 + per);
  float s = 0.f;
  if (col < nf * d) {
    for (int r = r0 + rl; r < r1; r += 4) s += partial[(size_t)r * nf * d + col];
  }
  __shared__ float red[4][64];
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && col < nf * d) {
    s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    const int which = col / d, c = col - which * d;
    float* out = which == 0 ? dgamma : (which == 1 ? dbeta : (which == 2 ? dcolsum : out3));
    if (out != nullptr) atomicAdd(out + c, s);
  }
}

__global__ __launch_bounds__(256) void ln_bwd_finalize_kernel(const float* __restrict__ partial, int nblk, int d,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                               float* __restrict__ dcolsum, const float* __restrict__ xslab = nullptr,
                                                               int xrows = 0, int xn = 0, float* __restrict__ xout = nullptr, int nf = 3,
                                                               float* __restrict__ out3 = nullptr) {
  ln_finalize_body((int)blockIdx.x, (int)blockIdx.y, partial, nblk, d, dgamma, dbeta, dcolsum, xslab, xrows, xn, xout, nf, out3);
}

// Several finalizes in ONE launch (blockIdx.z = job): a backward pass issues one LayerNorm backward per sub-block and nothing inside
// backward reads the column sums they leave (dgamma, dbeta, bias gradients: the optimizer's inputs), so the slabs of many launches -
// each in a workspace of its own - are reduced together at the points where the gradients must be final (a data-parallel bucket
// trigger, the end of backward): 25 launches of 7.5 us + a kernel boundary each become one or a few (round 5).
constexpr int LN_FIN_JOBS = 28;
struct LnFinJobs {
  struct {
    const float* partial;
    float* out[4];
    const float* xslab;
    float* xout;
    int nblk, d, nf, xrows, xn;
  } j[LN_FIN_JOBS];
};
__global__ __launch_bounds__(256) void ln_bwd_finalize_jobs_kernel(const LnFinJobs jobs) {
  const auto& q = jobs.j[blockIdx.z];
  const int need = (q.nf * q.d + 63) / 64 + (q.xn + 63) / 64;
  if ((int)blockIdx.x >= need) return;
  ln_finalize_body((int)blockIdx.x, (int)blockIdx.y, q.partial, q.nblk, q.d, q.out[0], q.out[1], q.out[2], q.xslab, q.xrows, q.xn, q.xout, q.nf,
                   q.out[3]);
}


// LayerScale (+ stochastic depth) backward: out = res + rs[m/rps] * ls * branch  (layerscale.py:23, stochastic_depth.py:16-27)
//   dbr[m,:]  = bf16( dres[m,:] * rs * ls )                 cotangent of the bf16 branch (feeds the wgrad / dgrad GEMMs)
//   d_ls     += sum_m dres[m,:] * rs * branch[m,:]          dbias += sum_m dbr[m,:]   (bias of the Dense that produced the branch)
// Same wave-per-row / partial-slab structure as ln_bwd_kernel; slab rows are [d_ls | dbias | unused].
template <int CH>
__global__ __launch_bounds__(LN_THREADS) void layerscale_bwd_kernel(const float* __restrict__ dres, const bf16_t* __restrict__ branch,
                                                                     const float* __restrict__ ls, const float* __restrict__ rowscale,
                                                                     int rows_per_sample, bf16_t* __restrict__ dbr, float* __restrict__ partial,
                                                                     int rows, int d, long dres_stride) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int nchunk = d >> 2;
  float4 g[CH], dg[CH], db[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int ci = lane + 64 * c;
    g[c] = (ci < nchunk) ? reinterpret_cast<const float4*>(ls)[ci] : make_float4(0, 0, 0, 0);
    dg[c] = make_float4(0, 0, 0, 0);
    db[c] = make_float4(0, 0, 0, 0);
  }
  // two rows per iteration: both rows' loads are in flight before the first is used (one row per iteration left this kernel at
  // 3.2 TB/s on CaiT-S24's 50 176 x 384 launches)
  const int stride = gridDim.x * LN_WAVES;
  for (int row = blockIdx.x * LN_WAVES + wave; row < rows; row += 2 * stride) {
    const int rowb = row + stride;
    const bool hasb = rowb < rows;
    const float rsa = rowscale ? rowscale[row / rows_per_sample] : 1.0f;
    const float rsb = (rowscale && hasb) ? rowscale[rowb / rows_per_sample] : 1.0f;
    float4 dra[CH], drb[CH];
    uint2 bva[CH], bvb[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int ci = lane + 64 * c;
      dra[c] = drb[c] = make_float4(0, 0, 0, 0);
      bva[c] = bvb[c] = make_uint2(0u, 0u);
      if (ci < nchunk) {
        dra[c] = reinterpret_cast<const float4*>(dres + (size_t)row * dres_stride)[ci];
        bva[c] = reinterpret_cast<const uint2*>(branch + (size_t)row * d)[ci];
        if (hasb) {
          drb[c] = reinterpret_cast<const float4*>(dres + (size_t)rowb * dres_stride)[ci];
          bvb[c] = reinterpret_cast<const uint2*>(branch + (size_t)rowb * d)[ci];
        }
      }
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (half == 1 && !hasb) break;
      const int r = half ? rowb : row;
      const float rs = half ? rsb : rsa;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int ci = lane + 64 * c;
        if (ci < nchunk) {
          const float4 dr = half ? drb[c] : dra[c];
          const uint2 bv = half ? bvb[c] : bva[c];
          const float b0 = __uint_as_float(bv.x << 16), b1 = __uint_as_float(bv.x & 0xffff0000u);
          const float b2 = __uint_as_float(bv.y << 16), b3 = __uint_as_float(bv.y & 0xffff0000u);
          const float o0 = round_bf16(dr.x * rs * g[c].x), o1 = round_bf16(dr.y * rs * g[c].y);
          const float o2 = round_bf16(dr.z * rs * g[c].z), o3 = round_bf16(dr.w * rs * g[c].w);
          reinterpret_cast<uint2*>(dbr + (size_t)r * d)[ci] = make_uint2(pack_bf16x2(o0, o1), pack_bf16x2(o2, o3));
          dg[c].x += dr.x * rs * b0; dg[c].y += dr.y * rs * b1; dg[c].z += dr.z * rs * b2; dg[c].w += dr.w * rs * b3;
          db[c].x += o0; db[c].y += o1; db[c].z += o2; db[c].w += o3;
        }
      }
    }
  }
  __shared__ float4 red[LN_WAVES][64];
#pragma unroll
  for (int which = 0; which < 3; ++which) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const float4 v = which == 0 ? dg[c] : (which == 1 ? db[c] : make_float4(0, 0, 0, 0));
      __syncthreads();
      red[wave][lane] = v;
      __syncthreads();
      if (wave == 0) {
        float4 t = red[0][lane];
#pragma unroll
        for (int w = 1; w < LN_WAVES; ++w) {
          t.x += red[w][lane].x; t.y += red[w][lane].y; t.z += red[w][lane].z; t.w += red[w][lane].w;
        }
        const int ci = lane + 64 * c;
        if (ci < nchunk) reinterpret_cast<float4*>(partial + ((size_t)blockIdx.x * 3 + which) * d)[ci] = t;
      }
    }
  }
}


// LayerNorm backward with the LayerScale (+ stochastic depth) backward of the sub-block whose output cotangent it produces (CaiT: the
// residual gradient this kernel writes is exactly what layerscale_bwd_kernel would read next, cait.py:28-60 in reverse):
//   out = LN-VJP(dy) + dres_in  -> fp32 dx_out;   dbr = bf16(out * rs * ls);   d_ls += sum_rows out * rs * branch;   dbias += sum_rows dbr
// One pass over the residual gradient instead of two (round 4: the separate launch re-read 4 + 2 B and wrote 2 B per element, 30 + 5 us
// per sub-block at CaiT-S24).  Slab rows are [dgamma | dbeta | d_ls | dbias] (4 families); same arithmetic per element as the two kernels.
template <int CH>
__global__ __launch_bounds__(LN_THREADS) void ln_bwd_ls_kernel(const bf16_t* __restrict__ dy, const float* __restrict__ x,
                                                                const float* __restrict__ gamma, const float* __restrict__ mean_in,
                                                                const float* __restrict__ rstd_in, const float* dres_in,
                                                                float* dx_out, float* __restrict__ partial, int rows, int d, long x_stride,
                                                                long out_stride, int round_params, const bf16_t* __restrict__ branch,
                                                                const float* __restrict__ ls, const float* __restrict__ rowscale,
                                                                int rows_per_sample, bf16_t* __restrict__ dbr) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int nchunk = d >> 2;
  float4 g[CH], gl[CH], dg[CH], db[CH], dl[CH], dbl[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int ci = lane + 64 * c;
    g[c] = (ci < nchunk) ? reinterpret_cast<const float4*>(gamma)[ci] : make_float4(0, 0, 0, 0);
    if (round_params) g[c] = make_float4(round_bf16(g[c].x), round_bf16(g[c].y), round_bf16(g[c].z), round_bf16(g[c].w));
    gl[c] = (ci < nchunk) ? reinterpret_cast<const float4*>(ls)[ci] : make_float4(0, 0, 0, 0);
    dg[c] = db[c] = dl[c] = dbl[c] = make_float4(0, 0, 0, 0);
  }
  const float inv_d = 1.0f / (float)d;
  float4 xv_n[CH], rs_n[CH];
  uint2 dv_n[CH], bv_n[CH];
  float mean_n = 0.f, rstd_n = 0.f, rsc_n = 1.f;
  auto load_row = [&](int row) {
    const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * x_stride);
    const uint2* dyr = reinterpret_cast<const uint2*>(dy + (size_t)row * d);
    const uint2* brr = reinterpret_cast<const uint2*>(branch + (size_t)row * d);
    mean_n = mean_in[row];
    rstd_n = rstd_in[row];
    rsc_n = rowscale ? rowscale[row / rows_per_sample] : 1.0f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int ci = lane + 64 * c;
      xv_n[c] = make_float4(0, 0, 0, 0);
      rs_n[c] = make_float4(0, 0, 0, 0);
      dv_n[c] = make_uint2(0u, 0u);
      bv_n[c] = make_uint2(0u, 0u);
      if (ci < nchunk) {
        xv_n[c] = xr[ci];
        dv_n[c] = dyr[ci];
        bv_n[c] = brr[ci];
        if (dres_in) rs_n[c] = reinterpret_cast<const float4*>(dres_in + (size_t)row * out_stride)[ci];
      }
    }
  };
  const int row_step = gridDim.x * LN_WAVES;
  int row = blockIdx.x * LN_WAVES + wave;
  if (row < rows) load_row(row);
  for (; row < rows; row += row_step) {
    const float mean = mean_n, rstd = rstd_n, rsc = rsc_n;
    float4 xh[CH], gy[CH], rs[CH];
    uint2 dvc[CH], bvc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      xh[c] = xv_n[c];
      rs[c] = rs_n[c];
      dvc[c] = dv_n[c];
      bvc[c] = bv_n[c];
    }
    if (row + row_step < rows) load_row(row + row_step);
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int ci = lane + 64 * c;
      if (ci < nchunk) {
        const float4 xv = xh[c];
        const uint2 dv = dvc[c];
        const float d0 = __uint_as_float(dv.x << 16), d1 = __uint_as_float(dv.x & 0xffff0000u);
        const float d2 = __uint_as_float(dv.y << 16), d3 = __uint_as_float(dv.y & 0xffff0000u);
        xh[c] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
        gy[c] = make_float4(d0 * g[c].x, d1 * g[c].y, d2 * g[c].z, d3 * g[c].w);
        dg[c].x += d0 * xh[c].x; dg[c].y += d1 * xh[c].y; dg[c].z += d2 * xh[c].z; dg[c].w += d3 * xh[c].w;
        db[c].x += d0; db[c].y += d1; db[c].z += d2; db[c].w += d3;
        c1 += (gy[c].x + gy[c].y) + (gy[c].z + gy[c].w);
        c2 += (gy[c].x * xh[c].x + gy[c].y * xh[c].y) + (gy[c].z * xh[c].z + gy[c].w * xh[c].w);
      } else {
        xh[c] = make_float4(0, 0, 0, 0);
        gy[c] = make_float4(0, 0, 0, 0);
      }
    }
    c1 = wave_sum(c1) * inv_d;
    c2 = wave_sum(c2) * inv_d;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int ci = lane + 64 * c;
      if (ci < nchunk) {
        float4 o = make_float4(rstd * (gy[c].x - c1 - xh[c].x * c2), rstd * (gy[c].y - c1 - xh[c].y * c2),
                               rstd * (gy[c].z - c1 - xh[c].z * c2), rstd * (gy[c].w - c1 - xh[c].w * c2));
        o.x += rs[c].x; o.y += rs[c].y; o.z += rs[c].z; o.w += rs[c].w;
        reinterpret_cast<float4*>(dx_out + (size_t)row * out_stride)[ci] = o;
        const uint2 bv = bvc[c];
        const float b0 = __uint_as_float(bv.x << 16), b1 = __uint_as_float(bv.x & 0xffff0000u);
        const float b2 = __uint_as_float(bv.y << 16), b3 = __uint_as_float(bv.y & 0xffff0000u);
        const float o0 = round_bf16(o.x * rsc * gl[c].x), o1 = round_bf16(o.y * rsc * gl[c].y);
        const float o2 = round_bf16(o.z * rsc * gl[c].z), o3 = round_bf16(o.w * rsc * gl[c].w);
        reinterpret_cast<uint2*>(dbr + (size_t)row * d)[ci] = make_uint2(pack_bf16x2(o0, o1), pack_bf16x2(o2, o3));
        dl[c].x += o.x * rsc * b0; dl[c].y += o.y * rsc * b1; dl[c].z += o.z * rsc * b2; dl[c].w += o.w * rsc * b3;
        dbl[c].x += o0; dbl[c].y += o1; dbl[c].z += o2; dbl[c].w += o3;
      }
    }
  }
  __shared__ float4 red[LN_WAVES][64];
#pragma unroll
  for (int which = 0; which < 4; ++which) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const float4 v = which == 0 ? dg[c] : (which == 1 ? db[c] : (which == 2 ? dl[c] : dbl[c]));
      __syncthreads();
      red[wave][lane] = v;
      __syncthreads();
      if (wave == 0) {
        float4 t = red[0][lane];
#pragma unroll
        for (int w = 1; w < LN_WAVES; ++w) {
          t.x += red[w][lane].x; t.y += red[w][lane].y; t.z += red[w][lane].z; t.w += red[w][lane].w;
        }
        const int ci = lane + 64 * c;
        if (ci < nchunk) reinterpret_cast<float4*>(partial + ((size_t)blockIdx.x * 4 + which) * d)[ci] = t;
      }
    }
  }
}


}  // namespace

static int ln_bwd_grid(int rows) { return ln_grid(rows, 256 * 3); }
extern "C" int savit_layernorm_bwd_mapped(const void* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                          const float* dres_in, float* dx, void* dx_bf16, float* dgamma, float* dbeta, float* dcolsum, int rows,
                                          int d, long x_stride, long out_stride, int round_params_bf16, int dy_grp, int dy_grp_stride,
                                          int dy_grp_off, void* workspace, long workspace_bytes, void* stream);

extern "C" int savit_layernorm_bwd_grid(int rows) { return rows > 0 ? ln_bwd_grid(rows) : 0; }

extern "C" int savit_layernorm_bwd_finalize_jobs(const savit_colsum_job* jobs, int count, void* stream) {
  SAVIT_CHECK_ARG(jobs != nullptr && count >= 0);
  for (int base = 0; base < count; base += LN_FIN_JOBS) {
    LnFinJobs J{};
    const int n = count - base < LN_FIN_JOBS ? count - base : LN_FIN_JOBS;
    int gx = 1;
    for (int i = 0; i < n; ++i) {
      const savit_colsum_job& q = jobs[base + i];
      SAVIT_CHECK_ARG(q.partial && q.nblk > 0 && q.d > 0 && (q.nf == 3 || q.nf == 4) && ((uintptr_t)q.partial % 16) == 0);
      SAVIT_CHECK_ARG(q.extra_slab == nullptr || (q.extra_out && q.extra_rows >= 0 && q.extra_n > 0 && q.extra_n % 4 == 0 &&
                                                  ((uintptr_t)q.extra_slab % 16) == 0 && ((uintptr_t)q.extra_out % 16) == 0));
      auto& o = J.j[i];
      o.partial = q.partial; o.nblk = q.nblk; o.d = q.d; o.nf = q.nf;
      for (int k = 0; k < 4; ++k) o.out[k] = q.out[k];
      o.xslab = q.extra_slab; o.xout = q.extra_out; o.xrows = q.extra_rows; o.xn = q.extra_slab ? q.extra_n : 0;
      const int need = (q.nf * q.d + 63) / 64 + (o.xn + 63) / 64;
      if (need > gx) gx = need;
    }
    hipLaunchKernelGGL(ln_bwd_finalize_jobs_kernel, dim3(gx, FIN_SPLIT, n), dim3(256), 0, (hipStream_t)stream, J);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  }
  return SAVIT_OK;
}

extern "C" long savit_layernorm_bwd_workspace_bytes(int rows, int d) {
  if (rows <= 0 || d <= 0) return 0;
  return (long)ln_bwd_grid(rows) * 4 * d * (long)sizeof(float);  // 4 families of partial column sums (savit_layernorm_bwd_ls; 3 otherwise)
}

extern "C" int savit_layernorm_bwd(const void* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                   const float* dres_in, float* dx, void* dx_bf16, float* dgamma, float* dbeta, float* dcolsum,
                                   int rows, int d, long x_stride, long out_stride, int round_params_bf16, void* workspace,
                                   long workspace_bytes, void* stream) {
  return savit_layernorm_bwd_mapped(dy, x, gamma, mean, rstd, dres_in, dx, dx_bf16, dgamma, dbeta, dcolsum, rows, d, x_stride, out_stride,
                                    round_params_bf16, 0, 0, 0, workspace, workspace_bytes, stream);
}

extern "C" int savit_layernorm_bwd_ex(const void* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                      const float* dres_in, float* dx, void* dx_bf16, float* dgamma, float* dbeta, float* dcolsum,
                                      int rows, int d, long x_stride, long out_stride, int round_params_bf16, void* workspace,
                                      long workspace_bytes, const float* extra_slab, int extra_rows, int extra_n, float* extra_out,
                                      void* stream) {
  SAVIT_CHECK_ARG(dy && x && gamma && mean && rstd && dx && x_stride >= d && out_stride >= d && (x_stride % 4) == 0 &&
                  (out_stride % 4) == 0 && rows > 0 && d > 64 && (d % 4) == 0 && d <= 64 * 4 * LN_MAX_CHUNKS);
  SAVIT_CHECK_ARG(extra_slab && extra_out && extra_rows >= 0 && extra_n > 0 && extra_n % 4 == 0 && (dgamma || dbeta || dcolsum) &&
                  ((uintptr_t)extra_slab % 16) == 0 && ((uintptr_t)extra_out % 16) == 0);
  SAVIT_CHECK_ARG(workspace != nullptr && ((uintptr_t)workspace % 16) == 0 &&
                  workspace_bytes >= savit_layernorm_bwd_workspace_bytes(rows, d));
  hipStream_t s = (hipStream_t)stream;
  const int ch = (d / 4 + 63) / 64;
  float* partial = (float*)workspace;
  const int grid = ln_bwd_grid(rows);
  LN_DISPATCH(ch, ln_bwd_kernel, grid, (const bf16_t*)dy, x, gamma, mean, rstd, dres_in, dx, (bf16_t*)dx_bf16, partial, rows, d,
              x_stride, out_stride, round_params_bf16, 0, 0, 0);
  hipLaunchKernelGGL(ln_bwd_finalize_kernel, dim3((3 * d + 63) / 64 + (extra_n + 63) / 64, FIN_SPLIT), dim3(256), 0, s, partial, grid, d,
                     dgamma, dbeta, dcolsum, extra_slab, extra_rows, extra_n, extra_out);
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_layernorm_bwd_sparse(const void* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                          int stat_stride, const float* dres_in, int res_mod, long res_stride, float* dx, void* dx_bf16,
                                          float* dgamma, float* dbeta, float* dcolsum, int rows, int d, long x_stride, long out_stride,
                                          int round_params_bf16, void* workspace, long workspace_bytes, const float* extra_slab,
                                          int extra_rows, int extra_n, float* extra_out, void* stream) {
  SAVIT_CHECK_ARG(dy && x && gamma && mean && rstd && dx && x_stride >= d && out_stride >= d && (x_stride % 4) == 0 &&
                  (out_stride % 4) == 0 && rows > 0 && d > 64 && (d % 4) == 0 && d <= 64 * 4 * LN_MAX_CHUNKS && stat_stride >= 1);
  SAVIT_CHECK_ARG(res_mod >= 0 && (res_mod == 0 || (dres_in != nullptr && res_stride >= d && (res_stride % 4) == 0)));
  SAVIT_CHECK_ARG(extra_slab == nullptr || (extra_out && extra_rows >= 0 && extra_n > 0 && extra_n % 4 == 0 && (dgamma || dbeta || dcolsum) &&
                                            ((uintptr_t)extra_slab % 16) == 0 && ((uintptr_t)extra_out % 16) == 0));
  SAVIT_CHECK_ARG(workspace != nullptr && ((uintptr_t)workspace % 16) == 0 && workspace_bytes >= savit_layernorm_bwd_workspace_bytes(rows, d));
  hipStream_t s = (hipStream_t)stream;
  const int ch = (d / 4 + 63) / 64;
  float* partial = (float*)workspace;
  const int grid = ln_bwd_grid(rows);
  LN_DISPATCH(ch, ln_bwd_kernel, grid, (const bf16_t*)dy, x, gamma, mean, rstd, dres_in, dx, (bf16_t*)dx_bf16, partial, rows, d, x_stride,
              out_stride, round_params_bf16, 0, 0, 0, res_mod, res_stride, stat_stride);
  if (dgamma || dbeta || dcolsum) {
    const int xn = extra_slab ? extra_n : 0;
    hipLaunchKernelGGL(ln_bwd_finalize_kernel, dim3((3 * d + 63) / 64 + (xn + 63) / 64, FIN_SPLIT), dim3(256), 0, s, partial, grid, d, dgamma,
                       dbeta, dcolsum, extra_slab, extra_rows, xn, extra_out);
  }
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_layernorm_bwd_mapped(const void* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                          const float* dres_in, float* dx, void* dx_bf16, float* dgamma, float* dbeta, float* dcolsum, int rows,
                                          int d, long x_stride, long out_stride, int round_params_bf16, int dy_grp, int dy_grp_stride,
                                          int dy_grp_off, void* workspace, long workspace_bytes, void* stream) {
  SAVIT_CHECK_ARG(dy_grp == 0 || (dy_grp > 0 && dy_grp_stride >= dy_grp && dy_grp_off >= 0 && dy_grp_off + dy_grp <= dy_grp_stride));
  SAVIT_CHECK_ARG(dy && x && gamma && mean && rstd && dx && x_stride >= d && out_stride >= d && (x_stride % 4) == 0 &&
                  (out_stride % 4) == 0 && rows >= 0 && d > 0 && (d % 4) == 0 && d <= 64 * 4 * LN_MAX_CHUNKS);
  if (rows == 0) return SAVIT_OK;
  SAVIT_CHECK_ARG(workspace != nullptr && ((uintptr_t)workspace % 16) == 0 &&
                  workspace_bytes >= savit_layernorm_bwd_workspace_bytes(rows, d));
  hipStream_t s = (hipStream_t)stream;
  const int ch = (d / 4 + 63) / 64;
  float* partial = (float*)workspace;
  if (d <= 64 && dy_grp == 0) {  // narrow rows; its grid never exceeds ln_bwd_grid(rows), which sized the workspace
    const int g16 = (rows + 15) / 16, cap = ln_bwd_grid(rows);
    const int ngrid = g16 < cap ? g16 : cap;
    hipLaunchKernelGGL(ln_bwd_narrow_kernel, dim3(ngrid), dim3(LN_THREADS), 0, s, (const bf16_t*)dy, x, gamma, mean, rstd, dres_in, dx,
                       (bf16_t*)dx_bf16, partial, rows, d, x_stride, out_stride, round_params_bf16);
    if (dgamma || dbeta || dcolsum)
      hipLaunchKernelGGL(ln_bwd_finalize_kernel, dim3((3 * d + 63) / 64, FIN_SPLIT), dim3(256), 0, s, partial, ngrid, d, dgamma, dbeta, dcolsum);
    SAVIT_LAUNCH_RET();
  }
  const int grid = ln_bwd_grid(rows);
  LN_DISPATCH(ch, ln_bwd_kernel, grid, (const bf16_t*)dy, x, gamma, mean, rstd, dres_in, dx, (bf16_t*)dx_bf16, partial, rows, d,
              x_stride, out_stride, round_params_bf16, dy_grp, dy_grp_stride, dy_grp_off);
  if (dgamma || dbeta || dcolsum) {
    hipLaunchKernelGGL(ln_bwd_finalize_kernel, dim3((3 * d + 63) / 64, FIN_SPLIT), dim3(256), 0, s, partial, grid, d, dgamma,
                       dbeta, dcolsum);
  }
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_layerscale_bwd(const float* dres, const void* branch_bf16, const float* layerscale, const float* rowscale,
                                    int rows_per_sample, void* dbranch_bf16, float* d_layerscale, float* dbias, int rows, int d,
                                    long dres_stride, void* workspace, long workspace_bytes, void* stream) {
  SAVIT_CHECK_ARG(dres && branch_bf16 && layerscale && dbranch_bf16 && (d_layerscale || !dbias) && rows >= 0 && d > 0 && (d % 4) == 0 &&
                  d <= 64 * 4 * LN_MAX_CHUNKS && dres_stride >= d && (dres_stride % 4) == 0 && (rowscale == nullptr || rows_per_sample >= 1));
  if (rows == 0) return SAVIT_OK;
  SAVIT_CHECK_ARG(workspace != nullptr && ((uintptr_t)workspace % 16) == 0 && workspace_bytes >= savit_layernorm_bwd_workspace_bytes(rows, d));
  hipStream_t s = (hipStream_t)stream;
  const int ch = (d / 4 + 63) / 64;
  const int grid = ln_bwd_grid(rows);
  float* partial = (float*)workspace;
  LN_DISPATCH(ch, layerscale_bwd_kernel, grid, dres, (const bf16_t*)branch_bf16, layerscale, rowscale, rows_per_sample, (bf16_t*)dbranch_bf16,
              partial, rows, d, dres_stride);
  if (d_layerscale == nullptr) SAVIT_LAUNCH_RET();  // deferred: savit_layernorm_bwd_finalize_jobs reduces the slab (out[0] = d_layerscale, out[1] = dbias)
  hipLaunchKernelGGL(ln_bwd_finalize_kernel, dim3((3 * d + 63) / 64, FIN_SPLIT), dim3(256), 0, s, partial, grid, d, d_layerscale, dbias,
                     (float*)nullptr);
  SAVIT_LAUNCH_RET();
}

extern "C" int savit_layernorm_bwd_ls(const void* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                      const float* dres_in, float* dx, float* dgamma, float* dbeta, int rows, int d, long x_stride,
                                      long out_stride, int round_params_bf16, const void* branch_bf16, const float* layerscale,
                                      const float* rowscale, int rows_per_sample, void* dbranch_bf16, float* d_layerscale, float* dbias,
                                      void* workspace, long workspace_bytes, const float* extra_slab, int extra_rows, int extra_n,
                                      float* extra_out, void* stream) {
  SAVIT_CHECK_ARG(dy && x && gamma && mean && rstd && dx && x_stride >= d && out_stride >= d && (x_stride % 4) == 0 &&
                  (out_stride % 4) == 0 && rows >= 0 && d > 64 && (d % 4) == 0 && d <= 64 * 4 * LN_MAX_CHUNKS);
  const bool deferred = !dgamma && !dbeta && !d_layerscale && !dbias;  // the caller reduces the slab later (savit_layernorm_bwd_finalize_jobs)
  SAVIT_CHECK_ARG(branch_bf16 && layerscale && dbranch_bf16 && (d_layerscale || deferred) && (rowscale == nullptr || rows_per_sample >= 1));
  SAVIT_CHECK_ARG(extra_slab == nullptr || (!deferred && extra_out && extra_rows >= 0 && extra_n > 0 && extra_n % 4 == 0 && ((uintptr_t)extra_slab % 16) == 0 &&
                                            ((uintptr_t)extra_out % 16) == 0));
  if (rows == 0) return SAVIT_OK;
  SAVIT_CHECK_ARG(workspace != nullptr && ((uintptr_t)workspace % 16) == 0 && workspace_bytes >= savit_layernorm_bwd_workspace_bytes(rows, d));
  hipStream_t s = (hipStream_t)stream;
  const int ch = (d / 4 + 63) / 64;
  const int grid = ln_bwd_grid(rows);
  float* partial = (float*)workspace;
  LN_DISPATCH(ch, ln_bwd_ls_kernel, grid, (const bf16_t*)dy, x, gamma, mean, rstd, dres_in, dx, partial, rows, d, x_stride, out_stride,
              round_params_bf16, (const bf16_t*)branch_bf16, layerscale, rowscale, rows_per_sample, (bf16_t*)dbranch_bf16);
  if (deferred) SAVIT_LAUNCH_RET();
  const int xn = extra_slab ? extra_n : 0;  // the extra slab's column sums ride along as in savit_layernorm_bwd_ex
  hipLaunchKernelGGL(ln_bwd_finalize_kernel, dim3((4 * d + 63) / 64 + (xn + 63) / 64, FIN_SPLIT), dim3(256), 0, s, partial, grid, d, dgamma, dbeta,
                     d_layerscale, extra_slab, extra_rows, xn, extra_out, 4, dbias);
  SAVIT_LAUNCH_RET();
}
