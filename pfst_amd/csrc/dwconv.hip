// Depthwise 3x3 convolution, stride 1, padding = dilation: the HBM-bound dilated stencils of the ASPP head.
// Reference: mmcv DepthwiseSeparableConvModule.depthwise_conv inside
// rsiseg/models/decode_heads/sep_aspp_head.py:17-26 (ASPP, dilation 12/24/36) and :63-77 (sep_bottleneck).
//
// Design (MI355X): one workgroup owns a strip of output rows of one (image, channel) plane and stages the strip
// plus its +-dilation row halo in LDS with ONE contiguous, 16-byte-vectorised read (full rows are contiguous in NCHW,
// so there is no index arithmetic in the copy).  With 160 KB of LDS per CU a 128x128 fp32 plane (64 KB) is staged
// whole, which is what makes dilation 12/24/36 cheap: the halo (2d rows) would otherwise exceed the strip.  Every
// input element is fetched from HBM exactly once; all 9 taps are LDS reads (ds_read_b128 when dilation % 4 == 0).
// Algorithmic traffic = read plane + write plane = 8 B per output element.
// The data gradient is the same stencil with mirrored taps; the weight gradient reuses the staging and reduces
// 9 partial sums per block (wavefront shuffles, one atomic per tap per block).
#include <stdlib.h>
#include "common.h"
#include "../../include/pfst_hip.h"

namespace {

constexpr int DW_LDS_BYTES = 64 * 1024;

struct Strip { int y0, y1, lo, hi; };   // output rows [y0,y1), staged rows [lo,hi)

__device__ __forceinline__ Strip make_strip(int strip_idx, int R, int H, int dil) {
  Strip s;
  s.y0 = strip_idx * R;
  s.y1 = min(H, s.y0 + R);
  s.lo = max(0, s.y0 - dil);
  s.hi = min(H, s.y1 + dil);
  return s;
}

// y = relu(x * sc + sh): the producing conv -> BN -> ReLU layer's normalisation applied on load (bnl), with bn_apply's pinned roundings
__device__ __forceinline__ float bn_on_load(float v, float sc, float sh) { return fmaxf(__fmaf_rn(v, sc, sh), 0.f); }
__device__ __forceinline__ float4 bn_on_load4(float4 v, float sc, float sh) {
  return make_float4(bn_on_load(v.x, sc, sh), bn_on_load(v.y, sc, sh), bn_on_load(v.z, sc, sh), bn_on_load(v.w, sc, sh));
}

// contiguous copy of rows [lo,hi) of one plane into LDS; bnl: the rows hold the PRE-normalisation tensor of the layer that feeds this
// convolution and are normalised + rectified while they are staged (the normalised tensor itself is never written)
__device__ __forceinline__ void stage_rows(const float* __restrict__ plane, float* __restrict__ tile, int lo, int hi, int W, bool bnl = false,
                                           float sc = 1.f, float sh = 0.f) {
  const int n = (hi - lo) * W;
  const float* src = plane + (i64)lo * W;
  if ((W & 3) == 0 && (((uintptr_t)src) & 15) == 0) {
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* t4 = reinterpret_cast<float4*>(tile);
    // eight 16-byte loads in flight per thread before the first LDS store (a 128x128 plane is exactly one batch of 512 threads): the
    // plain copy loop compiles to load -> wait -> store, 16 KB in flight per CU, and is latency-bound at half the HBM rate
    // (buffer loads: an offset past the staged rows returns zeros instead of needing a branch around the load)
    const int n4 = n >> 2, bd = blockDim.x;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(s4), 0, n4 * 16, 0x00020000);
    for (int i0 = threadIdx.x; i0 < n4; i0 += 8 * bd) {
      float4 r[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) r[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, 16 * (i0 + u * bd), 0, 0));
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + u * bd < n4) t4[i0 + u * bd] = bnl ? bn_on_load4(r[u], sc, sh) : r[u];
    }
  } else {
    for (int i = threadIdx.x; i < n; i += blockDim.x) tile[i] = bnl ? bn_on_load(src[i], sc, sh) : src[i];
  }
}

// the second pass of BatchNorm backward for one element, exactly as bn_bwd_apply_kernel evaluates it (ReLU gate recomputed from the
// pre-normalisation value like bn_apply decided it, projections subtracted in fp64)
__device__ __forceinline__ float bn_bwd_elem(float g, float xv, const pfst_bn_bwd_rec_t& r) {
  const float dz = __fmaf_rn(xv, r.sc, r.sh) > 0.f ? g : 0.f;
#ifdef PFST_DIAG_DW_F32
  return (float)r.gs * (dz - (float)r.m1 - ((xv - r.mu) * r.is) * (float)r.m2);
#endif
  return (float)(r.gs * ((double)dz - r.m1 - (((double)xv - (double)r.mu) * (double)r.is) * r.m2));
}
__device__ __forceinline__ float4 bn_bwd_elem4(float4 g, float4 xv, const pfst_bn_bwd_rec_t& r) {
  return make_float4(bn_bwd_elem(g.x, xv.x, r), bn_bwd_elem(g.y, xv.y, r), bn_bwd_elem(g.z, xv.z, r), bn_bwd_elem(g.w, xv.w, r));
}

// rows [lo,hi) of dL/dpre = BatchNorm-backward(dy, pre) of one plane into LDS: the depthwise layer's backward stages the gradient of its own
// convolution output without that tensor ever being written (pfst_bn_backward_sums left the per-channel record)
__device__ __forceinline__ void stage_rows_bnbwd(const float* __restrict__ dyp, const float* __restrict__ prep, float* __restrict__ tile, int lo,
                                                 int hi, int W, const pfst_bn_bwd_rec_t& rec) {
  const int n = (hi - lo) * W;
  const float* g = dyp + (i64)lo * W;
  const float* x = prep + (i64)lo * W;
  if ((W & 3) == 0 && ((((uintptr_t)g) | ((uintptr_t)x)) & 15) == 0) {
    float4* t4 = reinterpret_cast<float4*>(tile);
    const int n4 = n >> 2, bd = blockDim.x;
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g), 0, n4 * 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, n4 * 16, 0x00020000);
    for (int i0 = threadIdx.x; i0 < n4; i0 += 4 * bd) {
      float4 a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(gr, 16 * (i0 + u * bd), 0, 0));
        b[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, 16 * (i0 + u * bd), 0, 0));
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (i0 + u * bd < n4) t4[i0 + u * bd] = bn_bwd_elem4(a[u], b[u], rec);
    }
  } else {
    for (int i = threadIdx.x; i < n; i += blockDim.x) tile[i] = bn_bwd_elem(g[i], x[i], rec);
  }
}

// load 4 consecutive floats starting at column sx (any alignment, zero outside [0,W))
template <bool ALIGNED>
__device__ __forceinline__ float4 row4(const float* __restrict__ row, int sx, int W) {
  if (ALIGNED) {
    if (sx < 0 || sx >= W) return make_float4(0.f, 0.f, 0.f, 0.f);
    return *reinterpret_cast<const float4*>(row + sx);
  }
  float4 v;
  v.x = (sx >= 0 && sx < W) ? row[sx] : 0.f;
  v.y = (sx + 1 >= 0 && sx + 1 < W) ? row[sx + 1] : 0.f;
  v.z = (sx + 2 >= 0 && sx + 2 < W) ? row[sx + 2] : 0.f;
  v.w = (sx + 3 >= 0 && sx + 3 < W) ? row[sx + 3] : 0.f;
  return v;
}

// dilation 1: the three taps of one row from ONE aligned 16-byte read plus the two neighbouring elements (3 LDS reads instead of 12)
__device__ __forceinline__ void row_taps_d1(const float* __restrict__ row, int c4, int W, float4& vl, float4& vc, float4& vr) {
  vc = *reinterpret_cast<const float4*>(row + c4 * 4);
  const float lw = c4 > 0 ? row[c4 * 4 - 1] : 0.f;
  const float rx = c4 * 4 + 4 < W ? row[c4 * 4 + 4] : 0.f;
  vl = make_float4(lw, vc.x, vc.y, vc.z);
  vr = make_float4(vc.y, vc.z, vc.w, rx);
}

// nine per-thread partial sums -> nine atomics per workgroup (wave shuffles, then one value per wave through `scratch`: >= 8 x 9 floats;
// every thread of the workgroup calls it; the scratch may be reused after the call)
__device__ __forceinline__ void block_add9(const float (&acc)[9], float* __restrict__ dst, float* __restrict__ scratch) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();                                 // the scratch may still be read by a previous reduction
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const float v = wave_sum(acc[t]);
    if (lane == 0) scratch[wid * 9 + t] = v;
  }
  __syncthreads();
  if (threadIdx.x < 9) {
    float v = 0.f;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) v += scratch[k * 9 + threadIdx.x];
    atomicAdd(&dst[threadIdx.x], v);
  }
}

// Deterministic mode (api.cpp): det_T > 0 = the weight gradient of channel c goes to slot `slot` of a zeroed [C][det_T][9] scratch instead of
// being added into dw[c] by every workgroup of the channel (each slot has ONE writer); dw_det_reduce_kernel then adds the slots in index order.
__device__ __forceinline__ float* dw_dst(float* __restrict__ dw, int c, int det_T, int slot) {
  return det_T > 0 ? dw + ((i64)c * det_T + slot) * 9 : dw + c * 9;
}
__global__ __launch_bounds__(64) void dw_det_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int C, int T) {
  const int c = blockIdx.x, t = threadIdx.x;
  if (c >= C || t >= 9) return;
  float v = 0.f;
  for (int k = 0; k < T; ++k) v += part[((i64)c * T + k) * 9 + t];
  dw[c * 9 + t] += v;
}

// block-wide (minimum, maximum) of per-thread running extrema: thread 0 gets the result (scratch: >= 32 floats of LDS, barriers inside)
__device__ __forceinline__ void block_minmax(float& lo, float& hi, float* scratch) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, o));
    hi = fmaxf(hi, __shfl_xor(hi, o));
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) { scratch[wid] = lo; scratch[16 + wid] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int i = 1; i < nw; ++i) { lo = fminf(lo, scratch[i]); hi = fmaxf(hi, scratch[16 + i]); }
  }
}

// MODE 0: scalar (any W); MODE 1: float4 outputs, dilation % 4 == 0 (ds_read_b128 taps); MODE 2: float4 outputs, any dilation;
// MODE 3: float4 outputs, dilation 1 (row_taps_d1)
// WG (backward only, flip = 1: x = dY, y = dX): the same pass also forms the WEIGHT gradient.  With v_t = dY[p + off(t)] the nine
// neighbours the data gradient reads for output position p,  dW[8 - t] = sum_p X[p] v_t(p)  (substitute p = q + off in
// dW[t'] = sum_q dY[q] X[q + off(t')]): one more 16-byte load (the layer's forward input X at p) and 9 x 4 fused multiply-adds per
// output quad instead of a second kernel that stages X and re-reads dY -- 3 N of traffic for the two gradients instead of 4 N.
template <int MODE, bool WG = false>
__global__ __launch_bounds__(512) void dwconv3x3_kernel(const float* __restrict__ x, i64 x_bs, const float* __restrict__ w,
                                                        float* __restrict__ y, i64 y_bs, int C, int H, int W, int dil, int R,
                                                        int flip, int accumulate, float* __restrict__ stats,
                                                        const float* __restrict__ fx = nullptr, i64 fx_bs = 0, float* __restrict__ dw = nullptr,
                                                        const float4* __restrict__ bnl = nullptr, const float* __restrict__ bnpre = nullptr,
                                                        i64 bnpre_bs = 0, const pfst_bn_bwd_rec_t* __restrict__ bnrec = nullptr, int det_T = 0,
                                                        int stats_minmax = 0) {
  // bnrec != NULL (WG): x = the gradient of this layer's BatchNorm + ReLU output, bnpre = the layer's own convolution output: the rows
  // staged are dL/dpre = the second pass of BatchNorm backward, formed on the fly (stage_rows_bnbwd)
  // bnl != NULL: coef[C] = (mean, invstd, sc, sh) of the conv -> BN -> ReLU layer whose PRE-normalisation output is this convolution's
  // forward input: forward (WG = false) x is that tensor and is normalised while staged; backward (WG) fx is, and its quads are
  // normalised as they are loaded.  The normalised tensor is never materialised (its only consumer is this depthwise layer).
  extern __shared__ float tile[];
  __shared__ double red[40];                       // (also the 8 x 9 floats of block_add9)
  float st_s = 0.f, st_q = 0.f;                    // fused BatchNorm statistics of this block's outputs (stats != NULL)
  float st_lo = __builtin_inff(), st_hi = -__builtin_inff();      // stats_minmax: and their extrema (predicted max |relu(bn(y))|, bn.hip)
  const int c = blockIdx.y, n = blockIdx.z;
  const Strip s = make_strip(blockIdx.x, R, H, dil);
  const float* xp = x + (i64)n * x_bs + (i64)c * H * W;
  float* yp = y + (i64)n * y_bs + (i64)c * H * W;
  const float* fxp = WG ? fx + (i64)n * fx_bs + (i64)c * H * W : nullptr;
  float accw[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) accw[t] = 0.f;
  float wt[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wt[t] = w[c * 9 + (flip ? 8 - t : t)];
  const float bsc = bnl ? bnl[c].z : 1.f, bsh = bnl ? bnl[c].w : 0.f;
  if (WG && bnrec) {
    const pfst_bn_bwd_rec_t rec = bnrec[c];
    stage_rows_bnbwd(xp, bnpre + (i64)n * bnpre_bs + (i64)c * H * W, tile, s.lo, s.hi, W, rec);
  } else {
    stage_rows(xp, tile, s.lo, s.hi, W, bnl != nullptr && !WG, bsc, bsh);
  }
  __syncthreads();
  if (MODE != 0) {
    const int W4 = W >> 2;
    const int total = (s.y1 - s.y0) * W4;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
      const int r = i / W4, c4 = i - r * W4;
      const int yy = s.y0 + r;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      float4 xq = make_float4(0.f, 0.f, 0.f, 0.f);
      if (WG) {
        xq = *(reinterpret_cast<const float4*>(fxp + (i64)yy * W) + c4);
        if (bnl) xq = bn_on_load4(xq, bsc, bsh);
      }
#pragma unroll
      for (int ty = 0; ty < 3; ++ty) {
        const int sy = yy + (ty - 1) * dil;
        if (sy < 0 || sy >= H) continue;
        const float* row = tile + (sy - s.lo) * W;
        float4 tv[3];
        if (MODE == 3) row_taps_d1(row, c4, W, tv[0], tv[1], tv[2]);
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) {
          const float4 v = MODE == 3 ? tv[tx] : row4<MODE == 1>(row, c4 * 4 + (tx - 1) * dil, W);
          const float k = wt[ty * 3 + tx];
          acc.x = fmaf(k, v.x, acc.x); acc.y = fmaf(k, v.y, acc.y); acc.z = fmaf(k, v.z, acc.z); acc.w = fmaf(k, v.w, acc.w);
          if (WG) accw[8 - (ty * 3 + tx)] += (xq.x * v.x + xq.y * v.y) + (xq.z * v.z + xq.w * v.w);
        }
      }
      float4* out = reinterpret_cast<float4*>(yp + (i64)yy * W) + c4;
      if (accumulate) { const float4 o = *out; acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w; }
      *out = acc;
      st_s += (acc.x + acc.y) + (acc.z + acc.w);
      st_q = fmaf(acc.x, acc.x, fmaf(acc.y, acc.y, fmaf(acc.z, acc.z, fmaf(acc.w, acc.w, st_q))));
      st_lo = fminf(fminf(st_lo, fminf(acc.x, acc.y)), fminf(acc.z, acc.w));
      st_hi = fmaxf(fmaxf(st_hi, fmaxf(acc.x, acc.y)), fmaxf(acc.z, acc.w));
    }
  } else {
    const int total = (s.y1 - s.y0) * W;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
      const int r = i / W, col = i - r * W;
      const int yy = s.y0 + r;
      float acc = 0.f;
#pragma unroll
      for (int ty = 0; ty < 3; ++ty) {
        const int sy = yy + (ty - 1) * dil;
        if (sy < 0 || sy >= H) continue;
        const float* row = tile + (sy - s.lo) * W;
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) {
          const int sx = col + (tx - 1) * dil;
          if (sx >= 0 && sx < W) {
            acc = fmaf(wt[ty * 3 + tx], row[sx], acc);
            if (WG) accw[8 - (ty * 3 + tx)] = fmaf(bnl ? bn_on_load(fxp[(i64)yy * W + col], bsc, bsh) : fxp[(i64)yy * W + col], row[sx], accw[8 - (ty * 3 + tx)]);
          }
        }
      }
      const i64 o = (i64)yy * W + col;
      acc = accumulate ? yp[o] + acc : acc;
      yp[o] = acc;
      st_s += acc;
      st_q = fmaf(acc, acc, st_q);
      st_lo = fminf(st_lo, acc);
      st_hi = fmaxf(st_hi, acc);
    }
  }
  // per-(channel, strip, image) partial sums for pfst_bn_finalize_partials: stats[c][n * gridDim.x + strip][2]  (the kernel is
  // HBM-bound, the arithmetic is free; saves the separate bn_stats read of the depthwise output)
  if (stats) {
    double bs = (double)st_s, bq = (double)st_q;
    block_sum2_d<true>(bs, bq, red);
    if (threadIdx.x == 0) {
      const i64 T = (i64)gridDim.x * gridDim.z;
      float2* dst = reinterpret_cast<float2*>(stats) + ((i64)c * T + (i64)n * gridDim.x + blockIdx.x);
      *dst = make_float2((float)bs, (float)bq);
    }
    if (stats_minmax) {                              // [C][T][2] behind the sums
      block_minmax(st_lo, st_hi, reinterpret_cast<float*>(red));
      if (threadIdx.x == 0) {
        const i64 T = (i64)gridDim.x * gridDim.z;
        reinterpret_cast<float2*>(stats)[(i64)gridDim.y * T + (i64)c * T + (i64)n * gridDim.x + blockIdx.x] = make_float2(st_lo, st_hi);
      }
    }
  }
  if (WG) block_add9(accw, dw_dst(dw, c, det_T, blockIdx.z * gridDim.x + blockIdx.x), reinterpret_cast<float*>(red));
}

// Whole-plane variant (the plane fits the LDS budget: the 128x128 ASPP planes): a workgroup walks `cpb` consecutive channels of one image
// and fetches the NEXT plane into registers (8 x 16 bytes per thread, issued right after the barrier) while it computes the current one
// from LDS -- the load latency of a plane is hidden behind the stencil of the previous one instead of being exposed once per plane.
// grid: (1, ceil(C / cpb), N), 512 threads, float4 outputs (MODE 1: dilation % 4 == 0, MODE 2: any dilation).
template <int MODE, bool WG = false>
__global__ __launch_bounds__(512) void dwconv3x3_plane_kernel(const float* __restrict__ x, i64 x_bs, const float* __restrict__ w,
                                                              float* __restrict__ y, i64 y_bs, int C, int H, int W, int dil, int cpb,
                                                              int flip, int accumulate, float* __restrict__ stats,
                                                              const float* __restrict__ fx = nullptr, i64 fx_bs = 0, float* __restrict__ dw = nullptr,
                                                              int det_T = 0, int stats_minmax = 0) {
  extern __shared__ float tile[];
  __shared__ double red[40];                       // (also the 8 x 9 floats of block_add9)
  const int n = blockIdx.z, c0 = blockIdx.y * cpb, c1 = min(C, c0 + cpb);
  const int HW = H * W, n4 = HW >> 2, W4 = W >> 2, tid = threadIdx.x;
  float4 r[8];                                     // host: n4 <= 8 * 512
  auto fetch = [&](int c) {
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (i64)n * x_bs + (i64)c * HW), 0, HW * 4, 0x00020000);
#pragma unroll
    for (int u = 0; u < 8; ++u) r[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, 16 * (tid + u * 512), 0, 0));
  };
  fetch(c0);
  for (int c = c0; c < c1; ++c) {
    float4* t4 = reinterpret_cast<float4*>(tile);
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (tid + u * 512 < n4) t4[tid + u * 512] = r[u];
    __syncthreads();
    if (c + 1 < c1) fetch(c + 1);
    float wt[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wt[t] = w[c * 9 + (flip ? 8 - t : t)];
    float* yp = y + (i64)n * y_bs + (i64)c * HW;
    float st_s = 0.f, st_q = 0.f;
    float st_lo = __builtin_inff(), st_hi = -__builtin_inff();
    float accw[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) accw[t] = 0.f;
    float4 xr[8];                                  // WG: the forward input's quads this thread's outputs pair with, all in flight at once
    if (WG) {
      const __amdgpu_buffer_rsrc_t xs =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(fx + (i64)n * fx_bs + (i64)c * HW), 0, HW * 4, 0x00020000);
#pragma unroll
      for (int u = 0; u < 8; ++u) xr[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xs, 16 * (tid + u * 512), 0, 0));
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = tid + u * 512;
      if (i >= n4) break;
      const int yy = i / W4, c4 = i - yy * W4;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int ty = 0; ty < 3; ++ty) {
        const int sy = yy + (ty - 1) * dil;
        if (sy < 0 || sy >= H) continue;
        const float* row = tile + sy * W;
        float4 tv[3];
        if (MODE == 3) row_taps_d1(row, c4, W, tv[0], tv[1], tv[2]);
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) {
          const float4 v = MODE == 3 ? tv[tx] : row4<MODE == 1>(row, c4 * 4 + (tx - 1) * dil, W);
          const float k = wt[ty * 3 + tx];
          acc.x = fmaf(k, v.x, acc.x); acc.y = fmaf(k, v.y, acc.y); acc.z = fmaf(k, v.z, acc.z); acc.w = fmaf(k, v.w, acc.w);
          if (WG) accw[8 - (ty * 3 + tx)] += (xr[u].x * v.x + xr[u].y * v.y) + (xr[u].z * v.z + xr[u].w * v.w);
        }
      }
      float4* out = reinterpret_cast<float4*>(yp) + i;
      if (accumulate) { const float4 o = *out; acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w; }
      *out = acc;
      st_s += (acc.x + acc.y) + (acc.z + acc.w);
      st_q = fmaf(acc.x, acc.x, fmaf(acc.y, acc.y, fmaf(acc.z, acc.z, fmaf(acc.w, acc.w, st_q))));
      st_lo = fminf(fminf(st_lo, fminf(acc.x, acc.y)), fminf(acc.z, acc.w));
      st_hi = fmaxf(fmaxf(st_hi, fmaxf(acc.x, acc.y)), fmaxf(acc.z, acc.w));
    }
    if (stats) {                                   // stats[c][n][2]: one strip per plane (pfst_dwconv_stats_slots == 1)
      double bs = (double)st_s, bq = (double)st_q;
      block_sum2_d<true>(bs, bq, red);
      if (tid == 0) reinterpret_cast<float2*>(stats)[(i64)c * gridDim.z + n] = make_float2((float)bs, (float)bq);
      if (stats_minmax) {                          // [C][N][2] behind the sums
        block_minmax(st_lo, st_hi, reinterpret_cast<float*>(red));
        if (tid == 0) reinterpret_cast<float2*>(stats)[(i64)C * gridDim.z + (i64)c * gridDim.z + n] = make_float2(st_lo, st_hi);
      }
    }
    if (WG) block_add9(accw, dw_dst(dw, c, det_T, blockIdx.z), reinterpret_cast<float*>(red));
    __syncthreads();                               // every tap of this plane has been read: the tile may be overwritten
  }
}

// dw[c][t] += sum_{n,p} dy[n][c][p] * x[n][c][p + off(t)]     VEC: W % 4 == 0 and 16-byte aligned planes
template <bool VEC, bool ALIGNED, bool D1 = false>
__global__ __launch_bounds__(512) void dwconv3x3_wgrad_kernel(const float* __restrict__ x, i64 x_bs, const float* __restrict__ dy,
                                                              i64 dy_bs, float* __restrict__ dw, int C, int H, int W, int dil, int R, int det_T = 0) {
  extern __shared__ float tile[];
  __shared__ float red[8][9];
  const int c = blockIdx.y, n = blockIdx.z;
  const Strip s = make_strip(blockIdx.x, R, H, dil);
  const float* xp = x + (i64)n * x_bs + (i64)c * H * W;
  const float* gp = dy + (i64)n * dy_bs + (i64)c * H * W;
  // the strip's output gradients: up to eight 16-byte loads per thread in flight beside the staging copy (a strip of at most 64 KB is
  // exactly one batch; larger strips take further batches inside the loop)
  const int W4v = W >> 2, totalv = (s.y1 - s.y0) * W4v;
  float4 gq[8];
  if (VEC) {
    const __amdgpu_buffer_rsrc_t grs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gp + (i64)s.y0 * W), 0, totalv * 16, 0x00020000);
#pragma unroll
    for (int u = 0; u < 8; ++u)
      gq[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(grs, 16 * (threadIdx.x + u * 512), 0, 0));
  }
  stage_rows(xp, tile, s.lo, s.hi, W);
  __syncthreads();
  float acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = 0.f;
  if (VEC) {
    const int W4 = W >> 2;
    const int total = (s.y1 - s.y0) * W4;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = threadIdx.x + u * 512;
      if (i >= total) break;
      const int r = i / W4, c4 = i - r * W4;
      const int yy = s.y0 + r;
      const float4 g = gq[u];
#pragma unroll
      for (int ty = 0; ty < 3; ++ty) {
        const int sy = yy + (ty - 1) * dil;
        if (sy < 0 || sy >= H) continue;
        const float* row = tile + (sy - s.lo) * W;
        float4 tv[3];
        if (D1) row_taps_d1(row, c4, W, tv[0], tv[1], tv[2]);
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) {
          const float4 v = D1 ? tv[tx] : row4<ALIGNED>(row, c4 * 4 + (tx - 1) * dil, W);
          acc[ty * 3 + tx] += (g.x * v.x + g.y * v.y) + (g.z * v.z + g.w * v.w);
        }
      }
    }
    for (int i = threadIdx.x + 8 * 512; i < total; i += blockDim.x) {
      const int r = i / W4, c4 = i - r * W4;
      const int yy = s.y0 + r;
      const float4 g = *(reinterpret_cast<const float4*>(gp + (i64)yy * W) + c4);
#pragma unroll
      for (int ty = 0; ty < 3; ++ty) {
        const int sy = yy + (ty - 1) * dil;
        if (sy < 0 || sy >= H) continue;
        const float* row = tile + (sy - s.lo) * W;
        float4 tv[3];
        if (D1) row_taps_d1(row, c4, W, tv[0], tv[1], tv[2]);
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) {
          const float4 v = D1 ? tv[tx] : row4<ALIGNED>(row, c4 * 4 + (tx - 1) * dil, W);
          acc[ty * 3 + tx] += (g.x * v.x + g.y * v.y) + (g.z * v.z + g.w * v.w);
        }
      }
    }
  } else {
    const int total = (s.y1 - s.y0) * W;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
      const int r = i / W, col = i - r * W;
      const int yy = s.y0 + r;
      const float g = gp[(i64)yy * W + col];
#pragma unroll
      for (int ty = 0; ty < 3; ++ty) {
        const int sy = yy + (ty - 1) * dil;
        if (sy < 0 || sy >= H) continue;
        const float* row = tile + (sy - s.lo) * W;
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) {
          const int sx = col + (tx - 1) * dil;
          if (sx >= 0 && sx < W) acc[ty * 3 + tx] = fmaf(g, row[sx], acc[ty * 3 + tx]);
        }
      }
    }
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const float v = wave_sum(acc[t]);
    if (lane == 0) red[wid][t] = v;
  }
  __syncthreads();
  if (threadIdx.x < 9) {
    float v = 0.f;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) v += red[k][threadIdx.x];
    atomicAdd(&dw_dst(dw, c, det_T, blockIdx.z * gridDim.x + blockIdx.x)[threadIdx.x], v);
  }
}

// ---- the three atrous depthwise branches of the ASPP head in one pass (sep_aspp_head.py:63-77: dilations 12 / 24 / 36 on the SAME 2048-channel
// map).  Separately each branch stages the input plane again (forward: 3 x (N + N) of traffic) and, in backward, reads and re-writes the
// shared input gradient (3 x (dy + x + old dx + dx)).  Here a plane is staged ONCE for all NS branches (forward: N + NS N), and the backward
// stages the NS output gradients of a channel one after the other while the input-gradient quads and the forward input's quads stay in
// registers: x once, every dy once, dx written once (x + NS dy + [old dx] + dx).  Whole planes of at most 8 x 512 quads, dilations % 4 == 0.
struct DwSets {
  const float* w[3];        // [C][9] filters
  float* y[3];              // forward: outputs; backward: unused
  const float* dy[3];       // backward: output gradients
  float* dw[3];             // backward: weight gradients (+=)
  float* stats[3];          // forward: BatchNorm partials [C][N][2] or NULL
  int stats_minmax;         // forward: the partials are followed by the (minimum, maximum) partials [C][N][2]
  long long bs[3];          // batch strides of y / dy
  int dil[3];
  const float* pre[3];      // backward with rec: the branches' own convolution outputs (pre-normalisation), batch stride bs[i]
  const pfst_bn_bwd_rec_t* rec[3];   // backward: NULL, or the records of pfst_bn_backward_sums: dy[i] is then the gradient of the BN + ReLU output
  float* pool;              // forward: NULL, or [N][C] plane means of x (nn.AdaptiveAvgPool2d(1) of the image-pool branch, aspp_head.py:69-77)
  const float* pool_grad;   // backward: NULL, or [N][C] gradients of those means: dx += pool_grad[n][c] * pool_scale
  float pool_scale;         // 1 / (H W)
};

template <int NS>
__global__ __launch_bounds__(512) void dwconv3x3_multi_fwd_kernel(const float* __restrict__ x, i64 x_bs, DwSets S, int C, int H, int W, int cpb) {
  extern __shared__ float tile[];
  __shared__ double red[40];
  const int n = blockIdx.z, c0 = blockIdx.y * cpb, c1 = min(C, c0 + cpb);
  const int HW = H * W, n4 = HW >> 2, W4 = W >> 2, tid = threadIdx.x;
  float4 r[8];
  auto fetch = [&](int c) {
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (i64)n * x_bs + (i64)c * HW), 0, HW * 4, 0x00020000);
#pragma unroll
    for (int u = 0; u < 8; ++u) r[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, 16 * (tid + u * 512), 0, 0));
  };
  fetch(c0);
  for (int c = c0; c < c1; ++c) {
    float4* t4 = reinterpret_cast<float4*>(tile);
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (tid + u * 512 < n4) t4[tid + u * 512] = r[u];
    if (S.pool) {                                  // the plane's mean, in fp64 like pfst_global_avgpool (one pass over x less for the image-pool branch)
      double ps = 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (tid + u * 512 < n4) ps += ((double)r[u].x + (double)r[u].y) + ((double)r[u].z + (double)r[u].w);
      ps = block_sum_d(ps, red);
      if (tid == 0) S.pool[n * C + c] = (float)(ps * (double)S.pool_scale);
    }
    __syncthreads();
    if (c + 1 < c1) fetch(c + 1);
#pragma unroll 1
    for (int si = 0; si < NS; ++si) {
      const int dil = S.dil[si];
      float wt[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) wt[t] = S.w[si][c * 9 + t];
      float* yp = S.y[si] + (i64)n * S.bs[si] + (i64)c * HW;
      float st_s = 0.f, st_q = 0.f;
      float st_lo = __builtin_inff(), st_hi = -__builtin_inff();
      for (int i = tid; i < n4; i += 512) {
        const int yy = i / W4, c4 = i - yy * W4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) {
          const int sy = yy + (ty - 1) * dil;
          if (sy < 0 || sy >= H) continue;
          const float* row = tile + sy * W;
#pragma unroll
          for (int tx = 0; tx < 3; ++tx) {
            const float4 v = row4<true>(row, c4 * 4 + (tx - 1) * dil, W);
            const float k = wt[ty * 3 + tx];
            acc.x = fmaf(k, v.x, acc.x); acc.y = fmaf(k, v.y, acc.y); acc.z = fmaf(k, v.z, acc.z); acc.w = fmaf(k, v.w, acc.w);
          }
        }
        reinterpret_cast<float4*>(yp)[i] = acc;
        st_s += (acc.x + acc.y) + (acc.z + acc.w);
        st_q = fmaf(acc.x, acc.x, fmaf(acc.y, acc.y, fmaf(acc.z, acc.z, fmaf(acc.w, acc.w, st_q))));
        st_lo = fminf(fminf(st_lo, fminf(acc.x, acc.y)), fminf(acc.z, acc.w));
        st_hi = fmaxf(fmaxf(st_hi, fmaxf(acc.x, acc.y)), fmaxf(acc.z, acc.w));
      }
      if (S.stats[si]) {                           // same partial layout as dwconv3x3_plane_kernel: stats[c][n][2]
        double bs = (double)st_s, bq = (double)st_q;
        block_sum2_d<true>(bs, bq, red);
        if (tid == 0) reinterpret_cast<float2*>(S.stats[si])[(i64)c * gridDim.z + n] = make_float2((float)bs, (float)bq);
        if (S.stats_minmax) {                      // [C][N][2] behind the sums
          block_minmax(st_lo, st_hi, reinterpret_cast<float*>(red));
          if (tid == 0) reinterpret_cast<float2*>(S.stats[si])[(i64)C * gridDim.z + (i64)c * gridDim.z + n] = make_float2(st_lo, st_hi);
        }
      }
    }
    __syncthreads();                               // every tap of this plane has been read: the tile may be overwritten
  }
}

// NT threads (512 or 1024), QPT = 4096 / NT quads each: 1024 threads keep half the planes' quads per thread -- under 128 registers, so the one
// workgroup a CU holds (two planes of LDS) is sixteen waves instead of eight (2.51 -> 2.33 ms per launch at the bench shape,
// tools/dw_multi_microbench.py).  [Gradient planes two iterations ahead in two register sets: 65+ spilled registers at either size.]
template <int NS, bool BNB, int NT>
__global__ __launch_bounds__(NT) void dwconv3x3_multi_bwd_kernel(const float* __restrict__ x, i64 x_bs, DwSets S, float* __restrict__ dx,
                                                                  i64 dx_bs, int accumulate, int C, int H, int W, int cpb, int det_T = 0) {
  // LDS: [gradient plane | forward-input plane].  BNB: the gradient plane staged is dL/dpre = BatchNorm-backward(dy, pre), formed from the two
  // planes fetched into registers one branch ahead (bn_bwd_elem4) -- the branches' dL/dpre tensors are never written.
  extern __shared__ float tile[];
  __shared__ double red[72];                        // block_add9: NT / 64 x 9 floats
  constexpr int QPT = 8 * 512 / NT;
  const int n = blockIdx.z, c0 = blockIdx.y * cpb, c1 = min(C, c0 + cpb);
  const int HW = H * W, n4 = HW >> 2, W4 = W >> 2, tid = threadIdx.x;
  float4* const t4 = reinterpret_cast<float4*>(tile);
  float4* const x4 = t4 + n4;
  float4 r[QPT], p[QPT];
  auto fetch = [&](int c, int si) {                // the gradient plane (BNB: and the pre-normalisation plane) of branch si, channel c -> registers
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(S.dy[si] + (i64)n * S.bs[si] + (i64)c * HW), 0, HW * 4, 0x00020000);
#pragma unroll
    for (int u = 0; u < QPT; ++u) r[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, 16 * (tid + u * NT), 0, 0));
    if (BNB) {
      const __amdgpu_buffer_rsrc_t ps =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(S.pre[si] + (i64)n * S.bs[si] + (i64)c * HW), 0, HW * 4, 0x00020000);
#pragma unroll
      for (int u = 0; u < QPT; ++u) p[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ps, 16 * (tid + u * NT), 0, 0));
    }
  };
  fetch(c0, 0);
  for (int c = c0; c < c1; ++c) {
    float4 dxa[QPT];
    {                                              // the forward input's plane: each thread keeps its own quads in LDS (read back per branch)
      const __amdgpu_buffer_rsrc_t xs =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (i64)n * x_bs + (i64)c * HW), 0, HW * 4, 0x00020000);
#pragma unroll
      for (int u = 0; u < QPT; ++u) {
        const float4 v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xs, 16 * (tid + u * NT), 0, 0));
        if (tid + u * NT < n4) x4[tid + u * NT] = v;
        dxa[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll 1
    for (int si = 0; si < NS; ++si) {              // (a run-time loop: unrolled, the 8 x 9 x NS tap bodies spill)
      if (BNB) {
        const pfst_bn_bwd_rec_t rec = S.rec[si][c];
#pragma unroll
        for (int u = 0; u < QPT; ++u)
          if (tid + u * NT < n4) t4[tid + u * NT] = bn_bwd_elem4(r[u], p[u], rec);
      } else {
#pragma unroll
        for (int u = 0; u < QPT; ++u)
          if (tid + u * NT < n4) t4[tid + u * NT] = r[u];
      }
      __syncthreads();
      if (si + 1 < NS) fetch(c, si + 1);
      else if (c + 1 < c1) fetch(c + 1, 0);
      const int dil = S.dil[si];
      float wt[9], accw[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) { wt[t] = S.w[si][c * 9 + 8 - t]; accw[t] = 0.f; }       // mirrored taps: the data gradient
#pragma unroll
      for (int u = 0; u < QPT; ++u) {
        const int i = tid + u * NT;
        if (i >= n4) break;
        const int yy = i / W4, c4 = i - yy * W4;
        const float4 xq = x4[i];
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) {
          const int sy = yy + (ty - 1) * dil;
          if (sy < 0 || sy >= H) continue;
          const float* row = tile + sy * W;
#pragma unroll
          for (int tx = 0; tx < 3; ++tx) {
            const float4 v = row4<true>(row, c4 * 4 + (tx - 1) * dil, W);
            const float k = wt[ty * 3 + tx];
            dxa[u].x = fmaf(k, v.x, dxa[u].x); dxa[u].y = fmaf(k, v.y, dxa[u].y); dxa[u].z = fmaf(k, v.z, dxa[u].z); dxa[u].w = fmaf(k, v.w, dxa[u].w);
            accw[8 - (ty * 3 + tx)] += (xq.x * v.x + xq.y * v.y) + (xq.z * v.z + xq.w * v.w);
          }
        }
      }
      block_add9(accw, dw_dst(S.dw[si], c, det_T, blockIdx.z), reinterpret_cast<float*>(red));
      __syncthreads();                             // every tap of this gradient plane has been read: the tile may be overwritten
    }
    float4* out = reinterpret_cast<float4*>(dx + (i64)n * dx_bs + (i64)c * HW);
    const float pg = S.pool_grad ? S.pool_grad[n * C + c] * S.pool_scale : 0.f;     // adjoint of the plane mean: the same value to every element
#pragma unroll
    for (int u = 0; u < QPT; ++u) {
      const int i = tid + u * NT;
      if (i >= n4) break;
      float4 a = dxa[u];
      if (S.pool_grad) { a.x += pg; a.y += pg; a.z += pg; a.w += pg; }
      if (accumulate) { const float4 o = out[i]; a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w; }
      out[i] = a;
    }
  }
}

// rows per strip so that (R + 2*dil) * W floats fit the LDS budget; whole plane when it fits
inline int strip_rows(int H, int W, int dil) {
  if ((i64)H * W * 4 <= DW_LDS_BYTES) return H;
  int rows = DW_LDS_BYTES / (W * 4) - 2 * dil;
  if (rows < 1) rows = 1;
  return rows < H ? rows : H;
}

inline size_t strip_lds(int R, int H, int W, int dil) {
  i64 rows = (i64)R + 2 * dil;
  if (rows > H) rows = H;
  return (size_t)rows * W * sizeof(float);
}

}  // namespace

extern "C" int pfst_dwconv_stats_slots(int H, int W, int dil) {
  if (H <= 0 || W <= 0 || dil < 1) return 0;
  return cdiv(H, strip_rows(H, W, dil));
}

extern "C" int pfst_dwconv3x3(const float* x, long long x_bs, const float* w, float* y, long long y_bs,
                              int N, int C, int H, int W, int dil, int flip, int accumulate, float* stats, int stats_minmax,
                              const float* bn_on_load_coef, pfst_stream_t stream) {
  PFST_CHECK_ARG(x && w && y && N > 0 && C > 0 && H > 0 && W > 0 && dil >= 1);
  PFST_CHECK_ARG(!stats_minmax || stats);
  PFST_CHECK_ARG(!bn_on_load_coef || !flip);         // the forward input is normalised on load; a data gradient has nothing to normalise
  const float4* bnl = reinterpret_cast<const float4*>(bn_on_load_coef);
  PFST_CHECK_ARG(x_bs >= (i64)C * H * W && y_bs >= (i64)C * H * W && C <= 65535 && N <= 65535);
  const int R = strip_rows(H, W, dil);
  const size_t lds = strip_lds(R, H, W, dil);
  PFST_CHECK_ARG(lds <= 150 * 1024);
  static bool set = false;
  if (!set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv3x3_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv3x3_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv3x3_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv3x3_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    set = true;
  }
  const bool vec = (W % 4 == 0) && (((uintptr_t)x | (uintptr_t)y) % 16 == 0) && (x_bs % 4 == 0) && (y_bs % 4 == 0) &&
                   (((i64)H * W) % 4 == 0);
  const int mode = !vec ? 0 : (dil % 4 == 0 ? 1 : (dil == 1 ? 3 : 2));
  dim3 grid(cdiv(H, R), C, N);
  hipStream_t st = (hipStream_t)stream;
  constexpr int cpb = 4;            // channels per workgroup of the plane kernel
  if (mode != 0 && R == H && (i64)H * W <= 8 * 512 * 4 && cpb > 0 && !bnl) {      // (the strip kernel carries the normalise-on-load variant)
    static bool set2 = false;
    if (!set2) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv3x3_plane_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
      hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv3x3_plane_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
      hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv3x3_plane_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
      set2 = true;
    }
    dim3 gp(1, cdiv(C, cpb), N);
    if (mode == 1)
      hipLaunchKernelGGL(dwconv3x3_plane_kernel<1>, gp, dim3(512), lds, st, x, x_bs, w, y, y_bs, C, H, W, dil, cpb, flip, accumulate, stats,
                         (const float*)nullptr, (i64)0, (float*)nullptr, 0, stats_minmax);
    else if (mode == 3)
      hipLaunchKernelGGL(dwconv3x3_plane_kernel<3>, gp, dim3(512), lds, st, x, x_bs, w, y, y_bs, C, H, W, dil, cpb, flip, accumulate, stats,
                         (const float*)nullptr, (i64)0, (float*)nullptr, 0, stats_minmax);
    else
      hipLaunchKernelGGL(dwconv3x3_plane_kernel<2>, gp, dim3(512), lds, st, x, x_bs, w, y, y_bs, C, H, W, dil, cpb, flip, accumulate, stats,
                         (const float*)nullptr, (i64)0, (float*)nullptr, 0, stats_minmax);
    PFST_CHECK_LAUNCH();
    return PFST_OK;
  }
  if (mode == 1)
    hipLaunchKernelGGL(dwconv3x3_kernel<1>, grid, dim3(512), lds, st, x, x_bs, w, y, y_bs, C, H, W, dil, R, flip, accumulate, stats, (const float*)nullptr, (i64)0, (float*)nullptr, bnl,
                       (const float*)nullptr, (i64)0, (const pfst_bn_bwd_rec_t*)nullptr, 0, stats_minmax);
  else if (mode == 2)
    hipLaunchKernelGGL(dwconv3x3_kernel<2>, grid, dim3(512), lds, st, x, x_bs, w, y, y_bs, C, H, W, dil, R, flip, accumulate, stats, (const float*)nullptr, (i64)0, (float*)nullptr, bnl,
                       (const float*)nullptr, (i64)0, (const pfst_bn_bwd_rec_t*)nullptr, 0, stats_minmax);
  else if (mode == 3)
    hipLaunchKernelGGL(dwconv3x3_kernel<3>, grid, dim3(512), lds, st, x, x_bs, w, y, y_bs, C, H, W, dil, R, flip, accumulate, stats, (const float*)nullptr, (i64)0, (float*)nullptr, bnl,
                       (const float*)nullptr, (i64)0, (const pfst_bn_bwd_rec_t*)nullptr, 0, stats_minmax);
  else
    hipLaunchKernelGGL(dwconv3x3_kernel<0>, grid, dim3(512), lds, st, x, x_bs, w, y, y_bs, C, H, W, dil, R, flip, accumulate, stats, (const float*)nullptr, (i64)0, (float*)nullptr, bnl,
                       (const float*)nullptr, (i64)0, (const pfst_bn_bwd_rec_t*)nullptr, 0, stats_minmax);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

// Backward of the depthwise convolution in ONE pass: dx (+)= the stencil of dy with mirrored taps, dw += the weight gradient, from one
// staging of dy and one read of the forward input x (see the WG note above dwconv3x3_kernel): 3 N of traffic instead of 2 N + 2 N.
extern "C" int pfst_dwconv3x3_bwd(const float* dy, long long dy_bs, const float* x, long long x_bs, const float* w, float* dx, long long dx_bs,
                                  float* dw, int N, int C, int H, int W, int dil, int accumulate, const float* bn_on_load_coef,
                                  const float* bn_pre, long long bn_pre_bs, const pfst_bn_bwd_rec_t* bn_rec, pfst_stream_t stream) {
  PFST_CHECK_ARG(dy && x && w && dx && dw && N > 0 && C > 0 && H > 0 && W > 0 && dil >= 1);
  PFST_CHECK_ARG((bn_rec == nullptr) == (bn_pre == nullptr) && (!bn_rec || bn_pre_bs >= (i64)C * H * W));
  const float4* bnl = reinterpret_cast<const float4*>(bn_on_load_coef);
  PFST_CHECK_ARG(dy_bs >= (i64)C * H * W && x_bs >= (i64)C * H * W && dx_bs >= (i64)C * H * W && C <= 65535 && N <= 65535);
  const int R = strip_rows(H, W, dil);
  const size_t lds = strip_lds(R, H, W, dil);
  PFST_CHECK_ARG(lds <= 150 * 1024);
  static bool set = false;
  if (!set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv3x3_kernel<0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv3x3_kernel<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv3x3_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv3x3_kernel<3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv3x3_plane_kernel<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv3x3_plane_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv3x3_plane_kernel<3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    set = true;
  }
  const bool vec = (W % 4 == 0) && (((uintptr_t)dy | (uintptr_t)dx | (uintptr_t)x) % 16 == 0) && (dy_bs % 4 == 0) && (dx_bs % 4 == 0) &&
                   (x_bs % 4 == 0) && (((i64)H * W) % 4 == 0) && (!bn_pre || (((uintptr_t)bn_pre % 16 == 0) && bn_pre_bs % 4 == 0));
#ifdef PFST_DIAG_DW_MODE2
  const int mode = !vec ? 0 : (dil % 4 == 0 ? 1 : 2);
#else
  const int mode = !vec ? 0 : (dil % 4 == 0 ? 1 : (dil == 1 ? 3 : 2));
#endif
  dim3 grid(cdiv(H, R), C, N);
  hipStream_t st = (hipStream_t)stream;
  constexpr int cpb = 4;
  float* const none = nullptr;
  const bool plane = mode != 0 && R == H && (i64)H * W <= 8 * 512 * 4 && cpb > 0 && !bnl && !bn_rec;
  // deterministic mode: per-workgroup slots of a zeroed scratch + an ordered reduction instead of the workgroups' atomic adds into dw
  const int det_T = pfst_deterministic() ? (plane ? N : (int)grid.x * N) : 0;
  float* dwk = dw;
  if (det_T) {
    const size_t bytes = (size_t)C * det_T * 9 * sizeof(float);
    dwk = static_cast<float*>(pfst_det_scratch(bytes, st));
    PFST_CHECK_ARG(dwk != nullptr);
    if (hipMemsetAsync(dwk, 0, bytes, st) != hipSuccess) return PFST_ERR_LAUNCH;
  }
  if (plane) {
    dim3 gp(1, cdiv(C, cpb), N);
    if (mode == 1)
      hipLaunchKernelGGL((dwconv3x3_plane_kernel<1, true>), gp, dim3(512), lds, st, dy, dy_bs, w, dx, dx_bs, C, H, W, dil, cpb, 1, accumulate, none, x, (i64)x_bs, dwk, det_T);
    else if (mode == 3)
      hipLaunchKernelGGL((dwconv3x3_plane_kernel<3, true>), gp, dim3(512), lds, st, dy, dy_bs, w, dx, dx_bs, C, H, W, dil, cpb, 1, accumulate, none, x, (i64)x_bs, dwk, det_T);
    else
      hipLaunchKernelGGL((dwconv3x3_plane_kernel<2, true>), gp, dim3(512), lds, st, dy, dy_bs, w, dx, dx_bs, C, H, W, dil, cpb, 1, accumulate, none, x, (i64)x_bs, dwk, det_T);
  } else if (mode == 1)
    hipLaunchKernelGGL((dwconv3x3_kernel<1, true>), grid, dim3(512), lds, st, dy, dy_bs, w, dx, dx_bs, C, H, W, dil, R, 1, accumulate, none, x, (i64)x_bs, dwk, bnl, bn_pre, (i64)bn_pre_bs, bn_rec, det_T);
  else if (mode == 2)
    hipLaunchKernelGGL((dwconv3x3_kernel<2, true>), grid, dim3(512), lds, st, dy, dy_bs, w, dx, dx_bs, C, H, W, dil, R, 1, accumulate, none, x, (i64)x_bs, dwk, bnl, bn_pre, (i64)bn_pre_bs, bn_rec, det_T);
  else if (mode == 3)
    hipLaunchKernelGGL((dwconv3x3_kernel<3, true>), grid, dim3(512), lds, st, dy, dy_bs, w, dx, dx_bs, C, H, W, dil, R, 1, accumulate, none, x, (i64)x_bs, dwk, bnl, bn_pre, (i64)bn_pre_bs, bn_rec, det_T);
  else
    hipLaunchKernelGGL((dwconv3x3_kernel<0, true>), grid, dim3(512), lds, st, dy, dy_bs, w, dx, dx_bs, C, H, W, dil, R, 1, accumulate, none, x, (i64)x_bs, dwk, bnl, bn_pre, (i64)bn_pre_bs, bn_rec, det_T);
  if (det_T) hipLaunchKernelGGL(dw_det_reduce_kernel, dim3(C), dim3(64), 0, st, dwk, dw, C, det_T);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

// The NS <= 3 depthwise branches that share one input (the ASPP head's atrous branches): forward in one pass over x, backward in one pass over
// x with every branch's output gradient read once and the input gradient written once.  Whole planes (H * W <= 16384, W % 4 == 0),
// dilations % 4 == 0, 16-byte aligned dense planes: pfst_dwconv3x3_multi_ok says whether a shape qualifies.
extern "C" int pfst_dwconv3x3_multi_ok(int H, int W, int ns, const int* dils) {
  if (ns < 1 || ns > 3 || H <= 0 || W <= 0 || W % 4 != 0 || (i64)H * W > 8 * 512 * 4 || (i64)H * W * 4 > DW_LDS_BYTES) return 0;
  for (int i = 0; i < ns; ++i)
    if (dils[i] < 4 || dils[i] % 4 != 0) return 0;
  return 1;
}

static int dw_multi_sets(DwSets& S, int ns, const float* const* w, float* const* y, const float* const* dy, float* const* dw, float* const* stats,
                         const long long* bs, const int* dils, i64 plane_elems) {
  S.pool = nullptr;
  S.pool_grad = nullptr;
  S.pool_scale = 0.f;
  S.stats_minmax = 0;
  for (int i = 0; i < 3; ++i) {
    const int k = i < ns ? i : 0;
    S.pre[i] = nullptr;
    S.rec[i] = nullptr;
    S.w[i] = w[k];
    S.y[i] = y ? y[k] : nullptr;
    S.dy[i] = dy ? dy[k] : nullptr;
    S.dw[i] = dw ? dw[k] : nullptr;
    S.stats[i] = stats ? stats[k] : nullptr;
    S.bs[i] = bs[k];
    S.dil[i] = dils[k];
    if (!S.w[i] || S.bs[i] < plane_elems || (S.bs[i] & 3)) return 0;
    if (((uintptr_t)S.y[i] | (uintptr_t)S.dy[i]) & 15) return 0;
  }
  return 1;
}

extern "C" int pfst_dwconv3x3_multi_fwd(const float* x, long long x_bs, int ns, const float* const* w, float* const* y, const long long* y_bs,
                                        float* const* stats, int stats_minmax, const int* dils, float* plane_mean, int N, int C, int H, int W,
                                        pfst_stream_t stream) {
  PFST_CHECK_ARG(x && w && y && y_bs && dils && N > 0 && C > 0 && C <= 65535 && N <= 65535 && pfst_dwconv3x3_multi_ok(H, W, ns, dils));
  PFST_CHECK_ARG(!stats_minmax || stats);
  PFST_CHECK_ARG(x_bs >= (i64)C * H * W && (x_bs & 3) == 0 && ((uintptr_t)x & 15) == 0);
  DwSets S;
  PFST_CHECK_ARG(dw_multi_sets(S, ns, w, y, nullptr, nullptr, stats, y_bs, dils, (i64)C * H * W));
  for (int i = 0; i < ns; ++i) PFST_CHECK_ARG(y[i] != nullptr);
  S.pool = plane_mean;
  S.pool_scale = 1.0f / (float)(H * W);
  S.stats_minmax = stats_minmax;
  static bool set = false;
  if (!set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv3x3_multi_fwd_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv3x3_multi_fwd_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv3x3_multi_fwd_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    set = true;
  }
  const int cpb = 4;
  const size_t lds = (size_t)H * W * sizeof(float);
  dim3 gp(1, cdiv(C, cpb), N);
  hipStream_t st = (hipStream_t)stream;
  if (ns == 1) hipLaunchKernelGGL(dwconv3x3_multi_fwd_kernel<1>, gp, dim3(512), lds, st, x, (i64)x_bs, S, C, H, W, cpb);
  else if (ns == 2) hipLaunchKernelGGL(dwconv3x3_multi_fwd_kernel<2>, gp, dim3(512), lds, st, x, (i64)x_bs, S, C, H, W, cpb);
  else hipLaunchKernelGGL(dwconv3x3_multi_fwd_kernel<3>, gp, dim3(512), lds, st, x, (i64)x_bs, S, C, H, W, cpb);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

extern "C" int pfst_dwconv3x3_multi_bwd(const float* x, long long x_bs, int ns, const float* const* w, const float* const* dy,
                                        const long long* dy_bs, float* const* dw, const int* dils, const float* plane_mean_grad, float* dx,
                                        long long dx_bs, int accumulate, const float* const* bn_pre, const pfst_bn_bwd_rec_t* const* bn_rec,
                                        int N, int C, int H, int W, pfst_stream_t stream) {
  PFST_CHECK_ARG(x && w && dy && dy_bs && dw && dils && dx && N > 0 && C > 0 && C <= 65535 && N <= 65535 && pfst_dwconv3x3_multi_ok(H, W, ns, dils));
  PFST_CHECK_ARG(x_bs >= (i64)C * H * W && dx_bs >= (i64)C * H * W && ((x_bs | dx_bs) & 3) == 0 && (((uintptr_t)x | (uintptr_t)dx) & 15) == 0);
  DwSets S;
  PFST_CHECK_ARG(dw_multi_sets(S, ns, w, nullptr, dy, dw, nullptr, dy_bs, dils, (i64)C * H * W));
  for (int i = 0; i < ns; ++i) PFST_CHECK_ARG(dy[i] != nullptr && dw[i] != nullptr);
  S.pool_grad = plane_mean_grad;
  S.pool_scale = 1.0f / (float)(H * W);
  const bool bnb = bn_rec != nullptr;
  PFST_CHECK_ARG(bnb == (bn_pre != nullptr));
  if (bnb)
    for (int i = 0; i < 3; ++i) {
      const int k = i < ns ? i : 0;
      PFST_CHECK_ARG(bn_pre[k] && bn_rec[k] && ((uintptr_t)bn_pre[k] & 15) == 0);
      S.pre[i] = bn_pre[k];
      S.rec[i] = bn_rec[k];
    }
  static bool set = false;
  if (!set) {
#define PFST_DW_MULTI_ATTR(NS_, B_) \
    hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv3x3_multi_bwd_kernel<NS_, B_, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)
    PFST_DW_MULTI_ATTR(1, false); PFST_DW_MULTI_ATTR(2, false); PFST_DW_MULTI_ATTR(3, false);
    PFST_DW_MULTI_ATTR(1, true); PFST_DW_MULTI_ATTR(2, true); PFST_DW_MULTI_ATTR(3, true);
#undef PFST_DW_MULTI_ATTR
    set = true;
  }
  const int cpb = 4;
  const size_t lds = 2 * (size_t)H * W * sizeof(float);          // gradient plane + forward-input plane
  dim3 gp(1, cdiv(C, cpb), N);
  hipStream_t st = (hipStream_t)stream;
  // deterministic mode (see pfst_dwconv3x3_bwd): every branch's weight gradient through [C][N][9] slots of the scratch
  const int det_T = pfst_deterministic() ? N : 0;
  float* real_dw[3] = {S.dw[0], S.dw[1], S.dw[2]};
  if (det_T) {
    const size_t per = (size_t)C * det_T * 9, bytes = per * ns * sizeof(float);
    float* part = static_cast<float*>(pfst_det_scratch(bytes, st));
    PFST_CHECK_ARG(part != nullptr);
    if (hipMemsetAsync(part, 0, bytes, st) != hipSuccess) return PFST_ERR_LAUNCH;
    for (int i = 0; i < 3; ++i) S.dw[i] = part + (size_t)(i < ns ? i : 0) * per;
  }
  // 1024 threads: 2.33 ms per launch against 2.50 ms with 512 (profiles/r04_dw_multi_microbench.txt)
#define PFST_DW_MULTI_LAUNCH(NS_, B_) \
  hipLaunchKernelGGL((dwconv3x3_multi_bwd_kernel<NS_, B_, 1024>), gp, dim3(1024), lds, st, x, (i64)x_bs, S, dx, (i64)dx_bs, accumulate, C, H, W, cpb, det_T)
  if (bnb) {
    if (ns == 1) { PFST_DW_MULTI_LAUNCH(1, true); } else if (ns == 2) { PFST_DW_MULTI_LAUNCH(2, true); } else { PFST_DW_MULTI_LAUNCH(3, true); }
  } else {
    if (ns == 1) { PFST_DW_MULTI_LAUNCH(1, false); } else if (ns == 2) { PFST_DW_MULTI_LAUNCH(2, false); } else { PFST_DW_MULTI_LAUNCH(3, false); }
  }
#undef PFST_DW_MULTI_LAUNCH
  if (det_T)
    for (int i = 0; i < ns; ++i) hipLaunchKernelGGL(dw_det_reduce_kernel, dim3(C), dim3(64), 0, st, S.dw[i], real_dw[i], C, det_T);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

extern "C" int pfst_dwconv3x3_wgrad(const float* x, long long x_bs, const float* dy, long long dy_bs, float* dw,
                                    int N, int C, int H, int W, int dil, pfst_stream_t stream) {
  PFST_CHECK_ARG(x && dy && dw && N > 0 && C > 0 && H > 0 && W > 0 && dil >= 1 && N <= 65535 && C <= 65535);
  const int R = strip_rows(H, W, dil);
  const size_t lds = strip_lds(R, H, W, dil);
  PFST_CHECK_ARG(lds <= 150 * 1024);
  static bool set = false;
  if (!set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv3x3_wgrad_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv3x3_wgrad_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv3x3_wgrad_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv3x3_wgrad_kernel<true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    set = true;
  }
  const bool vec = (W % 4 == 0) && (((uintptr_t)x | (uintptr_t)dy) % 16 == 0) && (x_bs % 4 == 0) && (dy_bs % 4 == 0) &&
                   (((i64)H * W) % 4 == 0);
  dim3 grid(cdiv(H, R), C, N);
  hipStream_t st = (hipStream_t)stream;
  const int det_T = pfst_deterministic() ? (int)grid.x * N : 0;        // deterministic mode: see pfst_dwconv3x3_bwd
  float* dwk = dw;
  if (det_T) {
    const size_t bytes = (size_t)C * det_T * 9 * sizeof(float);
    dwk = static_cast<float*>(pfst_det_scratch(bytes, st));
    PFST_CHECK_ARG(dwk != nullptr);
    if (hipMemsetAsync(dwk, 0, bytes, st) != hipSuccess) return PFST_ERR_LAUNCH;
  }
  if (vec && dil % 4 == 0)
    hipLaunchKernelGGL((dwconv3x3_wgrad_kernel<true, true>), grid, dim3(512), lds, st, x, x_bs, dy, dy_bs, dwk, C, H, W, dil, R, det_T);
  else if (vec && dil == 1)
    hipLaunchKernelGGL((dwconv3x3_wgrad_kernel<true, false, true>), grid, dim3(512), lds, st, x, x_bs, dy, dy_bs, dwk, C, H, W, dil, R, det_T);
  else if (vec)
    hipLaunchKernelGGL((dwconv3x3_wgrad_kernel<true, false>), grid, dim3(512), lds, st, x, x_bs, dy, dy_bs, dwk, C, H, W, dil, R, det_T);
  else
    hipLaunchKernelGGL((dwconv3x3_wgrad_kernel<false, false>), grid, dim3(512), lds, st, x, x_bs, dy, dy_bs, dwk, C, H, W, dil, R, det_T);
  if (det_T) hipLaunchKernelGGL(dw_det_reduce_kernel, dim3(C), dim3(64), 0, st, dwk, dw, C, det_T);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}
