// BatchNorm2d in TRAINING mode (+ ReLU, + residual add) and its backward: HBM-bound passes.
// Reference: nn.BatchNorm2d(eps=1e-5, momentum=0.1) inside mmcv ConvModule and
// rsiseg/models/backbones/resnet.py:273-305 (Bottleneck: bn -> relu, bn3 + identity -> relu).
// The teacher also runs BN in train mode (pfgst.py:247-251 only switches dropout off).
//
// Statistics are accumulated in fp64 (per-thread fp32 loads, fp64 sums), one atomic pair per block.
#include <stdlib.h>
#include "common.h"
#include "amax.h"
#include "../../include/pfst_hip.h"

// cache policy of bn_apply's streams (raw buffer aux bits on gfx950: 2 = nt): build-time knobs for A/B builds
#ifndef PFST_BN_LOAD_AUX
#define PFST_BN_LOAD_AUX 2      // the pre-BN tensor is streamed (nt): bn_apply -1.5 ms per step, profiles/r05_ab_cache_policy.txt (nt stores: +0.5 ms)
#endif
#ifndef PFST_BN_RES_AUX
#define PFST_BN_RES_AUX 0
#endif
#ifndef PFST_BN_STORE_AUX
#define PFST_BN_STORE_AUX 0
#endif

namespace {

constexpr int BN_SPLIT_TARGET = 2048;  // aim for this many blocks in the reduction passes

// grid: (splits, C, N); each block reduces a contiguous chunk of one (image, channel) plane
// det_part != NULL (deterministic mode, api.cpp): this block's two sums go to slot (n, split) of det_part[c][gridDim.z * gridDim.x][2] instead of
// being added atomically into ws; bn_det_sum_kernel adds the slots in index order
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, i64 x_bs, int HW, int chunk,
                                                       double* __restrict__ ws, double* __restrict__ det_part = nullptr) {
  __shared__ double sm[32];
  const int c = blockIdx.y, n = blockIdx.z;
  const float* xp = x + (i64)n * x_bs + (i64)c * HW;
  const int beg = blockIdx.x * chunk;
  const int end = min(beg + chunk, HW);
  double s = 0.0, ss = 0.0;
  if ((chunk & 3) == 0 && (HW & 3) == 0 && ((uintptr_t)xp & 15) == 0) {
    for (int i = (beg >> 2) + threadIdx.x; i < (end >> 2); i += blockDim.x) {
      const float4 v = reinterpret_cast<const float4*>(xp)[i];
      s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
      ss += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
    }
  } else {
    for (int i = beg + threadIdx.x; i < end; i += blockDim.x) {
      const float v = xp[i];
      s += (double)v;
      ss += (double)v * (double)v;
    }
  }
  block_sum2_d(s, ss, sm);
  if (threadIdx.x == 0) {
    if (det_part) {
      double* d = det_part + (((i64)c * gridDim.z + n) * gridDim.x + blockIdx.x) * 2;
      d[0] = s;
      d[1] = ss;
    } else {
      atomicAdd(&ws[2 * c], s);
      atomicAdd(&ws[2 * c + 1], ss);
    }
  }
}

// deterministic mode: ws[2c], ws[2c+1] = the T slots of channel c summed in index order     grid: ceil(C / 256)
__global__ __launch_bounds__(256) void bn_det_sum_kernel(const double* __restrict__ part, int T, int C, double* __restrict__ ws) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double a = 0.0, b = 0.0;
  for (int k = 0; k < T; ++k) {
    a += part[((i64)c * T + k) * 2];
    b += part[((i64)c * T + k) * 2 + 1];
  }
  ws[2 * c] = a;
  ws[2 * c + 1] = b;
}

// coef[c] = (mean, invstd, sc, sh): what the fused BatchNorm-backward sums of a data-gradient launch read (pfst_bnb_fuse_t)
__device__ __forceinline__ void write_coef(float4* __restrict__ coef, const float* __restrict__ gamma, const float* __restrict__ beta,
                                           int c, float mean, float invstd) {
  if (!coef) return;
  float sc, sh;
  bn_affine(mean, invstd, gamma[c], beta[c], sc, sh);
  coef[c] = make_float4(mean, invstd, sc, sh);
}

__global__ void bn_finalize_kernel(const double* __restrict__ ws, int C, double count, float* __restrict__ mean,
                                   float* __restrict__ invstd, float* __restrict__ rmean, float* __restrict__ rvar,
                                   float momentum, float eps, const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float4* __restrict__ coef) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double m = ws[2 * c] / count;
  double var = ws[2 * c + 1] / count - m * m;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)m;
  invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  write_coef(coef, gamma, beta, c, mean[c], invstd[c]);
  if (rmean) {
    const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)m;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
  }
}

// this thread's share of the T (sum, sum') slots of one channel, in fp64.  The slots of the high-resolution layers are many (T = 8192 at
// 1/4, 32768 at 1/2 resolution: 64-256 KB per channel) and one workgroup per channel walks them: eight slots per thread in flight (four
// independent 16-byte loads) instead of one -- the 1-slot loop spent 20-100 us per launch on the stem / layer1 channels, latency-bound.
__device__ __forceinline__ void sum_partial_slots(const float2* __restrict__ p, int T, double& s, double& ss) {
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
  int done = 0;
  if (((uintptr_t)p & 15) == 0) {
    const float4* p4 = reinterpret_cast<const float4*>(p);
    const int T2 = T >> 1, bd = blockDim.x;
    for (int i = threadIdx.x; i < T2; i += 4 * bd) {
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 v0 = p4[i];
      const float4 v1 = i + bd < T2 ? p4[i + bd] : z;
      const float4 v2 = i + 2 * bd < T2 ? p4[i + 2 * bd] : z;
      const float4 v3 = i + 3 * bd < T2 ? p4[i + 3 * bd] : z;
      a0 += (double)v0.x + (double)v0.z; b0 += (double)v0.y + (double)v0.w;
      a1 += (double)v1.x + (double)v1.z; b1 += (double)v1.y + (double)v1.w;
      a2 += (double)v2.x + (double)v2.z; b2 += (double)v2.y + (double)v2.w;
      a3 += (double)v3.x + (double)v3.z; b3 += (double)v3.y + (double)v3.w;
    }
    done = T2 * 2;
  }
  for (int i = done + threadIdx.x; i < T; i += blockDim.x) {
    const float2 v = p[i];
    a0 += (double)v.x;
    b0 += (double)v.y;
  }
  s = (a0 + a1) + (a2 + a3);
  ss = (b0 + b1) + (b2 + b3);
}

// statistics from the conv epilogue's partials: part[c][T][2] (sum, sum of squares per slot)   grid: C blocks
// BD threads: 256, or 1024 for the long slot rows of the full- / half- / quarter-resolution layers (T >= 4096: few channels, so few
// workgroups, each walking 64-256 KB per array -- latency-bound on the loads one workgroup keeps in flight)
template <int BD>
__global__ __launch_bounds__(BD) void bn_finalize_partials_kernel(const float* __restrict__ part, int T, double count,
                                                                    float* __restrict__ mean, float* __restrict__ invstd,
                                                                    float* __restrict__ rmean, float* __restrict__ rvar,
                                                                    float momentum, float eps, const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, float4* __restrict__ coef,
                                                                    const float* __restrict__ minmax, int relu, float* __restrict__ y_amax) {
  __shared__ double sm[32];
  __shared__ float mm[2][BD / 64];
  const int c = blockIdx.x;
  // minmax != NULL: [C][T][2] (minimum, maximum) partials of the same producer -> the channel's extrema of the pre-activation.  Its loads are
  // issued BEFORE the sums' barrier, so that both slot arrays are in flight together (the kernel is one round trip of latency, not two)
  float lo = __builtin_inff(), hi = -__builtin_inff();
  if (minmax) {
    const float2* q = reinterpret_cast<const float2*>(minmax) + (i64)c * T;
    int done = 0;
    if (((uintptr_t)q & 15) == 0) {
      const float4* q4 = reinterpret_cast<const float4*>(q);
      const int T2 = T >> 1;
      for (int i = threadIdx.x; i < T2; i += blockDim.x) {
        const float4 v = q4[i];
        lo = fminf(lo, fminf(v.x, v.z));
        hi = fmaxf(hi, fmaxf(v.y, v.w));
      }
      done = T2 * 2;
    }
    for (int i = done + threadIdx.x; i < T; i += blockDim.x) {
      const float2 v = q[i];
      lo = fminf(lo, v.x);
      hi = fmaxf(hi, v.y);
    }
  }
  double s, ss;
  sum_partial_slots(reinterpret_cast<const float2*>(part) + (i64)c * T, T, s, ss);
  if (minmax) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      lo = fminf(lo, __shfl_xor(lo, o));
      hi = fmaxf(hi, __shfl_xor(hi, o));
    }
    if ((threadIdx.x & 63) == 0) {
      mm[0][threadIdx.x >> 6] = lo;
      mm[1][threadIdx.x >> 6] = hi;
    }
  }
  // the per-channel scalars of the tail, requested before the barrier as well
  const float g_c = gamma ? gamma[c] : 1.f, b_c = beta ? beta[c] : 0.f;
  const float rm_c = rmean ? rmean[c] : 0.f, rv_c = rmean ? rvar[c] : 0.f;
  block_sum2_d(s, ss, sm);                 // (its barriers also publish mm)
  if (threadIdx.x == 0) {
    const double m = s / count;
    double var = ss / count - m * m;
    if (var < 0.0) var = 0.0;
    const float mf = (float)m, isf = (float)(1.0 / sqrt(var + (double)eps));
    mean[c] = mf;
    invstd[c] = isf;
    float sc, sh;
    bn_affine(mf, isf, g_c, b_c, sc, sh);
    if (coef) coef[c] = make_float4(mf, isf, sc, sh);
    if (minmax && y_amax) {
      // y = [relu](fma(x, sc, sh)) is monotone in x for fixed (sc, sh) -- fma rounds monotonically --, so the channel's extreme outputs are the
      // images of its extreme inputs: max |y| of the tensor the normalisation pass WOULD write, exactly, before (or without) writing it
      for (int w = 1; w < (int)(blockDim.x >> 6); ++w) {
        lo = fminf(lo, mm[0][w]);
        hi = fmaxf(hi, mm[1][w]);
      }
      const float ya = __fmaf_rn(lo, sc, sh), yb = __fmaf_rn(hi, sc, sh);
      const float am = relu ? fmaxf(fmaxf(ya, yb), 0.f) : fmaxf(fabsf(ya), fabsf(yb));
      if (am > 0.f) atomicMax(reinterpret_cast<unsigned*>(y_amax) + (c & (PFST_AMAX_SUB - 1)), __builtin_bit_cast(unsigned, am));
    }
    if (rmean) {
      const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
      rmean[c] = (1.f - momentum) * rm_c + momentum * (float)m;
      rvar[c] = (1.f - momentum) * rv_c + momentum * (float)unb;
    }
  }
}

// grid: (blocks over HW, C, N)
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, i64 x_bs, const float* __restrict__ res,
                                                       i64 res_bs, float* __restrict__ y, i64 y_bs,
                                                       const float* __restrict__ mean, const float* __restrict__ invstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       int C, int HW, int relu, unsigned long long* __restrict__ mask, int rev,
                                                       float* __restrict__ amax, const float* __restrict__ post,
                                                       const float4* __restrict__ res_coef) {
  // res_coef != NULL: `res` is the PRE-normalisation output of the downsample conv -> BN layer of a residual block (resnet.py:298-303: no
  // ReLU there) and res_coef[C] = (mean, invstd, sc, sh) its record: the residual is normalised as it is loaded -- fma(r, sc, sh), the value the
  // layer's own normalisation pass would have written -- and that tensor is never written
  // post != NULL: y is multiplied by post[n][c] after the ReLU -- nn.Dropout2d's keep / (1 - p) factor of the layer that feeds conv_seg
  // (decode_head.py:103-107,242-247) folded into this pass: the same product the separate scaling pass formed, one tensor round trip less
  // rev: walk the tensor from its end (planes, images and blocks in descending order) -- see pfst_bn_order()
  // amax != NULL: max |y| of what this launch writes goes to that slot group (amax.h): the scale of the next f16x3 GEMM's operand
  float am = 0.f;
  const int c = rev ? gridDim.y - 1 - blockIdx.y : blockIdx.y, n = rev ? gridDim.z - 1 - blockIdx.z : blockIdx.z;
  const int bxi = rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
  float sc, sh;
  bn_affine(mean[c], invstd[c], gamma[c], beta[c], sc, sh);
  const float pm = post ? post[n * C + c] : 1.f;
  const float rsc = res_coef ? res_coef[c].z : 1.f, rsh = res_coef ? res_coef[c].w : 0.f;
  const float* xp = x + (i64)n * x_bs + (i64)c * HW;
  const float* rp = res ? res + (i64)n * res_bs + (i64)c * HW : nullptr;
  float* yp = y + (i64)n * y_bs + (i64)c * HW;
  const int stride = gridDim.x * blockDim.x;
  if ((HW & 3) == 0 && (((uintptr_t)xp | (uintptr_t)yp | (uintptr_t)rp) & 15) == 0) {
    const int n4 = HW >> 2;
    // four 16-byte loads per operand in flight per thread (buffer loads: offsets past the plane return zeros and their stores are dropped)
    constexpr int U = 4;
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xp), 0, HW * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(rp ? rp : xp), 0, HW * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(yp, 0, HW * 4, 0x00020000);
    for (int i0 = bxi * blockDim.x + threadIdx.x; i0 < n4; i0 += U * stride) {
      float4 v[U], r[U];
      unsigned off[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * stride;
        off[u] = i < n4 ? 16u * (unsigned)i : OOB;
        v[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, off[u], 0, PFST_BN_LOAD_AUX));
      }
      if (rp) {
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rr, off[u], 0, PFST_BN_RES_AUX));
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * stride;
        float4 w = v[u];
        w.x = __fmaf_rn(w.x, sc, sh); w.y = __fmaf_rn(w.y, sc, sh); w.z = __fmaf_rn(w.z, sc, sh); w.w = __fmaf_rn(w.w, sc, sh);
        if (rp) {
          if (res_coef) { r[u].x = __fmaf_rn(r[u].x, rsc, rsh); r[u].y = __fmaf_rn(r[u].y, rsc, rsh); r[u].z = __fmaf_rn(r[u].z, rsc, rsh); r[u].w = __fmaf_rn(r[u].w, rsc, rsh); }
          w.x += r[u].x; w.y += r[u].y; w.z += r[u].z; w.w += r[u].w;
        }
        if (mask && i < n4) {
          // ReLU bitmask for the backward pass (the host guarantees HW % 256 == 0, so all 64 lanes are here and hold 256
          // consecutive elements): word [i >> 6][k] bit (lane) <-> component k of lane's float4, i.e. element 4*lane + k
          const unsigned long long b0 = __ballot(w.x > 0.f), b1 = __ballot(w.y > 0.f), b2 = __ballot(w.z > 0.f), b3 = __ballot(w.w > 0.f);
          const int l = threadIdx.x & 63;
          if (l < 4) mask[((i64)n * C + c) * (HW >> 6) + (i64)(i >> 6) * 4 + l] = l == 0 ? b0 : (l == 1 ? b1 : (l == 2 ? b2 : b3));
        }
        if (relu) { w.x = fmaxf(w.x, 0.f); w.y = fmaxf(w.y, 0.f); w.z = fmaxf(w.z, 0.f); w.w = fmaxf(w.w, 0.f); }
        if (post) { w.x *= pm; w.y *= pm; w.z *= pm; w.w *= pm; }
        if (i < n4) am = fmaxf(fmaxf(am, fmaxf(fabsf(w.x), fabsf(w.y))), fmaxf(fabsf(w.z), fabsf(w.w)));
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, w), yr, off[u], 0, PFST_BN_STORE_AUX);
      }
    }
  } else {
    for (int i = bxi * blockDim.x + threadIdx.x; i < HW; i += stride) {
      float v = __fmaf_rn(xp[i], sc, sh);
      if (rp) v += res_coef ? __fmaf_rn(rp[i], rsc, rsh) : rp[i];
      if (relu) v = fmaxf(v, 0.f);
      if (post) v *= pm;
      am = fmaxf(am, fabsf(v));
      yp[i] = v;
    }
  }
  if (amax) amax_publish(amax, am);
}

// ReLU gate of element i of a plane: from the bitmask bn_apply wrote (layout there), else from the saved output y, else
// (no residual) recomputed from x exactly as bn_apply did
__device__ __forceinline__ bool relu_on(const unsigned long long* __restrict__ mp, const float* __restrict__ yp, int i, float xv,
                                        float sc, float sh) {
  if (mp) return (mp[(i >> 8) * 4 + (i & 3)] >> ((i >> 2) & 63)) & 1ull;
  return (yp ? yp[i] : __fmaf_rn(xv, sc, sh)) > 0.f;
}

// ReLU gates of the 4 elements of float4 number i4 of a plane (same sources as relu_on)
__device__ __forceinline__ void relu_on4(const unsigned long long* __restrict__ mp, const float* __restrict__ yp, int i4, const float4& xv,
                                         float sc, float sh, bool (&on)[4]) {
  if (mp) {
    const ulonglong2 a = *reinterpret_cast<const ulonglong2*>(mp + (i4 >> 6) * 4);       // wave-uniform address: broadcast loads
    const ulonglong2 b = *reinterpret_cast<const ulonglong2*>(mp + (i4 >> 6) * 4 + 2);
    const int l = i4 & 63;
    on[0] = (a.x >> l) & 1ull; on[1] = (a.y >> l) & 1ull; on[2] = (b.x >> l) & 1ull; on[3] = (b.y >> l) & 1ull;
  } else if (yp) {
    const float4 yv = reinterpret_cast<const float4*>(yp)[i4];
    on[0] = yv.x > 0.f; on[1] = yv.y > 0.f; on[2] = yv.z > 0.f; on[3] = yv.w > 0.f;
  } else {
    on[0] = __fmaf_rn(xv.x, sc, sh) > 0.f; on[1] = __fmaf_rn(xv.y, sc, sh) > 0.f; on[2] = __fmaf_rn(xv.z, sc, sh) > 0.f; on[3] = __fmaf_rn(xv.w, sc, sh) > 0.f;
  }
}

// backward pass 1: ws[2c] += sum dz, ws[2c+1] += sum dz*xhat     grid: (splits, C, N)
// VEC (host: HW % 4 == 0, 16-byte aligned planes): float4 streams
template <bool VEC>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dy, i64 dy_bs, const float* __restrict__ y,
                                                            i64 y_bs, const float* __restrict__ x, i64 x_bs,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            int HW, int chunk, int relu, const unsigned long long* __restrict__ mask,
                                                            double* __restrict__ ws, int rev, const float* __restrict__ post,
                                                            double* __restrict__ det_part = nullptr) {
  // post != NULL: the incoming gradient is the gradient of y * post[n][c] (bn_apply's folded Dropout2d factor): dz = dy * post * gate
  __shared__ double sm[32];
  const int c = rev ? gridDim.y - 1 - blockIdx.y : blockIdx.y, n = rev ? gridDim.z - 1 - blockIdx.z : blockIdx.z;
  const int bxi = rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
  const float mu = mean[c], is = invstd[c];
  const float pm = post ? post[n * gridDim.y + c] : 1.f;
  // ReLU mask: from the saved output y, or (no residual) recomputed bit-identically to bn_apply from x -- saves the y read
  float sc, sh;
  bn_affine(mu, is, gamma[c], beta ? beta[c] : 0.f, sc, sh);
  const i64 base = (i64)c * HW;
  const float* gp = dy + (i64)n * dy_bs + base;
  const float* yp = y ? y + (i64)n * y_bs + base : nullptr;
  const unsigned long long* mp = mask ? mask + ((i64)n * gridDim.y + c) * (HW >> 6) : nullptr;
  const float* xp = x + (i64)n * x_bs + base;
  const int beg = bxi * chunk;
  const int end = min(beg + chunk, HW);
  double s = 0.0, sx = 0.0;
  if (VEC) {
    // four 16-byte loads per operand in flight per thread (as in the apply passes); the sums keep the order of the plain loop
    constexpr int U = 4;
    const int e4 = end >> 2;
    for (int i0 = (beg >> 2) + threadIdx.x; i0 < e4; i0 += U * blockDim.x) {
      float4 gq[U], xq[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i4 = i0 + u * blockDim.x;
        if (i4 < e4) {
          gq[u] = reinterpret_cast<const float4*>(gp)[i4];
          xq[u] = reinterpret_cast<const float4*>(xp)[i4];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i4 = i0 + u * blockDim.x;
        if (i4 >= e4) break;
        float4 g = gq[u];
        const float4 xv = xq[u];
        if (post) { g.x *= pm; g.y *= pm; g.z *= pm; g.w *= pm; }
        if (relu) {
          bool on[4];
          relu_on4(mp, yp, i4, xv, sc, sh, on);
          if (!on[0]) g.x = 0.f;
          if (!on[1]) g.y = 0.f;
          if (!on[2]) g.z = 0.f;
          if (!on[3]) g.w = 0.f;
        }
        s += ((double)g.x + (double)g.y) + ((double)g.z + (double)g.w);
        sx += ((double)g.x * (double)((xv.x - mu) * is) + (double)g.y * (double)((xv.y - mu) * is)) +
              ((double)g.z * (double)((xv.z - mu) * is) + (double)g.w * (double)((xv.w - mu) * is));
      }
    }
  } else {
    for (int i = beg + threadIdx.x; i < end; i += blockDim.x) {
      float dz = gp[i];
      const float xv = xp[i];
      if (post) dz *= pm;
      if (relu && !relu_on(mp, yp, i, xv, sc, sh)) dz = 0.f;
      const float xh = (xv - mu) * is;
      s += (double)dz;
      sx += (double)dz * (double)xh;
    }
  }
  block_sum2_d(s, sx, sm);
  if (threadIdx.x == 0) {
    if (det_part) {                                  // deterministic mode: slot (n, chunk) of channel c (see bn_stats_kernel)
      double* d = det_part + (((i64)c * gridDim.z + n) * gridDim.x + bxi) * 2;
      d[0] = s;
      d[1] = sx;
    } else {
      atomicAdd(&ws[2 * c], s);
      atomicAdd(&ws[2 * c + 1], sx);
    }
  }
}

// the two sums from the data-gradient epilogue's partials part[c][T][2] (pfst_bnb_fuse_t) -> ws[2c], ws[2c+1]     grid: C blocks
// part = (sum dz, sum dz * x):  sum dz * xhat = invstd * (sum dz*x - mean * sum dz), in fp64
__global__ __launch_bounds__(256) void bn_bwd_partials_kernel(const float* __restrict__ part, int T, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, double* __restrict__ ws) {
  __shared__ double sm[32];
  const int c = blockIdx.x;
  double s, sx;
  sum_partial_slots(reinterpret_cast<const float2*>(part) + (i64)c * T, T, s, sx);
  const float mu = mean[c], is = invstd[c];        // requested before the barrier
  block_sum2_d(s, sx, sm);
  if (threadIdx.x == 0) {
    ws[2 * c] = s;
    ws[2 * c + 1] = (double)is * (sx - (double)mu * s);
  }
}

#ifndef PFST_BN_BWD_APPLY_U
#define PFST_BN_BWD_APPLY_U 1
#endif
// backward pass 2     grid: (blocks over HW, C, N)
template <bool VEC>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, i64 dy_bs, const float* __restrict__ y,
                                                           i64 y_bs, const float* __restrict__ x, i64 x_bs,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ dx, i64 dx_bs,
                                                           float* __restrict__ dres, i64 dres_bs, int dres_acc,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           int C, int HW, double inv_count, int relu,
                                                           const unsigned long long* __restrict__ mask, const double* __restrict__ ws, int rev,
                                                           float* __restrict__ amax, const float* __restrict__ post) {
  const int c = rev ? gridDim.y - 1 - blockIdx.y : blockIdx.y, n = rev ? gridDim.z - 1 - blockIdx.z : blockIdx.z;
  const int bxi = rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
  float am = 0.f;                                      // max |dx| written here -> slot group `amax` (f16x3 scale of the gradient operand)
  const float mu = mean[c], is = invstd[c];
  const float pm = post ? post[n * C + c] : 1.f;       // folded Dropout2d factor (see bn_bwd_reduce_kernel)
  // the two projections are subtracted in fp64: dz - mean(dz) cancels heavily when dz has a large common mode
  const double m1 = ws[2 * c] * inv_count, m2 = ws[2 * c + 1] * inv_count;
  const double gs = (double)gamma[c] * (double)is;
  float sc, sh;
  bn_affine(mu, is, gamma[c], beta ? beta[c] : 0.f, sc, sh);
  if (bxi == 0 && n == 0 && threadIdx.x == 0) {
    if (dgamma) dgamma[c] += (float)ws[2 * c + 1];
    if (dbeta) dbeta[c] += (float)ws[2 * c];
  }
  const i64 base = (i64)c * HW;
  const float* gp = dy + (i64)n * dy_bs + base;
  const float* yp = y ? y + (i64)n * y_bs + base : nullptr;
  const unsigned long long* mp = mask ? mask + ((i64)n * gridDim.y + c) * (HW >> 6) : nullptr;
  const float* xp = x + (i64)n * x_bs + base;
  float* dxp = dx + (i64)n * dx_bs + base;
  float* drp = dres ? dres + (i64)n * dres_bs + base : nullptr;
  const int stride = gridDim.x * blockDim.x;
  if (VEC) {
    // PFST_BN_BWD_APPLY_U 16-byte loads per operand in flight per thread (the host sizes the grid to match)
    constexpr int U = PFST_BN_BWD_APPLY_U;
    const int n4 = HW >> 2;
    for (int i0 = bxi * blockDim.x + threadIdx.x; i0 < n4; i0 += U * stride) {
      float4 gq[U], xq[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i4 = i0 + u * stride;
        if (i4 < n4) {
          gq[u] = reinterpret_cast<const float4*>(gp)[i4];
          xq[u] = reinterpret_cast<const float4*>(xp)[i4];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i4 = i0 + u * stride;
        if (i4 >= n4) break;
        float4 g = gq[u];
        const float4 xv = xq[u];
        if (post) { g.x *= pm; g.y *= pm; g.z *= pm; g.w *= pm; }
        if (relu) {
          bool on[4];
          relu_on4(mp, yp, i4, xv, sc, sh, on);
          if (!on[0]) g.x = 0.f;
          if (!on[1]) g.y = 0.f;
          if (!on[2]) g.z = 0.f;
          if (!on[3]) g.w = 0.f;
        }
        float4 o;
        o.x = (float)(gs * ((double)g.x - m1 - (((double)xv.x - (double)mu) * (double)is) * m2));
        o.y = (float)(gs * ((double)g.y - m1 - (((double)xv.y - (double)mu) * (double)is) * m2));
        o.z = (float)(gs * ((double)g.z - m1 - (((double)xv.z - (double)mu) * (double)is) * m2));
        o.w = (float)(gs * ((double)g.w - m1 - (((double)xv.w - (double)mu) * (double)is) * m2));
        reinterpret_cast<float4*>(dxp)[i4] = o;
        am = fmaxf(fmaxf(am, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        if (drp) {
          if (dres_acc) {
            const float4 old = reinterpret_cast<const float4*>(drp)[i4];
            g.x += old.x; g.y += old.y; g.z += old.z; g.w += old.w;
          }
          reinterpret_cast<float4*>(drp)[i4] = g;
        }
      }
    }
  } else {
    for (int i = bxi * blockDim.x + threadIdx.x; i < HW; i += stride) {
      float dz = gp[i];
      const float xv = xp[i];
      if (post) dz *= pm;
      if (relu && !relu_on(mp, yp, i, xv, sc, sh)) dz = 0.f;
      const double xh = ((double)xv - (double)mu) * (double)is;
      const float o = (float)(gs * ((double)dz - m1 - xh * m2));
      dxp[i] = o;
      am = fmaxf(am, fabsf(o));
      if (drp) drp[i] = dres_acc ? drp[i] + dz : dz;
    }
  }
  if (amax) amax_publish(amax, am);
}

// ---- two BatchNorm layers behind ONE gated gradient (pfst_bn_backward_dual) --------------------------------------------------------------
// A stage's first Bottleneck ends in out = relu(bn3(conv3(.)) + bn_d(conv_d(x))) (resnet.py:298-307): bn3 and the downsample branch's BN both
// receive g = [out > 0] dL/dout.  Layer by layer that is four passes reading g (two reductions, two apply passes: 10 N of traffic); here one
// reduction reads (g, xa, xb) and one apply pass writes both input gradients: 8 N.  Same arithmetic as the single-layer kernels, element by
// element (fp64 projections, the same summation order inside a workgroup).
struct BnDualSide {
  const float* x; i64 x_bs;             // the layer's pre-BN tensor
  const float* mean; const float* invstd; const float* gamma;
  float* dx; i64 dx_bs;                 // out: dL/dx
  float* dgamma; float* dbeta;          // += (may be NULL)
  double* ws;                           // [2 C]: (sum g, sum g * xhat)
  float* amax;                          // NULL, or the slot group receiving max |dx|
};

// reduction: ws_a / ws_b [2c], [2c+1] += over the chunk     grid: (splits, C, N); planes 16-byte aligned, HW % 256 == 0
__global__ __launch_bounds__(256) void bn_bwd_reduce_dual_kernel(const float* __restrict__ dy, i64 dy_bs, const unsigned long long* __restrict__ mask,
                                                                 BnDualSide a, BnDualSide b, int HW, int chunk, int rev) {
  __shared__ double sm[32];
  const int c = rev ? gridDim.y - 1 - blockIdx.y : blockIdx.y, n = rev ? gridDim.z - 1 - blockIdx.z : blockIdx.z;
  const int bxi = rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
  const float mua = a.mean[c], isa = a.invstd[c], mub = b.mean[c], isb = b.invstd[c];
  const i64 base = (i64)c * HW;
  const float* gp = dy + (i64)n * dy_bs + base;
  const unsigned long long* mp = mask + ((i64)n * gridDim.y + c) * (HW >> 6);
  const float* xap = a.x + (i64)n * a.x_bs + base;
  const float* xbp = b.x + (i64)n * b.x_bs + base;
  const int beg = bxi * chunk;
  const int end = min(beg + chunk, HW);
  double s = 0.0, sa = 0.0, sb = 0.0;
  constexpr int U = 4;
  const int e4 = end >> 2;
  for (int i0 = (beg >> 2) + threadIdx.x; i0 < e4; i0 += U * blockDim.x) {
    float4 gq[U], aq[U], bq[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i4 = i0 + u * blockDim.x;
      if (i4 < e4) {
        gq[u] = reinterpret_cast<const float4*>(gp)[i4];
        aq[u] = reinterpret_cast<const float4*>(xap)[i4];
        bq[u] = reinterpret_cast<const float4*>(xbp)[i4];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i4 = i0 + u * blockDim.x;
      if (i4 >= e4) break;
      float4 g = gq[u];
      const float4 xa = aq[u], xb = bq[u];
      bool on[4];
      relu_on4(mp, nullptr, i4, xa, 0.f, 0.f, on);
      if (!on[0]) g.x = 0.f;
      if (!on[1]) g.y = 0.f;
      if (!on[2]) g.z = 0.f;
      if (!on[3]) g.w = 0.f;
      s += ((double)g.x + (double)g.y) + ((double)g.z + (double)g.w);
      sa += ((double)g.x * (double)((xa.x - mua) * isa) + (double)g.y * (double)((xa.y - mua) * isa)) +
            ((double)g.z * (double)((xa.z - mua) * isa) + (double)g.w * (double)((xa.w - mua) * isa));
      sb += ((double)g.x * (double)((xb.x - mub) * isb) + (double)g.y * (double)((xb.y - mub) * isb)) +
            ((double)g.z * (double)((xb.z - mub) * isb) + (double)g.w * (double)((xb.w - mub) * isb));
    }
  }
  double s2 = s;
  block_sum2_d(s, sa, sm);
  __syncthreads();
  block_sum2_d(s2, sb, sm);
  if (threadIdx.x == 0) {
    atomicAdd(&a.ws[2 * c], s);
    atomicAdd(&a.ws[2 * c + 1], sa);
    atomicAdd(&b.ws[2 * c], s2);
    atomicAdd(&b.ws[2 * c + 1], sb);
  }
}

// apply: dxa, dxb from one read of g     grid: (blocks over HW, C, N)
__global__ __launch_bounds__(256) void bn_bwd_apply_dual_kernel(const float* __restrict__ dy, i64 dy_bs, const unsigned long long* __restrict__ mask,
                                                                BnDualSide a, BnDualSide b, int C, int HW, double inv_count, int rev) {
  const int c = rev ? C - 1 - blockIdx.y : blockIdx.y, n = rev ? gridDim.z - 1 - blockIdx.z : blockIdx.z;
  const int bxi = rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
  float ama = 0.f, amb = 0.f;
  const float mua = a.mean[c], isa = a.invstd[c], mub = b.mean[c], isb = b.invstd[c];
  const double m1a = a.ws[2 * c] * inv_count, m2a = a.ws[2 * c + 1] * inv_count;
  const double m1b = b.ws[2 * c] * inv_count, m2b = b.ws[2 * c + 1] * inv_count;
  const double gsa = (double)a.gamma[c] * (double)isa, gsb = (double)b.gamma[c] * (double)isb;
  if (bxi == 0 && n == 0 && threadIdx.x == 0) {
    if (a.dgamma) a.dgamma[c] += (float)a.ws[2 * c + 1];
    if (a.dbeta) a.dbeta[c] += (float)a.ws[2 * c];
    if (b.dgamma) b.dgamma[c] += (float)b.ws[2 * c + 1];
    if (b.dbeta) b.dbeta[c] += (float)b.ws[2 * c];
  }
  const i64 base = (i64)c * HW;
  const float* gp = dy + (i64)n * dy_bs + base;
  const unsigned long long* mp = mask + ((i64)n * C + c) * (HW >> 6);
  const float* xap = a.x + (i64)n * a.x_bs + base;
  const float* xbp = b.x + (i64)n * b.x_bs + base;
  float* dap = a.dx + (i64)n * a.dx_bs + base;
  float* dbp = b.dx + (i64)n * b.dx_bs + base;
  const int stride = gridDim.x * blockDim.x;
  const int n4 = HW >> 2;
  for (int i4 = bxi * blockDim.x + threadIdx.x; i4 < n4; i4 += stride) {
    float4 g = reinterpret_cast<const float4*>(gp)[i4];
    const float4 xa = reinterpret_cast<const float4*>(xap)[i4];
    const float4 xb = reinterpret_cast<const float4*>(xbp)[i4];
    bool on[4];
    relu_on4(mp, nullptr, i4, xa, 0.f, 0.f, on);
    if (!on[0]) g.x = 0.f;
    if (!on[1]) g.y = 0.f;
    if (!on[2]) g.z = 0.f;
    if (!on[3]) g.w = 0.f;
    float4 o;
    o.x = (float)(gsa * ((double)g.x - m1a - (((double)xa.x - (double)mua) * (double)isa) * m2a));
    o.y = (float)(gsa * ((double)g.y - m1a - (((double)xa.y - (double)mua) * (double)isa) * m2a));
    o.z = (float)(gsa * ((double)g.z - m1a - (((double)xa.z - (double)mua) * (double)isa) * m2a));
    o.w = (float)(gsa * ((double)g.w - m1a - (((double)xa.w - (double)mua) * (double)isa) * m2a));
    reinterpret_cast<float4*>(dap)[i4] = o;
    ama = fmaxf(fmaxf(ama, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    o.x = (float)(gsb * ((double)g.x - m1b - (((double)xb.x - (double)mub) * (double)isb) * m2b));
    o.y = (float)(gsb * ((double)g.y - m1b - (((double)xb.y - (double)mub) * (double)isb) * m2b));
    o.z = (float)(gsb * ((double)g.z - m1b - (((double)xb.z - (double)mub) * (double)isb) * m2b));
    o.w = (float)(gsb * ((double)g.w - m1b - (((double)xb.w - (double)mub) * (double)isb) * m2b));
    reinterpret_cast<float4*>(dbp)[i4] = o;
    amb = fmaxf(fmaxf(amb, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
  }
  if (a.amax) amax_publish(a.amax, ama);
  __syncthreads();                     // (amax_publish's LDS slots are reused by the second call)
  if (b.amax) amax_publish(b.amax, amb);
}

// after the two sums are in ws: parameter gradients + the per-channel record a consumer needs to apply the second pass itself
__global__ void bn_bwd_rec_kernel(const double* __restrict__ ws, const float* __restrict__ mean, const float* __restrict__ invstd,
                                  const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ dgamma,
                                  float* __restrict__ dbeta, int C, double inv_count, pfst_bn_bwd_rec_t* __restrict__ rec) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (dgamma) dgamma[c] += (float)ws[2 * c + 1];
  if (dbeta) dbeta[c] += (float)ws[2 * c];
  pfst_bn_bwd_rec_t r;
  r.m1 = ws[2 * c] * inv_count;
  r.m2 = ws[2 * c + 1] * inv_count;
  r.gs = (double)gamma[c] * (double)invstd[c];
  r.mu = mean[c];
  r.is = invstd[c];
  bn_affine(mean[c], invstd[c], gamma[c], beta[c], r.sc, r.sh);
  rec[c] = r;
}

// split one HW plane into `splits` chunks (multiples of 4) so the reduction launches ~BN_SPLIT_TARGET blocks
inline void split_for(int HW, int C, int N, int& splits, int& chunk) {
  i64 planes = (i64)C * N;
  splits = (int)(BN_SPLIT_TARGET / (planes > 0 ? planes : 1));
  const int max_splits = (HW + 1023) / 1024;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  chunk = (HW + splits - 1) / splits;
  chunk = (chunk + 3) & ~3;
  splits = (HW + chunk - 1) / chunk;
}

// out (+)= g where the ReLU bitmask has the element's bit (the identity branch of a residual block written out after all: engine.Var._flush_pending)
__global__ __launch_bounds__(256) void relu_gate_kernel(const float* __restrict__ g, i64 g_bs, const unsigned long long* __restrict__ mask,
                                                        float* __restrict__ out, i64 out_bs, int C, int HW, int accumulate) {
  const int c = blockIdx.y, n = blockIdx.z;
  const float* gp = g + (i64)n * g_bs + (i64)c * HW;
  float* op = out + (i64)n * out_bs + (i64)c * HW;
  const unsigned long long* mp = mask + ((i64)n * C + c) * (HW >> 6);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
    const float v = relu_on(mp, nullptr, i, 0.f, 0.f, 0.f) ? gp[i] : 0.f;
    op[i] = accumulate ? op[i] + v : v;
  }
}

// deterministic mode: the [C][N * splits][2] partial slots of a reduction launch (NULL when the mode is off; `ok` false if the scratch failed)
inline double* bn_det_part(int C, int N, int splits, hipStream_t s, bool& ok) {
  ok = true;
  if (!pfst_deterministic()) return nullptr;
  double* p = static_cast<double*>(pfst_det_scratch((size_t)C * N * splits * 2 * sizeof(double), s));
  ok = p != nullptr;
  return p;
}

}  // namespace

extern "C" int pfst_bn_stats(const float* x, long long x_bs, int N, int C, int HW, float* mean, float* invstd,
                             float* running_mean, float* running_var, float momentum, float eps, double* ws,
                             const float* gamma, const float* beta, float* coef, pfst_stream_t stream) {
  PFST_CHECK_ARG(!coef || (gamma && beta));
  PFST_CHECK_ARG(x && mean && invstd && ws && N > 0 && C > 0 && HW > 0 && C <= 65535 && N <= 65535);
  PFST_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr));
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(ws, 0, sizeof(double) * 2 * C, s) != hipSuccess) return PFST_ERR_LAUNCH;
  int splits, chunk;
  split_for(HW, C, N, splits, chunk);
  bool det_ok;
  double* const det = bn_det_part(C, N, splits, s, det_ok);
  PFST_CHECK_ARG(det_ok);
  hipLaunchKernelGGL(bn_stats_kernel, dim3(splits, C, N), dim3(256), 0, s, x, x_bs, HW, chunk, ws, det);
  if (det) hipLaunchKernelGGL(bn_det_sum_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, det, N * splits, C, ws);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, ws, C, (double)N * HW, mean, invstd, running_mean,
                     running_var, momentum, eps, gamma, beta, reinterpret_cast<float4*>(coef));
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

extern "C" int pfst_bn_finalize_partials(const float* partials, int T, int C, double count, float* mean, float* invstd,
                                         float* running_mean, float* running_var, float momentum, float eps,
                                         const float* gamma, const float* beta, float* coef, const float* minmax, int relu,
                                         float* y_amax, pfst_stream_t stream) {
  PFST_CHECK_ARG(partials && mean && invstd && T > 0 && C > 0 && count > 0);
  PFST_CHECK_ARG(!coef || (gamma && beta));
  PFST_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr));
  PFST_CHECK_ARG((minmax == nullptr) == (y_amax == nullptr) && (!minmax || (gamma && beta)));
  if (T >= 4096)
    hipLaunchKernelGGL(bn_finalize_partials_kernel<1024>, dim3(C), dim3(1024), 0, (hipStream_t)stream, partials, T, count, mean, invstd,
                       running_mean, running_var, momentum, eps, gamma, beta, reinterpret_cast<float4*>(coef), minmax, relu, y_amax);
  else
    hipLaunchKernelGGL(bn_finalize_partials_kernel<256>, dim3(C), dim3(256), 0, (hipStream_t)stream, partials, T, count, mean, invstd,
                       running_mean, running_var, momentum, eps, gamma, beta, reinterpret_cast<float4*>(coef), minmax, relu, y_amax);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

// Traversal order of the streaming BatchNorm kernels (bit 0: normalise pass, bit 1: backward reduction, bit 2: backward apply run from the
// END of the tensor).  A pass that starts where its producer stopped finds the producer's last ~100-200 MB in the Infinity Cache.
static constexpr int pfst_bn_order() { return 3; }      // measured (round 3): 3 and 7 -0.4 % on the step against 0 = all ascending

extern "C" int pfst_bn_apply(const float* x, long long x_bs, const float* residual, long long res_bs, float* y, long long y_bs,
                             const float* mean, const float* invstd, const float* gamma, const float* beta,
                             int N, int C, int HW, int relu, unsigned long long* relu_mask, float* y_amax, const float* post_scale,
                             const float* residual_coef, pfst_stream_t stream) {
  PFST_CHECK_ARG(x && y && mean && invstd && gamma && beta && N > 0 && C > 0 && HW > 0 && C <= 65535 && N <= 65535);
  PFST_CHECK_ARG(!residual_coef || residual);
  PFST_CHECK_ARG(!post_scale || !residual);          // the folded Dropout2d factor belongs to a plain conv -> BN -> ReLU layer
  // the bitmask comes out of the float4 path only: whole 256-element groups per wave, 16-byte aligned planes
  PFST_CHECK_ARG(!relu_mask || (relu && HW % 256 == 0 && ((x_bs | y_bs | (residual ? res_bs : 0)) & 3) == 0 &&
                                (((uintptr_t)x | (uintptr_t)y | (uintptr_t)residual) & 15) == 0));
  int gx = cdiv(HW, 256 * 4 * 4);
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(bn_apply_kernel, dim3(gx, C, N), dim3(256), 0, (hipStream_t)stream, x, x_bs, residual, res_bs, y, y_bs, mean,
                     invstd, gamma, beta, C, HW, relu, relu_mask, pfst_bn_order() & 1, y_amax, post_scale,
                     reinterpret_cast<const float4*>(residual_coef));
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

extern "C" int pfst_bn_backward(const float* dy, long long dy_bs, const float* y, long long y_bs, const float* x, long long x_bs,
                                const float* mean, const float* invstd, const float* gamma, const float* beta,
                                float* dx, long long dx_bs, float* dres, long long dres_bs, int dres_accumulate,
                                float* dgamma, float* dbeta, int N, int C, int HW, int relu, const unsigned long long* relu_mask,
                                double* ws, const float* bwd_partials, int bwd_slots, float* dx_amax, const float* post_scale,
                                pfst_stream_t stream) {
  PFST_CHECK_ARG(dy && x && mean && invstd && gamma && dx && ws && N > 0 && C > 0 && HW > 0 && C <= 65535 && N <= 65535);
  PFST_CHECK_ARG(!post_scale || (!dres && !bwd_partials));     // no residual branch, no fused sums (they were formed without the factor)
  PFST_CHECK_ARG(!relu_mask || (relu && HW % 256 == 0));
  PFST_CHECK_ARG(!relu || relu_mask || y || beta);   // ReLU mask from y, or recomputed from x with beta (no residual)
  PFST_CHECK_ARG(!bwd_partials || bwd_slots > 0);
  hipStream_t s = (hipStream_t)stream;
  const bool fused = bwd_partials != nullptr;      // the sums came out of the launch that produced dy: no reduction pass
  if (fused) hipLaunchKernelGGL(bn_bwd_partials_kernel, dim3(C), dim3(256), 0, s, bwd_partials, bwd_slots, mean, invstd, ws);
  else if (hipMemsetAsync(ws, 0, sizeof(double) * 2 * C, s) != hipSuccess) return PFST_ERR_LAUNCH;
  int splits, chunk;
  split_for(HW, C, N, splits, chunk);
  bool det_ok = true;
  double* const det = fused ? nullptr : bn_det_part(C, N, splits, s, det_ok);      // deterministic mode: slots + ordered sum instead of atomics
  PFST_CHECK_ARG(det_ok);
  const bool vec = (HW & 3) == 0 && ((dy_bs | x_bs | dx_bs | (y ? y_bs : 0) | (dres ? dres_bs : 0)) & 3) == 0 &&
                   (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dx | (uintptr_t)y | (uintptr_t)dres) & 15) == 0;
  int gx = cdiv(HW, 256 * 4 * (vec ? PFST_BN_BWD_APPLY_U : 1));
  if (gx < 1) gx = 1;
  const double inv_count = 1.0 / ((double)N * HW);
  if (vec) {
    if (!fused)
      hipLaunchKernelGGL(bn_bwd_reduce_kernel<true>, dim3(splits, C, N), dim3(256), 0, s, dy, dy_bs, y, y_bs, x, x_bs, mean, invstd, gamma,
                         beta, HW, chunk, relu, relu_mask, ws, (pfst_bn_order() >> 1) & 1, post_scale, det);
    if (!fused && det) hipLaunchKernelGGL(bn_det_sum_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, det, N * splits, C, ws);
    hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3(gx, C, N), dim3(256), 0, s, dy, dy_bs, y, y_bs, x, x_bs, mean, invstd, gamma, beta,
                       dx, dx_bs, dres, dres_bs, dres_accumulate, dgamma, dbeta, C, HW, inv_count, relu, relu_mask, ws, (pfst_bn_order() >> 2) & 1, dx_amax, post_scale);
  } else {
    if (!fused)
      hipLaunchKernelGGL(bn_bwd_reduce_kernel<false>, dim3(splits, C, N), dim3(256), 0, s, dy, dy_bs, y, y_bs, x, x_bs, mean, invstd, gamma,
                         beta, HW, chunk, relu, relu_mask, ws, (pfst_bn_order() >> 1) & 1, post_scale, det);
    if (!fused && det) hipLaunchKernelGGL(bn_det_sum_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, det, N * splits, C, ws);
    hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(gx, C, N), dim3(256), 0, s, dy, dy_bs, y, y_bs, x, x_bs, mean, invstd, gamma, beta,
                       dx, dx_bs, dres, dres_bs, dres_accumulate, dgamma, dbeta, C, HW, inv_count, relu, relu_mask, ws, (pfst_bn_order() >> 2) & 1, dx_amax, post_scale);
  }
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

extern "C" int pfst_bn_backward_dual(const float* dy, long long dy_bs, const unsigned long long* relu_mask,
                                     const float* xa, long long xa_bs, const float* mean_a, const float* invstd_a, const float* gamma_a,
                                     float* dxa, long long dxa_bs, float* dgamma_a, float* dbeta_a, double* ws_a,
                                     const float* bwd_partials_a, int bwd_slots_a, float* dxa_amax,
                                     const float* xb, long long xb_bs, const float* mean_b, const float* invstd_b, const float* gamma_b,
                                     float* dxb, long long dxb_bs, float* dgamma_b, float* dbeta_b, double* ws_b, float* dxb_amax,
                                     int N, int C, int HW, pfst_stream_t stream) {
  PFST_CHECK_ARG(dy && relu_mask && xa && mean_a && invstd_a && gamma_a && dxa && ws_a && xb && mean_b && invstd_b && gamma_b && dxb && ws_b);
  PFST_CHECK_ARG(ws_a != ws_b && dxa != dxb && N > 0 && C > 0 && HW > 0 && C <= 65535 && N <= 65535);
  PFST_CHECK_ARG(!bwd_partials_a || bwd_slots_a > 0);
  if (pfst_deterministic() || HW % 256 != 0 || ((dy_bs | xa_bs | xb_bs | dxa_bs | dxb_bs) & 3) != 0 ||
      (((uintptr_t)dy | (uintptr_t)xa | (uintptr_t)xb | (uintptr_t)dxa | (uintptr_t)dxb) & 15) != 0) {
    pfst_set_error(__FILE__, __LINE__, "dual BatchNorm backward needs HW % 256 == 0, 16-byte aligned planes and the default (atomic) reductions: "
                                       "run the two layers through pfst_bn_backward");
    return PFST_ERR_UNSUPPORTED;
  }
  hipStream_t s = (hipStream_t)stream;
  const BnDualSide a{xa, (i64)xa_bs, mean_a, invstd_a, gamma_a, dxa, (i64)dxa_bs, dgamma_a, dbeta_a, ws_a, dxa_amax};
  const BnDualSide b{xb, (i64)xb_bs, mean_b, invstd_b, gamma_b, dxb, (i64)dxb_bs, dgamma_b, dbeta_b, ws_b, dxb_amax};
  int splits, chunk;
  split_for(HW, C, N, splits, chunk);
  if (hipMemsetAsync(ws_b, 0, sizeof(double) * 2 * C, s) != hipSuccess) return PFST_ERR_LAUNCH;
  if (bwd_partials_a) {
    // layer a's sums came out of the launch that completed dL/dout (pfst_bnb_fuse_t): only layer b's reduction reads the tensors
    const float* none = nullptr;
    double* nodet = nullptr;
    hipLaunchKernelGGL(bn_bwd_partials_kernel, dim3(C), dim3(256), 0, s, bwd_partials_a, bwd_slots_a, mean_a, invstd_a, ws_a);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<true>, dim3(splits, C, N), dim3(256), 0, s, dy, (i64)dy_bs, none, (i64)0, xb, (i64)xb_bs, mean_b, invstd_b,
                       gamma_b, none, HW, chunk, 1, relu_mask, ws_b, (pfst_bn_order() >> 1) & 1, none, nodet);
  } else {
    if (hipMemsetAsync(ws_a, 0, sizeof(double) * 2 * C, s) != hipSuccess) return PFST_ERR_LAUNCH;
    hipLaunchKernelGGL(bn_bwd_reduce_dual_kernel, dim3(splits, C, N), dim3(256), 0, s, dy, (i64)dy_bs, relu_mask, a, b, HW, chunk,
                       (pfst_bn_order() >> 1) & 1);
  }
  const int gx = cdiv(HW, 256 * 4);
  hipLaunchKernelGGL(bn_bwd_apply_dual_kernel, dim3(gx, C, N), dim3(256), 0, s, dy, (i64)dy_bs, relu_mask, a, b, C, HW, 1.0 / ((double)N * HW),
                     (pfst_bn_order() >> 2) & 1);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

extern "C" int pfst_bn_backward_sums(const float* dy, long long dy_bs, const float* x, long long x_bs, const float* mean, const float* invstd,
                                     const float* gamma, const float* beta, float* dgamma, float* dbeta, int N, int C, int HW,
                                     double* ws, const float* bwd_partials, int bwd_slots, pfst_bn_bwd_rec_t* rec, pfst_stream_t stream) {
  PFST_CHECK_ARG(dy && x && mean && invstd && gamma && beta && ws && rec && N > 0 && C > 0 && HW > 0 && C <= 65535 && N <= 65535);
  PFST_CHECK_ARG(!bwd_partials || bwd_slots > 0);
  hipStream_t s = (hipStream_t)stream;
  if (bwd_partials) {
    hipLaunchKernelGGL(bn_bwd_partials_kernel, dim3(C), dim3(256), 0, s, bwd_partials, bwd_slots, mean, invstd, ws);
  } else {
    if (hipMemsetAsync(ws, 0, sizeof(double) * 2 * C, s) != hipSuccess) return PFST_ERR_LAUNCH;
    int splits, chunk;
    split_for(HW, C, N, splits, chunk);
    const bool vec = (HW & 3) == 0 && ((dy_bs | x_bs) & 3) == 0 && (((uintptr_t)dy | (uintptr_t)x) & 15) == 0;
    const float* none = nullptr;
    const unsigned long long* nomask = nullptr;
    bool det_ok;
    double* const det = bn_det_part(C, N, splits, s, det_ok);
    PFST_CHECK_ARG(det_ok);
    if (vec)
      hipLaunchKernelGGL(bn_bwd_reduce_kernel<true>, dim3(splits, C, N), dim3(256), 0, s, dy, (i64)dy_bs, none, (i64)0, x, (i64)x_bs, mean, invstd, gamma,
                         beta, HW, chunk, 1, nomask, ws, (pfst_bn_order() >> 1) & 1, none, det);
    else
      hipLaunchKernelGGL(bn_bwd_reduce_kernel<false>, dim3(splits, C, N), dim3(256), 0, s, dy, (i64)dy_bs, none, (i64)0, x, (i64)x_bs, mean, invstd, gamma,
                         beta, HW, chunk, 1, nomask, ws, (pfst_bn_order() >> 1) & 1, none, det);
    if (det) hipLaunchKernelGGL(bn_det_sum_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, det, N * splits, C, ws);
  }
  hipLaunchKernelGGL(bn_bwd_rec_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, ws, mean, invstd, gamma, beta, dgamma, dbeta, C,
                     1.0 / ((double)N * HW), rec);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

extern "C" int pfst_relu_gate(const float* g, long long g_bs, const unsigned long long* relu_mask, float* out, long long out_bs, int N, int C,
                              int HW, int accumulate, pfst_stream_t stream) {
  PFST_CHECK_ARG(g && relu_mask && out && N > 0 && C > 0 && HW > 0 && HW % 256 == 0 && C <= 65535 && N <= 65535);
  hipLaunchKernelGGL(relu_gate_kernel, dim3(cdiv(HW, 1024), C, N), dim3(256), 0, (hipStream_t)stream, g, (i64)g_bs, relu_mask, out, (i64)out_bs, C, HW,
                     accumulate);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}
