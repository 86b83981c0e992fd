// Elementwise utilities, max-pool, bilinear resize, global average pool, channel dropout.
// Reference ops: nn.MaxPool2d(3,2,1) resnet.py:638; rsiseg/ops/wrappers.py:8-27 (F.interpolate bilinear,
// align_corners=False) at sep_aspp_head.py:81-100; nn.AdaptiveAvgPool2d(1) aspp_head.py:69-77;
// nn.Dropout2d decode_head.py:103-107,242-247.  All HBM-bound; one pass each.
#include "common.h"
#include "amax.h"
#include "../../include/pfst_hip.h"

namespace {

__global__ void fill_kernel(float* __restrict__ p, i64 n, float v) {
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void axpy_kernel(float* y, const float* x, float a, i64 n) {   // x may alias y (in-place scaling)
  const i64 stride = (i64)gridDim.x * blockDim.x;
  i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if ((((uintptr_t)y | (uintptr_t)x) & 15) == 0) {
    const i64 n4 = n >> 2;
    for (; i < n4; i += stride) {
      float4 a4 = reinterpret_cast<float4*>(y)[i];
      const float4 b4 = reinterpret_cast<const float4*>(x)[i];
      a4.x = fmaf(a, b4.x, a4.x); a4.y = fmaf(a, b4.y, a4.y); a4.z = fmaf(a, b4.z, a4.z); a4.w = fmaf(a, b4.w, a4.w);
      reinterpret_cast<float4*>(y)[i] = a4;
    }
    for (i64 j = (n4 << 2) + (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) y[j] = fmaf(a, x[j], y[j]);
  } else {
    for (; i < n; i += stride) y[i] = fmaf(a, x[i], y[i]);
  }
}
__global__ void i64_to_u8_kernel(const long long* __restrict__ s, unsigned char* __restrict__ d, i64 n) {
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) d[i] = (unsigned char)s[i];
}
__global__ void u8_to_i64_kernel(const unsigned char* __restrict__ s, long long* __restrict__ d, i64 n) {
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) d[i] = (long long)s[i];
}

// ---- max pool 3x3 stride 2 pad 1; first maximum in row-major scan order wins (torch CPU/GPU semantics)
// bnl != NULL: x is the PRE-normalisation output of the conv -> BN -> ReLU layer in front of the pool (stem.6); coef[c] = (mean, invstd, sc, sh)
// of that layer, applied to every tap as it is read (bn_apply's pinned arithmetic) -- the normalised tensor is never written
__global__ void maxpool_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ idx, int H, int W,
                               int Ho, int Wo, const float4* __restrict__ bnl, int C, float* __restrict__ amax) {
  // amax != NULL: max |y| of this launch's outputs -> that slot group (amax.h): the f16x3 GEMMs of layer1 read the pooled map
  float am = 0.f;
  const int nc = blockIdx.y;
  const float* xp = x + (i64)nc * H * W;
  const float bsc = bnl ? bnl[nc % C].z : 1.f, bsh = bnl ? bnl[nc % C].w : 0.f;
  for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < Ho * Wo; o += gridDim.x * blockDim.x) {
    const int oy = o / Wo, ox = o - oy * Wo;
    float best = -INFINITY;
    int bt = 0;
    bool first = true;
#pragma unroll
    for (int ty = 0; ty < 3; ++ty) {
      const int sy = oy * 2 - 1 + ty;
      if (sy < 0 || sy >= H) continue;
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) {
        const int sx = ox * 2 - 1 + tx;
        if (sx < 0 || sx >= W) continue;
        float v = xp[(i64)sy * W + sx];
        if (bnl) v = fmaxf(__fmaf_rn(v, bsc, bsh), 0.f);
        if (first || v > best || v != v) { best = v; bt = ty * 3 + tx; first = false; }
      }
    }
    y[(i64)nc * Ho * Wo + o] = best;
    idx[(i64)nc * Ho * Wo + o] = (unsigned char)bt;
    am = fmaxf(am, fabsf(best));
  }
  if (amax) amax_publish(amax, am);
}
// the same pool, two adjacent outputs per thread (W % 4 == 0, 16-byte aligned planes): their windows cover input columns 4k-1 .. 4k+3 of three
// rows -- one 16-byte load + one scalar per row instead of nine scalar loads per output, every tap normalised once -- scanned per output in
// the same row-major order with the same comparison (first maximum wins, NaN propagates): identical values and indices.
__global__ __launch_bounds__(256) void maxpool_pair_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ idx,
                                                           int H, int W, int Ho, int Wo, const float4* __restrict__ bnl, int C,
                                                           float* __restrict__ amax) {
  float am = 0.f;
  const int nc = blockIdx.y;
  const float* xp = x + (i64)nc * H * W;
  const float bsc = bnl ? bnl[nc % C].z : 1.f, bsh = bnl ? bnl[nc % C].w : 0.f;
  const int W2 = Wo >> 1;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < Ho * W2; q += gridDim.x * blockDim.x) {
    const int oy = q / W2, k = q - oy * W2;
    float v[3][5];                       // columns 4k-1 .. 4k+3 of rows 2oy-1 .. 2oy+1
    bool rok[3];
#pragma unroll
    for (int ty = 0; ty < 3; ++ty) {
      const int sy = oy * 2 - 1 + ty;
      rok[ty] = sy >= 0 && sy < H;
      const float* row = xp + (i64)(rok[ty] ? sy : 0) * W + 4 * k;
      const float4 m = *reinterpret_cast<const float4*>(row);
      const float l = row[k > 0 ? -1 : 0];
      v[ty][0] = l; v[ty][1] = m.x; v[ty][2] = m.y; v[ty][3] = m.z; v[ty][4] = m.w;
      if (bnl) {
#pragma unroll
        for (int i = 0; i < 5; ++i) v[ty][i] = fmaxf(__fmaf_rn(v[ty][i], bsc, bsh), 0.f);
      }
    }
    float best[2];
    int bt[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {        // output ox = 2k + e: taps at columns 2ox-1 .. 2ox+1 = registers 2e .. 2e+2
      float b = -INFINITY;
      int t = 0;
      bool first = true;
#pragma unroll
      for (int ty = 0; ty < 3; ++ty) {
        if (!rok[ty]) continue;
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) {
          if (e == 0 && tx == 0 && k == 0) continue;          // column -1
          const float u = v[ty][2 * e + tx];
          if (first || u > b || u != u) { b = u; t = ty * 3 + tx; first = false; }
        }
      }
      best[e] = b;
      bt[e] = t;
      am = fmaxf(am, fabsf(b));
    }
    const i64 o = (i64)nc * Ho * Wo + (i64)oy * Wo + 2 * k;
    *reinterpret_cast<float2*>(y + o) = make_float2(best[0], best[1]);
    *reinterpret_cast<uchar2*>(idx + o) = make_uchar2((unsigned char)bt[0], (unsigned char)bt[1]);
  }
  if (amax) amax_publish(amax, am);
}
__global__ void maxpool_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ idx, float* __restrict__ dx,
                                   int H, int W, int Ho, int Wo) {
  const int nc = blockIdx.y;
  const float* gp = dy + (i64)nc * Ho * Wo;
  const unsigned char* ip = idx + (i64)nc * Ho * Wo;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < H * W; i += gridDim.x * blockDim.x) {
    const int iy = i / W, ix = i - iy * W;
    float acc = 0.f;
#pragma unroll
    for (int ty = 0; ty < 3; ++ty) {
      const int ny = iy + 1 - ty;  // = 2*oy
      if (ny < 0 || (ny & 1)) continue;
      const int oy = ny >> 1;
      if (oy >= Ho) continue;
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) {
        const int nx = ix + 1 - tx;
        if (nx < 0 || (nx & 1)) continue;
        const int ox = nx >> 1;
        if (ox >= Wo) continue;
        if (ip[oy * Wo + ox] == ty * 3 + tx) acc += gp[oy * Wo + ox];
      }
    }
    dx[(i64)nc * H * W + i] = acc;
  }
}

// even H, W: one thread per 2x2 input block (a, b).  Only the windows (a,b), (a,b+1), (a+1,b), (a+1,b+1) reach it, each routing its
// gradient to the ONE pixel its saved arg-max names: 4 index bytes + 4 gradients in, two 8-byte stores out.
__global__ __launch_bounds__(256) void maxpool_bwd2x2_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ idx,
                                                             float* __restrict__ dx, int H, int W) {
  const int nc = blockIdx.y;
  const int Ho = H >> 1, Wo = W >> 1;                 // (H + 2 - 3) / 2 + 1 for even H
  const float* gp = dy + (i64)nc * Ho * Wo;
  const unsigned char* ip = idx + (i64)nc * Ho * Wo;
  float* xp = dx + (i64)nc * H * W;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < Ho * Wo; i += gridDim.x * blockDim.x) {
    const int a = i / Wo, b = i - a * Wo;
    const bool hb = b + 1 < Wo, ha = a + 1 < Ho;
    const int k00 = ip[i], k01 = hb ? ip[i + 1] : -1, k10 = ha ? ip[i + Wo] : -1, k11 = (ha && hb) ? ip[i + Wo + 1] : -1;
    const float g00 = gp[i], g01 = hb ? gp[i + 1] : 0.f, g10 = ha ? gp[i + Wo] : 0.f, g11 = (ha && hb) ? gp[i + Wo + 1] : 0.f;
    // window (oy, ox) covers rows 2oy-1..2oy+1: tap index k = ty*3 + tx
    const float o00 = k00 == 4 ? g00 : 0.f;                                                      // (2a,   2b)
    const float o01 = (k00 == 5 ? g00 : 0.f) + (k01 == 3 ? g01 : 0.f);                           // (2a,   2b+1)
    const float o10 = (k00 == 7 ? g00 : 0.f) + (k10 == 1 ? g10 : 0.f);                           // (2a+1, 2b)
    const float o11 = (k00 == 8 ? g00 : 0.f) + (k01 == 6 ? g01 : 0.f) + (k10 == 2 ? g10 : 0.f) + (k11 == 0 ? g11 : 0.f);
    *reinterpret_cast<float2*>(xp + (i64)(2 * a) * W + 2 * b) = make_float2(o00, o01);
    *reinterpret_cast<float2*>(xp + (i64)(2 * a + 1) * W + 2 * b) = make_float2(o10, o11);
  }
}

// ---- bilinear resize, align_corners=False.  grid: (blocks over Ho*Wo, C, N)
__global__ void resize_bilinear_kernel(const float* __restrict__ x, i64 x_bs, float* __restrict__ y, i64 y_bs, int C, int Hi, int Wi,
                                       int Ho, int Wo, float sh, float sw) {
  const int c = blockIdx.y, n = blockIdx.z;
  const float* xp = x + (i64)n * x_bs + (i64)c * Hi * Wi;
  float* yp = y + (i64)n * y_bs + (i64)c * Ho * Wo;
  for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < Ho * Wo; o += gridDim.x * blockDim.x) {
    const int oy = o / Wo, ox = o - oy * Wo;
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    bilin_src(oy, sh, Hi, y0, y1, ly0, ly1);
    bilin_src(ox, sw, Wi, x0, x1, lx0, lx1);
    const float v = bilin_blend(xp[y0 * Wi + x0], xp[y0 * Wi + x1], xp[y1 * Wi + x0], xp[y1 * Wi + x1], lx0, lx1, ly0, ly1);
    yp[o] = v;
  }
}
// x2 up-sampling (the decoder's resize of the ASPP output to the c1 resolution, sep_aspp_head.py:96-100).  At scale 1/2 the source index of
// output o is o/2 - 1/4: outputs 4j .. 4j+3 of a row blend the source column pairs (2j-1, 2j), (2j, 2j+1), (2j, 2j+1), (2j+1, 2j+2) with
// weights (.25, .75), (.75, .25), (.25, .75), (.75, .25) -- exactly what bilin_src returns there (0.5 (o + 0.5) - 0.5 and its fraction are
// exact in fp32) --, the first output of a row being the clamped case (weights 1, 0 on columns 0, 1); output rows 2p-1 and 2p blend the SAME
// source rows p-1, p.  One thread: one 8-byte load + two clamped scalars from each of the two source rows, eight outputs, two 16-byte stores;
// row weights from bilin_src, the blend is bilin_blend: bit-identical to the generic kernel, which gathers four values, runs two index
// computations and stores four bytes per output (2.3 TB/s).   grid: ((Hi + 1) / 4 rounded up, C, N), 256 threads = four row pairs; Wi even
__global__ __launch_bounds__(256) void resize_bilinear2x_kernel(const float* __restrict__ x, i64 x_bs, float* __restrict__ y, i64 y_bs, int C,
                                                                int Hi, int Wi) {
  const int c = blockIdx.y, n = blockIdx.z;
  const int Ho = 2 * Hi, Wo = 2 * Wi, W4 = Wo >> 2;
  const float* xp = x + (i64)n * x_bs + (i64)c * Hi * Wi;
  float* yp = y + (i64)n * y_bs + (i64)c * Ho * Wo;
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);            // output rows 2p-1 (p >= 1) and 2p (p <= Hi-1) from source rows p-1, p
  if (p > Hi) return;
  const int ra = max(p - 1, 0), rb = min(p, Hi - 1);
  const float* r0 = xp + (i64)ra * Wi;
  const float* r1 = xp + (i64)rb * Wi;
  const bool hasA = p >= 1, hasB = p <= Hi - 1;
  int y0, y1;
  float la0 = 0.f, la1 = 0.f, lb0 = 0.f, lb1 = 0.f;
  if (hasA) bilin_src(2 * p - 1, 0.5f, Hi, y0, y1, la0, la1);  // (y0, y1) = (ra, rb)
  if (hasB) bilin_src(2 * p, 0.5f, Hi, y0, y1, lb0, lb1);      // (ra, rb) too; p = 0: (0, 1) with weights (1, 0) -- row 1 is not read, its weight is 0
  float4* outA = reinterpret_cast<float4*>(yp + (i64)(2 * p - 1) * Wo);
  float4* outB = reinterpret_cast<float4*>(yp + (i64)(2 * p) * Wo);
  for (int j = threadIdx.x & 63; j < W4; j += 64) {
    const int cl = max(2 * j - 1, 0), cr = min(2 * j + 2, Wi - 1);
    const float2 m0 = *reinterpret_cast<const float2*>(r0 + 2 * j), m1 = *reinterpret_cast<const float2*>(r1 + 2 * j);
    const float l0 = r0[cl], l1 = r1[cl], g0 = r0[cr], g1 = r1[cr];
    auto row = [&](float ly0, float ly1) {
      float4 o;
      o.x = j > 0 ? bilin_blend(l0, m0.x, l1, m1.x, 0.25f, 0.75f, ly0, ly1) : bilin_blend(m0.x, m0.y, m1.x, m1.y, 1.f, 0.f, ly0, ly1);
      o.y = bilin_blend(m0.x, m0.y, m1.x, m1.y, 0.75f, 0.25f, ly0, ly1);
      o.z = bilin_blend(m0.x, m0.y, m1.x, m1.y, 0.25f, 0.75f, ly0, ly1);
      o.w = bilin_blend(m0.y, g0, m1.y, g1, 0.75f, 0.25f, ly0, ly1);
      return o;
    };
    if (hasA) outA[j] = row(la0, la1);
    if (hasB) outB[j] = row(lb0, lb1);
  }
}
// adjoint as a gather over input pixels (deterministic, no atomics)
__global__ void resize_bilinear_bwd_kernel(const float* __restrict__ dy, i64 dy_bs, float* __restrict__ dx, i64 dx_bs, int C, int Hi,
                                           int Wi, int Ho, int Wo, float sh, float sw, int accumulate) {
  const int c = blockIdx.y, n = blockIdx.z;
  const float* gp = dy + (i64)n * dy_bs + (i64)c * Ho * Wo;
  float* dp = dx + (i64)n * dx_bs + (i64)c * Hi * Wi;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < Hi * Wi; i += gridDim.x * blockDim.x) {
    const int iy = i / Wi, ix = i - iy * Wi;
    int oy_lo = (int)floorf(((float)iy - 0.5f) / sh - 0.5f) - 1, oy_hi = (int)ceilf(((float)iy + 1.5f) / sh - 0.5f) + 1;
    int ox_lo = (int)floorf(((float)ix - 0.5f) / sw - 0.5f) - 1, ox_hi = (int)ceilf(((float)ix + 1.5f) / sw - 0.5f) + 1;
    oy_lo = max(oy_lo, 0); oy_hi = min(oy_hi, Ho - 1);
    ox_lo = max(ox_lo, 0); ox_hi = min(ox_hi, Wo - 1);
    float acc = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      int y0, y1; float ly0, ly1;
      bilin_src(oy, sh, Hi, y0, y1, ly0, ly1);
      const float wy = (y0 == iy ? ly0 : 0.f) + (y1 == iy ? ly1 : 0.f);
      if (wy == 0.f) continue;
      float row = 0.f;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        int x0, x1; float lx0, lx1;
        bilin_src(ox, sw, Wi, x0, x1, lx0, lx1);
        const float wx = (x0 == ix ? lx0 : 0.f) + (x1 == ix ? lx1 : 0.f);
        if (wx != 0.f) row = fmaf(wx, gp[oy * Wo + ox], row);
      }
      acc = fmaf(wy, row, acc);
    }
    dp[i] = accumulate ? dp[i] + acc : acc;
  }
}

// exact 2x up-sampling (align_corners=False): output o = 2k reads inputs (k-1: .25, k: .75), o = 2k+1 reads (k: .75, k+1: .25),
// clamped at the borders -- so input i receives {2i-1: .25, 2i: .75, 2i+1: .75, 2i+2: .25}, with the missing border tap's weight
// folded onto the edge output (weight 1).  A separable 4x4 gather, one thread per input pixel.   grid: (blocks over Hi*Wi, C, N)
__global__ __launch_bounds__(256) void resize_bilinear2x_bwd_kernel(const float* __restrict__ dy, i64 dy_bs, float* __restrict__ dx,
                                                                    i64 dx_bs, int C, int Hi, int Wi, int accumulate) {
  const int c = blockIdx.y, n = blockIdx.z;
  const int Wo = 2 * Wi;
  const float* gp = dy + (i64)n * dy_bs + (i64)c * 4 * Hi * Wi;
  float* dp = dx + (i64)n * dx_bs + (i64)c * Hi * Wi;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < Hi * Wi; i += gridDim.x * blockDim.x) {
    const int iy = i / Wi, ix = i - iy * Wi;
    float wy[4], wx[4];
    wy[0] = iy > 0 ? 0.25f : 0.f;  wy[1] = iy > 0 ? 0.75f : 1.f;  wy[2] = iy < Hi - 1 ? 0.75f : 1.f;  wy[3] = iy < Hi - 1 ? 0.25f : 0.f;
    wx[0] = ix > 0 ? 0.25f : 0.f;  wx[1] = ix > 0 ? 0.75f : 1.f;  wx[2] = ix < Wi - 1 ? 0.75f : 1.f;  wx[3] = ix < Wi - 1 ? 0.25f : 0.f;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int oy = min(max(2 * iy - 1 + a, 0), 2 * Hi - 1);          // clamped rows/cols carry weight 0
      const float* row = gp + (i64)oy * Wo + 2 * ix;
      const float2 mid = *reinterpret_cast<const float2*>(row);
      const float lft = row[ix > 0 ? -1 : 0], rgt = row[ix < Wi - 1 ? 2 : 1];
      acc = fmaf(wy[a], wx[0] * lft + wx[1] * mid.x + wx[2] * mid.y + wx[3] * rgt, acc);
    }
    dp[i] = accumulate ? dp[i] + acc : acc;
  }
}

// the same adjoint, a 2 x 2 block of input pixels per thread (Hi, Wi even, 16-byte aligned dy rows): the block's taps lie in 6 rows x 6
// columns of dy -- one 16-byte load + two clamped scalars per row instead of (8-byte load + two scalars) x 4 rows per input pixel -- with the
// taps and weights of resize_bilinear2x_bwd_kernel (a weight-0 tap multiplies whichever finite value sits in its register).
// grid: (blocks over Hi*Wi/4, C, N)
__global__ __launch_bounds__(256) void resize_bilinear2x_bwd_q_kernel(const float* __restrict__ dy, i64 dy_bs, float* __restrict__ dx,
                                                                      i64 dx_bs, int C, int Hi, int Wi, int accumulate) {
  const int c = blockIdx.y, n = blockIdx.z;
  const int Wo = 2 * Wi, Ho = 2 * Hi, W2 = Wi >> 1, H2 = Hi >> 1;
  const float* gp = dy + (i64)n * dy_bs + (i64)c * 4 * Hi * Wi;
  float* dp = dx + (i64)n * dx_bs + (i64)c * Hi * Wi;
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < H2 * W2; b += gridDim.x * blockDim.x) {
    const int by = b / W2, bx = b - by * W2;
    const int iy0 = 2 * by, ix0 = 2 * bx;
    // columns 2 ix0 - 1 .. 2 ix0 + 4 of dy rows 2 iy0 - 1 .. 2 iy0 + 4 (clamped: those taps carry weight 0)
    float v[6][6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const int oy = min(max(2 * iy0 - 1 + r, 0), Ho - 1);
      const float* row = gp + (i64)oy * Wo + 2 * ix0;
      const float4 mid = *reinterpret_cast<const float4*>(row);
      v[r][0] = row[ix0 > 0 ? -1 : 0];
      v[r][1] = mid.x; v[r][2] = mid.y; v[r][3] = mid.z; v[r][4] = mid.w;
      v[r][5] = row[ix0 + 1 < Wi - 1 ? 4 : 3];
    }
    float o[2][2];
#pragma unroll
    for (int dyi = 0; dyi < 2; ++dyi) {
      const int iy = iy0 + dyi;
      float wy[4];
      wy[0] = iy > 0 ? 0.25f : 0.f;  wy[1] = iy > 0 ? 0.75f : 1.f;  wy[2] = iy < Hi - 1 ? 0.75f : 1.f;  wy[3] = iy < Hi - 1 ? 0.25f : 0.f;
#pragma unroll
      for (int dxi = 0; dxi < 2; ++dxi) {
        const int ix = ix0 + dxi;
        float wx[4];
        wx[0] = ix > 0 ? 0.25f : 0.f;  wx[1] = ix > 0 ? 0.75f : 1.f;  wx[2] = ix < Wi - 1 ? 0.75f : 1.f;  wx[3] = ix < Wi - 1 ? 0.25f : 0.f;
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const float* q = &v[2 * dyi + a][2 * dxi];
          acc = fmaf(wy[a], wx[0] * q[0] + wx[1] * q[1] + wx[2] * q[2] + wx[3] * q[3], acc);
        }
        o[dyi][dxi] = acc;
      }
    }
#pragma unroll
    for (int dyi = 0; dyi < 2; ++dyi) {
      float2* d2 = reinterpret_cast<float2*>(dp + (i64)(iy0 + dyi) * Wi + ix0);
      float2 w2 = make_float2(o[dyi][0], o[dyi][1]);
      if (accumulate) { const float2 old = *d2; w2.x = old.x + w2.x; w2.y = old.y + w2.y; }
      *d2 = w2;
    }
  }
}

// ---- per-(n,c) plane reductions / broadcasts.   grid: (C, N)
__global__ void plane_sum_kernel(const float* __restrict__ x, i64 x_bs, float* __restrict__ v, int C, int HW, float scale) {
  __shared__ double sm[16];
  const int c = blockIdx.x, n = blockIdx.y;
  const float* xp = x + (i64)n * x_bs + (i64)c * HW;
  double s = 0.0;
  for (int i = threadIdx.x; i < HW; i += blockDim.x) s += (double)xp[i];
  s = block_sum_d(s, sm);
  if (threadIdx.x == 0) v[n * C + c] = (float)(s * (double)scale);
}
__global__ void broadcast_hw_kernel(const float* __restrict__ v, float* __restrict__ y, i64 y_bs, int C, int HW, float scale, int accumulate,
                                    float* __restrict__ amax) {
  const int c = blockIdx.y, n = blockIdx.z;
  const float val = v[n * C + c] * scale;
  if (amax) amax_publish(amax, accumulate ? 0.f : fabsf(val));      // (plain writes only: the value every element of the plane receives)
  float* yp = y + (i64)n * y_bs + (i64)c * HW;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) yp[i] = accumulate ? yp[i] + val : val;
}
__global__ void channel_scale_kernel(const float* __restrict__ x, const float* __restrict__ mask, float* __restrict__ y, int C, int HW) {
  const int c = blockIdx.y, n = blockIdx.z;
  const float m = mask[n * C + c];
  const i64 base = ((i64)n * C + c) * HW;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) y[base + i] = x[base + i] * m;
}

// ---- test-time inference beyond the whole-image argmax (encoder_decoder.py:220-263 slide_inference, :284-327 inference, :355-372 aug_test):
// class probabilities, the sliding-window sum of crop logits with its count map, flips of probability maps, arg-max of averaged maps
// y[n][c][p] = softmax over c of x[n][:][p] with torch's arithmetic (max, sequential fp32 sum of expf(z - max), IEEE division)
__global__ void softmax_nchw_kernel(const float* __restrict__ x, i64 x_bs, float* __restrict__ y, i64 y_bs, int C, int HW) {
  const int n = blockIdx.y;
  const float* xp = x + (i64)n * x_bs;
  float* yp = y + (i64)n * y_bs;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, xp[(i64)c * HW + p]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(xp[(i64)c * HW + p] - mx);
    for (int c = 0; c < C; ++c) yp[(i64)c * HW + p] = __fdiv_rn(expf(xp[(i64)c * HW + p] - mx), se);
  }
}
// preds[n][c][y1 + i][x1 + j] += crop[n][c][i][j] (F.pad + add, :246-248); count[n][y1 + i][x1 + j] += 1 (:250)       grid: (blocks, C + 1, N)
__global__ void window_accumulate_kernel(const float* __restrict__ crop, float* __restrict__ preds, float* __restrict__ count, int C, int Hc,
                                         int Wc, int H, int W, int y1, int x1) {
  const int c = blockIdx.y, n = blockIdx.z;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < Hc * Wc; i += gridDim.x * blockDim.x) {
    const int iy = i / Wc, ix = i - iy * Wc;
    const i64 dst = (i64)(y1 + iy) * W + (x1 + ix);
    if (c < C) preds[((i64)n * C + c) * H * W + dst] += crop[((i64)n * C + c) * Hc * Wc + i];
    else count[(i64)n * H * W + dst] += 1.f;
  }
}
__global__ void window_normalize_kernel(float* __restrict__ preds, const float* __restrict__ count, int C, int HW) {      // preds / count_mat (:256)
  const int c = blockIdx.y, n = blockIdx.z;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x)
    preds[((i64)n * C + c) * HW + i] = __fdiv_rn(preds[((i64)n * C + c) * HW + i], count[(i64)n * HW + i]);
}
__global__ void argmax_nchw_kernel(const float* __restrict__ x, i64 x_bs, unsigned char* __restrict__ lab, int C, int HW) {  // first maximal class
  const int n = blockIdx.y;
  const float* xp = x + (i64)n * x_bs;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
    float best = xp[p];
    int arg = 0;
    for (int c = 1; c < C; ++c) {
      const float v = xp[(i64)c * HW + p];
      if (v > best) { best = v; arg = c; }
    }
    lab[(i64)n * HW + p] = (unsigned char)arg;
  }
}
__global__ void flip_planes_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int hflip, int vflip) {   // output.flip (:316-325)
  const i64 base = (i64)blockIdx.y * H * W;
  for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < H * W; o += gridDim.x * blockDim.x) {
    const int oy = o / W, ox = o - oy * W;
    y[base + o] = x[base + (i64)(vflip ? H - 1 - oy : oy) * W + (hflip ? W - 1 - ox : ox)];
  }
}
__global__ void div_scalar_kernel(float* __restrict__ x, i64 n, float d) {                                               // seg_logit /= len(imgs) (:367)
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) x[i] = __fdiv_rn(x[i], d);
}

// ---- integer-factor nearest up-sampling of small maps and its adjoint (F.interpolate(mode='nearest'),
// pfgst_loss.py:57-58 when downscale == 1: features at 1/8 are resized to the 1/4 logits grid)
__global__ void upsample_nearest_kernel(const float* __restrict__ x, float* __restrict__ y, int h, int w, int u) {
  const int nc = blockIdx.y, H = h * u, W = w * u;
  for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < H * W; o += gridDim.x * blockDim.x) {
    const int oy = o / W, ox = o - oy * W;
    y[(i64)nc * H * W + o] = x[(i64)nc * h * w + (oy / u) * w + ox / u];
  }
}
__global__ void upsample_nearest_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int h, int w, int u) {
  const int nc = blockIdx.y, W = w * u;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < h * w; i += gridDim.x * blockDim.x) {
    const int iy = i / w, ix = i - iy * w;
    float acc = 0.f;
    for (int a = 0; a < u; ++a)
      for (int b = 0; b < u; ++b) acc += dy[(i64)nc * h * u * W + (i64)(iy * u + a) * W + ix * u + b];
    dx[(i64)nc * h * w + i] = acc;
  }
}

inline int hw_blocks(int HW) {
  int g = cdiv(HW, 256 * 4);
  return g < 1 ? 1 : g;
}

}  // namespace

extern "C" int pfst_fill_f32(float* p, long long n, float value, pfst_stream_t stream) {
  PFST_CHECK_ARG(p && n >= 0);
  if (n == 0) return PFST_OK;
  hipLaunchKernelGGL(fill_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, p, n, value);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}
extern "C" int pfst_axpy_f32(float* y, const float* x, float alpha, long long n, pfst_stream_t stream) {
  PFST_CHECK_ARG(y && x && n >= 0);
  if (n == 0) return PFST_OK;
  hipLaunchKernelGGL(axpy_kernel, dim3(ew_grid(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, y, x, alpha, n);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}
extern "C" int pfst_i64_to_u8(const long long* src, unsigned char* dst, long long n, pfst_stream_t stream) {
  PFST_CHECK_ARG(src && dst && n > 0);
  hipLaunchKernelGGL(i64_to_u8_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, src, dst, n);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}
extern "C" int pfst_u8_to_i64(const unsigned char* src, long long* dst, long long n, pfst_stream_t stream) {
  PFST_CHECK_ARG(src && dst && n > 0);
  hipLaunchKernelGGL(u8_to_i64_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, src, dst, n);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

extern "C" int pfst_maxpool3x3s2(const float* x, float* y, unsigned char* idx, int NC, int H, int W, int Ho, int Wo,
                                 const float* bn_on_load_coef, int C, float* y_amax, pfst_stream_t stream) {
  PFST_CHECK_ARG(!bn_on_load_coef || (C > 0 && NC % C == 0));
  PFST_CHECK_ARG(x && y && idx && NC > 0 && NC <= 65535 * 16 && H > 0 && W > 0);
  PFST_CHECK_ARG(Ho == (H + 2 - 3) / 2 + 1 && Wo == (W + 2 - 3) / 2 + 1 && NC <= 65535);
  if ((W & 3) == 0 && (H & 1) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 7) == 0 &&
      (reinterpret_cast<uintptr_t>(idx) & 1) == 0) {                                   // the stem's pool: two outputs per thread
    hipLaunchKernelGGL(maxpool_pair_kernel, dim3(hw_blocks(Ho * Wo / 2), NC), dim3(256), 0, (hipStream_t)stream, x, y, idx, H, W, Ho, Wo,
                       reinterpret_cast<const float4*>(bn_on_load_coef), C > 0 ? C : 1, y_amax);
    PFST_CHECK_LAUNCH();
    return PFST_OK;
  }
  hipLaunchKernelGGL(maxpool_kernel, dim3(hw_blocks(Ho * Wo), NC), dim3(256), 0, (hipStream_t)stream, x, y, idx, H, W, Ho, Wo,
                     reinterpret_cast<const float4*>(bn_on_load_coef), C > 0 ? C : 1, y_amax);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}
extern "C" int pfst_maxpool3x3s2_bwd(const float* dy, const unsigned char* idx, float* dx, int NC, int H, int W, int Ho, int Wo, pfst_stream_t stream) {
  PFST_CHECK_ARG(dy && idx && dx && NC > 0 && NC <= 65535 && H > 0 && W > 0);
  PFST_CHECK_ARG(Ho == (H + 2 - 3) / 2 + 1 && Wo == (W + 2 - 3) / 2 + 1);
  if ((H & 1) == 0 && (W & 1) == 0 && (reinterpret_cast<uintptr_t>(dx) & 7) == 0) {
    hipLaunchKernelGGL(maxpool_bwd2x2_kernel, dim3(hw_blocks(Ho * Wo), NC), dim3(256), 0, (hipStream_t)stream, dy, idx, dx, H, W);
    PFST_CHECK_LAUNCH();
    return PFST_OK;
  }
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(hw_blocks(H * W), NC), dim3(256), 0, (hipStream_t)stream, dy, idx, dx, H, W, Ho, Wo);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

extern "C" int pfst_resize_bilinear(const float* x, long long x_bs, float* y, long long y_bs, int N, int C, int Hi, int Wi, int Ho, int Wo, pfst_stream_t stream) {
  PFST_CHECK_ARG(x && y && N > 0 && C > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C <= 65535 && N <= 65535);
  if (Ho == 2 * Hi && Wo == 2 * Wi && Hi > 1 && Wi > 1 && (Wi & 1) == 0 && (x_bs & 1) == 0 && (y_bs & 3) == 0 &&
      (reinterpret_cast<uintptr_t>(x) & 7) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0) {                  // decoder up-sampling
    hipLaunchKernelGGL(resize_bilinear2x_kernel, dim3(cdiv(Hi + 1, 4), C, N), dim3(256), 0, (hipStream_t)stream, x, x_bs, y, y_bs, C, Hi, Wi);
    PFST_CHECK_LAUNCH();
    return PFST_OK;
  }
  hipLaunchKernelGGL(resize_bilinear_kernel, dim3(hw_blocks(Ho * Wo), C, N), dim3(256), 0, (hipStream_t)stream, x, x_bs, y, y_bs, C, Hi,
                     Wi, Ho, Wo, (float)Hi / (float)Ho, (float)Wi / (float)Wo);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}
extern "C" int pfst_resize_bilinear_bwd(const float* dy, long long dy_bs, float* dx, long long dx_bs, int N, int C, int Hi, int Wi, int Ho, int Wo, int accumulate, pfst_stream_t stream) {
  PFST_CHECK_ARG(dy && dx && N > 0 && C > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C <= 65535 && N <= 65535);
  if (Ho == 2 * Hi && Wo == 2 * Wi && Hi > 1 && Wi > 1 && (Hi & 1) == 0 && (Wi & 1) == 0 && (dy_bs & 3) == 0 && (dx_bs & 1) == 0 &&
      (reinterpret_cast<uintptr_t>(dy) & 15) == 0 && (reinterpret_cast<uintptr_t>(dx) & 7) == 0) {       // decoder up-sampling, 2 x 2 blocks
    hipLaunchKernelGGL(resize_bilinear2x_bwd_q_kernel, dim3(hw_blocks(Hi * Wi / 4), C, N), dim3(256), 0, (hipStream_t)stream, dy, dy_bs, dx,
                       dx_bs, C, Hi, Wi, accumulate);
    PFST_CHECK_LAUNCH();
    return PFST_OK;
  }
  if (Ho == 2 * Hi && Wo == 2 * Wi && Hi > 1 && Wi > 1 && (dy_bs & 1) == 0 && (reinterpret_cast<uintptr_t>(dy) & 7) == 0) {   // decoder up-sampling
    hipLaunchKernelGGL(resize_bilinear2x_bwd_kernel, dim3(hw_blocks(Hi * Wi), C, N), dim3(256), 0, (hipStream_t)stream, dy, dy_bs, dx,
                       dx_bs, C, Hi, Wi, accumulate);
    PFST_CHECK_LAUNCH();
    return PFST_OK;
  }
  hipLaunchKernelGGL(resize_bilinear_bwd_kernel, dim3(hw_blocks(Hi * Wi), C, N), dim3(256), 0, (hipStream_t)stream, dy, dy_bs, dx, dx_bs,
                     C, Hi, Wi, Ho, Wo, (float)Hi / (float)Ho, (float)Wi / (float)Wo, accumulate);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

extern "C" int pfst_global_avgpool(const float* x, long long x_bs, float* y, int N, int C, int HW, pfst_stream_t stream) {
  PFST_CHECK_ARG(x && y && N > 0 && C > 0 && HW > 0 && N <= 65535);
  hipLaunchKernelGGL(plane_sum_kernel, dim3(C, N), dim3(256), 0, (hipStream_t)stream, x, x_bs, y, C, HW, 1.0f / (float)HW);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}
extern "C" int pfst_reduce_hw(const float* dy, long long dy_bs, float* v, int N, int C, int HW, pfst_stream_t stream) {
  PFST_CHECK_ARG(dy && v && N > 0 && C > 0 && HW > 0 && N <= 65535);
  hipLaunchKernelGGL(plane_sum_kernel, dim3(C, N), dim3(256), 0, (hipStream_t)stream, dy, dy_bs, v, C, HW, 1.0f);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}
extern "C" int pfst_broadcast_hw(const float* v, float* y, long long y_bs, int N, int C, int HW, float scale, int accumulate, float* y_amax,
                                 pfst_stream_t stream) {
  PFST_CHECK_ARG(v && y && N > 0 && C > 0 && HW > 0 && C <= 65535 && N <= 65535 && (!y_amax || !accumulate));
  hipLaunchKernelGGL(broadcast_hw_kernel, dim3(hw_blocks(HW), C, N), dim3(256), 0, (hipStream_t)stream, v, y, y_bs, C, HW, scale, accumulate, y_amax);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}
extern "C" int pfst_channel_scale(const float* x, const float* mask, float* y, int N, int C, int HW, pfst_stream_t stream) {
  PFST_CHECK_ARG(x && mask && y && N > 0 && C > 0 && HW > 0 && C <= 65535 && N <= 65535);
  hipLaunchKernelGGL(channel_scale_kernel, dim3(hw_blocks(HW), C, N), dim3(256), 0, (hipStream_t)stream, x, mask, y, C, HW);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

extern "C" int pfst_softmax_nchw(const float* x, long long x_bs, float* y, long long y_bs, int N, int C, int HW, pfst_stream_t stream) {
  PFST_CHECK_ARG(x && y && N > 0 && N <= 65535 && C > 0 && HW > 0 && x_bs >= (i64)C * HW && y_bs >= (i64)C * HW);
  hipLaunchKernelGGL(softmax_nchw_kernel, dim3(hw_blocks(HW), N), dim3(256), 0, (hipStream_t)stream, x, (i64)x_bs, y, (i64)y_bs, C, HW);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}
extern "C" int pfst_window_accumulate(const float* crop, float* preds, float* count, int N, int C, int Hc, int Wc, int H, int W, int y1, int x1,
                                      pfst_stream_t stream) {
  PFST_CHECK_ARG(crop && preds && count && N > 0 && N <= 65535 && C > 0 && C < 65535 && Hc > 0 && Wc > 0);
  PFST_CHECK_ARG(y1 >= 0 && x1 >= 0 && y1 + Hc <= H && x1 + Wc <= W);          // the window lies inside the image: nothing is written outside preds
  hipLaunchKernelGGL(window_accumulate_kernel, dim3(hw_blocks(Hc * Wc), C + 1, N), dim3(256), 0, (hipStream_t)stream, crop, preds, count, C, Hc,
                     Wc, H, W, y1, x1);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}
extern "C" int pfst_window_normalize(float* preds, const float* count, int N, int C, int HW, pfst_stream_t stream) {
  PFST_CHECK_ARG(preds && count && N > 0 && N <= 65535 && C > 0 && C <= 65535 && HW > 0);
  hipLaunchKernelGGL(window_normalize_kernel, dim3(hw_blocks(HW), C, N), dim3(256), 0, (hipStream_t)stream, preds, count, C, HW);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}
extern "C" int pfst_argmax_nchw(const float* x, long long x_bs, unsigned char* label_u8, int N, int C, int HW, pfst_stream_t stream) {
  PFST_CHECK_ARG(x && label_u8 && N > 0 && N <= 65535 && C > 0 && C <= 255 && HW > 0 && x_bs >= (i64)C * HW);
  hipLaunchKernelGGL(argmax_nchw_kernel, dim3(hw_blocks(HW), N), dim3(256), 0, (hipStream_t)stream, x, (i64)x_bs, label_u8, C, HW);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}
extern "C" int pfst_flip_planes(const float* x, float* y, int planes, int H, int W, int horizontal, int vertical, pfst_stream_t stream) {
  PFST_CHECK_ARG(x && y && x != y && planes > 0 && planes <= 65535 && H > 0 && W > 0);
  hipLaunchKernelGGL(flip_planes_kernel, dim3(hw_blocks(H * W), planes), dim3(256), 0, (hipStream_t)stream, x, y, H, W, horizontal, vertical);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}
extern "C" int pfst_div_scalar(float* x, long long n, float divisor, pfst_stream_t stream) {
  PFST_CHECK_ARG(x && n > 0 && divisor != 0.f);
  hipLaunchKernelGGL(div_scalar_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, (i64)n, divisor);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

extern "C" int pfst_upsample_nearest(const float* x, float* y, int NC, int h, int w, int factor, pfst_stream_t stream) {
  PFST_CHECK_ARG(x && y && NC > 0 && NC <= 65535 && h > 0 && w > 0 && factor >= 1);
  hipLaunchKernelGGL(upsample_nearest_kernel, dim3(hw_blocks(h * w * factor * factor), NC), dim3(256), 0, (hipStream_t)stream, x, y, h, w, factor);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}
extern "C" int pfst_upsample_nearest_bwd(const float* dy, float* dx, int NC, int h, int w, int factor, pfst_stream_t stream) {
  PFST_CHECK_ARG(dy && dx && NC > 0 && NC <= 65535 && h > 0 && w > 0 && factor >= 1);
  hipLaunchKernelGGL(upsample_nearest_bwd_kernel, dim3(hw_blocks(h * w), NC), dim3(256), 0, (hipStream_t)stream, dy, dx, h, w, factor);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}
