// Fused bilinear-upsample + softmax cross-entropy + accuracy, pseudo-labelling and class-mix.
// The full-resolution logits (b x C x S x S; 201 MB per head per pass at b=8, S=1024, C=6) are never
// materialised: every full-resolution pixel interpolates its C logits from the low-resolution map
// (which stays L2 resident) in registers.
// Reference: rsiseg/models/decode_heads/decode_head.py:249-283 (resize + CE + accuracy),
// losses/cross_entropy_loss.py:45-65, losses/utils.py:60-69, losses/accuracy.py:6-61,
// uda/pfgst.py:259-300 (pseudo labels, class mix), utils/dacs_transforms.py:110-144.
#include <stdlib.h>
#include <algorithm>
#include "common.h"
#include "../../include/pfst_hip.h"

namespace {

struct Bilin {
  int y0, y1, x0, x1;
  float ly0, ly1, lx0, lx1;
};
__device__ __forceinline__ Bilin make_bilin(int oy, int ox, float sh, float sw, int h, int w) {
  Bilin b;
  bilin_src(oy, sh, h, b.y0, b.y1, b.ly0, b.ly1);
  bilin_src(ox, sw, w, b.x0, b.x1, b.lx0, b.lx1);
  return b;
}
__device__ __forceinline__ float interp(const float* __restrict__ p, int w, const Bilin& b) {
  return bilin_blend(p[b.y0 * w + b.x0], p[b.y0 * w + b.x1], p[b.y1 * w + b.x0], p[b.y1 * w + b.x1], b.lx0, b.lx1, b.ly0, b.ly1);
}

// grid: (blocks over H*W, N).  One thread per full-resolution pixel.
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits, int C, int h, int w,
                                                     const unsigned char* __restrict__ label, const float* __restrict__ pw,
                                                     const float* __restrict__ cw, int H, int W, int ignore, float sh, float sw,
                                                     float* __restrict__ lse, double* __restrict__ acc) {
  __shared__ double sm[16];
  const int n = blockIdx.y;
  const float* lp = logits + (i64)n * C * h * w;
  const int hw = h * w;
  double loss = 0.0, correct = 0.0, valid = 0.0, bad = 0.0;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < H * W; p += gridDim.x * blockDim.x) {
    const int oy = p / W, ox = p - oy * W;
    const Bilin b = make_bilin(oy, ox, sh, sw, h, w);
    const int lab = label[(i64)n * H * W + p];
    float mx = -INFINITY, zl = 0.f;
    int arg = 0;
    float se = 0.f;
    if (C <= 8) {
      // up to eight classes (the ISPRS / INRIA configs: 6 / 2): the interpolated logits stay in registers between the max and the sum pass
      // (same values, same order of operations: bit-identical to the two-pass form below)
      float zc[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        zc[c] = c < C ? interp(lp + (i64)c * hw, w, b) : -INFINITY;
        if (c < C && zc[c] > mx) { mx = zc[c]; arg = c; }
        if (c == lab) zl = zc[c];
      }
#pragma unroll
      for (int c = 0; c < 8; ++c)
        if (c < C) se += expf(zc[c] - mx);
    } else {
      for (int c = 0; c < C; ++c) {
        const float z = interp(lp + (i64)c * hw, w, b);
        if (z > mx) { mx = z; arg = c; }
        if (c == lab) zl = z;
      }
      for (int c = 0; c < C; ++c) se += expf(interp(lp + (i64)c * hw, w, b) - mx);
    }
    const float l = mx + logf(se);
    lse[(i64)n * H * W + p] = l;
    if (lab != ignore && lab < C) {
      float wgt = pw ? pw[(i64)n * H * W + p] : 1.f;
      if (cw) wgt *= cw[lab];
      loss += (double)(wgt * (l - zl));
      valid += 1.0;
      if (arg == lab) correct += 1.0;
    } else if (lab != ignore) {
      bad += 1.0;       // a label in [C, 255] other than ignore_index: F.cross_entropy raises on it, so must the caller (acc[3])
    }
  }
  bad = block_sum_d(bad, sm);
  if (threadIdx.x == 0 && bad > 0.0) atomicAdd(&acc[3], bad);
  loss = block_sum_d(loss, sm);
  correct = block_sum_d(correct, sm);
  valid = block_sum_d(valid, sm);
  if (threadIdx.x == 0) {
    atomicAdd(&acc[0], loss);
    atomicAdd(&acc[1], correct);
    atomicAdd(&acc[2], valid);
  }
}

// The x4 / x8 cases (H == S h, W == S w, S = 4: the decode head's logits at 1/4 resolution, 8: the auxiliary head's at 1/8; C <= 8) by
// inter-cell blocks, as ce_bwd_blocks_kernel below: one thread per S x S block of full-resolution
// pixels that interpolate from the same four cells -- their 4 x C logits are loaded once instead of once per pixel.  Per pixel the arithmetic
// is the one above (same coordinates and weights from bilin_src, same class order): bit-identical lse, the same loss terms.
// Two tiles of 16 x 16 blocks per workgroup: every workgroup ends in four block sums and three fp64 atomics on one cache line; measured at
// 8 x 1024^2: one tile 122 us, two 97, four 109 (too few workgroups), 1024-thread workgroups of four tiles 113 (profiles/r05_small_kernels.txt).
// grid: (ceil((w + 1) / 16), ceil((h + 1) / 32), N), 16 x 16 threads
template <int S, int TILES>
__global__ __launch_bounds__(256) void ce_fwd_blocks_kernel(const float* __restrict__ logits, int C, int h, int w,
                                                        const unsigned char* __restrict__ label, const float* __restrict__ pw,
                                                        const float* __restrict__ cw, int ignore, float* __restrict__ lse,
                                                        double* __restrict__ acc) {
  __shared__ double sm[16];
  const int n = blockIdx.z, H = S * h, W = S * w, hw = h * w;
  const int bx = blockIdx.x * 16 + (threadIdx.x & 15) - 1;
  const float* lp = logits + (i64)n * C * hw;
  const unsigned char* lab = label + (i64)n * H * W;
  float* ls = lse + (i64)n * H * W;
  const float* pwp = pw ? pw + (i64)n * H * W : nullptr;
  double loss = 0.0, correct = 0.0, valid = 0.0, bad = 0.0;
  for (int it = 0; it < TILES; ++it) {
    const int by = (blockIdx.y * TILES + it) * 16 + (threadIdx.x >> 4) - 1;
    if (bx > w - 1 || by > h - 1) continue;
    const int xl = max(bx, 0), xr = min(xl + 1, w - 1), yt = max(by, 0), yb = min(yt + 1, h - 1);
    float v[2][2][8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float* q = lp + (i64)(c < C ? c : 0) * hw;
      v[0][0][c] = q[yt * w + xl];
      v[0][1][c] = q[yt * w + xr];
      v[1][0][c] = q[yb * w + xl];
      v[1][1][c] = q[yb * w + xr];
    }
    float lx0[S], lx1[S];
#pragma unroll
    for (int j = 0; j < S; ++j) {
      const int ox = S * bx + S / 2 + j;
      int x0, x1;
      lx0[j] = lx1[j] = 0.f;
      if (ox >= 0 && ox < W) bilin_src(ox, 1.f / S, w, x0, x1, lx0[j], lx1[j]);
    }
#pragma unroll
    for (int r = 0; r < S; ++r) {
      const int oy = S * by + S / 2 + r;
      if (oy < 0 || oy >= H) continue;
      int y0, y1;
      float ly0, ly1;
      bilin_src(oy, 1.f / S, h, y0, y1, ly0, ly1);
#pragma unroll
      for (int half = 0; half < S / 2; ++half) {
        const int ox = S * bx + S / 2 + 2 * half;
        if (ox < 0 || ox >= W) continue;
        const i64 p = (i64)oy * W + ox;
        const unsigned short l2 = *reinterpret_cast<const unsigned short*>(lab + p);
        float2 g2 = make_float2(1.f, 1.f);
        if (pwp) g2 = *reinterpret_cast<const float2*>(pwp + p);
        float lo[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int j = 2 * half + k;
          const int l = k ? (l2 >> 8) : (l2 & 255);
          float mx = -INFINITY, zl = 0.f, se = 0.f;
          int arg = 0;
          float zc[8];
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            zc[c] = c < C ? bilin_blend(v[0][0][c], v[0][1][c], v[1][0][c], v[1][1][c], lx0[j], lx1[j], ly0, ly1) : -INFINITY;
            if (c < C && zc[c] > mx) { mx = zc[c]; arg = c; }
            if (c == l) zl = zc[c];
          }
#pragma unroll
          for (int c = 0; c < 8; ++c)
            if (c < C) se += expf(zc[c] - mx);
          const float e = mx + logf(se);
          lo[k] = e;
          if (l != ignore && l < C) {
            float wgt = k ? g2.y : g2.x;
            if (!pwp) wgt = 1.f;
            if (cw) wgt *= cw[l];
            loss += (double)(wgt * (e - zl));
            valid += 1.0;
            if (arg == l) correct += 1.0;
          } else if (l != ignore) {
            bad += 1.0;
          }
        }
        *reinterpret_cast<float2*>(ls + p) = make_float2(lo[0], lo[1]);
      }
    }
  }
  bad = block_sum_d(bad, sm);
  if (threadIdx.x == 0 && bad > 0.0) atomicAdd(&acc[3], bad);
  loss = block_sum_d(loss, sm);
  correct = block_sum_d(correct, sm);
  valid = block_sum_d(valid, sm);
  if (threadIdx.x == 0) {
    atomicAdd(&acc[0], loss);
    atomicAdd(&acc[1], correct);
    atomicAdd(&acc[2], valid);
  }
}

// grid: (blocks over h*w, C, N).  One thread per low-resolution logit; gathers its full-res footprint.
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, int C, int h, int w,
                                                     const unsigned char* __restrict__ label, const float* __restrict__ pw,
                                                     const float* __restrict__ cw, int H, int W, int ignore, float sh, float sw,
                                                     const float* __restrict__ lse, float scale, float* __restrict__ dlogits,
                                                     int accumulate) {
  const int c = blockIdx.y, n = blockIdx.z;
  const float* lp = logits + ((i64)n * C + c) * h * w;
  const unsigned char* lab = label + (i64)n * H * W;
  const float* ls = lse + (i64)n * H * W;
  const float* pwp = pw ? pw + (i64)n * H * W : nullptr;
  float* dp = dlogits + ((i64)n * C + c) * h * w;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < h * w; i += gridDim.x * blockDim.x) {
    const int iy = i / w, ix = i - iy * w;
    int oy_lo = (int)floorf(((float)iy - 0.5f) / sh - 0.5f) - 1, oy_hi = (int)ceilf(((float)iy + 1.5f) / sh - 0.5f) + 1;
    int ox_lo = (int)floorf(((float)ix - 0.5f) / sw - 0.5f) - 1, ox_hi = (int)ceilf(((float)ix + 1.5f) / sw - 0.5f) + 1;
    oy_lo = max(oy_lo, 0); oy_hi = min(oy_hi, H - 1);
    ox_lo = max(ox_lo, 0); ox_hi = min(ox_hi, W - 1);
    float acc = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      int y0, y1; float ly0, ly1;
      bilin_src(oy, sh, h, y0, y1, ly0, ly1);
      const float wy = (y0 == iy ? ly0 : 0.f) + (y1 == iy ? ly1 : 0.f);
      if (wy == 0.f) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        Bilin b;
        b.y0 = y0; b.y1 = y1; b.ly0 = ly0; b.ly1 = ly1;
        bilin_src(ox, sw, w, b.x0, b.x1, b.lx0, b.lx1);
        const float wx = (b.x0 == ix ? b.lx0 : 0.f) + (b.x1 == ix ? b.lx1 : 0.f);
        if (wx == 0.f) continue;
        const int p = oy * W + ox;
        const int l = lab[p];
        if (l == ignore || l >= C) continue;
        float g = pwp ? pwp[p] : 1.f;
        if (cw) g *= cw[l];
        const float prob = expf(interp(lp, w, b) - ls[p]);
        acc = fmaf(wy * wx * g, prob - (l == c ? 1.f : 0.f), acc);
      }
    }
    acc *= scale;
    dp[i] = accumulate ? dp[i] + acc : acc;
  }
}

// The same gather with ONE thread per low-resolution cell for all classes (C <= 8): the per-pixel work that does not depend on the class --
// source coordinates and weights, label, log-sum-exp, pixel weight -- is done once instead of C times, the C accumulators live in
// registers.  Per class the terms and their order are those of ce_bwd_kernel: bit-identical gradients.       grid: (blocks over h*w, 1, N)
__global__ __launch_bounds__(256) void ce_bwd_cells_kernel(const float* __restrict__ logits, int C, int h, int w,
                                                           const unsigned char* __restrict__ label, const float* __restrict__ pw,
                                                           const float* __restrict__ cw, int H, int W, int ignore, float sh, float sw,
                                                           const float* __restrict__ lse, float scale, float* __restrict__ dlogits,
                                                           int accumulate) {
  const int n = blockIdx.z;
  const int hw = h * w;
  const float* lp = logits + (i64)n * C * hw;
  const unsigned char* lab = label + (i64)n * H * W;
  const float* ls = lse + (i64)n * H * W;
  const float* pwp = pw ? pw + (i64)n * H * W : nullptr;
  float* dp = dlogits + (i64)n * C * hw;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += gridDim.x * blockDim.x) {
    const int iy = i / w, ix = i - iy * w;
    int oy_lo = (int)floorf(((float)iy - 0.5f) / sh - 0.5f) - 1, oy_hi = (int)ceilf(((float)iy + 1.5f) / sh - 0.5f) + 1;
    int ox_lo = (int)floorf(((float)ix - 0.5f) / sw - 0.5f) - 1, ox_hi = (int)ceilf(((float)ix + 1.5f) / sw - 0.5f) + 1;
    oy_lo = max(oy_lo, 0); oy_hi = min(oy_hi, H - 1);
    ox_lo = max(ox_lo, 0); ox_hi = min(ox_hi, W - 1);
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      int y0, y1; float ly0, ly1;
      bilin_src(oy, sh, h, y0, y1, ly0, ly1);
      const float wy = (y0 == iy ? ly0 : 0.f) + (y1 == iy ? ly1 : 0.f);
      if (wy == 0.f) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        Bilin b;
        b.y0 = y0; b.y1 = y1; b.ly0 = ly0; b.ly1 = ly1;
        bilin_src(ox, sw, w, b.x0, b.x1, b.lx0, b.lx1);
        const float wx = (b.x0 == ix ? b.lx0 : 0.f) + (b.x1 == ix ? b.lx1 : 0.f);
        if (wx == 0.f) continue;
        const int p = oy * W + ox;
        const int l = lab[p];
        if (l == ignore || l >= C) continue;
        float g = pwp ? pwp[p] : 1.f;
        if (cw) g *= cw[l];
        const float wg = wy * wx * g, lsp = ls[p];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          if (c < C) {
            const float prob = expf(interp(lp + (i64)c * hw, w, b) - lsp);
            acc[c] = fmaf(wg, prob - (l == c ? 1.f : 0.f), acc[c]);
          }
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if (c < C) {
        const float a = acc[c] * scale;
        dp[(i64)c * hw + i] = accumulate ? dp[(i64)c * hw + i] + a : a;
      }
    }
  }
}

// The x4 / x8 cases of the same gradient (H == S h, W == S w: the decode head's logits at 1/4, the auxiliary head's at 1/8 resolution; C <= 8)
// by INTER-CELL BLOCKS instead of cells.  The S x S full-resolution pixels between four neighbouring cell centres -- pixels S b + S/2 ...
// S b + 3S/2 - 1 of block b in each direction (4b+2 ... 4b+5 at S = 4),
// b = -1 ... w-1 (the outermost blocks are the half-width borders whose taps are clamped) -- all interpolate from the same four cells:
// one thread loads those 4 x C logits ONCE, evaluates each pixel's softmax once (the per-cell gather above evaluates every pixel in each
// of the up to four cells it touches: 4x the exponentials, 16x the logit loads), and keeps the 4 x C corner sums in registers.  A
// workgroup is 16 x 16 blocks; the corner sums meet in LDS in four conflict-free phases (in a phase every thread adds to a different
// cell), and the 15 x 15 cells whose four blocks all lie inside the workgroup are written out -- neighbouring workgroups overlap by one
// block row / column.  Per pixel and tap the term is the one of ce_bwd_cells_kernel (same source coordinates, weights, interpolation and
// fma); the ORDER of a cell's <= 64 terms differs (block by block instead of raster order), deterministically.
// grid: (ceil(w / 15), ceil(h / 15), N)
template <int S>
__global__ __launch_bounds__(256) void ce_bwd_blocks_kernel(const float* __restrict__ logits, int C, int h, int w,
                                                        const unsigned char* __restrict__ label, const float* __restrict__ pw,
                                                        const float* __restrict__ cw, int ignore, const float* __restrict__ lse, float scale,
                                                        float* __restrict__ dlogits, int accumulate) {
  constexpr int TB = 16, OWN = TB - 1;
  __shared__ float cell[8][TB + 1][TB + 1];
  const int n = blockIdx.z, H = S * h, W = S * w, hw = h * w;
  const int tx = threadIdx.x & (TB - 1), ty = threadIdx.x >> 4;
  const int cx0 = blockIdx.x * OWN, cy0 = blockIdx.y * OWN;        // first cell this workgroup owns
  const int bx = cx0 - 1 + tx, by = cy0 - 1 + ty;                  // this thread's block: taps (b, b + 1), clamped at the borders
  const float* lp = logits + (i64)n * C * hw;
  const unsigned char* lab = label + (i64)n * H * W;
  const float* ls = lse + (i64)n * H * W;
  const float* pwp = pw ? pw + (i64)n * H * W : nullptr;
  for (int i = threadIdx.x; i < 8 * (TB + 1) * (TB + 1); i += 256) (&cell[0][0][0])[i] = 0.f;
  float acc[2][2][8];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[a][b][c] = 0.f;
  if (bx <= w - 1 && by <= h - 1) {
    // the four cells every pixel of the block interpolates from (what bilin_src returns for each of them)
    const int xl = max(bx, 0), xr = min(xl + 1, w - 1), yt = max(by, 0), yb = min(yt + 1, h - 1);
    float v[2][2][8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float* q = lp + (i64)(c < C ? c : 0) * hw;
      v[0][0][c] = q[yt * w + xl];
      v[0][1][c] = q[yt * w + xr];
      v[1][0][c] = q[yb * w + xl];
      v[1][1][c] = q[yb * w + xr];
    }
    // column taps of the block's four pixel columns: weight towards cell bx (L) and cell bx + 1 (R)
    float lx0[S], lx1[S], wxl[S], wxr[S];
#pragma unroll
    for (int j = 0; j < S; ++j) {
      const int ox = S * bx + S / 2 + j;
      int x0 = 0, x1 = 0;
      lx0[j] = lx1[j] = 0.f;
      if (ox >= 0 && ox < W) bilin_src(ox, 1.f / S, w, x0, x1, lx0[j], lx1[j]);
      wxl[j] = (x0 == bx ? lx0[j] : 0.f) + (x1 == bx ? lx1[j] : 0.f);
      wxr[j] = (x0 == bx + 1 ? lx0[j] : 0.f) + (x1 == bx + 1 ? lx1[j] : 0.f);
    }
#pragma unroll
    for (int r = 0; r < S; ++r) {
      const int oy = S * by + S / 2 + r;
      if (oy < 0 || oy >= H) continue;
      int y0, y1;
      float ly0, ly1;
      bilin_src(oy, 1.f / S, h, y0, y1, ly0, ly1);
      const float wyt = (y0 == by ? ly0 : 0.f) + (y1 == by ? ly1 : 0.f);
      const float wyb = (y0 == by + 1 ? ly0 : 0.f) + (y1 == by + 1 ? ly1 : 0.f);
#pragma unroll
      for (int half = 0; half < S / 2; ++half) {
        const int ox = S * bx + S / 2 + 2 * half;                  // pixel pairs are inside the image together (W = 4w, ox even)
        if (ox < 0 || ox >= W) continue;
        const i64 p = (i64)oy * W + ox;
        const unsigned short l2 = *reinterpret_cast<const unsigned short*>(lab + p);
        const float2 ls2 = *reinterpret_cast<const float2*>(ls + p);
        float2 g2 = make_float2(1.f, 1.f);
        if (pwp) g2 = *reinterpret_cast<const float2*>(pwp + p);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int j = 2 * half + k;
          const int l = k ? (l2 >> 8) : (l2 & 255);
          if (l == ignore || l >= C) continue;
          float g = k ? g2.y : g2.x;
          if (cw) g *= cw[l];
          const float lsp = k ? ls2.y : ls2.x;
          const float wtl = wyt * wxl[j] * g, wtr = wyt * wxr[j] * g, wbl = wyb * wxl[j] * g, wbr = wyb * wxr[j] * g;
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            if (c < C) {
              const float z = bilin_blend(v[0][0][c], v[0][1][c], v[1][0][c], v[1][1][c], lx0[j], lx1[j], ly0, ly1);
              const float d = expf(z - lsp) - (l == c ? 1.f : 0.f);
              acc[0][0][c] = fmaf(wtl, d, acc[0][0][c]);
              acc[0][1][c] = fmaf(wtr, d, acc[0][1][c]);
              acc[1][0][c] = fmaf(wbl, d, acc[1][0][c]);
              acc[1][1][c] = fmaf(wbr, d, acc[1][1][c]);
            }
          }
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 2; ++a) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
#pragma unroll
      for (int c = 0; c < 8; ++c)
        if (c < C) cell[c][ty + a][tx + b] += acc[a][b][c];
      __syncthreads();
    }
  }
  float* dp = dlogits + (i64)n * C * hw;
  for (int i = threadIdx.x; i < OWN * OWN; i += 256) {
    const int u = i / OWN, q = i - u * OWN;
    const int gy = cy0 + u, gx = cx0 + q;
    if (gy >= h || gx >= w) continue;
    for (int c = 0; c < C; ++c) {
      const float a = cell[c][u + 1][q + 1] * scale;
      float* o = dp + (i64)c * hw + gy * w + gx;
      *o = accumulate ? *o + a : a;
    }
  }
}

// grid: (blocks over H*W, N).  The reference's rule (pfgst.py:259-261): p = softmax(z) = exp(z - max) / sum (torch's softmax
// arithmetic: sequential fp32 sum over the classes, IEEE division), then torch.max over p -- the FIRST class whose ROUNDED
// probability equals the maximum wins.  That differs from the arg-max of z whenever two distinct logits give probabilities
// that round to the same float (flat early-training logits), so the comparison is made on p, not on z.
__global__ __launch_bounds__(256) void pseudo_label_kernel(const float* __restrict__ logits, int C, int h, int w, int H, int W,
                                                           float sh, float sw, float thr, long long* __restrict__ l64,
                                                           unsigned char* __restrict__ l8, unsigned long long* __restrict__ count,
                                                           float* __restrict__ conf, float* __restrict__ prob) {
  __shared__ double sm[16];
  const int n = blockIdx.y;
  const float* lp = logits + (i64)n * C * h * w;
  const int hw = h * w;
  double cnt = 0.0;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < H * W; p += gridDim.x * blockDim.x) {
    const int oy = p / W, ox = p - oy * W;
    const Bilin b = make_bilin(oy, ox, sh, sw, h, w);
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, interp(lp + (i64)c * hw, w, b));
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(interp(lp + (i64)c * hw, w, b) - mx);
    float pmax = -1.f;
    int arg = 0;
    for (int c = 0; c < C; ++c) {
      const float pc = __fdiv_rn(expf(interp(lp + (i64)c * hw, w, b) - mx), se);
      if (pc > pmax) { pmax = pc; arg = c; }   // strict '>' keeps the FIRST maximal probability like torch.max
    }
    if (pmax >= thr) cnt += 1.0;
    if (conf) conf[(i64)n * H * W + p] = pmax >= thr ? 1.f : 0.f;     // thre_type='part' (pfgst.py:267-268)
    if (prob) prob[(i64)n * H * W + p] = pmax;
    if (l64) l64[(i64)n * H * W + p] = arg;
    if (l8) l8[(i64)n * H * W + p] = (unsigned char)arg;
  }
  cnt = block_sum_d(cnt, sm);
  if (threadIdx.x == 0 && cnt > 0.0) atomicAdd(count, (unsigned long long)cnt);
}

__global__ void label_presence_kernel(const unsigned char* __restrict__ label, i64 n, int* __restrict__ presence) {
  __shared__ int flags[256];
  flags[threadIdx.x] = 0;
  __syncthreads();
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) flags[label[i]] = 1;
  __syncthreads();
  if (flags[threadIdx.x]) presence[threadIdx.x] = 1;
}

// grid: (blocks over HW, N)
__global__ void class_mask_kernel(const unsigned char* __restrict__ gt, const int* __restrict__ classes, int K,
                                  unsigned char* __restrict__ mask, i64 HW) {
  __shared__ unsigned char lut[256];
  const int n = blockIdx.y;
  lut[threadIdx.x] = 0;
  __syncthreads();
  if (threadIdx.x < K) {
    const int c = classes[n * K + threadIdx.x];
    if (c >= 0 && c < 256) lut[c] = 1;
  }
  __syncthreads();
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (i64)gridDim.x * blockDim.x)
    mask[(i64)n * HW + i] = lut[gt[(i64)n * HW + i]];
}

// grid: (blocks over HW, N)
__global__ void class_mix_kernel(const float* __restrict__ img, const float* __restrict__ trg, const unsigned char* __restrict__ gt,
                                 const unsigned char* __restrict__ pseudo, const unsigned char* __restrict__ mask,
                                 const unsigned long long* __restrict__ conf, const float* __restrict__ trg_w, float* __restrict__ mimg,
                                 unsigned char* __restrict__ mlbl, long long* __restrict__ mlbl64, float* __restrict__ mw,
                                 int Cimg, i64 HW, double numel) {
  const int n = blockIdx.y;
  // q exactly as the reference: python float (double) count/size, stored into a float32 tensor
  const float q = trg_w ? 0.f : (float)((double)conf[0] / numel);
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (i64)gridDim.x * blockDim.x) {
    const i64 p = (i64)n * HW + i;
    const int m = mask[p];
    const float fm = (float)m, fi = (float)(1 - m);
    for (int c = 0; c < Cimg; ++c) {
      const i64 o = ((i64)n * Cimg + c) * HW + i;
      mimg[o] = fm * img[o] + fi * trg[o];
    }
    const int l = m ? gt[p] : pseudo[p];
    mlbl[p] = (unsigned char)l;
    if (mlbl64) mlbl64[p] = l;
    mw[p] = fm * 1.0f + fi * (trg_w ? trg_w[p] : q);
  }
}

// confusion statistics of rsiseg/core/evaluation/metrics.py:26-86 (intersect_and_union):
// hist[c] = #(pred==label==c), hist[C+c] = #(pred==c), hist[2C+c] = #(label==c), over pixels with label != ignore
__global__ __launch_bounds__(256) void confusion_hist_kernel(const unsigned char* __restrict__ pred, const unsigned char* __restrict__ label,
                                                             i64 n, int C, int ignore, unsigned long long* __restrict__ hist) {
  extern __shared__ unsigned int sh[];   // [3*C]
  for (int i = threadIdx.x; i < 3 * C; i += blockDim.x) sh[i] = 0;
  __syncthreads();
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
    const int l = label[i];
    if (l == ignore) continue;
    const int p = pred[i];
    if (p < C) atomicAdd(&sh[C + p], 1u);
    if (l < C) atomicAdd(&sh[2 * C + l], 1u);
    if (p == l && p < C) atomicAdd(&sh[p], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * C; i += blockDim.x)
    if (sh[i]) atomicAdd(&hist[i], (unsigned long long)sh[i]);
}

__global__ void ce_finalize_kernel(const double* __restrict__ acc, double numel, float loss_weight, float* __restrict__ out) {
  const double eps = 1.1920928955078125e-07;  // torch.finfo(float32).eps
  out[0] = (float)((double)loss_weight * (acc[0] / numel));
  out[1] = (float)((acc[1] + eps) * (100.0 / (acc[2] + eps)));
  out[2] = (float)acc[3];
}

inline int px_blocks(i64 n) {
  i64 g = (n + 1023) / 1024;
  if (g > 4096) g = 4096;
  return g < 1 ? 1 : (int)g;
}

}  // namespace

extern "C" int pfst_ce_upsample_fwd(const float* logits, int N, int C, int h, int w, const unsigned char* label, const float* pix_weight,
                                    const float* class_weight, int H, int W, int ignore_index, float* lse, double* acc, pfst_stream_t stream) {
  PFST_CHECK_ARG(logits && label && lse && acc && N > 0 && C > 0 && C <= 255 && h > 0 && w > 0 && H > 0 && W > 0 && N <= 65535);
  // 16 pixels per thread: every workgroup ends in three same-address fp64 atomics, and 8192 workgroups of four pixels per thread spent
  // half of the launch in that tail (profiles/r05_small_kernels.txt: 322 -> 149 us at 8 x 1024^2)
  const int gx = (int)std::max<i64>(1, std::min<i64>(4096, ((i64)H * W + 4095) / 4096));
  const bool blocks_ok = C <= 8 && (((uintptr_t)lse | (uintptr_t)pix_weight) & 7) == 0 && ((uintptr_t)label & 1) == 0 && h < 65535 * 16;
  if (blocks_ok && H == 4 * h && W == 4 * w)
    hipLaunchKernelGGL((ce_fwd_blocks_kernel<4, 2>), dim3(cdiv(w + 1, 16), cdiv(h + 1, 32), N), dim3(256), 0, (hipStream_t)stream, logits, C, h, w,
                       label, pix_weight, class_weight, ignore_index, lse, acc);
  else if (blocks_ok && H == 8 * h && W == 8 * w)        // the auxiliary head's logits at 1/8 resolution: 64 pixels per thread, one tile per workgroup
    hipLaunchKernelGGL((ce_fwd_blocks_kernel<8, 1>), dim3(cdiv(w + 1, 16), cdiv(h + 1, 16), N), dim3(256), 0, (hipStream_t)stream, logits, C, h, w,
                       label, pix_weight, class_weight, ignore_index, lse, acc);
  else
    hipLaunchKernelGGL(ce_fwd_kernel, dim3(gx, N), dim3(256), 0, (hipStream_t)stream, logits, C, h, w, label,
                       pix_weight, class_weight, H, W, ignore_index, (float)h / (float)H, (float)w / (float)W, lse, acc);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

extern "C" int pfst_ce_upsample_bwd(const float* logits, int N, int C, int h, int w, const unsigned char* label, const float* pix_weight,
                                    const float* class_weight, int H, int W, int ignore_index, const float* lse, float scale,
                                    float* dlogits, int accumulate, pfst_stream_t stream) {
  PFST_CHECK_ARG(logits && label && lse && dlogits && N > 0 && C > 0 && C <= 255 && h > 0 && w > 0 && H > 0 && W > 0 && N <= 65535);
  int gx = cdiv((i64)h * w, 256);
  const bool blocks_ok = C <= 8 && (((uintptr_t)lse | (uintptr_t)pix_weight) & 7) == 0 && ((uintptr_t)label & 1) == 0 && h < 65535 * 15;
  if (blocks_ok && H == 4 * h && W == 4 * w)
    hipLaunchKernelGGL(ce_bwd_blocks_kernel<4>, dim3(cdiv(w, 15), cdiv(h, 15), N), dim3(256), 0, (hipStream_t)stream, logits, C, h, w, label,
                       pix_weight, class_weight, ignore_index, lse, scale, dlogits, accumulate);
  else if (blocks_ok && H == 8 * h && W == 8 * w)
    hipLaunchKernelGGL(ce_bwd_blocks_kernel<8>, dim3(cdiv(w, 15), cdiv(h, 15), N), dim3(256), 0, (hipStream_t)stream, logits, C, h, w, label,
                       pix_weight, class_weight, ignore_index, lse, scale, dlogits, accumulate);
  else if (C <= 8)     // one thread per low-resolution cell, all classes in registers; more classes: one launch row per class
    hipLaunchKernelGGL(ce_bwd_cells_kernel, dim3(gx, 1, N), dim3(256), 0, (hipStream_t)stream, logits, C, h, w, label, pix_weight,
                       class_weight, H, W, ignore_index, (float)h / (float)H, (float)w / (float)W, lse, scale, dlogits, accumulate);
  else
    hipLaunchKernelGGL(ce_bwd_kernel, dim3(gx, C, N), dim3(256), 0, (hipStream_t)stream, logits, C, h, w, label, pix_weight,
                       class_weight, H, W, ignore_index, (float)h / (float)H, (float)w / (float)W, lse, scale, dlogits, accumulate);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

extern "C" int pfst_pseudo_label(const float* logits, int N, int C, int h, int w, int H, int W, float threshold,
                                 long long* label_i64, unsigned char* label_u8, unsigned long long* count, float* conf_mask,
                                 float* max_prob, pfst_stream_t stream) {
  PFST_CHECK_ARG(logits && (label_i64 || label_u8) && count && N > 0 && C > 0 && C <= 255 && h > 0 && w > 0 && H > 0 && W > 0 && N <= 65535);
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(count, 0, sizeof(unsigned long long), s) != hipSuccess) return PFST_ERR_LAUNCH;
  hipLaunchKernelGGL(pseudo_label_kernel, dim3(px_blocks((i64)H * W), N), dim3(256), 0, s, logits, C, h, w, H, W,
                     (float)h / (float)H, (float)w / (float)W, threshold, label_i64, label_u8, count, conf_mask, max_prob);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

extern "C" int pfst_label_presence(const unsigned char* label, long long n, int* presence256, pfst_stream_t stream) {
  PFST_CHECK_ARG(label && presence256 && n > 0);
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(presence256, 0, sizeof(int) * 256, s) != hipSuccess) return PFST_ERR_LAUNCH;
  hipLaunchKernelGGL(label_presence_kernel, dim3(px_blocks(n)), dim3(256), 0, s, label, (i64)n, presence256);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

extern "C" int pfst_class_mask(const unsigned char* gt, const int* classes, int K, unsigned char* mask, int N, long long HW, pfst_stream_t stream) {
  PFST_CHECK_ARG(gt && classes && mask && K > 0 && K <= 256 && N > 0 && N <= 65535 && HW > 0);
  hipLaunchKernelGGL(class_mask_kernel, dim3(px_blocks(HW), N), dim3(256), 0, (hipStream_t)stream, gt, classes, K, mask, (i64)HW);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

extern "C" int pfst_class_mix(const float* img, const float* trg_img, const unsigned char* gt, const unsigned char* pseudo,
                              const unsigned char* mask, const unsigned long long* conf_count, const float* trg_weight, float* mixed_img,
                              unsigned char* mixed_lbl, long long* mixed_lbl_i64, float* mixed_w, int N, int Cimg, long long HW, pfst_stream_t stream) {
  PFST_CHECK_ARG(img && trg_img && gt && pseudo && mask && (conf_count || trg_weight) && mixed_img && mixed_lbl && mixed_w);
  PFST_CHECK_ARG(N > 0 && N <= 65535 && Cimg > 0 && HW > 0);
  hipLaunchKernelGGL(class_mix_kernel, dim3(px_blocks(HW), N), dim3(256), 0, (hipStream_t)stream, img, trg_img, gt, pseudo, mask,
                     conf_count, trg_weight, mixed_img, mixed_lbl, mixed_lbl_i64, mixed_w, Cimg, (i64)HW, (double)N * (double)HW);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

extern "C" int pfst_ce_finalize(const double* acc, double numel, float loss_weight, float* out, pfst_stream_t stream) {
  PFST_CHECK_ARG(acc && out && numel > 0);
  hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, acc, numel, loss_weight, out);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

extern "C" int pfst_confusion_hist(const unsigned char* pred, const unsigned char* label, long long n, int C, int ignore_index,
                                   unsigned long long* hist, pfst_stream_t stream) {
  PFST_CHECK_ARG(pred && label && hist && n > 0 && C > 0 && C <= 255);
  hipLaunchKernelGGL(confusion_hist_kernel, dim3(px_blocks(n)), dim3(256), 3 * C * sizeof(unsigned int), (hipStream_t)stream, pred,
                     label, (i64)n, C, ignore_index, hist);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}
