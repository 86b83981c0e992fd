// PFGSTLoss: local pseudo-feature similarity losses, fused so that the reference's unfold tensors
// (b x 512 x 9 x H x W, 302 MB per image per call at S=1024) never exist.
// Reference: rsiseg/models/losses/pfgst_loss.py:44-234: kernel 3, dilation d, cross_prob_type 'trg'; the shipped options
// (sim_type 'cosine', detach_unfold=True, src_loss_type 'mean_std', top_k, downscale 0.5) and the variants reachable from
// the same configs: sim_type 'gaussian' (:199-201), src_loss_type 'margin' / 'margin2' (:116-131), detach_unfold=False
// (:151-152), top_k=None (:229-231), downscale None / 1.
// Neighbour index k = ty*3+tx, offset ((ty-1)*d, (tx-1)*d) -- the order nn.Unfold produces.
#include "common.h"
#include <stdlib.h>
#include "../../include/pfst_hip.h"

namespace {

constexpr float COS_EPS = 1e-8f;

__device__ __forceinline__ int nearest_src(int dst, float scale, int in_size) {
  // ATen nearest_neighbor_compute_source_index: min(floor(dst*scale), in-1)
  int s = (int)floorf((float)dst * scale);
  return s < in_size - 1 ? s : in_size - 1;
}

// ---- cosine similarity to the 9 dilated neighbours.  One thread per pixel, loop over channels;
// loads are coalesced along x and the 9 taps of a channel hit L1/L2 (feature map read once from HBM).
// grid: (blocks over H*W, N)
// GAUSS: sim_k = exp(-|F(r+D_k) - F(r)|^2 / sigma^2), the zero padding of nn.Unfold taking part as F = 0 (:199-201).
template <bool GAUSS>
__global__ __launch_bounds__(256) void sim_map_kernel(const float* __restrict__ feat, int C, int H, int W, int dil, float inv_sigma2,
                                                      float* __restrict__ sim, float* __restrict__ norm) {
  const int n = blockIdx.y;
  const int HW = H * W;
  const float* fp = feat + (i64)n * C * HW;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
    const int y = p / W, x = p - y * W;
    int off[9];
    bool ok[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int sy = y + (k / 3 - 1) * dil, sx = x + (k % 3 - 1) * dil;
      ok[k] = sy >= 0 && sy < H && sx >= 0 && sx < W;
      off[k] = ok[k] ? sy * W + sx : p;
    }
    float dot[9], nn[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) { dot[k] = 0.f; nn[k] = 0.f; }
    for (int c = 0; c < C; ++c) {
      const float* ch = fp + (i64)c * HW;
      const float a = ch[p];
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const float b = ok[k] ? ch[off[k]] : 0.f;
        if (GAUSS) {
          const float d = b - a;
          dot[k] = fmaf(d, d, dot[k]);
        } else {
          dot[k] = fmaf(a, b, dot[k]);
          nn[k] = fmaf(b, b, nn[k]);
        }
      }
    }
    if (GAUSS) {
#pragma unroll
      for (int k = 0; k < 9; ++k) sim[((i64)n * 9 + k) * HW + p] = expf(-dot[k] * inv_sigma2);
      if (norm) norm[(i64)n * HW + p] = 0.f;
      continue;
    }
    const float na = fmaxf(sqrtf(nn[4]), COS_EPS);
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const float nb = fmaxf(sqrtf(nn[k]), COS_EPS);
      sim[((i64)n * 9 + k) * HW + p] = dot[k] / (na * nb);
    }
    norm[(i64)n * HW + p] = sqrtf(nn[4]);
  }
}

// ---- adjoint: dF(r) = sum_k A_k(r) F(r+D_k) + B(r) F(r)   (PFGSTLoss backward)
//   A_k = (G[k,r] + G[8-k, r+D_k]) / (n(r) n(r+D_k)),  B = -(sum_k G[k,r] s_k(r) + G[8-k,r+D_k] s_{8-k}(r+D_k)) / n(r)^2
// GAUSS: d s_k(r) / dF(r) = -2/sigma^2 s_k(r) (F(r) - F(r+D_k)) and symmetrically for the neighbour, so with
//   c_k = -2/sigma^2 (G[k,r] s_k(r) + G[8-k,r+D_k] s_{8-k}(r+D_k)):  A_k = -c_k,  B = sum_k c_k  (+ the padded taps, whose
//   neighbour is the constant 0: c = -2/sigma^2 G[k,r] s_k(r) goes to B only).  Same final loop.
template <bool GAUSS>
__global__ __launch_bounds__(256) void sim_map_bwd_kernel(const float* __restrict__ feat, const float* __restrict__ sim,
                                                          const float* __restrict__ norm, const float* __restrict__ gsim, int C,
                                                          int H, int W, int dil, float inv_sigma2, float* __restrict__ dfeat, int accumulate) {
  const int n = blockIdx.y;
  const int HW = H * W;
  const float* fp = feat + (i64)n * C * HW;
  float* dp = dfeat + (i64)n * C * HW;
  const float* sp = sim + (i64)n * 9 * HW;
  const float* gp = gsim + (i64)n * 9 * HW;
  const float* np_ = norm + (i64)n * HW;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
    const int y = p / W, x = p - y * W;
    const float nr = GAUSS ? 1.f : fmaxf(np_[p], COS_EPS);
    float A[9];
    int off[9];
    float B = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int sy = y + (k / 3 - 1) * dil, sx = x + (k % 3 - 1) * dil;
      const bool in = sy >= 0 && sy < H && sx >= 0 && sx < W;
      const bool ok = in && k != 4;
      off[k] = ok ? sy * W + sx : p;
      A[k] = 0.f;
      if (GAUSS) {
        if (ok) {
          const int q = off[k];
          const float c = -2.f * inv_sigma2 * (gp[(i64)k * HW + p] * sp[(i64)k * HW + p] + gp[(i64)(8 - k) * HW + q] * sp[(i64)(8 - k) * HW + q]);
          A[k] = -c;
          B += c;
        } else if (!in) {
          B += -2.f * inv_sigma2 * gp[(i64)k * HW + p] * sp[(i64)k * HW + p];
        }
      } else if (ok) {
        const int q = off[k];
        const float g1 = gp[(i64)k * HW + p], g2 = gp[(i64)(8 - k) * HW + q];
        const float nq = fmaxf(np_[q], COS_EPS);
        A[k] = (g1 + g2) / (nr * nq);
        B -= g1 * sp[(i64)k * HW + p] + g2 * sp[(i64)(8 - k) * HW + q];
      }
    }
    if (!GAUSS) B /= nr * nr;
    for (int c = 0; c < C; ++c) {
      const float* ch = fp + (i64)c * HW;
      float v = B * ch[p];
#pragma unroll
      for (int k = 0; k < 9; ++k) v = fmaf(A[k], ch[off[k]], v);
      const i64 o = (i64)c * HW + p;
      dp[o] = accumulate ? dp[o] + v : v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Fast path of the cosine similarity (the shipped configuration): a wave owns a strip of 64 float4 = 256 consecutive pixels
// (whole rows: W/4 in {16, 32, 64}) and a slice of the channels.  Per channel a lane issues THREE aligned 16-byte loads (rows
// y-d, y, y+d); the +-d column shifts of the nine taps come from the neighbouring lanes' registers (two or four wave shuffles
// per row), the centre row stays in registers, the zero padding of nn.Unfold is a mask at the row ends / missing rows.  The 36
// dot products and the squared norms of the three rows accumulate per lane; the four waves of a block (channel quarters) are
// combined through LDS, and the neighbours' norms of the cosine denominator are the SAME sums shifted by the tap -- no
// per-tap norm accumulation (the first kernel spent 9 of its 18 fma per channel on them).
// ---------------------------------------------------------------------------------------------------------------------------
// Strip owned by workgroup bx of gx: a strip reads the rows y-d..y+d, i.e. the centre rows of its neighbour strips, so neighbouring
// strips must share an L2.  Workgroups are dealt round-robin over the 8 XCDs (bx % 8, as gx % 8 == 0 keeps the rows of the grid
// aligned), hence XCD x gets the CONTIGUOUS strips x*gx/8 .. (x+1)*gx/8 - 1: only the rows at the 16 chunk borders are fetched twice
// (with the identity mapping every row crossed the fabric three times: the first strip kernel ran at 6.5 TB/s of fabric reads).
__device__ __forceinline__ int strip_of(int bx, int gx) {
  return (gx & 7) == 0 ? (bx & 7) * (gx >> 3) + (bx >> 3) : bx;
}

template <int D>
__device__ __forceinline__ void shift_pair(const float4& v, bool has_prev, bool has_next, float4& left, float4& right) {
  // left = elements (x0 - D .. x0 - D + 3), right = (x0 + D .. x0 + D + 3) of the row whose aligned float4 at x0 this lane holds
  static_assert(D == 1 || D == 2, "feature-grid dilation 1 or 2");
  if (D == 2) {
    const float pz = __shfl_up(v.z, 1, 64), pw = __shfl_up(v.w, 1, 64), nx = __shfl_down(v.x, 1, 64), ny = __shfl_down(v.y, 1, 64);
    left = make_float4(has_prev ? pz : 0.f, has_prev ? pw : 0.f, v.x, v.y);
    right = make_float4(v.z, v.w, has_next ? nx : 0.f, has_next ? ny : 0.f);
  } else {
    const float pw = __shfl_up(v.w, 1, 64), nx = __shfl_down(v.x, 1, 64);
    left = make_float4(has_prev ? pw : 0.f, v.x, v.y, v.z);
    right = make_float4(v.y, v.z, v.w, has_next ? nx : 0.f);
  }
}
__device__ __forceinline__ void fma4(float4& acc, const float4& a, const float4& b) {
  acc.x = fmaf(a.x, b.x, acc.x); acc.y = fmaf(a.y, b.y, acc.y); acc.z = fmaf(a.z, b.z, acc.z); acc.w = fmaf(a.w, b.w, acc.w);
}
__device__ __forceinline__ float4 ld4_or_zero(const float* __restrict__ p, bool ok) {
  return ok ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
}

constexpr int SIMQ_VALS = 48;       // per lane: 9 taps x 4 pixels of dot products + 3 rows x 4 pixels of squared norms

// grid: (H*W / 256, N), 64 * NW threads (NW = 4 or 8 channel slices per strip)
template <int D, int NW>
__global__ __launch_bounds__(64 * NW) void sim_map_cos_q_kernel(const float* __restrict__ feat, int C, int H, int W, float* __restrict__ sim,
                                                            float* __restrict__ norm) {
  extern __shared__ float red[];                      // [4 waves][SIMQ_VALS][64 lanes]
  const int n = blockIdx.y, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int HW = H * W, wq = W >> 2;
  const int pix4 = strip_of(blockIdx.x, gridDim.x) * 64 + lane;
  const int y = pix4 / wq, xq = pix4 - y * wq;
  const bool has_prev = xq > 0, has_next = xq < wq - 1;
  const bool up = y - D >= 0, dn = y + D < H;
  const int cpw = C / NW;                             // channels of this wave (host: C % NW == 0)
  const float* fp = feat + ((i64)n * C + (i64)wid * cpw) * HW + (i64)y * W + 4 * xq;
  float4 dot[9], nn[3];
#pragma unroll
  for (int k = 0; k < 9; ++k) dot[k] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int t = 0; t < 3; ++t) nn[t] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
  for (int c = 0; c < cpw; ++c, fp += HW) {
    float4 r[3];
    r[0] = ld4_or_zero(fp - D * W, up);
    r[1] = *reinterpret_cast<const float4*>(fp);
    r[2] = ld4_or_zero(fp + D * W, dn);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      float4 l, rr;
      shift_pair<D>(r[t], has_prev, has_next, l, rr);
      fma4(dot[3 * t + 0], r[1], l);
      fma4(dot[3 * t + 1], r[1], r[t]);
      fma4(dot[3 * t + 2], r[1], rr);
      fma4(nn[t], r[t], r[t]);
    }
  }
  // combine the channel slices: (NW = 8: waves 4..7 hand their sums to waves 0..3 first, so the LDS block stays 4 slices)
  auto put = [&](int slot) {
    float* mine = red + (slot * SIMQ_VALS) * 64 + lane;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      mine[(4 * k + 0) * 64] = dot[k].x; mine[(4 * k + 1) * 64] = dot[k].y; mine[(4 * k + 2) * 64] = dot[k].z; mine[(4 * k + 3) * 64] = dot[k].w;
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      mine[(36 + 4 * t + 0) * 64] = nn[t].x; mine[(36 + 4 * t + 1) * 64] = nn[t].y; mine[(36 + 4 * t + 2) * 64] = nn[t].z; mine[(36 + 4 * t + 3) * 64] = nn[t].w;
    }
  };
  if (NW == 8) {
    if (wid >= 4) put(wid - 4);
    __syncthreads();
    if (wid < 4) {
      const float* o = red + (wid * SIMQ_VALS) * 64 + lane;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        dot[k].x += o[(4 * k + 0) * 64]; dot[k].y += o[(4 * k + 1) * 64]; dot[k].z += o[(4 * k + 2) * 64]; dot[k].w += o[(4 * k + 3) * 64];
      }
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        nn[t].x += o[(36 + 4 * t + 0) * 64]; nn[t].y += o[(36 + 4 * t + 1) * 64]; nn[t].z += o[(36 + 4 * t + 2) * 64]; nn[t].w += o[(36 + 4 * t + 3) * 64];
      }
    }
    __syncthreads();
  }
  if (wid < 4) put(wid);
  __syncthreads();
  if (wid != 0) return;
  auto total = [&](int v) {
    return (red[(0 * SIMQ_VALS + v) * 64 + lane] + red[(1 * SIMQ_VALS + v) * 64 + lane]) +
           (red[(2 * SIMQ_VALS + v) * 64 + lane] + red[(3 * SIMQ_VALS + v) * 64 + lane]);
  };
#pragma unroll
  for (int k = 0; k < 9; ++k) dot[k] = make_float4(total(4 * k), total(4 * k + 1), total(4 * k + 2), total(4 * k + 3));
#pragma unroll
  for (int t = 0; t < 3; ++t) nn[t] = make_float4(total(36 + 4 * t), total(37 + 4 * t), total(38 + 4 * t), total(39 + 4 * t));
  auto clampn = [](const float4& q) {
    return make_float4(fmaxf(sqrtf(q.x), COS_EPS), fmaxf(sqrtf(q.y), COS_EPS), fmaxf(sqrtf(q.z), COS_EPS), fmaxf(sqrtf(q.w), COS_EPS));
  };
  const float4 na = clampn(nn[1]);
  const i64 o = (i64)n * 9 * HW + 4 * (i64)pix4;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    float4 l, rr;
    shift_pair<D>(nn[t], has_prev, has_next, l, rr);         // squared norms of the tap pixels (0 in the padding -> COS_EPS)
    const float4 nb[3] = {clampn(l), clampn(nn[t]), clampn(rr)};
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const float4 d = dot[3 * t + u];
      const float4 q = make_float4(d.x / (na.x * nb[u].x), d.y / (na.y * nb[u].y), d.z / (na.z * nb[u].z), d.w / (na.w * nb[u].w));
      *reinterpret_cast<float4*>(sim + o + (i64)(3 * t + u) * HW) = q;
    }
  }
  *reinterpret_cast<float4*>(norm + (i64)n * HW + 4 * (i64)pix4) = make_float4(sqrtf(nn[1].x), sqrtf(nn[1].y), sqrtf(nn[1].z), sqrtf(nn[1].w));
}

// adjoint, same data movement: dF(r) = B(r) F(r) + sum_k A_k(r) F(r + D_k).  The per-pixel coefficients (sim_map_bwd_kernel's
// prologue) are computed once by sim_bwd_coef_kernel into coef[n][10][HW] = (A_0..A_8 with A_4 = 0, B); the stencil kernel then
// reads them as ten aligned float4 and streams the channels: three 16-byte loads + one 16-byte store each.
// grid: (blocks over H*W, N)
__global__ __launch_bounds__(256) void sim_bwd_coef_kernel(const float* __restrict__ sim, const float* __restrict__ norm,
                                                           const float* __restrict__ gsim, int H, int W, int dil, float* __restrict__ coef) {
  const int n = blockIdx.y;
  const int HW = H * W;
  const float* sp = sim + (i64)n * 9 * HW;
  const float* gp = gsim + (i64)n * 9 * HW;
  const float* np_ = norm + (i64)n * HW;
  float* cp = coef + (i64)n * 10 * HW;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
    const int y = p / W, x = p - y * W;
    const float nr = fmaxf(np_[p], COS_EPS);
    float b = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int sy = y + (k / 3 - 1) * dil, sx = x + (k % 3 - 1) * dil;
      const bool ok = sy >= 0 && sy < H && sx >= 0 && sx < W && k != 4;
      float a = 0.f;
      if (ok) {
        const int q = sy * W + sx;
        const float g1 = gp[(i64)k * HW + p], g2 = gp[(i64)(8 - k) * HW + q];
        a = (g1 + g2) / (nr * fmaxf(np_[q], COS_EPS));
        b -= g1 * sp[(i64)k * HW + p] + g2 * sp[(i64)(8 - k) * HW + q];
      }
      cp[(i64)k * HW + p] = a;
    }
    cp[(i64)9 * HW + p] = b / (nr * nr);
  }
}

// grid: (H*W / 256, channel chunks, N), 256 threads; wave w of chunk j handles channels [(4 j + w) cpw, +cpw)
template <int D>
__global__ __launch_bounds__(256) void sim_map_bwd_cos_q_kernel(const float* __restrict__ feat, const float* __restrict__ coef, int C, int H,
                                                                int W, int cpw, float* __restrict__ dfeat, int accumulate) {
  const int n = blockIdx.z, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int HW = H * W, wq = W >> 2;
  const int pix4 = strip_of(blockIdx.x, gridDim.x) * 64 + lane;
  const int y = pix4 / wq, xq = pix4 - y * wq;
  const bool has_prev = xq > 0, has_next = xq < wq - 1;
  const bool up = y - D >= 0, dn = y + D < H;
  const float* cp = coef + (i64)n * 10 * HW + 4 * (i64)pix4;
  float4 A[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) A[k] = *reinterpret_cast<const float4*>(cp + (i64)k * HW);
  const float4 B = *reinterpret_cast<const float4*>(cp + (i64)9 * HW);
  const int c0 = (blockIdx.y * 4 + wid) * cpw;
  const i64 base = ((i64)n * C + c0) * HW + (i64)y * W + 4 * xq;
  const float* fp = feat + base;
  float* dp = dfeat + base;
#pragma unroll 4
  for (int c = 0; c < cpw && c0 + c < C; ++c, fp += HW, dp += HW) {
    float4 r[3];
    r[0] = ld4_or_zero(fp - D * W, up);
    r[1] = *reinterpret_cast<const float4*>(fp);
    r[2] = ld4_or_zero(fp + D * W, dn);
    float4 v = make_float4(B.x * r[1].x, B.y * r[1].y, B.z * r[1].z, B.w * r[1].w);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      float4 l, rr;
      shift_pair<D>(r[t], has_prev, has_next, l, rr);
      fma4(v, A[3 * t + 0], l);
      fma4(v, A[3 * t + 1], r[t]);
      fma4(v, A[3 * t + 2], rr);
    }
    if (accumulate) {
      const float4 o = *reinterpret_cast<const float4*>(dp);
      v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    *reinterpret_cast<float4*>(dp) = v;
  }
}

// the fast path applies to whole-row strips of 256 pixels and feature-grid dilation 1 or 2 (every shipped configuration)
inline bool simq_ok(const void* a, const void* b, int C, int H, int W, int dil) {
  const int wq = W / 4;
  return (dil == 1 || dil == 2) && W % 4 == 0 && (wq == 16 || wq == 32 || wq == 64) && ((i64)H * W) % 256 == 0 && C % 4 == 0 &&
         ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0;
}

// ---- source statistics.  grid: (blocks over H*W, N)
__device__ __forceinline__ int src_pair_class(const unsigned char* __restrict__ gt, int Hg, int Wg, float sgy, float sgx, int H, int W,
                                               int y, int x, int k, int dil, int ctr) {
  // returns 0 = skip, 1 = positive pair, 2 = negative pair
  const int sy = y + (k / 3 - 1) * dil, sx = x + (k % 3 - 1) * dil;
  int nb = 0;  // nn.Unfold zero padding: label 0 outside
  if (sy >= 0 && sy < H && sx >= 0 && sx < W) nb = gt[(i64)nearest_src(sy, sgy, Hg) * Wg + nearest_src(sx, sgx, Wg)];
  return nb == ctr ? 1 : 2;
}

// ---- src_perc (pfgst_loss.py:98-102): only the hardest fraction of the source pairs enters the source losses -- the smallest
// positive and the largest negative similarities: `sorted[:int(n * src_perc)]`.  A sorted prefix is {s < t} plus some of the ties
// {s == t}, t the K-th smallest value, so no sort is needed: an exact radix select (4 passes of 8 bits over order-preserving
// integer keys) finds t and the number of ties inside the prefix; ties share that number equally (they are equal values, so the
// losses are the reference's; only the split of the gradient among EXACTLY equal similarities is even instead of by sort position).
// Negative pairs use the complemented key (descending order).
struct SrcSel {                       // one per set (0: positive pairs, 1: negative pairs)
  unsigned int prefix;                // key bits fixed so far; after the last pass: key of the K-th element
  unsigned int shift;                 // bit position of the digit the next histogram pass resolves (24, 16, 8, 0)
  unsigned long long rank;            // 1-based rank of the wanted element among the keys matching `prefix`
  unsigned long long K;               // int(n * src_perc)
  float w_tie;                        // weight of the elements whose key == prefix (ties inside the prefix / all ties)
  int done;
};
__device__ __forceinline__ unsigned int order_key(float v, bool descending) {
  const unsigned int u = __float_as_uint(v);
  const unsigned int k = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return descending ? ~k : k;
}
__device__ __forceinline__ float sel_weight(const SrcSel* __restrict__ sel, int set, float v) {
  if (!sel) return 1.f;
  const unsigned int k = order_key(v, set == 1);
  return k < sel[set].prefix ? 1.f : (k == sel[set].prefix ? sel[set].w_tie : 0.f);
}

// grid: (blocks over H*W, N): histogram of the current digit of the keys matching the prefix
__global__ __launch_bounds__(256) void src_sel_hist_kernel(const float* __restrict__ sim, const unsigned char* __restrict__ gt, int H, int W,
                                                           int Hg, int Wg, int dil, const SrcSel* __restrict__ sel, unsigned int* __restrict__ hist) {
  __shared__ unsigned int sh[2 * 256];
  sh[threadIdx.x] = 0; sh[256 + threadIdx.x] = 0;
  __syncthreads();
  const int n = blockIdx.y, HW = H * W;
  const unsigned char* g = gt + (i64)n * Hg * Wg;
  const float sgy = (float)Hg / (float)H, sgx = (float)Wg / (float)W;
  const unsigned int shift0 = sel[0].shift, shift1 = sel[1].shift;
  // keys match when their bits ABOVE the current digit equal the prefix (first pass: shift = 24, nothing fixed yet)
  const unsigned int hi0 = shift0 >= 24 ? 0u : (0xFFFFFFFFu << (shift0 + 8)), hi1 = shift1 >= 24 ? 0u : (0xFFFFFFFFu << (shift1 + 8));
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
    const int y = p / W, x = p - y * W;
    const int ctr = g[(i64)nearest_src(y, sgy, Hg) * Wg + nearest_src(x, sgx, Wg)];
    if (ctr == 255) continue;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int cls = src_pair_class(g, Hg, Wg, sgy, sgx, H, W, y, x, k, dil, ctr);
      const float v = sim[((i64)n * 9 + k) * HW + p];
      if (cls == 1) {
        const unsigned int key = order_key(v, false);
        if (((key ^ sel[0].prefix) & hi0) == 0) atomicAdd(&sh[(key >> shift0) & 255u], 1u);
      } else {
        const unsigned int key = order_key(v, true);
        if (((key ^ sel[1].prefix) & hi1) == 0) atomicAdd(&sh[256 + ((key >> shift1) & 255u)], 1u);
      }
    }
  }
  __syncthreads();
  if (sh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], sh[threadIdx.x]);
  if (sh[256 + threadIdx.x]) atomicAdd(&hist[256 + threadIdx.x], sh[256 + threadIdx.x]);
}

// one block, 2 threads used: resolve the digit, advance the state, clear the histogram
// first: 2 = initialise the state (before the first histogram), 1 = first digit (also fixes K from the set size), 0 = later digits
__global__ void src_sel_scan_kernel(SrcSel* __restrict__ sel, unsigned int* __restrict__ hist, double perc, int first) {
  const int set = threadIdx.x;
  if (first == 2) {
    if (set < 2) {
      sel[set].prefix = 0; sel[set].shift = 24; sel[set].rank = 0; sel[set].K = 0; sel[set].w_tie = 0.f; sel[set].done = 0;
    }
    for (int i = threadIdx.x; i < 512; i += blockDim.x) hist[i] = 0;
    return;
  }
  if (set < 2) {
    SrcSel& st = sel[set];
    unsigned int* h = hist + 256 * set;
    if (first) {
      unsigned long long n = 0;
      for (int d = 0; d < 256; ++d) n += h[d];
      st.K = (unsigned long long)((double)n * perc);         // int(n * src_perc): truncation of the double product
      st.rank = st.K;
      st.prefix = 0;
      st.done = st.K == 0;
      st.w_tie = 0.f;                                          // K == 0: nothing selected (no key is below prefix 0)
    }
    if (!st.done) {
      unsigned long long cum = 0;
      int d = 0;
      for (; d < 256; ++d) {
        if (cum + h[d] >= st.rank) break;
        cum += h[d];
      }
      st.prefix |= (unsigned int)d << st.shift;
      st.rank -= cum;
      if (st.shift == 0) {
        st.w_tie = (float)((double)st.rank / (double)h[d]);   // `rank` of the h[d] ties lie inside the sorted prefix
        st.done = 1;
      }
    }
    st.shift = st.shift >= 8 ? st.shift - 8 : 0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += blockDim.x) hist[i] = 0;
}

// loss_type 0: count / sum / sum of squares (mean & unbiased std, :107-115); 1 / 2: count / sum of relu(m0 - s)^e for positive
// pairs and relu(s - m1)^e for negative pairs, e = loss_type (:116-131).
__global__ __launch_bounds__(256) void src_stats_kernel(const float* __restrict__ sim, const unsigned char* __restrict__ gt, int H, int W,
                                                        int Hg, int Wg, int dil, int loss_type, float m0, float m1,
                                                        double* __restrict__ stats, const SrcSel* __restrict__ sel,
                                                        double* __restrict__ det_part = nullptr) {
  // det_part != NULL (deterministic mode, api.cpp): this block's six sums go to slot blockIdx.y * gridDim.x + blockIdx.x of det_part[slots][6]
  // instead of being added atomically into stats; src_stats_det_sum_kernel adds the slots in index order
  __shared__ double sm[16];
  const int n = blockIdx.y, HW = H * W;
  const unsigned char* g = gt + (i64)n * Hg * Wg;
  const float sgy = (float)Hg / (float)H, sgx = (float)Wg / (float)W;
  double a[6] = {0, 0, 0, 0, 0, 0};
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
    const int y = p / W, x = p - y * W;
    const int ctr = g[(i64)nearest_src(y, sgy, Hg) * Wg + nearest_src(x, sgx, Wg)];
    if (ctr == 255) continue;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int cls = src_pair_class(g, Hg, Wg, sgy, sgx, H, W, y, x, k, dil, ctr);
      const float sv = sim[((i64)n * 9 + k) * HW + p];
      const double s = (double)sv;
      const double wt = (double)sel_weight(sel, cls == 1 ? 0 : 1, sv);       // src_perc: 1 inside the kept prefix, 0 outside
      const int o = cls == 1 ? 0 : 3;
      if (loss_type == 0) {
        a[o] += wt; a[o + 1] += wt * s; a[o + 2] += wt * s * s;
      } else {
        const double h = cls == 1 ? fmax((double)m0 - s, 0.0) : fmax(s - (double)m1, 0.0);
        a[o] += wt; a[o + 1] += wt * (loss_type == 1 ? h : h * h);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const double r = block_sum_d(a[i], sm);
    if (threadIdx.x == 0) {
      if (det_part) det_part[((i64)blockIdx.y * gridDim.x + blockIdx.x) * 6 + i] = r;
      else if (r != 0.0) atomicAdd(&stats[i], r);
    }
  }
}
__global__ void src_stats_det_sum_kernel(const double* __restrict__ part, int T, double* __restrict__ stats) {
  const int i = threadIdx.x;
  if (i >= 6) return;
  double a = 0.0;
  for (int k = 0; k < T; ++k) a += part[(i64)k * 6 + i];
  stats[i] = a;
}

struct SrcMoments { double n, mean, std; };
__device__ __forceinline__ SrcMoments moments(const double* st) {
  SrcMoments m;
  m.n = st[0];
  m.mean = st[0] > 0 ? st[1] / st[0] : 0.0;
  const double var = st[0] > 1 ? (st[2] - st[1] * st[1] / st[0]) / (st[0] - 1.0) : 0.0;
  m.std = var > 0 ? sqrt(var) : 0.0;
  return m;
}

__global__ __launch_bounds__(256) void src_grad_kernel(const float* __restrict__ sim, const unsigned char* __restrict__ gt, int H, int W,
                                                       int Hg, int Wg, int dil, int loss_type, float m0, float m1,
                                                       const double* __restrict__ stats, float w_pos,
                                                       float w_neg, float w_pos_std, float w_neg_std, float* __restrict__ gsim,
                                                       float* __restrict__ losses, const SrcSel* __restrict__ sel) {
  const int n = blockIdx.y, HW = H * W;
  const unsigned char* g = gt + (i64)n * Hg * Wg;
  const float sgy = (float)Hg / (float)H, sgx = (float)Wg / (float)W;
  if (loss_type != 0) {       // hinge losses: mean over the pairs of relu(.)^e; d/ds = -/+ e relu(.)^(e-1) w / n
    const double np = stats[0], nn = stats[3];
    if (blockIdx.x == 0 && n == 0 && threadIdx.x == 0) {
      losses[0] = np > 0 ? (float)((double)w_pos * stats[1] / np) : 0.f;     // (an empty set gives NaN in the reference)
      losses[1] = nn > 0 ? (float)((double)w_neg * stats[4] / nn) : 0.f;
      losses[2] = 0.f;
      losses[3] = 0.f;
    }
    const double cp = np > 0 ? (double)w_pos / np : 0.0, cn = nn > 0 ? (double)w_neg / nn : 0.0;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
      const int y = p / W, x = p - y * W;
      const int ctr = g[(i64)nearest_src(y, sgy, Hg) * Wg + nearest_src(x, sgx, Wg)];
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const i64 o = ((i64)n * 9 + k) * HW + p;
        float gr = 0.f;
        if (ctr != 255) {
          const int cls = src_pair_class(g, Hg, Wg, sgy, sgx, H, W, y, x, k, dil, ctr);
          const double s = (double)sim[o];
          const double h = cls == 1 ? (double)m0 - s : s - (double)m1;
          if (h > 0.0) gr = (float)((cls == 1 ? -cp : cn) * (loss_type == 1 ? 1.0 : 2.0 * h)) * sel_weight(sel, cls == 1 ? 0 : 1, sim[o]);
        }
        gsim[o] = gr;
      }
    }
    return;
  }
  const SrcMoments mp = moments(stats), mn = moments(stats + 3);
  if (blockIdx.x == 0 && n == 0 && threadIdx.x == 0) {
    losses[0] = (float)(-mp.mean * w_pos);
    losses[1] = (float)(mn.mean * w_neg);
    losses[2] = (float)(mp.std * w_pos_std);
    losses[3] = (float)(mn.std * w_neg_std);
  }
  // d(-w mean)/ds = -w/n ; d(w std)/ds = w (s-mean)/((n-1) std)
  const double pa = mp.n > 0 ? -(double)w_pos / mp.n : 0.0;
  const double pb = (mp.n > 1 && mp.std > 0) ? (double)w_pos_std / ((mp.n - 1.0) * mp.std) : 0.0;
  const double na = mn.n > 0 ? (double)w_neg / mn.n : 0.0;
  const double nb = (mn.n > 1 && mn.std > 0) ? (double)w_neg_std / ((mn.n - 1.0) * mn.std) : 0.0;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
    const int y = p / W, x = p - y * W;
    const int ctr = g[(i64)nearest_src(y, sgy, Hg) * Wg + nearest_src(x, sgx, Wg)];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const i64 o = ((i64)n * 9 + k) * HW + p;
      float gr = 0.f;
      if (ctr != 255) {
        const int cls = src_pair_class(g, Hg, Wg, sgy, sgx, H, W, y, x, k, dil, ctr);
        const double s = (double)sim[o];
        gr = (cls == 1 ? (float)(pa + pb * (s - mp.mean)) : (float)(na + nb * (s - mn.mean))) * sel_weight(sel, cls == 1 ? 0 : 1, sim[o]);
      }
      gsim[o] = gr;
    }
  }
}

// ---- softmax of the nearest-down-scaled student logits.  grid: (blocks over H*W, N)
__global__ void softmax_down_kernel(const float* __restrict__ logits, int C, int h, int w, int ds, float* __restrict__ prob, int H, int W) {
  const int n = blockIdx.y, HW = H * W;
  const float* lp = logits + (i64)n * C * h * w;
  float* pp = prob + (i64)n * C * HW;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
    const int y = p / W, x = p - y * W;
    const int so = (y * ds) * w + x * ds;
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, lp[(i64)c * h * w + so]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(lp[(i64)c * h * w + so] - mx);
    const float inv = 1.f / se;
    for (int c = 0; c < C; ++c) pp[(i64)c * HW + p] = expf(lp[(i64)c * h * w + so] - mx) * inv;
  }
}

// ---- target validity: centre label != 255 and all nine dilated neighbours un-mixed (and inside the map)
__global__ __launch_bounds__(256) void trg_valid_kernel(const unsigned char* __restrict__ gt, const unsigned char* __restrict__ mix,
                                                        int H, int W, int Hg, int Wg, int dil, unsigned char* __restrict__ valid,
                                                        unsigned char* __restrict__ all9, unsigned long long* __restrict__ count) {
  __shared__ double sm[16];
  const int n = blockIdx.y, HW = H * W;
  const unsigned char* g = gt + (i64)n * Hg * Wg;
  const unsigned char* m = mix + (i64)n * Hg * Wg;
  const float sgy = (float)Hg / (float)H, sgx = (float)Wg / (float)W;
  double cnt = 0.0;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
    const int y = p / W, x = p - y * W;
    bool all = true;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int sy = y + (k / 3 - 1) * dil, sx = x + (k % 3 - 1) * dil;
      const bool in = sy >= 0 && sy < H && sx >= 0 && sx < W;
      all = all && in && (m[(i64)nearest_src(in ? sy : 0, sgy, Hg) * Wg + nearest_src(in ? sx : 0, sgx, Wg)] == 0);
    }
    const int ctr = g[(i64)nearest_src(y, sgy, Hg) * Wg + nearest_src(x, sgx, Wg)];
    const bool v = all && ctr != 255;
    valid[(i64)n * HW + p] = v;
    if (all9) all9[(i64)n * HW + p] = all;
    cnt += v ? 1.0 : 0.0;
  }
  cnt = block_sum_d(cnt, sm);
  if (threadIdx.x == 0 && cnt > 0.0) atomicAdd(count, (unsigned long long)cnt);
}

// ---- top-k target losses.  One thread per pixel: 9-element sort in registers.
__global__ __launch_bounds__(256) void topk_loss_kernel(const float* __restrict__ ema_sim, const float* __restrict__ prob,
                                                        const unsigned char* __restrict__ valid, const unsigned long long* __restrict__ count,
                                                        int C, int H, int W, int dil, int top_k, float w_pos, float w_neg,
                                                        float* __restrict__ gP, double* __restrict__ acc, float* __restrict__ gS) {
  __shared__ double sm[16];
  const int n = blockIdx.y, HW = H * W;
  const double cnt = (double)count[0];
  const bool all_pairs = top_k == 0;          // top_k=None in the reference: every one of the 9 pairs, both losses (:229-231)
  const float cpos = cnt > 1.0 ? (float)((double)w_pos / (cnt * (all_pairs ? 9 : top_k + 1))) : 0.f;
  const float cneg = cnt > 1.0 ? (float)((double)w_neg / (cnt * (all_pairs ? 9 : top_k))) : 0.f;
  const float* pp = prob + (i64)n * C * HW;
  double spos = 0.0, sneg = 0.0;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
    const i64 base = (i64)n * 9 * HW + p;
    if (!valid[(i64)n * HW + p]) {
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        gP[base + (i64)k * HW] = 0.f;
        if (gS) gS[base + (i64)k * HW] = 0.f;
      }
      continue;
    }
    const int y = p / W, x = p - y * W;
    float s[9], P[9];
    int id[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      s[k] = ema_sim[base + (i64)k * HW];
      id[k] = k;
      const int sy = y + (k / 3 - 1) * dil, sx = x + (k % 3 - 1) * dil;  // always inside for valid pixels
      const int q = sy * W + sx;
      float d = 0.f;
      for (int c = 0; c < C; ++c) d = fmaf(pp[(i64)c * HW + p], pp[(i64)c * HW + q], d);
      P[k] = d;
    }
    if (all_pairs) {
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        spos += (double)(-s[k] * P[k]);
        sneg += (double)(-(1.f - s[k]) * (1.f - P[k]));
        gP[base + (i64)k * HW] = -s[k] * cpos + (1.f - s[k]) * cneg;
        if (gS) gS[base + (i64)k * HW] = -P[k] * cpos + (1.f - P[k]) * cneg;     // d/d sim: the teacher side (proj_net's weights)
      }
      continue;
    }
    // stable insertion sort, descending similarity (ties keep the lower index first)
#pragma unroll
    for (int i = 1; i < 9; ++i) {
#pragma unroll
      for (int j = i; j > 0; --j) {
        if (s[j] > s[j - 1]) {
          const float ts = s[j]; s[j] = s[j - 1]; s[j - 1] = ts;
          const float tp = P[j]; P[j] = P[j - 1]; P[j - 1] = tp;
          const int ti = id[j]; id[j] = id[j - 1]; id[j - 1] = ti;
        }
      }
    }
    float g[9], gs[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) { g[k] = 0.f; gs[k] = 0.f; }
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      float gj = 0.f, gsj = 0.f;
      if (j <= top_k) {            // top-(k+1) largest: loc_pos = -sim * P
        spos += (double)(-s[j] * P[j]);
        gj = -s[j] * cpos;
        gsj = -P[j] * cpos;
      } else if (j >= 9 - top_k) { // top-k smallest: loc_neg = -(1-sim) * (1-P)
        sneg += (double)(-(1.f - s[j]) * (1.f - P[j]));
        gj = (1.f - s[j]) * cneg;
        gsj = (1.f - P[j]) * cneg;
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) if (id[j] == k) { g[k] = gj; gs[k] = gsj; }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      gP[base + (i64)k * HW] = g[k];
      if (gS) gS[base + (i64)k * HW] = gs[k];
    }
  }
  spos = block_sum_d(spos, sm);
  sneg = block_sum_d(sneg, sm);
  if (threadIdx.x == 0) {
    if (spos != 0.0) atomicAdd(&acc[0], spos);
    if (sneg != 0.0) atomicAdd(&acc[1], sneg);
  }
}

// ---- gradient of the cross-probabilities P_k(r) = sum_c p_c(r) p_c(r+D_k) into the (full 1/4-res) student logits.
// detach_unfold=True: only the centre factor p_c(r) carries gradient.  unfold_grad (detach_unfold=False): the unfolded
// factor does too -- pixel r is the k-neighbour of r+D_{8-k}, so its coefficient gains gP[8-k, r+D_k] (gather form, no atomics).
__global__ __launch_bounds__(256) void cross_prob_bwd_kernel(const float* __restrict__ prob, const float* __restrict__ gP, int C, int H,
                                                             int W, int dil, int ds, int unfold_grad, float* __restrict__ dlogits, int h, int w) {
  const int n = blockIdx.y, HW = H * W;
  const float* pp = prob + (i64)n * C * HW;
  float* dl = dlogits + (i64)n * C * h * w;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
    const int y = p / W, x = p - y * W;
    float g[9];
    int off[9];
    bool any = false;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      g[k] = gP[((i64)n * 9 + k) * HW + p];
      const int sy = y + (k / 3 - 1) * dil, sx = x + (k % 3 - 1) * dil;
      const bool in = sy >= 0 && sy < H && sx >= 0 && sx < W;
      off[k] = in ? sy * W + sx : -1;
      if (unfold_grad && in) g[k] += gP[((i64)n * 9 + (8 - k)) * HW + off[k]];
      any = any || g[k] != 0.f;
    }
    if (!any) continue;
    float dot = 0.f;  // sum_j p_j * dprob_j
    for (int c = 0; c < C; ++c) {
      float d = 0.f;
#pragma unroll
      for (int k = 0; k < 9; ++k) if (off[k] >= 0) d = fmaf(g[k], pp[(i64)c * HW + off[k]], d);
      dot = fmaf(pp[(i64)c * HW + p], d, dot);
    }
    const int so = (y * ds) * w + x * ds;
    for (int c = 0; c < C; ++c) {
      float d = 0.f;
#pragma unroll
      for (int k = 0; k < 9; ++k) if (off[k] >= 0) d = fmaf(g[k], pp[(i64)c * HW + off[k]], d);
      dl[(i64)c * h * w + so] += pp[(i64)c * HW + p] * (d - dot);
    }
  }
}

__global__ void sim_loss_finalize_kernel(const double* __restrict__ acc, const unsigned long long* __restrict__ count, int top_k,
                                         float w_pos, float w_neg, float* __restrict__ out) {
  const double cnt = (double)count[0];
  out[0] = cnt > 1.0 ? (float)((double)w_pos * acc[0] / (cnt * (top_k == 0 ? 9 : top_k + 1))) : 0.f;
  out[1] = cnt > 1.0 ? (float)((double)w_neg * acc[1] / (cnt * (top_k == 0 ? 9 : top_k))) : 0.f;
}

inline int px_blocks(i64 n) {
  i64 g = (n + 255) / 256;
  if (g > 4096) g = 4096;
  return g < 1 ? 1 : (int)g;
}

}  // namespace

extern "C" int pfst_sim_map(const float* feat, int N, int C, int H, int W, int dil, int sim_type, float sigma, float* sim, float* norm,
                            pfst_stream_t stream) {
  PFST_CHECK_ARG(feat && sim && norm && N > 0 && N <= 65535 && C > 0 && H > 0 && W > 0 && dil >= 1);
  PFST_CHECK_ARG(sim_type == 0 || (sim_type == 1 && sigma > 0.f));
  const dim3 grid(px_blocks((i64)H * W), N);
  if (sim_type == 0 && simq_ok(feat, sim, C, H, W, dil) && (reinterpret_cast<uintptr_t>(norm) & 15) == 0) {
    const dim3 gq((unsigned)(((i64)H * W) / 256), N);
    const size_t lds = 4 * SIMQ_VALS * 64 * sizeof(float);
    if (C % 8 == 0 && C >= 64) {                   // 8 channel slices per strip: 16 waves per CU at the BASELINE shape
      if (dil == 1) hipLaunchKernelGGL((sim_map_cos_q_kernel<1, 8>), gq, dim3(512), lds, (hipStream_t)stream, feat, C, H, W, sim, norm);
      else hipLaunchKernelGGL((sim_map_cos_q_kernel<2, 8>), gq, dim3(512), lds, (hipStream_t)stream, feat, C, H, W, sim, norm);
    } else {
      if (dil == 1) hipLaunchKernelGGL((sim_map_cos_q_kernel<1, 4>), gq, dim3(256), lds, (hipStream_t)stream, feat, C, H, W, sim, norm);
      else hipLaunchKernelGGL((sim_map_cos_q_kernel<2, 4>), gq, dim3(256), lds, (hipStream_t)stream, feat, C, H, W, sim, norm);
    }
  } else if (sim_type == 1)
    hipLaunchKernelGGL(sim_map_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, feat, C, H, W, dil, 1.f / (sigma * sigma), sim, norm);
  else
    hipLaunchKernelGGL(sim_map_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, feat, C, H, W, dil, 0.f, sim, norm);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

extern "C" int pfst_sim_map_bwd(const float* feat, const float* sim, const float* norm, const float* gsim, int N, int C, int H, int W, int dil,
                                int sim_type, float sigma, float* dfeat, int accumulate, float* coef_ws, pfst_stream_t stream) {
  PFST_CHECK_ARG(feat && sim && norm && gsim && dfeat && N > 0 && N <= 65535 && C > 0 && H > 0 && W > 0 && dil >= 1);
  PFST_CHECK_ARG(sim_type == 0 || (sim_type == 1 && sigma > 0.f));
  const dim3 grid(px_blocks((i64)H * W), N);
  if (sim_type == 0 && coef_ws && simq_ok(feat, dfeat, C, H, W, dil) && (reinterpret_cast<uintptr_t>(coef_ws) & 15) == 0) {
    PFST_CHECK_ARG(coef_ws != nullptr);
    hipLaunchKernelGGL(sim_bwd_coef_kernel, grid, dim3(256), 0, (hipStream_t)stream, sim, norm, gsim, H, W, dil, coef_ws);
    const int cpw = 16;                                   // channels per wave
    const dim3 gq((unsigned)(((i64)H * W) / 256), cdiv(C, 4 * cpw), N);
    if (dil == 1) hipLaunchKernelGGL(sim_map_bwd_cos_q_kernel<1>, gq, dim3(256), 0, (hipStream_t)stream, feat, coef_ws, C, H, W, cpw, dfeat, accumulate);
    else hipLaunchKernelGGL(sim_map_bwd_cos_q_kernel<2>, gq, dim3(256), 0, (hipStream_t)stream, feat, coef_ws, C, H, W, cpw, dfeat, accumulate);
  } else if (sim_type == 1)
    hipLaunchKernelGGL(sim_map_bwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, feat, sim, norm, gsim, C, H, W, dil,
                       1.f / (sigma * sigma), dfeat, accumulate);
  else
    hipLaunchKernelGGL(sim_map_bwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, feat, sim, norm, gsim, C, H, W, dil, 0.f, dfeat,
                       accumulate);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

extern "C" int pfst_src_sim_stats(const float* sim, const unsigned char* gt, int N, int H, int W, int Hg, int Wg, int dil, int loss_type,
                                  float margin_pos, float margin_neg, double* stats, const void* select, pfst_stream_t stream) {
  PFST_CHECK_ARG(sim && gt && stats && N > 0 && N <= 65535 && H > 0 && W > 0 && Hg > 0 && Wg > 0 && dil >= 1);
  PFST_CHECK_ARG(loss_type >= 0 && loss_type <= 2);
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(stats, 0, 6 * sizeof(double), s) != hipSuccess) return PFST_ERR_LAUNCH;
  const int gxs = px_blocks((i64)H * W);
  double* det = nullptr;
  if (pfst_deterministic()) {
    det = static_cast<double*>(pfst_det_scratch((size_t)gxs * N * 6 * sizeof(double), s));
    PFST_CHECK_ARG(det != nullptr);
  }
  hipLaunchKernelGGL(src_stats_kernel, dim3(gxs, N), dim3(256), 0, s, sim, gt, H, W, Hg, Wg, dil, loss_type, margin_pos,
                     margin_neg, stats, reinterpret_cast<const SrcSel*>(select), det);
  if (det) hipLaunchKernelGGL(src_stats_det_sum_kernel, dim3(1), dim3(64), 0, s, det, gxs * N, stats);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

extern "C" int pfst_src_sim_grad(const float* sim, const unsigned char* gt, int N, int H, int W, int Hg, int Wg, int dil, int loss_type,
                                 float margin_pos, float margin_neg, const double* stats,
                                 float w_pos, float w_neg, float w_pos_std, float w_neg_std, float* gsim, float* losses,
                                 const void* select, pfst_stream_t stream) {
  PFST_CHECK_ARG(sim && gt && stats && gsim && losses && N > 0 && N <= 65535 && H > 0 && W > 0 && Hg > 0 && Wg > 0 && dil >= 1);
  PFST_CHECK_ARG(loss_type >= 0 && loss_type <= 2);
  hipLaunchKernelGGL(src_grad_kernel, dim3(px_blocks((i64)H * W), N), dim3(256), 0, (hipStream_t)stream, sim, gt, H, W, Hg, Wg, dil,
                     loss_type, margin_pos, margin_neg, stats, w_pos, w_neg, w_pos_std, w_neg_std, gsim, losses,
                     reinterpret_cast<const SrcSel*>(select));
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

extern "C" int pfst_src_sim_select_bytes(void) { return (int)(2 * sizeof(SrcSel) + 512 * sizeof(unsigned int)); }

extern "C" int pfst_src_sim_select(const float* sim, const unsigned char* gt, int N, int H, int W, int Hg, int Wg, int dil, double src_perc,
                                   void* select, pfst_stream_t stream) {
  PFST_CHECK_ARG(sim && gt && select && N > 0 && N <= 65535 && H > 0 && W > 0 && Hg > 0 && Wg > 0 && dil >= 1);
  PFST_CHECK_ARG(src_perc >= 0.0 && src_perc <= 1.0 && (reinterpret_cast<uintptr_t>(select) & 7) == 0);
  hipStream_t s = (hipStream_t)stream;
  SrcSel* st = reinterpret_cast<SrcSel*>(select);
  unsigned int* hist = reinterpret_cast<unsigned int*>(st + 2);
  for (int pass = 0; pass < 4; ++pass) {
    if (pass == 0) hipLaunchKernelGGL(src_sel_scan_kernel, dim3(1), dim3(256), 0, s, st, hist, src_perc, 2);   // initialise shift = 24
    hipLaunchKernelGGL(src_sel_hist_kernel, dim3(px_blocks((i64)H * W), N), dim3(256), 0, s, sim, gt, H, W, Hg, Wg, dil, st, hist);
    hipLaunchKernelGGL(src_sel_scan_kernel, dim3(1), dim3(256), 0, s, st, hist, src_perc, pass == 0 ? 1 : 0);
  }
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

extern "C" int pfst_softmax_down(const float* logits, int N, int C, int h, int w, int ds, float* prob, int H, int W, pfst_stream_t stream) {
  PFST_CHECK_ARG(logits && prob && N > 0 && N <= 65535 && C > 0 && h > 0 && w > 0 && ds >= 1 && H > 0 && W > 0);
  PFST_CHECK_ARG((H - 1) * ds < h && (W - 1) * ds < w);
  hipLaunchKernelGGL(softmax_down_kernel, dim3(px_blocks((i64)H * W), N), dim3(256), 0, (hipStream_t)stream, logits, C, h, w, ds, prob, H, W);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

extern "C" int pfst_trg_valid_mask(const unsigned char* gt, const unsigned char* mix_mask, int N, int H, int W, int Hg, int Wg, int dil,
                                   unsigned char* valid, unsigned char* all9, unsigned long long* count, pfst_stream_t stream) {
  PFST_CHECK_ARG(gt && mix_mask && valid && count && N > 0 && N <= 65535 && H > 0 && W > 0 && Hg > 0 && Wg > 0 && dil >= 1);
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(count, 0, sizeof(unsigned long long), s) != hipSuccess) return PFST_ERR_LAUNCH;
  hipLaunchKernelGGL(trg_valid_kernel, dim3(px_blocks((i64)H * W), N), dim3(256), 0, s, gt, mix_mask, H, W, Hg, Wg, dil, valid, all9, count);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

extern "C" int pfst_sim_topk_loss(const float* ema_sim, const float* prob, const unsigned char* valid, const unsigned long long* count,
                                  int N, int C, int H, int W, int dil, int top_k, float w_pos, float w_neg, float* gP, double* acc,
                                  float* g_sim, pfst_stream_t stream) {
  PFST_CHECK_ARG(ema_sim && prob && valid && count && gP && acc && N > 0 && N <= 65535 && C > 0 && H > 0 && W > 0 && dil >= 1);
  PFST_CHECK_ARG(top_k >= 0 && 2 * top_k + 1 <= 9);    // 0 = all nine pairs (top_k=None)
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(acc, 0, 2 * sizeof(double), s) != hipSuccess) return PFST_ERR_LAUNCH;
  hipLaunchKernelGGL(topk_loss_kernel, dim3(px_blocks((i64)H * W), N), dim3(256), 0, s, ema_sim, prob, valid, count, C, H, W, dil, top_k,
                     w_pos, w_neg, gP, acc, g_sim);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

extern "C" int pfst_cross_prob_bwd(const float* prob, const float* gP, int N, int C, int H, int W, int dil, int ds, int unfold_grad,
                                   float* dlogits, int h, int w, pfst_stream_t stream) {
  PFST_CHECK_ARG(prob && gP && dlogits && N > 0 && N <= 65535 && C > 0 && H > 0 && W > 0 && dil >= 1 && ds >= 1);
  PFST_CHECK_ARG((H - 1) * ds < h && (W - 1) * ds < w);
  hipLaunchKernelGGL(cross_prob_bwd_kernel, dim3(px_blocks((i64)H * W), N), dim3(256), 0, (hipStream_t)stream, prob, gP, C, H, W, dil, ds,
                     unfold_grad, dlogits, h, w);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

extern "C" int pfst_sim_loss_finalize(const double* acc, const unsigned long long* count, int top_k, float w_pos, float w_neg, float* out, pfst_stream_t stream) {
  PFST_CHECK_ARG(acc && count && out && top_k >= 0);
  hipLaunchKernelGGL(sim_loss_finalize_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, acc, count, top_k, w_pos, w_neg, out);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}
