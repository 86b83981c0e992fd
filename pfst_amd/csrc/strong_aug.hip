// DACS strong augmentation of the mixed image: colour jitter and gaussian blur (HBM-bound, 3-channel images).
// Reference: rsiseg/models/utils/dacs_transforms.py:44-107 which delegates the arithmetic to kornia
// (kornia.augmentation.ColorJitter(brightness=contrast=saturation=hue=s), kornia.filters.GaussianBlur2d, reflect
// border).  kornia is an un-vendored, un-pinned third-party dependency (requirements.sh:1) that is absent from the
// image: the formulas below restate kornia 0.6's documented behaviour (additive brightness, multiplicative contrast,
// HSV saturation scale / hue shift, each followed by clamp to [0,1]; transform order = a random permutation) --
// PARITY UNPINNED for these two transforms (no reference test or fixture covers them).
#include "common.h"
#include "../../include/pfst_hip.h"

namespace {

constexpr float TWO_PI = 6.283185307179586f;

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }

__device__ __forceinline__ void rgb2hsv(float r, float g, float b, float& h, float& s, float& v) {
  const float mx = fmaxf(r, fmaxf(g, b)), mn = fminf(r, fminf(g, b));
  float d = mx - mn;
  v = mx;
  s = d / (mx + 1e-8f);
  if (d == 0.f) d = 1.f;
  const float rc = mx - r, gc = mx - g, bc = mx - b;
  float hh;
  if (r >= g && r >= b) hh = bc - gc;                 // first arg-max like torch.max
  else if (g >= b) hh = (rc - bc) + 2.f * d;
  else hh = (gc - rc) + 4.f * d;
  hh = hh / d / 6.f;
  hh = hh - floorf(hh);
  h = TWO_PI * hh;
}
__device__ __forceinline__ void hsv2rgb(float h, float s, float v, float& r, float& g, float& b) {
  const float h6 = h / TWO_PI * 6.f;
  float fl = floorf(h6);
  int hi = (int)fl % 6;
  if (hi < 0) hi += 6;
  const float f = h6 - fl;
  const float p = v * (1.f - s), q = v * (1.f - f * s), t = v * (1.f - (1.f - f) * s);
  switch (hi) {
    case 0: r = v; g = t; b = p; break;
    case 1: r = q; g = v; b = p; break;
    case 2: r = p; g = v; b = t; break;
    case 3: r = p; g = q; b = v; break;
    case 4: r = t; g = p; b = v; break;
    default: r = v; g = p; b = q; break;
  }
}

// params per image: [brightness, contrast, saturation, hue(rad), order0..3]   grid: (blocks over HW, N)
__global__ void color_jitter_kernel(float* __restrict__ img, const float* __restrict__ params, const float* __restrict__ mean,
                                    const float* __restrict__ stdv, i64 HW, int denorm) {
  const int n = blockIdx.y;
  const float* pr = params + n * 8;
  const float bf = pr[0], cf = pr[1], sf = pr[2], hf = pr[3];
  int order[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) order[i] = (int)pr[4 + i];
  float m[3] = {0.f, 0.f, 0.f}, sd[3] = {1.f, 1.f, 1.f};
  if (denorm) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { m[c] = mean[c]; sd[c] = stdv[c]; }
  }
  float* base = img + (i64)n * 3 * HW;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (i64)gridDim.x * blockDim.x) {
    float r = base[i], g = base[HW + i], b = base[2 * HW + i];
    if (denorm) { r = (r * sd[0] + m[0]) / 255.f; g = (g * sd[1] + m[1]) / 255.f; b = (b * sd[2] + m[2]) / 255.f; }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int op = order[k];
      if (op == 0) { r = clamp01(r + bf - 1.f); g = clamp01(g + bf - 1.f); b = clamp01(b + bf - 1.f); }
      else if (op == 1) { r = clamp01(r * cf); g = clamp01(g * cf); b = clamp01(b * cf); }
      else {
        float h, s, v;
        rgb2hsv(r, g, b, h, s, v);
        if (op == 2) s = clamp01(s * sf);
        else { h = fmodf(h + hf, TWO_PI); if (h < 0.f) h += TWO_PI; }
        hsv2rgb(h, s, v, r, g, b);
      }
    }
    if (denorm) { r = (r * 255.f - m[0]) / sd[0]; g = (g * 255.f - m[1]) / sd[1]; b = (b * 255.f - m[2]) / sd[2]; }
    base[i] = r; base[HW + i] = g; base[2 * HW + i] = b;
  }
}

__device__ __forceinline__ int reflect(int i, int n) {   // torch 'reflect' padding (no edge repeat)
  if (n == 1) return 0;
  const int period = 2 * (n - 1);
  i = i % period;
  if (i < 0) i += period;
  return i < n ? i : period - i;
}

// one separable pass; taps[n][K] per image.  grid: (blocks over HW, C, N)
__global__ void blur_pass_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ taps, int K, int C, int H,
                                 int W, int horizontal, int r_lo, int r_hi) {
  const int c = blockIdx.y, n = blockIdx.z;
  const float* xp = x + ((i64)n * C + c) * H * W;
  float* yp = y + ((i64)n * C + c) * H * W;
  const float* tp = taps + (i64)n * K;
  const int half = K / 2;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < H * W; i += gridDim.x * blockDim.x) {
    const int yy = i / W, xx = i - yy * W;
    float acc = 0.f;
    for (int t = r_lo; t <= r_hi; ++t) {       // taps outside [r_lo, r_hi] are below fp32 resolution of the sum
      const float wgt = tp[t];
      const float v = horizontal ? xp[yy * W + reflect(xx + t - half, W)] : xp[reflect(yy + t - half, H) * W + xx];
      acc = fmaf(wgt, v, acc);
    }
    yp[i] = acc;
  }
}

}  // namespace

extern "C" int pfst_color_jitter(float* img, const float* params, const float* mean3, const float* std3, int N, long long HW,
                                 int denorm, pfst_stream_t stream) {
  PFST_CHECK_ARG(img && params && N > 0 && N <= 65535 && HW > 0 && (!denorm || (mean3 && std3)));
  i64 g = (HW + 1023) / 1024;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(color_jitter_kernel, dim3((int)g, N), dim3(256), 0, (hipStream_t)stream, img, params, mean3, std3, (i64)HW, denorm);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

extern "C" int pfst_gaussian_blur(const float* x, float* tmp, float* y, const float* taps_y, int Ky, const float* taps_x, int Kx,
                                  int N, int C, int H, int W, int reach, pfst_stream_t stream) {
  PFST_CHECK_ARG(x && tmp && y && taps_y && taps_x && (Ky & 1) && (Kx & 1) && N > 0 && C > 0 && H > 0 && W > 0 && N <= 65535 && reach >= 0);
  int gx = cdiv((i64)H * W, 1024);
  hipStream_t s = (hipStream_t)stream;
  const int lo_x = max(0, Kx / 2 - reach), hi_x = min(Kx - 1, Kx / 2 + reach);
  const int lo_y = max(0, Ky / 2 - reach), hi_y = min(Ky - 1, Ky / 2 + reach);
  hipLaunchKernelGGL(blur_pass_kernel, dim3(gx, C, N), dim3(256), 0, s, x, tmp, taps_x, Kx, C, H, W, 1, lo_x, hi_x);
  hipLaunchKernelGGL(blur_pass_kernel, dim3(gx, C, N), dim3(256), 0, s, tmp, y, taps_y, Ky, C, H, W, 0, lo_y, hi_y);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}
