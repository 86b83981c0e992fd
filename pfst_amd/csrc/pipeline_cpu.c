// CPU pixel kernels of the data pipeline (pfst_amd/pipeline.py): plain C, one pass per operation, compiled with gcc
// (-O2 -ffp-contract=off: every float operation rounds exactly as the NumPy expression it mirrors, so the two implementations
// agree bit for bit -- tests/test_data_loader_cpu.py).  The reference does these steps in OpenCV / mmcv C++
// (rsiseg/datasets/pipelines/transforms.py:11-260 Resize, :942-1059 PhotoMetricDistortion, :404-451 Normalize); NumPy's
// temporaries made them 50-100 ms per sample, i.e. the loader, not the GPU, set the pace of real-data training.
// Images: H x W x 3 uint8, interleaved.  No global state, re-entrant.
#include <math.h>
#include <stdint.h>
#include <stddef.h>

static inline uint8_t sat_u8(float v) {          // np.clip(np.rint(v), 0, 255).astype(np.uint8)
  float r = rintf(v);                            // round half to even, like np.rint
  if (r < 0.f) r = 0.f;
  if (r > 255.f) r = 255.f;
  return (uint8_t)r;
}

static inline uint8_t clip_trunc_u8(float v) {   // np.clip(v, 0, 255).astype(np.uint8): truncation
  if (v < 0.f) v = 0.f;
  if (v > 255.f) v = 255.f;
  return (uint8_t)v;
}

// rows [y_beg, y_end) x columns [x_beg, x_end) of the bilinear resize of src (h x w x 3) to H x W:
// lo / hi source indices and the weight of `hi` per output row / column are passed in (pipeline._src_index), so the geometry has ONE
// definition.  out: (y_end - y_beg) x (x_end - x_beg) x 3.
void pfst_cpu_resize_window_u8(const uint8_t* src, int h, int w, const int64_t* y0, const int64_t* y1, const float* fy,
                               const int64_t* x0, const int64_t* x1, const float* fx, int ny, int nx, uint8_t* out) {
  (void)h;
  for (int i = 0; i < ny; ++i) {
    const uint8_t* r0 = src + (size_t)y0[i] * w * 3;
    const uint8_t* r1 = src + (size_t)y1[i] * w * 3;
    const float wy1 = fy[i], wy0 = 1.0f - fy[i];
    uint8_t* o = out + (size_t)i * nx * 3;
    for (int j = 0; j < nx; ++j) {
      const float wx1 = fx[j], wx0 = 1.0f - fx[j];
      const uint8_t *a = r0 + x0[j] * 3, *b = r0 + x1[j] * 3, *c = r1 + x0[j] * 3, *d = r1 + x1[j] * 3;
      for (int ch = 0; ch < 3; ++ch) {
        const float top = (float)a[ch] * wx0 + (float)b[ch] * wx1;
        const float bot = (float)c[ch] * wx0 + (float)d[ch] * wx1;
        o[j * 3 + ch] = sat_u8(top * wy0 + bot * wy1);
      }
    }
  }
}

// cv2.cvtColor(BGR2HSV) restated (pipeline.bgr2hsv_u8): H in [0, 180), S, V in [0, 255]
void pfst_cpu_bgr2hsv_u8(const uint8_t* img, int64_t n, uint8_t* out) {
  for (int64_t i = 0; i < n; ++i) {
    const float b = img[3 * i], g = img[3 * i + 1], r = img[3 * i + 2];
    const float v = fmaxf(fmaxf(b, g), r), mn = fminf(fminf(b, g), r);
    const float d = v - mn;
    const float s = v > 0.f ? d / fmaxf(v, 1e-12f) * 255.0f : 0.0f;
    const float dd = fmaxf(d, 1e-12f);
    float hh = (v == r ? (g - b) / dd : (v == g ? 2.0f + (b - r) / dd : 4.0f + (r - g) / dd)) * 60.0f;
    if (d == 0.f) hh = 0.f;
    if (hh < 0.f) hh = hh + 360.0f;
    hh = hh / 2.0f;
    float hr = rintf(hh);                          // in [0, 180]: `% 180` only folds 180 back to 0
    if (hr >= 180.0f) hr -= 180.0f;
    out[3 * i] = clip_trunc_u8(hr);
    out[3 * i + 1] = clip_trunc_u8(rintf(s));
    out[3 * i + 2] = clip_trunc_u8(v);
  }
}

void pfst_cpu_hsv2bgr_u8(const uint8_t* hsv, int64_t n, uint8_t* out) {
  for (int64_t i = 0; i < n; ++i) {
    const float h = (float)hsv[3 * i] * 2.0f;
    const float s = (float)hsv[3 * i + 1] / 255.0f;
    const float v = hsv[3 * i + 2];
    const float c = v * s;
    const float hp = h / 60.0f;
    const float fl = floorf(hp);                   // hp in [0, 8.5]
    const float hm2 = hp - 2.0f * floorf(hp * 0.5f);   // == fmodf(hp, 2): every step is exact for these magnitudes
    const float x = c * (1.0f - fabsf(hm2 - 1.0f));
    const int sector = (int)fl % 6;
    float r, g, b;
    switch (sector) {
      case 0: r = c; g = x; b = 0.f; break;
      case 1: r = x; g = c; b = 0.f; break;
      case 2: r = 0.f; g = c; b = x; break;
      case 3: r = 0.f; g = x; b = c; break;
      case 4: r = x; g = 0.f; b = c; break;
      default: r = c; g = 0.f; b = x; break;
    }
    const float m = v - c;
    out[3 * i] = sat_u8(b + m);
    out[3 * i + 1] = sat_u8(g + m);
    out[3 * i + 2] = sat_u8(r + m);
  }
}

// transforms.py:975-979 `convert`: clip(img * alpha + beta, 0, 255) truncated to uint8; stride: bytes between consecutive elements
// (1 for a whole image, 3 for one channel of an interleaved image)
void pfst_cpu_convert_u8(const uint8_t* in, int64_t n, int stride, float alpha, float beta, uint8_t* out) {
  for (int64_t i = 0; i < n; ++i) out[i * stride] = clip_trunc_u8((float)in[i * stride] * alpha + beta);
}

// hue shift: (h + delta) % 180 on channel 0 of an interleaved HSV image (python-style modulo of ints)
void pfst_cpu_hue_shift_u8(uint8_t* hsv, int64_t n, int delta) {
  for (int64_t i = 0; i < n; ++i) {
    int v = ((int)hsv[3 * i] + delta) % 180;
    if (v < 0) v += 180;
    hsv[3 * i] = (uint8_t)v;
  }
}

// mmcv.imnormalize: out[p][c] = (img[p][to_rgb ? 2 - c : c] - mean[c]) / std[c], float32, still H x W x 3
void pfst_cpu_normalize_u8(const uint8_t* img, int64_t n, const float* mean, const float* std, int to_rgb, float* out) {
  for (int64_t i = 0; i < n; ++i)
    for (int c = 0; c < 3; ++c) out[3 * i + c] = ((float)img[3 * i + (to_rgb ? 2 - c : c)] - mean[c]) / std[c];
}
