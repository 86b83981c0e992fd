// Depthwise 3x3 convolution, stride 1, padding = dilation (HBM-bound 9-tap stencil).
// Reference: mmcv DepthwiseSeparableConvModule.depthwise_conv inside
// rsiseg/models/decode_heads/sep_aspp_head.py:17-26 (ASPP, dilation 12/24/36) and :63-77 (sep_bottleneck).
//
// One workgroup owns a strip of rows of one (image, channel) plane.  The strip plus its two halo
// row-bands is staged in LDS as three row-bands (top / middle / bottom taps) so each input element is
// fetched from HBM/L2 once per band and every tap is an LDS read; rows are read with coalesced
// 256-byte wave accesses.  The data gradient is the same stencil with mirrored taps.
#include "common.h"
#include "../../include/pfst_hip.h"

namespace {

constexpr int DW_ROWS = 8;  // output rows per workgroup

// LDS: 3 bands x DW_ROWS rows x W floats (W <= 1024 -> 96 KB max; typical 128/256 -> 12/24 KB)
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const float* __restrict__ x, i64 x_bs, const float* __restrict__ w,
                                                        float* __restrict__ y, i64 y_bs, int C, int H, int W, int dil,
                                                        int flip, int accumulate) {
  extern __shared__ float tile[];  // [3][DW_ROWS][W]
  const int c = blockIdx.y, n = blockIdx.z;
  const int y0 = blockIdx.x * DW_ROWS;
  const float* xp = x + (i64)n * x_bs + (i64)c * H * W;
  float* yp = y + (i64)n * y_bs + (i64)c * H * W;
  float wt[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wt[t] = w[c * 9 + (flip ? 8 - t : t)];
  const int rows = min(DW_ROWS, H - y0);
  // stage the three row bands (zero-filled outside the image)
  for (int band = 0; band < 3; ++band) {
    const int dy = (band - 1) * dil;
    for (int i = threadIdx.x; i < rows * W; i += blockDim.x) {
      const int r = i / W, col = i - r * W;
      const int sy = y0 + r + dy;
      tile[(band * DW_ROWS + r) * W + col] = (sy >= 0 && sy < H) ? xp[(i64)sy * W + col] : 0.f;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < rows * W; i += blockDim.x) {
    const int r = i / W, col = i - r * W;
    float acc = 0.f;
#pragma unroll
    for (int ty = 0; ty < 3; ++ty) {
      const float* row = tile + (ty * DW_ROWS + r) * W;
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) {
        const int sx = col + (tx - 1) * dil;
        const float v = (sx >= 0 && sx < W) ? row[sx] : 0.f;
        acc = fmaf(wt[ty * 3 + tx], v, acc);
      }
    }
    const i64 o = (i64)(y0 + r) * W + col;
    yp[o] = accumulate ? yp[o] + acc : acc;
  }
}

// dw[c][t] += sum_{n,p} dy[n][c][p] * x[n][c][p + off(t)]
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_kernel(const float* __restrict__ x, i64 x_bs, const float* __restrict__ dy,
                                                              i64 dy_bs, float* __restrict__ dw, int C, int H, int W, int dil) {
  __shared__ float red[4][9];
  const int c = blockIdx.x, n = blockIdx.y;
  const float* xp = x + (i64)n * x_bs + (i64)c * H * W;
  const float* gp = dy + (i64)n * dy_bs + (i64)c * H * W;
  float acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = 0.f;
  for (int i = threadIdx.x; i < H * W; i += blockDim.x) {
    const int r = i / W, col = i - r * W;
    const float g = gp[i];
#pragma unroll
    for (int ty = 0; ty < 3; ++ty) {
      const int sy = r + (ty - 1) * dil;
      if (sy < 0 || sy >= H) continue;
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) {
        const int sx = col + (tx - 1) * dil;
        if (sx >= 0 && sx < W) acc[ty * 3 + tx] = fmaf(g, xp[(i64)sy * W + sx], acc[ty * 3 + tx]);
      }
    }
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const float s = wave_sum(acc[t]);
    if (lane == 0) red[wid][t] = s;
  }
  __syncthreads();
  if (threadIdx.x < 9) {
    const float s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    atomicAdd(&dw[c * 9 + threadIdx.x], s);
  }
}

}  // namespace

extern "C" int pfst_dwconv3x3(const float* x, long long x_bs, const float* w, float* y, long long y_bs,
                              int N, int C, int H, int W, int dil, int flip, int accumulate, pfst_stream_t stream) {
  PFST_CHECK_ARG(x && w && y && N > 0 && C > 0 && H > 0 && W > 0 && dil >= 1);
  PFST_CHECK_ARG(x_bs >= (i64)C * H * W && y_bs >= (i64)C * H * W && C <= 65535 && N <= 65535 && W <= 4096);
  const size_t lds = (size_t)3 * DW_ROWS * W * sizeof(float);
  PFST_CHECK_ARG(lds <= 160 * 1024);
  if (lds > 64 * 1024) {
    static bool set = false;
    if (!set) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv3x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      set = true;
    }
  }
  dim3 grid(cdiv(H, DW_ROWS), C, N);
  hipLaunchKernelGGL(dwconv3x3_kernel, grid, dim3(256), lds, (hipStream_t)stream, x, x_bs, w, y, y_bs, C, H, W, dil, flip, accumulate);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

extern "C" int pfst_dwconv3x3_wgrad(const float* x, long long x_bs, const float* dy, long long dy_bs, float* dw,
                                    int N, int C, int H, int W, int dil, pfst_stream_t stream) {
  PFST_CHECK_ARG(x && dy && dw && N > 0 && C > 0 && H > 0 && W > 0 && dil >= 1 && N <= 65535);
  dim3 grid(C, N);
  hipLaunchKernelGGL(dwconv3x3_wgrad_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, x_bs, dy, dy_bs, dw, C, H, W, dil);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}
