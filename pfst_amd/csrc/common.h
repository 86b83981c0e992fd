// Shared device helpers for the pfst_hip kernels (gfx950 / CDNA4 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PFST_OK 0
#define PFST_ERR_ARG (-1)
#define PFST_ERR_LAUNCH (-2)
#define PFST_ERR_UNSUPPORTED (-3)

#define PFST_CHECK_ARG(cond)                                   \
  do {                                                         \
    if (!(cond)) {                                             \
      pfst_set_error(__FILE__, __LINE__, #cond);               \
      return PFST_ERR_ARG;                                     \
    }                                                          \
  } while (0)

#define PFST_CHECK_LAUNCH()                                    \
  do {                                                         \
    hipError_t e_ = hipGetLastError();                         \
    if (e_ != hipSuccess) {                                    \
      pfst_set_error(__FILE__, __LINE__, hipGetErrorString(e_)); \
      return PFST_ERR_LAUNCH;                                  \
    }                                                          \
  } while (0)

void pfst_set_error(const char* file, int line, const char* msg);
int pfst_deterministic(void);            // api.cpp: pfst_set_deterministic -- fixed-order sums instead of atomics between workgroups
void* pfst_det_scratch(size_t bytes, void* stream);   // api.cpp: that mode's partial-sum scratch of `stream` (NULL: allocation failed)

typedef long long i64;

static inline int cdiv(i64 a, i64 b) { return (int)((a + b - 1) / b); }

// Tile order of the implicit-GEMM kernels: workgroup `lin` of an image's gx x gy tiles -> (pixel tile bx, row tile by).  Workgroups are
// dealt round-robin over the 8 XCDs (each with its own L2), so for gx % 8 == 0
//  - the row tiles that share a pixel tile (= one activation tile) get ids 8 apart: same XCD, back to back, L2 hits for all but one;
//  - band != 0 (3x3 kernels): every XCD walks its OWN contiguous eighth of the pixel tiles, so that the tiles of adjacent image rows,
//    which re-read each other's rows through the taps, meet in one L2.  Dealt round-robin, vertical neighbours sit Wo / 128 ids apart,
//    i.e. on different XCDs: 2.6x the activation bytes came from beyond L2 on the 32-channel stem layers (HBM-bound at 4.7 TB/s).
#ifdef __HIPCC__
__device__ __forceinline__ void pfst_tile_order(int lin, int gx, int gy, int band, int& bx, int& by) {
  if ((gx & 7) == 0) {
    if (band) {
      const int j = lin >> 3;
      const int idx = j / gy;
      by = j - idx * gy;
      bx = (lin & 7) * (gx >> 3) + idx;
    } else {
      const int grp = lin / (8 * gy), r = lin - grp * 8 * gy;
      by = r >> 3;
      bx = grp * 8 + (r & 7);
    }
  } else {
    by = lin / gx;
    bx = lin - by * gx;
  }
}
#endif

// grid size for a grid-stride elementwise kernel: cap at 256 CUs x 8 blocks
static inline int ew_grid(i64 n, int block = 256) {
  i64 g = (n + block - 1) / block;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide sum of a double; result valid in thread 0.  blockDim.x must be a multiple of 64, <= 1024.
__device__ __forceinline__ double block_sum_d(double v, double* smem /* >= 16 doubles */) {
  v = wave_sum_d(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) smem[wid] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int i = 0; i < nw; ++i) r += smem[i];
  }
  return r;
}

// two block-wide sums at once (the (sum, sum of squares) pairs of the fused BatchNorm statistics): one pair of barriers instead of two;
// results valid in thread 0.  smem >= 32 doubles.
// WAVE_F32: the 64 lane values are added in fp32 (DPP), the per-wave results in fp64 -- for per-thread partials that are fp32 anyway
template <bool WAVE_F32 = false>
__device__ __forceinline__ void block_sum2_d(double& a, double& b, double* smem) {
  if (WAVE_F32) {
    a = (double)wave_sum((float)a);
    b = (double)wave_sum((float)b);
  } else {
    a = wave_sum_d(a);
    b = wave_sum_d(b);
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) { smem[wid] = a; smem[16 + wid] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    double ra = 0.0, rb = 0.0;
    for (int i = 0; i < nw; ++i) { ra += smem[i]; rb += smem[16 + i]; }
    a = ra; b = rb;
  }
}

// torch's area_pixel_compute_source_index for bilinear, align_corners=False (ATen/native/UpSample.h):
// src = max(0, scale*(dst+0.5)-0.5).  The arithmetic is PINNED (explicit fma / rn intrinsics, immune to -ffp-contract) to
// what torch's kernels evaluate, determined bit-for-bit against F.interpolate (tests/test_hip_ops.py): the source index is one
// fma, the four-tap blend is t = fma(lx0, v00, lx1*v01); out = fma(ly0, t0, ly1*t1).  The pseudo-label index map depends
// on these roundings wherever two classes tie to within an ulp.
__device__ __forceinline__ void bilin_src(int dst, float scale, int in_size, int& i0, int& i1, float& l0, float& l1) {
  float s = __fmaf_rn(scale, __fadd_rn((float)dst, 0.5f), -0.5f);
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = __fsub_rn(s, (float)i0);
  l0 = __fsub_rn(1.f, l1);
}
// scale / shift of the normalisation, ONE definition for bn_apply, bn_backward and the fused data-gradient epilogue: the ReLU gate
// recomputed in backward must be bit-identical to the forward decision, so the roundings are pinned
__device__ __forceinline__ void bn_affine(float mean, float invstd, float gamma, float beta, float& sc, float& sh) {
  sc = __fmul_rn(invstd, gamma);
  sh = __fmaf_rn(-mean, sc, beta);
}
__device__ __forceinline__ float bilin_blend(float v00, float v01, float v10, float v11, float lx0, float lx1, float ly0, float ly1) {
  const float t0 = __fmaf_rn(lx0, v00, __fmul_rn(lx1, v01));
  const float t1 = __fmaf_rn(lx0, v10, __fmul_rn(lx1, v11));
  return __fmaf_rn(ly0, t0, __fmul_rn(ly1, t1));
}

// internal (not part of the C ABI): K-quad weight-gradient fast path, conv_wgrad_q.hip
bool pfst_wgrad_q_eligible(const float* x, i64 x_bs, const float* dy, i64 dy_bs, int Hi, int Wi, int Ho, int Wo, int ksize, int stride, int dil);
int pfst_wgrad_q_launch(const float* x, i64 x_bs, const float* dy, i64 dy_bs, float* dw, int N, int Cin, int Hi, int Wi, int Cout,
                        int Ho, int Wo, int ksize, int dil, int pad, int groups, i64 x_gs, i64 dy_gs, i64 dw_gs, hipStream_t s);
// internal: bf16x6-split K-quad weight gradient of 1x1 / grouped transform-domain products, conv_split.hip
bool pfst_wgrad_split_q_eligible(const float* x, i64 x_bs, const float* dy, i64 dy_bs, int Hi, int Wi, int Ho, int Wo, int ksize, int stride);
int pfst_wgrad_split_q_launch(const float* x, i64 x_bs, const float* dy, i64 dy_bs, float* dw, int N, int Cin, int Cout, int P, int groups,
                              i64 x_gs, i64 dy_gs, i64 dw_gs, hipStream_t s);
// internal: f16x3 weight gradient of 1x1 / grouped transform-domain products (P % 4 == 0, Cout > 64), conv_f16x3.hip
int pfst_wgrad_f16x3_launch(const float* x, i64 x_bs, const float* dy, i64 dy_bs, float* dw, int N, int J, int M, int P, int groups,
                            i64 x_gs, i64 dy_gs, i64 dw_gs, const float* x_amax, const float* dy_amax, int packed, hipStream_t s, const float* bnl = nullptr);
// internal: K-quad implicit-GEMM convolution (Cin % 16 == 0), conv_igemm_q.hip
struct pfst_bnb_fuse;
typedef struct pfst_bnb_fuse PfstBnbArgs;   // include/pfst_hip.h: fused BatchNorm-backward sums of a data-gradient launch (null = off)
int pfst_igemm_q_launch(const float* in, i64 in_bs, const float* wq, const float* bias, float* out, i64 out_bs, int N, int C, int Hi,
                        int Wi, int M, int Ho, int Wo, int ks, int a, int b, int c, int d, int acc, float* stats, int stats_T, int groups,
                        hipStream_t s, const PfstBnbArgs* bnb = nullptr);
