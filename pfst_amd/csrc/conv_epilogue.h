// Shared epilogue of the implicit-GEMM convolution kernels (fp32-MFMA, K-quad and bf16x6-split): optional fused BatchNorm
// statistics, bias, accumulate, store.  C/D layout of the 32x32 MFMA accumulator: column = lane & 31 (pixel),
// row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) (output channel).
#pragma once
#include "common.h"

typedef float pfst_f32x16 __attribute__((ext_vector_type(16)));

template <int TM, int TN, int WAVES_N, int BN>
__device__ __forceinline__ void conv_epilogue(const pfst_f32x16 (&acc)[TM][TN], float* __restrict__ out, const float* __restrict__ bias,
                                              float* __restrict__ stats, int stats_T, int accumulate, int M, int P, int m0, int p0,
                                              int wm0, int wn0, int bx, int n, int wid, int lane) {
  const int l31 = lane & 31, lh = lane >> 5;
  // Fused BatchNorm statistics: per-row (output channel) sum / sum of squares over this wave's pixels, reduced across
  // the 32 lanes of each half-wave with shuffles and written (no atomics) to stats[m][slot][2]; pfst_bn_finalize_partials
  // reduces the slots in fp64.  Saves the separate full-tensor read of bn_stats.
  if (stats) {
    const int gx = (P + BN - 1) / BN;
    const int slot = (n * gx + bx) * WAVES_N + (wid % WAVES_N);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float sv = 0.f, sq = 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int pp = p0 + wn0 + j * 32 + l31;
          const float v = pp < P ? acc[i][j][r] : 0.f;
          sv += v;
          sq = fmaf(v, v, sq);
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
          sv += __shfl_xor(sv, o, 64);
          sq += __shfl_xor(sq, o, 64);
        }
        const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (l31 == 0 && m < M) {
          float2* dst = reinterpret_cast<float2*>(stats) + ((i64)m * stats_T + slot);
          *dst = make_float2(sv, sq);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int pp = p0 + wn0 + j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < M && pp < P) {
          float v = acc[i][j][r];
          if (bias) v += bias[m];
          const i64 idx = (i64)m * P + pp;
          if (accumulate) v += out[idx];
          out[idx] = v;
        }
      }
    }
  }
}
