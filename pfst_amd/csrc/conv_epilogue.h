// Shared epilogue of the implicit-GEMM convolution kernels (fp32-MFMA, K-quad and bf16x6-split): optional fused BatchNorm
// statistics, bias, accumulate, store.  C/D layout of the 32x32 MFMA accumulator: column = lane & 31 (pixel),
// row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) (output channel).
#pragma once
#include "common.h"

typedef float pfst_f32x16 __attribute__((ext_vector_type(16)));

// sum over each 32-lane half of the wave, result in every lane: four DPP-fused adds (quad swaps, half-row and row mirrors)
// and one ds_swizzle for the 16 <-> 16 exchange -- a tenth of the LDS traffic of five ds_bpermute shuffles.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float half_wave_sum(float v) {
  v = dpp_add<0xB1>(v);     // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);     // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);    // row_half_mirror
  v = dpp_add<0x140>(v);    // row_mirror
  return v + __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));   // lane ^ 16
}

template <int TM, int TN, int WAVES_N, int BN>
__device__ __forceinline__ void conv_epilogue(const pfst_f32x16 (&acc)[TM][TN], float* __restrict__ out, const float* __restrict__ bias,
                                              float* __restrict__ stats, int stats_T, int accumulate, int M, int P, int m0, int p0,
                                              int wm0, int wn0, int bx, int n, int wid, int lane) {
  const int l31 = lane & 31, lh = lane >> 5;
  // Fused BatchNorm statistics: per-row (output channel) sum / sum of squares over this wave's pixels, reduced across
  // the 32 lanes of each half-wave (half_wave_sum) and written (no atomics) to stats[m][slot][2]; pfst_bn_finalize_partials
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
        sv = half_wave_sum(sv);
        sq = half_wave_sum(sq);
        const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (l31 == 0 && m < M) {
          float2* dst = reinterpret_cast<float2*>(stats) + ((i64)m * stats_T + slot);
          *dst = make_float2(sv, sq);
        }
      }
    }
  }
  // Store.  fp32 MFMA and VALU share the vector pipe, so per-element address arithmetic and bounds checks in a 64-store
  // epilogue cost the co-resident waves real matrix throughput (measured: 10-17 % on tiles that live only 16-32 K-steps).
  // Fast path (all rows of this wave's sub-tile inside M, no bias): BUFFER stores -- one voffset per lane and column block
  // (pixel + the half-wave's 4-row shift; ragged pixel tiles use an out-of-range offset that the hardware drops), the row as
  // a SCALAR soffset: no vector arithmetic per element at all.
  if (!bias && m0 + wm0 + TM * 32 <= M) {
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, M * P * 4, 0x00020000);
    unsigned voff[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int pp = p0 + wn0 + j * 32 + l31;
      voff[j] = pp < P ? 4u * ((unsigned)pp + 4u * (unsigned)lh * (unsigned)P) : OOB;
    }
    const int row0 = m0 + wm0;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int soff = 4 * P * (row0 + i * 32 + (r & 3) + 8 * (r >> 2));     // wave-uniform
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          float v = acc[i][j][r];
          if (accumulate) v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff[j], soff, 0));
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc, voff[j], soff, 0);
        }
      }
    }
    return;
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
