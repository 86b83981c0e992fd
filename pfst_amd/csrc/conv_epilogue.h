// Shared epilogue of the implicit-GEMM convolution kernels (fp32-MFMA, K-quad and bf16x6-split): optional fused BatchNorm
// statistics, bias, accumulate, store.  C/D layout of the 32x32 MFMA accumulator: column = lane & 31 (pixel),
// row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) (output channel).
#pragma once
#include "common.h"

typedef float pfst_f32x16 __attribute__((ext_vector_type(16)));

// Transposing butterfly over the 32 lanes of each half-wave: every lane enters with NV values (one per accumulator row) and
// leaves with ONE fully reduced value, lane l holding value index l & (NV-1).  Each step exchanges half of the remaining
// values with the partner lane l ^ mask and adds, so the whole reduction costs NV-1 exchanges instead of 5 NV.  The partner
// permutations are XORs the DPP unit (or one ds_swizzle) provides: 16 | 15 = row mirror | 7 = half-row mirror | 2, 1 = quad
// perms; they are independent over GF(2) and the kept half is chosen by the top bit of each mask, which the lower masks
// leave alone, so partners always hold the same value subset and every lane's data is counted exactly once.
template <int CTRL>
__device__ __forceinline__ float lane_xchg(float v) {
  if (CTRL == 0) return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));   // lane ^ 16
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int H, int CTRL>
__device__ __forceinline__ void butterfly_step(float* a, bool upper) {   // 2H values -> H values
#pragma unroll
  for (int k = 0; k < H; ++k) {
    const float send = upper ? a[k] : a[k + H];
    const float keep = upper ? a[k + H] : a[k];
    a[k] = keep + lane_xchg<CTRL>(send);
  }
}
template <int NV>
__device__ __forceinline__ float half_wave_transpose_sum(float (&a)[NV], int l31) {
  static_assert(NV == 32 || NV == 16, "one value per accumulator row of a 64- or 32-row wave tile");
  if (NV == 32) {
    butterfly_step<16, 0>(a, (l31 & 16) != 0);
  } else {
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] += lane_xchg<0>(a[k]);          // both 16-lane rows need all 16 values: plain add
  }
  butterfly_step<8, 0x140>(a, (l31 & 8) != 0);     // row_mirror      (lane ^ 15)
  butterfly_step<4, 0x141>(a, (l31 & 4) != 0);     // row_half_mirror (lane ^ 7)
  butterfly_step<2, 0x4E>(a, (l31 & 2) != 0);      // quad_perm [2,3,0,1]
  butterfly_step<1, 0xB1>(a, (l31 & 1) != 0);      // quad_perm [1,0,3,2]
  return a[0];
}

template <int TM, int TN, int WAVES_N, int BN>
__device__ __forceinline__ void conv_epilogue(const pfst_f32x16 (&acc)[TM][TN], float* __restrict__ out, const float* __restrict__ bias,
                                              float* __restrict__ stats, int stats_T, int accumulate, int M, int P, int m0, int p0,
                                              int wm0, int wn0, int bx, int n, int wid, int lane) {
  const int l31 = lane & 31, lh = lane >> 5;
  // Fused BatchNorm statistics: per-row (output channel) sum / sum of squares over this wave's pixels, reduced across
  // the 32 lanes of each half-wave (transposing butterfly) and written (no atomics) to stats[m][slot][2]; pfst_bn_finalize_partials
  // reduces the slots in fp64.  Saves the separate full-tensor read of bn_stats.
  if (stats) {
    const int gx = (P + BN - 1) / BN;
    const int slot = (n * gx + bx) * WAVES_N + (wid % WAVES_N);
    constexpr int NV = TM * 16;
    float sv[NV], sq[NV];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int pp = p0 + wn0 + j * 32 + l31;
          const float v = pp < P ? acc[i][j][r] : 0.f;
          a += v;
          b = fmaf(v, v, b);
        }
        sv[i * 16 + r] = a;
        sq[i * 16 + r] = b;
      }
    }
    const float ts = half_wave_transpose_sum<NV>(sv, l31);
    const float tq = half_wave_transpose_sum<NV>(sq, l31);
    // lane l31 now owns value index e = l31 & (NV-1) = i*16 + r of its half-wave: one 8-byte store per lane
    const int e = l31 & (NV - 1), r = e & 15;
    const int m = m0 + wm0 + (e >> 4) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    if (l31 < NV && m < M) {
      float2* dst = reinterpret_cast<float2*>(stats) + ((i64)m * stats_T + slot);
      *dst = make_float2(ts, tq);
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
    // NOTE for callers: wm0 must be provably wave-uniform (derive the wave id with __builtin_amdgcn_readfirstlane(tid >> 6));
    // otherwise every store below (and every accumulate load, each with its own vmcnt(0) wait) is wrapped in a waterfall
    // loop over the 'divergent' soffset -- measured 120-300 k cycles per workgroup.
    const int row0 = m0 + wm0;
    if (!accumulate) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = acc[i][j][r];       // (a float temporary: __builtin_bit_cast of the vector-element lvalue reads element 0)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc, voff[j],
                                                  4 * P * (row0 + i * 32 + (r & 3) + 8 * (r >> 2)), 0);
          }
        }
      return;
    }
    // accumulate: the 16 loads of a 32x32 block are issued back to back, then added and stored (one latency per block, not 64)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float old[16];
#pragma unroll
        for (int r = 0; r < 16; ++r)
          old[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff[j], 4 * P * (row0 + i * 32 + (r & 3) + 8 * (r >> 2)), 0));
#pragma unroll
        for (int r = 0; r < 16; ++r)
        {
          const float v = acc[i][j][r] + old[r];
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc, voff[j],
                                                4 * P * (row0 + i * 32 + (r & 3) + 8 * (r >> 2)), 0);
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
