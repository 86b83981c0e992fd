// fp32-faithful convolution on the bf16 matrix cores: every fp32 operand is split EXACTLY into three bf16 pieces
// x = x0 + x1 + x2 (8 + 8 + 8 mantissa bits) and the product is rebuilt from the six piece-products of weight >= 2^-16
//   a*b ~= a0b0 + (a0b1 + a1b0) + (a0b2 + a1b1 + a2b0)          (dropped terms <= 2^-24 relative: fp32's own ulp)
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  Six bf16 MFMAs replace sixteen fp32 MFMA-equivalents: the
// fp32-input MFMA runs at 1/16 of the bf16 rate on gfx950 (no xf32/TF32), so this is 2.67x the arithmetic throughput
// at fp32 accuracy.  Measured on the oracle: the 3-term "bf16x3" variant is NOT enough -- logits off by
// 6e-3, random-init gradients by 16 % -- while the 6-term split is at least as close to fp64 as fp32 arithmetic.
//
// Same implicit-GEMM structure, tiling, buffer-load addressing and epilogue as conv_mfma.hip (fprop + dgrad; layers
// whose Cin is not a multiple of 16 stay on the fp32-MFMA kernel).  Weights are pre-split once per step by
// pfst_conv_pack_weight_split into the exact LDS image [k16-group][piece][k-half][row][8 x bf16]; activations are split
// in registers on their way from HBM to LDS (fp32 NCHW stays the storage format everywhere).
#include "conv_epilogue.h"
#include "det.h"
#include <math.h>
#include <type_traits>
#include <utility>
#include <stdlib.h>
#include "../../include/pfst_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int BN = 128;
constexpr int NP = 3;  // pieces

__device__ __forceinline__ bool src_coord(int o, int t, int a, int b, int c0, int div, int lim, int& s) {
  const int v = o * a + t * b + c0;
  const int odd = v & (div - 1);
  s = v >> (div >> 1);
  return (odd == 0) & (s >= 0) & (s < lim);
}

__device__ __forceinline__ unsigned short bf16_bits(float x) { return __builtin_bit_cast(unsigned short, (__bf16)x); }
__device__ __forceinline__ float bf16_to_f32(unsigned short b) { return __builtin_bit_cast(float, (unsigned)b << 16); }

// exact 3-way split of 8 fp32 values into three packed bf16x8 vectors
__device__ __forceinline__ void split8(const float (&v)[8], uint4& p0, uint4& p1, uint4& p2) {
  unsigned short h0[8], h1[8], h2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    h0[i] = bf16_bits(v[i]);
    const float r1 = v[i] - bf16_to_f32(h0[i]);
    h1[i] = bf16_bits(r1);
    const float r2 = r1 - bf16_to_f32(h1[i]);
    h2[i] = bf16_bits(r2);
  }
  p0 = make_uint4(h0[0] | (unsigned)h0[1] << 16, h0[2] | (unsigned)h0[3] << 16, h0[4] | (unsigned)h0[5] << 16, h0[6] | (unsigned)h0[7] << 16);
  p1 = make_uint4(h1[0] | (unsigned)h1[1] << 16, h1[2] | (unsigned)h1[3] << 16, h1[4] | (unsigned)h1[5] << 16, h1[6] | (unsigned)h1[7] << 16);
  p2 = make_uint4(h2[0] | (unsigned)h2[1] << 16, h2[2] | (unsigned)h2[3] << 16, h2[4] | (unsigned)h2[5] << 16, h2[6] | (unsigned)h2[7] << 16);
}

// BNB != 0: the data-gradient launch also emits the BatchNorm-backward sums of the layer that owns `out` (conv_epilogue.h); on the bf16
// matrix pipe the epilogue's VALU work co-issues with the other waves' MFMAs
// DIAG != 0: the timing-ablation instantiations of round 2 (profiles/r02_split_phase_stamps.txt; results WRONG by construction).  No entry point
// launches them any more (the PFST_SPLIT_DIAG dispatch and tools/split_ablation.py were removed in round 5); instantiate by hand to repeat it:
// 1 no split + LDS store, 2 no global loads, 3 no MFMAs, 4 no LDS fragment reads, 5 no in-loop barrier,
// 6 s_memtime stamps around the four phases of a K-step, summed per wave into the `stats` buffer as 6 (of 8) x u32 (not an output value)
template <int BM, int BNB = 0, int DIAG = 0>
__global__ __launch_bounds__(256) void conv_igemm_split_kernel(
    const float* __restrict__ in, i64 in_bs, const uint4* __restrict__ wk6, const float* __restrict__ bias,
    float* __restrict__ out, i64 out_bs, int C, int Hi, int Wi, int M, int Ho, int Wo, int ks,
    int ca, int cb, int cc, int cdivv, int accumulate, float* __restrict__ stats, int stats_T, PfstBnbArgs bnb) {
  constexpr int WM = BM >= 64 ? 64 : 32;
  constexpr int WAVES_M = BM / WM;
  constexpr int WAVES_N = 4 / WAVES_M;
  constexpr int WN = BN / WAVES_N;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int A_CHUNKS = 2 * NP * BM;                 // 16-byte chunks of the A tile: [piece][half][row]
  constexpr int A_N = (A_CHUNKS + 255) / 256;

  __shared__ uint4 As[2][2 * NP * BM];
  __shared__ uint4 Bs[2][2 * NP * BN];

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave id in an SGPR
  const int wm0 = (wid / WAVES_N) * WM, wn0 = (wid % WAVES_N) * WN;
  const int P = Ho * Wo, HiWi = Hi * Wi;
  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so the m-tiles that share one pixel tile
  // (= one activation tile) are given ids 8 apart: they run back to back on the SAME XCD and hit its L2.
  int bx, by;
  {
    const int gx = (P + BN - 1) / BN, gy = (M + BM - 1) / BM;
    const int lin = blockIdx.x;
    pfst_tile_order(lin, gx, gy, ks == 3, bx, by);
  }
  const int p0 = bx * BN, m0 = by * BM, n = blockIdx.y * gridDim.z + blockIdx.z;   // gridDim.y > 1: batched GEMMs (Winograd), group = blockIdx.y
  const int K = C * ks * ks;
  const int KT = K / 16;
  in += (i64)n * in_bs;
  out += (i64)n * out_bs;

  const int pix = tid & (BN - 1), kh = tid >> 7;          // B staging: this thread's pixel and k-half
  const int p = p0 + pix;
  const bool pvalid = p < P;
  const int oy = pvalid ? p / Wo : 0;
  const int ox = pvalid ? p - oy * Wo : 0;

  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(wk6) + (i64)blockIdx.y * KT * 2 * NP * M, 0,
                                                                         KT * 2 * NP * M * 16, 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, C * HiWi * 4, 0x00020000);
  unsigned a_voff[A_N], b_voff[8];
#pragma unroll
  for (int i = 0; i < A_N; ++i) {
    const int c = tid + 256 * i;                           // chunk -> (segment = piece*2+half, row)
    const int seg = c / BM, row = c - seg * BM;
    a_voff[i] = (c < A_CHUNKS && m0 + row < M) ? 16u * ((unsigned)seg * (unsigned)M + (unsigned)(m0 + row)) : OOB;
  }
  int ld_ty = 0, ld_tx = 0, ld_ci0 = 0;
  auto set_tap = [&]() {
    int sy, sx;
    const bool ok = pvalid & src_coord(oy, ld_ty, ca, cb, cc, cdivv, Hi, sy) & src_coord(ox, ld_tx, ca, cb, cc, cdivv, Wi, sx);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      b_voff[i] = ok ? 4u * ((unsigned)(kh * 8 + i) * (unsigned)HiWi + (unsigned)(sy * Wi + sx)) : OOB;
  };
  set_tap();

  uint4 areg[A_N];
  float breg[8];
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto load_tile = [&](int kt) {
    const int a_soff = kt * 2 * NP * M * 16, b_soff = ld_ci0 * HiWi * 4;
#pragma unroll
    for (int i = 0; i < A_N; ++i)
      areg[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, a_voff[i], a_soff, 0));
#pragma unroll
    for (int i = 0; i < 8; ++i)
      breg[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b_rsrc, b_voff[i], b_soff, 0));
    ld_ci0 += 16;
    if (ld_ci0 >= C) {
      ld_ci0 = 0; ld_tx += 1;
      if (ld_tx == ks) { ld_tx = 0; ld_ty += 1; }
      set_tap();
    }
  };
  uint4 bq0, bq1, bq2;                               // the split pieces of this thread's 8 activations of the NEXT K-step
  auto split_tile = [&]() { split8(breg, bq0, bq1, bq2); };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_N; ++i) {
      const int c = tid + 256 * i;
      if (c < A_CHUNKS) As[buf][c] = areg[i];
    }
    Bs[buf][(0 * 2 + kh) * BN + pix] = bq0;
    Bs[buf][(1 * 2 + kh) * BN + pix] = bq1;
    Bs[buf][(2 * 2 + kh) * BN + pix] = bq2;
  };

  load_tile(0);
  split_tile();
  store_tile(0);
  __syncthreads();

  const int l31 = lane & 31, lh = lane >> 5;
  unsigned long long tprev = 0;
  unsigned ph[6] = {0, 0, 0, 0, 0, 0};
  if (DIAG == 6) tprev = __builtin_amdgcn_s_memtime();
  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    if (DIAG != 2) {
      if (kt + 1 < KT) load_tile(kt + 1);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(breg[i]));
#pragma unroll
      for (int i = 0; i < A_N; ++i) asm volatile("" : "+v"(areg[i].x), "+v"(areg[i].y), "+v"(areg[i].z), "+v"(areg[i].w));
    }
    if (DIAG == 6) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      ph[4] += (unsigned)(t - tprev);          // global loads issued
      tprev = t;
      __builtin_amdgcn_sched_barrier(0);
    }
    bf16x8 af[TM][NP], bf[TN][NP];
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
      if (DIAG != 4) {
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i][pl] = __builtin_bit_cast(bf16x8, As[cur][(pl * 2 + lh) * BM + wm0 + i * 32 + l31]);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j][pl] = __builtin_bit_cast(bf16x8, Bs[cur][(pl * 2 + lh) * BN + wn0 + j * 32 + l31]);
      } else {
#pragma unroll
        for (int i = 0; i < TM; ++i) { bf16x8 t = __builtin_bit_cast(bf16x8, areg[0]); asm volatile("" : "+v"(t)); af[i][pl] = t; }
#pragma unroll
        for (int j = 0; j < TN; ++j) { bf16x8 t = __builtin_bit_cast(bf16x8, areg[0]); asm volatile("" : "+v"(t)); bf[j][pl] = t; }
      }
    }
    // smallest terms first: (2,0) (1,1) (0,2) | (1,0) (0,1) | (0,0).  This loop (now used for contractions below 512 and the 64- / 32-row
    // tiles) leaves the order of everything else to the compiler: all MFMAs back to back, then the split and the LDS stores.  Its matrix
    // pipe is 0.54-0.58 busy; the phase stamps (DIAG 6) show why, and conv_igemm_split_pipe_body / _pair_body below are the answer.
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0};
    constexpr int PB[6] = {0, 1, 2, 0, 1, 0};
    if (DIAG == 6) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      ph[0] += (unsigned)(t - tprev);
      tprev = t;
      __builtin_amdgcn_sched_barrier(0);
    }
    if (DIAG != 3) {
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PA[t]], bf[j][PB[t]], acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(af[i][pl]));
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(bf[j][pl]));
      }
    }
    if (DIAG == 6) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      ph[1] += (unsigned)(t - tprev);
      tprev = t;
      __builtin_amdgcn_sched_barrier(0);
    }
    if (DIAG != 1) {
      if (DIAG == 6) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        ph[5] += (unsigned)(t - tprev);          // global loads landed
        tprev = t;
        __builtin_amdgcn_sched_barrier(0);
      }
      if (kt + 1 < KT) {
        split_tile();
        store_tile(cur ^ 1);
      }
      if (DIAG == 6) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        ph[2] += (unsigned)(t - tprev);
        tprev = t;
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(breg[i]));
#pragma unroll
      for (int i = 0; i < A_N; ++i) asm volatile("" ::"v"(areg[i].x), "v"(areg[i].y), "v"(areg[i].z), "v"(areg[i].w));
    }
    if (DIAG != 5) __syncthreads();
    if (DIAG == 6) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      ph[3] += (unsigned)(t - tprev);
      tprev = t;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (DIAG == 6) {
    if (lane == 0 && stats) {
      unsigned* dbg = reinterpret_cast<unsigned*>(stats) + ((((i64)n * gridDim.x + blockIdx.x) * 4 + wid) * 8);
      dbg[0] = ph[0]; dbg[1] = ph[1]; dbg[2] = ph[2]; dbg[3] = ph[3]; dbg[4] = ph[4]; dbg[5] = ph[5];
    }
    stats = nullptr;
  }

  if (BNB != 0) {
    // (the main loop ended with a workgroup barrier: Bs is free) the per-row sums are reduced through 20 KB of the B double buffer
    static_assert(sizeof(Bs) >= 4 * PFST_ROWSUM_LDS_FLOATS * sizeof(float), "epilogue scratch must fit into the B double buffer");
    conv_epilogue<TM, TN, WAVES_N, BN, BNB, true>(acc, out, bias, stats, stats_T, accumulate, M, P, m0, p0, wm0, wn0, bx, n, wid, lane, bnb,
                                                  reinterpret_cast<float*>(&Bs[0][0]));
  } else {
    conv_epilogue<TM, TN, WAVES_N, BN>(acc, out, bias, stats, stats_T, accumulate, M, P, m0, p0, wm0, wn0, bx, n, wid, lane);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same 128 x 128 tile with a software-pipelined K-step.  In-kernel stamps on the kernel above (tools/split_ablation.py, DIAG 6,
// three waves per SIMD) per wave and K-step: 581 cycles to ISSUE the 11 global loads, 228 until the fragments landed, 1080 for the 24
// MFMAs (768 alone), 10 waiting for the global loads, 1363 for the 44-instruction split + 6 LDS stores, 232 at the barrier.  Beside
// ANOTHER wave's back-to-back MFMAs a wave gets roughly one vector instruction per MFMA slot issued; inside its OWN MFMA stream a wave
// hides up to five (MI355X guide, 'single-issue instructions hidden per MFMA gap').  So every wave here carries its own loads,
// fragment reads, split arithmetic and LDS stores in the gaps of its own 24 MFMAs (sched_group_barrier), and the K loop is cut into
// branch-free blocks: the tap switch happens between them, not inside.
template <int... Is, class F>
__device__ __forceinline__ void static_for_seq(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>()), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {       // f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>), always unrolled
  static_for_seq(std::make_integer_sequence<int, N>(), static_cast<F&&>(f));
}

// The in-register split as 44 single VALU instructions pinned in place (volatile asm): plain arithmetic is sunk to the LDS stores by the
// compiler and packed into v_pk_add_f32, which is slow beside MFMAs.  K = 0..43 on the eight values v[0..8): two pairs in lockstep
// (independent neighbours), 11 ops per pair -- stage 0/1 {cvt_pk, << 16, & 0xffff0000, sub, sub}, stage 2 {cvt_pk}.
struct SplitState {
  unsigned hp[NP][4];            // packed bf16 pairs of the three pieces
  unsigned tl[4], th[4];         // a piece widened back to fp32 (low / high element)
  float ra[4], rb[4];            // running remainders of the pairs
};
template <int K>
__device__ __forceinline__ void split_op(const float (&v)[8], SplitState& s) {
  constexpr int half = K / 22, r = K % 22, q = 2 * half + (r & 1), o = r >> 1;    // o = 0..10
  constexpr int st = o / 5, op = o % 5;
  if constexpr (op == 0) {
    if constexpr (st == 0) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(s.hp[0][q]) : "v"(v[2 * q]), "v"(v[2 * q + 1]));
    else asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(s.hp[st][q]) : "v"(s.ra[q]), "v"(s.rb[q]));
  } else if constexpr (op == 1) {
    asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(s.tl[q]) : "v"(s.hp[st][q]));
  } else if constexpr (op == 2) {
    asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(s.th[q]) : "v"(s.hp[st][q]));
  } else if constexpr (op == 3) {
    if constexpr (st == 0) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(s.ra[q]) : "v"(v[2 * q]), "v"(s.tl[q]));
    else asm volatile("v_sub_f32 %0, %1, %2" : "=v"(s.ra[q]) : "v"(s.ra[q]), "v"(s.tl[q]));
  } else {
    if constexpr (st == 0) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(s.rb[q]) : "v"(v[2 * q + 1]), "v"(s.th[q]));
    else asm volatile("v_sub_f32 %0, %1, %2" : "=v"(s.rb[q]) : "v"(s.rb[q]), "v"(s.th[q]));
  }
}
__device__ __forceinline__ uint4 split_piece(const SplitState& s, int pl) { return make_uint4(s.hp[pl][0], s.hp[pl][1], s.hp[pl][2], s.hp[pl][3]); }
// SHAPE16: timing experiment only (WRONG results): every 32x32x16 MFMA replaced by two 16x16x32 MFMAs on the same operand registers,
// to see which clock the chip holds for that shape (MI355X guide, DVFS give-back item 7)
template <int BNB, bool SHAPE16 = false>
__device__ __forceinline__ void conv_igemm_split_pipe_body(
    const float* __restrict__ in, i64 in_bs, const uint4* __restrict__ wk6, const float* __restrict__ bias,
    float* __restrict__ out, i64 out_bs, int C, int Hi, int Wi, int M, int Ho, int Wo, int ks,
    int ca, int cb, int cc, int cdivv, int accumulate, float* __restrict__ stats, int stats_T, const PfstBnbArgs& bnb) {
  constexpr int BM = 128, WM = 64, WAVES_N = 2, WN = 64, TM = 2, TN = 2;
  constexpr int A_N = 2 * NP * BM / 256;                 // 3 chunks of 16 bytes per thread: [piece][half][row]
  __shared__ uint4 As[2][2 * NP * BM];
  __shared__ uint4 Bs[2][2 * NP * BN];

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wid / WAVES_N) * WM, wn0 = (wid % WAVES_N) * WN;
  const int P = Ho * Wo, HiWi = Hi * Wi;
  int bx, by;
  {
    const int gx = (P + BN - 1) / BN, gy = (M + BM - 1) / BM;
    const int lin = blockIdx.x;
    pfst_tile_order(lin, gx, gy, ks == 3, bx, by);
  }
  const int p0 = bx * BN, m0 = by * BM, n = blockIdx.y * gridDim.z + blockIdx.z;
  const int spt = C / 16;                                // K-steps per filter tap
  const int KT = spt * ks * ks;
  in += (i64)n * in_bs;
  out += (i64)n * out_bs;

  const int pix = tid & (BN - 1), kh = tid >> 7;
  const int p = p0 + pix;
  const bool pvalid = p < P;
  const int oy = pvalid ? p / Wo : 0;
  const int ox = pvalid ? p - oy * Wo : 0;

  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(wk6) + (i64)blockIdx.y * KT * 2 * NP * M, 0,
                                                                         KT * 2 * NP * M * 16, 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, C * HiWi * 4, 0x00020000);
  // one vector offset per operand; chunk i of the weight image / channel i of the activations is a SCALAR offset on top (an
  // out-of-range base stays out of range: the scalar parts are far below 2 GiB)
  unsigned a_voff, b_voff;
  {
    const int seg = tid / BM, row = tid - seg * BM;        // chunk c = tid + 256 i -> segment seg + 2 i, same row
    a_voff = (m0 + row < M) ? 16u * ((unsigned)seg * (unsigned)M + (unsigned)(m0 + row)) : OOB;
  }
  const int a_chunk = 2 * M * 16, b_chan = HiWi * 4;
  auto set_tap = [&](int ty, int tx) {
    int sy, sx;
    const bool ok = pvalid & src_coord(oy, ty, ca, cb, cc, cdivv, Hi, sy) & src_coord(ox, tx, ca, cb, cc, cdivv, Wi, sx);
    b_voff = ok ? 4u * ((unsigned)(kh * 8) * (unsigned)HiWi + (unsigned)(sy * Wi + sx)) : OOB;
  };

  uint4 areg[A_N];
  float breg[8];
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int a_step = 2 * NP * M * 16;                    // bytes of one K-step of the weight image
  int a_soff = 0;                                        // of the tile the NEXT load fetches
  auto load_tile = [&](int b_soff) {
#pragma unroll
    for (int i = 0; i < A_N; ++i)
      areg[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, a_voff, a_soff + i * a_chunk, 0));
#pragma unroll
    for (int i = 0; i < 8; ++i)
      breg[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b_rsrc, b_voff, b_soff + i * b_chan, 0));
    a_soff += a_step;
  };
  auto split_store = [&](int buf) {
    uint4 bq0, bq1, bq2;
    split8(breg, bq0, bq1, bq2);
#pragma unroll
    for (int i = 0; i < A_N; ++i) As[buf][tid + 256 * i] = areg[i];
    Bs[buf][(0 * 2 + kh) * BN + pix] = bq0;
    Bs[buf][(1 * 2 + kh) * BN + pix] = bq1;
    Bs[buf][(2 * 2 + kh) * BN + pix] = bq2;
  };

  set_tap(0, 0);
  load_tile(0);
  split_store(0);
  __syncthreads();

  const int l31 = lane & 31, lh = lane >> 5;
  int cur = 0;
  // one K-step: consumes the tile in LDS buffer `cur`; LOAD: also fetches, splits and stores the next tile into the other buffer.
  // The issue order is fixed slot by slot (a full scheduling barrier after each): 4 fragment reads, then 24 slots of one MFMA + fillers --
  //   slots 0-7   one fragment read (in the order the terms need them) + one activation load
  //   slots 8-10  one weight-image load
  //   slots 11-21 four instructions of the 44-instruction split of the eight activations (split_op)
  //   slots 22-23 the three weight-image and the three activation LDS stores (the matrix pipe is still busy with the last MFMAs)
  auto step = [&](auto load_tag, int b_soff) {
    constexpr bool LOAD = decltype(load_tag)::value;
    bf16x8 af[TM][NP], bf[TN][NP];
    auto read_frag = [&](int r) {       // r = 0..11: (a2, b0) | (a1, b1) | (a0, b2), rows i / columns j inside
      const int q = r >> 2, e = r & 3;
      if (e < 2) af[e][2 - q] = __builtin_bit_cast(bf16x8, As[cur][((2 - q) * 2 + lh) * BM + wm0 + e * 32 + l31]);
      else bf[e - 2][q] = __builtin_bit_cast(bf16x8, Bs[cur][(q * 2 + lh) * BN + wn0 + (e - 2) * 32 + l31]);
    };
    static_for<4>([&](auto rc) { read_frag(decltype(rc)::value); });
    __builtin_amdgcn_sched_barrier(0);
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0};            // smallest terms first
    constexpr int PB[6] = {0, 1, 2, 0, 1, 0};
    SplitState sp;
    static_for<24>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      constexpr int t = m >> 2, i = (m >> 1) & 1, j = m & 1;
      if constexpr (SHAPE16) {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        f32x4 lo = __builtin_shufflevector(acc[i][j], acc[i][j], 0, 1, 2, 3), hi = __builtin_shufflevector(acc[i][j], acc[i][j], 4, 5, 6, 7);
        lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][PA[t]], bf[j][PB[t]], lo, 0, 0, 0);
        hi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][PA[t]], bf[j][PB[t]], hi, 0, 0, 0);
        acc[i][j][0] = lo[0]; acc[i][j][1] = lo[1]; acc[i][j][2] = lo[2]; acc[i][j][3] = lo[3];
        acc[i][j][4] = hi[0]; acc[i][j][5] = hi[1]; acc[i][j][6] = hi[2]; acc[i][j][7] = hi[3];
      } else {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PA[t]], bf[j][PB[t]], acc[i][j], 0, 0, 0);
      }
      if constexpr (m < 8) read_frag(4 + m);
      if constexpr (LOAD) {
        if constexpr (m < 8) breg[m] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b_rsrc, b_voff, b_soff + m * b_chan, 0));
        else if constexpr (m < 11)
          areg[m - 8] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, a_voff, a_soff + (m - 8) * a_chunk, 0));
        else if constexpr (m < 22) {
          static_for<4>([&](auto kc) { split_op<(m - 11) * 4 + decltype(kc)::value>(breg, sp); });
        } else if constexpr (m == 22) {
          As[cur ^ 1][tid] = areg[0];
          As[cur ^ 1][tid + 256] = areg[1];
        } else {
          As[cur ^ 1][tid + 512] = areg[2];
#pragma unroll
          for (int pl = 0; pl < NP; ++pl) Bs[cur ^ 1][(pl * 2 + kh) * BN + pix] = split_piece(sp, pl);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    if (LOAD) a_soff += a_step;
    __syncthreads();
    cur ^= 1;
  };
  const int chan_step = 16 * HiWi * 4;
  for (int ty = 0; ty < ks; ++ty)
    for (int tx = 0; tx < ks; ++tx) {
      for (int sidx = 1; sidx < spt; ++sidx) step(std::true_type(), sidx * chan_step);
      // the last K-step of this tap fetches the first tile of the next one
      int nty = ty, ntx = tx + 1;
      if (ntx == ks) { ntx = 0; nty += 1; }
      if (nty < ks) {
        set_tap(nty, ntx);
        step(std::true_type(), 0);
      } else {
        step(std::false_type(), 0);
      }
    }

  if (BNB != 0) {
    static_assert(sizeof(Bs) >= 4 * PFST_ROWSUM_LDS_FLOATS * sizeof(float), "epilogue scratch must fit into the B double buffer");
    conv_epilogue<TM, TN, WAVES_N, BN, BNB, true>(acc, out, bias, stats, stats_T, accumulate, M, P, m0, p0, wm0, wn0, bx, n, wid, lane, bnb,
                                                  reinterpret_cast<float*>(&Bs[0][0]));
  } else {
    conv_epilogue<TM, TN, WAVES_N, BN>(acc, out, bias, stats, stats_T, accumulate, M, P, m0, p0, wm0, wn0, bx, n, wid, lane);
  }
}

// the plain variant is held to 168 registers (three workgroups per CU, what its 48 KB of LDS allow); the fused-BatchNorm-backward
// epilogues need more and run two per CU
__global__ __launch_bounds__(256, 3) void conv_igemm_split_pipe_kernel(
    const float* __restrict__ in, i64 in_bs, const uint4* __restrict__ wk6, const float* __restrict__ bias,
    float* __restrict__ out, i64 out_bs, int C, int Hi, int Wi, int M, int Ho, int Wo, int ks,
    int ca, int cb, int cc, int cdivv, int accumulate, float* __restrict__ stats, int stats_T) {
  conv_igemm_split_pipe_body<0>(in, in_bs, wk6, bias, out, out_bs, C, Hi, Wi, M, Ho, Wo, ks, ca, cb, cc, cdivv, accumulate, stats, stats_T,
                                PfstBnbArgs());
}
template <int BNB>
__global__ __launch_bounds__(256, 2) void conv_igemm_split_pipe_bnb_kernel(
    const float* __restrict__ in, i64 in_bs, const uint4* __restrict__ wk6, const float* __restrict__ bias,
    float* __restrict__ out, i64 out_bs, int C, int Hi, int Wi, int M, int Ho, int Wo, int ks,
    int ca, int cb, int cc, int cdivv, int accumulate, float* __restrict__ stats, int stats_T, PfstBnbArgs bnb) {
  conv_igemm_split_pipe_body<BNB>(in, in_bs, wk6, bias, out, out_bs, C, Hi, Wi, M, Ho, Wo, ks, ca, cb, cc, cdivv, accumulate, stats, stats_T,
                                  bnb);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The pipelined loop on v_mfma_f32_16x16x32_bf16 (the shape that holds the higher clock): the two K=16 LDS tiles are ONE
// K=32 step -- k-quarter q of the MFMA operand is (tile q / 2, half q % 2) of the unchanged [piece][half][row] images.  All 24 fragments of
// a step live in registers (96), so the LDS image is free once they have landed: barrier A sits in the middle of the MFMA stream, the
// next pair of tiles is loaded, split and stored in the gaps of the same 96 MFMAs, barrier B ends the step.  Two 48 KB workgroups per CU
// (~240 registers).  The accumulators leave in the 16x16 layout and are re-laid through LDS, 32 rows per wave at a time, for conv_epilogue.
template <int BNB>
__device__ __forceinline__ void conv_igemm_split_pair_body(
    const float* __restrict__ in, i64 in_bs, const uint4* __restrict__ wk6, const float* __restrict__ bias,
    float* __restrict__ out, i64 out_bs, int C, int Hi, int Wi, int M, int Ho, int Wo, int ks,
    int ca, int cb, int cc, int cdivv, int accumulate, float* __restrict__ stats, int stats_T, const PfstBnbArgs& bnb) {
  constexpr int BM = 128, WAVES_N = 2;
  constexpr int TILE_A = 2 * NP * BM, TILE_B = 2 * NP * BN;     // 16-byte chunks of one K=16 tile
  __shared__ uint4 smem[2 * TILE_A + 2 * TILE_B];
  uint4* const As = smem;
  uint4* const Bs = smem + 2 * TILE_A;
  typedef float f32x4 __attribute__((ext_vector_type(4)));

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wid / WAVES_N) * 64, wn0 = (wid % WAVES_N) * 64;
  const int P = Ho * Wo, HiWi = Hi * Wi;
  int bx, by;
  {
    const int gx = (P + BN - 1) / BN, gy = (M + BM - 1) / BM;
    const int lin = blockIdx.x;
    pfst_tile_order(lin, gx, gy, ks == 3, bx, by);
  }
  const int p0 = bx * BN, m0 = by * BM, n = blockIdx.y * gridDim.z + blockIdx.z;
  const int spt = C / 32;                                // K=32 steps per filter tap
  const int KT16 = (C / 16) * ks * ks;
  in += (i64)n * in_bs;
  out += (i64)n * out_bs;

  const int pix = tid & (BN - 1), kh = tid >> 7;
  const int p = p0 + pix;
  const bool pvalid = p < P;
  const int oy = pvalid ? p / Wo : 0;
  const int ox = pvalid ? p - oy * Wo : 0;

  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(wk6) + (i64)blockIdx.y * KT16 * 2 * NP * M, 0,
                                                                         KT16 * 2 * NP * M * 16, 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, C * HiWi * 4, 0x00020000);
  unsigned a_voff, b_voff;
  {
    const int seg = tid / BM, row = tid - seg * BM;        // chunk c = tid + 256 i of a tile -> segment seg + 2 i, same row
    a_voff = (m0 + row < M) ? 16u * ((unsigned)seg * (unsigned)M + (unsigned)(m0 + row)) : OOB;
  }
  const int a_chunk = 2 * M * 16, a_tile = 2 * NP * M * 16, b_chan = HiWi * 4;
  auto set_tap = [&](int ty, int tx) {
    int sy, sx;
    const bool ok = pvalid & src_coord(oy, ty, ca, cb, cc, cdivv, Hi, sy) & src_coord(ox, tx, ca, cb, cc, cdivv, Wi, sx);
    b_voff = ok ? 4u * ((unsigned)(kh * 8) * (unsigned)HiWi + (unsigned)(sy * Wi + sx)) : OOB;
  };

  uint4 areg[6];                                         // [tile 2][chunk 3]
  float breg[2][8];                                      // [tile][channel kh * 8 + i of the tile's 16]
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  int a_soff = 0;
  // load v = 0..21 of the next tile pair: the 16 activations first (the split needs them early), then the 6 weight chunks
  auto load_one = [&](auto vc, int b_soff) {
    constexpr int v = decltype(vc)::value;
    if constexpr (v < 16)
      breg[v >> 3][v & 7] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b_rsrc, b_voff, b_soff + ((v >> 3) * 16 + (v & 7)) * b_chan, 0));
    else
      areg[v - 16] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, a_voff, a_soff + ((v - 16) / 3) * a_tile + ((v - 16) % 3) * a_chunk, 0));
  };

  set_tap(0, 0);
  static_for<22>([&](auto vc) { load_one(vc, 0); });
  a_soff += 2 * a_tile;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    uint4 q0, q1, q2;
    split8(breg[t], q0, q1, q2);
#pragma unroll
    for (int i = 0; i < 3; ++i) As[t * TILE_A + tid + 256 * i] = areg[t * 3 + i];
    Bs[t * TILE_B + (0 * 2 + kh) * BN + pix] = q0;
    Bs[t * TILE_B + (1 * 2 + kh) * BN + pix] = q1;
    Bs[t * TILE_B + (2 * 2 + kh) * BN + pix] = q2;
  }
  __syncthreads();

  const int l15 = lane & 15, lq = lane >> 4;
  // fragment of k-quarter lq: tile lq / 2, half lq % 2
  const int a_frag = (lq >> 1) * TILE_A + (lq & 1) * BM + wm0 + l15;
  const int b_frag = (lq >> 1) * TILE_B + (lq & 1) * BN + wn0 + l15;
  auto step = [&](auto load_tag, int b_soff) {
    constexpr bool LOAD = decltype(load_tag)::value;
    bf16x8 af[4][NP], bf[4][NP];
    auto rd_a = [&](int i, int pl) { af[i][pl] = __builtin_bit_cast(bf16x8, As[a_frag + pl * 2 * BM + i * 16]); };
    auto rd_b = [&](int j, int pl) { bf[j][pl] = __builtin_bit_cast(bf16x8, Bs[b_frag + pl * 2 * BN + j * 16]); };
    // fragment reads in the order the terms (a2 b0) (a1 b1) (a0 b2) need them; r = 0..23, the first five before the first MFMA
    auto read_frag = [&](auto rc) {
      constexpr int r = decltype(rc)::value;
      if constexpr (r == 0) rd_a(0, 2);
      else if constexpr (r < 5) rd_b(r - 1, 0);
      else if constexpr (r < 8) rd_a(r - 4, 2);
      else if constexpr (r == 8) rd_a(0, 1);
      else if constexpr (r < 13) rd_b(r - 9, 1);
      else if constexpr (r < 16) rd_a(r - 12, 1);
      else if constexpr (r == 16) rd_a(0, 0);
      else if constexpr (r < 21) rd_b(r - 17, 2);
      else rd_a(r - 20, 0);
    };
    static_for<5>([&](auto rc) { read_frag(rc); });
    __builtin_amdgcn_sched_barrier(0);
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0};            // smallest terms first
    constexpr int PB[6] = {0, 1, 2, 0, 1, 0};
    SplitState s0, s1;
    // slots: 0-18 the remaining fragment reads, 0-21 the next pair's global loads, 32 barrier A (every wave holds its fragments: the
    // LDS image may be overwritten), 40-83 two split instructions each, 84-95 the twelve LDS stores; barrier B after the last MFMA
    static_for<96>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      constexpr int t = m >> 4, i = (m >> 2) & 3, j = m & 3;
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][PA[t]], bf[j][PB[t]], acc[i][j], 0, 0, 0);
      if constexpr (m < 19) read_frag(std::integral_constant<int, 5 + m>());
      if constexpr (m == 32) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's fragment reads have landed
        __builtin_amdgcn_s_barrier();
      }
      if constexpr (LOAD) {
        if constexpr (m < 22) load_one(mc, b_soff);
        if constexpr (m >= 40 && m < 84) {
          constexpr int k = (m - 40) * 2;
          if constexpr (k < 44) { split_op<k>(breg[0], s0); split_op<k + 1>(breg[0], s0); }
          else { split_op<k - 44>(breg[1], s1); split_op<k - 43>(breg[1], s1); }
        }
        if constexpr (m >= 84 && m < 90) As[((m - 84) / 3) * TILE_A + tid + 256 * ((m - 84) % 3)] = areg[m - 84];
        if constexpr (m >= 90 && m < 93) Bs[((m - 90) * 2 + kh) * BN + pix] = split_piece(s0, m - 90);
        if constexpr (m >= 93) Bs[TILE_B + ((m - 93) * 2 + kh) * BN + pix] = split_piece(s1, m - 93);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    if (LOAD) a_soff += 2 * a_tile;
    __syncthreads();
  };
  const int chan_step = 32 * HiWi * 4;
  for (int ty = 0; ty < ks; ++ty)
    for (int tx = 0; tx < ks; ++tx) {
      for (int sidx = 1; sidx < spt; ++sidx) step(std::true_type(), sidx * chan_step);
      int nty = ty, ntx = tx + 1;                        // the last step of this tap fetches the first pair of the next one
      if (ntx == ks) { ntx = 0; nty += 1; }
      if (nty < ks) {
        set_tap(nty, ntx);
        step(std::true_type(), 0);
      } else {
        step(std::false_type(), 0);
      }
    }

  // D of v_mfma_f32_16x16x32: lane holds rows 4 (lane / 16) + r, column lane % 16 of its 16 x 16 tile.  Two passes of 32 rows per wave
  // through LDS (row stride 68 floats, 34 KB for the four waves) into the 32x32 layout conv_epilogue expects.
  static_assert(sizeof(smem) >= 4 * 32 * 68 * sizeof(float), "re-layout scratch must fit into the tiles");
  float* const ws = reinterpret_cast<float*>(smem) + wid * (32 * 68);
  const int l31 = lane & 31, lh = lane >> 5;
  pfst_f32x16 acc32[2][2];
#pragma unroll
  for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) ws[(i * 16 + 4 * lq + r) * 68 + j * 16 + l15] = acc[hb * 2 + i][j][r];
    // a wave reads back only what its own lanes wrote and its LDS operations complete in order; the fence keeps the compiler from
    // moving the loads of other lanes' values above the stores (and the next pass's stores above these loads)
    wave_lds_phase_fence();
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc32[hb][j][r] = ws[((r & 3) + 8 * (r >> 2) + 4 * lh) * 68 + j * 32 + l31];
    wave_lds_phase_fence();
  }
  if (BNB != 0) {
    __syncthreads();                                     // the reduction scratch overlaps other waves' re-layout areas
    static_assert(sizeof(smem) >= 4 * PFST_ROWSUM_LDS_FLOATS * sizeof(float), "epilogue scratch must fit into the tiles");
    conv_epilogue<2, 2, WAVES_N, BN, BNB, true>(acc32, out, bias, stats, stats_T, accumulate, M, P, m0, p0, wm0, wn0, bx, n, wid, lane, bnb,
                                                reinterpret_cast<float*>(smem));
  } else {
    conv_epilogue<2, 2, WAVES_N, BN>(acc32, out, bias, stats, stats_T, accumulate, M, P, m0, p0, wm0, wn0, bx, n, wid, lane);
  }
}

__global__ __launch_bounds__(256, 2) void conv_igemm_split_pair_kernel(
    const float* __restrict__ in, i64 in_bs, const uint4* __restrict__ wk6, const float* __restrict__ bias,
    float* __restrict__ out, i64 out_bs, int C, int Hi, int Wi, int M, int Ho, int Wo, int ks,
    int ca, int cb, int cc, int cdivv, int accumulate, float* __restrict__ stats, int stats_T) {
  conv_igemm_split_pair_body<0>(in, in_bs, wk6, bias, out, out_bs, C, Hi, Wi, M, Ho, Wo, ks, ca, cb, cc, cdivv, accumulate, stats, stats_T,
                                PfstBnbArgs());
}
template <int BNB>
__global__ __launch_bounds__(256, 2) void conv_igemm_split_pair_bnb_kernel(
    const float* __restrict__ in, i64 in_bs, const uint4* __restrict__ wk6, const float* __restrict__ bias,
    float* __restrict__ out, i64 out_bs, int C, int Hi, int Wi, int M, int Ho, int Wo, int ks,
    int ca, int cb, int cc, int cdivv, int accumulate, float* __restrict__ stats, int stats_T, PfstBnbArgs bnb) {
  conv_igemm_split_pair_body<BNB>(in, in_bs, wk6, bias, out, out_bs, C, Hi, Wi, M, Ho, Wo, ks, ca, cb, cc, cdivv, accumulate, stats, stats_T, bnb);
}

// w[Cout][Cin][T] -> split K-major images.  fprop: k = t*Cin+ci, row m = co;  dgrad: k = t*Cout+co, row m = ci.
// layout: [k/16][piece 3][k-half 2][row][8 x bf16]  (one uint4 per (k16-group, piece, half, row))
// blockIdx.y: filter set (the (m+2)^2 transform-domain sets of a Winograd layer: `set_in` floats / `set_out` 16-byte chunks apart; 0 otherwise)
__global__ void pack_weight_split_kernel(const float* __restrict__ w, uint4* __restrict__ wf, uint4* __restrict__ wd, int Cout, int Cin, int T,
                                         i64 set_in = 0, i64 set_out = 0) {
  w += blockIdx.y * set_in;
  if (wf) wf += blockIdx.y * set_out;
  if (wd) wd += blockIdx.y * set_out;
  const i64 nf = (i64)(T * Cin / 16) * 2 * Cout, nd = (i64)(T * Cout / 16) * 2 * Cin;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < nf + nd; i += (i64)gridDim.x * blockDim.x) {
    const bool dg = i >= nf;
    const i64 e = dg ? i - nf : i;
    const int M = dg ? Cin : Cout, Cq = dg ? Cout : Cin;   // rows, channels-per-tap along K
    if ((dg ? wd : wf) == nullptr) continue;
    if (Cq % 16 != 0) continue;
    const int row = (int)(e % M);
    const i64 gh = e / M;
    const int h = (int)(gh & 1);
    const int g = (int)(gh >> 1);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = g * 16 + h * 8 + j;
      const int t = k / Cq, c = k - t * Cq;
      const int co = dg ? c : row, ci = dg ? row : c;
      v[j] = w[((i64)co * Cin + ci) * T + t];
    }
    uint4 q0, q1, q2;
    split8(v, q0, q1, q2);
    uint4* dst = dg ? wd : wf;
    const i64 base = (i64)g * 2 * NP * M;
    dst[base + (i64)(0 * 2 + h) * M + row] = q0;
    dst[base + (i64)(1 * 2 + h) * M + row] = q1;
    dst[base + (i64)(2 * 2 + h) * M + row] = q2;
  }
}

// ---------------------------------------------------------------------------------------------
// weight gradient with the same 6-term split: dW[m][j] += sum_p dY[m][p] * X[ci(j)][src(p, tap(j))],  j = ci*T + tap.
// Both operands are activations and pixel(K)-contiguous: each thread stages 8 consecutive pixels of one row
// (dY: two 16-byte loads; X: eight range-checked buffer loads), splits them in registers and writes three 16-byte
// pieces per operand into the [piece][k-half][row][8] LDS image.
// ---------------------------------------------------------------------------------------------
template <int BM, int T>
__global__ __launch_bounds__(256) void conv_wgrad_split_kernel(
    const float* __restrict__ x, i64 x_bs, const float* __restrict__ dy, i64 dy_bs, float* __restrict__ dw,
    int Cin, int Hi, int Wi, int M, int Ho, int Wo, int stride, int dil, int pad, int chunks, int chunk_len, i64 det_stride = 0) {
  constexpr int BJ = 128;
  constexpr int WM = BM >= 64 ? 64 : 32;
  constexpr int WAVES_M = BM / WM;
  constexpr int WAVES_N = 4 / WAVES_M;
  constexpr int WN = BJ / WAVES_N;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int KS = T == 9 ? 3 : 1;
  constexpr unsigned OOB = 0x80000000u;

  __shared__ uint4 As[2][2 * NP * BM];
  __shared__ uint4 Bs[2][2 * NP * BJ];

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave id in an SGPR
  const int wm0 = (wid / WAVES_N) * WM, wn0 = (wid % WAVES_N) * WN;
  const int P = Ho * Wo, HiWi = Hi * Wi, J = Cin * T;
  const int j0 = blockIdx.x * BJ, m0 = blockIdx.y * BM;
  const int n = blockIdx.z / chunks, chunk = blockIdx.z - n * chunks;
  const int pbeg = chunk * chunk_len;
  const int pend = min(P, pbeg + chunk_len);
  if (pbeg >= pend) return;
  x += (i64)n * x_bs;
  dy += (i64)n * dy_bs;
  dw += (i64)blockIdx.z * det_stride;         // deterministic mode (det.h): one scratch tile-set per grid slice
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy), 0, M * P * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, Cin * HiWi * 4, 0x00020000);

  // staging role: thread -> (row, k-half).  A rows only exist for tid < 2*BM.
  const int srow = tid >> 1, half = tid & 1;
  const bool a_thread = srow < BM;
  const int am = m0 + srow;
  const unsigned a_row_off = (a_thread && am < M) ? 4u * (unsigned)am * (unsigned)P : OOB;
  const int bj = j0 + srow;                       // always < j0 + 128
  int b_coff = -1, b_dy = 0, b_dx = 0;
  if (bj < J) {
    const int ci = bj / T, tap = bj - ci * T;
    const int ty = tap / KS, tx = tap - ty * KS;
    b_coff = ci * HiWi;
    b_dy = ty * dil - pad;
    b_dx = tx * dil - pad;
  }

  float areg[8], breg[8];
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto load_tile = [&](int pk0) {
    const int pb = pk0 + half * 8;                 // first of this thread's 8 pixels
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int p = pb + e;
      const bool pv = p < pend;
      const unsigned va = (pv && a_row_off != OOB) ? a_row_off + 4u * (unsigned)p : OOB;
      areg[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(a_rsrc, va, 0, 0));
      const int pc = pv ? p : pend - 1;
      const int oy = pc / Wo, ox = pc - oy * Wo;
      const int sy = oy * stride + b_dy, sx = ox * stride + b_dx;
      const bool ok = pv & (b_coff >= 0) & ((unsigned)sy < (unsigned)Hi) & ((unsigned)sx < (unsigned)Wi);
      const unsigned vb = ok ? 4u * (unsigned)(b_coff + sy * Wi + sx) : OOB;
      breg[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b_rsrc, vb, 0, 0));
    }
  };
  auto store_tile = [&](int buf) {
    uint4 q0, q1, q2;
    if (a_thread) {
      split8(areg, q0, q1, q2);
      As[buf][(0 * 2 + half) * BM + srow] = q0;
      As[buf][(1 * 2 + half) * BM + srow] = q1;
      As[buf][(2 * 2 + half) * BM + srow] = q2;
    }
    split8(breg, q0, q1, q2);
    Bs[buf][(0 * 2 + half) * BJ + srow] = q0;
    Bs[buf][(1 * 2 + half) * BJ + srow] = q1;
    Bs[buf][(2 * 2 + half) * BJ + srow] = q2;
  };

  const int KT = (pend - pbeg + 15) / 16;
  load_tile(pbeg);
  store_tile(0);
  __syncthreads();
  const int l31 = lane & 31, lh = lane >> 5;
  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < KT) load_tile(pbeg + (kt + 1) * 16);
    bf16x8 af[TM][NP], bf[TN][NP];
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i][pl] = __builtin_bit_cast(bf16x8, As[cur][(pl * 2 + lh) * BM + wm0 + i * 32 + l31]);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j][pl] = __builtin_bit_cast(bf16x8, Bs[cur][(pl * 2 + lh) * BJ + wn0 + j * 32 + l31]);
    }
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0};
    constexpr int PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PA[t]], bf[j][PB[t]], acc[i][j], 0, 0, 0);
    if (kt + 1 < KT) store_tile(cur ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int jj = j0 + wn0 + j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < M && jj < J) atomicAdd(&dw[(i64)m * J + jj], acc[i][j][r]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Weight gradient of 1x1 convolutions (and the transform-domain products of the Winograd weight gradient) with the 6-term
// split, "K-quad" data movement: both operands are pixel(K)-contiguous, so a thread stages 8 consecutive pixels of one row
// with TWO buffer_load_dwordx4 (no per-element address arithmetic, no per-element range checks: whole quads are in or out),
// splits them once in registers and writes the three 16-byte pieces of the [piece][k-half][row][8 x bf16] LDS image.
// On the bf16 matrix pipe VALU work co-issues with the MFMAs (unlike the fp32-input MFMA), so the 16 splits per thread and
// K-step (~90 VALU) hide behind the 24 MFMAs of the step.  Tiling, split-K chunking and the XCD-contiguous slice order are
// those of conv_wgrad_q_kernel (conv_wgrad_q.hip).  The first split kernel (conv_wgrad_split_kernel below: 16 scalar loads
// with index arithmetic per thread and step) ran at 70-90 TFLOP/s-equivalent, slower than the fp32-MFMA kernel.
// ---------------------------------------------------------------------------------------------
template <int BM>
__global__ __launch_bounds__(256) void conv_wgrad_split_q_kernel(
    const float* __restrict__ x, i64 x_bs, const float* __restrict__ dy, i64 dy_bs, float* __restrict__ dw,
    int J, int M, int P, int chunks, int chunk_len, int N, i64 x_gs, i64 dy_gs, i64 dw_gs, int gx, int gy, int gz, int xcd_order) {
  constexpr int BJ = 128;
  constexpr int WM = BM >= 64 ? 64 : 32;
  constexpr int WAVES_M = BM / WM;
  constexpr int WAVES_N = 4 / WAVES_M;
  constexpr int WN = BJ / WAVES_N;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr unsigned OOB = 0x80000000u;

  __shared__ uint4 As[2][2 * NP * BM];
  __shared__ uint4 Bs[2][2 * NP * BJ];

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wid / WAVES_N) * WM, wn0 = (wid % WAVES_N) * WN;
  int bx, by, bz;                                   // tile and K slice: all tiles of a slice on one XCD (see conv_wgrad_q_kernel)
  {
    const int lin = blockIdx.x, tiles = gx * gy, z8 = gz & ~7;
    if (xcd_order && lin < tiles * z8) {
      const int xcd = lin & 7, idx = lin >> 3;
      const int sl = idx / tiles, t = idx - sl * tiles;
      bz = sl * 8 + xcd;
      by = t / gx;
      bx = t - by * gx;
    } else {
      bz = lin / tiles;
      const int t = lin - bz * tiles;
      by = t / gx;
      bx = t - by * gx;
    }
  }
  const int j0 = bx * BJ, m0 = by * BM;
  const int ng = bz / chunks, chunk = bz - ng * chunks;
  const int grp = ng / N, n = ng - grp * N;
  const int pbeg = chunk * chunk_len;
  const int pend = min(P, pbeg + chunk_len);
  if (pbeg >= pend) return;
  x += (i64)grp * x_gs + (i64)n * x_bs;
  dy += (i64)grp * dy_gs + (i64)n * dy_bs;
  dw += dw_gs >= 0 ? (i64)grp * dw_gs : (i64)bz * -dw_gs;      // < 0: deterministic mode, one scratch tile-set per grid slice (det.h)
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy), 0, M * P * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, J * P * 4, 0x00020000);

  // staging role: thread -> (row, k-half): 8 consecutive pixels of one row per operand
  const int srow = tid >> 1, half = tid & 1;
  const bool a_thread = srow < BM;
  const unsigned a_voff = (a_thread && m0 + srow < M) ? 4u * ((unsigned)(m0 + srow) * (unsigned)P + 8u * half) : OOB;
  const unsigned b_voff = (j0 + srow < J) ? 4u * ((unsigned)(j0 + srow) * (unsigned)P + 8u * half) : OOB;

  float4 areg[2], breg[2];
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto ld4 = [](const __amdgpu_buffer_rsrc_t& rs, unsigned vo, int so) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, 0));
  };
  auto load_tile = [&](int pk0) {
    const int p = pk0 + 8 * half;                 // P % 4 == 0 and chunk_len % 16 == 0: a quad is entirely in or out
    const bool v0 = p < pend, v1 = p + 4 < pend;
    const int soff = pk0 * 4;
    areg[0] = ld4(a_rsrc, v0 ? a_voff : OOB, soff);
    areg[1] = ld4(a_rsrc, v1 ? a_voff : OOB, soff + 16);
    breg[0] = ld4(b_rsrc, v0 ? b_voff : OOB, soff);
    breg[1] = ld4(b_rsrc, v1 ? b_voff : OOB, soff + 16);
  };
  auto store_tile = [&](int buf) {
    uint4 q0, q1, q2;
    if (a_thread) {
      const float va[8] = {areg[0].x, areg[0].y, areg[0].z, areg[0].w, areg[1].x, areg[1].y, areg[1].z, areg[1].w};
      split8(va, q0, q1, q2);
      As[buf][(0 * 2 + half) * BM + srow] = q0;
      As[buf][(1 * 2 + half) * BM + srow] = q1;
      As[buf][(2 * 2 + half) * BM + srow] = q2;
    }
    const float vb[8] = {breg[0].x, breg[0].y, breg[0].z, breg[0].w, breg[1].x, breg[1].y, breg[1].z, breg[1].w};
    split8(vb, q0, q1, q2);
    Bs[buf][(0 * 2 + half) * BJ + srow] = q0;
    Bs[buf][(1 * 2 + half) * BJ + srow] = q1;
    Bs[buf][(2 * 2 + half) * BJ + srow] = q2;
  };

  const int KT = (pend - pbeg + 15) / 16;
  load_tile(pbeg);
  store_tile(0);
  __syncthreads();
  const int l31 = lane & 31, lh = lane >> 5;
  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < KT) load_tile(pbeg + (kt + 1) * 16);
    bf16x8 af[TM][NP], bf[TN][NP];
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i][pl] = __builtin_bit_cast(bf16x8, As[cur][(pl * 2 + lh) * BM + wm0 + i * 32 + l31]);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j][pl] = __builtin_bit_cast(bf16x8, Bs[cur][(pl * 2 + lh) * BJ + wn0 + j * 32 + l31]);
    }
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0};
    constexpr int PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PA[t]], bf[j][PB[t]], acc[i][j], 0, 0, 0);
    if (kt + 1 < KT) store_tile(cur ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int jj = j0 + wn0 + j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < M && jj < J) atomicAdd(&dw[(i64)m * J + jj], acc[i][j][r]);
      }
    }
  }
}

// The K-quad split weight gradient with the wave's own staging work inside its MFMA stream (see conv_igemm_split_pipe_body for the
// measurement behind it).  Both operands are split in registers here (88 VALU per K-step), so the loads run TWO tiles ahead into
// alternating register sets: step k consumes LDS buffer k & 1, splits tile k+1 (loaded during step k-1) in the gaps of its 24 MFMAs
// and stores it to the other buffer, and issues the loads of tile k+2.  Tiles past the end of the K slice load zeros (out-of-range
// buffer offsets) and are never consumed, so the step is branch-free.  BM = 128 only.
__global__ __launch_bounds__(256, 2) void conv_wgrad_split_q_pipe_kernel(
    const float* __restrict__ x, i64 x_bs, const float* __restrict__ dy, i64 dy_bs, float* __restrict__ dw,
    int J, int M, int P, int chunks, int chunk_len, int N, i64 x_gs, i64 dy_gs, i64 dw_gs, int gx, int gy, int gz, int xcd_order) {
  constexpr int BM = 128, BJ = 128, WM = 64, WAVES_N = 2, WN = 64, TM = 2, TN = 2;
  constexpr unsigned OOB = 0x80000000u;
  __shared__ uint4 As[2][2 * NP * BM];
  __shared__ uint4 Bs[2][2 * NP * BJ];

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wid / WAVES_N) * WM, wn0 = (wid % WAVES_N) * WN;
  int bx, by, bz;                                   // tile and K slice: all tiles of a slice on one XCD (see conv_wgrad_q_kernel)
  {
    const int lin = blockIdx.x, tiles = gx * gy, z8 = gz & ~7;
    if (xcd_order && lin < tiles * z8) {
      const int xcd = lin & 7, idx = lin >> 3;
      const int sl = idx / tiles, t = idx - sl * tiles;
      bz = sl * 8 + xcd;
      by = t / gx;
      bx = t - by * gx;
    } else {
      bz = lin / tiles;
      const int t = lin - bz * tiles;
      by = t / gx;
      bx = t - by * gx;
    }
  }
  const int j0 = bx * BJ, m0 = by * BM;
  const int ng = bz / chunks, chunk = bz - ng * chunks;
  const int grp = ng / N, n = ng - grp * N;
  const int pbeg = chunk * chunk_len;
  const int pend = min(P, pbeg + chunk_len);
  if (pbeg >= pend) return;
  x += (i64)grp * x_gs + (i64)n * x_bs;
  dy += (i64)grp * dy_gs + (i64)n * dy_bs;
  dw += dw_gs >= 0 ? (i64)grp * dw_gs : (i64)bz * -dw_gs;      // < 0: deterministic mode, one scratch tile-set per grid slice (det.h)
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy), 0, M * P * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, J * P * 4, 0x00020000);

  const int srow = tid >> 1, half = tid & 1;        // staging role: 8 consecutive pixels of one row per operand
  const unsigned a_voff = (m0 + srow < M) ? 4u * ((unsigned)(m0 + srow) * (unsigned)P + 8u * half) : OOB;
  const unsigned b_voff = (j0 + srow < J) ? 4u * ((unsigned)(j0 + srow) * (unsigned)P + 8u * half) : OOB;

  float la[2][8], lb[2][8];                         // two register sets of loaded tiles
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // load q = 0..3 of the tile starting at pixel pk0: (A, B) x (first, second quad); P % 4 == 0 and chunk_len % 16 == 0: a quad is
  // entirely in or out
  auto load_quad = [&](auto qc, auto setc, int pk0) {
    constexpr int q = decltype(qc)::value, SET = decltype(setc)::value;
    const int p = pk0 + 8 * half + 4 * (q & 1);
    const bool v = p < pend;
    const float4 t = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(q < 2 ? a_rsrc : b_rsrc, v ? (q < 2 ? a_voff : b_voff) : OOB,
                                                                                      pk0 * 4 + 16 * (q & 1), 0));
    float (&dst)[8] = q < 2 ? la[SET] : lb[SET];
    dst[4 * (q & 1) + 0] = t.x; dst[4 * (q & 1) + 1] = t.y; dst[4 * (q & 1) + 2] = t.z; dst[4 * (q & 1) + 3] = t.w;
  };
  const int l31 = lane & 31, lh = lane >> 5;

  // prologue: tile 0 through the plain split into buffer 0, tile 1 into register set 1
  static_for<4>([&](auto qc) { load_quad(qc, std::integral_constant<int, 0>(), pbeg); });
  {
    uint4 q0, q1, q2;
    split8(la[0], q0, q1, q2);
    As[0][(0 * 2 + half) * BM + srow] = q0; As[0][(1 * 2 + half) * BM + srow] = q1; As[0][(2 * 2 + half) * BM + srow] = q2;
    split8(lb[0], q0, q1, q2);
    Bs[0][(0 * 2 + half) * BJ + srow] = q0; Bs[0][(1 * 2 + half) * BJ + srow] = q1; Bs[0][(2 * 2 + half) * BJ + srow] = q2;
  }
  static_for<4>([&](auto qc) { load_quad(qc, std::integral_constant<int, 1>(), pbeg + 16); });
  __syncthreads();

  // one K-step on LDS buffer CUR = k & 1.  Issue order, a full scheduling barrier after each slot: 4 fragment reads, then 24 slots of
  // one MFMA + fillers -- slots 0-7 one fragment read, 0-3 one global load of tile k+2, 0-21 four split instructions of tile k+1
  // (dy first), 11 / 22 the three LDS stores of the dy / x pieces.
  auto step = [&](auto curc, int kt) {
    constexpr int CUR = decltype(curc)::value, NXT = CUR ^ 1;
    bf16x8 af[TM][NP], bf[TN][NP];
    auto read_frag = [&](int r) {       // r = 0..11: (a2, b0) | (a1, b1) | (a0, b2), rows i / columns j inside
      const int q = r >> 2, e = r & 3;
      if (e < 2) af[e][2 - q] = __builtin_bit_cast(bf16x8, As[CUR][((2 - q) * 2 + lh) * BM + wm0 + e * 32 + l31]);
      else bf[e - 2][q] = __builtin_bit_cast(bf16x8, Bs[CUR][(q * 2 + lh) * BJ + wn0 + (e - 2) * 32 + l31]);
    };
    static_for<4>([&](auto rc) { read_frag(decltype(rc)::value); });
    __builtin_amdgcn_sched_barrier(0);
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0};            // smallest terms first
    constexpr int PB[6] = {0, 1, 2, 0, 1, 0};
    SplitState sa, sb;
    const int pk2 = pbeg + (kt + 2) * 16;
    static_for<24>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      constexpr int t = m >> 2, i = (m >> 1) & 1, j = m & 1;
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PA[t]], bf[j][PB[t]], acc[i][j], 0, 0, 0);
      if constexpr (m < 8) read_frag(4 + m);
      if constexpr (m < 22) {
        static_for<4>([&](auto kc) {
          constexpr int k = m * 4 + decltype(kc)::value;
          if constexpr (k < 44) split_op<k>(la[NXT], sa);
          else split_op<k - 44>(lb[NXT], sb);
        });
      }
      // (the loads of tile k+2 overwrite set CUR: its values were split during the previous step)
      if constexpr (m < 4) load_quad(mc, curc, pk2);
      if constexpr (m == 11) {
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) As[NXT][(pl * 2 + half) * BM + srow] = split_piece(sa, pl);
      }
      if constexpr (m == 22) {
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) Bs[NXT][(pl * 2 + half) * BJ + srow] = split_piece(sb, pl);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    __syncthreads();
  };
  const int KT = (pend - pbeg + 15) / 16;
  for (int kt = 0; kt < KT; kt += 2) {
    step(std::integral_constant<int, 0>(), kt);
    if (kt + 1 < KT) step(std::integral_constant<int, 1>(), kt + 1);
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int jj = j0 + wn0 + j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < M && jj < J) atomicAdd(&dw[(i64)m * J + jj], acc[i][j][r]);
      }
    }
  }
}

template <int BM>
int launch_wgrad_split_q(const float* x, i64 x_bs, const float* dy, i64 dy_bs, float* dw, int N, int J, int M, int P, int groups,
                         i64 x_gs, i64 dy_gs, i64 dw_gs, hipStream_t s) {
  const int tiles = cdiv(J, 128) * cdiv(M, BM) * groups;
  // split-K chunking: whole rounds of resident workgroups (49 KB of LDS: 3 per CU)
  const double slots = 256.0 * 3;
  int chunks = 1;
  double best = -1.0;
  for (int c = 1; c <= 64 && (c == 1 || P / c >= 512); ++c) {
    const double rounds = (double)tiles * N * c / slots;
    const double eff = rounds < 2.0 ? 0.45 * rounds : rounds / ceil(rounds);
    if (eff > best + 0.02) { best = eff; chunks = c; }
    if (eff >= 0.93) break;
  }
  int chunk_len = ((cdiv(P, chunks) + 15) / 16) * 16;
  chunks = cdiv(P, chunk_len);
  const int gx = cdiv(J, 128), gy = cdiv(M, BM), gz = N * groups * chunks;
  PFST_CHECK_ARG((i64)gx * gy * gz < (1ll << 31));
  const i64 elems = (i64)M * J;
  bool det_ok;
  float* const ws = wgrad_det_scratch(elems, gz, s, det_ok);          // deterministic mode (det.h): one scratch tile-set per grid slice
  PFST_CHECK_ARG(det_ok);
  float* const dwk = ws ? ws : dw;
  const i64 gsk = ws ? -elems : dw_gs;
  if (BM == 128)
    hipLaunchKernelGGL(conv_wgrad_split_q_pipe_kernel, dim3(gx * gy * gz), dim3(256), 0, s, x, x_bs, dy, dy_bs, dwk, J, M, P, chunks, chunk_len,
                       N, x_gs, dy_gs, gsk, gx, gy, gz, 1);
  else
    hipLaunchKernelGGL((conv_wgrad_split_q_kernel<BM>), dim3(gx * gy * gz), dim3(256), 0, s, x, x_bs, dy, dy_bs, dwk, J, M, P, chunks, chunk_len,
                       N, x_gs, dy_gs, gsk, gx, gy, gz, 1);
  if (ws) wgrad_det_reduce(ws, dw, elems, groups, N * chunks, dw_gs, s);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

template <int BM, int T>
int launch_wgrad_split(const float* x, i64 x_bs, const float* dy, i64 dy_bs, float* dw, int N, int Cin, int Hi, int Wi, int M,
                       int Ho, int Wo, int stride, int dil, int pad, hipStream_t s) {
  const int P = Ho * Wo, J = Cin * T;
  const int tiles = cdiv(J, 128) * cdiv(M, BM);
  int chunks = 1;
  while ((i64)tiles * N * chunks < 1024 && P / (chunks * 2) >= 512) chunks *= 2;
  int chunk_len = ((cdiv(P, chunks) + 15) / 16) * 16;
  chunks = cdiv(P, chunk_len);
  dim3 grid(cdiv(J, 128), cdiv(M, BM), N * chunks);
  const i64 elems = (i64)M * J;
  bool det_ok;
  float* const ws = wgrad_det_scratch(elems, (i64)N * chunks, s, det_ok);          // deterministic mode (det.h)
  PFST_CHECK_ARG(det_ok);
  hipLaunchKernelGGL((conv_wgrad_split_kernel<BM, T>), grid, dim3(256), 0, s, x, x_bs, dy, dy_bs, ws ? ws : dw, Cin, Hi, Wi, M, Ho, Wo, stride,
                     dil, pad, chunks, chunk_len, ws ? elems : (i64)0);
  if (ws) wgrad_det_reduce(ws, dw, elems, 1, N * chunks, 0, s);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

template <int BM>
int launch_split(const float* in, i64 in_bs, const void* wk6, const float* bias, float* out, i64 out_bs, int N, int C, int Hi,
                 int Wi, int M, int Ho, int Wo, int ks, int a, int b, int c, int d, int acc, float* stats, int stats_T, int groups,
                 hipStream_t s, const PfstBnbArgs* bnb = nullptr) {
  dim3 grid(cdiv((i64)Ho * Wo, BN) * cdiv(M, BM), groups, N);
  constexpr int lds_pad = 0;
  // measured per layer (bf16x6 train step): the pipelined loop wins from K = 512 up, loses 1-5 % on the short 1x1 / Winograd-domain GEMMs
  const bool pipe = (i64)C * ks * ks >= 512;
  // the K = 32 pairing on the 16x16x32 MFMA shape: whole 32-channel steps per tap
  if (BM == 128 && pipe && C % 32 == 0) {
    if (bnb && bnb->x) {
      PFST_CHECK_ARG(M % BM == 0 && !bias && !stats && groups == 1 && bnb->coef && bnb->partials);
#define PFST_LAUNCH_PAIR_BNB(MODE_)                                                                                                        \
      hipLaunchKernelGGL((conv_igemm_split_pair_bnb_kernel<MODE_>), grid, dim3(256), 0, s, in, in_bs, (const uint4*)wk6, bias, out, out_bs, C, \
                         Hi, Wi, M, Ho, Wo, ks, a, b, c, d, acc, stats, stats_T, *bnb)
      if (!bnb->relu) PFST_LAUNCH_PAIR_BNB(3);
      else if (bnb->y) PFST_LAUNCH_PAIR_BNB(2);
      else PFST_LAUNCH_PAIR_BNB(1);
#undef PFST_LAUNCH_PAIR_BNB
    } else {
      hipLaunchKernelGGL(conv_igemm_split_pair_kernel, grid, dim3(256), 0, s, in, in_bs, (const uint4*)wk6, bias, out, out_bs, C, Hi, Wi, M, Ho,
                         Wo, ks, a, b, c, d, acc, stats, stats_T);
    }
    PFST_CHECK_LAUNCH();
    return PFST_OK;
  }
  if (bnb && bnb->x) {
    PFST_CHECK_ARG(M % BM == 0 && !bias && !stats && groups == 1 && bnb->coef && bnb->partials);
#define PFST_LAUNCH_SPLIT_BNB(MODE_)                                                                                              \
    if (BM == 128 && pipe)                                                                                                         \
      hipLaunchKernelGGL((conv_igemm_split_pipe_bnb_kernel<MODE_>), grid, dim3(256), 0, s, in, in_bs, (const uint4*)wk6, bias, out, out_bs, C, \
                         Hi, Wi, M, Ho, Wo, ks, a, b, c, d, acc, stats, stats_T, *bnb);                                            \
    else                                                                                                                           \
      hipLaunchKernelGGL((conv_igemm_split_kernel<BM, MODE_>), grid, dim3(256), 0, s, in, in_bs, (const uint4*)wk6, bias, out, out_bs, C,  \
                         Hi, Wi, M, Ho, Wo, ks, a, b, c, d, acc, stats, stats_T, *bnb)
    if (!bnb->relu) PFST_LAUNCH_SPLIT_BNB(3);
    else if (bnb->y) PFST_LAUNCH_SPLIT_BNB(2);
    else PFST_LAUNCH_SPLIT_BNB(1);
#undef PFST_LAUNCH_SPLIT_BNB
  } else {
    if (BM == 128 && pipe)
      hipLaunchKernelGGL(conv_igemm_split_pipe_kernel, grid, dim3(256), lds_pad, s, in, in_bs, (const uint4*)wk6, bias, out, out_bs, C, Hi, Wi,
                         M, Ho, Wo, ks, a, b, c, d, acc, stats, stats_T);
    else
      hipLaunchKernelGGL((conv_igemm_split_kernel<BM, 0>), grid, dim3(256), lds_pad, s, in, in_bs, (const uint4*)wk6, bias, out, out_bs, C, Hi,
                         Wi, M, Ho, Wo, ks, a, b, c, d, acc, stats, stats_T, PfstBnbArgs());
  }
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

}  // namespace

extern "C" int pfst_conv_pack_weight_split(const float* w, void* wk6_fprop, void* wk6_dgrad, int Cout, int Cin, int T, pfst_stream_t stream) {
  PFST_CHECK_ARG(w && (wk6_fprop || wk6_dgrad) && Cout > 0 && Cin > 0 && (T == 1 || T == 9));
  PFST_CHECK_ARG(!wk6_fprop || Cin % 16 == 0);
  PFST_CHECK_ARG(!wk6_dgrad || Cout % 16 == 0);
  const i64 n = (i64)Cout * Cin * T / 4;
  hipLaunchKernelGGL(pack_weight_split_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, w, (uint4*)wk6_fprop,
                     (uint4*)wk6_dgrad, Cout, Cin, T);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

extern "C" int pfst_conv_igemm_split(const float* in, long long in_bs, const void* wk6, const float* bias, float* out, long long out_bs,
                                     int N, int C, int Hi, int Wi, int M, int Ho, int Wo, int ksize, int stride, int dil, int pad,
                                     int mode, int accumulate, float* stats, const pfst_bnb_fuse_t* bnb, pfst_stream_t stream) {
  PFST_CHECK_ARG(in && wk6 && out && N > 0 && C > 0 && M > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0);
  PFST_CHECK_ARG(!bnb || (bnb->x && bnb->x_bs >= (i64)M * Ho * Wo && (!bnb->y || bnb->y_bs >= (i64)M * Ho * Wo)));
  PFST_CHECK_ARG((ksize == 1 || ksize == 3) && (stride == 1 || stride == 2) && dil >= 1 && pad >= 0 && (mode == 0 || mode == 1));
  PFST_CHECK_ARG(in_bs >= (i64)C * Hi * Wi && out_bs >= (i64)M * Ho * Wo && N <= 65535);
  if (C % 16 != 0) {
    pfst_set_error(__FILE__, __LINE__, "split kernel needs C % 16 == 0 (use pfst_conv_igemm)");
    return PFST_ERR_UNSUPPORTED;
  }
  const int span = (ksize - 1) * dil;
  if (mode == 0) {
    PFST_CHECK_ARG(Ho == (Hi + 2 * pad - span - 1) / stride + 1 && Wo == (Wi + 2 * pad - span - 1) / stride + 1);
  } else {
    PFST_CHECK_ARG(Hi == (Ho + 2 * pad - span - 1) / stride + 1 && Wi == (Wo + 2 * pad - span - 1) / stride + 1);
  }
  int a, b, c, d;
  if (mode == 0) { a = stride; b = dil; c = -pad; d = 1; } else { a = 1; b = -dil; c = pad; d = stride; }
  hipStream_t s = (hipStream_t)stream;
  const int stats_T = N * pfst_conv_stats_slots(M, Ho, Wo);
  if (M > 64) return launch_split<128>(in, in_bs, wk6, bias, out, out_bs, N, C, Hi, Wi, M, Ho, Wo, ksize, a, b, c, d, accumulate, stats, stats_T, 1, s, bnb);
  if (M > 32) return launch_split<64>(in, in_bs, wk6, bias, out, out_bs, N, C, Hi, Wi, M, Ho, Wo, ksize, a, b, c, d, accumulate, stats, stats_T, 1, s, bnb);
  return launch_split<32>(in, in_bs, wk6, bias, out, out_bs, N, C, Hi, Wi, M, Ho, Wo, ksize, a, b, c, d, accumulate, stats, stats_T, 1, s, bnb);
}

extern "C" int pfst_conv_wgrad_split(const float* x, long long x_bs, const float* dy, long long dy_bs, float* dw,
                                     int N, int Cin, int Hi, int Wi, int Cout, int Ho, int Wo, int ksize, int stride, int dil, int pad,
                                     pfst_stream_t stream) {
  PFST_CHECK_ARG(x && dy && dw && N > 0 && Cin > 0 && Cout > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0);
  PFST_CHECK_ARG((ksize == 1 || ksize == 3) && (stride == 1 || stride == 2) && dil >= 1 && pad >= 0);
  const int span = (ksize - 1) * dil;
  PFST_CHECK_ARG(Ho == (Hi + 2 * pad - span - 1) / stride + 1 && Wo == (Wi + 2 * pad - span - 1) / stride + 1);
  PFST_CHECK_ARG(x_bs >= (i64)Cin * Hi * Wi && dy_bs >= (i64)Cout * Ho * Wo);
  hipStream_t s = (hipStream_t)stream;
  if (pfst_wgrad_split_q_eligible(x, x_bs, dy, dy_bs, Hi, Wi, Ho, Wo, ksize, stride))
    return pfst_wgrad_split_q_launch(x, x_bs, dy, dy_bs, dw, N, Cin, Cout, Ho * Wo, 1, 0, 0, 0, s);
#define PFST_WGS(BM_)                                                                                                  \
  return ksize == 3 ? launch_wgrad_split<BM_, 9>(x, x_bs, dy, dy_bs, dw, N, Cin, Hi, Wi, Cout, Ho, Wo, stride, dil, pad, s) \
                    : launch_wgrad_split<BM_, 1>(x, x_bs, dy, dy_bs, dw, N, Cin, Hi, Wi, Cout, Ho, Wo, stride, dil, pad, s)
  if (Cout > 64) { PFST_WGS(128); }
  if (Cout > 32) { PFST_WGS(64); }
  PFST_WGS(32);
#undef PFST_WGS
}

// internal: K-quad split weight gradient of 1x1 / grouped transform-domain products (used by pfst_conv_wgrad_split and pfst_wino_wgrad)
bool pfst_wgrad_split_q_eligible(const float* x, i64 x_bs, const float* dy, i64 dy_bs, int Hi, int Wi, int Ho, int Wo, int ksize, int stride) {
  if (ksize != 1 || stride != 1 || Hi != Ho || Wi != Wo) return false;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) return false;
  if ((x_bs | dy_bs) & 3) return false;
  return ((i64)Ho * Wo) % 4 == 0;
}
int pfst_wgrad_split_q_launch(const float* x, i64 x_bs, const float* dy, i64 dy_bs, float* dw, int N, int Cin, int Cout, int P, int groups,
                              i64 x_gs, i64 dy_gs, i64 dw_gs, hipStream_t s) {
  PFST_CHECK_ARG((i64)Cin * P * 4 < (1ll << 31) && (i64)Cout * P * 4 < (1ll << 31));
  if (Cout > 64) return launch_wgrad_split_q<128>(x, x_bs, dy, dy_bs, dw, N, Cin, Cout, P, groups, x_gs, dy_gs, dw_gs, s);
  if (Cout > 32) return launch_wgrad_split_q<64>(x, x_bs, dy, dy_bs, dw, N, Cin, Cout, P, groups, x_gs, dy_gs, dw_gs, s);
  return launch_wgrad_split_q<32>(x, x_bs, dy, dy_bs, dw, N, Cin, Cout, P, groups, x_gs, dy_gs, dw_gs, s);
}

// ---- Winograd on the bf16x6 GEMM (conv_winograd.hip supplies the transforms): the (m+2)^2 transform-domain products as ONE
// grouped launch of the split kernel, filter sets packed per transform index.
extern "C" int pfst_wino_gemm_split(const float* V, const void* U6, float* Mbuf, int N, int K, int M, int T, int m, pfst_stream_t stream) {
  PFST_CHECK_ARG(V && U6 && Mbuf && N > 0 && N <= 65535 && K > 0 && K % 16 == 0 && M > 0 && T > 0 && (m == 2 || m == 4));
  const int nx = (m + 2) * (m + 2);
  PFST_CHECK_ARG((i64)K * T * 4 < (1ll << 31) && (i64)M * T * 4 < (1ll << 31) && (i64)K * M * 6 < (1ll << 31));
  hipStream_t s = (hipStream_t)stream;
  if (M > 64) return launch_split<128>(V, (i64)K * T, U6, nullptr, Mbuf, (i64)M * T, N, K, 1, T, M, 1, T, 1, 1, 1, 0, 1, 0, nullptr, 0, nx, s);
  if (M > 32) return launch_split<64>(V, (i64)K * T, U6, nullptr, Mbuf, (i64)M * T, N, K, 1, T, M, 1, T, 1, 1, 1, 0, 1, 0, nullptr, 0, nx, s);
  return launch_split<32>(V, (i64)K * T, U6, nullptr, Mbuf, (i64)M * T, N, K, 1, T, M, 1, T, 1, 1, 1, 0, 1, 0, nullptr, 0, nx, s);
}

// plain[(m+2)^2][Cout][Cin] transform-domain filters (pfst_wino_filter_plain: normal and flipped) -> split-packed sets of 6*Cout*Cin bytes
extern "C" int pfst_wino_pack_weight_split(const float* plain_f, const float* plain_d, void* U6_fprop, void* U6_dgrad, int Cout, int Cin,
                                           int m, pfst_stream_t stream) {
  PFST_CHECK_ARG(((plain_f && U6_fprop) || (plain_d && U6_dgrad)) && (m == 2 || m == 4));
  PFST_CHECK_ARG(Cout > 0 && Cin > 0 && (!U6_fprop || Cin % 16 == 0) && (!U6_dgrad || Cout % 16 == 0));
  const i64 n = (i64)Cout * Cin, set = 6 * n;
  const int nx = (m + 2) * (m + 2);
  // one launch per layout for all (m+2)^2 sets (was one per set: 72 small launches per layer and step)
  int gx = ew_grid(n / 8 + 1);
  if (gx > 4096) gx = 4096;
  if (U6_fprop) {
    hipLaunchKernelGGL(pack_weight_split_kernel, dim3(gx, nx), dim3(256), 0, (hipStream_t)stream, plain_f, (uint4*)U6_fprop, (uint4*)nullptr, Cout,
                       Cin, 1, n, set / 16);
    PFST_CHECK_LAUNCH();
  }
  if (U6_dgrad) {
    hipLaunchKernelGGL(pack_weight_split_kernel, dim3(gx, nx), dim3(256), 0, (hipStream_t)stream, plain_d, (uint4*)nullptr, (uint4*)U6_dgrad, Cout,
                       Cin, 1, n, set / 16);
    PFST_CHECK_LAUNCH();
  }
  return PFST_OK;
}
