// Weight gradient, "K-quad" fast path (stride 1, rows of 4-pixel quads):
//   dW[m][j] += sum_{n,p} dY[n][m][p] * X[n][ci(j)][src(p, tap(j))],  j = ci*T + tap        GEMM M = Cout, N = Cin*T, K = pixels
// Both operands are K(pixel)-contiguous in memory, so the whole staging path moves 16-byte quads of 4 consecutive pixels:
//   global  : one buffer_load_dwordx4 per (row, quad)          (4x fewer VMEM instructions and address VALU than dword gathers)
//   LDS     : image [quad][row (+ pad)] of float4, planes padded by 8/NQ rows -> conflict-free ds_write_b128 / ds_read_b128,
//             every address = per-thread base + immediate
//   MFMA    : v_mfma_f32_32x32x2_f32; lane (row l31, half lh) holds quad 2g+lh of its row, and MFMA #e of group g consumes
//             element e of both operands: k = 8g + 4*lh + e.  The K sum is order-free, so A and B only have to agree.
// 3x3 taps shift the source quad by dx = tx*dil - pad (4-byte-aligned dwordx4 loads).  A K-step is 16 pixels of one output
// row, walked on the scalar unit: interior steps use constant voffsets + a scalar soffset (no per-lane arithmetic); the
// first/last step of a row and the top/bottom `dil` rows load element-wise with out-of-range offsets for the padding.
// Split-K over images and pixel chunks, fp32 atomics into the flat gradient arena -- as conv_wgrad_kernel (conv_mfma.hip),
// which remains the generic path (stride 2, ragged widths).
#include "common.h"
#include "det.h"
#include "amax.h"
#include "../../include/pfst_hip.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

int g_wgrad_lds_pad = 0;                  // pfst_conv_wgrad_set_lds_pad


constexpr int QBJ = 128;

template <int BM, int T, int WBK>
__global__ __launch_bounds__(256) void conv_wgrad_q_kernel(
    const float* __restrict__ x, i64 x_bs, const float* __restrict__ dy, i64 dy_bs, float* __restrict__ dw,
    int Cin, int Hi, int Wi, int M, int Ho, int Wo, int dil, int pad, int chunks, int chunk_len, int N, i64 x_gs, i64 dy_gs, i64 dw_gs,
    int gx, int gy, int gz, int xcd_order) {
  constexpr int WM = BM >= 64 ? 64 : 32;
  constexpr int WAVES_M = BM / WM;
  constexpr int WAVES_N = 4 / WAVES_M;
  constexpr int WN = QBJ / WAVES_N;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int NQ = WBK / 4;                 // quads per K-step
  constexpr int RPP = 256 / NQ;               // tile rows staged per pass
  constexpr int A_N = (BM + RPP - 1) / RPP, B_N = QBJ / RPP;
  constexpr int SW = 8 / NQ;                  // plane padding: the 8 lanes of a ds_write_b128 group hit 8 distinct 16-B slots
  constexpr int PA = BM + SW, PB = QBJ + SW;  // rows per quad plane of the A / B image
  constexpr int KS = T == 9 ? 3 : 1;
  constexpr unsigned OOB = 0x80000000u;

  __shared__ float4 As[2][NQ * PA];
  __shared__ float4 Bs[2][NQ * PB];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm0 = (wid / WAVES_N) * WM, wn0 = (wid % WAVES_N) * WN;
  const int P = Ho * Wo, HiWi = Hi * Wi, J = Cin * T;
  // Workgroup -> (j tile, m tile, K slice).  The gx*gy tiles of ONE K slice z (an image's pixel chunk) read the same dy rows
  // (shared along j) and x rows (shared along m): (M + J) * chunk_len * 4 bytes in all, against gx*gy times that if every tile
  // fetched its own.  Workgroups are dealt round-robin over the 8 XCDs (MI355X_MICROARCH.md), each with its own L2, so the
  // linear id is decoded such that all tiles of a slice land on ONE XCD (id % 8) and run there back to back: slice z = 8 s + xcd.
  // (The plain 3-D grid put the tiles of a slice on all 8 XCDs: every L2 fetched every operand row.)  The last gz % 8 slices
  // keep the plain order.
  int bx, by, bz;
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
  const int j0 = bx * QBJ, m0 = by * BM;
  // bz = ((group * N) + image) * chunks + chunk; groups > 1: batched products that share shapes (the 16 transform
  // indices of the Winograd weight gradient), each with its own x / dy / dw
  const int ng = bz / chunks, chunk = bz - ng * chunks;
  const int grp = ng / N, n = ng - grp * N;
  const int pbeg = chunk * chunk_len;
  const int pend = min(P, pbeg + chunk_len);
  if (pbeg >= pend) return;
  x += (i64)grp * x_gs + (i64)n * x_bs;
  dy += (i64)grp * dy_gs + (i64)n * dy_bs;
  dw += dw_gs >= 0 ? (i64)grp * dw_gs : (i64)bz * -dw_gs;      // < 0: deterministic mode, one scratch tile-set per grid slice (det.h)
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy), 0, M * P * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, Cin * HiWi * 4, 0x00020000);

  // staging role: quad q of rows r0 + RPP*i (4 x NQ consecutive lanes read one row's WBK*4 contiguous bytes)
  const int q = tid % NQ, r0 = tid / NQ;
  unsigned a_voff[A_N], b_voff[B_N];
  int j_coff[B_N], j_dy[B_N], j_dx[B_N];
#pragma unroll
  for (int i = 0; i < A_N; ++i) {
    const int r = r0 + RPP * i;
    const int m = m0 + r;
    a_voff[i] = (r < BM && m < M) ? 4u * ((unsigned)m * (unsigned)P + 4u * q) : OOB;
  }
#pragma unroll
  for (int i = 0; i < B_N; ++i) {
    const int j = j0 + r0 + RPP * i;
    j_coff[i] = -1; j_dy[i] = 0; j_dx[i] = 0;
    b_voff[i] = OOB;
    if (j < J) {
      if (T == 1) {
        b_voff[i] = 4u * ((unsigned)j * (unsigned)HiWi + 4u * q);
      } else {
        const int ci = j / T, tap = j - ci * T;
        const int ty = tap / KS, tx = tap - ty * KS;
        j_coff[i] = ci * HiWi;
        j_dy[i] = ty * dil - pad;
        j_dx[i] = tx * dil - pad;
      }
    }
  }

  float4 areg[A_N], breg[B_N];
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // 3x3: a K-step is 16 consecutive pixels of ONE output row (Wo % 16 == 0), so (oy0, ox0) are wave-uniform and advance
  // on the scalar unit.  Interior steps (not the first/last step of a row, not within `dil` rows of the top/bottom) need no
  // per-lane address or validity arithmetic at all: constant voffsets (biased by +dil rows/cols so they are never negative:
  // the hardware range check looks at the voffset alone) plus a scalar soffset.  Edge steps take the general path.
  int oy0 = 0, ox0 = 0;
  if (T != 1) {
    oy0 = pbeg / Wo;
    ox0 = pbeg - oy0 * Wo;
#pragma unroll
    for (int i = 0; i < B_N; ++i)
      b_voff[i] = j_coff[i] >= 0 ? 4u * (unsigned)(j_coff[i] + (j_dy[i] + dil) * Wi + (j_dx[i] + dil) + 4 * q) : OOB;
  }
  const int bias_px = dil * Wi + dil;

  auto ld4 = [](const __amdgpu_buffer_rsrc_t& rs, unsigned vo, int so) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, 0));
  };
  auto load_tile = [&](int pk0) {
    const int p = pk0 + 4 * q;                 // P % 4 == 0 and chunk_len % WBK == 0: a quad is entirely in or out
    const bool pv = p < pend;
    const int soff = pk0 * 4;
#pragma unroll
    for (int i = 0; i < A_N; ++i) areg[i] = ld4(a_rsrc, pv ? a_voff[i] : OOB, soff);
    if (T == 1) {
#pragma unroll
      for (int i = 0; i < B_N; ++i) breg[i] = ld4(b_rsrc, pv ? b_voff[i] : OOB, soff);
    } else {
      const bool interior = (ox0 >= 16) & (ox0 + 32 <= Wo) & (oy0 >= dil) & (oy0 + dil < Hi) & (pk0 + WBK <= pend);
      if (interior) {
        const int so = (oy0 * Wi + ox0 - bias_px) * 4;
#pragma unroll
        for (int i = 0; i < B_N; ++i) breg[i] = ld4(b_rsrc, b_voff[i], so);
      } else {                                 // edge step: element-wise dword loads, padding / ragged ends via OOB offsets
        const int ox = ox0 + 4 * q;
#pragma unroll
        for (int i = 0; i < B_N; ++i) {
          const int sy = oy0 + j_dy[i], sx0 = ox + j_dx[i];
          const bool rowok = pv & (j_coff[i] >= 0) & ((unsigned)sy < (unsigned)Hi);
          const int base = j_coff[i] + sy * Wi + sx0;
          float e[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const bool ok = rowok & ((unsigned)(sx0 + k) < (unsigned)Wi);
            e[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b_rsrc, ok ? 4u * (unsigned)(base + k) : OOB, 0, 0));
          }
          breg[i] = make_float4(e[0], e[1], e[2], e[3]);
        }
      }
      ox0 += WBK;                              // scalar walk over the output rows
      if (ox0 >= Wo) { ox0 = 0; oy0 += 1; }
    }
  };
  const int l31 = lane & 31, lh = lane >> 5;
  // LDS pointers of this thread: every access is base + immediate; the double-buffer flip is one add per pointer and K-step
  const float4* a_rd = &As[0][lh * PA + wm0 + l31];     // + 2g * PA + 32 i
  const float4* b_rd = &Bs[0][lh * PB + wn0 + l31];     // + 2g * PB + 32 j
  float4* a_wr = &As[0][q * PA + r0];                   // + RPP i
  float4* b_wr = &Bs[0][q * PB + r0];
  int da = NQ * PA, db = NQ * PB;
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < A_N; ++i)
      if (A_N * RPP == BM || r0 + RPP * i < BM) a_wr[RPP * i] = areg[i];
#pragma unroll
    for (int i = 0; i < B_N; ++i) b_wr[RPP * i] = breg[i];
  };
  auto mma_step = [&]() {
#pragma unroll
    for (int g = 0; g < WBK / 8; ++g) {
      float4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = a_rd[2 * g * PA + 32 * i];
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = b_rd[2 * g * PB + 32 * j];
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const float a = e == 0 ? af[i].x : e == 1 ? af[i].y : e == 2 ? af[i].z : af[i].w;
            const float b = e == 0 ? bf[j].x : e == 1 ? bf[j].y : e == 2 ? bf[j].z : bf[j].w;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i][j], 0, 0, 0);
          }
    }
  };

  const int KT = (pend - pbeg + WBK - 1) / WBK;
  load_tile(pbeg);
  store_tile();
  __syncthreads();
  a_wr += da; b_wr += db;
  for (int kt = 0; kt + 1 < KT; ++kt) {             // steady state: prefetch K-step kt+1, multiply step kt
    load_tile(pbeg + (kt + 1) * WBK);
    mma_step();
    store_tile();
    __syncthreads();
    a_rd += da; b_rd += db; a_wr -= da; b_wr -= db;
    da = -da; db = -db;
  }
  mma_step();                                       // last K-step
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

// ---- the same kernel on the fp16 matrix cores with the f16x3 arithmetic (conv_f16x3.hip: two scaled fp16 pieces per operand, three MFMAs
// per product, scales from the tensors' slot groups).  Both operands are activations, so both are split as they are staged: a 4-pixel quad
// becomes 4 + 4 halfs, and the LDS image [piece][k-half][row] of 16-byte fragments (8 consecutive pixels of a row = two quads) is what
// v_mfma_f32_32x32x16_f16 consumes -- lane (row l31, half lh) holds pixels 8 lh .. 8 lh + 7 of the 16-pixel K-step for A and B alike.
// One K-step costs 3 MFMAs of 32 cycles per 32x32 block instead of 8 fp32-input MFMAs of 64: the layers this serves (direct 3x3 of
// stem / layer1, 1x1 with <= 64 output channels) ran at 73-95 TF-eq on the fp32 pipe.
template <int BM, int T>
__global__ __launch_bounds__(256) void conv_wgrad_q16_kernel(
    const float* __restrict__ x, i64 x_bs, const float* __restrict__ dy, i64 dy_bs, float* __restrict__ dw,
    int Cin, int Hi, int Wi, int M, int Ho, int Wo, int dil, int pad, int chunks, int chunk_len, int N, i64 x_gs, i64 dy_gs, i64 dw_gs,
    int gx, int gy, int gz, int xcd_order, const float* __restrict__ x_amax, const float* __restrict__ dy_amax) {
  constexpr int WBK = 16;
  constexpr int WM = BM >= 64 ? 64 : 32;
  constexpr int WAVES_M = BM / WM;
  constexpr int WAVES_N = 4 / WAVES_M;
  constexpr int WN = QBJ / WAVES_N;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int NQ = WBK / 4;                 // quads per K-step
  constexpr int RPP = 256 / NQ;               // tile rows staged per pass
  constexpr int A_N = (BM + RPP - 1) / RPP, B_N = QBJ / RPP;
  constexpr int SW = 2;                       // plane padding
  constexpr int PA = BM + SW, PB = QBJ + SW;  // rows per quad plane of the A / B image
  constexpr int KS = T == 9 ? 3 : 1;
  constexpr unsigned OOB = 0x80000000u;

  __shared__ uint4 As[2][4 * PA];             // [buffer][(piece 2 x k-half 2) planes][row]: 8 halfs per slot
  __shared__ uint4 Bs[2][4 * PB];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm0 = (wid / WAVES_N) * WM, wn0 = (wid % WAVES_N) * WN;
  const int P = Ho * Wo, HiWi = Hi * Wi, J = Cin * T;
  // Workgroup -> (j tile, m tile, K slice).  The gx*gy tiles of ONE K slice z (an image's pixel chunk) read the same dy rows
  // (shared along j) and x rows (shared along m): (M + J) * chunk_len * 4 bytes in all, against gx*gy times that if every tile
  // fetched its own.  Workgroups are dealt round-robin over the 8 XCDs (MI355X_MICROARCH.md), each with its own L2, so the
  // linear id is decoded such that all tiles of a slice land on ONE XCD (id % 8) and run there back to back: slice z = 8 s + xcd.
  // (The plain 3-D grid put the tiles of a slice on all 8 XCDs: every L2 fetched every operand row.)  The last gz % 8 slices
  // keep the plain order.
  int bx, by, bz;
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
  const int j0 = bx * QBJ, m0 = by * BM;
  // bz = ((group * N) + image) * chunks + chunk; groups > 1: batched products that share shapes (the 16 transform
  // indices of the Winograd weight gradient), each with its own x / dy / dw
  const int ng = bz / chunks, chunk = bz - ng * chunks;
  const int grp = ng / N, n = ng - grp * N;
  const int pbeg = chunk * chunk_len;
  const int pend = min(P, pbeg + chunk_len);
  if (pbeg >= pend) return;
  x += (i64)grp * x_gs + (i64)n * x_bs;
  dy += (i64)grp * dy_gs + (i64)n * dy_bs;
  dw += dw_gs >= 0 ? (i64)grp * dw_gs : (i64)bz * -dw_gs;      // < 0: deterministic mode, one scratch tile-set per grid slice (det.h)
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy), 0, M * P * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, Cin * HiWi * 4, 0x00020000);

  const int ea = amax_exponent(amax_read(dy_amax)), eb = amax_exponent(amax_read(x_amax));
  const float sa = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, scale_of(ea))));       // SGPRs
  const float sb = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, scale_of(eb))));
  // staging role: quad q of rows r0 + RPP*i (4 x NQ consecutive lanes read one row's WBK*4 contiguous bytes)
  const int q = tid % NQ, r0 = tid / NQ;
  unsigned a_voff[A_N], b_voff[B_N];
  int j_coff[B_N], j_dy[B_N], j_dx[B_N];
#pragma unroll
  for (int i = 0; i < A_N; ++i) {
    const int r = r0 + RPP * i;
    const int m = m0 + r;
    a_voff[i] = (r < BM && m < M) ? 4u * ((unsigned)m * (unsigned)P + 4u * q) : OOB;
  }
#pragma unroll
  for (int i = 0; i < B_N; ++i) {
    const int j = j0 + r0 + RPP * i;
    j_coff[i] = -1; j_dy[i] = 0; j_dx[i] = 0;
    b_voff[i] = OOB;
    if (j < J) {
      if (T == 1) {
        b_voff[i] = 4u * ((unsigned)j * (unsigned)HiWi + 4u * q);
      } else {
        const int ci = j / T, tap = j - ci * T;
        const int ty = tap / KS, tx = tap - ty * KS;
        j_coff[i] = ci * HiWi;
        j_dy[i] = ty * dil - pad;
        j_dx[i] = tx * dil - pad;
      }
    }
  }

  float4 areg[A_N], breg[B_N];
  f32x16 acc[TM][TN];
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // 3x3: a K-step is 16 consecutive pixels of ONE output row (Wo % 16 == 0), so (oy0, ox0) are wave-uniform and advance
  // on the scalar unit.  Interior steps (not the first/last step of a row, not within `dil` rows of the top/bottom) need no
  // per-lane address or validity arithmetic at all: constant voffsets (biased by +dil rows/cols so they are never negative:
  // the hardware range check looks at the voffset alone) plus a scalar soffset.  Edge steps take the general path.
  int oy0 = 0, ox0 = 0;
  if (T != 1) {
    oy0 = pbeg / Wo;
    ox0 = pbeg - oy0 * Wo;
#pragma unroll
    for (int i = 0; i < B_N; ++i)
      b_voff[i] = j_coff[i] >= 0 ? 4u * (unsigned)(j_coff[i] + (j_dy[i] + dil) * Wi + (j_dx[i] + dil) + 4 * q) : OOB;
  }
  const int bias_px = dil * Wi + dil;

  auto ld4 = [](const __amdgpu_buffer_rsrc_t& rs, unsigned vo, int so) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, 0));
  };
  auto load_tile = [&](int pk0) {
    const int p = pk0 + 4 * q;                 // P % 4 == 0 and chunk_len % WBK == 0: a quad is entirely in or out
    const bool pv = p < pend;
    const int soff = pk0 * 4;
#pragma unroll
    for (int i = 0; i < A_N; ++i) areg[i] = ld4(a_rsrc, pv ? a_voff[i] : OOB, soff);
    if (T == 1) {
#pragma unroll
      for (int i = 0; i < B_N; ++i) breg[i] = ld4(b_rsrc, pv ? b_voff[i] : OOB, soff);
    } else {
      const bool interior = (ox0 >= 16) & (ox0 + 32 <= Wo) & (oy0 >= dil) & (oy0 + dil < Hi) & (pk0 + WBK <= pend);
      if (interior) {
        const int so = (oy0 * Wi + ox0 - bias_px) * 4;
#pragma unroll
        for (int i = 0; i < B_N; ++i) breg[i] = ld4(b_rsrc, b_voff[i], so);
      } else {                                 // edge step: element-wise dword loads, padding / ragged ends via OOB offsets
        const int ox = ox0 + 4 * q;
#pragma unroll
        for (int i = 0; i < B_N; ++i) {
          const int sy = oy0 + j_dy[i], sx0 = ox + j_dx[i];
          const bool rowok = pv & (j_coff[i] >= 0) & ((unsigned)sy < (unsigned)Hi);
          const int base = j_coff[i] + sy * Wi + sx0;
          float e[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const bool ok = rowok & ((unsigned)(sx0 + k) < (unsigned)Wi);
            e[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b_rsrc, ok ? 4u * (unsigned)(base + k) : OOB, 0, 0));
          }
          breg[i] = make_float4(e[0], e[1], e[2], e[3]);
        }
      }
      ox0 += WBK;                              // scalar walk over the output rows
      if (ox0 >= Wo) { ox0 = 0; oy0 += 1; }
    }
  };
  const int l31 = lane & 31, lh = lane >> 5;
  // LDS pointers of this thread: fragment reads at base + immediate; a staged quad is the (q & 1) half of the 16-byte fragment
  // (k-half q >> 1) of its row, once per piece; the double-buffer flip is one add per pointer and K-step
  const uint4* a_rd = &As[0][lh * PA + wm0 + l31];      // + pl * 2 * PA + 32 i
  const uint4* b_rd = &Bs[0][lh * PB + wn0 + l31];      // + pl * 2 * PB + 32 j
  uint2* a_wr = reinterpret_cast<uint2*>(&As[0][(q >> 1) * PA + r0]) + (q & 1);        // + 2 RPP i (rows), + 4 PA (piece l, in uint2 units)
  uint2* b_wr = reinterpret_cast<uint2*>(&Bs[0][(q >> 1) * PB + r0]) + (q & 1);
  int da = 4 * PA, db = 4 * PB;                          // buffer stride in uint4 (reads); writes: 2x in uint2
  // h = f16(v s), l = f16(v s - h): the f16x3 pieces, four instructions per pair of values as in the GEMM loops (conv_f16x3.hip split_op_f16:
  // v_fma_mixlo / mixhi_f16 round fma(x, s, c) once into the low / high half; bit-identical to amax.h pack_f16x2_pieces)
  auto split_pair = [](float a, float b, float s, unsigned& h, unsigned& l) {
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(h) : "v"(a), "s"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(h) : "v"(b), "s"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l) : "v"(a), "s"(s), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(b), "s"(s), "v"(h));
  };
  auto split_quad = [&](const float4& v, float s, uint2& h, uint2& l) {
    split_pair(v.x, v.y, s, h.x, l.x);
    split_pair(v.z, v.w, s, h.y, l.y);
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < A_N; ++i)
      if (A_N * RPP == BM || r0 + RPP * i < BM) {
        uint2 h, l;
        split_quad(areg[i], sa, h, l);
        a_wr[2 * RPP * i] = h;
        a_wr[2 * RPP * i + 4 * PA] = l;
      }
#pragma unroll
    for (int i = 0; i < B_N; ++i) {
      uint2 h, l;
      split_quad(breg[i], sb, h, l);
      b_wr[2 * RPP * i] = h;
      b_wr[2 * RPP * i + 4 * PB] = l;
    }
  };
  auto mma_step = [&]() {
    f16x8 af[TM][2], bf[TN][2];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i][pl] = __builtin_bit_cast(f16x8, a_rd[pl * 2 * PA + 32 * i]);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j][pl] = __builtin_bit_cast(f16x8, b_rd[pl * 2 * PB + 32 * j]);
    }
    // al bh | ah bl | ah bh: smallest terms first
#pragma unroll
    for (int prod = 0; prod < 3; ++prod)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][prod == 0 ? 1 : 0], bf[j][prod == 1 ? 1 : 0], acc[i][j], 0, 0, 0);
  };

  const int KT = (pend - pbeg + WBK - 1) / WBK;
  load_tile(pbeg);
  store_tile();
  __syncthreads();
  a_wr += 2 * da; b_wr += 2 * db;
  for (int kt = 0; kt + 1 < KT; ++kt) {             // steady state: prefetch K-step kt+1, multiply step kt
    load_tile(pbeg + (kt + 1) * WBK);
    mma_step();
    store_tile();
    __syncthreads();
    a_rd += da; b_rd += db; a_wr -= 2 * da; b_wr -= 2 * db;
    da = -da; db = -db;
  }
  mma_step();                                       // last K-step
  const float ua = unscale_of(ea), ub = unscale_of(eb);       // exact powers of two, in two factors so that neither over- nor underflows
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int jj = j0 + wn0 + j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < M && jj < J) atomicAdd(&dw[(i64)m * J + jj], acc[i][j][r] * ua * ub);
      }
    }
  }
}

template <int BM, int T, int WBK>
int launch_q(const float* x, i64 x_bs, const float* dy, i64 dy_bs, float* dw, int N, int Cin, int Hi, int Wi, int M,
             int Ho, int Wo, int dil, int pad, int groups, i64 x_gs, i64 dy_gs, i64 dw_gs, hipStream_t s) {
  const int P = Ho * Wo, J = Cin * T;
  const int tiles = cdiv(J, QBJ) * cdiv(M, BM) * groups;
  // split-K chunking: whole rounds of resident blocks (see launch_wgrad_k in conv_mfma.hip); 32 KB LDS at WBK = 16, 64 KB at 32
  const double slots = 256.0 * (WBK == 16 ? 4 : 2);
  int chunks = 1;
  double best = -1.0;
  for (int c = 1; c <= 64 && (c == 1 || P / c >= 512); ++c) {
    const double rounds = (double)tiles * N * c / slots;
    const double eff = rounds < 2.0 ? 0.45 * rounds : rounds / ceil(rounds);
    if (eff > best + 0.02) { best = eff; chunks = c; }
    if (eff >= 0.93) break;
  }
  int chunk_len = ((cdiv(P, chunks) + WBK - 1) / WBK) * WBK;
  chunks = cdiv(P, chunk_len);
  constexpr int xcd_env = 1;
  const int gx = cdiv(J, QBJ), gy = cdiv(M, BM), gz = N * groups * chunks;
  PFST_CHECK_ARG((i64)gx * gy * gz < (1ll << 31));
  dim3 grid(gx * gy * gz);
  // occupancy cap (pfst_conv_wgrad_set_lds_pad): unused dynamic LDS, so that fewer workgroups fit per CU and a concurrently running
  // HBM-bound kernel of another stream finds registers and wave slots
  const i64 elems = (i64)M * J;
  bool det_ok;
  float* const ws = wgrad_det_scratch(elems, gz, s, det_ok);          // deterministic mode (det.h): one scratch tile-set per grid slice
  PFST_CHECK_ARG(det_ok);
  hipLaunchKernelGGL((conv_wgrad_q_kernel<BM, T, WBK>), grid, dim3(256), g_wgrad_lds_pad, s, x, x_bs, dy, dy_bs, ws ? ws : dw, Cin, Hi, Wi, M, Ho, Wo,
                     dil, pad, chunks, chunk_len, N, x_gs, dy_gs, ws ? -elems : dw_gs, gx, gy, gz, xcd_env);
  if (ws) wgrad_det_reduce(ws, dw, elems, groups, N * chunks, dw_gs, s);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

template <int BM, int T>
int launch_q_bk(const float* x, i64 x_bs, const float* dy, i64 dy_bs, float* dw, int N, int Cin, int Hi, int Wi, int M,
                int Ho, int Wo, int dil, int pad, int groups, i64 x_gs, i64 dy_gs, i64 dw_gs, hipStream_t s) {
  return launch_q<BM, T, 16>(x, x_bs, dy, dy_bs, dw, N, Cin, Hi, Wi, M, Ho, Wo, dil, pad, groups, x_gs, dy_gs, dw_gs, s);
}

template <int BM, int T>
int launch_q16(const float* x, i64 x_bs, const float* dy, i64 dy_bs, float* dw, int N, int Cin, int Hi, int Wi, int M, int Ho, int Wo, int dil,
               int pad, const float* x_amax, const float* dy_amax, hipStream_t s) {
  constexpr int WBK = 16;
  const int P = Ho * Wo, J = Cin * T;
  const int tiles = cdiv(J, QBJ) * cdiv(M, BM);
  const double slots = 256.0 * 4;                // as launch_q at WBK = 16 (same LDS footprint)
  int chunks = 1;
  double best = -1.0;
  for (int c = 1; c <= 64 && (c == 1 || P / c >= 512); ++c) {
    const double rounds = (double)tiles * N * c / slots;
    const double eff = rounds < 2.0 ? 0.45 * rounds : rounds / ceil(rounds);
    if (eff > best + 0.02) { best = eff; chunks = c; }
    if (eff >= 0.93) break;
  }
  int chunk_len = ((cdiv(P, chunks) + WBK - 1) / WBK) * WBK;
  chunks = cdiv(P, chunk_len);
  constexpr int xcd_env = 1;
  const int gx = cdiv(J, QBJ), gy = cdiv(M, BM), gz = N * chunks;
  PFST_CHECK_ARG((i64)gx * gy * gz < (1ll << 31));
  const i64 elems = (i64)M * J;
  bool det_ok;
  float* const ws = wgrad_det_scratch(elems, gz, s, det_ok);          // deterministic mode (det.h): one scratch tile-set per grid slice
  PFST_CHECK_ARG(det_ok);
  hipLaunchKernelGGL((conv_wgrad_q16_kernel<BM, T>), dim3(gx * gy * gz), dim3(256), g_wgrad_lds_pad, s, x, x_bs, dy, dy_bs, ws ? ws : dw, Cin, Hi, Wi, M, Ho,
                     Wo, dil, pad, chunks, chunk_len, N, (i64)0, (i64)0, ws ? -elems : (i64)0, gx, gy, gz, xcd_env, x_amax, dy_amax);
  if (ws) wgrad_det_reduce(ws, dw, elems, 1, N * chunks, 0, s);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

}  // namespace

// dw += dL/dw of a stride-1 'same' convolution (1x1, or 3x3 with pad == dil) with the f16x3 arithmetic on the K-quad kernel: the layers the
// whole-line kernel (pfst_conv_wgrad_f16x3: 1x1, > 64 output channels) does not take.  x_amax / dy_amax: slot groups with max |x|, max |dy|.
extern "C" int pfst_conv_wgrad_f16x3_q(const float* x, long long x_bs, const float* dy, long long dy_bs, float* dw, int N, int Cin, int H, int W,
                                       int Cout, int ksize, int dil, const float* x_amax, const float* dy_amax, pfst_stream_t stream) {
  PFST_CHECK_ARG(x && dy && dw && x_amax && dy_amax && N > 0 && N <= 65535 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && dil >= 1);
  PFST_CHECK_ARG(ksize == 1 || ksize == 3);
  PFST_CHECK_ARG(x_bs >= (i64)Cin * H * W && dy_bs >= (i64)Cout * H * W && (i64)Cin * H * W * 4 < (1ll << 31) && (i64)Cout * H * W * 4 < (1ll << 31));
  if (!pfst_wgrad_q_eligible(x, x_bs, dy, dy_bs, H, W, H, W, ksize, 1, dil)) {
    pfst_set_error(__FILE__, __LINE__, "f16x3 K-quad weight gradient needs 16-byte aligned planes and HW % 4 == 0 (1x1) / W % 16 == 0, dil <= 8 (3x3)");
    return PFST_ERR_UNSUPPORTED;
  }
  hipStream_t s = (hipStream_t)stream;
  const int pad = ksize == 3 ? dil : 0;
#define PFST_WGQ16(BM_)                                                                                                         \
  return ksize == 3 ? launch_q16<BM_, 9>(x, x_bs, dy, dy_bs, dw, N, Cin, H, W, Cout, H, W, dil, pad, x_amax, dy_amax, s)         \
                    : launch_q16<BM_, 1>(x, x_bs, dy, dy_bs, dw, N, Cin, H, W, Cout, H, W, dil, pad, x_amax, dy_amax, s)
  if (Cout > 64) { PFST_WGQ16(128); }
  if (Cout > 32) { PFST_WGQ16(64); }
  PFST_WGQ16(32);
#undef PFST_WGQ16
}

// true if the quad path applies (the caller has validated the geometry already)
bool pfst_wgrad_q_eligible(const float* x, i64 x_bs, const float* dy, i64 dy_bs, int Hi, int Wi, int Ho, int Wo, int ksize, int stride, int dil) {
  if (stride != 1) return false;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) return false;
  if ((x_bs | dy_bs) & 3) return false;
  if (ksize == 1) return ((i64)Ho * Wo) % 4 == 0 && Hi == Ho && Wi == Wo;
  return Wo % 16 == 0 && Hi == Ho && Wi == Wo && dil <= 8;    // 'same' 3x3 (pad == dil); 16-pixel K-steps walk whole rows
}

int pfst_wgrad_q_launch(const float* x, i64 x_bs, const float* dy, i64 dy_bs, float* dw, int N, int Cin, int Hi, int Wi, int Cout,
                        int Ho, int Wo, int ksize, int dil, int pad, int groups, i64 x_gs, i64 dy_gs, i64 dw_gs, hipStream_t s) {
#define PFST_WGQ(BM_)                                                                                                                        \
  return ksize == 3 ? launch_q_bk<BM_, 9>(x, x_bs, dy, dy_bs, dw, N, Cin, Hi, Wi, Cout, Ho, Wo, dil, pad, groups, x_gs, dy_gs, dw_gs, s)      \
                    : launch_q_bk<BM_, 1>(x, x_bs, dy, dy_bs, dw, N, Cin, Hi, Wi, Cout, Ho, Wo, dil, pad, groups, x_gs, dy_gs, dw_gs, s)
  if (Cout > 64) { PFST_WGQ(128); }
  if (Cout > 32) { PFST_WGQ(64); }
  PFST_WGQ(32);
#undef PFST_WGQ
}

extern "C" int pfst_conv_wgrad_set_lds_pad(int bytes) {
  PFST_CHECK_ARG(bytes >= 0 && bytes <= 96 * 1024);
  g_wgrad_lds_pad = bytes;
  return PFST_OK;
}
