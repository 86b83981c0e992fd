// Dense convolution (groups = 1) as implicit GEMM on the CDNA4 fp32 matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate -- keeps the reference's
// fp32 numerics; gfx950 has no xf32/TF32 path).
//
// Reference ops replaced: every F.conv2d with groups=1 on the PFST hot path
// (rsiseg/models/backbones/resnet.py:169-209,593-624, decode_heads/aspp_head.py:32-42,85-92,
// fcn_head.py:40-49, decode_head.py:242-247) and their autograd data / weight gradients.
//
// This file: the C entry points, weight packing, and the GENERIC kernels -- any channel count (the 3- / 10-band stems,
// K steps that straddle taps), stride 2, ragged widths.  The fast paths live in conv_igemm_q.hip (fprop / dgrad with
// Cin % 16 == 0), conv_wgrad_q.hip (stride-1 weight gradients) and conv_winograd.hip (wide 3x3 layers).
//
// Layout (NCHW fp32, per image):   OUT[m][p] = sum_k WK[k][m] * IN[c(k)][src(p, tap(k))]
//   GEMM M = output channels, N = output pixels (contiguous in memory -> coalesced), K = (tap, c).
//   WK is the weight re-packed K-major ([tap*C + c][M]) so both LDS tiles are filled by
//   row-contiguous, coalesced global reads and read back conflict-free (lane = column).
//   One 256-thread block (4 waves) computes a BM x 128 tile; each wave owns a 64x64 (or 32x32)
//   sub-tile as 32x32 MFMA accumulators; K advances 16 per step through a double-buffered LDS
//   tile with register-staged prefetch (one barrier per step).
#include "conv_epilogue.h"
#include "det.h"
#include "../../include/pfst_hip.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));


namespace {

constexpr int BK_MIN = 16;  // K granularity of the wave-uniform-tap fast path (Cin % 16 == 0)

// source coordinate of output index o for tap t: (o*a + t*b + c0) / div, div in {1,2}; branch-free validity
__device__ __forceinline__ bool src_coord(int o, int t, int a, int b, int c0, int div, int lim, int& s) {
  const int v = o * a + t * b + c0;
  const int odd = v & (div - 1);
  s = v >> (div >> 1);
  return (odd == 0) & (s >= 0) & (s < lim);
}

template <int BM>
__global__ __launch_bounds__(256) void conv_igemm_kernel(
    const float* __restrict__ in, i64 in_bs, const float* __restrict__ wk, const float* __restrict__ bias,
    float* __restrict__ out, i64 out_bs, int C, int Hi, int Wi, int M, int Ho, int Wo, int ks,
    int ca, int cb, int cc, int cdivv, int accumulate, float* __restrict__ stats, int stats_T) {
  constexpr int BK = 16, BN = 128;
  constexpr int WM = BM >= 64 ? 64 : 32;
  constexpr int WAVES_M = BM / WM;
  constexpr int WAVES_N = 4 / WAVES_M;
  constexpr int WN = BN / WAVES_N;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int A_RPP = 256 / BM;   // A rows loaded per pass
  constexpr int A_N = BK / A_RPP;   // A loads per thread per step
  constexpr int B_RPP = 256 / BN;   // B rows loaded per pass
  constexpr int B_N = BK / B_RPP;   // B loads per thread per step

  __shared__ float As[2][BK][BM];
  __shared__ float Bs[2][BK][BN];

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave id in an SGPR
  const int wm0 = (wid / WAVES_N) * WM, wn0 = (wid % WAVES_N) * WN;
  const int P = Ho * Wo, HiWi = Hi * Wi;
  // XCD-aware tile order (1-D grid over tiles): workgroups are dealt round-robin over the 8 XCDs, so the m-tiles that
  // share one pixel tile (= one activation tile) get ids 8 apart: they run back to back on the SAME XCD and reuse its
  // L2 copy instead of re-fetching the activations once per m-tile (measured: -26 % FETCH_SIZE, +1.6 % TFLOP/s).
  int bx, by;
  {
    const int gx = (P + BN - 1) / BN, gy = (M + BM - 1) / BM;
    const int lin = blockIdx.x;
    pfst_tile_order(lin, gx, gy, ks == 3, bx, by);
  }
  const int p0 = bx * BN, m0 = by * BM, n = blockIdx.z;
  const int K = C * ks * ks;
  const int KT = (K + BK - 1) / BK;
  in += (i64)n * in_bs;
  out += (i64)n * out_bs;

  const int bj = tid % BN, br0 = tid / BN;
  const int p = p0 + bj;
  const bool pvalid = p < P;
  const int oy = pvalid ? p / Wo : 0;
  const int ox = pvalid ? p - oy * Wo : 0;
  const int am = tid % BM, ar0 = tid / BM;
  const bool amvalid = (m0 + am) < M;

  float areg[A_N], breg[B_N];
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Generic loader: any (tap, channel) per K index, validity applied when the registers are written to LDS (clamped,
  // branch-free loads).  The fast kernels for Cin % 16 == 0 keep all of this off the vector pipe (conv_igemm_q.hip).
  const int amc = min(m0 + am, M - 1);
  unsigned a_mask = 0, b_mask = 0;
  auto load_tile = [&](int kt) {
    const int k0 = kt * BK;
    a_mask = 0; b_mask = 0;
#pragma unroll
    for (int i = 0; i < A_N; ++i) {
      const int k = k0 + ar0 + i * A_RPP;
      areg[i] = wk[(i64)min(k, K - 1) * M + amc];
      a_mask |= (unsigned)(amvalid && k < K) << i;
    }
#pragma unroll
    for (int i = 0; i < B_N; ++i) {
      const int k = k0 + br0 + B_RPP * i;
      const int kc = min(k, K - 1);
      const int tap = kc / C, ci = kc - tap * C;
      const int ty = tap / ks, tx = tap - ty * ks;
      int sy, sx;
      const bool ok = pvalid & (k < K) & src_coord(oy, ty, ca, cb, cc, cdivv, Hi, sy) & src_coord(ox, tx, ca, cb, cc, cdivv, Wi, sx);
      breg[i] = in[(i64)ci * HiWi + (ok ? sy * Wi + sx : 0)];
      b_mask |= (unsigned)ok << i;
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_N; ++i) As[buf][ar0 + i * A_RPP][am] = ((a_mask >> i) & 1u) ? areg[i] : 0.f;
#pragma unroll
    for (int i = 0; i < B_N; ++i) Bs[buf][br0 + B_RPP * i][bj] = ((b_mask >> i) & 1u) ? breg[i] : 0.f;
  };

  load_tile(0);
  store_tile(0);
  __syncthreads();

  const int l31 = lane & 31, lh = lane >> 5;
  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < KT) load_tile(kt + 1);
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const int krow = kk * 2 + lh;
      float af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = As[cur][krow][wm0 + i * 32 + l31];
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = Bs[cur][krow][wn0 + j * 32 + l31];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < KT) store_tile(cur ^ 1);
    __syncthreads();
  }

  conv_epilogue<TM, TN, WAVES_N, BN>(acc, out, bias, stats, stats_T, accumulate, M, P, m0, p0, wm0, wn0, bx, n, wid, lane);
}

// ---------------------------------------------------------------------------------------------
// weight gradient: dW[m][j] += sum_{n,p} dY[n][m][p] * X[n][ci(j)][src(p, tap(j))],  j = ci*T + tap
// GEMM M = Cout, N = Cin*T, K = pixels (split over images and pixel chunks; fp32 atomics combine).
// Both operands are K(pixel)-contiguous in memory: LDS tiles are [row][k] with a +1 pad.
// ---------------------------------------------------------------------------------------------
constexpr int WBJ = 128;

template <int BM, int T, int WBK>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(
    const float* __restrict__ x, i64 x_bs, const float* __restrict__ dy, i64 dy_bs, float* __restrict__ dw,
    int Cin, int Hi, int Wi, int M, int Ho, int Wo, int stride, int dil, int pad, int chunks, int chunk_len, i64 det_stride = 0) {
  constexpr int WM = BM >= 64 ? 64 : 32;
  constexpr int WAVES_M = BM / WM;
  constexpr int WAVES_N = 4 / WAVES_M;
  constexpr int WN = WBJ / WAVES_N;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int WLD = WBK + 1;
  constexpr int RPP = 256 / WBK;            // tile rows loaded per pass
  constexpr int A_N = BM / RPP, B_N = WBJ / RPP;
  constexpr int KS = T == 9 ? 3 : 1;

  extern __shared__ float smem[];
  float(*As)[BM][WLD] = reinterpret_cast<float(*)[BM][WLD]>(smem);
  float(*Bs)[WBJ][WLD] = reinterpret_cast<float(*)[WBJ][WLD]>(smem + 2 * BM * WLD);
  // per-column (j = ci*T + tap) gather descriptors, decoded ONCE per block: {channel offset, dy, dx}
  int* jtab = reinterpret_cast<int*>(smem + 2 * (BM + WBJ) * WLD);   // [WBJ][2]

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave id in an SGPR
  const int wm0 = (wid / WAVES_N) * WM, wn0 = (wid % WAVES_N) * WN;
  const int P = Ho * Wo, HiWi = Hi * Wi, J = Cin * T;
  const int j0 = blockIdx.x * WBJ, m0 = blockIdx.y * BM;
  const int n = blockIdx.z / chunks, chunk = blockIdx.z - n * chunks;
  const int pbeg = chunk * chunk_len;
  const int pend = min(P, pbeg + chunk_len);
  if (pbeg >= pend) return;
  x += (i64)n * x_bs;
  dy += (i64)n * dy_bs;
  dw += (i64)blockIdx.z * det_stride;         // deterministic mode (det.h): one scratch tile-set per grid slice

  if (tid < WBJ) {
    const int j = j0 + tid;
    int coff = -1, pk = 0;
    if (j < J) {
      const int ci = j / T, tap = j - ci * T;
      const int ty = tap / KS, tx = tap - ty * KS;
      coff = ci * HiWi;
      pk = ((ty * dil - pad) << 16) | ((tx * dil - pad) & 0xffff);
    }
    jtab[2 * tid] = coff;
    jtab[2 * tid + 1] = pk;
  }
  __syncthreads();

  const int kcol = tid % WBK, r0 = tid / WBK;
  float areg[A_N], breg[B_N];
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Buffer loads with hardware range check (see conv_igemm_kernel): rows / pixels / taps outside the problem use an
  // out-of-range byte offset and read as 0; the K(pixel) advance is a scalar soffset wherever the mapping allows it.
  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy), 0, M * P * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, Cin * HiWi * 4, 0x00020000);
  constexpr bool POINTWISE = (T == 1);
  const bool unit = POINTWISE && stride == 1 && pad == 0;       // input pixel == output pixel: constant offsets
  unsigned a_voff[A_N], b_voff[B_N];
#pragma unroll
  for (int i = 0; i < A_N; ++i) {
    const int m = m0 + r0 + RPP * i;
    a_voff[i] = m < M ? 4u * ((unsigned)m * (unsigned)P + (unsigned)kcol) : OOB;
  }
#pragma unroll
  for (int i = 0; i < B_N; ++i) {
    const int j = j0 + r0 + RPP * i;
    b_voff[i] = (unit && j < J) ? 4u * ((unsigned)j * (unsigned)HiWi + (unsigned)kcol) : OOB;
  }
  // this thread's 16 gather descriptors, kept in registers for the whole K loop
  int j_coff[B_N], j_dyx[B_N];
#pragma unroll
  for (int i = 0; i < B_N; ++i) {
    const int2 e = reinterpret_cast<const int2*>(jtab)[r0 + RPP * i];
    j_coff[i] = e.x;
    j_dyx[i] = e.y;
  }
  auto load_tile = [&](int pk0) {
    const int p = pk0 + kcol;
    const bool tail = pk0 + WBK > pend;           // wave-uniform: only the last step of the last chunk
    const int soff = pk0 * 4;
#pragma unroll
    for (int i = 0; i < A_N; ++i) {
      const unsigned vo = (tail && p >= pend) ? OOB : a_voff[i];
      areg[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(a_rsrc, vo, soff, 0));
    }
    if (unit) {
#pragma unroll
      for (int i = 0; i < B_N; ++i) {
        const unsigned vo = (tail && p >= pend) ? OOB : b_voff[i];
        breg[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b_rsrc, vo, soff, 0));
      }
    } else {
      const bool pv = p < pend;
      const int pc = pv ? p : pend - 1;
      const int oy = pc / Wo, ox = pc - oy * Wo;
      const int by = oy * stride, bx = ox * stride;
#pragma unroll
      for (int i = 0; i < B_N; ++i) {
        const int sy = by + (j_dyx[i] >> 16), sx = bx + (int)(short)(j_dyx[i] & 0xffff);
        const bool ok = pv & (j_coff[i] >= 0) & ((unsigned)sy < (unsigned)Hi) & ((unsigned)sx < (unsigned)Wi);
        const unsigned vo = ok ? 4u * (unsigned)(j_coff[i] + sy * Wi + sx) : OOB;
        breg[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b_rsrc, vo, 0, 0));
      }
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_N; ++i) As[buf][r0 + RPP * i][kcol] = areg[i];
#pragma unroll
    for (int i = 0; i < B_N; ++i) Bs[buf][r0 + RPP * i][kcol] = breg[i];
  };

  const int KT = (pend - pbeg + WBK - 1) / WBK;
  load_tile(pbeg);
  store_tile(0);
  __syncthreads();
  const int l31 = lane & 31, lh = lane >> 5;
  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < KT) load_tile(pbeg + (kt + 1) * WBK);
#pragma unroll
    for (int kk = 0; kk < WBK / 2; ++kk) {
      const int kc = kk * 2 + lh;
      float af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = As[cur][wm0 + i * 32 + l31][kc];
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = Bs[cur][wn0 + j * 32 + l31][kc];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
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

// K-major packings for the implicit GEMM.  Plain [k][m] when the channels per tap are not a multiple of 16 (generic kernel);
// otherwise the K-quad image [k/4][m][4] that conv_igemm_q.hip reads with one 16-byte load per (quad, row).
__device__ __forceinline__ i64 packed_index(i64 k, int m, int M, bool quad) {
  return quad ? (((k >> 2) * M + m) << 2) + (k & 3) : k * M + m;
}

__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wf, float* __restrict__ wd,
                                   int Cout, int Cin, int T) {
  const i64 total = (i64)Cout * Cin * T;
  const bool qf = (Cin % 16) == 0, qd = (Cout % 16) == 0;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
    if (wf) {  // i = (t*Cin + ci)*Cout + co : k = t*Cin + ci, row m = co
      const int co = (int)(i % Cout);
      const i64 k = i / Cout;
      const int ci = (int)(k % Cin), t = (int)(k / Cin);
      wf[packed_index(k, co, Cout, qf)] = w[((i64)co * Cin + ci) * T + t];
    }
    if (wd) {  // i = (t*Cout + co)*Cin + ci : k = t*Cout + co, row m = ci
      const int ci = (int)(i % Cin);
      const i64 k = i / Cin;
      const int co = (int)(k % Cout), t = (int)(k / Cout);
      wd[packed_index(k, ci, Cin, qd)] = w[((i64)co * Cin + ci) * T + t];
    }
  }
}

// grid: (splits over HW, C, N): one fp32 atomic per block
__global__ void bias_grad_kernel(const float* __restrict__ dy, i64 dy_bs, float* __restrict__ db, int C, int HW, int chunk) {
  __shared__ double sm[16];
  const int c = blockIdx.y, n = blockIdx.z;
  const float* p = dy + (i64)n * dy_bs + (i64)c * HW;
  const int beg = blockIdx.x * chunk, end = min(beg + chunk, HW);
  double s = 0.0;
  for (int i = beg + threadIdx.x; i < end; i += blockDim.x) s += (double)p[i];
  s = block_sum_d(s, sm);
  if (threadIdx.x == 0) atomicAdd(&db[c], (float)s);
}

template <int BM>
int launch_igemm_generic(const float* in, i64 in_bs, const float* wk, const float* bias, float* out, i64 out_bs, int N, int C,
                         int Hi, int Wi, int M, int Ho, int Wo, int ks, int a, int b, int c, int d, int acc, float* stats, int stats_T, hipStream_t s) {
  dim3 grid(cdiv((i64)Ho * Wo, 128) * cdiv(M, BM), 1, N);
  hipLaunchKernelGGL((conv_igemm_kernel<BM>), grid, dim3(256), 0, s, in, in_bs, wk, bias, out, out_bs, C, Hi, Wi, M,
                     Ho, Wo, ks, a, b, c, d, acc, stats, stats_T);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

template <int BM, int T, int WBK>
int launch_wgrad_k(const float* x, i64 x_bs, const float* dy, i64 dy_bs, float* dw, int N, int Cin, int Hi, int Wi, int M,
                   int Ho, int Wo, int stride, int dil, int pad, hipStream_t s) {
  const int P = Ho * Wo, J = Cin * T;
  const int tiles = cdiv(J, WBJ) * cdiv(M, BM);
  // Split the pixel (K) range (chunks of >= 512 pixels, fp32 atomics combine) so that the launch fills the chip in
  // whole "rounds": `slots` blocks are resident at once (LDS-limited: 4 per CU at WBK=16, 2 at WBK=32), and e.g.
  // 1152 blocks on 1024 slots leave 7/8 of the CUs idle for the last ninth of the work (measured -10 % on the 3x3
  // layers).  Take the smallest split whose last round is >= 90 % full, else the best one.
  const double slots = 256.0 * (WBK == 16 ? 4 : 2);
  int chunks = 1;
  double best = -1.0;
  for (int c = 1; c <= 64 && (c == 1 || P / c >= 512); ++c) {
    const double rounds = (double)tiles * N * c / slots;
    const double eff = rounds < 2.0 ? 0.45 * rounds : rounds / ceil(rounds);    // want >= 2 rounds: 1x1 layers measured best there
    if (eff > best + 0.02) { best = eff; chunks = c; }
    if (eff >= 0.93) break;
  }
  while ((i64)N * chunks > 65535 && chunks > 1) --chunks;       // gridDim.z limit
  int chunk_len = ((cdiv(P, chunks) + WBK - 1) / WBK) * WBK;
  chunks = cdiv(P, chunk_len);
  dim3 grid(cdiv(J, WBJ), cdiv(M, BM), N * chunks);
  const size_t lds = (size_t)2 * (BM + WBJ) * (WBK + 1) * sizeof(float) + (size_t)WBJ * 2 * sizeof(int);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<BM, T, WBK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  const i64 elems = (i64)M * J;
  bool det_ok;
  float* const ws = wgrad_det_scratch(elems, (i64)N * chunks, s, det_ok);          // deterministic mode (det.h)
  PFST_CHECK_ARG(det_ok);
  hipLaunchKernelGGL((conv_wgrad_kernel<BM, T, WBK>), grid, dim3(256), lds, s, x, x_bs, dy, dy_bs, ws ? ws : dw, Cin, Hi, Wi, M, Ho, Wo,
                     stride, dil, pad, chunks, chunk_len, ws ? elems : (i64)0);
  if (ws) wgrad_det_reduce(ws, dw, elems, 1, N * chunks, 0, s);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

template <int BM, int T>
int launch_wgrad(const float* x, i64 x_bs, const float* dy, i64 dy_bs, float* dw, int N, int Cin, int Hi, int Wi, int M,
                 int Ho, int Wo, int stride, int dil, int pad, hipStream_t s) {
  // measured (tools/conv_microbench.py): 3x3 gathers prefer the shallower K tile (4 blocks/CU: +7-10 %), 1x1 the deeper one
  const int wbk = T == 9 ? 16 : 32;
  if (wbk == 16) return launch_wgrad_k<BM, T, 16>(x, x_bs, dy, dy_bs, dw, N, Cin, Hi, Wi, M, Ho, Wo, stride, dil, pad, s);
  return launch_wgrad_k<BM, T, 32>(x, x_bs, dy, dy_bs, dw, N, Cin, Hi, Wi, M, Ho, Wo, stride, dil, pad, s);
}

}  // namespace

extern "C" int pfst_conv_pack_weight(const float* w, float* wk_fprop, float* wk_dgrad, int Cout, int Cin, int T, pfst_stream_t stream) {
  PFST_CHECK_ARG(w && (wk_fprop || wk_dgrad) && Cout > 0 && Cin > 0 && (T == 1 || T == 9));
  const i64 n = (i64)Cout * Cin * T;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, w, wk_fprop, wk_dgrad, Cout, Cin, T);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

extern "C" int pfst_conv_stats_slots(int M, int Ho, int Wo) {
  const int waves_n = M > 64 ? 2 : 4;             // WAVES_N of the BM = 128 / 64 / 32 tile variants
  return cdiv((i64)Ho * Wo, 128) * waves_n;
}

extern "C" int pfst_conv_igemm(const float* in, long long in_bs, const float* wk, const float* bias, float* out, long long out_bs,
                               int N, int C, int Hi, int Wi, int M, int Ho, int Wo, int ksize, int stride, int dil, int pad,
                               int mode, int accumulate, float* stats, const pfst_bnb_fuse_t* bnb, pfst_stream_t stream) {
  PFST_CHECK_ARG(in && wk && out && N > 0 && C > 0 && M > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0);
  PFST_CHECK_ARG(!bnb || ((C % BK_MIN) == 0 && bnb->x && bnb->x_bs >= (i64)M * Ho * Wo && (!bnb->y || bnb->y_bs >= (i64)M * Ho * Wo)));
  PFST_CHECK_ARG((ksize == 1 || ksize == 3) && (stride == 1 || stride == 2) && dil >= 1 && pad >= 0 && (mode == 0 || mode == 1));
  PFST_CHECK_ARG(in_bs >= (i64)C * Hi * Wi && out_bs >= (i64)M * Ho * Wo && N <= 65535);
  // 32-bit byte offsets inside one image / one weight tensor (buffer descriptors): fail loudly beyond 2 GiB
  PFST_CHECK_ARG((i64)C * Hi * Wi * 4 < (1ll << 31) && (i64)M * Ho * Wo * 4 < (1ll << 31) && (i64)C * ksize * ksize * M * 4 < (1ll << 31));
  const int span = (ksize - 1) * dil;
  if (mode == 0) {
    PFST_CHECK_ARG(Ho == (Hi + 2 * pad - span - 1) / stride + 1 && Wo == (Wi + 2 * pad - span - 1) / stride + 1);
  } else {  // dgrad: `in` is dy (Hi x Wi = forward output), `out` is dx (Ho x Wo = forward input)
    PFST_CHECK_ARG(Hi == (Ho + 2 * pad - span - 1) / stride + 1 && Wi == (Wo + 2 * pad - span - 1) / stride + 1);
  }
  int a, b, c, d;
  if (mode == 0) { a = stride; b = dil; c = -pad; d = 1; } else { a = 1; b = -dil; c = pad; d = stride; }
  hipStream_t s = (hipStream_t)stream;
  const int stats_T = N * pfst_conv_stats_slots(M, Ho, Wo);
  const bool generic = (C % BK_MIN) != 0;
  if (!generic)      // Cin % 16 == 0: K-quad kernel (conv_igemm_q.hip), weights packed [K/4][M][4]
    return pfst_igemm_q_launch(in, in_bs, wk, bias, out, out_bs, N, C, Hi, Wi, M, Ho, Wo, ksize, a, b, c, d, accumulate, stats, stats_T, 1, s, bnb);
  if (M > 64) return launch_igemm_generic<128>(in, in_bs, wk, bias, out, out_bs, N, C, Hi, Wi, M, Ho, Wo, ksize, a, b, c, d, accumulate, stats, stats_T, s);
  if (M > 32) return launch_igemm_generic<64>(in, in_bs, wk, bias, out, out_bs, N, C, Hi, Wi, M, Ho, Wo, ksize, a, b, c, d, accumulate, stats, stats_T, s);
  return launch_igemm_generic<32>(in, in_bs, wk, bias, out, out_bs, N, C, Hi, Wi, M, Ho, Wo, ksize, a, b, c, d, accumulate, stats, stats_T, s);
}

extern "C" int pfst_conv_wgrad(const float* x, long long x_bs, const float* dy, long long dy_bs, float* dw,
                               int N, int Cin, int Hi, int Wi, int Cout, int Ho, int Wo, int ksize, int stride, int dil, int pad,
                               pfst_stream_t stream) {
  PFST_CHECK_ARG(x && dy && dw && N > 0 && Cin > 0 && Cout > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0);
  PFST_CHECK_ARG((ksize == 1 || ksize == 3) && (stride == 1 || stride == 2) && dil >= 1 && pad >= 0);
  const int span = (ksize - 1) * dil;
  PFST_CHECK_ARG(Ho == (Hi + 2 * pad - span - 1) / stride + 1 && Wo == (Wi + 2 * pad - span - 1) / stride + 1);
  PFST_CHECK_ARG(x_bs >= (i64)Cin * Hi * Wi && dy_bs >= (i64)Cout * Ho * Wo && N <= 65535);
  PFST_CHECK_ARG((i64)Cin * Hi * Wi * 4 < (1ll << 31) && (i64)Cout * Ho * Wo * 4 < (1ll << 31));     // 32-bit offsets per image
  hipStream_t s = (hipStream_t)stream;
  if (pfst_wgrad_q_eligible(x, x_bs, dy, dy_bs, Hi, Wi, Ho, Wo, ksize, stride, dil))       // K-quad fast path (conv_wgrad_q.hip)
    return pfst_wgrad_q_launch(x, x_bs, dy, dy_bs, dw, N, Cin, Hi, Wi, Cout, Ho, Wo, ksize, dil, pad, 1, 0, 0, 0, s);
#define PFST_WGRAD(BM_)                                                                                          \
  return ksize == 3 ? launch_wgrad<BM_, 9>(x, x_bs, dy, dy_bs, dw, N, Cin, Hi, Wi, Cout, Ho, Wo, stride, dil, pad, s) \
                    : launch_wgrad<BM_, 1>(x, x_bs, dy, dy_bs, dw, N, Cin, Hi, Wi, Cout, Ho, Wo, stride, dil, pad, s)
  if (Cout > 64) { PFST_WGRAD(128); }
  if (Cout > 32) { PFST_WGRAD(64); }
  PFST_WGRAD(32);
#undef PFST_WGRAD
}

extern "C" int pfst_bias_grad(const float* dy, long long dy_bs, float* db, int N, int C, int HW, pfst_stream_t stream) {
  PFST_CHECK_ARG(dy && db && N > 0 && C > 0 && HW > 0 && C <= 65535 && N <= 65535);
  if (pfst_deterministic()) {          // one workgroup per channel and launch, one image per launch: a single, ordered writer per db[c]
    for (int n = 0; n < N; ++n)
      hipLaunchKernelGGL(bias_grad_kernel, dim3(1, C, 1), dim3(256), 0, (hipStream_t)stream, dy + (i64)n * dy_bs, dy_bs, db, C, HW, HW);
    PFST_CHECK_LAUNCH();
    return PFST_OK;
  }
  int splits = cdiv(HW, 4096);
  if (splits > 256) splits = 256;
  const int chunk = cdiv(HW, splits);
  hipLaunchKernelGGL(bias_grad_kernel, dim3(cdiv(HW, chunk), C, N), dim3(256), 0, (hipStream_t)stream, dy, dy_bs, db, C, HW, chunk);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}
