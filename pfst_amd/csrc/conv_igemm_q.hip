// Implicit-GEMM convolution on the fp32-input MFMA, "K-quad" main loop (Cin % 16 == 0; fprop and dgrad):
//   out[n][m][p] (+)= sum_k W[m][k] * in[n][ci(k)][src(p, tap(k))],  k = tap*C + ci,   GEMM M = Cout, N = pixels, K = taps*C.
// v_mfma_f32_32x32x2_f32 executes at the fp32 VECTOR rate on the SIMD's vector pipe: every VALU instruction in the main
// loop takes ~4 cycles away from the matrix instructions (measured with SQ_INSTS_VALU / SQ_VALU_MFMA_BUSY_CYCLES: MFMA
// utilisation = 64 N_mfma / (64 N_mfma + 4 N_valu) on all our fp32 kernels).  So the loop is built to need almost none:
//   * both operands are staged as 16-byte quads of 4 consecutive k: LDS image [quad][row] of float4, one ds_write_b128 per
//     quad and one ds_read_b128 per MFMA fragment of FOUR k-steps, every LDS address = per-thread base + immediate
//     (the double-buffer flip costs one add per base pointer and K-step);
//   * weights are pre-packed [K/4][M][4] (pfst_conv_pack_weight), read with one buffer_load_dwordx4 per quad;
//   * activations are read with buffer_load_dword, voffsets constant while the tap is unchanged, channel advance as a
//     scalar soffset, padding / ragged tiles as out-of-range offsets (hardware returns 0);
//   * lane (l31, lh) holds quad 2g+lh of its row; MFMA #e of group g consumes element e of both operands
//     (k = 8g + 4 lh + e) -- the K sum is order-free, A and B only have to agree.
// Tiling, XCD-aware tile order and the epilogue (bias / accumulate / fused BN statistics) are those of conv_mfma.hip,
// which keeps the generic path (Cin % 16 != 0: the 3- and 10-band stems).
#include "conv_epilogue.h"
#include "../../include/pfst_hip.h"
#include <stdlib.h>

#ifdef PFST_CLOCK_STAMPS
// DIAGNOSTIC BUILD ONLY (tools/clock_probe.py; the product library never contains this): per-workgroup shader-clock stamps
// {prologue, main loop, epilogue} + real-time {lifetime, start}, written to a buffer that nothing else reads.
__device__ unsigned long long g_pfst_stamps[5 * 65536];
extern "C" int pfst_debug_read_stamps(unsigned long long* host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pfst_stamps), sizeof(unsigned long long) * 5 * n) == hipSuccess ? 0 : -2;
}
#endif

namespace {

constexpr int QBN = 128, QBK = 16, QNQ = 4;

__device__ __forceinline__ bool src_coord_q(int o, int t, int a, int b, int c0, int div, int lim, int& s) {
  const int v = o * a + t * b + c0;
  const int odd = v & (div - 1);
  s = v >> (div >> 1);
  return (odd == 0) & (s >= 0) & (s < lim);
}

// BNB: the data-gradient launch that completes dL/dy of a BatchNorm layer also emits that layer's backward sums (conv_epilogue.h)
template <int BM, int BNB = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void conv_igemm_q_kernel(
    const float* __restrict__ in_all, i64 in_bs, const float4* __restrict__ wq, const float* __restrict__ bias,
    float* __restrict__ out_all, i64 out_bs, int N, int C, int Hi, int Wi, int M, int Ho, int Wo, int ks,
    int ca, int cb, int cc, int cdivv, int accumulate, float* __restrict__ stats, int stats_T, PfstBnbArgs bnb) {
  constexpr int BN = QBN, BK = QBK, NQ = QNQ;
  constexpr int WM = BM >= 64 ? 64 : 32;
  constexpr int WAVES_M = BM / WM;
  constexpr int WAVES_N = 4 / WAVES_M;
  constexpr int WN = BN / WAVES_N;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int A_CH = NQ * BM;                       // weight quads per K-step
  constexpr int A_N = (A_CH + 255) / 256;
  constexpr unsigned OOB = 0x80000000u;

  // one LDS block: the A / B double buffers of the main loop, reused by the epilogue's row reductions (4 waves x 5 KB)
  constexpr int LDS_F4 = 2 * NQ * BM + 2 * NQ * BN;
  static_assert(LDS_F4 * 4 >= 4 * PFST_ROWSUM_LDS_FLOATS, "epilogue scratch must fit into the main loop's LDS");
  __shared__ float4 smem[LDS_F4];
  float4 (*As)[NQ * BM] = reinterpret_cast<float4 (*)[NQ * BM]>(smem);
  float4 (*Bs)[NQ * BN] = reinterpret_cast<float4 (*)[NQ * BN]>(smem + 2 * NQ * BM);

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave id in an SGPR
#ifdef PFST_CLOCK_STAMPS
  const unsigned long long t_clk0 = __builtin_amdgcn_s_memtime(), t_rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  const int wm0 = (wid / WAVES_N) * WM, wn0 = (wid % WAVES_N) * WN;
  const int P = Ho * Wo, HiWi = Hi * Wi;
  const int gx = (P + BN - 1) / BN, gy = (M + BM - 1) / BM;
  const int tiles_img = gx * gy, tiles = tiles_img * N;
  const int K = C * ks * ks;
  const int KT = K / BK;

  // B staging role: one pixel, one k-half (8 channels = 2 quads);  A staging role: quads tid + 256 i of the [quad][row] image
  const int pix = tid & (BN - 1), kh = tid >> 7;
  // gridDim.y > 1: a batch of GEMMs that share shapes but not weights (the 16 transform-domain products of the Winograd
  // path, conv_winograd.hip): group g = blockIdx.y uses weight set g and images g*N .. g*N+N-1.
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(wq) + (i64)blockIdx.y * (K / 4) * M, 0,
                                                                         K * M * 4, 0x00020000);

  // ---- tile of this workgroup (grid: tiles x 1 x images).  XCD-aware order inside an image: workgroups are dealt
  // round-robin over the 8 XCDs, so the m-tiles that share one pixel tile (= one activation tile) get ids 8 apart: they run
  // back to back on the SAME XCD and reuse its L2 copy (measured on the first kernel: -26 % FETCH_SIZE).
  // (Persistent workgroups looping over tiles with cross-tile prefetch were measured 3-4 % SLOWER than letting the
  // dispatcher hand out one tile per workgroup.)
  int t_n = 0, t_bx = 0, t_m0 = 0, t_p0 = 0;          // tile being loaded
  const float* in = in_all;
  bool pvalid = false;
  int oy = 0, ox = 0;
  unsigned a_voff[A_N], b_voff = OOB;                // one activation voffset: the 8 channels of the k-half ride on the soffset
  int ld_ty = 0, ld_tx = 0, ld_ci0 = 0;              // (tap, first channel) of the K-slice being prefetched
  auto set_tap = [&]() {
    int sy, sx;
    const bool ok = pvalid & src_coord_q(oy, ld_ty, ca, cb, cc, cdivv, Hi, sy) & src_coord_q(ox, ld_tx, ca, cb, cc, cdivv, Wi, sx);
    b_voff = ok ? 4u * ((unsigned)(kh * 8) * (unsigned)HiWi + (unsigned)(sy * Wi + sx)) : OOB;
  };
  auto setup_tile = [&](int tile) {
    t_n = blockIdx.y * N + blockIdx.z;              // image index straight from SGPRs: the buffer descriptors stay scalar
    const int lin = tile;
    int by;
    pfst_tile_order(lin, gx, gy, ks == 3, t_bx, by);
    t_p0 = t_bx * BN;
    t_m0 = by * BM;
    in = in_all + (i64)t_n * in_bs;
    const int p = t_p0 + pix;
    pvalid = p < P;
    oy = pvalid ? p / Wo : 0;
    ox = pvalid ? p - oy * Wo : 0;
#pragma unroll
    for (int i = 0; i < A_N; ++i) {
      const int c = tid + 256 * i;
      const int kq = c / BM, row = c - kq * BM;
      a_voff[i] = (c < A_CH && t_m0 + row < M) ? 16u * ((unsigned)kq * (unsigned)M + (unsigned)(t_m0 + row)) : OOB;
    }
    ld_ty = 0; ld_tx = 0; ld_ci0 = 0;
    set_tap();
  };

  float4 areg[A_N];
  float breg[8];
  pfst_f32x16 acc[TM][TN];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };

  auto load_tile = [&](int kt) {
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, C * HiWi * 4, 0x00020000);
    const int a_soff = kt * (BK / 4) * M * 16, b_soff = ld_ci0 * HiWi * 4;
#pragma unroll
    for (int i = 0; i < A_N; ++i)
      areg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, a_voff[i], a_soff, 0));
#pragma unroll
    for (int i = 0; i < 8; ++i)
      breg[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b_rsrc, b_voff, b_soff + i * HiWi * 4, 0));
    ld_ci0 += BK;
    if (ld_ci0 >= C) {
      ld_ci0 = 0; ld_tx += 1;
      if (ld_tx == ks) { ld_tx = 0; ld_ty += 1; }
      set_tap();
    }
  };
  const int l31 = lane & 31, lh = lane >> 5;
  // LDS pointers of this thread; the double-buffer flip is one add per pointer and K-step, everything else is immediates
  const float4* a_rd = &As[0][lh * BM + wm0 + l31];     // + 2g * BM + 32 i
  const float4* b_rd = &Bs[0][lh * BN + wn0 + l31];     // + 2g * BN + 32 j
  float4* a_wr = &As[0][tid];                           // + 256 i
  float4* b_wr = &Bs[0][(2 * kh) * BN + pix];           // + BN for the second quad
  int da = NQ * BM, db = NQ * BN;

  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < A_N; ++i)
      if (A_CH % 256 == 0 || tid + 256 * i < A_CH) a_wr[256 * i] = areg[i];
    b_wr[0] = make_float4(breg[0], breg[1], breg[2], breg[3]);
    b_wr[BN] = make_float4(breg[4], breg[5], breg[6], breg[7]);
  };

  setup_tile(blockIdx.x);
  load_tile(0);
  store_tile();
  __syncthreads();
  a_wr += da; b_wr += db;
  zero_acc();
#ifdef PFST_CLOCK_STAMPS
  const unsigned long long t_clk1 = __builtin_amdgcn_s_memtime();
#endif
  auto mma_step = [&]() {
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      float4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = a_rd[2 * g * BM + 32 * i];
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = b_rd[2 * g * BN + 32 * j];
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
  for (int kt = 0; kt + 1 < KT; ++kt) {             // steady state: prefetch K-slice kt+1, multiply slice kt
    load_tile(kt + 1);
    mma_step();
    store_tile();
    __syncthreads();
    a_rd += da; b_rd += db; a_wr -= da; b_wr -= db;
    da = -da; db = -db;
  }
  mma_step();                                       // last K-slice
#ifdef PFST_CLOCK_STAMPS
  const unsigned long long t_clk2 = __builtin_amdgcn_s_memtime();
#endif
  float* red = nullptr;
  if (BNB != 0 || stats) {                          // wave-uniform: the epilogue reduces its per-row sums through LDS
    __syncthreads();                                // every wave is done reading the last K-slice
    red = reinterpret_cast<float*>(smem);
  }
  conv_epilogue<TM, TN, WAVES_N, BN, BNB, true>(acc, out_all + (i64)t_n * out_bs, bias, stats, stats_T, accumulate, M, P, t_m0, t_p0, wm0, wn0,
                                          t_bx, t_n, wid, lane, bnb, red);
#ifdef PFST_CLOCK_STAMPS
  if (tid == 0) {
    __builtin_amdgcn_s_waitcnt(0);                  // the stores have left the wave
    const unsigned long long t_clk3 = __builtin_amdgcn_s_memtime();
    const unsigned b = (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) & 65535u;
    g_pfst_stamps[5 * b + 0] = t_clk1 - t_clk0;
    g_pfst_stamps[5 * b + 1] = t_clk2 - t_clk1;
    g_pfst_stamps[5 * b + 2] = t_clk3 - t_clk2;
    g_pfst_stamps[5 * b + 3] = __builtin_amdgcn_s_memrealtime() - t_rt0;
    g_pfst_stamps[5 * b + 4] = t_rt0;
  }
#endif
}

template <int BM>
int launch_q(const float* in, i64 in_bs, const float* wq, const float* bias, float* out, i64 out_bs, int N, int C, int Hi,
             int Wi, int M, int Ho, int Wo, int ks, int a, int b, int c, int d, int acc, float* stats, int stats_T, int groups,
             const PfstBnbArgs* bnb, hipStream_t s) {
  dim3 grid(cdiv((i64)Ho * Wo, QBN) * cdiv(M, BM), groups, N);
  constexpr int lds_pad = 0;
  if (bnb && bnb->x) {
    // the fused sums use the epilogue's full-tile store path: whole row tiles, no bias, one GEMM group
    PFST_CHECK_ARG(M % BM == 0 && !bias && !stats && groups == 1 && bnb->coef && bnb->partials);
#define PFST_LAUNCH_BNB(MODE_)                                                                                                        \
    hipLaunchKernelGGL((conv_igemm_q_kernel<BM, MODE_>), grid, dim3(256), lds_pad, s, in, in_bs, reinterpret_cast<const float4*>(wq), bias, out, \
                       out_bs, N, C, Hi, Wi, M, Ho, Wo, ks, a, b, c, d, acc, stats, stats_T, *bnb)
    if (!bnb->relu) PFST_LAUNCH_BNB(3);
    else if (bnb->y) PFST_LAUNCH_BNB(2);
    else PFST_LAUNCH_BNB(1);
#undef PFST_LAUNCH_BNB
  } else {
    hipLaunchKernelGGL((conv_igemm_q_kernel<BM, 0>), grid, dim3(256), lds_pad, s, in, in_bs, reinterpret_cast<const float4*>(wq), bias, out,
                       out_bs, N, C, Hi, Wi, M, Ho, Wo, ks, a, b, c, d, acc, stats, stats_T, PfstBnbArgs());
  }
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

}  // namespace

int pfst_igemm_q_launch(const float* in, i64 in_bs, const float* wq, const float* bias, float* out, i64 out_bs, int N, int C, int Hi,
                        int Wi, int M, int Ho, int Wo, int ks, int a, int b, int c, int d, int acc, float* stats, int stats_T, int groups,
                        hipStream_t s, const PfstBnbArgs* bnb) {
  if (M > 64) return launch_q<128>(in, in_bs, wq, bias, out, out_bs, N, C, Hi, Wi, M, Ho, Wo, ks, a, b, c, d, acc, stats, stats_T, groups, bnb, s);
  if (M > 32) return launch_q<64>(in, in_bs, wq, bias, out, out_bs, N, C, Hi, Wi, M, Ho, Wo, ks, a, b, c, d, acc, stats, stats_T, groups, bnb, s);
  return launch_q<32>(in, in_bs, wq, bias, out, out_bs, N, C, Hi, Wi, M, Ho, Wo, ks, a, b, c, d, acc, stats, stats_T, groups, bnb, s);
}
