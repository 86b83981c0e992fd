// fp32-faithful convolution on the fp16 matrix cores with THREE MFMAs per product term instead of six (conv_split.hip: bf16x6).
//
// A bf16 piece carries 8 significand bits, so an fp32 value needs three pieces and a product six piece-products.  An fp16 piece
// carries 11 bits: two pieces h + l hold 22 bits (|x - h - l| <= 2^-22 |x|, remainders computed exactly) and the product is
//     a*b ~= ah*bh + ah*bl + al*bh            (dropped: al*bl <= 2^-22 |ab|, zero-mean; round-off of the pieces 2^-22)
// accumulated in fp32 by v_mfma_f32_16x16x32_f16, which runs at the bf16 rate -- half the matrix work of bf16x6.  What fp16 lacks is
// exponent range (5 bits), so every operand tensor is scaled by a power of two taken from its absolute maximum:
//     s = 2^(14 - floor(log2 max|x|))  ->  max|x s| < 2^15 (no overflow), both pieces of every element down to 2^-18 of the tensor's
//     maximum are fp16 NORMALS (full 22 bits), smaller elements keep an absolute error <= 2^-25 / s = 2^-40 of the maximum;
// the accumulator is multiplied by 1 / (s_a s_b), an exact power of two, in the epilogue.  The absolute maxima are device scalars
// (one fp32 slot per tensor, or per transform-domain plane of a Winograd layer) written by the kernels that PRODUCE the tensors
// (atomicMax on the bit pattern) or by pfst_absmax; nothing is read back to the host.  Emulation against fp64:
// 3.6e-7 at K = 2048 against 5.0e-7 for bf16x6 and 5.6e-7 for an fp32 fma chain -- fewer roundings per accumulated product.
//
// This file: the K=32 "pair" loop of conv_split.hip re-scheduled for 48 MFMAs per step (fprop / dgrad / the grouped Winograd-domain
// GEMM), the weight packing and the absmax reduction.  Layers it does not cover (Cin % 32 != 0, fewer than 65 output channels) stay
// on the bf16x6 or fp32-input MFMA kernels.
#include "conv_epilogue.h"
#include "amax.h"
#include "det.h"
#include "weight_jobs.h"
#include <math.h>
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include "../../include/pfst_hip.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BN = 128;
constexpr int NP = 2;      // pieces: 0 = h (high 11 bits), 1 = l (next 11 bits)

template <int... Is, class F>
__device__ __forceinline__ void static_for_seq(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>()), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_seq(std::make_integer_sequence<int, N>(), static_cast<F&&>(f));
}

__device__ __forceinline__ bool src_coord(int o, int t, int a, int b, int c0, int div, int lim, int& s) {
  const int v = o * a + t * b + c0;
  const int odd = v & (div - 1);
  s = v >> (div >> 1);
  return (odd == 0) & (s >= 0) & (s < lim);
}

// ---- absolute maximum of a tensor into its slot group (amax.h)
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, i64 n, i64 plane_stride, int slot_stride,
                                                     float* __restrict__ slots) {
  // blockIdx.y: plane; a plane is n contiguous floats, planes `plane_stride` apart; plane y goes to group y * slot_stride (0: all planes
  // into one group, e.g. the per-image blocks of a channel slice of a concat buffer)
  const float* p = x + (i64)blockIdx.y * plane_stride;
  float m = 0.f;
  const i64 n4 = (((uintptr_t)p & 15) == 0) ? n / 4 : 0;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (i64)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(p)[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  for (i64 i = n4 * 4 + (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(p[i]));
  amax_publish(slots + (i64)blockIdx.y * slot_stride * PFST_AMAX_SUB, m);
}

// two-piece fp16 split of 8 values already multiplied by the tensor's scale (plain code: prologue and packing)
__device__ __forceinline__ void split8_f16(const float (&v)[8], float s, uint4& ph, uint4& pl) {
  unsigned h[4], l[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float t0 = v[2 * q] * s, t1 = v[2 * q + 1] * s;
    const _Float16 h0 = (_Float16)t0, h1 = (_Float16)t1;                 // round to nearest even
    const _Float16 l0 = (_Float16)(t0 - (float)h0), l1 = (_Float16)(t1 - (float)h1);   // the remainders are exact in fp32
    h[q] = (unsigned)__builtin_bit_cast(unsigned short, h0) | (unsigned)__builtin_bit_cast(unsigned short, h1) << 16;
    l[q] = (unsigned)__builtin_bit_cast(unsigned short, l0) | (unsigned)__builtin_bit_cast(unsigned short, l1) << 16;
  }
  ph = make_uint4(h[0], h[1], h[2], h[3]);
  pl = make_uint4(l[0], l[1], l[2], l[3]);
}

// a PRE-SPLIT operand (the Winograd transforms write V / dM that way: one dword per element, h in the low half, l in the high half):
// the two piece vectors of 8 values are byte permutes of the 8 loaded dwords -- 8 instructions instead of 24
__device__ __forceinline__ void unpack8_f16(const float (&v)[8], uint4& ph, uint4& pl) {
  unsigned h[4], l[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned d0 = __builtin_bit_cast(unsigned, v[2 * q]), d1 = __builtin_bit_cast(unsigned, v[2 * q + 1]);
    h[q] = __builtin_amdgcn_perm(d1, d0, 0x05040100u);
    l[q] = __builtin_amdgcn_perm(d1, d0, 0x07060302u);
  }
  ph = make_uint4(h[0], h[1], h[2], h[3]);
  pl = make_uint4(l[0], l[1], l[2], l[3]);
}
struct SplitF16;
template <int K> __device__ __forceinline__ void unpack_op_f16(const float (&v)[8], SplitF16& st);

// The same split as single VALU instructions pinned in place (volatile asm, like split_op of conv_split.hip), K = 0..23 on the eight
// values, six slots per pair q of which FOUR carry an instruction (round 4; PFST_F16X3_SPLIT6: the six-instruction form of round 3 --
// two scale multiplies, v_cvt_pk_f16_f32, two v_fma_mix_f32, v_cvt_pk_f16_f32).  v_fma_mixlo_f16 / v_fma_mixhi_f16 compute
// fma(x, s, c) in fp32 and round ONCE to fp16 into the low / high half of the destination:
//     h.lo = f16(x0 s)   h.hi = f16(x1 s)            x s is exact (a power of two), so this is the round-to-nearest-even of x s
//     l.lo = f16(x0 s - h.lo)   l.hi = f16(x1 s - h.hi)    the remainder is exact in fp32 (|x s - h| <= ulp_f16 / 2), rounded once
// -- the same two pieces bit for bit (tests/test_f16x3_elementwise_gpu.py::test_split_instructions_against_the_definition), 16
// instead of 24 vector instructions per 8 values on loops that are bound by what they issue beside the MFMAs.  The scale sits in an SGPR.
struct SplitF16 {
  unsigned h[4], l[4];
  float t0[4], t1[4];
};
#ifdef PFST_F16X3_SPLIT6
template <int K>
__device__ __forceinline__ void split_op_f16(const float (&v)[8], float s, SplitF16& st) {
  constexpr int q = K / 6, op = K % 6;
  if constexpr (op == 0) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(st.t0[q]) : "s"(s), "v"(v[2 * q]));
  else if constexpr (op == 1) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(st.t1[q]) : "s"(s), "v"(v[2 * q + 1]));
  else if constexpr (op == 2) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(st.h[q]) : "v"(st.t0[q]), "v"(st.t1[q]));
  else if constexpr (op == 3)
    asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(st.t0[q]) : "v"(v[2 * q]), "s"(s), "v"(st.h[q]));
  else if constexpr (op == 4)
    asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(st.t1[q]) : "v"(v[2 * q + 1]), "s"(s), "v"(st.h[q]));
  else asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(st.l[q]) : "v"(st.t0[q]), "v"(st.t1[q]));
}
#else
template <int K>
__device__ __forceinline__ void split_op_f16(const float (&v)[8], float s, SplitF16& st) {
  constexpr int q = K / 6, op = K % 6;               // slots 0, 1 of a pair are empty: the issue schedules of the loops keep their shape
  if constexpr (op == 2) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(st.h[q]) : "v"(v[2 * q]), "s"(s));
  else if constexpr (op == 3) asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(st.h[q]) : "v"(v[2 * q + 1]), "s"(s));
  else if constexpr (op == 4)
    asm volatile("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(st.l[q]) : "v"(v[2 * q]), "s"(s), "v"(st.h[q]));
  else if constexpr (op == 5)
    asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(st.l[q]) : "v"(v[2 * q + 1]), "s"(s), "v"(st.h[q]));
}
#endif

// K = 0..7: permute number K of unpack8_f16 as one pinned instruction (h pairs first)
template <int K>
__device__ __forceinline__ void unpack_op_f16(const float (&v)[8], SplitF16& st) {
  constexpr int q = K & 3;
  if constexpr (K < 4) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(st.h[q]) : "v"(v[2 * q + 1]), "v"(v[2 * q]), "s"(0x05040100u));
  else asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(st.l[q]) : "v"(v[2 * q + 1]), "v"(v[2 * q]), "s"(0x07060302u));
}

// w[Cout][Cin][T] -> two-piece K-major images, scaled by the set's power of two.
// layout: [k/16][piece 2][k-half 2][row][8 x f16]  (one uint4 per (k16-group, piece, half, row)); fprop: k = t*Cin+ci, row = co;
// dgrad: k = t*Cout+co, row = ci.  blockIdx.y: filter set (Winograd transform index), `set_in` floats / `set_out` chunks apart.
__global__ void pack_weight_f16x2_kernel(const float* __restrict__ w, uint4* __restrict__ wf, uint4* __restrict__ wd, int Cout, int Cin, int T,
                                         i64 set_in, i64 set_out, const float* __restrict__ amax) {
  w += blockIdx.y * set_in;
  if (wf) wf += blockIdx.y * set_out;
  if (wd) wd += blockIdx.y * set_out;
  const float s = scale_of(amax_exponent(amax_read(amax + (i64)blockIdx.y * PFST_AMAX_SUB)));
  const i64 nf = (i64)(T * Cin / 16) * 2 * Cout, nd = (i64)(T * Cout / 16) * 2 * Cin;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < nf + nd; i += (i64)gridDim.x * blockDim.x) {
    const bool dg = i >= nf;
    const i64 e = dg ? i - nf : i;
    const int M = dg ? Cin : Cout, Cq = dg ? Cout : Cin;
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
    uint4 ph, pl;
    split8_f16(v, s, ph, pl);
    uint4* dst = dg ? wd : wf;
    const i64 base = (i64)g * 2 * NP * M;
    dst[base + (i64)(0 * 2 + h) * M + row] = ph;
    dst[base + (i64)(1 * 2 + h) * M + row] = pl;
  }
}

// the same images for every layer of a network in one launch (pfst_conv_pack_weight_f16x2_batched): whole workgroups per filter set
__global__ __launch_bounds__(256) void pack_weight_f16x2_batched_kernel(const pfst_weight_job_t* __restrict__ jobs, int njobs) {
  const pfst_weight_job_t J = jobs[weight_job_of_block(jobs, njobs, blockIdx.x)];
  const int lb = blockIdx.x - J.first_block;
  const int Cout = J.Cout, Cin = J.Cin, T = J.T;
  const i64 nf = J.dst_f ? (i64)(T * Cin / 16) * 2 * Cout : 0, nd = J.dst_d ? (i64)(T * Cout / 16) * 2 * Cin : 0;
  const int bps = (int)((nf + nd + 255) / 256);          // workgroups per set
  const int set = lb / bps;
  const i64 n = (i64)Cout * Cin * T;
  const float s = scale_of(amax_exponent(amax_read(J.amax_f + (i64)set * PFST_AMAX_SUB)));     // every lane takes part
  const i64 i = (i64)(lb - set * bps) * 256 + threadIdx.x;
  if (i >= nf + nd) return;
  const float* w = J.src + set * n;
  const bool dg = i >= nf;
  const i64 e = dg ? i - nf : i;
  const int M = dg ? Cin : Cout, Cq = dg ? Cout : Cin;
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
  uint4 ph, pl;
  split8_f16(v, s, ph, pl);
  uint4* dst = reinterpret_cast<uint4*>(dg ? J.dst_d : J.dst_f) + set * (4 * n / 16);
  const i64 base = (i64)g * 2 * NP * M;
  dst[base + (i64)(0 * 2 + h) * M + row] = ph;
  dst[base + (i64)(1 * 2 + h) * M + row] = pl;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The implicit GEMM.  Workgroup tile 128 x 128, four waves of 64 x 64 = 4 x 4 accumulators of v_mfma_f32_16x16x32_f16; one step
// consumes K = 32 (two K=16 LDS tiles of [piece][half][row] images, k-quarter q of the MFMA operand = tile q / 2, half q % 2): 16
// fragments in registers, 48 MFMAs (al bh | ah bl | ah bh, smallest terms first).  One LDS buffer: barrier A sits in the MFMA stream
// once every wave holds its fragments, the next pair is split and stored behind it, barrier B ends the step.  The activations of pair
// k+2 are loaded (16 scalar buffer loads per thread, alternating register sets) while pair k+1 is split and pair k multiplied, so a
// load has a whole step to land; the four pre-split weight chunks run two pairs ahead as well (loaded behind the stores of the set that
// is being written to LDS).
// Issue order is fixed slot by slot (one MFMA + at most two fillers, a scheduling barrier after each slot).
// ---------------------------------------------------------------------------------------------------------------------------------
// SHAPE: the MFMA instruction, 32 = v_mfma_f32_32x32x16_f16 (24 per step; round 3's 16 = v_mfma_f32_16x16x32_f16 form, 48 per step, is gone: half the MFMA
// issue slots, accumulators already in the 32x32 layout of conv_epilogue -- no re-layout through LDS)
// BNB != 0 (SHAPE 32 only): the data-gradient launch also emits the BatchNorm-backward sums of the layer that owns `out` (conv_epilogue.h)
// BPACK: the activation operand is stored pre-split (Winograd-domain V written by pfst_wino_input in packed mode)
// ONE: a 1x1 convolution with stride 1 and no padding (and the Winograd-domain GEMMs): output pixel = input pixel, no tap arithmetic --
// the form the tile chain exists for (fewer live scalars: the chain's two-sided tile state must not push the loop's SGPRs out)
// BMT: rows of the workgroup tile.  256 (512 threads, eight waves of 64 x 64, one workgroup per CU; 1x1 / Winograd-domain launches with
// M % 256 == 0): every thread stages ONE K=16 tile of a pair instead of both, i.e. half the split instructions, activation loads and
// activation LDS stores per MFMA, and an activation tile is fetched once for 256 output rows.  (Diagnostic: the 128-row loop with the
// second tile's loads and split removed -- wrong results, timing only -- runs 10-11 % faster at every K.)
struct PfstResGate {                       // value-initialised = off
  const float* g = nullptr;                // [N][M][P] tensor added to the output where its bit is set
  i64 g_bs = 0;
  const unsigned long long* mask = nullptr;    // bn_apply's ReLU bitmask of an [N][M][P] tensor (dense)
};

// BNL (256-row pixel-to-pixel tile, plain operand): `in` is the PRE-normalisation output of the conv -> BN -> ReLU layer feeding this 1x1
// convolution (Bottleneck conv2 -> bn2 -> relu -> conv3); every activation element is normalised -- max(fma(x, sc, sh), 0) with the (sc, sh)
// of its channel from bnl[C] = (mean, invstd, sc, sh) -- between its load and its split, the normalised tensor is never written; in_amax
// then holds the PREDICTED max of the normalised tensor (pfst_bn_finalize_partials).  The coefficients sit in LDS (C <= 2048: 16 KB).
template <int SHAPE, int BNB = 0, bool BPACK = false, bool ONE = false, int BMT = 128, bool BNL = false>
__device__ __forceinline__ void conv_igemm_f16x3_body(
    const float* __restrict__ in, i64 in_bs, const uint4* __restrict__ wk4, const float* __restrict__ bias,
    float* __restrict__ out, i64 out_bs, int C, int Hi, int Wi, int M, int Ho, int Wo, int ks,
    int ca, int cb, int cc, int cdivv, int accumulate, float* __restrict__ stats, int stats_T,
    const float* __restrict__ w_amax, const float* __restrict__ in_amax, int Y, int Z, int chain, const PfstBnbArgs& bnb,
    const PfstResGate& gate, const float4* __restrict__ bnl = nullptr) {
  static_assert(!BNL || (BMT == 256 && ONE && !BPACK && BNB == 0), "normalise-on-load exists for the 256-row 1x1 forward tile");
  static_assert(SHAPE == 32, "one MFMA shape: v_mfma_f32_32x32x16_f16 (the 16x16x32 form of round 3 was removed in round 5)");
  static_assert(BMT == 128 || (BMT == 256 && SHAPE == 32 && ONE) || (BMT == 64 && SHAPE == 32 && BNB == 0 && !BPACK),
                "the 256-row tile exists for the pixel-to-pixel 32x32 loop, the 64-row tile for the plain 32x32 loop");
  // BMT 64 (layers with 33 ... 64 output rows: layer1 conv2, stem.6): 256 threads as for 128 rows, the four waves 2 x 2 over 64 rows x 128
  // pixels, i.e. ONE 32-row block per wave (TMW = 1) -- the activation staging is the 128-row tile's (same split work per pixel, spread over
  // half the MFMAs), a weight tile is one 16-byte chunk per thread
  constexpr int BM = BMT, WAVES_N = 2, NT = BM == 64 ? 256 : 2 * BM;     // NT threads: BM / 64 x 2 waves (64 rows: 2 x 2 waves of 32 rows)
  constexpr int TMW = BM == 64 ? 1 : 2;                         // 32-row accumulator blocks of a wave tile
  constexpr int TPT = BM == 256 ? 1 : 2;                        // K=16 activation tiles of a pair one thread stages
  constexpr int NA = 2 * (2 * NP * BM) / NT;                    // 16-byte weight chunks of a pair one thread stages: 4 (2 at 64 rows)
  constexpr int TILE_A = 2 * NP * BM, TILE_B = 2 * NP * BN;     // 16-byte chunks of one K=16 tile
  // SHAPE 32: two LDS buffers of a pair (2 x 32 KB; two workgroups per CU either way: 176 registers), so a step needs ONE barrier and the
  // stores of pair k+1 go to the other buffer whenever their data is ready.  SHAPE 16: one buffer, barrier A in the MFMA stream.
  constexpr int PAIR_CHUNKS = 2 * TILE_A + 2 * TILE_B;
  constexpr int SMEM_CHUNKS = 2 * PAIR_CHUNKS;
  __shared__ uint4 smem[SMEM_CHUNKS];
  uint4* const As = smem;
  uint4* const Bs = smem + 2 * TILE_A;

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wid / WAVES_N) * (32 * TMW), wn0 = (wid % WAVES_N) * 64;
  const int P = Ho * Wo, HiWi = Hi * Wi;
  constexpr int BNL_ROWS = 2048 + 32;                    // C <= 2048 (host) + the zero rows a last half block reads
  __shared__ uint4 bnl_s[BNL ? BNL_ROWS / 2 : 1];        // [(sc, sh) of two channels]
  if constexpr (BNL) {
    for (int i = tid; i < BNL_ROWS; i += NT) {
      const float4 cf = i < C ? bnl[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      reinterpret_cast<float2*>(bnl_s)[i] = make_float2(cf.z, cf.w);
    }
    __syncthreads();
  }
  const int gx = (P + BN - 1) / BN, gy = (M + BM - 1) / BM, X = gx * gy;
  const int total = X * Y * Z, G = gridDim.x;
  // tile number -> (pixel tile, row tile, filter set y, image z): the tiles of one GEMM fastest, then the IMAGES of one filter set, then
  // the sets -- a set's weights (1 MB at 512 x 512) are then reused by the 8 images that follow each other on an XCD; with the sets
  // inside the images (the order of an (X, Y, Z) grid) every image re-read all 36 sets: 2.9 GB of L2 misses per launch against 0.6 GB of
  // activations, 5.3 TB/s of fabric traffic under a kernel meant to be matrix-bound
  auto decode = [&](int lin, int& bx, int& by, int& y, int& n) {
    const int zy = lin / X, x = lin - zy * X;
    y = zy / Z;
    const int z = zy - y * Z;
    n = y * Z + z;
    pfst_tile_order(x, gx, gy, !ONE && ks == 3, bx, by);   // XCD-aware (common.h)
  };
  const int spt = (C + 31) / 32;                         // K=32 steps per filter tap (1x1: the last one may be half empty)
  const int ntaps = ONE ? 1 : ks * ks;
  const int KP = spt * ntaps;                            // steps in all
  const int KT16 = (C / 16) * ntaps;
  // CHAIN: the tiles of a workgroup form ONE software pipeline -- the loads that run two pairs ahead move on to the workgroup's next tile
  // when a tile's pairs are used up, so the next tile's first pair is split and stored, and its second loaded, inside the last two steps
  // of the current one: after the first tile there is no prologue and no memory latency in front of a tile's first MFMA.  (K <= 512
  // launches spent 4-6 K-steps' worth of time per tile outside the loop: tools/gemm_k_sweep.py.)  Needs an even number of steps (the
  // LDS buffer / register set of a pair is its parity) and the 32x32 loop; otherwise every tile runs its own prologue.
  const bool CHAIN = ONE && SHAPE == 32 && (KP & 1) == 0 && (chain & 1) != 0;       // `chain`: bit 0 the tile chain, bit 1 min / max partials
  float* const stats_mm = (stats && (chain & 2)) ? stats + (i64)2 * M * stats_T : nullptr;      // behind the (sum, sum of squares) partials
  // Short contractions write their output almost as fast as HBM takes it (K = 256: 128 KB per tile every ~10 us per CU) and the dirty lines
  // they park in L2 get in the way of the operand reads: with streaming (nt) stores the same launch runs 15 % (K = 256) / 6 % (K = 512) faster,
  // a K = 2048 launch 1.6 % slower (profiles/r05_gemm_store_cost.txt; a build whose stores all land in one 64 KB window is as fast as a
  // build without stores: the cost is the write traffic, not the issue).  chain bit 2: the host enables it
#ifdef PFST_NT_ALL_K
  const bool nt_out = (chain & 4) != 0;
#else
  const bool nt_out = KP <= 16 && (chain & 4) != 0;
#endif

  const int pix = tid & (BN - 1), kh = (tid >> 7) & 1;   // activation staging: pixel, k-half ...
  const int bt = BM == 256 ? tid >> 8 : 0;               // ... and (256-row tile) which K=16 tile of the pair
  const int a_seg = tid / BM, a_row = tid - a_seg * BM;  // chunk c = tid + NT i of a tile -> segment seg + 2 i, same row
  constexpr unsigned OOB = 0x80000000u;
  const int a_chunk = 2 * M * 16, a_tile = 2 * NP * M * 16, b_chan = HiWi * 4;
  const int chan_step = 32 * HiWi * 4;

  // ---- the load side: tile ld_lin, its buffer resources and this thread's pixel / weight row
  int ld_lin = blockIdx.x;
  __amdgpu_buffer_rsrc_t a_rsrc, b_rsrc;
  unsigned a_voff;
  int ld_oy, ld_ox;
  bool ld_pvalid;
  auto set_load_side = [&](int lin) {
    int bx, by, y, n;
    decode(lin, bx, by, y, n);
    const int p = bx * BN + pix, m0 = by * BM;
    ld_pvalid = p < P;
    if constexpr (ONE) {
      ld_oy = p;                                         // the pixel itself
      ld_ox = 0;
    } else {
      ld_oy = ld_pvalid ? p / Wo : 0;
      ld_ox = ld_pvalid ? p - ld_oy * Wo : 0;
    }
    a_voff = (m0 + a_row < M) ? 16u * ((unsigned)a_seg * (unsigned)M + (unsigned)(m0 + a_row)) : OOB;
    a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(wk4) + (i64)y * KT16 * 2 * NP * M, 0, KT16 * 2 * NP * M * 16, 0x00020000);
    b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in) + (i64)n * in_bs, 0, C * HiWi * 4, 0x00020000);
  };
  // activation voffset of the pixel for filter tap `tap` (0 .. ks*ks-1): constant while the tap does not change
  auto tap_voff = [&](int tap) -> unsigned {
    if constexpr (ONE) return ld_pvalid ? 4u * ((unsigned)(bt * 16 + kh * 8) * (unsigned)HiWi + (unsigned)ld_oy) : OOB;
    const int ty = tap / ks, tx = tap - ty * ks;
    int sy, sx;
    const bool ok = ld_pvalid & src_coord(ld_oy, ty, ca, cb, cc, cdivv, Hi, sy) & src_coord(ld_ox, tx, ca, cb, cc, cdivv, Wi, sx);
    return ok ? 4u * ((unsigned)(kh * 8) * (unsigned)HiWi + (unsigned)(sy * Wi + sx)) : OOB;
  };

  uint4 areg[2][NA];                                     // [register set][tile 2][chunk NA / 2]: like the activations, two pairs ahead
  float breg[2][TPT][8];                                 // [register set][tile][channel kh * 8 + i of the tile's 16]
  pfst_f32x16 acc32[TMW][2];                             // the epilogue's 32 x 32 block layout
#pragma unroll
  for (int i = 0; i < TMW; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc32[i][j][r] = 0.f;

  // loads of pair q: activations into register set SET (v = 0..15), weights into areg (v = 16..19)
  auto load_b = [&](auto vc, auto setc, unsigned voff, int soff) {
    constexpr int v = decltype(vc)::value, SET = decltype(setc)::value;                // v = 0 .. 8 TPT - 1 (256-row tile: the thread's one tile sits in voff)
    breg[SET][v >> 3][v & 7] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b_rsrc, voff, soff + ((v >> 3) * 16 + (v & 7)) * b_chan, 0));
  };
  auto load_a = [&](auto vc, auto setc, int a_soff) {
    constexpr int v = decltype(vc)::value, SET = decltype(setc)::value;                // v = 0..NA-1: tile v / (NA / 2), chunk v % (NA / 2)
    areg[SET][v] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, a_voff, a_soff + (v / (NA / 2)) * a_tile + (v % (NA / 2)) * a_chunk, 0));
  };
  // pairs are numbered (tap, channel block); the address of pair k+2 advances by one channel block per step, the pixel offset is
  // recomputed only when the tap changes (never for a 1x1 convolution): no integer divisions in the loop
  int tap2 = 0, sidx2 = 0;                               // (tap, channel block) of the pair whose activations are loaded next
  int sidx1 = 0;                                         // channel block of the pair loaded last (BNL: the one a step normalises and splits)
  unsigned voff2 = OOB;
  auto advance = [&]() {
    sidx1 = sidx2;
    if (++sidx2 == spt) {
      sidx2 = 0;
      ++tap2;
      if (tap2 < ntaps) {
        voff2 = tap_voff(tap2);
      } else if (CHAIN && tap2 == ntaps && ld_lin + G < total) {       // on to the workgroup's next tile
        ld_lin += G;
        set_load_side(ld_lin);
        tap2 = 0;
        voff2 = tap_voff(0);
      } else {
        voff2 = OOB;                                     // past the last pair: zeros (the weight offset leaves its buffer's range)
      }
    }
  };

  const int eb = amax_exponent(amax_read(in_amax));
  const float sb = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, scale_of(eb))));    // an SGPR
  int ea = 0, ea_y = -1;                                 // the weight exponent of filter set ea_y

  // pair 0 through the plain split into LDS buffer 0, pair 1 into register set 1 (+ its weights into areg)
  auto prologue = [&](int lin) {
    ld_lin = lin;
    set_load_side(lin);
    tap2 = sidx2 = 0;
    voff2 = tap_voff(0);
    static_for<8 * TPT>([&](auto vc) { load_b(vc, std::integral_constant<int, 0>(), voff2, 0); });
    static_for<NA>([&](auto vc) { load_a(vc, std::integral_constant<int, 0>(), 0); });
    advance();
#pragma unroll
    for (int v = 0; v < NA; ++v) As[(v / (NA / 2)) * TILE_A + tid + NT * (v % (NA / 2))] = areg[0][v];
#pragma unroll
    for (int t = 0; t < TPT; ++t) {
      uint4 ph, pl;
      if constexpr (BNL) {                               // pair 0: channel block 0
        const float2* cf = reinterpret_cast<const float2*>(bnl_s) + (t + bt) * 16 + kh * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) breg[0][t][i] = fmaxf(__fmaf_rn(breg[0][t][i], cf[i].x, cf[i].y), 0.f);
      }
      if constexpr (BPACK) unpack8_f16(breg[0][t], ph, pl);
      else split8_f16(breg[0][t], sb, ph, pl);
      Bs[(t + bt) * TILE_B + (0 * 2 + kh) * BN + pix] = ph;
      Bs[(t + bt) * TILE_B + (1 * 2 + kh) * BN + pix] = pl;
    }
    // pair 1 (past the end of a one-step contraction the offset is out of range: zeros)
    {
      const int soff = sidx2 * chan_step, a_soff = (tap2 * spt + sidx2) * 2 * a_tile;
      static_for<8 * TPT>([&](auto vc) { load_b(vc, std::integral_constant<int, 1>(), voff2, soff); });
      static_for<NA>([&](auto vc) { load_a(vc, std::integral_constant<int, 1>(), a_soff); });
    }
    advance();
    __syncthreads();
  };

  const int l15 = lane & 15, lq = lane >> 4;
  const int a_frag = (lq >> 1) * TILE_A + (lq & 1) * BM + wm0 + l15;
  const int b_frag = (lq >> 1) * TILE_B + (lq & 1) * BN + wn0 + l15;

  // one step on pair k: SETN = (k + 1) & 1 holds pair k+1 (split and stored here), pair k+2 is loaded into set k & 1.
  // v_mfma_f32_32x32x16_f16: fragments [k16 tile][32-row block][piece] (lane l31 = row, lh = k-half), 24 MFMAs in the
  // order product-major, tile-minor (al bh | ah bl | ah bh): slots 0-11 one fragment read each, 0-7 two activation loads each, 8-11 the
  // weight stores of pair k+1 (to the OTHER LDS buffer: no barrier inside the step), 12-23 four split instructions each, 12-15 the weight
  // loads of pair k+2, 18 / 23 the activation stores
  const int l31 = lane & 31, lh = lane >> 5;
  auto step32 = [&](auto setn_c, int k) {
    constexpr int SETN = decltype(setn_c)::value, SETL = SETN ^ 1;
    constexpr int CUR = SETL * PAIR_CHUNKS, NXT = SETN * PAIR_CHUNKS;       // pair k sits in LDS buffer k & 1 = SETL, pair k+1 goes to the other
    f16x8 af[2][TMW][NP], bf[2][2][NP];
    auto rd_a = [&](int t, int i, int pl) { af[t][i][pl] = __builtin_bit_cast(f16x8, As[CUR + t * TILE_A + (pl * 2 + lh) * BM + wm0 + i * 32 + l31]); };
    auto rd_b = [&](int t, int j, int pl) { bf[t][j][pl] = __builtin_bit_cast(f16x8, Bs[CUR + t * TILE_B + (pl * 2 + lh) * BN + wn0 + j * 32 + l31]); };
    // fragment reads, RG = 2 + TMW per group: (al, bh) of tile 0, of tile 1, then (ah, bl) of tile 0, of tile 1 -- within a group a0 b0 b1 [a1],
    // the MFMAs' order
    constexpr int RG = 2 + TMW;
    auto read_frag = [&](auto rc) {
      constexpr int r = decltype(rc)::value, grp = r / RG, e = r % RG, t = grp & 1, pa = grp < 2 ? 1 : 0, pb = grp < 2 ? 0 : 1;
      if constexpr (e == 0) rd_a(t, 0, pa);
      else if constexpr (e == 1) rd_b(t, 0, pb);
      else if constexpr (e == 2) rd_b(t, 1, pb);
      else rd_a(t, 1, pa);
    };
    const int soff2 = sidx2 * chan_step;
    const int a_soff2 = (tap2 * spt + sidx2) * 2 * a_tile;
    constexpr int PRE = TMW == 2 ? RG : 2 * RG;                    // reads in front of the first MFMA (MFMA m needs read <= PRE + m - 1; with one
    static_for<PRE>([&](auto rc) { read_frag(rc); });              // block per wave a group of three reads lasts only two MFMAs)
    __builtin_amdgcn_sched_barrier(0);
    SplitF16 s0, s1;
    uint4 cfr[BNL ? 4 : 1];                                          // BNL: (sc, sh) of the 8 channels this thread stages for pair k+1
    const int cf_at = BNL ? sidx1 * 16 + bt * 8 + kh * 4 : 0;        // in two-channel uint4 rows
    // the staging work of one of the 24 issue slots (TMW = 2: one slot per MFMA; TMW = 1: two per MFMA): 0-7 two activation loads each,
    // 8-11 the weight stores of pair k+1 (to the OTHER LDS buffer: no barrier inside the step), 12-23 four split instructions each, 12-15 the
    // weight loads of pair k+2, 18 / 23 the activation stores
    auto filler = [&](auto sc) {
      constexpr int m = decltype(sc)::value;
      if constexpr (m < 8) {                                        // (256-row tile: one load per slot)
        if constexpr (TPT == 2) {
          load_b(std::integral_constant<int, 2 * m>(), std::integral_constant<int, SETL>(), voff2, soff2);
          load_b(std::integral_constant<int, 2 * m + 1>(), std::integral_constant<int, SETL>(), voff2, soff2);
        } else {
          load_b(std::integral_constant<int, m>(), std::integral_constant<int, SETL>(), voff2, soff2);
        }
      }
      if constexpr (m >= 8 && m < 8 + NA) As[NXT + ((m - 8) / (NA / 2)) * TILE_A + tid + NT * ((m - 8) % (NA / 2))] = areg[SETN][m - 8];
      if constexpr (BNL && m >= 8 && m < 12) cfr[m - 8] = bnl_s[cf_at + (m - 8)];          // one 16-byte LDS read per slot (a wave-uniform address)
      if constexpr (m >= 12) {
        constexpr int SPS = 2 * TPT;                                // split instructions per slot: 48 (24: one tile) over slots 12-23
        static_for<SPS>([&](auto kc) {
          constexpr int kk = (m - 12) * SPS + decltype(kc)::value;
          if constexpr (BPACK) {                                  // 16 (8) permutes instead of 48 (24) split instructions
            if constexpr (kk < 8) unpack_op_f16<kk>(breg[SETN][0], s0);
            else if constexpr (kk < 16 && TPT == 2) unpack_op_f16<kk - 8>(breg[SETN][TPT - 1], s1);
          } else {
            if constexpr (BNL && kk < 24 && kk % 6 < 2) {            // the two empty slots of a pair's six: normalise the pair's two elements
              constexpr int e = 2 * (kk / 6) + kk % 6;
              const uint4 c2 = cfr[e >> 1];
              const float sc = __builtin_bit_cast(float, (e & 1) ? c2.z : c2.x), sh = __builtin_bit_cast(float, (e & 1) ? c2.w : c2.y);
              breg[SETN][0][e] = fmaxf(__fmaf_rn(breg[SETN][0][e], sc, sh), 0.f);
            }
            if constexpr (kk < 24) split_op_f16<kk>(breg[SETN][0], sb, s0);
            else split_op_f16<kk - 24>(breg[SETN][TPT - 1], sb, s1);
          }
        });
      }
      if constexpr (m >= 12 && m < 12 + NA) load_a(std::integral_constant<int, m - 12>(), std::integral_constant<int, SETL>(), a_soff2);
      if constexpr (m == 18 && TPT == 2) {
        Bs[NXT + (0 * 2 + kh) * BN + pix] = make_uint4(s0.h[0], s0.h[1], s0.h[2], s0.h[3]);
        Bs[NXT + (1 * 2 + kh) * BN + pix] = make_uint4(s0.l[0], s0.l[1], s0.l[2], s0.l[3]);
      }
      if constexpr (m == 23 && TPT == 2) {
        Bs[NXT + TILE_B + (0 * 2 + kh) * BN + pix] = make_uint4(s1.h[0], s1.h[1], s1.h[2], s1.h[3]);
        Bs[NXT + TILE_B + (1 * 2 + kh) * BN + pix] = make_uint4(s1.l[0], s1.l[1], s1.l[2], s1.l[3]);
      }
      if constexpr (m == 23 && TPT == 1) {                          // the thread's one tile
        Bs[NXT + bt * TILE_B + (0 * 2 + kh) * BN + pix] = make_uint4(s0.h[0], s0.h[1], s0.h[2], s0.h[3]);
        Bs[NXT + bt * TILE_B + (1 * 2 + kh) * BN + pix] = make_uint4(s0.l[0], s0.l[1], s0.l[2], s0.l[3]);
      }
    };
    // 12 TMW MFMAs in the order product-major, tile-minor (al bh | ah bl | ah bh); the remaining fragment reads one per MFMA
    static_for<12 * TMW>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      constexpr int prod = m / (4 * TMW), rem = m % (4 * TMW), t = rem / (2 * TMW), i = (rem >> 1) % TMW, j = rem & 1;
      constexpr int pa = prod == 0 ? 1 : 0, pb = prod == 1 ? 1 : 0;
      acc32[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t][i][pa], bf[t][j][pb], acc32[i][j], 0, 0, 0);
      if constexpr (PRE + m < 4 * RG) read_frag(std::integral_constant<int, PRE + m>());
      if constexpr (TMW == 2) {
        filler(std::integral_constant<int, m>());
      } else {
        filler(std::integral_constant<int, 2 * m>());
        filler(std::integral_constant<int, 2 * m + 1>());
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    advance();
    __syncthreads();
  };
  // un-scale (an exact power of two, in two factors so that neither over- nor underflows) and hand over to the common epilogue
  auto epilogue = [&](int lin) {
    int bx, by, y, n;
    decode(lin, bx, by, y, n);
    if (y != ea_y) {
      ea = amax_exponent(amax_read(w_amax + (i64)y * PFST_AMAX_SUB));
      ea_y = y;
    }
    const int p0 = bx * BN, m0 = by * BM;
    float* const outn = out + (i64)n * out_bs;
    // the gated addend of a residual block's identity branch (conv_epilogue.h), this image's planes
    const float* const gsrc = gate.g ? gate.g + (i64)n * gate.g_bs : nullptr;
    const unsigned long long* const gmask = gate.g ? gate.mask + (i64)n * M * (P >> 6) : nullptr;
    const float ua = unscale_of(ea), ub = unscale_of(eb);
    {
#pragma unroll
      for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc32[i][j][r] = acc32[i][j][r] * ua * ub;
      if constexpr (BNB != 0) {
        // the reduction scratch is LDS buffer 1: the last step (odd, or the only one) read it and ended with a barrier; buffer 0 may
        // already hold the next tile's first pair
        static_assert(PAIR_CHUNKS * sizeof(uint4) >= (NT / 64) * PFST_ROWSUM_LDS_FLOATS * sizeof(float), "epilogue scratch must fit into one pair buffer");
        conv_epilogue<2, 2, WAVES_N, BN, BNB, true>(acc32, outn, bias, stats, stats_T, accumulate, M, P, m0, p0, wm0, wn0, bx, n, wid, lane, bnb,
                                                    reinterpret_cast<float*>(smem + PAIR_CHUNKS), gsrc, gmask, nullptr, nt_out);
        __syncthreads();                                 // before the next tile's first step stores into that buffer
      } else {
#ifdef PFST_STATS_LDSRED                           // A/B build: the forward statistics' row sums through LDS (as the fused BatchNorm-backward sums) instead of the DPP butterfly
        if (stats && !stats_mm && BMT != 64) {
          conv_epilogue<TMW, 2, WAVES_N, BN, 0, true>(acc32, outn, bias, stats, stats_T, accumulate, M, P, m0, p0, wm0, wn0, bx, n, wid, lane, PfstBnbArgs(),
                                                      reinterpret_cast<float*>(smem + PAIR_CHUNKS), gsrc, gmask, nullptr, nt_out);
          __syncthreads();
        } else
#endif
        conv_epilogue<TMW, 2, WAVES_N, BN>(acc32, outn, bias, stats, stats_T, accumulate, M, P, m0, p0, wm0, wn0, bx, n, wid, lane, PfstBnbArgs(), nullptr,
                                           gsrc, gmask, stats_mm, nt_out);
      }
#pragma unroll
      for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc32[i][j][r] = 0.f;
    }
  };

  // ---- the workgroup's tiles: blockIdx.x, + gridDim.x, ...  (every tile index below `total`; gridDim.x <= total)
  for (int lin = blockIdx.x; lin < total; lin += G) {
    if (!CHAIN || lin == (int)blockIdx.x) prologue(lin);
    for (int k = 0; k < KP; k += 2) {
      step32(std::integral_constant<int, 1>(), k);
      if (k + 1 < KP) step32(std::integral_constant<int, 0>(), k + 1);
    }
    epilogue(lin);
  }
}

template <int SHAPE, bool BPACK = false, bool ONE = false, int BMT = 128>
__global__ __launch_bounds__(BMT == 64 ? 256 : 2 * BMT, BMT == 256 ? 1 : 2) void conv_igemm_f16x3_kernel(
    const float* __restrict__ in, i64 in_bs, const uint4* __restrict__ wk4, const float* __restrict__ bias,
    float* __restrict__ out, i64 out_bs, int C, int Hi, int Wi, int M, int Ho, int Wo, int ks,
    int ca, int cb, int cc, int cdivv, int accumulate, float* __restrict__ stats, int stats_T,
    const float* __restrict__ w_amax, const float* __restrict__ in_amax, int Y, int Z, int chain, PfstResGate gate) {
  conv_igemm_f16x3_body<SHAPE, 0, BPACK, ONE, BMT>(in, in_bs, wk4, bias, out, out_bs, C, Hi, Wi, M, Ho, Wo, ks, ca, cb, cc, cdivv, accumulate, stats, stats_T,
                               w_amax, in_amax, Y, Z, chain, PfstBnbArgs(), gate);
}
__global__ __launch_bounds__(512, 1) void conv_igemm_f16x3_bnl_kernel(
    const float* __restrict__ in, i64 in_bs, const uint4* __restrict__ wk4, const float* __restrict__ bias,
    float* __restrict__ out, i64 out_bs, int C, int Hi, int Wi, int M, int Ho, int Wo, int ks,
    int ca, int cb, int cc, int cdivv, int accumulate, float* __restrict__ stats, int stats_T,
    const float* __restrict__ w_amax, const float* __restrict__ in_amax, int Y, int Z, int chain, const float4* __restrict__ bnl) {
  conv_igemm_f16x3_body<32, 0, false, true, 256, true>(in, in_bs, wk4, bias, out, out_bs, C, Hi, Wi, M, Ho, Wo, ks, ca, cb, cc, cdivv, accumulate, stats,
                                                       stats_T, w_amax, in_amax, Y, Z, chain, PfstBnbArgs(), PfstResGate(), bnl);
}
template <int BNB, bool ONE = false, int BMT = 128>
__global__ __launch_bounds__(2 * BMT, BMT == 128 ? 2 : 1) void conv_igemm_f16x3_bnb_kernel(
    const float* __restrict__ in, i64 in_bs, const uint4* __restrict__ wk4, const float* __restrict__ bias,
    float* __restrict__ out, i64 out_bs, int C, int Hi, int Wi, int M, int Ho, int Wo, int ks,
    int ca, int cb, int cc, int cdivv, int accumulate, float* __restrict__ stats, int stats_T,
    const float* __restrict__ w_amax, const float* __restrict__ in_amax, int Y, int Z, int chain, PfstBnbArgs bnb, PfstResGate gate) {
  conv_igemm_f16x3_body<32, BNB, false, ONE, BMT>(in, in_bs, wk4, bias, out, out_bs, C, Hi, Wi, M, Ho, Wo, ks, ca, cb, cc, cdivv, accumulate, stats, stats_T,
                                 w_amax, in_amax, Y, Z, chain, bnb, gate);
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Weight gradient of a 1x1 convolution / the grouped transform-domain products of a Winograd layer on the same arithmetic:
//     dW[m][j] += sum_p dY[m][p] X[j][p]        (m: output channel, j: input channel, p: pixel or tile of one image)
// Both operands are activations and pixel-contiguous: a thread stages 8 consecutive pixels of one row per operand (two 16-byte buffer
// loads), both are split in registers (2 x 24 VALU per K=16 step) and written as [piece][k-half][row] images.  The loads run TWO tiles
// ahead into alternating register sets (conv_wgrad_split_q_pipe_kernel's scheme): step k multiplies LDS buffer k & 1 (12 MFMAs
// 32x32x16: al bh | ah bl | ah bh for the 2 x 2 blocks of a wave), splits tile k+1 in their gaps, stores it to the other buffer and
// issues the loads of tile k+2.  Split-K over pixel chunks and images, fp32 atomics into dW, all tiles of a K slice on one XCD.
// ---------------------------------------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

// (Round 5 removed the two earlier forms of this kernel -- K = 16 per step with 8-pixel row pieces, then K = 32 "pair" steps -- together with their
// PFST_F16X3_WGRAD_PAIR / _LINE switches: the whole-line kernel below has been the only one launched since round 3; git history has them.)
// PACK: both operands are stored pre-split (the Winograd-domain V and dM): permutes instead of split instructions.
// BMT = 256: 256 rows of dY per workgroup (512 threads, one workgroup per CU): an X tile is fetched once for 256 rows of dY.
// K = 32 per step (a pair of K=16 tiles per LDS buffer, the implicit GEMM's step32 shape: 24 MFMAs per barrier) with WHOLE-LINE loads.  The earlier
// form loaded 8 consecutive pixels of a row per thread (two lanes per row): one load instruction
// touched 32 rows x 32 bytes, a quarter of each 128-byte line, and the other three quarters came from three more instructions -- the vector
// L1 had to hold every line across four instructions; timing with fully coalesced (wrong) addresses: 51.8 -> 44.1 ms per step.  Here a
// row's 32 pixels of a pair (128 bytes) are ONE line read by 8 adjacent lanes, a load instruction covers 8 whole lines: thread
// (row slot r = tid / 8, segment s = tid % 8) loads pixels 4 s .. 4 s + 3 of rows r, r + RPP, ... (RPP = threads / 8 rows per pass), i.e.
// half a 16-byte LDS chunk of tile s / 4, k-half (s / 2) % 2, and stores its two 8-byte piece halves with ds_write_b64.  The planes are
// padded (+2 chunks per (piece, half) plane, +4 per tile) so that the 16 lanes of a ds_write_b64 group -- two rows x eight segments --
// fall into 16 different bank pairs; the fragment reads stay 16 consecutive chunks per group.
struct Split4 {
  unsigned h[2], l[2];
  float t0[2], t1[2];
};
#ifdef PFST_F16X3_SPLIT6
template <int K>
__device__ __forceinline__ void split4_op_f16(const float (&v)[4], float s, Split4& st) {        // K = 0..11: split_op_f16 on two pairs
  constexpr int q = K / 6, op = K % 6;
  if constexpr (op == 0) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(st.t0[q]) : "s"(s), "v"(v[2 * q]));
  else if constexpr (op == 1) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(st.t1[q]) : "s"(s), "v"(v[2 * q + 1]));
  else if constexpr (op == 2) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(st.h[q]) : "v"(st.t0[q]), "v"(st.t1[q]));
  else if constexpr (op == 3)
    asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(st.t0[q]) : "v"(v[2 * q]), "s"(s), "v"(st.h[q]));
  else if constexpr (op == 4)
    asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(st.t1[q]) : "v"(v[2 * q + 1]), "s"(s), "v"(st.h[q]));
  else asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(st.l[q]) : "v"(st.t0[q]), "v"(st.t1[q]));
}
#else
template <int K>
__device__ __forceinline__ void split4_op_f16(const float (&v)[4], float s, Split4& st) {        // K = 0..11: split_op_f16 on two pairs
  constexpr int q = K / 6, op = K % 6;
  if constexpr (op == 2) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(st.h[q]) : "v"(v[2 * q]), "s"(s));
  else if constexpr (op == 3) asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(st.h[q]) : "v"(v[2 * q + 1]), "s"(s));
  else if constexpr (op == 4)
    asm volatile("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(st.l[q]) : "v"(v[2 * q]), "s"(s), "v"(st.h[q]));
  else if constexpr (op == 5)
    asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(st.l[q]) : "v"(v[2 * q + 1]), "s"(s), "v"(st.h[q]));
}
#endif
template <int K>
__device__ __forceinline__ void unpack4_op_f16(const float (&v)[4], Split4& st) {                 // K = 0..3: the permutes of a pre-split quad
  constexpr int q = K & 1;
  if constexpr (K < 2) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(st.h[q]) : "v"(v[2 * q + 1]), "v"(v[2 * q]), "s"(0x05040100u));
  else asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(st.l[q]) : "v"(v[2 * q + 1]), "v"(v[2 * q]), "s"(0x07060302u));
}

// BNL: x is the PRE-normalisation output of the conv -> BN -> ReLU layer whose (never written) result this 1x1 convolution read in forward
// (conv_igemm_f16x3_body BNL): the rows of X are normalised -- max(fma(x, sc, sh), 0), (sc, sh) of row j from bnl[J] = (mean, invstd, sc, sh) --
// between their load and their split; a thread's rows are fixed for the whole launch, so their coefficients sit in registers.
template <bool PACK, int BMT = 128, bool BNL = false>
__global__ __launch_bounds__(2 * BMT, BMT == 128 ? 2 : 1) void conv_wgrad_f16x3_line_kernel(
    const float* __restrict__ x, i64 x_bs, const float* __restrict__ dy, i64 dy_bs, float* __restrict__ dw,
    int J, int M, int P, int chunks, int chunk_len, int N, i64 x_gs, i64 dy_gs, i64 dw_gs, int gx, int gy, int gz,
    const float* __restrict__ x_amax, const float* __restrict__ dy_amax, const float4* __restrict__ bnl) {
  static_assert(!BNL || !PACK, "a pre-split operand is already normalised");
  constexpr int BM = BMT, BJ = 128, WM = 64, WAVES_N = 2, WN = 64, TM = 2, TN = 2, NT = 2 * BM;
  constexpr int RPP = NT / 8;                               // rows per load pass
  constexpr int PA = BM / RPP, PB = BJ / RPP;               // passes (= loads per thread and pair) over the dY / X rows: 4 + 4, or 4 + 2
  constexpr int ROWS_A = BM + 2, ROWS_B = BJ + 2;           // padded (piece, half) planes, in 16-byte chunks
  constexpr int TILE_A = 4 * ROWS_A + 4, TILE_B = 4 * ROWS_B + 4;
  __shared__ uint4 As[2][2 * TILE_A];                     // [buffer][tile][piece][k-half][row]
  __shared__ uint4 Bs[2][2 * TILE_B];

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wid / WAVES_N) * WM, wn0 = (wid % WAVES_N) * WN;
  int bx, by, bz;
  {
    const int lin = blockIdx.x, tiles = gx * gy, z8 = gz & ~7;
    if (lin < tiles * z8) {
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
  const int ea = amax_exponent(amax_read(dy_amax)), eb = amax_exponent(amax_read(x_amax));
  const float sa = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, scale_of(ea))));
  const float sb = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, scale_of(eb))));
  // rows past M / J lie outside the buffers' ranges and read zeros (the range check includes the scalar offset)
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy), 0, M * P * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, J * P * 4, 0x00020000);

  const int rslot = tid >> 3, seg = tid & 7;               // staging role: row slot, 16-byte segment of the pair's 128-byte line
  constexpr unsigned OOB = 0x80000000u;
  const unsigned a_voff = 4u * ((unsigned)(m0 + rslot) * (unsigned)P) + 16u * seg;
  const unsigned b_voff = 4u * ((unsigned)(j0 + rslot) * (unsigned)P) + 16u * seg;
  const int row_step = RPP * P * 4;                        // bytes between the rows of two passes (scalar offset)
  // this thread's half chunk: tile seg / 4, k-half (seg / 2) % 2, 8-byte half seg % 2 of the chunk of row `rslot` (+ RPP per pass)
  const int a_half = 2 * ((seg >> 2) * TILE_A + ((seg >> 1) & 1) * ROWS_A + rslot) + (seg & 1);     // in 8-byte units
  const int b_half = 2 * ((seg >> 2) * TILE_B + ((seg >> 1) & 1) * ROWS_B + rslot) + (seg & 1);

  float la[2][PA][4], lb[2][PB][4];                 // [register set][pass][4 pixels]
  float bsc[BNL ? PB : 1], bsh[BNL ? PB : 1];       // BNL: (sc, sh) of this thread's X rows (rows past J: zeros, like their data)
  if constexpr (BNL) {
#pragma unroll
    for (int p = 0; p < PB; ++p) {
      const int j = j0 + rslot + p * RPP;
      const float4 cf = j < J ? bnl[j] : make_float4(0.f, 0.f, 0.f, 0.f);
      bsc[p] = cf.z;
      bsh[p] = cf.w;
    }
  }
  auto normalise = [&](float (&v)[4], int p) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(__fmaf_rn(v[e], bsc[p], bsh[p]), 0.f);
  };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // load q = 0 .. PA + PB - 1 of the pair starting at pixel pk0: dY passes first.  P % 4 == 0 and chunk_len % 16 == 0: a quad is entirely
  // in or out of the chunk (out: zeros)
  auto load_quad = [&](auto qc, auto setc, int pk0) {
    constexpr int q = decltype(qc)::value, SET = decltype(setc)::value, opb = q >= PA, pass = opb ? q - PA : q;
    const bool v = pk0 + 4 * seg < pend;
    const float4 w = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(opb ? b_rsrc : a_rsrc, v ? (opb ? b_voff : a_voff) : OOB,
                                                                                      pk0 * 4 + pass * row_step, 0));
    float (&dst)[4] = opb ? lb[SET][pass] : la[SET][pass];
    dst[0] = w.x; dst[1] = w.y; dst[2] = w.z; dst[3] = w.w;
  };
  // the two 8-byte stores of half chunk q (pieces h, l: planes 2 apart)
  auto store_half = [&](auto qc, int buf, const Split4& st) {
    constexpr int q = decltype(qc)::value, opb = q >= PA, pass = opb ? q - PA : q;
    if constexpr (opb) {
      uint2* b2 = reinterpret_cast<uint2*>(&Bs[buf][0]);
      b2[b_half + 2 * pass * RPP] = make_uint2(st.h[0], st.h[1]);
      b2[b_half + 2 * (pass * RPP + 2 * ROWS_B)] = make_uint2(st.l[0], st.l[1]);
    } else {
      uint2* a2 = reinterpret_cast<uint2*>(&As[buf][0]);
      a2[a_half + 2 * pass * RPP] = make_uint2(st.h[0], st.h[1]);
      a2[a_half + 2 * (pass * RPP + 2 * ROWS_A)] = make_uint2(st.l[0], st.l[1]);
    }
  };
  const int l31 = lane & 31, lh = lane >> 5;

  // prologue: pair 0 through the plain split into buffer 0, pair 1 into register set 1
  static_for<PA + PB>([&](auto qc) { load_quad(qc, std::integral_constant<int, 0>(), pbeg); });
  static_for<PA + PB>([&](auto qc) {
    constexpr int q = decltype(qc)::value, opb = q >= PA, pass = opb ? q - PA : q;
    Split4 st;
    if constexpr (BNL && opb) normalise(lb[0][pass], pass);
    static_for<PACK ? 4 : 12>([&](auto kc) {
      if constexpr (PACK) unpack4_op_f16<decltype(kc)::value>(opb ? lb[0][pass] : la[0][pass], st);
      else split4_op_f16<decltype(kc)::value>(opb ? lb[0][pass] : la[0][pass], opb ? sb : sa, st);
    });
    store_half(qc, 0, st);
  });
  static_for<PA + PB>([&](auto qc) { load_quad(qc, std::integral_constant<int, 1>(), pbeg + 32); });
  __syncthreads();

  // one step on LDS buffer CUR = k & 1 (register set NXT holds pair k+1; pair k+2 is loaded into set CUR).  24 MFMAs product-major,
  // tile-minor; slots 0-11 one fragment read, 0 .. PA+PB-1 one load, three slots of four split instructions per half chunk (its two
  // stores behind the MFMA of the third slot).
  auto step = [&](auto curc, int kp) {
    constexpr int CUR = decltype(curc)::value, NXT = CUR ^ 1;
    f16x8 af[2][TM][NP], bf[2][TN][NP];
    auto rd_a = [&](int t, int i, int pl) { af[t][i][pl] = __builtin_bit_cast(f16x8, As[CUR][t * TILE_A + (pl * 2 + lh) * ROWS_A + wm0 + i * 32 + l31]); };
    auto rd_b = [&](int t, int j, int pl) { bf[t][j][pl] = __builtin_bit_cast(f16x8, Bs[CUR][t * TILE_B + (pl * 2 + lh) * ROWS_B + wn0 + j * 32 + l31]); };
    auto read_frag = [&](auto rc) {
      constexpr int r = decltype(rc)::value, grp = r >> 2, e = r & 3, t = grp & 1, pa = grp < 2 ? 1 : 0, pb = grp < 2 ? 0 : 1;
      if constexpr (e == 0) rd_a(t, 0, pa);
      else if constexpr (e == 1) rd_b(t, 0, pb);
      else if constexpr (e == 2) rd_b(t, 1, pb);
      else rd_a(t, 1, pa);
    };
    static_for<4>([&](auto rc) { read_frag(rc); });
    __builtin_amdgcn_sched_barrier(0);
    Split4 st[2];
    const int pk2 = pbeg + (kp + 2) * 32;
    static_for<24>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      constexpr int prod = m >> 3, t = (m >> 2) & 1, i = (m >> 1) & 1, j = m & 1;
      constexpr int pa = prod == 0 ? 1 : 0, pb = prod == 1 ? 1 : 0;
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t][i][pa], bf[t][j][pb], acc[i][j], 0, 0, 0);
      if constexpr (m < 12) read_frag(std::integral_constant<int, 4 + m>());
      if constexpr (m < PA + PB) load_quad(mc, curc, pk2);
      constexpr int q = m / 3, part = m % 3;                          // half chunk q = 0 .. 7 in slots 3 q .. 3 q + 2
      if constexpr (q < PA + PB) {
        constexpr int opb = q >= PA, pass = opb ? q - PA : q;
        if constexpr (BNL && opb && part == 0) normalise(lb[NXT][pass], pass);       // in front of the half chunk's first split instructions
        static_for<4>([&](auto kc) {
          constexpr int k = part * 4 + decltype(kc)::value;
          if constexpr (PACK) {
            if constexpr (k < 4) unpack4_op_f16<k>(opb ? lb[NXT][pass] : la[NXT][pass], st[q & 1]);
          } else {
            split4_op_f16<k>(opb ? lb[NXT][pass] : la[NXT][pass], opb ? sb : sa, st[q & 1]);
          }
        });
        if constexpr (part == 2) store_half(std::integral_constant<int, q>(), NXT, st[q & 1]);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    __syncthreads();
  };
  const int KPAIRS = (pend - pbeg + 31) / 32;
  for (int kp = 0; kp < KPAIRS; kp += 2) {
    step(std::integral_constant<int, 0>(), kp);
    if (kp + 1 < KPAIRS) step(std::integral_constant<int, 1>(), kp + 1);
  }
  const float ua = unscale_of(ea), ub = unscale_of(eb);
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

// grid of the tile-chain GEMM.  A workgroup walks tiles blockIdx.x, + gridDim.x, ... as one software pipeline; how many tiles it gets is a
// balance: long chains amortise the one prologue (4-6 K-steps' worth, tools/gemm_k_sweep.py), but the hardware can only even out the
// CUs' speeds by handing out whole workgroups, and a last round of workgroups that fills a fraction of the slots wastes the rest.
// -> of the grids total / t (t = 1..8) and one-workgroup-per-slot, the one whose rounds waste the least (ties: the longer chains).
int f16x3_slots_override = 0;                             // pfst_f16x3_set_slots
unsigned f16x3_grid(i64 total, bool chainable, int wg_per_cu = 2) {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  }
  // resident workgroups: two 256-thread workgroups per CU, one of the 512-thread 256-row tile (the override counts 256-thread slots)
  int slots = (f16x3_slots_override > 0 ? f16x3_slots_override : 2 * cus) * wg_per_cu / 2;
  if (slots < 1) slots = 1;
  if (!chainable || total <= slots) return (unsigned)total;
  i64 best_g = total;
  double best_waste = 1e30;
  for (int t = 0; t <= 32; ++t) {                         // t = 0: one workgroup per slot
    const i64 g = t == 0 ? slots : (total + t - 1) / t;
    if (g < slots) continue;
    const i64 per_wg = (total + g - 1) / g, rounds = (g + slots - 1) / slots;
    if (per_wg > 8) continue;                             // measured: 4-8 tiles per workgroup beat both 1 and 32 at every K
    const double waste = (double)(rounds * per_wg * slots) / (double)total;
    if (waste < best_waste - 1e-9 || (waste < best_waste + 1e-9 && g < best_g)) { best_waste = waste; best_g = g; }
  }
  return (unsigned)best_g;
}


}  // namespace

// number of resident workgroup slots the tile-chain grid is sized for (0: two per CU of the current device).  A test hook: with a few
// slots small problems run as chains of up to 8 tiles per workgroup, the shape the b = 8 x 1024^2 launches have.
extern "C" int pfst_f16x3_set_slots(int slots) {
  PFST_CHECK_ARG(slots >= 0);
  f16x3_slots_override = slots;
  return PFST_OK;
}

// the grid pfst_conv_igemm_f16x3 / pfst_wino_gemm_f16x3 launch for `total_tiles` 128 x 128 tiles (chainable: a 1x1 / Winograd-domain
// contraction with an even number of K = 32 steps): total_tiles itself, or fewer workgroups that each walk up to 8 tiles
extern "C" int pfst_f16x3_chain_grid(long long total_tiles, int chainable) {
  if (total_tiles <= 0 || total_tiles >= (1ll << 31)) return 0;
  return (int)f16x3_grid(total_tiles, chainable != 0);
}

// Test hook: the split exactly as the GEMM loops issue it (split_op_f16 / split4_op_f16, the pinned instruction sequences) and as the
// prologues and packing kernels do it (split8_f16, plain code), on n values (n % 8 == 0) scaled from the slot group `amax`:
// pieces_loop[i], pieces_loop4[i], pieces_plain[i] = (h | l << 16) of x[i].
namespace {
__global__ __launch_bounds__(256) void f16x3_split_probe_kernel(const float* __restrict__ x, i64 n8, const float* __restrict__ amax,
                                                                unsigned* __restrict__ pieces_loop, unsigned* __restrict__ pieces_loop4,
                                                                unsigned* __restrict__ pieces_plain) {
  const float s = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, scale_of(amax_exponent(amax_read(amax))))));
  for (i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x; g < n8; g += (i64)gridDim.x * blockDim.x) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = x[g * 8 + j];
    SplitF16 st;
    static_for<24>([&](auto kc) { split_op_f16<decltype(kc)::value>(v, s, st); });
    uint4 ph, pl;
    split8_f16(v, s, ph, pl);
    const unsigned hp[4] = {ph.x, ph.y, ph.z, ph.w}, lp[4] = {pl.x, pl.y, pl.z, pl.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      pieces_loop[g * 8 + 2 * q] = (st.h[q] & 0xffffu) | (st.l[q] << 16);
      pieces_loop[g * 8 + 2 * q + 1] = (st.h[q] >> 16) | (st.l[q] & 0xffff0000u);
      pieces_plain[g * 8 + 2 * q] = (hp[q] & 0xffffu) | (lp[q] << 16);
      pieces_plain[g * 8 + 2 * q + 1] = (hp[q] >> 16) | (lp[q] & 0xffff0000u);
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const float v4[4] = {v[4 * half], v[4 * half + 1], v[4 * half + 2], v[4 * half + 3]};
      Split4 s4;
      static_for<12>([&](auto kc) { split4_op_f16<decltype(kc)::value>(v4, s, s4); });
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        pieces_loop4[g * 8 + 4 * half + 2 * q] = (s4.h[q] & 0xffffu) | (s4.l[q] << 16);
        pieces_loop4[g * 8 + 4 * half + 2 * q + 1] = (s4.h[q] >> 16) | (s4.l[q] & 0xffff0000u);
      }
    }
  }
}
}  // namespace

extern "C" int pfst_f16x3_split_probe(const float* x, long long n, const float* amax, unsigned* pieces_loop, unsigned* pieces_loop4,
                                      unsigned* pieces_plain, pfst_stream_t stream) {
  PFST_CHECK_ARG(x && amax && pieces_loop && pieces_loop4 && pieces_plain && n > 0 && n % 8 == 0);
  hipLaunchKernelGGL(f16x3_split_probe_kernel, dim3(ew_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, x, (i64)(n / 8), amax, pieces_loop,
                     pieces_loop4, pieces_plain);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

// max |x| of `planes` planes of `n` floats (plane_stride apart) into the slot GROUP (1024 floats, amax.h) number plane * slot_stride; the
// groups must have been zeroed (or hold an earlier maximum to extend).  pfst_absmax is the stand-alone form; producers of GEMM operands write their slots themselves.
extern "C" int pfst_absmax(const float* x, long long n, int planes, long long plane_stride, int slot_stride, float* slots,
                           pfst_stream_t stream) {
  PFST_CHECK_ARG(x && slots && n > 0 && planes > 0 && planes <= 65535 && (slot_stride == 0 || slot_stride == 1));
  int gx = (int)((n / 4 + 255) / 256);
  if (gx < 1) gx = 1;
  const int cap = planes > 1 ? 256 : 2048;
  if (gx > cap) gx = cap;
  hipLaunchKernelGGL(absmax_kernel, dim3(gx, planes), dim3(256), 0, (hipStream_t)stream, x, (i64)n, (i64)plane_stride, slot_stride, slots);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

// w [sets][Cout][Cin][T] -> two-piece fp16 K-major images (4 bytes per weight per layout), scaled per set by the power of two that
// amax[set] implies.  sets > 1: the transform-domain filter sets of a Winograd layer (T = 1, `set` floats apart).
extern "C" int pfst_conv_pack_weight_f16x2(const float* w, void* wk4_fprop, void* wk4_dgrad, int Cout, int Cin, int T, int sets,
                                           const float* amax, pfst_stream_t stream) {
  PFST_CHECK_ARG(w && amax && (wk4_fprop || wk4_dgrad) && Cout > 0 && Cin > 0 && (T == 1 || T == 9) && sets >= 1 && sets <= 65535);
  PFST_CHECK_ARG(!wk4_fprop || Cin % 16 == 0);
  PFST_CHECK_ARG(!wk4_dgrad || Cout % 16 == 0);
  PFST_CHECK_ARG(sets == 1 || T == 1);
  const i64 n = (i64)Cout * Cin * T;
  int gx = ew_grid(n / 8 + 1);
  if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(pack_weight_f16x2_kernel, dim3(gx, sets), dim3(256), 0, (hipStream_t)stream, w, (uint4*)wk4_fprop, (uint4*)wk4_dgrad,
                     Cout, Cin, T, n, 4 * n / 16, amax);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

extern "C" int pfst_conv_pack_weight_f16x2_batched(const pfst_weight_job_t* jobs_host, const pfst_weight_job_t* jobs_dev, int njobs,
                                                   pfst_stream_t stream) {
  PFST_CHECK_ARG(jobs_host && jobs_dev && njobs > 0);
  i64 blocks = 0;
  for (int j = 0; j < njobs; ++j) {
    const pfst_weight_job_t& J = jobs_host[j];
    PFST_CHECK_ARG(J.src && J.amax_f && (J.dst_f || J.dst_d) && J.Cout > 0 && J.Cin > 0 && (J.T == 1 || J.T == 9) && J.sets >= 1);
    PFST_CHECK_ARG((!J.dst_f || J.Cin % 16 == 0) && (!J.dst_d || J.Cout % 16 == 0) && (J.sets == 1 || J.T == 1) && J.first_block == blocks);
    blocks += weight_job_blocks(J, 1);
  }
  PFST_CHECK_ARG(blocks < (1ll << 31));
  hipLaunchKernelGGL(pack_weight_f16x2_batched_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, jobs_dev, njobs);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

// fprop (mode 0) / dgrad (mode 1) on the f16x3 kernel; in_amax: one slot holding max |in|.  Needs C % 32 == 0 (1x1: C % 16 == 0) and M > 32 (33 ... 64 rows: the 64-row tile).
extern "C" int pfst_conv_igemm_f16x3(const float* in, long long in_bs, const void* wk4, const float* w_amax, const float* in_amax,
                                     const float* bias, float* out, long long out_bs, int N, int C, int Hi, int Wi, int M, int Ho, int Wo,
                                     int ksize, int stride, int dil, int pad, int mode, int accumulate, float* stats,
                                     const pfst_bnb_fuse_t* bnb, const float* gate_dy, long long gate_dy_bs,
                                     const unsigned long long* gate_mask, int stats_minmax, const float* bnl, pfst_stream_t stream) {
  PFST_CHECK_ARG(in && wk4 && w_amax && in_amax && out && N > 0 && C > 0 && M > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0);
  // bnl: the input is normalised as it is loaded (1x1 forward launches on the 256-row tile; C <= 512 coefficient rows in LDS)
  PFST_CHECK_ARG(!bnl || (mode == 0 && ksize == 1 && stride == 1 && pad == 0 && M % 256 == 0 && C <= 2048 && !bias && !gate_dy && !(bnb && bnb->x) &&
                          !accumulate));
  // stats_minmax: `stats` has room for 4 * M * slots floats and also receives the per-channel (minimum, maximum) partials behind the sums
  PFST_CHECK_ARG(!stats_minmax || (stats && !bias && !(bnb && bnb->x)));
  PfstResGate gate;
  if (gate_dy) {
    // out = conv + (bit ? gate_dy : 0): the epilogue's whole-tile path, the mask's 256-element groups, one writer (no old values)
    PFST_CHECK_ARG(gate_mask && mode == 1 && !accumulate && !bias && !stats && M % 128 == 0 && ((i64)Ho * Wo) % 256 == 0 &&
                   gate_dy_bs >= (i64)M * Ho * Wo);
    gate.g = gate_dy;
    gate.g_bs = gate_dy_bs;
    gate.mask = gate_mask;
  }
  PFST_CHECK_ARG(!bnb || (bnb->x && bnb->x_bs >= (i64)M * Ho * Wo && (!bnb->y || bnb->y_bs >= (i64)M * Ho * Wo)));
  PFST_CHECK_ARG(!bnb || !bnb->y_mask || (bnb->y && bnb->relu && ((i64)Ho * Wo) % 256 == 0));      // the gate bits of a residual layer's y
  PFST_CHECK_ARG((ksize == 1 || ksize == 3) && (stride == 1 || stride == 2) && dil >= 1 && pad >= 0 && (mode == 0 || mode == 1));
  PFST_CHECK_ARG(in_bs >= (i64)C * Hi * Wi && out_bs >= (i64)M * Ho * Wo && N <= 65535);
  // 32-bit buffer ranges and offsets (the out-of-range marker 0x80000000 is ADDED to them): one image of either operand and the weight
  // image must stay below 2 GB, as in pfst_wino_gemm_f16x3 / pfst_conv_wgrad_f16x3
  PFST_CHECK_ARG((i64)C * Hi * Wi * 4 < (1ll << 31) && (i64)M * Ho * Wo * 4 < (1ll << 31) && (i64)ksize * ksize * C * M * 4 < (1ll << 31));
  // a 1x1 convolution may end in half a channel block: the loads of the 16 missing channels (activations and weight chunks alike) lie
  // outside their buffers' ranges and return zeros (the range check of gfx950 includes the scalar offset: tools/probes/soffset_range_probe.hip)
  if ((C % 32 != 0 && !(ksize == 1 && C % 16 == 0)) || M <= 32) {
    pfst_set_error(__FILE__, __LINE__, "f16x3 kernel needs C % 32 == 0 (1x1: C % 16 == 0) and more than 32 output channels (use pfst_conv_igemm_split)");
    return PFST_ERR_UNSUPPORTED;
  }
  const bool small = M <= 64;                       // 33 ... 64 output rows: the 64-row tile (one 32-row block per wave)
  PFST_CHECK_ARG(!small || !(bnb && bnb->x));
  const int span = (ksize - 1) * dil;
  if (mode == 0) {
    PFST_CHECK_ARG(Ho == (Hi + 2 * pad - span - 1) / stride + 1 && Wo == (Wi + 2 * pad - span - 1) / stride + 1);
  } else {
    PFST_CHECK_ARG(Hi == (Ho + 2 * pad - span - 1) / stride + 1 && Wi == (Wo + 2 * pad - span - 1) / stride + 1);
  }
  int a, b, c, d;
  if (mode == 0) { a = stride; b = dil; c = -pad; d = 1; } else { a = 1; b = -dil; c = pad; d = stride; }
  const int stats_T = N * (int)cdiv((i64)Ho * Wo, BN) * 2;      // two pixel-waves per tile at every tile height (= pfst_conv_stats_slots for M > 64)
  const bool one = !small && ksize == 1 && stride == 1 && pad == 0;      // pixel-to-pixel: the tile-chain variant
  const bool big = one && M % 256 == 0;                              // 256-row tiles, 512 threads, one workgroup per CU (measured and not kept: the 128-row
                                                                     // tile for K <= 512 launches with a memory-heavy epilogue, profiles/r05_ab_heavy_epilogue.txt)
  const i64 total = (i64)cdiv((i64)Ho * Wo, BN) * cdiv(M, big ? 256 : small ? 64 : 128) * N;
  PFST_CHECK_ARG(total < (1ll << 31));
  const dim3 grid(f16x3_grid(total, one && ((C + 31) / 32) % 2 == 0, big ? 1 : 2));
  const int chain = 1 | (stats_minmax ? 2 : 0) | 4;            // bit 0: tile chains, bit 1: (min, max) partials, bit 2: nt stores for short contractions
  if (bnb && bnb->x) {
    // the fused sums use the epilogue's full-tile store path: whole row tiles, no bias, no forward statistics
    PFST_CHECK_ARG(M % 128 == 0 && !bias && !stats && bnb->coef && bnb->partials);
#define PFST_LAUNCH_F16_BNB(MODE_, ONE_)                                                                                                    \
    hipLaunchKernelGGL((conv_igemm_f16x3_bnb_kernel<MODE_, ONE_>), grid, dim3(256), 0, (hipStream_t)stream, in, (i64)in_bs, (const uint4*)wk4, bias, \
                       out, (i64)out_bs, C, Hi, Wi, M, Ho, Wo, ksize, a, b, c, d, accumulate, stats, stats_T, w_amax, in_amax, 1, N, chain, *bnb, gate)
#define PFST_LAUNCH_F16_BNB_BIG(MODE_)                                                                                                      \
    hipLaunchKernelGGL((conv_igemm_f16x3_bnb_kernel<MODE_, true, 256>), grid, dim3(512), 0, (hipStream_t)stream, in, (i64)in_bs, (const uint4*)wk4, bias, \
                       out, (i64)out_bs, C, Hi, Wi, M, Ho, Wo, ksize, a, b, c, d, accumulate, stats, stats_T, w_amax, in_amax, 1, N, chain, *bnb, gate)
    if (big) {
      if (!bnb->relu) PFST_LAUNCH_F16_BNB_BIG(3);
      else if (bnb->y_mask) PFST_LAUNCH_F16_BNB_BIG(4);
      else if (bnb->y) PFST_LAUNCH_F16_BNB_BIG(2);
      else PFST_LAUNCH_F16_BNB_BIG(1);
    } else if (one) {
      if (!bnb->relu) PFST_LAUNCH_F16_BNB(3, true);
      else if (bnb->y_mask) PFST_LAUNCH_F16_BNB(4, true);
      else if (bnb->y) PFST_LAUNCH_F16_BNB(2, true);
      else PFST_LAUNCH_F16_BNB(1, true);
    } else {
      if (!bnb->relu) PFST_LAUNCH_F16_BNB(3, false);
      else if (bnb->y_mask) PFST_LAUNCH_F16_BNB(4, false);
      else if (bnb->y) PFST_LAUNCH_F16_BNB(2, false);
      else PFST_LAUNCH_F16_BNB(1, false);
    }
#undef PFST_LAUNCH_F16_BNB
#undef PFST_LAUNCH_F16_BNB_BIG
    PFST_CHECK_LAUNCH();
    return PFST_OK;
  }
  if (bnl)
    hipLaunchKernelGGL(conv_igemm_f16x3_bnl_kernel, grid, dim3(512), 0, (hipStream_t)stream, in, (i64)in_bs, (const uint4*)wk4, bias, out, (i64)out_bs, C, Hi, Wi,
                       M, Ho, Wo, ksize, a, b, c, d, accumulate, stats, stats_T, w_amax, in_amax, 1, N, chain, reinterpret_cast<const float4*>(bnl));
  else if (small)
    hipLaunchKernelGGL((conv_igemm_f16x3_kernel<32, false, false, 64>), grid, dim3(256), 0, (hipStream_t)stream, in, (i64)in_bs, (const uint4*)wk4, bias, out,
                       (i64)out_bs, C, Hi, Wi, M, Ho, Wo, ksize, a, b, c, d, accumulate, stats, stats_T, w_amax, in_amax, 1, N, chain, gate);
  else if (big)
    hipLaunchKernelGGL((conv_igemm_f16x3_kernel<32, false, true, 256>), grid, dim3(512), 0, (hipStream_t)stream, in, (i64)in_bs, (const uint4*)wk4, bias, out,
                       (i64)out_bs, C, Hi, Wi, M, Ho, Wo, ksize, a, b, c, d, accumulate, stats, stats_T, w_amax, in_amax, 1, N, chain, gate);
  else if (one)
    hipLaunchKernelGGL((conv_igemm_f16x3_kernel<32, false, true>), grid, dim3(256), 0, (hipStream_t)stream, in, (i64)in_bs, (const uint4*)wk4, bias, out,
                       (i64)out_bs, C, Hi, Wi, M, Ho, Wo, ksize, a, b, c, d, accumulate, stats, stats_T, w_amax, in_amax, 1, N, chain, gate);
  else
    hipLaunchKernelGGL(conv_igemm_f16x3_kernel<32>, grid, dim3(256), 0, (hipStream_t)stream, in, (i64)in_bs, (const uint4*)wk4, bias, out,
                       (i64)out_bs, C, Hi, Wi, M, Ho, Wo, ksize, a, b, c, d, accumulate, stats, stats_T, w_amax, in_amax, 1, N, chain, gate);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

// the (m+2)^2 transform-domain GEMMs of a Winograd layer as one grouped launch: V [X][N][K][T] with ONE slot group (max over all planes:
// the planes of a transformed activation lie within ~2^7 of each other, far inside the 2^18 full-precision window), U4 [X] packed sets
// with a group per set
extern "C" int pfst_wino_gemm_f16x3(const float* V, const void* U4, const float* u_amax, const float* v_amax, float* Mbuf, int N, int K,
                                    int M, int T, int m, int v_packed, pfst_stream_t stream) {
  PFST_CHECK_ARG(V && U4 && u_amax && v_amax && Mbuf && N > 0 && N <= 65535 && K > 0 && K % 32 == 0 && M > 64 && T > 0 && (m == 2 || m == 4));
  const int nx = (m + 2) * (m + 2);
  PFST_CHECK_ARG((i64)K * T * 4 < (1ll << 31) && (i64)M * T * 4 < (1ll << 31) && (i64)K * M * 4 < (1ll << 31));
  const bool big = M % 256 == 0;
  const i64 total = (i64)cdiv((i64)T, BN) * cdiv(M, big ? 256 : 128) * nx * N;
  PFST_CHECK_ARG(total < (1ll << 31));
  const dim3 grid(f16x3_grid(total, (K / 32) % 2 == 0, big ? 1 : 2));
  const int chain = 1 | 4;
  // v_packed: V holds pre-split elements (pfst_wino_input with pack_x_amax) and v_amax the bound they were scaled by
  if (big && v_packed)
    hipLaunchKernelGGL((conv_igemm_f16x3_kernel<32, true, true, 256>), grid, dim3(512), 0, (hipStream_t)stream, V, (i64)K * T, (const uint4*)U4,
                       (const float*)nullptr, Mbuf, (i64)M * T, K, 1, T, M, 1, T, 1, 1, 1, 0, 1, 0, (float*)nullptr, 0, u_amax, v_amax, nx, N, chain, PfstResGate());
  else if (big)
    hipLaunchKernelGGL((conv_igemm_f16x3_kernel<32, false, true, 256>), grid, dim3(512), 0, (hipStream_t)stream, V, (i64)K * T, (const uint4*)U4,
                       (const float*)nullptr, Mbuf, (i64)M * T, K, 1, T, M, 1, T, 1, 1, 1, 0, 1, 0, (float*)nullptr, 0, u_amax, v_amax, nx, N, chain, PfstResGate());
  else if (v_packed)
    hipLaunchKernelGGL((conv_igemm_f16x3_kernel<32, true, true>), grid, dim3(256), 0, (hipStream_t)stream, V, (i64)K * T, (const uint4*)U4,
                       (const float*)nullptr, Mbuf, (i64)M * T, K, 1, T, M, 1, T, 1, 1, 1, 0, 1, 0, (float*)nullptr, 0, u_amax, v_amax, nx, N, chain, PfstResGate());
  else
    hipLaunchKernelGGL((conv_igemm_f16x3_kernel<32, false, true>), grid, dim3(256), 0, (hipStream_t)stream, V, (i64)K * T, (const uint4*)U4, (const float*)nullptr,
                       Mbuf, (i64)M * T, K, 1, T, M, 1, T, 1, 1, 1, 0, 1, 0, (float*)nullptr, 0, u_amax, v_amax, nx, N, chain, PfstResGate());
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

// internal: dW[grp][M][J] += sum over images and pixels; x [grp][N][J][P], dy [grp][N][M][P]; needs P % 4 == 0, M > 64
int pfst_wgrad_f16x3_launch(const float* x, i64 x_bs, const float* dy, i64 dy_bs, float* dw, int N, int J, int M, int P, int groups,
                            i64 x_gs, i64 dy_gs, i64 dw_gs, const float* x_amax, const float* dy_amax, int packed, hipStream_t s, const float* bnl) {
  PFST_CHECK_ARG(!bnl || (!packed && groups == 1));             // bnl: the rows of x are normalised as they are loaded (coef [J][4])
  const float4* const bnl4 = reinterpret_cast<const float4*>(bnl);
  const bool big = M % 256 == 0;                                // 256 rows of dY per workgroup (512 threads, one workgroup per CU)
  const int bm = big ? 256 : 128;
  const int tiles = cdiv(J, 128) * cdiv(M, bm) * groups;
  // split-K chunking: whole rounds of resident workgroups (2 per CU; the 256-row tile: 1)
  const double slots = big ? 256.0 : 256.0 * 2;
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
  const int gx = cdiv(J, 128), gy = cdiv(M, bm), gz = N * groups * chunks;
  PFST_CHECK_ARG((i64)gx * gy * gz < (1ll << 31));
  const i64 elems = (i64)M * J;
  bool det_ok;
  float* const ws = wgrad_det_scratch(elems, gz, s, det_ok);          // deterministic mode (det.h): one scratch tile-set per grid slice
  PFST_CHECK_ARG(det_ok);
  float* const dwk = ws ? ws : dw;
  const i64 gsk = ws ? -elems : dw_gs;
#define PFST_LAUNCH_WGRAD(KERNEL_, THREADS_)                                                                                               \
  hipLaunchKernelGGL(KERNEL_, dim3(gx * gy * gz), dim3(THREADS_), 0, s, x, x_bs, dy, dy_bs, dwk, J, M, P, chunks, chunk_len, N, x_gs, dy_gs, gsk, \
                     gx, gy, gz, x_amax, dy_amax, bnl4)
  if (bnl && big) PFST_LAUNCH_WGRAD((conv_wgrad_f16x3_line_kernel<false, 256, true>), 512);
  else if (bnl) PFST_LAUNCH_WGRAD((conv_wgrad_f16x3_line_kernel<false, 128, true>), 256);
  else if (big && packed) PFST_LAUNCH_WGRAD((conv_wgrad_f16x3_line_kernel<true, 256>), 512);
  else if (big) PFST_LAUNCH_WGRAD((conv_wgrad_f16x3_line_kernel<false, 256>), 512);
  else if (packed) PFST_LAUNCH_WGRAD((conv_wgrad_f16x3_line_kernel<true>), 256);
  else PFST_LAUNCH_WGRAD((conv_wgrad_f16x3_line_kernel<false>), 256);
#undef PFST_LAUNCH_WGRAD
  if (ws) wgrad_det_reduce(ws, dw, elems, groups, N * chunks, dw_gs, s);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

// dw[co][ci] += sum_{n,p} dy[n][co][p] x[n][ci][p]: the weight gradient of a stride-1 1x1 convolution (atomic fp32 adds)
extern "C" int pfst_conv_wgrad_f16x3(const float* x, long long x_bs, const float* dy, long long dy_bs, float* dw, int N, int Cin, int Cout,
                                     int HW, const float* x_amax, const float* dy_amax, const float* bnl, pfst_stream_t stream) {
  PFST_CHECK_ARG(x && dy && dw && x_amax && dy_amax && N > 0 && Cin > 0 && Cout > 0 && HW > 0);
  PFST_CHECK_ARG(x_bs >= (i64)Cin * HW && dy_bs >= (i64)Cout * HW && (x_bs & 3) == 0 && (dy_bs & 3) == 0);
  PFST_CHECK_ARG((((uintptr_t)x | (uintptr_t)dy) & 15) == 0 && (i64)Cin * HW * 4 < (1ll << 31) && (i64)Cout * HW * 4 < (1ll << 31));
  if (HW % 4 != 0 || Cout <= 64) {
    pfst_set_error(__FILE__, __LINE__, "f16x3 weight gradient needs HW % 4 == 0 and more than 64 output channels (use pfst_conv_wgrad_split)");
    return PFST_ERR_UNSUPPORTED;
  }
  return pfst_wgrad_f16x3_launch(x, x_bs, dy, dy_bs, dw, N, Cin, Cout, HW, 1, 0, 0, 0, x_amax, dy_amax, 0, (hipStream_t)stream, bnl);
}
