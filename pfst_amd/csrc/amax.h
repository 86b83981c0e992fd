// Absolute-maximum slots of the f16x3 arithmetic (conv_f16x3.hip): a tensor's scale comes from max |x|, kept on the device as a GROUP
// of PFST_AMAX_SUB fp32 sub-slots (bit patterns of non-negative floats, so an unsigned atomicMax orders them).  Producers (BatchNorm
// apply / backward, the Winograd transforms, pfst_absmax) call amax_publish once per workgroup: the maxima of the waves meet in LDS and
// ONE atomic per workgroup goes to the sub-slot its linear id selects, so the 16-64 K workgroups of a producing launch put a few dozen
// atomics on each address (one group of 64 sub-slots, one atomic per wave, serialised ~4000 same-address atomics per launch and
// tripled the BatchNorm kernels' time).  Consumers take the maximum of the group with amax_read (4 KB per wave, L2 hits).
#pragma once
#include "common.h"

constexpr int PFST_AMAX_SUB = 1024;

// m: this thread's maximum.  Every thread of the workgroup must call it (it holds a barrier).
__device__ __forceinline__ void amax_publish(float* __restrict__ group, float m) {
  __shared__ float amax_red[16];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) amax_red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int w = 1; w < nw; ++w) m = fmaxf(m, amax_red[w]);
    if (m > 0.f) {
      const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
      atomicMax(reinterpret_cast<unsigned*>(group) + (lin & (PFST_AMAX_SUB - 1)), __builtin_bit_cast(unsigned, m));
    }
  }
}

// wave-uniform maximum of a group
__device__ __forceinline__ float amax_read(const float* __restrict__ group) {
  const float4* g4 = reinterpret_cast<const float4*>(group);
  float m = 0.f;
#pragma unroll
  for (int i = 0; i < PFST_AMAX_SUB / 256; ++i) {
    const float4 v = g4[(threadIdx.x & 63) + 64 * i];
    m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, m)));
}

// biased exponent of an absolute maximum, clamped so that both the scale 2^(268 - E - 127) and its inverse are normal floats
__device__ __forceinline__ int amax_exponent(float amax) {
  int e = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 0xffu);
  return e < 16 ? 16 : (e > 254 ? 254 : e);            // zero / denormal maxima: any scale will do (the tensor is ~0)
}
__device__ __forceinline__ float scale_of(int e) { return __builtin_bit_cast(float, (unsigned)(268 - e) << 23); }       // 2^(14 - (e - 127))
__device__ __forceinline__ float unscale_of(int e) { return __builtin_bit_cast(float, (unsigned)(e - 14) << 23); }      // 2^((e - 127) - 14)

// x s as two fp16 pieces h + l in one dword (h in the low half): the storage format of a pre-split GEMM operand
__device__ __forceinline__ unsigned pack_f16x2_pieces(float xs) {
  const _Float16 h = (_Float16)xs;                       // round to nearest even
  const _Float16 l = (_Float16)(xs - (float)h);          // the remainder is exact in fp32
  return (unsigned)__builtin_bit_cast(unsigned short, h) | (unsigned)__builtin_bit_cast(unsigned short, l) << 16;
}
