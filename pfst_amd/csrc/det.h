// Deterministic mode (api.cpp: pfst_set_deterministic) of the split-K weight-gradient launchers.  A launch splits the contraction over images and
// pixel chunks (grid slices bz = (group * N + image) * chunks + chunk) and normally every slice adds its partial tile into dW with fp32 atomics.  In
// deterministic mode the launcher hands the kernel a zeroed scratch with ONE tile-set per slice instead (dw_gs < 0: `dw += bz * -dw_gs`) -- every
// scratch element then has a single writer -- and wgrad_det_reduce_kernel adds a group's N * chunks slices into dW in index order.  Same launch
// shape and parallelism as the default mode; costs the scratch traffic (one write + one read of slices x |dW|).
#pragma once
#include "common.h"

namespace {

// grid: (ceil(elems / 256), groups); S slices per group
__global__ __launch_bounds__(256) void wgrad_det_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, i64 elems, int S, i64 dw_gs) {
  const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
  if (i >= elems) return;
  const int grp = blockIdx.y;
  float v = 0.f;
  for (int k = 0; k < S; ++k) v += ws[((i64)grp * S + k) * elems + i];
  dw[(i64)grp * dw_gs + i] += v;
}

// the zeroed scratch of a launch with `slices` grid slices (NULL: not in deterministic mode; ok = false: the scratch could not be had)
inline float* wgrad_det_scratch(i64 elems, i64 slices, hipStream_t s, bool& ok) {
  ok = true;
  if (!pfst_deterministic()) return nullptr;
  const size_t bytes = sizeof(float) * (size_t)elems * (size_t)slices;
  float* ws = static_cast<float*>(pfst_det_scratch(bytes, s));
  ok = ws != nullptr && hipMemsetAsync(ws, 0, bytes, s) == hipSuccess;
  return ok ? ws : nullptr;
}

inline void wgrad_det_reduce(const float* ws, float* dw, i64 elems, int groups, int slices_per_group, i64 dw_gs, hipStream_t s) {
  hipLaunchKernelGGL(wgrad_det_reduce_kernel, dim3((unsigned)((elems + 255) / 256), groups), dim3(256), 0, s, ws, dw, elems, slices_per_group, dw_gs);
}

}  // namespace
