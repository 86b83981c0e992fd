// EMA teacher update and AdamW on FLAT parameter arenas: one launch each instead of the reference's
// ~850 tiny kernels (pfgst.py:105-127 loops over ~214 tensors) / torch.optim.AdamW's per-tensor loop.
// All student parameters live in one contiguous fp32 buffer (same for grads, Adam moments and the
// teacher), which is also what RCCL all-reduces.
#include "common.h"
#include "../../include/pfst_hip.h"

namespace {

__global__ void ema_kernel(float* __restrict__ t, const float* __restrict__ s, i64 n, float alpha) {
  const float one_m = 1.f - alpha;
  const i64 stride = (i64)gridDim.x * blockDim.x;
  const i64 n4 = n >> 2;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 a = reinterpret_cast<float4*>(t)[i];
    const float4 b = reinterpret_cast<const float4*>(s)[i];
    // reference arithmetic: alpha*ema + (1-alpha)*param, two roundings of the products then the sum
    a.x = alpha * a.x + one_m * b.x; a.y = alpha * a.y + one_m * b.y;
    a.z = alpha * a.z + one_m * b.z; a.w = alpha * a.w + one_m * b.w;
    reinterpret_cast<float4*>(t)[i] = a;
  }
  for (i64 i = (n4 << 2) + (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) t[i] = alpha * t[i] + one_m * s[i];
}

// torch.optim.AdamW (amsgrad=False, maximize=False), single-tensor formulation
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, i64 n,
                             float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt, float gscale) {
  const i64 stride = (i64)gridDim.x * blockDim.x;
  const float step_size = lr / bc1;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float gi = g[i] * gscale;
    float pi = p[i] * (1.f - lr * wd);
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= step_size * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}

}  // namespace

extern "C" int pfst_ema_update(float* teacher, const float* student, long long n, float alpha, pfst_stream_t stream) {
  PFST_CHECK_ARG(teacher && student && n > 0);
  PFST_CHECK_ARG(((((uintptr_t)teacher) | ((uintptr_t)student)) & 15) == 0);
  hipLaunchKernelGGL(ema_kernel, dim3(ew_grid(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, teacher, student, (i64)n, alpha);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

extern "C" int pfst_adamw_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                               float eps, float weight_decay, int step, float grad_scale, pfst_stream_t stream) {
  PFST_CHECK_ARG(p && g && m && v && n > 0 && step >= 1);
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adamw_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (i64)n, lr, beta1, beta2, eps,
                     weight_decay, (float)bc1, (float)sqrt(bc2), grad_scale);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}
