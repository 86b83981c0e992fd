// Does the buffer range check of gfx950 include the scalar offset?  The GEMM kernels address "past the end" pairs through soffset and rely
// on zeros / no access there.  Build: hipcc --offload-arch=gfx950 -O2 tools/probes/soffset_range_probe.hip -o gpurun_out/soffset_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(const float* buf, float* out, int soff) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(buf), 0, 64, 0x00020000);   // 16 floats in range
  out[threadIdx.x] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, 4u * threadIdx.x, soff, 0));
}
int main() {
  float h[64], *d, *o;
  for (int i = 0; i < 64; ++i) h[i] = 100.f + i;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, 16 * 4);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  for (int soff : {0, 32, 64, 128}) {
    probe<<<1, 16>>>(d, o, soff);
    float r[16];
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    printf("soffset %3d:", soff);
    for (int i = 0; i < 16; ++i) printf(" %g", r[i]);
    printf("\n");
  }
  return 0;
}
