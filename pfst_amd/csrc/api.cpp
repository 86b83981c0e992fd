// Process-wide error string + ABI version of libpfst_hip.so.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include "../../include/pfst_hip.h"

static char g_err[512] = "no error";

void pfst_set_error(const char* file, int line, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s:%d: %s", file, line, msg);
}

// Deterministic mode (pfst_set_deterministic): every sum that is normally completed by fp32 / fp64 atomic adds of several workgroups -- the
// split-K slices of the weight gradients, the BatchNorm-backward reductions, the depthwise and bias gradients, PFGSTLoss's source statistics --
// goes through per-workgroup (per grid slice) partials in a scratch and is added up by a second kernel in index order (det.h, bn.hip,
// dwconv.hip): same launch shapes, no sum depends on which workgroup finishes first.  +3 % on the b = 8 x 1024^2 step; the gradient of a step is
// bit-identical run to run and for any stream schedule (tests/test_deterministic_gpu.py, tools/det_repro_fullsize.py).
static int g_deterministic = 0;
int pfst_deterministic(void) { return g_deterministic; }
extern "C" int pfst_set_deterministic(int on) {
  g_deterministic = on != 0;
  return 0;
}
extern "C" int pfst_get_deterministic(void) { return g_deterministic; }

// Scratch of the deterministic mode's partial sums (the only memory the library allocates itself; nothing outside that mode touches it): one
// grow-only buffer per stream -- a launcher fills it, a second kernel of the same launcher reduces it, both queued on that stream.
void* pfst_det_scratch(size_t bytes, void* stream) {
  static struct { void* stream; void* buf; size_t cap; } slots[8];
  int at = -1;
  for (int i = 0; i < 8; ++i)
    if (slots[i].buf && slots[i].stream == stream) at = i;
  if (at < 0)
    for (int i = 0; i < 8 && at < 0; ++i)
      if (!slots[i].buf) at = i;
  if (at < 0) return nullptr;
  if (slots[at].cap < bytes) {
    if (slots[at].buf) {
      if (hipDeviceSynchronize() != hipSuccess) return nullptr;       // earlier launches may still read the old buffer
      (void)hipFree(slots[at].buf);
      slots[at].buf = nullptr;
      slots[at].cap = 0;
    }
    const size_t cap = bytes < (size_t(8) << 20) ? (size_t(8) << 20) : bytes + bytes / 2;
    if (hipMalloc(&slots[at].buf, cap) != hipSuccess) return nullptr;
    slots[at].cap = cap;
    slots[at].stream = stream;
  }
  return slots[at].buf;
}

extern "C" const char* pfst_last_error(void) { return g_err; }
extern "C" int pfst_abi_version(void) { return 1; }
