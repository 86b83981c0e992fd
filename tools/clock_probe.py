"""Diagnostic: shader clock held by the chip while the fp32-MFMA implicit-GEMM kernel runs (s_memtime / s_memrealtime
stamps, MI355X_MICROARCH.md 'DVFS give-back' item 6).  Builds a SEPARATE library with -DPFST_CLOCK_STAMPS; the product
library never contains the stamps."""
import ctypes, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
src = os.path.join(ROOT, 'pfst_amd', 'csrc')
out = '/tmp/libpfst_clock.so'
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-DPFST_CLOCK_STAMPS',
                       '-Wno-unused-value', '-Wno-unused-result', os.path.join(src, 'conv_mfma.hip'), '-x', 'hip', os.path.join(src, 'api.cpp'), '-o', out])
L = ctypes.CDLL(out)
B, C, M, H = 8, 2048, 512, 128
x = torch.randn(B, C, H, H, device='cuda'); w = torch.randn(M, C, 1, 1, device='cuda') * 0.05
wf = torch.empty(C, M, device='cuda'); y = torch.empty(B, M, H, H, device='cuda')
vp = ctypes.c_void_p
L.pfst_conv_pack_weight(vp(w.data_ptr()), vp(wf.data_ptr()), None, M, C, 1, None)
def run():
    L.pfst_conv_igemm(vp(x.data_ptr()), ctypes.c_longlong(C * H * H), vp(wf.data_ptr()), None, vp(y.data_ptr()), ctypes.c_longlong(M * H * H),
                      B, C, H, H, M, H, H, 1, 1, 1, 0, 0, 0, None, None)
t0 = time.time()
while time.time() - t0 < 3.0:       # >= 2 s of back-to-back launches on random data before reading the stamps
    for _ in range(50): run()
    torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); run(); e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e)
n = 4096
buf = (ctypes.c_ulonglong * (2 * n))()
assert L.pfst_debug_read_stamps(buf, n) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 2).astype(np.float64)
a = a[a[:, 1] > 0]
clk = a[:, 0] / a[:, 1] * 100e6        # memrealtime ticks at 100 MHz
print(f'kernel {ms:.3f} ms = {2*B*M*C*H*H/ms/1e9:.1f} TFLOP/s; blocks sampled {len(a)}; shader clock median {np.median(clk)/1e9:.3f} GHz '
      f'(p10 {np.percentile(clk,10)/1e9:.3f}, p90 {np.percentile(clk,90)/1e9:.3f}); block duration median {np.median(a[:,1])/100:.1f} us')
print(f'clock-adjusted fp32 MFMA peak = {157.3*np.median(clk)/2.4e9:.1f} TFLOP/s')
