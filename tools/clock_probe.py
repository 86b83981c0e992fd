"""Diagnostic: where a workgroup of the K-quad implicit-GEMM kernel spends its life, and the shader clock the chip holds
(s_memtime / s_memrealtime stamps, MI355X_MICROARCH.md 'DVFS give-back' item 6).  Builds a SEPARATE library with
-DPFST_CLOCK_STAMPS; the product library never contains the stamps.   usage: python3 tools/clock_probe.py"""
import ctypes, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
src = os.path.join(ROOT, 'pfst_amd', 'csrc')
out = '/tmp/libpfst_clock.so'
files = ['conv_mfma.hip', 'conv_igemm_q.hip', 'conv_wgrad_q.hip', 'conv_winograd.hip', 'conv_split.hip']
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-DPFST_CLOCK_STAMPS',
                       '-Wno-unused-value', '-Wno-unused-result'] + [os.path.join(src, f) for f in files] +
                      ['-x', 'hip', os.path.join(src, 'api.cpp'), '-o', out])
L = ctypes.CDLL(out)
vp = ctypes.c_void_p
B, H = 8, 128
for C, M, with_stats in [(2048, 512, 0), (512, 2048, 0), (512, 2048, 1), (256, 1024, 0), (256, 1024, 1), (512, 512, 0)]:
    x = torch.randn(B, C, H, H, device='cuda'); w = torch.randn(M, C, 1, 1, device='cuda') * 0.05
    wf = torch.empty(C * M, device='cuda'); y = torch.empty(B, M, H, H, device='cuda')
    L.pfst_conv_pack_weight(vp(w.data_ptr()), vp(wf.data_ptr()), None, M, C, 1, None)
    st = torch.empty(2 * M * B * L.pfst_conv_stats_slots(M, H, H), device='cuda') if with_stats else None
    def run():
        L.pfst_conv_igemm(vp(x.data_ptr()), ctypes.c_longlong(C * H * H), vp(wf.data_ptr()), None, vp(y.data_ptr()), ctypes.c_longlong(M * H * H),
                          B, C, H, H, M, H, H, 1, 1, 1, 0, 0, 0, vp(st.data_ptr()) if with_stats else None, None, None)
    t0 = time.time()
    while time.time() - t0 < 1.0:           # sustained load on random data before reading the stamps
        for _ in range(20): run()
        torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); run(); e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e)
    nblk = (H * H // 128) * ((M + 127) // 128) * B
    n = min(nblk, 65536)
    buf = (ctypes.c_ulonglong * (5 * n))()
    assert L.pfst_debug_read_stamps(buf, n) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 5).astype(np.float64)
    a = a[a[:, 3] > 0]
    cyc = a[:, 0] + a[:, 1] + a[:, 2]
    clk = cyc / a[:, 3] * 100e6           # s_memrealtime ticks at 100 MHz
    span = (a[:, 4].max() - a[:, 4].min() + a[:, 3].max()) / 100.0   # us from first start to last end (approx.)
    kt = C // 16
    print(f'C={C} M={M} stats={with_stats}: kernel {ms:.3f} ms = {2*B*M*C*H*H/ms/1e9:.1f} TF/s; {nblk} tiles of {kt} K-steps | per workgroup (median cycles): '
          f'prologue {np.median(a[:,0]):.0f}, loop {np.median(a[:,1]):.0f} ({np.median(a[:,1])/kt:.0f}/step), epilogue {np.median(a[:,2]):.0f} | '
          f'lifetime {np.median(a[:,3])/100:.1f} us, clock {np.median(clk)/1e9:.2f} GHz | starts span {span:.0f} us', flush=True)
