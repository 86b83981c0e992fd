// Winograd F(2x2, 3x3) for the wide stride-1 3x3 convolutions (fp32 throughout).
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A      per 2x2 output tile / 4x4 input patch, summed over input channels
// 16 multiplies instead of 36 per tile and channel pair: 2.25x fewer MACs on a path that is bound by the fp32 MFMA rate.
// The channel sum is 16 independent GEMMs  M_xi[co][t] = sum_ci U_xi[co][ci] V_xi[ci][t]  (xi = transform index, t = tile), run as
// ONE launch of the K-quad implicit-GEMM kernel (conv_igemm_q.hip, gridDim.y = 16, 1x1 mode) on transform-domain tensors
//   V [16][N][C][T]   U [16][K/4][M][4] (the kernel's packed weight layout)   Mbuf [16][N][Cout][T].
// Data gradient = the same pipeline on dY with the flipped / transposed filter.  Weight gradient:
//   dU_xi[co][ci] = sum_{n,t} dM_xi[n][co][t] V_xi[n][ci][t]   (one grouped launch of the 1x1 K-quad wgrad kernel),  dM = A dY A^T,
//   dW = G^T dU G.
// Dilation d: a dilated 3x3 convolution is d*d independent dilation-1 convolutions on the interleaved sub-grids
// (y mod d, x mod d); the tile index enumerates them (column offset fastest), so the same kernels serve d = 1, 2, 4.
// Transform matrices (Lavin & Gray 2016, correlation form):
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]   A^T = [1 1 1 0; 0 1 -1 -1]
#include "common.h"
#include "../../include/pfst_hip.h"

namespace {

struct WinoGeom {
  int H, W, d, Hs, Ws, Th, Tw, T;   // image, dilation, sub-grid size, tiles per sub-grid, tiles per image
};
__host__ __device__ inline WinoGeom wino_geom(int H, int W, int d) {
  WinoGeom g;
  g.H = H; g.W = W; g.d = d;
  g.Hs = (H + d - 1) / d; g.Ws = (W + d - 1) / d;
  g.Th = (g.Hs + 1) / 2; g.Tw = (g.Ws + 1) / 2;
  g.T = d * d * g.Th * g.Tw;
  return g;
}
// tile index -> (sub-grid offsets sy, sx; tile coordinates ty, tx).  The sub-grid column offset sx runs fastest: consecutive
// threads then touch x = sx + d*(2tx + const), i.e. runs of d contiguous pixels, instead of pixels 2d apart.
__device__ __forceinline__ void tile_coord(const WinoGeom& g, int t, int& sy, int& sx, int& ty, int& tx) {
  sx = t % g.d; t /= g.d;
  tx = t % g.Tw; t /= g.Tw;
  ty = t % g.Th;
  sy = t / g.Th;
}

// ---- filter transform.  One thread per (co, ci): U = G g G^T (4x4), written into the K-quad packed layouts
//   Uf[xi][(ci>>2)*Cout + co][ci&3]            (fprop:  K = Cin,  rows = Cout)
//   Ud[xi][(co>>2)*Cin  + ci][co&3]  with g flipped (dgrad: K = Cout, rows = Cin)
__device__ __forceinline__ void filter_tf(const float (&g)[3][3], float (&u)[4][4]) {
  float t[4][3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    t[0][j] = g[0][j];
    t[1][j] = 0.5f * (g[0][j] + g[1][j] + g[2][j]);
    t[2][j] = 0.5f * (g[0][j] - g[1][j] + g[2][j]);
    t[3][j] = g[2][j];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    u[i][0] = t[i][0];
    u[i][1] = 0.5f * (t[i][0] + t[i][1] + t[i][2]);
    u[i][2] = 0.5f * (t[i][0] - t[i][1] + t[i][2]);
    u[i][3] = t[i][2];
  }
}
__global__ void wino_filter_kernel(const float* __restrict__ w, float* __restrict__ Uf, float* __restrict__ Ud, int Cout, int Cin) {
  const i64 total = (i64)Cout * Cin;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Cin), co = (int)(i / Cin);
    float g[3][3], gf[3][3], u[4][4];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        g[a][b] = w[i * 9 + a * 3 + b];
        gf[2 - a][2 - b] = g[a][b];
      }
    if (Uf) {
      filter_tf(g, u);
#pragma unroll
      for (int xi = 0; xi < 16; ++xi) Uf[(i64)xi * total + ((i64)(ci >> 2) * Cout + co) * 4 + (ci & 3)] = u[xi >> 2][xi & 3];
    }
    if (Ud) {
      filter_tf(gf, u);
#pragma unroll
      for (int xi = 0; xi < 16; ++xi) Ud[(i64)xi * total + ((i64)(co >> 2) * Cin + ci) * 4 + (co & 3)] = u[xi >> 2][xi & 3];
    }
  }
}

// plain [16][Cout][Cin] filter sets (normal and flipped), input of the bf16x6 packer (pfst_wino_pack_weight_split)
__global__ void wino_filter_plain_kernel(const float* __restrict__ w, float* __restrict__ Pf, float* __restrict__ Pd, int Cout, int Cin) {
  const i64 total = (i64)Cout * Cin;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
    float g[3][3], gf[3][3], u[4][4];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        g[a][b] = w[i * 9 + a * 3 + b];
        gf[2 - a][2 - b] = g[a][b];
      }
    if (Pf) {
      filter_tf(g, u);
#pragma unroll
      for (int xi = 0; xi < 16; ++xi) Pf[(i64)xi * total + i] = u[xi >> 2][xi & 3];
    }
    if (Pd) {
      filter_tf(gf, u);
#pragma unroll
      for (int xi = 0; xi < 16; ++xi) Pd[(i64)xi * total + i] = u[xi >> 2][xi & 3];
    }
  }
}

// ---- input transform V = B^T d B.   grid: (blocks over T, C, N); one thread per tile
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, i64 x_bs, float* __restrict__ V, int N, int C,
                                                         WinoGeom g) {
  const int c = blockIdx.y, n = blockIdx.z;
  const float* xp = x + (i64)n * x_bs + (i64)c * g.H * g.W;
  const i64 plane = (i64)N * C * g.T;                        // stride between transform indices
  float* vp = V + ((i64)n * C + c) * g.T;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < g.T; t += gridDim.x * blockDim.x) {
    int sy, sx, ty, tx;
    tile_coord(g, t, sy, sx, ty, tx);
    float d[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int y = sy + g.d * (2 * ty - 1 + a);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int xx = sx + g.d * (2 * tx - 1 + b);
        d[a][b] = (y >= 0 && y < g.H && xx >= 0 && xx < g.W) ? xp[(i64)y * g.W + xx] : 0.f;
      }
    }
    float r[4][4];                                           // B^T d
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      r[0][b] = d[0][b] - d[2][b];
      r[1][b] = d[1][b] + d[2][b];
      r[2][b] = d[2][b] - d[1][b];
      r[3][b] = d[1][b] - d[3][b];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {                            // (B^T d) B
      vp[(i64)(a * 4 + 0) * plane + t] = r[a][0] - r[a][2];
      vp[(i64)(a * 4 + 1) * plane + t] = r[a][1] + r[a][2];
      vp[(i64)(a * 4 + 2) * plane + t] = r[a][2] - r[a][1];
      vp[(i64)(a * 4 + 3) * plane + t] = r[a][1] - r[a][3];
    }
  }
}

// ---- output transform Y = A^T m A (2x2 per tile), optional accumulate.   grid: (blocks over T, Cout, N)
__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ Mb, float* __restrict__ y, i64 y_bs, int N, int Cout,
                                                          WinoGeom g, int accumulate, float* __restrict__ stats) {
  __shared__ double red[16];
  float st_s = 0.f, st_q = 0.f;                    // fused BatchNorm statistics of the outputs this block writes (stats != NULL)
  const int c = blockIdx.y, n = blockIdx.z;
  float* yp = y + (i64)n * y_bs + (i64)c * g.H * g.W;
  const i64 plane = (i64)N * Cout * g.T;
  const float* mp = Mb + ((i64)n * Cout + c) * g.T;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < g.T; t += gridDim.x * blockDim.x) {
    int sy, sx, ty, tx;
    tile_coord(g, t, sy, sx, ty, tx);
    float m[4][4];
#pragma unroll
    for (int i = 0; i < 16; ++i) m[i >> 2][i & 3] = mp[(i64)i * plane + t];
    float r[2][4];                                           // A^T m
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      r[0][b] = m[0][b] + m[1][b] + m[2][b];
      r[1][b] = m[1][b] - m[2][b] - m[3][b];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int yy = sy + g.d * (2 * ty + i);
      if (yy >= g.H) continue;
      const float o0 = r[i][0] + r[i][1] + r[i][2], o1 = r[i][1] - r[i][2] - r[i][3];
      const int x0 = sx + g.d * (2 * tx), x1 = x0 + g.d;
      if (g.d == 1 && x1 < g.W && (g.W & 1) == 0) {           // the two outputs are adjacent: one 8-byte store
        float2* q = reinterpret_cast<float2*>(yp + (i64)yy * g.W + x0);
        float2 v = make_float2(o0, o1);
        if (accumulate) { const float2 old = *q; v.x += old.x; v.y += old.y; }
        *q = v;
        st_s += v.x + v.y;
        st_q = fmaf(v.x, v.x, fmaf(v.y, v.y, st_q));
        continue;
      }
      if (x0 < g.W) { float* q = yp + (i64)yy * g.W + x0; const float v = accumulate ? *q + o0 : o0; *q = v; st_s += v; st_q = fmaf(v, v, st_q); }
      if (x1 < g.W) { float* q = yp + (i64)yy * g.W + x1; const float v = accumulate ? *q + o1 : o1; *q = v; st_s += v; st_q = fmaf(v, v, st_q); }
    }
  }
  if (stats) {                                       // stats[c][n * gridDim.x + blockIdx.x][2] for pfst_bn_finalize_partials
    const double bs = block_sum_d((double)st_s, red);
    const double bq = block_sum_d((double)st_q, red);
    if (threadIdx.x == 0) {
      const i64 T = (i64)gridDim.x * gridDim.z;
      float2* dst = reinterpret_cast<float2*>(stats) + ((i64)c * T + (i64)n * gridDim.x + blockIdx.x);
      *dst = make_float2((float)bs, (float)bq);
    }
  }
}

// ---- adjoint of the output transform: dM = A dY A^T (4x4 per tile) for the weight gradient.   grid: (blocks over T, Cout, N)
__global__ __launch_bounds__(256) void wino_dy_kernel(const float* __restrict__ dy, i64 dy_bs, float* __restrict__ dM, int N, int Cout,
                                                      WinoGeom g) {
  const int c = blockIdx.y, n = blockIdx.z;
  const float* gp = dy + (i64)n * dy_bs + (i64)c * g.H * g.W;
  const i64 plane = (i64)N * Cout * g.T;
  float* mp = dM + ((i64)n * Cout + c) * g.T;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < g.T; t += gridDim.x * blockDim.x) {
    int sy, sx, ty, tx;
    tile_coord(g, t, sy, sx, ty, tx);
    float e[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int yy = sy + g.d * (2 * ty + i), xx = sx + g.d * (2 * tx + j);
        e[i][j] = (yy < g.H && xx < g.W) ? gp[(i64)yy * g.W + xx] : 0.f;
      }
    float r[4][2];                                           // A e   (A = [1 0; 1 1; 1 -1; 0 -1])
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      r[0][j] = e[0][j];
      r[1][j] = e[0][j] + e[1][j];
      r[2][j] = e[0][j] - e[1][j];
      r[3][j] = -e[1][j];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {                            // (A e) A^T
      mp[(i64)(a * 4 + 0) * plane + t] = r[a][0];
      mp[(i64)(a * 4 + 1) * plane + t] = r[a][0] + r[a][1];
      mp[(i64)(a * 4 + 2) * plane + t] = r[a][0] - r[a][1];
      mp[(i64)(a * 4 + 3) * plane + t] = -r[a][1];
    }
  }
}

// ---- dW += G^T dU G.   dU [16][Cout][Cin] (row-major as the 1x1 wgrad kernel writes it).  One thread per (co, ci)
__global__ void wino_dw_kernel(const float* __restrict__ dU, float* __restrict__ dw, int Cout, int Cin) {
  const i64 total = (i64)Cout * Cin;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
    float u[4][4];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) u[xi >> 2][xi & 3] = dU[(i64)xi * total + i];
    float t[3][4];                                           // G^T u
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      t[0][b] = u[0][b] + 0.5f * (u[1][b] + u[2][b]);
      t[1][b] = 0.5f * (u[1][b] - u[2][b]);
      t[2][b] = 0.5f * (u[1][b] + u[2][b]) + u[3][b];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {                            // (G^T u) G
      dw[i * 9 + a * 3 + 0] += t[a][0] + 0.5f * (t[a][1] + t[a][2]);
      dw[i * 9 + a * 3 + 1] += 0.5f * (t[a][1] - t[a][2]);
      dw[i * 9 + a * 3 + 2] += 0.5f * (t[a][1] + t[a][2]) + t[a][3];
    }
  }
}

inline int tile_blocks(int T) {
  int b = (T + 255) / 256;
  return b < 1 ? 1 : b;
}

}  // namespace

extern "C" int pfst_wino_tiles(int H, int W, int dil) {
  if (H <= 0 || W <= 0 || dil < 1) return 0;
  return wino_geom(H, W, dil).T;
}

extern "C" int pfst_wino_pack_weight(const float* w, float* U_fprop, float* U_dgrad, int Cout, int Cin, pfst_stream_t stream) {
  PFST_CHECK_ARG(w && (U_fprop || U_dgrad) && Cout > 0 && Cin > 0);
  PFST_CHECK_ARG((!U_fprop || Cin % 16 == 0) && (!U_dgrad || Cout % 16 == 0));     // K-quad GEMM kernel: K % 16 == 0
  hipLaunchKernelGGL(wino_filter_kernel, dim3(ew_grid((i64)Cout * Cin)), dim3(256), 0, (hipStream_t)stream, w, U_fprop, U_dgrad, Cout, Cin);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

extern "C" int pfst_wino_filter_plain(const float* w, float* P_fprop, float* P_dgrad, int Cout, int Cin, pfst_stream_t stream) {
  PFST_CHECK_ARG(w && (P_fprop || P_dgrad) && Cout > 0 && Cin > 0);
  hipLaunchKernelGGL(wino_filter_plain_kernel, dim3(ew_grid((i64)Cout * Cin)), dim3(256), 0, (hipStream_t)stream, w, P_fprop, P_dgrad, Cout, Cin);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

extern "C" int pfst_wino_input(const float* x, long long x_bs, float* V, int N, int C, int H, int W, int dil, pfst_stream_t stream) {
  PFST_CHECK_ARG(x && V && N > 0 && N <= 65535 && C > 0 && C <= 65535 && H > 0 && W > 0 && dil >= 1 && x_bs >= (i64)C * H * W);
  const WinoGeom g = wino_geom(H, W, dil);
  hipLaunchKernelGGL(wino_input_kernel, dim3(tile_blocks(g.T), C, N), dim3(256), 0, (hipStream_t)stream, x, x_bs, V, N, C, g);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

// the 16 transform-domain GEMMs: Mbuf[xi][n][M][T] = U[xi] (M x K) * V[xi][n] (K x T)
extern "C" int pfst_wino_gemm(const float* V, const float* U, float* Mbuf, int N, int K, int M, int T, pfst_stream_t stream) {
  PFST_CHECK_ARG(V && U && Mbuf && N > 0 && N <= 65535 && K > 0 && K % 16 == 0 && M > 0 && T > 0);
  PFST_CHECK_ARG((i64)K * T * 4 < (1ll << 31) && (i64)M * T * 4 < (1ll << 31) && (i64)K * M * 4 < (1ll << 31));
  return pfst_igemm_q_launch(V, (i64)K * T, U, nullptr, Mbuf, (i64)M * T, N, K, 1, T, M, 1, T, 1, 1, 1, 0, 1, 0, nullptr, 0, 16,
                             (hipStream_t)stream);
}

extern "C" int pfst_wino_stats_slots(int H, int W, int dil) {
  if (H <= 0 || W <= 0 || dil < 1) return 0;
  return tile_blocks(wino_geom(H, W, dil).T);
}

extern "C" int pfst_wino_output(const float* Mbuf, float* y, long long y_bs, int N, int Cout, int H, int W, int dil, int accumulate,
                                float* stats, pfst_stream_t stream) {
  PFST_CHECK_ARG(Mbuf && y && N > 0 && N <= 65535 && Cout > 0 && Cout <= 65535 && H > 0 && W > 0 && dil >= 1 && y_bs >= (i64)Cout * H * W);
  const WinoGeom g = wino_geom(H, W, dil);
  hipLaunchKernelGGL(wino_output_kernel, dim3(tile_blocks(g.T), Cout, N), dim3(256), 0, (hipStream_t)stream, Mbuf, y, y_bs, N, Cout, g,
                     accumulate, stats);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

extern "C" int pfst_wino_dy(const float* dy, long long dy_bs, float* dM, int N, int Cout, int H, int W, int dil, pfst_stream_t stream) {
  PFST_CHECK_ARG(dy && dM && N > 0 && N <= 65535 && Cout > 0 && Cout <= 65535 && H > 0 && W > 0 && dil >= 1 && dy_bs >= (i64)Cout * H * W);
  const WinoGeom g = wino_geom(H, W, dil);
  hipLaunchKernelGGL(wino_dy_kernel, dim3(tile_blocks(g.T), Cout, N), dim3(256), 0, (hipStream_t)stream, dy, dy_bs, dM, N, Cout, g);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

// dU[xi][Cout][Cin] = sum_{n,t} dM[xi][n][Cout][T] V[xi][n][Cin][T]  (one grouped launch of the 1x1 K-quad wgrad), then dW += G^T dU G.
// dU is scratch of 16*Cout*Cin floats (zeroed here).
extern "C" int pfst_wino_wgrad(const float* V, const float* dM, float* dU, float* dw, int N, int Cin, int Cout, int T, pfst_stream_t stream) {
  PFST_CHECK_ARG(V && dM && dU && dw && N > 0 && N <= 65535 && Cin > 0 && Cout > 0 && T > 0 && T % 4 == 0);
  hipStream_t s = (hipStream_t)stream;
  const i64 uc = (i64)Cout * Cin;
  if (hipMemsetAsync(dU, 0, 16 * uc * sizeof(float), s) != hipSuccess) return PFST_ERR_LAUNCH;
  // the 16 per-transform-index products as ONE grouped launch: enough workgroups without splitting the tile range further
  const int rc = pfst_wgrad_q_launch(V, (i64)Cin * T, dM, (i64)Cout * T, dU, N, Cin, 1, T, Cout, 1, T, 1, 1, 0, 16, (i64)N * Cin * T,
                                     (i64)N * Cout * T, uc, s);
  if (rc != PFST_OK) return rc;
  hipLaunchKernelGGL(wino_dw_kernel, dim3(ew_grid(uc)), dim3(256), 0, s, dU, dw, Cout, Cin);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}
