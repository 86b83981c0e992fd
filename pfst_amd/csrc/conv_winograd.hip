// Winograd F(m x m, 3x3), m = 2 or 4, for the wide stride-1 3x3 convolutions (fp32 throughout).
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A      per m x m output tile / (m+2) x (m+2) input patch, summed over input channels
// R*R multiplies (R = m + 2) instead of 9 m*m per tile and channel pair: 2.25x (m = 2) or 4x (m = 4) fewer MACs on a path that is
// bound by the fp32 MFMA rate; the transform-domain tensors shrink as well (R*R/m*m = 4x resp. 2.25x the plain tensor).
// The channel sum is R*R independent GEMMs  M_xi[co][t] = sum_ci U_xi[co][ci] V_xi[ci][t]  (xi = transform index, t = tile), run as
// ONE launch of the K-quad implicit-GEMM kernel (conv_igemm_q.hip, gridDim.y = R*R, 1x1 mode) on transform-domain tensors
//   V [R*R][N][C][T]   U [R*R][K/4][M][4] (the kernel's packed weight layout)   Mbuf [R*R][N][Cout][T].
// Data gradient = the same pipeline on dY with the flipped / transposed filter.  Weight gradient:
//   dU_xi[co][ci] = sum_{n,t} dM_xi[n][co][t] V_xi[n][ci][t]   (one grouped launch of the 1x1 K-quad wgrad kernel),  dM = A dY A^T,
//   dW = G^T dU G.
// Dilation d: a dilated 3x3 convolution is d*d independent dilation-1 convolutions on the interleaved sub-grids
// (y mod d, x mod d); the tile index enumerates them (column offset fastest), so the same kernels serve d = 1, 2, 4.
// Transform matrices (Lavin & Gray 2016, correlation form; interpolation points 0, +-1 [, +-2], inf):
//   m = 2:  B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]   A^T = [1 1 1 0; 0 1 -1 -1]
//   m = 4:  B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//           G = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
//           A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
// fp32 error against an fp64 direct convolution (512 channels): 3e-6 (m = 2), 3e-5 max / 5e-6 rms (m = 4) of the mean |y|.
#include "common.h"
#include "amax.h"
#include "weight_jobs.h"

namespace {

// constant transform matrices; every use below has compile-time indices after unrolling, so zeros vanish and +-1 become adds
template <int M> struct Wino;
template <> struct Wino<2> {
  static constexpr int R = 4;
  static constexpr float bt(int i, int j) {
    constexpr float m[4][4] = {{1, 0, -1, 0}, {0, 1, 1, 0}, {0, -1, 1, 0}, {0, 1, 0, -1}};
    return m[i][j];
  }
  static constexpr float g(int i, int j) {
    constexpr float m[4][3] = {{1, 0, 0}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0, 0, 1}};
    return m[i][j];
  }
  static constexpr float at(int i, int j) {
    constexpr float m[2][4] = {{1, 1, 1, 0}, {0, 1, -1, -1}};
    return m[i][j];
  }
};
template <> struct Wino<4> {
  static constexpr int R = 6;
  static constexpr float bt(int i, int j) {
    constexpr float m[6][6] = {{4, 0, -5, 0, 1, 0}, {0, -4, -4, 1, 1, 0}, {0, 4, -4, -1, 1, 0},
                               {0, -2, -1, 2, 1, 0}, {0, 2, -1, -2, 1, 0}, {0, 4, 0, -5, 0, 1}};
    return m[i][j];
  }
  static constexpr float g(int i, int j) {
    constexpr float m[6][3] = {{1.f / 4, 0, 0}, {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                               {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0, 0, 1}};
    return m[i][j];
  }
  static constexpr float at(int i, int j) {
    constexpr float m[4][6] = {{1, 1, 1, 1, 1, 0}, {0, 1, -1, 2, -2, 0}, {0, 1, 1, 4, 4, 0}, {0, 1, -1, 8, -8, 1}};
    return m[i][j];
  }
};
// acc += coef * v with the trivial coefficients folded
__device__ __forceinline__ float cmac(float acc, float coef, float v) {
  if (coef == 0.f) return acc;
  if (coef == 1.f) return acc + v;
  if (coef == -1.f) return acc - v;
  return fmaf(coef, v, acc);
}

struct WinoGeom {
  int H, W, d, Hs, Ws, Th, Tw, T;   // image, dilation, sub-grid size, tiles per sub-grid, tiles per image
};
inline WinoGeom wino_geom(int H, int W, int d, int m) {
  WinoGeom g;
  g.H = H; g.W = W; g.d = d;
  g.Hs = (H + d - 1) / d; g.Ws = (W + d - 1) / d;
  g.Th = (g.Hs + m - 1) / m; g.Tw = (g.Ws + m - 1) / m;
  g.T = d * d * g.Th * g.Tw;
  return g;
}
// tile index -> (sub-grid offsets sy, sx; tile coordinates ty, tx).  The sub-grid column offset sx runs fastest: consecutive
// threads then touch x = sx + d*(m*tx + const), i.e. runs of d contiguous pixels, instead of pixels m*d apart.
__device__ __forceinline__ void tile_coord(const WinoGeom& g, int t, int& sy, int& sx, int& ty, int& tx) {
  sx = t % g.d; t /= g.d;
  tx = t % g.Tw; t /= g.Tw;
  ty = t % g.Th;
  sy = t / g.Th;
}

// ---- filter transform U = G g G^T (R x R).  One thread per (co, ci); outputs in the K-quad packed layouts
//   Uf[xi][(ci>>2)*Cout + co][ci&3]            (fprop:  K = Cin,  rows = Cout)
//   Ud[xi][(co>>2)*Cin  + ci][co&3]  with g flipped (dgrad: K = Cout, rows = Cin)
// or (PLAIN) as [xi][Cout][Cin] sets, the input of the bf16x6 packer (pfst_wino_pack_weight_split).
template <int M>
__device__ __forceinline__ void filter_tf(const float (&g)[3][3], float (&u)[M + 2][M + 2]) {
  constexpr int R = M + 2;
  float t[R][3];
#pragma unroll
  for (int i = 0; i < R; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) a = cmac(a, Wino<M>::g(i, k), g[k][j]);
      t[i][j] = a;
    }
#pragma unroll
  for (int i = 0; i < R; ++i)
#pragma unroll
    for (int j = 0; j < R; ++j) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) a = cmac(a, Wino<M>::g(j, k), t[i][k]);
      u[i][j] = a;
    }
}
template <int M, bool PLAIN>
__global__ void wino_filter_kernel(const float* __restrict__ w, float* __restrict__ Uf, float* __restrict__ Ud, int Cout, int Cin) {
  constexpr int R = M + 2;
  const i64 total = (i64)Cout * Cin;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Cin), co = (int)(i / Cin);
    float g[3][3], gf[3][3], u[R][R];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        g[a][b] = w[i * 9 + a * 3 + b];
        gf[2 - a][2 - b] = g[a][b];
      }
    if (Uf) {
      filter_tf<M>(g, u);
      const i64 at = PLAIN ? i : ((i64)(ci >> 2) * Cout + co) * 4 + (ci & 3);
#pragma unroll
      for (int xi = 0; xi < R * R; ++xi) Uf[(i64)xi * total + at] = u[xi / R][xi % R];
    }
    if (Ud) {
      filter_tf<M>(gf, u);
      const i64 at = PLAIN ? i : ((i64)(co >> 2) * Cin + ci) * 4 + (co & 3);
#pragma unroll
      for (int xi = 0; xi < R * R; ++xi) Ud[(i64)xi * total + at] = u[xi / R][xi % R];
    }
  }
}

// ---- input transform V = B^T d B.   grid: (blocks over T, C, N); one thread per tile
// vec != 0 (host: dilation 1, W % M == 0, M-float aligned planes): the M interior columns of a patch row are one wide load
template <int M>
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, i64 x_bs, float* __restrict__ V, int N, int C,
                                                         WinoGeom g, int vec, float* __restrict__ amax, const float* __restrict__ pack_amax,
                                                         int pack_shift, const float4* __restrict__ bnl) {
  // bnl != NULL: x is the PRE-normalisation output of the conv -> BN -> ReLU layer feeding this convolution and coef = bnl[c] = (mean, invstd,
  // sc, sh) of that layer: every in-image element is normalised as it is loaded, max(fma(x, sc, sh), 0) -- the expression and rounding of
  // bn_apply -- so the normalised tensor is never written (the padding stays zero: it pads the NORMALISED image).  pack_amax then holds
  // the PREDICTED max |y| (pfst_bn_finalize_partials with the producer's min / max partials).
  // amax != NULL: max |V| over everything this launch writes -> that slot group (amax.h; the f16x3 GEMM's scale of V).
  // pack_amax != NULL (f16x3): V is written PRE-SPLIT -- every element as the two fp16 pieces of V s in one dword (amax.h
  // pack_f16x2_pieces), exactly what the GEMM's in-register split would produce, so the GEMMs that read V (forward product, weight
  // gradient) skip their 24 split instructions per 8 values.  The scale has to be known before V exists: |B^T d B| <= 2^pack_shift max|d|
  // (row sums of B^T: 10 for F(4x4), i.e. 100 < 2^7), so s derives from max|x| -- the slot group pack_amax -- and `amax` receives that
  // BOUND (one store) instead of the measured maximum; the consumers derive the same exponent from it.
  float am = 0.f;
  float ps = 0.f;
  if (pack_amax) {
    int e = amax_exponent(amax_read(pack_amax)) + pack_shift;
    e = e > 254 ? 254 : e;
    ps = scale_of(e);
    if (amax && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) amax[0] = __builtin_bit_cast(float, (unsigned)e << 23);
  }
  constexpr int R = M + 2;
  typedef float vecM __attribute__((ext_vector_type(M)));
  const int c = blockIdx.y, n = blockIdx.z;
  const float* xp = x + (i64)n * x_bs + (i64)c * g.H * g.W;
  const i64 plane = (i64)N * C * g.T;                        // stride between transform indices
  float* vp = V + ((i64)n * C + c) * g.T;
  float nsc = 1.f, nsh = 0.f;
  if (bnl) {
    const float4 cf = bnl[c];
    nsc = cf.z;
    nsh = cf.w;
  }
  auto norm = [&](float v, bool in) { return bnl ? (in ? fmaxf(__fmaf_rn(v, nsc, nsh), 0.f) : 0.f) : v; };
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < g.T; t += gridDim.x * blockDim.x) {
    int sy, sx, ty, tx;
    tile_coord(g, t, sy, sx, ty, tx);
    float d[R][R];
    if (vec) {                                               // d == 1: columns M*tx - 1 | M*tx .. M*tx + M-1 | M*tx + M
      const int x0 = M * tx;
#pragma unroll
      for (int a = 0; a < R; ++a) {
        const int y = M * ty - 1 + a;
        const bool in = y >= 0 && y < g.H;
        const float* row = xp + (i64)y * g.W + x0;
        vecM mid;
#pragma unroll
        for (int b = 0; b < M; ++b) mid[b] = 0.f;
        if (in) mid = *reinterpret_cast<const vecM*>(row);
        const bool inl = in && x0 > 0, inr = in && x0 + M < g.W;
        d[a][0] = norm(inl ? row[-1] : 0.f, inl);
#pragma unroll
        for (int b = 0; b < M; ++b) d[a][1 + b] = norm(mid[b], in);
        d[a][R - 1] = norm(inr ? row[M] : 0.f, inr);
      }
    } else {
#pragma unroll
      for (int a = 0; a < R; ++a) {
        const int y = sy + g.d * (M * ty - 1 + a);
#pragma unroll
        for (int b = 0; b < R; ++b) {
          const int xx = sx + g.d * (M * tx - 1 + b);
          const bool in = y >= 0 && y < g.H && xx >= 0 && xx < g.W;
          d[a][b] = norm(in ? xp[(i64)y * g.W + xx] : 0.f, in);
        }
      }
    }
    float r[R][R];                                           // B^T d
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
      for (int b = 0; b < R; ++b) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < R; ++k) a = cmac(a, Wino<M>::bt(i, k), d[k][b]);
        r[i][b] = a;
      }
#pragma unroll
    for (int i = 0; i < R; ++i)                              // (B^T d) B
#pragma unroll
      for (int j = 0; j < R; ++j) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < R; ++k) a = cmac(a, Wino<M>::bt(j, k), r[i][k]);
        if (pack_amax) {
          reinterpret_cast<unsigned*>(vp)[(i64)(i * R + j) * plane + t] = pack_f16x2_pieces(a * ps);
        } else {
          vp[(i64)(i * R + j) * plane + t] = a;
          am = fmaxf(am, fabsf(a));
        }
      }
  }
  if (amax && !pack_amax) amax_publish(amax, am);
}

// ---- output transform Y = A^T m A (M x M per tile), optional accumulate.   grid: (blocks over T, Cout, N)
// vec != 0 (host: dilation 1, W % M == 0, M-float aligned planes): the M outputs of a tile row are adjacent -> one wide store
template <int M>
__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ Mb, float* __restrict__ y, i64 y_bs, int N, int Cout,
                                                          WinoGeom g, int accumulate, int vec, float* __restrict__ stats, int stats_minmax,
                                                          const float* __restrict__ bnb_x, i64 bnb_x_bs, const float4* __restrict__ bnb_coef, int bnb_relu) {
  // bnb_x != NULL (data-gradient launches): y is the COMPLETE gradient of a conv -> BN [-> ReLU] layer's output (no residual), bnb_x that
  // layer's pre-BN tensor and bnb_coef its (mean, invstd, sc, sh) rows: `stats` receives the layer's BatchNorm-backward partials
  // (sum dz, sum dz * x) per block instead of forward statistics -- dz = the value written, gated by fma(x, sc, sh) > 0 --, in the layout
  // pfst_bn_backward(bwd_partials) reads (as the GEMM epilogue's fused sums, conv_epilogue.h): the layer's reduction pass is not launched
  constexpr int R = M + 2;
  typedef float vecM __attribute__((ext_vector_type(M)));
  __shared__ double red[32];
  float st_s = 0.f, st_q = 0.f;                    // fused BatchNorm statistics of the outputs this block writes (stats != NULL)
  float st_lo = __builtin_inff(), st_hi = -__builtin_inff();      // stats_minmax: and their (minimum, maximum), for the predicted max |relu(bn(y))|
  const int c = blockIdx.y, n = blockIdx.z;
  float* yp = y + (i64)n * y_bs + (i64)c * g.H * g.W;
  const float* bxp = bnb_x ? bnb_x + (i64)n * bnb_x_bs + (i64)c * g.H * g.W : nullptr;
  const float bsc = bnb_x ? bnb_coef[c].z : 0.f, bsh = bnb_x ? bnb_coef[c].w : 0.f;
  const i64 plane = (i64)N * Cout * g.T;
  const float* mp = Mb + ((i64)n * Cout + c) * g.T;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < g.T; t += gridDim.x * blockDim.x) {
    int sy, sx, ty, tx;
    tile_coord(g, t, sy, sx, ty, tx);
    float m[R][R];
#pragma unroll
    for (int i = 0; i < R * R; ++i) m[i / R][i % R] = __builtin_nontemporal_load(&mp[(i64)i * plane + t]);      // the GEMM's output is read once, here (streamed: -0.4 ms per step)
    float r[M][R];                                           // A^T m
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
      for (int b = 0; b < R; ++b) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < R; ++k) a = cmac(a, Wino<M>::at(i, k), m[k][b]);
        r[i][b] = a;
      }
#pragma unroll
    for (int i = 0; i < M; ++i) {
      const int yy = sy + g.d * (M * ty + i);
      if (yy >= g.H) continue;
      float o[M];
#pragma unroll
      for (int j = 0; j < M; ++j) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < R; ++k) a = cmac(a, Wino<M>::at(j, k), r[i][k]);
        o[j] = a;
      }
      const int x0 = sx + g.d * (M * tx);
      if (vec) {
        vecM* q = reinterpret_cast<vecM*>(yp + (i64)yy * g.W + x0);
        vecM v;
#pragma unroll
        for (int j = 0; j < M; ++j) v[j] = o[j];
        if (accumulate) v += *q;
        *q = v;
        if (bxp) {
          const vecM xv = *reinterpret_cast<const vecM*>(bxp + (i64)yy * g.W + x0);
#pragma unroll
          for (int j = 0; j < M; ++j) {
            const float dz = (!bnb_relu || __fmaf_rn(xv[j], bsc, bsh) > 0.f) ? v[j] : 0.f;
            st_s += dz;
            st_q = fmaf(dz, xv[j], st_q);
          }
          continue;
        }
#pragma unroll
        for (int j = 0; j < M; ++j) {
          const float e = v[j];
          st_s += e;
          st_q = fmaf(e, e, st_q);
          st_lo = fminf(st_lo, e);
          st_hi = fmaxf(st_hi, e);
        }
        continue;
      }
#pragma unroll
      for (int j = 0; j < M; ++j) {
        const int xx = x0 + g.d * j;
        if (xx < g.W) {
          float* q = yp + (i64)yy * g.W + xx;
          const float v = accumulate ? *q + o[j] : o[j];
          *q = v;
          if (bxp) {
            const float xs = bxp[(i64)yy * g.W + xx];
            const float dz = (!bnb_relu || __fmaf_rn(xs, bsc, bsh) > 0.f) ? v : 0.f;
            st_s += dz;
            st_q = fmaf(dz, xs, st_q);
          } else {
            st_s += v;
            st_q = fmaf(v, v, st_q);
            st_lo = fminf(st_lo, v);
            st_hi = fmaxf(st_hi, v);
          }
        }
      }
    }
  }
  if (stats) {                                       // stats[c][n * gridDim.x + blockIdx.x][2] for pfst_bn_finalize_partials
    double bs = (double)st_s, bq = (double)st_q;
    block_sum2_d<true>(bs, bq, red);
    if (threadIdx.x == 0) {
      const i64 T = (i64)gridDim.x * gridDim.z;
      float2* dst = reinterpret_cast<float2*>(stats) + ((i64)c * T + (i64)n * gridDim.x + blockIdx.x);
      *dst = make_float2((float)bs, (float)bq);
    }
    if (stats_minmax) {                              // [Cout][T][2] behind the sums, as the GEMM epilogue's stats_mm (conv_epilogue.h)
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        st_lo = fminf(st_lo, __shfl_xor(st_lo, o));
        st_hi = fmaxf(st_hi, __shfl_xor(st_hi, o));
      }
      float* mm = reinterpret_cast<float*>(red);
      __syncthreads();
      if ((threadIdx.x & 63) == 0) { mm[threadIdx.x >> 6] = st_lo; mm[8 + (threadIdx.x >> 6)] = st_hi; }
      __syncthreads();
      if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int i = 1; i < nw; ++i) { st_lo = fminf(st_lo, mm[i]); st_hi = fmaxf(st_hi, mm[8 + i]); }
        const i64 T = (i64)gridDim.x * gridDim.z;
        float2* dst = reinterpret_cast<float2*>(stats) + ((i64)Cout * T + (i64)c * T + (i64)n * gridDim.x + blockIdx.x);
        *dst = make_float2(st_lo, st_hi);
      }
    }
  }
}

// ---- adjoint of the output transform: dM = A dY A^T (R x R per tile) for the weight gradient.   grid: (blocks over T, Cout, N)
template <int M>
__global__ __launch_bounds__(256) void wino_dy_kernel(const float* __restrict__ dy, i64 dy_bs, float* __restrict__ dM, int N, int Cout,
                                                      WinoGeom g, float* __restrict__ amax, const float* __restrict__ pack_amax, int pack_shift) {
  float am = 0.f;                                            // max |dM| -> slot group `amax` when given
  float ps = 0.f;                                            // pack_amax: dM written pre-split (see wino_input_kernel); |A e A^T| <= 15^2 max|e| < 2^8
  if (pack_amax) {
    int e = amax_exponent(amax_read(pack_amax)) + pack_shift;
    e = e > 254 ? 254 : e;
    ps = scale_of(e);
    if (amax && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) amax[0] = __builtin_bit_cast(float, (unsigned)e << 23);
  }
  constexpr int R = M + 2;
  const int c = blockIdx.y, n = blockIdx.z;
  const float* gp = dy + (i64)n * dy_bs + (i64)c * g.H * g.W;
  const i64 plane = (i64)N * Cout * g.T;
  float* mp = dM + ((i64)n * Cout + c) * g.T;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < g.T; t += gridDim.x * blockDim.x) {
    int sy, sx, ty, tx;
    tile_coord(g, t, sy, sx, ty, tx);
    float e[M][M];
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
      for (int j = 0; j < M; ++j) {
        const int yy = sy + g.d * (M * ty + i), xx = sx + g.d * (M * tx + j);
        e[i][j] = (yy < g.H && xx < g.W) ? gp[(i64)yy * g.W + xx] : 0.f;
      }
    float r[R][M];                                           // A e   (A = (A^T)^T)
#pragma unroll
    for (int a = 0; a < R; ++a)
#pragma unroll
      for (int j = 0; j < M; ++j) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < M; ++i) s = cmac(s, Wino<M>::at(i, a), e[i][j]);
        r[a][j] = s;
      }
#pragma unroll
    for (int a = 0; a < R; ++a)                              // (A e) A^T
#pragma unroll
      for (int b = 0; b < R; ++b) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < M; ++j) s = cmac(s, Wino<M>::at(j, b), r[a][j]);
        if (pack_amax) {
          reinterpret_cast<unsigned*>(mp)[(i64)(a * R + b) * plane + t] = pack_f16x2_pieces(s * ps);
        } else {
          mp[(i64)(a * R + b) * plane + t] = s;
          am = fmaxf(am, fabsf(s));
        }
      }
  }
  if (amax && !pack_amax) amax_publish(amax, am);
}

// ---- dW += G^T dU G.   dU [R*R][Cout][Cin] (row-major as the 1x1 wgrad kernel writes it).  One thread per (co, ci)
template <int M>
__global__ void wino_dw_kernel(const float* __restrict__ dU, float* __restrict__ dw, int Cout, int Cin) {
  constexpr int R = M + 2;
  const i64 total = (i64)Cout * Cin;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
    float u[R][R];
#pragma unroll
    for (int xi = 0; xi < R * R; ++xi) u[xi / R][xi % R] = dU[(i64)xi * total + i];
    float t[3][R];                                           // G^T u
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int b = 0; b < R; ++b) {
        float s = 0.f;
#pragma unroll
        for (int a = 0; a < R; ++a) s = cmac(s, Wino<M>::g(a, p), u[a][b]);
        t[p][b] = s;
      }
#pragma unroll
    for (int p = 0; p < 3; ++p)                              // (G^T u) G
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        float s = 0.f;
#pragma unroll
        for (int b = 0; b < R; ++b) s = cmac(s, Wino<M>::g(b, q), t[p][b]);
        dw[i * 9 + p * 3 + q] += s;
      }
  }
}

// ---- batched weight preparation (pfst_weight_prep_batched): one workgroup = 256 filters of one Winograd layer (transform to the plain
// sets + every set's absolute maximum) or 4096 weights of one directly convolved layer (absolute maximum)
template <int M>
__device__ __forceinline__ void weight_prep_wino(const pfst_weight_job_t& J, int lb, float (*red)[36]) {
  constexpr int R = M + 2, X = R * R;
  const i64 total = (i64)J.Cout * J.Cin;
  const i64 i = (i64)lb * 256 + threadIdx.x;
  const bool live = i < total;
  float g[3][3], gf[3][3], u[R][R];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      g[a][b] = live ? J.src[i * 9 + a * 3 + b] : 0.f;
      gf[2 - a][2 - b] = g[a][b];
    }
#pragma unroll
  for (int dir = 0; dir < 2; ++dir) {
    float* dst = reinterpret_cast<float*>(dir ? J.dst_d : J.dst_f);
    float* amax = dir ? J.amax_d : J.amax_f;
    if (!dst) continue;                                  // uniform over the launch's job
    filter_tf<M>(dir ? gf : g, u);
#pragma unroll
    for (int xi = 0; xi < X; ++xi) {
      const float v = u[xi / R][xi % R];                 // dead lanes transformed zeros
      if (live) dst[(i64)xi * total + i] = v;
      float m = fabsf(v);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][xi] = m;
    }
    __syncthreads();
    if (threadIdx.x < X) {
      const float m = fmaxf(fmaxf(red[0][threadIdx.x], red[1][threadIdx.x]), fmaxf(red[2][threadIdx.x], red[3][threadIdx.x]));
      if (m > 0.f)
        atomicMax(reinterpret_cast<unsigned*>(amax) + (i64)threadIdx.x * PFST_AMAX_SUB + (lb & (PFST_AMAX_SUB - 1)), __builtin_bit_cast(unsigned, m));
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void weight_prep_batched_kernel(const pfst_weight_job_t* __restrict__ jobs, int njobs) {
  __shared__ float red[4][36];
  const pfst_weight_job_t J = jobs[weight_job_of_block(jobs, njobs, blockIdx.x)];
  const int lb = blockIdx.x - J.first_block;
  if (J.m == 2) return weight_prep_wino<2>(J, lb, red);
  if (J.m == 4) return weight_prep_wino<4>(J, lb, red);
  const i64 n = (i64)J.Cout * J.Cin * J.T;
  float m = 0.f;
#pragma unroll 4
  for (int k = 0; k < 16; ++k) {
    const i64 i = (i64)lb * 4096 + k * 256 + threadIdx.x;
    if (i < n) m = fmaxf(m, fabsf(J.src[i]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][0] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0][0], red[1][0]), fmaxf(red[2][0], red[3][0]));
    if (m > 0.f) atomicMax(reinterpret_cast<unsigned*>(J.amax_f) + (lb & (PFST_AMAX_SUB - 1)), __builtin_bit_cast(unsigned, m));
  }
}

inline int tile_blocks(int T) {
  int b = (T + 255) / 256;
  return b < 1 ? 1 : b;
}

#define PFST_WINO_M(m_, CALL2, CALL4) \
  do {                                \
    if ((m_) == 2) { CALL2; } else { CALL4; } \
  } while (0)

}  // namespace

#define PFST_CHECK_TILE(m) PFST_CHECK_ARG((m) == 2 || (m) == 4)

extern "C" int pfst_wino_tiles(int H, int W, int dil, int m) {
  if (H <= 0 || W <= 0 || dil < 1 || (m != 2 && m != 4)) return 0;
  return wino_geom(H, W, dil, m).T;
}

extern "C" int pfst_wino_pack_weight(const float* w, float* U_fprop, float* U_dgrad, int Cout, int Cin, int m, pfst_stream_t stream) {
  PFST_CHECK_ARG(w && (U_fprop || U_dgrad) && Cout > 0 && Cin > 0);
  PFST_CHECK_TILE(m);
  PFST_CHECK_ARG((!U_fprop || Cin % 16 == 0) && (!U_dgrad || Cout % 16 == 0));     // K-quad GEMM kernel: K % 16 == 0
  const dim3 grid(ew_grid((i64)Cout * Cin));
  PFST_WINO_M(m, hipLaunchKernelGGL((wino_filter_kernel<2, false>), grid, dim3(256), 0, (hipStream_t)stream, w, U_fprop, U_dgrad, Cout, Cin),
              hipLaunchKernelGGL((wino_filter_kernel<4, false>), grid, dim3(256), 0, (hipStream_t)stream, w, U_fprop, U_dgrad, Cout, Cin));
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

extern "C" int pfst_wino_filter_plain(const float* w, float* P_fprop, float* P_dgrad, int Cout, int Cin, int m, pfst_stream_t stream) {
  PFST_CHECK_ARG(w && (P_fprop || P_dgrad) && Cout > 0 && Cin > 0);
  PFST_CHECK_TILE(m);
  const dim3 grid(ew_grid((i64)Cout * Cin));
  PFST_WINO_M(m, hipLaunchKernelGGL((wino_filter_kernel<2, true>), grid, dim3(256), 0, (hipStream_t)stream, w, P_fprop, P_dgrad, Cout, Cin),
              hipLaunchKernelGGL((wino_filter_kernel<4, true>), grid, dim3(256), 0, (hipStream_t)stream, w, P_fprop, P_dgrad, Cout, Cin));
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

extern "C" int pfst_weight_job_blocks(const pfst_weight_job_t* job, int pack) { return job ? weight_job_blocks(*job, pack) : 0; }

extern "C" int pfst_weight_prep_batched(const pfst_weight_job_t* jobs_host, const pfst_weight_job_t* jobs_dev, int njobs, pfst_stream_t stream) {
  PFST_CHECK_ARG(jobs_host && jobs_dev && njobs > 0);
  i64 blocks = 0;
  for (int j = 0; j < njobs; ++j) {
    const pfst_weight_job_t& J = jobs_host[j];
    PFST_CHECK_ARG(J.src && J.Cout > 0 && J.Cin > 0 && (J.m == 0 || J.m == 2 || J.m == 4) && J.first_block == blocks);
    if (J.m == 0) PFST_CHECK_ARG(J.amax_f && (J.T == 1 || J.T == 9));
    else PFST_CHECK_ARG(J.T == 9 && (J.dst_f || J.dst_d) && (!J.dst_f || J.amax_f) && (!J.dst_d || J.amax_d));
    blocks += weight_job_blocks(J, 0);
  }
  PFST_CHECK_ARG(blocks < (1ll << 31));
  hipLaunchKernelGGL(weight_prep_batched_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, jobs_dev, njobs);
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

extern "C" int pfst_wino_input(const float* x, long long x_bs, float* V, int N, int C, int H, int W, int dil, int m, float* v_amax,
                               const float* pack_x_amax, const float* bnl, pfst_stream_t stream) {
  PFST_CHECK_ARG(!pack_x_amax || v_amax);
  PFST_CHECK_ARG(!bnl || ((uintptr_t)bnl & 15) == 0);
  const int shift_in = m == 4 ? 7 : 3;                // (max row sum of |B^T|)^2: 100 for F(4x4), 4 for F(2x2)
  PFST_CHECK_ARG(x && V && N > 0 && N <= 65535 && C > 0 && C <= 65535 && H > 0 && W > 0 && dil >= 1 && x_bs >= (i64)C * H * W);
  PFST_CHECK_TILE(m);
  const WinoGeom g = wino_geom(H, W, dil, m);
  const int vec = dil == 1 && W % m == 0 && x_bs % m == 0 && ((uintptr_t)x & (4 * m - 1)) == 0;
  const dim3 grid(tile_blocks(g.T), C, N);
  PFST_WINO_M(m, hipLaunchKernelGGL(wino_input_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, x, x_bs, V, N, C, g, vec, v_amax, pack_x_amax, shift_in, (const float4*)bnl),
              hipLaunchKernelGGL(wino_input_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, x, x_bs, V, N, C, g, vec, v_amax, pack_x_amax, shift_in, (const float4*)bnl));
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

// the R*R transform-domain GEMMs: Mbuf[xi][n][M][T] = U[xi] (M x K) * V[xi][n] (K x T)
extern "C" int pfst_wino_gemm(const float* V, const float* U, float* Mbuf, int N, int K, int M, int T, int m, pfst_stream_t stream) {
  PFST_CHECK_ARG(V && U && Mbuf && N > 0 && N <= 65535 && K > 0 && K % 16 == 0 && M > 0 && T > 0);
  PFST_CHECK_TILE(m);
  PFST_CHECK_ARG((i64)K * T * 4 < (1ll << 31) && (i64)M * T * 4 < (1ll << 31) && (i64)K * M * 4 < (1ll << 31));
  return pfst_igemm_q_launch(V, (i64)K * T, U, nullptr, Mbuf, (i64)M * T, N, K, 1, T, M, 1, T, 1, 1, 1, 0, 1, 0, nullptr, 0,
                             (m + 2) * (m + 2), (hipStream_t)stream);
}

extern "C" int pfst_wino_stats_slots(int H, int W, int dil, int m) {
  if (H <= 0 || W <= 0 || dil < 1 || (m != 2 && m != 4)) return 0;
  return tile_blocks(wino_geom(H, W, dil, m).T);
}

extern "C" int pfst_wino_output(const float* Mbuf, float* y, long long y_bs, int N, int Cout, int H, int W, int dil, int accumulate,
                                float* stats, int stats_minmax, const float* bnb_x, long long bnb_x_bs, const float* bnb_coef, int bnb_relu, int m,
                                pfst_stream_t stream) {
  PFST_CHECK_ARG(Mbuf && y && N > 0 && N <= 65535 && Cout > 0 && Cout <= 65535 && H > 0 && W > 0 && dil >= 1 && y_bs >= (i64)Cout * H * W);
  PFST_CHECK_ARG(!stats_minmax || stats);
  PFST_CHECK_ARG(!bnb_x || (stats && !stats_minmax && bnb_coef && bnb_x_bs >= (i64)Cout * H * W && (bnb_x_bs % 4) == 0 && ((uintptr_t)bnb_x & 15) == 0));
  PFST_CHECK_TILE(m);
  const WinoGeom g = wino_geom(H, W, dil, m);
  const int vec = dil == 1 && W % m == 0 && y_bs % m == 0 && ((uintptr_t)y & (4 * m - 1)) == 0;
  const dim3 grid(tile_blocks(g.T), Cout, N);
  PFST_WINO_M(m, hipLaunchKernelGGL(wino_output_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, Mbuf, y, y_bs, N, Cout, g, accumulate, vec, stats, stats_minmax, bnb_x,
                                 (i64)bnb_x_bs, reinterpret_cast<const float4*>(bnb_coef), bnb_relu),
              hipLaunchKernelGGL(wino_output_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, Mbuf, y, y_bs, N, Cout, g, accumulate, vec, stats, stats_minmax, bnb_x,
                                 (i64)bnb_x_bs, reinterpret_cast<const float4*>(bnb_coef), bnb_relu));
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

extern "C" int pfst_wino_dy(const float* dy, long long dy_bs, float* dM, int N, int Cout, int H, int W, int dil, int m, float* dm_amax,
                            const float* pack_dy_amax, pfst_stream_t stream) {
  PFST_CHECK_ARG(!pack_dy_amax || dm_amax);
  const int shift_dy = m == 4 ? 8 : 2;                // (max row sum of |A|)^2: 225 for F(4x4), 4 for F(2x2)
  PFST_CHECK_ARG(dy && dM && N > 0 && N <= 65535 && Cout > 0 && Cout <= 65535 && H > 0 && W > 0 && dil >= 1 && dy_bs >= (i64)Cout * H * W);
  PFST_CHECK_TILE(m);
  const WinoGeom g = wino_geom(H, W, dil, m);
  const dim3 grid(tile_blocks(g.T), Cout, N);
  PFST_WINO_M(m, hipLaunchKernelGGL(wino_dy_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, dy, dy_bs, dM, N, Cout, g, dm_amax, pack_dy_amax, shift_dy),
              hipLaunchKernelGGL(wino_dy_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, dy, dy_bs, dM, N, Cout, g, dm_amax, pack_dy_amax, shift_dy));
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}

// dU[xi][Cout][Cin] = sum_{n,t} dM[xi][n][Cout][T] V[xi][n][Cin][T]  (one grouped launch of the 1x1 K-quad wgrad), then dW += G^T dU G.
// dU is scratch of (m+2)^2*Cout*Cin floats (zeroed here).
extern "C" int pfst_wino_wgrad(const float* V, const float* dM, float* dU, float* dw, int N, int Cin, int Cout, int T, int m,
                               int split, const float* v_amax, const float* dm_amax, int packed, pfst_stream_t stream) {
  PFST_CHECK_ARG(V && dM && dU && dw && N > 0 && N <= 65535 && Cin > 0 && Cout > 0 && T > 0 && T % 4 == 0);
  PFST_CHECK_ARG(split != 2 || (v_amax && dm_amax && Cout > 64));
  PFST_CHECK_ARG(!packed || split == 2);       // packed: V and dM are stored pre-split (pfst_wino_input / pfst_wino_dy packed modes)
  PFST_CHECK_TILE(m);
  hipStream_t s = (hipStream_t)stream;
  const i64 uc = (i64)Cout * Cin;
  const int nx = (m + 2) * (m + 2);
  if (hipMemsetAsync(dU, 0, nx * uc * sizeof(float), s) != hipSuccess) return PFST_ERR_LAUNCH;
  // the per-transform-index products as ONE grouped launch: enough workgroups without splitting the tile range further
  // split != 0: the products on the bf16 matrix cores with the fp32-faithful 6-term split (conv_split.hip)
  // split == 2: the two-piece fp16 split (conv_f16x3.hip), scales from the slot groups the transforms published max |V| / max |dM| to
  const int rc = split == 2 ? pfst_wgrad_f16x3_launch(V, (i64)Cin * T, dM, (i64)Cout * T, dU, N, Cin, Cout, T, nx, (i64)N * Cin * T,
                                                      (i64)N * Cout * T, uc, v_amax, dm_amax, packed, s)
                 : split ? pfst_wgrad_split_q_launch(V, (i64)Cin * T, dM, (i64)Cout * T, dU, N, Cin, Cout, T, nx, (i64)N * Cin * T,
                                                   (i64)N * Cout * T, uc, s)
                       : pfst_wgrad_q_launch(V, (i64)Cin * T, dM, (i64)Cout * T, dU, N, Cin, 1, T, Cout, 1, T, 1, 1, 0, nx, (i64)N * Cin * T,
                                             (i64)N * Cout * T, uc, s);
  if (rc != PFST_OK) return rc;
  PFST_WINO_M(m, hipLaunchKernelGGL(wino_dw_kernel<2>, dim3(ew_grid(uc)), dim3(256), 0, s, dU, dw, Cout, Cin),
              hipLaunchKernelGGL(wino_dw_kernel<4>, dim3(ew_grid(uc)), dim3(256), 0, s, dU, dw, Cout, Cin));
  PFST_CHECK_LAUNCH();
  return PFST_OK;
}
