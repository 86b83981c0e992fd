/* pfst_hip.h -- C ABI of libpfst_hip.so: the MI355X (gfx950) kernels of the PFST train step.
 *
 * The reference (zhu-xlab/PFST, `rsiseg`) has no native layer: every op below is, in the
 * reference, a PyTorch/mmcv call inside PFGST.train_step (rsiseg/models/uda/pfgst.py:129-356).
 * Each entry point names the reference call it replaces.  Conventions:
 *   - every pointer is a DEVICE pointer owned by the caller (torch allocates); no hidden state,
 *     no allocation, no synchronisation inside; work is enqueued on `stream` (a hipStream_t);
 *   - tensors are fp32 NCHW; `*_bs` arguments are batch strides in elements so a channel slice of
 *     a bigger tensor can be passed (concatenations are never materialised);
 *   - return 0 on success, <0 on error (-1 bad argument, -2 launch failure, -3 unsupported);
 *     pfst_last_error() returns a static description of the last failure of the calling process;
 *   - re-entrant per stream.
 * The parser in pfst_amd/_lib.py reads this file to build the ctypes signatures, so keep one
 * declaration per statement and only the scalar types used below.
 */
#ifndef PFST_HIP_H
#define PFST_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

typedef void* pfst_stream_t; /* hipStream_t */

int pfst_abi_version(void);
const char* pfst_last_error(void);
/* Deterministic mode, the counterpart of torch.backends.cudnn.deterministic = True that the reference's `--deterministic` sets
 * (rsiseg/apis/train.py:52-68): weight gradients, BatchNorm-backward sums and depthwise weight gradients are summed in a fixed order instead of
 * by atomic adds of several workgroups -- the gradient of a step is bit-identical run to run; slower.  Process-wide, read at every launch. */
int pfst_set_deterministic(int on);
int pfst_get_deterministic(void);

/* ---- elementwise utilities -------------------------------------------------------------- */
int pfst_fill_f32(float* p, long long n, float value, pfst_stream_t stream);
/* y[i] += alpha * x[i]   (autograd's gradient accumulation) */
int pfst_axpy_f32(float* y, const float* x, float alpha, long long n, pfst_stream_t stream);
/* int64 / uint8 label conversions (decode_head.py:209 `.type(torch.LongTensor)`) */
int pfst_i64_to_u8(const long long* src, unsigned char* dst, long long n, pfst_stream_t stream);
int pfst_u8_to_i64(const unsigned char* src, long long* dst, long long n, pfst_stream_t stream);

/* fused BatchNorm-backward sums of a data-gradient launch, see pfst_conv_igemm */
typedef struct pfst_bnb_fuse {
  const float* x;            /* pre-BN tensor [N][M][P] of the layer that owns the gradient being written */
  long long x_bs;            /* its batch stride (elements) */
  const float* y;            /* that layer's output: ReLU gate = y > 0 (residual layers); NULL: gate recomputed from x as bn_apply did */
  long long y_bs;
  const float* coef;         /* [M][4] = (mean, invstd, sc, sh) of that layer as pfst_bn_finalize_partials / pfst_bn_stats wrote them */
  float* partials;           /* out: [M][T][2] = (sum dz, sum dz * x) per channel and slot */
  int relu;
  const unsigned long long* y_mask; /* optional, with y: pfst_bn_apply's ReLU bitmask of y ([N][M][HW / 64] words, HW % 256 == 0) -- the f16x3
                                      kernel then takes a residual layer's gate bits from it instead of reading y */
} pfst_bnb_fuse_t;

/* What a kernel needs to apply the SECOND pass of BatchNorm backward on the fly, per channel: dx = gs * (dz - m1 - xhat * m2) with
 * dz = dy * [x * sc + sh > 0], xhat = (x - mu) * is, in the arithmetic of pfst_bn_backward's apply pass (fp64 projections).  Written by
 * pfst_bn_backward_sums, read by pfst_dwconv3x3_bwd / pfst_dwconv3x3_multi_bwd (the depthwise layers' backward forms dL/dpre itself). */
typedef struct pfst_bn_bwd_rec {
  double m1, m2, gs;
  float mu, is, sc, sh;
} pfst_bn_bwd_rec_t;

/* ---- dense convolution as implicit GEMM on fp32 MFMA (F.conv2d, groups=1) ---------------
 * resnet.py:169-209 (Bottleneck 1x1/3x3), resnet.py:593-624 (stem), aspp_head.py:32-42,85-92,
 * fcn_head.py:40-49, decode_head.py:242-247 (conv_seg), mmcv pointwise convs. */
/* w[Cout][Cin][T] -> wk_fprop[(t*Cin+ci)][Cout] and wk_dgrad[(t*Cout+co)][Cin] (either may be NULL) */
int pfst_conv_pack_weight(const float* w, float* wk_fprop, float* wk_dgrad, int Cout, int Cin, int T, pfst_stream_t stream);
/* out[n][m][oy][ox] (+)= bias[m] + sum_{t,c} wk[(t*C+c)][m] * in[n][c][sy][sx]
 *   mode 0 (fprop): sy = oy*stride + ty*dil - pad          in = x,  C = Cin,  M = Cout
 *   mode 1 (dgrad): sy = (oy + pad - ty*dil) / stride      in = dy, C = Cout, M = Cin  (exact division only)
 * ksize in {1,3}; accumulate != 0 adds into `out`. */
int pfst_conv_igemm(const float* in, long long in_bs, const float* wk, const float* bias, float* out, long long out_bs,
                    int N, int C, int Hi, int Wi, int M, int Ho, int Wo, int ksize, int stride, int dil, int pad,
                    int mode, int accumulate, float* stats, const pfst_bnb_fuse_t* bnb, pfst_stream_t stream);
/* Fused BatchNorm-BACKWARD sums (bnb != NULL, data-gradient launches): when `out` is the COMPLETE gradient dL/dy of a
 * conv -> BN(train) -> [+residual] -> ReLU layer's output (this launch is the last writer of that buffer; with accumulate != 0 the
 * sums are taken over old + new), the epilogue also writes per-channel partials of  S1 = sum dz,  S2' = sum dz * x  (dz = dy * ReLU gate)
 * to bnb->partials[M][N * pfst_conv_stats_slots(M, Ho, Wo)][2]; pfst_bn_backward then skips its reduction pass (torch's
 * native_batch_norm_backward makes the same two sums).  Needs C % 16 == 0, M a multiple of the row tile (128 for M > 64, else 64 / 32),
 * no bias, stats == NULL; returns PFST_ERR_ARG otherwise (the caller then simply does not fuse). */
/* Fused BatchNorm statistics: if `stats` != NULL the epilogue also writes per-channel partial (sum, sum of squares)
 * pairs to stats[M][N * pfst_conv_stats_slots(M, Ho, Wo)][2] (fp32, no atomics); reduce them with
 * pfst_bn_finalize_partials.  Saves the separate full-tensor read of pfst_bn_stats. */
int pfst_conv_stats_slots(int M, int Ho, int Wo);
/* The same convolution on the bf16 matrix cores with fp32-faithful arithmetic: operands are split exactly into three
 * bf16 pieces and the six piece-products >= 2^-16 are accumulated in fp32 (csrc/conv_split.hip).  Needs C % 16 == 0
 * (returns -3 otherwise).  wk6_*: pfst_conv_pack_weight_split images, 6*Cout*Cin*T bytes each. */
int pfst_conv_pack_weight_split(const float* w, void* wk6_fprop, void* wk6_dgrad, int Cout, int Cin, int T, pfst_stream_t stream);
int pfst_conv_igemm_split(const float* in, long long in_bs, const void* wk6, const float* bias, float* out, long long out_bs,
                          int N, int C, int Hi, int Wi, int M, int Ho, int Wo, int ksize, int stride, int dil, int pad,
                          int mode, int accumulate, float* stats, const pfst_bnb_fuse_t* bnb, pfst_stream_t stream);
int pfst_conv_wgrad_split(const float* x, long long x_bs, const float* dy, long long dy_bs, float* dw,
                          int N, int Cin, int Hi, int Wi, int Cout, int Ho, int Wo, int ksize, int stride, int dil, int pad,
                          pfst_stream_t stream);
/* dw[co][ci][t] += sum_{n,oy,ox} dy[n][co][oy][ox] * x[n][ci][oy*stride+ty*dil-pad][...]   (atomic fp32 adds) */
int pfst_conv_wgrad(const float* x, long long x_bs, const float* dy, long long dy_bs, float* dw,
                    int N, int Cin, int Hi, int Wi, int Cout, int Ho, int Wo, int ksize, int stride, int dil, int pad,
                    pfst_stream_t stream);
/* Occupancy cap of the K-quad weight-gradient kernels: `bytes` of unused dynamic LDS per workgroup (0 = off, the default).  With
 * 24000 two instead of four workgroups fit per CU (the fp32-MFMA pipe stays saturated: -0.3 % on the step) and an HBM-bound kernel
 * of ANOTHER stream -- the BatchNorm-backward chain, when the host runs weight gradients on a side stream -- can be co-resident. */
int pfst_conv_wgrad_set_lds_pad(int bytes);
/* db[c] += sum_{n,hw} dy[n][c][hw] */
int pfst_bias_grad(const float* dy, long long dy_bs, float* db, int N, int C, int HW, pfst_stream_t stream);

/* ---- Winograd F(m x m, 3x3), m = 2 or 4, for wide stride-1 'same' 3x3 convolutions (csrc/conv_winograd.hip): the same F.conv2d /
 * autograd results with 2.25x (m = 2) or 4x (m = 4) fewer MACs.  X = (m+2)^2 transform indices; transform-domain tensors
 * V [X][N][C][T], U [X][K/4][M][4], Mbuf [X][N][M][T], T = pfst_wino_tiles(H, W, dil, m) tiles per image (dilation d = d*d
 * interleaved sub-grids).  fp32 error vs an fp64 direct convolution: 3e-6 (m = 2), 3e-5 (m = 4) of the mean |y|.
 *   fprop : pfst_wino_input(x) -> V;   pfst_wino_gemm(V, U_fprop) -> Mbuf;   pfst_wino_output(Mbuf) -> y
 *   dgrad : the same on dy with U_dgrad (flipped, transposed filter), pfst_wino_output(..., accumulate)
 *   wgrad : pfst_wino_input(x) -> V;  pfst_wino_dy(dy) -> dM;  pfst_wino_wgrad(V, dM, scratch dU[X*Cout*Cin]) : dw += ... */
int pfst_wino_tiles(int H, int W, int dil, int m);
int pfst_wino_pack_weight(const float* w, float* U_fprop, float* U_dgrad, int Cout, int Cin, int m, pfst_stream_t stream);
int pfst_wino_input(const float* x, long long x_bs, float* V, int N, int C, int H, int W, int dil, int m, float* v_amax,
                    const float* pack_x_amax, const float* bnl, pfst_stream_t stream);
/* bnl != NULL: coef [C][4] = (mean, invstd, sc, sh) of the conv -> BN -> ReLU layer whose PRE-normalisation output x is (Bottleneck conv1 ->
 * bn1 -> relu -> conv2, resnet.py:273-281): elements are normalised as they are loaded, the normalised tensor is never written; pack_x_amax
 * then is the slot group pfst_bn_finalize_partials published the predicted max |relu(bn(x))| to */
/* v_amax: NULL, or the slot group (1024 floats, zeroed; csrc/amax.h) that receives max |V| (f16x3 scale).  pack_x_amax != NULL: V is
 * written PRE-SPLIT for the f16x3 GEMMs (one dword per element: the two fp16 pieces of V s), scaled from the slot group holding max |x|
 * through the transform's norm bound, and v_amax receives that bound (pass v_packed = 1 / packed = 1 to the consumers) */
int pfst_wino_gemm(const float* V, const float* U, float* Mbuf, int N, int K, int M, int T, int m, pfst_stream_t stream);
int pfst_wino_output(const float* Mbuf, float* y, long long y_bs, int N, int Cout, int H, int W, int dil, int accumulate,
                     float* stats, int stats_minmax, const float* bnb_x, long long bnb_x_bs, const float* bnb_coef, int bnb_relu, int m,
                     pfst_stream_t stream);
/* bnb_x != NULL (the output transform of a DATA-GRADIENT launch whose `y` is the complete gradient of a conv -> BN [-> ReLU] layer's output,
 * no residual -- Bottleneck bn1 behind a Winograd conv2): bnb_x = that layer's pre-BN tensor, bnb_coef its coef [Cout][4]; `stats` then
 * receives the layer's BatchNorm-backward partials (sum dz, sum dz * x) [Cout][N * pfst_wino_stats_slots][2] for pfst_bn_backward(bwd_partials),
 * as pfst_conv_igemm's bnb does for the GEMM data gradients: the layer's reduction pass is not launched */
/* stats != NULL: BatchNorm partials of the output, stats[Cout][N * pfst_wino_stats_slots(H, W, dil, m)][2]; stats_minmax != 0: `stats` has room
 * for twice that and also receives the per-channel (minimum, maximum) partials behind the sums (as pfst_conv_igemm_f16x3's stats_minmax) */
int pfst_wino_stats_slots(int H, int W, int dil, int m);
int pfst_wino_dy(const float* dy, long long dy_bs, float* dM, int N, int Cout, int H, int W, int dil, int m, float* dm_amax,
                 const float* pack_dy_amax, pfst_stream_t stream);
int pfst_wino_wgrad(const float* V, const float* dM, float* dU, float* dw, int N, int Cin, int Cout, int T, int m,
                    int split, const float* v_amax, const float* dm_amax, int packed, pfst_stream_t stream);
/* split = 1: the transform-domain products with the fp32-faithful bf16x6 split on the bf16 matrix cores; split = 2: with the f16x3 split,
 * v_amax / dm_amax = the slot groups pfst_wino_input / pfst_wino_dy published the operands' absolute maxima to (NULL otherwise) */
/* the same GEMMs on the fp32-faithful bf16x6 path: plain [X][Cout][Cin] filter sets (normal / flipped) -> X split-packed sets of
 * 6*Cout*Cin bytes each -> pfst_wino_gemm_split */
int pfst_wino_filter_plain(const float* w, float* P_fprop, float* P_dgrad, int Cout, int Cin, int m, pfst_stream_t stream);
int pfst_wino_pack_weight_split(const float* plain_f, const float* plain_d, void* U6_fprop, void* U6_dgrad, int Cout, int Cin,
                                int m, pfst_stream_t stream);
int pfst_wino_gemm_split(const float* V, const void* U6, float* Mbuf, int N, int K, int M, int T, int m, pfst_stream_t stream);

/* ---- the same GEMMs with the fp32-faithful TWO-piece fp16 split (csrc/conv_f16x3.hip): three fp16 MFMAs per product instead of
 * six bf16 ones.  Every operand tensor is scaled by the power of two its absolute maximum implies; the maxima are device
 * scalars (fp32 slots) written by pfst_absmax or by the kernels that produce the operands -- never read back to the host.
 * Replaces the same reference calls as pfst_conv_igemm (F.conv2d / its data gradient inside mmcv ConvModule:
 * rsiseg/models/backbones/resnet.py:273-305, decode_heads/aspp_head.py:34-42,85-92).  Needs C % 32 == 0 and M > 64 (-3 otherwise). */
int pfst_absmax(const float* x, long long n, int planes, long long plane_stride, int slot_stride, float* slots, pfst_stream_t stream);
int pfst_conv_pack_weight_f16x2(const float* w, void* wk4_fprop, void* wk4_dgrad, int Cout, int Cin, int T, int sets,
                                const float* amax, pfst_stream_t stream);
/* Batched weight preparation of a whole network (two launches per model and step instead of 2-5 tiny ones per convolution; the
 * reference re-reads its weights inside every cuDNN call and has no such step).  A job table lives on the device; the host copy is
 * passed beside it and checked (pointers, shapes, block prefix) before the launch.
 *   pfst_weight_prep_batched:  m == 0: amax_f[0] (one slot group) <- max |src[Cout*Cin*T]|
 *                              m == 2/4: src = 3x3 filters; dst_f / dst_d (either may be NULL) <- the X = (m+2)^2 plain transform-domain
 *                              sets [X][Cout][Cin] of pfst_wino_filter_plain, amax_f / amax_d <- X consecutive slot groups with each set's maximum
 *   pfst_conv_pack_weight_f16x2_batched: src [sets][Cout][Cin][T] -> dst_f / dst_d two-piece fp16 images exactly as
 *                              pfst_conv_pack_weight_f16x2 writes them, scales from the `sets` consecutive slot groups at amax_f
 * The slot groups must be zeroed before the prep launch.  first_block: prefix sum of pfst_weight_job_blocks over the table. */
typedef struct pfst_weight_job {
  const float* src;
  void* dst_f;
  void* dst_d;
  float* amax_f;
  float* amax_d;
  int Cout, Cin, T, sets, m, first_block;
} pfst_weight_job_t;
int pfst_weight_job_blocks(const pfst_weight_job_t* job, int pack);     /* workgroups one job takes in the prep (0) / pack (1) launch */
int pfst_weight_prep_batched(const pfst_weight_job_t* jobs_host, const pfst_weight_job_t* jobs_dev, int njobs, pfst_stream_t stream);
int pfst_conv_pack_weight_f16x2_batched(const pfst_weight_job_t* jobs_host, const pfst_weight_job_t* jobs_dev, int njobs,
                                        pfst_stream_t stream);
int pfst_conv_igemm_f16x3(const float* in, long long in_bs, const void* wk4, const float* w_amax, const float* in_amax,
                          const float* bias, float* out, long long out_bs, int N, int C, int Hi, int Wi, int M, int Ho, int Wo,
                          int ksize, int stride, int dil, int pad, int mode, int accumulate, float* stats, const pfst_bnb_fuse_t* bnb,
                          const float* gate_dy, long long gate_dy_bs, const unsigned long long* gate_mask, int stats_minmax,
                          const float* bnl, pfst_stream_t stream);
/* bnl != NULL (forward 1x1 launches, M % 256 == 0, C <= 2048): `in` is the PRE-normalisation output of the conv -> BN -> ReLU layer feeding this
 * convolution (Bottleneck conv2 -> bn2 -> relu -> conv3, resnet.py:282-290) and bnl its coef [C][4] = (mean, invstd, sc, sh): elements are
 * normalised between their load and their split, the normalised tensor is never written; in_amax = the slot group
 * pfst_bn_finalize_partials published the predicted max |relu(bn(in))| to */
/* stats_minmax != 0 (with stats): stats has room for 4 * M * slots floats; behind the [M][slots][2] (sum, sum of squares) partials the
 * launch writes [M][slots][2] (minimum, maximum) partials of the output, from which pfst_bn_finalize_partials predicts max |relu(bn(out))| */
/* gate_dy != NULL (mode 1, accumulate 0, M % 128 == 0, Ho * Wo % 256 == 0): out = data gradient + (bit ? gate_dy : 0), gate_mask = the ReLU
 * bitmask pfst_bn_apply wrote for an [N][M][Ho * Wo] tensor -- the identity branch of a residual block (resnet.py:149-167, out = relu(bn3 +
 * identity)): dL/d(block input) = conv1's data gradient + dL/d(block output) gated by that ReLU, formed in conv1's epilogue instead of
 * being written by pfst_bn_backward (dres) and read back here */
/* The f16x3 GEMMs walk their tiles in chains of up to 8 per workgroup, the grid sized for the resident workgroup slots of the device
 * (2 per CU).  pfst_f16x3_set_slots overrides that number (0 = the device's): a test hook that makes small problems chain. */
int pfst_f16x3_set_slots(int slots);
int pfst_f16x3_chain_grid(long long total_tiles, int chainable);   /* the workgroup count such a launch uses */
/* Test hook: the two-piece split of n values (n % 8 == 0), scale from the slot group `amax`, as the GEMM loops issue it (8-value and
 * 4-value pinned instruction sequences) and as the prologues / packing kernels compute it; each output element = h | l << 16. */
int pfst_f16x3_split_probe(const float* x, long long n, const float* amax, unsigned* pieces_loop, unsigned* pieces_loop4,
                           unsigned* pieces_plain, pfst_stream_t stream);
int pfst_wino_gemm_f16x3(const float* V, const void* U4, const float* u_amax, const float* v_amax, float* Mbuf, int N, int K,
                         int M, int T, int m, int v_packed, pfst_stream_t stream);
int pfst_conv_wgrad_f16x3(const float* x, long long x_bs, const float* dy, long long dy_bs, float* dw, int N, int Cin, int Cout,
                          int HW, const float* x_amax, const float* dy_amax, const float* bnl, pfst_stream_t stream);
/* bnl != NULL: x is the PRE-normalisation tensor the forward launch read with its own bnl (pfst_conv_igemm_f16x3): rows normalised on load */
/* the same for the layers that kernel does not take -- stride-1 'same' 3x3 (pad == dil <= 8, W % 16 == 0) and 1x1 with <= 64 output channels
 * (the stems, layer1: resnet.py:593-624,169-209) -- on the K-quad kernel with both operands split as they are staged (csrc/conv_wgrad_q.hip) */
int pfst_conv_wgrad_f16x3_q(const float* x, long long x_bs, const float* dy, long long dy_bs, float* dw, int N, int Cin, int H, int W,
                            int Cout, int ksize, int dil, const float* x_amax, const float* dy_amax, pfst_stream_t stream);

/* ---- depthwise 3x3 convolution, stride 1, pad = dil (mmcv DepthwiseSeparableConvModule,
 * sep_aspp_head.py:17-26,63-77).  flip != 0 mirrors the taps (= data gradient). */
int pfst_dwconv3x3(const float* x, long long x_bs, const float* w, float* y, long long y_bs,
                   int N, int C, int H, int W, int dil, int flip, int accumulate, float* stats, int stats_minmax,
                   const float* bn_on_load_coef, pfst_stream_t stream);
/* stats_minmax != 0 (with stats): `stats` has room for twice the partials and also receives the per-channel (minimum, maximum) partials
 * behind the sums (as pfst_conv_igemm_f16x3's): the pointwise layer of the DepthwiseSeparableConvModule then normalises this output as it
 * loads it (pfst_conv_igemm_f16x3 bnl) from the predicted maximum */
/* bn_on_load_coef: NULL, or coef[C][4] = (mean, invstd, sc, sh) of the conv -> BN -> ReLU layer that feeds this depthwise layer: x is then that
 * layer's PRE-normalisation output, normalised + rectified while it is staged (forward only; the normalised tensor is never written) */
/* stats != NULL: per-channel partial (sum, sum of squares) of the outputs, stats[C][N * pfst_dwconv_stats_slots(H, W, dil)][2],
 * for pfst_bn_finalize_partials (as the `stats` argument of pfst_conv_igemm) */
int pfst_dwconv_stats_slots(int H, int W, int dil);
int pfst_dwconv3x3_wgrad(const float* x, long long x_bs, const float* dy, long long dy_bs, float* dw,
                         int N, int C, int H, int W, int dil, pfst_stream_t stream);
/* ns <= 3 depthwise branches that read the SAME input (the ASPP head's atrous branches, sep_aspp_head.py:63-77: dilations 12 / 24 / 36 on one
 * 2048-channel map), whole planes (H * W <= 16384, W % 4 == 0), dilations % 4 == 0, 16-byte aligned dense planes.
 *   fwd: y[i] = dwconv(x, w[i], dils[i]) (+ BatchNorm partials stats[i][C][N][2] or NULL) with every input plane staged once;
 *        plane_mean != NULL: also [N][C] means of x's planes = nn.AdaptiveAvgPool2d(1) of the image-pool branch (aspp_head.py:69-77),
 *        as pfst_global_avgpool forms them (fp64 sum);
 *   bwd: dw[i] += weight gradients, dx (+)= sum_i conv(dy[i], mirrored w[i]) [+ plane_mean_grad[n][c] / (H W), the pool's adjoint]:
 *        x read once, every dy[i] once, dx written once. */
int pfst_dwconv3x3_multi_ok(int H, int W, int ns, const int* dils);
int pfst_dwconv3x3_multi_fwd(const float* x, long long x_bs, int ns, const float* const* w, float* const* y, const long long* y_bs,
                             float* const* stats, int stats_minmax, const int* dils, float* plane_mean, int N, int C, int H, int W,
                             pfst_stream_t stream);
int pfst_dwconv3x3_multi_bwd(const float* x, long long x_bs, int ns, const float* const* w, const float* const* dy,
                             const long long* dy_bs, float* const* dw, const int* dils, const float* plane_mean_grad, float* dx,
                             long long dx_bs, int accumulate, const float* const* bn_pre, const pfst_bn_bwd_rec_t* const* bn_rec,
                             int N, int C, int H, int W, pfst_stream_t stream);
/* bn_pre / bn_rec != NULL (all branches or none): as in pfst_dwconv3x3_bwd, per branch (bn_pre[i] has dy[i]'s batch stride) */
/* both gradients of the depthwise convolution in one pass over dy and the forward input x (autograd of the same F.conv2d(groups = C)):
 * dx (+)= conv(dy, mirrored w), dw += sum dy * shifted x -- 3 N of HBM traffic instead of the two kernels' 4 N */
int pfst_dwconv3x3_bwd(const float* dy, long long dy_bs, const float* x, long long x_bs, const float* w, float* dx, long long dx_bs,
                       float* dw, int N, int C, int H, int W, int dil, int accumulate, const float* bn_on_load_coef,
                       const float* bn_pre, long long bn_pre_bs, const pfst_bn_bwd_rec_t* bn_rec, pfst_stream_t stream);
/* (bn_on_load_coef as in pfst_dwconv3x3: x holds the pre-normalisation tensor, its quads are normalised as they are loaded)
 * bn_rec != NULL: dy is the gradient of the BatchNorm + ReLU OUTPUT of this depthwise layer, bn_pre the layer's pre-normalisation tensor
 * (the depthwise convolution's own output): dL/dpre is formed while the rows are staged (pfst_bn_backward_sums ran before) */

/* ---- BatchNorm2d, training mode (nn.BatchNorm2d inside mmcv ConvModule; eps 1e-5, momentum .1) */
/* batch mean / 1/sqrt(biased var + eps) per channel; updates running stats (unbiased var) when
 * running_mean != NULL.  ws: >= 2*C doubles of scratch. */
int pfst_bn_stats(const float* x, long long x_bs, int N, int C, int HW, float* mean, float* invstd,
                  float* running_mean, float* running_var, float momentum, float eps, double* ws,
                  const float* gamma, const float* beta, float* coef, pfst_stream_t stream);
/* coef != NULL (needs gamma, beta): also writes coef[C][4] = (mean, invstd, sc = invstd*gamma, sh = beta - mean*sc), the per-channel
 * record the fused BatchNorm-backward sums of a data-gradient launch read (pfst_bnb_fuse_t) */
/* the same from the conv epilogue's partials[C][T][2] (count = N*H*W elements per channel) */
int pfst_bn_finalize_partials(const float* partials, int T, int C, double count, float* mean, float* invstd,
                              float* running_mean, float* running_var, float momentum, float eps,
                              const float* gamma, const float* beta, float* coef, const float* minmax, int relu, float* y_amax,
                              pfst_stream_t stream);
/* minmax != NULL (with gamma, beta, y_amax): the producer's [C][T][2] (minimum, maximum) partials (pfst_conv_igemm_f16x3 stats_minmax).  BatchNorm
 * [+ ReLU] is monotone per channel, so the extreme outputs are the images of the extreme inputs: the slot group y_amax receives
 * max |[relu](bn(x))| -- exactly what pfst_bn_apply(..., y_amax) would publish -- without the normalised tensor being written. */
/* y = [relu]( (x-mean)*invstd*gamma + beta [+ residual] ).  relu_mask != NULL (needs relu, HW % 256 == 0, 16-byte aligned planes):
 * also writes the ReLU gate as a bitmask of N*C*HW/64 words for pfst_bn_backward -- the backward of a residual layer then reads
 * 1 bit per element instead of the fp32 output y in both of its passes. */
int pfst_bn_apply(const float* x, long long x_bs, const float* residual, long long res_bs, float* y, long long y_bs,
                  const float* mean, const float* invstd, const float* gamma, const float* beta,
                  int N, int C, int HW, int relu, unsigned long long* relu_mask, float* y_amax, const float* post_scale,
                  const float* residual_coef, pfst_stream_t stream);
/* residual_coef: NULL, or coef [C][4] = (mean, invstd, sc, sh) of the downsample conv -> BN layer (no ReLU, resnet.py:298-303) whose
 * PRE-normalisation output `residual` then is: normalised as it is loaded, the normalised identity branch is never written */
/* y_amax: NULL, or the slot group (1024 floats, zeroed) that receives max |y|: the scale of an f16x3 GEMM reading y.
 * post_scale: NULL, or [N][C] factors y is multiplied by after the ReLU -- nn.Dropout2d's keep / (1 - p) mask of the layer feeding conv_seg
 * (decode_head.py:103-107,242-247) folded into this pass (no residual then) */
/* backward of the above: dz = dy * (y > 0 if relu); dres (+)= dz; dgamma += sum dz*xhat; dbeta += sum dz;
 * dx = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)).  ws: >= 2*C doubles.  The ReLU gate comes from relu_mask (as written by
 * pfst_bn_apply), else from the saved output y; if both are NULL and beta != NULL (layer without residual) it is recomputed from
 * x exactly as bn_apply did. */
int pfst_bn_backward(const float* dy, long long dy_bs, const float* y, long long y_bs, const float* x, long long x_bs,
                     const float* mean, const float* invstd, const float* gamma, const float* beta,
                     float* dx, long long dx_bs, float* dres, long long dres_bs, int dres_accumulate,
                     float* dgamma, float* dbeta, int N, int C, int HW, int relu, const unsigned long long* relu_mask,
                     double* ws, const float* bwd_partials, int bwd_slots, float* dx_amax, const float* post_scale, pfst_stream_t stream);
/* post_scale: the factors pfst_bn_apply folded into y: dy is the gradient of the SCALED output, dz = dy * post_scale[n][c] * gate (no dres, no
 * bwd_partials then) */
/* Two BatchNorm layers behind ONE gated gradient -- the last BN of a stage's first Bottleneck (a: bn3) and its downsample branch's BN (b),
 * out = relu(bn3(conv3(.)) + bn_d(conv_d(x))), resnet.py:298-307: both receive g = (relu_mask bit ? dy : 0).  One reduction pass over
 * (dy, xa, xb) and one apply pass writing dxa and dxb replace the two pfst_bn_backward calls (8 N instead of 10 N of traffic); the results
 * are those of pfst_bn_backward(relu = 1, relu_mask) per layer.  bwd_partials_a: layer a's sums from the launch that wrote dy (as in
 * pfst_bn_backward).  ws_a, ws_b: >= 2*C doubles each.  Needs HW % 256 == 0 and 16-byte aligned planes; PFST_ERR_UNSUPPORTED otherwise and in
 * deterministic mode (the caller then runs the layers one by one). */
int pfst_bn_backward_dual(const float* dy, long long dy_bs, const unsigned long long* relu_mask,
                          const float* xa, long long xa_bs, const float* mean_a, const float* invstd_a, const float* gamma_a,
                          float* dxa, long long dxa_bs, float* dgamma_a, float* dbeta_a, double* ws_a,
                          const float* bwd_partials_a, int bwd_slots_a, float* dxa_amax,
                          const float* xb, long long xb_bs, const float* mean_b, const float* invstd_b, const float* gamma_b,
                          float* dxb, long long dxb_bs, float* dgamma_b, float* dbeta_b, double* ws_b, float* dxb_amax,
                          int N, int C, int HW, pfst_stream_t stream);
/* The FIRST half of pfst_bn_backward alone, for a conv -> BN -> ReLU layer without residual whose convolution is depthwise: the two sums
 * (from bwd_partials, else by the reduction pass), dgamma += sum dz*xhat, dbeta += sum dz, and rec[C] for the consumer that applies the
 * second half while it loads dy and x (no dx is written here: 3 N of traffic less per layer).  ws: >= 2*C doubles. */
/* out (+)= (bit ? g : 0) with the ReLU bitmask pfst_bn_apply wrote (HW % 256 == 0): the gated identity-branch gradient of a residual block
 * written out by itself -- the product folds it into conv1's data-gradient epilogue (pfst_conv_igemm_f16x3 gate_dy); this is the fallback
 * when another reader needs the gradient first */
int pfst_relu_gate(const float* g, long long g_bs, const unsigned long long* relu_mask, float* out, long long out_bs, int N, int C, int HW,
                   int accumulate, pfst_stream_t stream);
int pfst_bn_backward_sums(const float* dy, long long dy_bs, const float* x, long long x_bs, const float* mean, const float* invstd,
                          const float* gamma, const float* beta, float* dgamma, float* dbeta, int N, int C, int HW,
                          double* ws, const float* bwd_partials, int bwd_slots, pfst_bn_bwd_rec_t* rec, pfst_stream_t stream);
/* bwd_partials != NULL: (sum dz, sum dz*x) were already produced by the launch that wrote dy (pfst_bnb_fuse_t, [C][bwd_slots][2]); the
 * reduction pass over dy and x is skipped and only the partials are summed (fp64). */

/* ---- pooling / resize ---------------------------------------------------------------------- */
/* nn.MaxPool2d(3, 2, 1) (resnet.py:638); idx holds the winning tap 0..8 */
int pfst_maxpool3x3s2(const float* x, float* y, unsigned char* idx, int NC, int H, int W, int Ho, int Wo, const float* bn_on_load_coef, int C,
                      float* y_amax, pfst_stream_t stream);
/* bn_on_load_coef: NULL, or coef[C][4] = (mean, invstd, sc, sh) (pfst_bn_finalize_partials / pfst_bn_stats) of the conv -> BN -> ReLU layer
 * in front of the pool: x is then that layer's PRE-normalisation output (NC = N * C planes) and y = maxpool(relu(x * sc + sh)) -- the
 * normalised tensor is never written (its only consumer is the pool) */
int pfst_maxpool3x3s2_bwd(const float* dy, const unsigned char* idx, float* dx, int NC, int H, int W, int Ho, int Wo, pfst_stream_t stream);
/* F.interpolate(mode='bilinear', align_corners=False) (ops/wrappers.py:27) and its adjoint */
int pfst_resize_bilinear(const float* x, long long x_bs, float* y, long long y_bs, int N, int C, int Hi, int Wi, int Ho, int Wo, pfst_stream_t stream);
int pfst_resize_bilinear_bwd(const float* dy, long long dy_bs, float* dx, long long dx_bs, int N, int C, int Hi, int Wi, int Ho, int Wo, int accumulate, pfst_stream_t stream);
/* nn.AdaptiveAvgPool2d(1) (aspp_head.py:69-77): y[n][c] = mean_hw x */
int pfst_global_avgpool(const float* x, long long x_bs, float* y, int N, int C, int HW, pfst_stream_t stream);
/* dx[n][c][hw] (+)= dy[n][c] * scale */
int pfst_broadcast_hw(const float* v, float* y, long long y_bs, int N, int C, int HW, float scale, int accumulate, float* y_amax,
                      pfst_stream_t stream);
/* y_amax (both): NULL, or the slot group (1024 floats, zeroed) that receives max |y| of what the launch writes (f16x3 scale, csrc/amax.h) */
/* v[n][c] = sum_hw dy[n][c][hw]  (adjoint of the 1x1 -> HxW bilinear broadcast) */
int pfst_reduce_hw(const float* dy, long long dy_bs, float* v, int N, int C, int HW, pfst_stream_t stream);
/* F.interpolate(mode='nearest') by an integer factor on [NC][h][w] maps and its adjoint (pfgst_loss.py:57-58) */
int pfst_upsample_nearest(const float* x, float* y, int NC, int h, int w, int factor, pfst_stream_t stream);
int pfst_upsample_nearest_bwd(const float* dy, float* dx, int NC, int h, int w, int factor, pfst_stream_t stream);
/* nn.Dropout2d (decode_head.py:103-107): y = x * mask[n][c] */
int pfst_channel_scale(const float* x, const float* mask, float* y, int N, int C, int HW, pfst_stream_t stream);
/* ---- test-time inference beyond the whole-image arg-max (rsiseg/models/segmentors/encoder_decoder.py): sliding windows, flips, averaging */
/* F.softmax(seg_logit, dim=1) (:311) with torch's arithmetic: y[n][c][p] = exp(x - max_c) / sum_c exp(x - max_c) */
int pfst_softmax_nchw(const float* x, long long x_bs, float* y, long long y_bs, int N, int C, int HW, pfst_stream_t stream);
/* slide_inference (:220-263): preds[:, :, y1:y1+Hc, x1:x1+Wc] += crop (F.pad + add, :246-248), count[:, y1:.., x1:..] += 1 (:250) */
int pfst_window_accumulate(const float* crop, float* preds, float* count, int N, int C, int Hc, int Wc, int H, int W, int y1, int x1,
                           pfst_stream_t stream);
int pfst_window_normalize(float* preds, const float* count, int N, int C, int HW, pfst_stream_t stream);      /* preds / count_mat (:256) */
/* seg_logit.argmax(dim=1) (:332, :368): first maximal class per pixel, uint8 */
int pfst_argmax_nchw(const float* x, long long x_bs, unsigned char* label_u8, int N, int C, int HW, pfst_stream_t stream);
/* output.flip(dims=(3,)) / (2,) (:316-325) on `planes` H x W maps, out of place */
int pfst_flip_planes(const float* x, float* y, int planes, int H, int W, int horizontal, int vertical, pfst_stream_t stream);
int pfst_div_scalar(float* x, long long n, float divisor, pfst_stream_t stream);                              /* seg_logit /= len(imgs) (:367) */

/* ---- fused bilinear-upsample + softmax cross-entropy + accuracy (decode_head.py:249-283,
 * cross_entropy_loss.py:45-65, accuracy.py:6-61).  logits are [N][C][h][w]; labels/weights [N][H][W].
 * acc[0] += sum_i w_i*cw[y_i]*nll_i (0 at ignore), acc[1] += #correct, acc[2] += #non-ignored, acc[3] += #labels that are neither
 * in [0, C) nor ignore_index (F.cross_entropy raises on those; pfst_ce_finalize reports the count so the caller can).  acc: 4 doubles.
 * lse[N][H][W] (log-sum-exp of the upsampled logits) is saved for the backward. */
int pfst_ce_upsample_fwd(const float* logits, int N, int C, int h, int w, const unsigned char* label, const float* pix_weight,
                         const float* class_weight, int H, int W, int ignore_index, float* lse, double* acc, pfst_stream_t stream);
/* dlogits[n][c][ly][lx] (+)= scale * sum_p bilin(p->l) * w_p*cw[y_p] * (softmax_c(p) - [c==y_p]) */
int pfst_ce_upsample_bwd(const float* logits, int N, int C, int h, int w, const unsigned char* label, const float* pix_weight,
                         const float* class_weight, int H, int W, int ignore_index, const float* lse, float scale,
                         float* dlogits, int accumulate, pfst_stream_t stream);

/* ---- pseudo labels (pfgst.py:259-268 + encoder_decoder.py:77-81): bilinear upsample of the teacher
 * logits, softmax (exp(z - max) / sum as torch computes it), then torch.max over the PROBABILITIES: the first class whose
 * rounded probability is maximal (not the arg-max of the logits: distinct logits may tie after rounding); prob >= threshold
 * counted into count[0].  conf_mask (0/1 per pixel) and max_prob (the softmax value itself) may be NULL. */
int pfst_pseudo_label(const float* logits, int N, int C, int h, int w, int H, int W, float threshold,
                      long long* label_i64, unsigned char* label_u8, unsigned long long* count, float* conf_mask, float* max_prob,
                      pfst_stream_t stream);

/* ---- evaluation: intersect_and_union (rsiseg/core/evaluation/metrics.py:26-86).  hist[3*C] (+)= per-class
 * #intersect, #pred, #label over pixels whose label != ignore_index (caller zeroes hist once per evaluation) */
int pfst_confusion_hist(const unsigned char* pred, const unsigned char* label, long long n, int C, int ignore_index,
                        unsigned long long* hist, pfst_stream_t stream);

/* ---- class mix (dacs_transforms.py:110-144, pfgst.py:281-300) ------------------------------- */
/* presence[v] = 1 if label value v occurs (torch.unique over the batch) */
int pfst_label_presence(const unsigned char* label, long long n, int* presence256, pfst_stream_t stream);
/* mask[n][p] = 1 if gt[n][p] is one of classes[n][0..K) (entries < 0 are padding) */
int pfst_class_mask(const unsigned char* gt, const int* classes, int K, unsigned char* mask, int N, long long HW, pfst_stream_t stream);
/* mixed = M*src + (1-M)*trg for image, label and pixel weight; the target weight is the scalar
 * q = conf_count[0] / (N*HW) (thre_type 'all') or, if trg_weight != NULL, the per-pixel map (thre_type 'part',
 * conf_mask of pfst_pseudo_label).  mixed_lbl_i64 may be NULL. */
int pfst_class_mix(const float* img, const float* trg_img, const unsigned char* gt, const unsigned char* pseudo,
                   const unsigned char* mask, const unsigned long long* conf_count, const float* trg_weight, float* mixed_img,
                   unsigned char* mixed_lbl, long long* mixed_lbl_i64, float* mixed_w, int N, int Cimg, long long HW, pfst_stream_t stream);

/* ---- DACS strong augmentation of the mixed image (dacs_transforms.py:44-107; kornia arithmetic restated,
 * parity unpinned).  params[n][8] = brightness, contrast, saturation, hue(rad), order[4] (0..3 = b,c,s,h) */
int pfst_color_jitter(float* img, const float* params, const float* mean3, const float* std3, int N, long long HW,
                      int denorm, pfst_stream_t stream);
/* separable gaussian blur, reflect border; taps_*[n][K*] per image; taps farther than `reach` from the centre are skipped */
int pfst_gaussian_blur(const float* x, float* tmp, float* y, const float* taps_y, int Ky, const float* taps_x, int Kx,
                       int N, int C, int H, int W, int reach, pfst_stream_t stream);

/* ---- PFGSTLoss (pfgst_loss.py:44-234), kernel 3x3 ------------------------------------------------ */
/* sim_type 0: sim[n][k][y][x] = cos(f[n][:,y,x], f[n][:,y+dy_k,x+dx_k]) (0 outside); norm[n][y][x] = |f|
 * sim_type 1: sim = exp(-|f(neighbour) - f(centre)|^2 / sigma^2), the zero padding counting as f = 0 (pfgst_loss.py:199-201) */
int pfst_sim_map(const float* feat, int N, int C, int H, int W, int dil, int sim_type, float sigma, float* sim, float* norm,
                 pfst_stream_t stream);
/* d feat (+)= adjoint of pfst_sim_map for upstream gradient gsim[n][9][H][W] */
int pfst_sim_map_bwd(const float* feat, const float* sim, const float* norm, const float* gsim, int N, int C, int H, int W, int dil,
                     int sim_type, float sigma, float* dfeat, int accumulate, float* coef_ws, pfst_stream_t stream);
/* coef_ws: 10*N*H*W floats of scratch (per-pixel stencil coefficients) for the strip kernel of the cosine path; NULL selects the
 * generic kernel */
/* source statistics: sets (neighbour label == / != centre label, centre != 255) of src sims.
 * gt is full resolution [N][Hg][Wg] uint8, nearest-sampled to HxW.
 * loss_type 0 (mean_std): stats[0..5] = n_pos, sum_pos, sumsq_pos, n_neg, sum_neg, sumsq_neg
 * loss_type 1 / 2 (margin / margin2, pfgst_loss.py:116-131): stats[1] = sum relu(margin_pos - s)^e over positive pairs,
 *   stats[4] = sum relu(s - margin_neg)^e over negative pairs, e = loss_type */
int pfst_src_sim_stats(const float* sim, const unsigned char* gt, int N, int H, int W, int Hg, int Wg, int dil, int loss_type,
                       float margin_pos, float margin_neg, double* stats, const void* select, pfst_stream_t stream);
/* loss_type 0: losses[0..3] = -w*mean_pos, w*mean_neg, w*std_pos, w*std_neg; 1 / 2: losses[0..1] = w_pos*mean hinge_pos,
 * w_neg*mean hinge_neg (losses[2..3] = 0); gsim = d(sum of the losses)/d sim */
int pfst_src_sim_grad(const float* sim, const unsigned char* gt, int N, int H, int W, int Hg, int Wg, int dil, int loss_type,
                      float margin_pos, float margin_neg, const double* stats,
                      float w_pos, float w_neg, float w_pos_std, float w_neg_std, float* gsim, float* losses, const void* select, pfst_stream_t stream);
/* src_perc (pfgst_loss.py:98-102: only the int(n * src_perc) smallest positive / largest negative similarities enter the source
 * losses): pfst_src_sim_select finds, with an exact radix select and no sort, the threshold value of each set and the share of its
 * ties that lies inside the sorted prefix; pass the filled `select` buffer (pfst_src_sim_select_bytes() bytes, 8-byte aligned) to
 * pfst_src_sim_stats / pfst_src_sim_grad, or NULL for all pairs. */
int pfst_src_sim_select_bytes(void);
int pfst_src_sim_select(const float* sim, const unsigned char* gt, int N, int H, int W, int Hg, int Wg, int dil, double src_perc,
                        void* select, pfst_stream_t stream);
/* prob[n][c][y][x] = softmax_c(logits[n][c][y*ds][x*ds]) (nearest down-scaling by ds) */
int pfst_softmax_down(const float* logits, int N, int C, int h, int w, int ds, float* prob, int H, int W, pfst_stream_t stream);
/* valid[n][y][x] = (gt != 255) && all 9 dilated neighbours un-mixed; count[0] = #valid */
int pfst_trg_valid_mask(const unsigned char* gt, const unsigned char* mix_mask, int N, int H, int W, int Hg, int Wg, int dil,
                        unsigned char* valid, unsigned char* all9, unsigned long long* count, pfst_stream_t stream);
/* top-k target losses (top_k = 0: all nine pairs, the reference's top_k=None); acc[0] += sum loc_pos, acc[1] += sum loc_neg
 * over valid pixels; gP[n][9][y][x] = d(w_pos*mean loc_pos + w_neg*mean loc_neg)/d cross_prob (0 when count <= 1) */
int pfst_sim_topk_loss(const float* ema_sim, const float* prob, const unsigned char* valid, const unsigned long long* count,
                       int N, int C, int H, int W, int dil, int top_k, float w_pos, float w_neg, float* gP, double* acc, float* g_sim, pfst_stream_t stream);
/* g_sim != NULL: also d(losses)/d ema_sim [N][9][H][W] -- needed only when the similarity has trainable inputs (proj_net) */
/* d logits[n][c][y*ds][x*ds] += softmax-backward( sum_k gP[k] * prob_c(neighbour k) ); unfold_grad != 0 (detach_unfold=False)
 * adds the gradient through the unfolded factor: coefficient gP[k][r] + gP[8-k][r+D_k] */
int pfst_cross_prob_bwd(const float* prob, const float* gP, int N, int C, int H, int W, int dil, int ds, int unfold_grad,
                        float* dlogits, int h, int w, pfst_stream_t stream);
/* out[0] = w_pos*acc[0]/((top_k+1)*count), out[1] = w_neg*acc[1]/(top_k*count); top_k = 0: both /(9*count)  (zeros when count <= 1) */
int pfst_sim_loss_finalize(const double* acc, const unsigned long long* count, int top_k, float w_pos, float w_neg, float* out, pfst_stream_t stream);

/* ---- EMA teacher + AdamW on flat parameter arenas (pfgst.py:105-127, torch.optim.AdamW) ------ */
int pfst_ema_update(float* teacher, const float* student, long long n, float alpha, pfst_stream_t stream);
int pfst_adamw_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, int step, float grad_scale, pfst_stream_t stream);

/* ---- loss bookkeeping (base.py:177-222) ------------------------------------------------------- */
/* out[0] = loss_weight*acc[0]/numel ; out[1] = 100*(acc[1]+eps)/(acc[2]+eps) ; out[2] = acc[3] (count of invalid labels) */
int pfst_ce_finalize(const double* acc, double numel, float loss_weight, float* out, pfst_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
