/*
 * ir2rgb_hip.h -- C ABI of libir2rgb_hip.so (MI355X / gfx950).
 *
 * This is the drop-in boundary for the ir2rgb hot path: what the reference binds through
 * three pybind11 extension modules (correlation_cuda / resample2d_cuda / channelnorm_cuda)
 * and what its models/networks.py obtains from cuDNN, expressed as plain C entry points
 * that take raw device pointers, explicit sizes and a HIP stream.  No torch types cross it.
 *
 * Conventions
 *   - every pointer is a device pointer unless the name ends in `_host`;
 *   - tensors are dense, in the layout named in the comment of each function;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all work is
 *     enqueued asynchronously on it, nothing synchronises the host;
 *   - return value: 0 on success, a positive hipError_t if a launch failed, or a negative
 *     IR2RGB_E* code for argument errors detected on the host.  Nothing throws.
 *   - "reference" paths are relative to /root/reference.
 */
#ifndef IR2RGB_HIP_H
#define IR2RGB_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IR2RGB_OK 0
#define IR2RGB_EINVAL (-1)  /* bad size / parameter combination            */
#define IR2RGB_ENOSUP (-2)  /* valid in the reference, not implemented here */
#define IR2RGB_EALIGN (-3)  /* pointer or pitch not aligned as required     */

/* element types of activation / weight buffers */
#define IR2RGB_F32 0
#define IR2RGB_BF16 1
#define IR2RGB_F16 2

/* library / build identification: "ir2rgb_hip <version> gfx950" */
const char *ir2rgb_version(void);

/* ------------------------------------------------------------------------------------------
 * FlowNet2 operators (fp32, NCHW)
 * ------------------------------------------------------------------------------------------ */

/* Output geometry of the cost volume.
 * Replaces the sizing arithmetic in
 *   models/flownet2_pytorch/networks/correlation_package/correlation_cuda.cc:25-42. */
int ir2rgb_correlation_out_shape(int C, int H, int W, int pad_size, int kernel_size, int max_displacement,
                                 int stride1, int stride2, int *outC, int *outH, int *outW);

/* Cost volume forward.  in1,in2 [N,C,H,W] -> out [N,outC,outH,outW]; out is fully written
 * (no pre-zeroing needed).  Unlike the reference no padded channels-last scratch copies
 * (rInput1/rInput2) are needed.
 * Replaces correlation_cuda.forward:
 *   correlation_package/correlation_cuda.cc:10-87 and
 *   correlation_package/correlation_cuda_kernel.cu:46-147, :336-427.
 * Status: operator API (the reference's fp32 NCHW boundary, BASELINE config 3: 65 us at [1,256,64,128] = 0.48 TB/s, fp32 VALU
 * bound).  The FlowNet2 pipeline of this package does not call it: FlowNetC's cost volume is taken from the half NHWC conv3
 * outputs by ir2rgb_correlation_nhwc_half below (18.6 us); both are parity-tested against the same oracle. */
int ir2rgb_correlation_fwd(const float *in1, const float *in2, float *out, int N, int C, int H, int W,
                           int pad_size, int kernel_size, int max_displacement, int stride1, int stride2,
                           void *stream);

/* Cost volume backward.  gout [N,outC,outH,outW] -> gin1, gin2 [N,C,H,W] (fully written).
 * stride1 must be 1 (the reference's launch geometry writes out of bounds otherwise).
 * Replaces correlation_cuda.backward:
 *   correlation_package/correlation_cuda.cc:89-167 and
 *   correlation_package/correlation_cuda_kernel.cu:150-334, :430-564. */
int ir2rgb_correlation_bwd(const float *in1, const float *in2, const float *gout, float *gin1, float *gin2,
                           int N, int C, int H, int W, int pad_size, int kernel_size, int max_displacement,
                           int stride1, int stride2, void *stream);

/* The cost volume of FlowNetC as it sits in the FlowNet2 pipeline (reference FlowNetC.py:27, :101-103: Correlation(pad 20,
 * kernel 1, max displacement 20, stride1 1, stride2 2) followed by LeakyReLU(0.1) and the concatenation with conv_redir):
 * a, b = half-precision NHWC feature maps [N][H][W][lda / ldb], channels [offa, offa+C) / [offb, offb+C) -- the outputs of
 * conv3 as ir2rgb_conv2d_fwd leaves them.  out_mode 0: out = fp32 [N][441][H][W], the layout of ir2rgb_correlation_fwd;
 * out_mode 1: out = half NHWC [N][H][W][ldo], channels [offo, offo+441) = LeakyReLU_slope(correlation) (slope 1: none).
 * Products of half inputs are exact in fp32; the result differs from ir2rgb_correlation_fwd on the same (half-rounded)
 * inputs only by the order of the fp32 additions.  Banded MFMA products (correlation_mfma.hip).  C in {128, 256},
 * W <= 128, otherwise IR2RGB_ENOSUP (use ir2rgb_correlation_fwd). */
int ir2rgb_correlation_nhwc_half(const void *a, int lda, int offa, const void *b, int ldb, int offb, void *out,
                                 int out_mode, int ldo, int offo, float slope, int N, int C, int H, int W, int dtype,
                                 void *stream);

/* Flow warp forward.  img [N,C,H,W], flow [N,2,H,W] (pixels; channel 0 = dx, 1 = dy)
 * -> out [N,C,H,W].  kernel_size must be 1 (the only value the reference passes; larger
 * values make its kernel read out of bounds).
 * Replaces resample2d_cuda.forward:
 *   resample2d_package/resample2d_cuda.cc:6-13, resample2d_kernel.cu:15-64, :192-232. */
int ir2rgb_resample2d_fwd(const float *img, const float *flow, float *out, int N, int C, int H, int W,
                          int kernel_size, void *stream);

/* Flow warp backward.  gout [N,C,H,W] -> gimg [N,C,H,W] (zeroed here, then scattered with
 * float atomics), gflow [N,2,H,W].  Keeps the reference's truncation-vs-floor quirk.
 * Replaces resample2d_cuda.backward:
 *   resample2d_package/resample2d_cuda.cc:15-24, resample2d_kernel.cu:67-190, :234-310. */
int ir2rgb_resample2d_bwd(const float *img, const float *flow, const float *gout, float *gimg, float *gflow,
                          int N, int C, int H, int W, int kernel_size, void *stream);

/* Channel L2 norm forward.  in [N,C,H,W] -> out [N,1,H,W].  norm_deg is accepted and
 * ignored exactly as in the reference.
 * Replaces channelnorm_cuda.forward:
 *   channelnorm_package/channelnorm_cuda.cc:6-13, channelnorm_kernel.cu:18-60, :98-129. */
int ir2rgb_channelnorm_fwd(const float *in, float *out, int N, int C, int H, int W, int norm_deg, void *stream);

/* Channel L2 norm backward.  gin = gout * in / (out + 1e-9).
 * Replaces channelnorm_cuda.backward:
 *   channelnorm_package/channelnorm_cuda.cc:16-25, channelnorm_kernel.cu:63-96, :131-177. */
int ir2rgb_channelnorm_bwd(const float *in, const float *out, const float *gout, float *gin, int N, int C,
                           int H, int W, int norm_deg, void *stream);

/* Fused FlowNet2 step: warped = resample2d(img2, flow); diff = img1 - warped;
 * norm = channelnorm(diff).  Any of warped / diff / norm may be NULL to skip that output.
 * Fuses the three-launch sequence at models/flownet2_pytorch/models.py:109-111 (and
 * :121-123, :133-137, :146-150) and the confidence test of models/flownet.py:50,56-57
 * (conf = norm^2 < 0.02 is left to the caller). */
int ir2rgb_warp_diff_norm_fwd(const float *img1, const float *img2, const float *flow, float *warped,
                              float *diff, float *norm, int N, int C, int H, int W, void *stream);

/* ------------------------------------------------------------------------------------------
 * Generator / discriminator convolutions (half precision NHWC, fp32 accumulate, MFMA)
 *
 * These replace what reference models/networks.py gets from torch.nn.Conv2d /
 * ConvTranspose2d (+ ReflectionPad2d) -> cuDNN for the dense layers of
 * CompositeGeneratorModule (:141-171), CompositeLocalGeneratorModule (:253-271),
 * ResnetBlock (:556-580) and NLayerDiscriminator (:678-699).
 * ------------------------------------------------------------------------------------------ */
typedef struct ir2rgb_conv_desc {
    int N, Hin, Win, Cin;   /* input  [N,Hin,Win,Cin]  NHWC, Cin % 64 == 0 */
    int Hout, Wout, Cout;   /* output [N,Hout,Wout,Cout] NHWC              */
    int kh, kw, stride_h, stride_w, pad_h, pad_w;
    int pad_mode;           /* 0: zero padding, 1: reflection padding (nn.ReflectionPad2d), 2: ADJOINT of reflection
                             * padding: the call is the data gradient of a reflection-padded 3x3 / stride-1 / pad-1
                             * convolution (x = its output gradient, wpacked = its adjoint-packed weights, y = the
                             * input gradient, same H x W); only where ir2rgb_conv2d_kernel_name() says
                             * "conv3x3_patch_kernel", IR2RGB_ENOSUP otherwise (then: zero padding 2 on the padded
                             * grid + ir2rgb_fold_reflect) */
    int transposed;         /* 0: Conv2d, 1: ConvTranspose2d (stride 1 or 2 per axis; Hout/Wout carry output_padding) */
    int dtype;              /* IR2RGB_BF16 or IR2RGB_F16: activations and packed weights */
    int act;                /* fused after bias: 0 none, 1 LeakyReLU(0.2), 2 LeakyReLU(0.1), 3 ReLU */
    int out_f32;            /* 0: y is half NHWC, 1: y is fp32 NHWC (head convolutions) */
    /* channel-slice views (0 = dense): x holds ldx channels per pixel of which [ci_off, ci_off+Cin) are
     * read; y holds ldy channels per pixel of which [co_off, co_off+Cout) are written.  Lets producers
     * write straight into the concatenation buffers of the FlowNet2 decoders (torch.cat, FlowNetS.py etc.) */
    int ldx, ci_off, ldy, co_off;
    /* 1: the rows of the statistics buffer are cut per sample -- rows [n * R / N, (n + 1) * R / N) hold sample n only
     * (R = ir2rgb_conv2d_stats_rows) -- so that BatchNorm can treat groups of samples as separate forwards (the
     * discriminators' real / generated batches, reference discriminator.py:154-166, run as ONE convolution).  Costs
     * partly filled pixel tiles at every sample's end; not for transposed convolutions.  0: tiles run over the batch. */
    int stats_per_sample;
} ir2rgb_conv_desc;

/* Number of half elements of the packed weight buffer for this convolution (< 0: error). */
long ir2rgb_conv2d_packed_weight_elems(const ir2rgb_conv_desc *d);

/* Rows of the per-tile BatchNorm statistics buffer written by ir2rgb_conv2d_fwd:
 * stats_partial is [rows][2][Cout] fp32 (sum, sum of squares over the tile's pixels). */
int ir2rgb_conv2d_stats_rows(const ir2rgb_conv_desc *d);

/* Repack torch-layout fp32 weights (Conv2d [Cout,Cin,kh,kw]; ConvTranspose2d [Cin,Cout,kh,kw])
 * into the K-contiguous half-precision layout the MFMA kernel streams. */
int ir2rgb_conv2d_pack_weight(const ir2rgb_conv_desc *d, const float *w, void *wpacked, void *stream);

/* Packs the ADJOINT weights of a stride-1 Conv2d for its data-gradient convolution: `d` describes
 * the adjoint convolution (Cin = forward Cout, Cout = forward Cin) and w is the FORWARD weight
 * [d->Cin, d->Cout, kh, kw]; the channel swap and the tap reversal happen in the packing pass. */
int ir2rgb_conv2d_pack_weight_adjoint(const ir2rgb_conv_desc *d, const float *w, void *wpacked, void *stream);

/* Batched packing -- all packed copies of a network refreshed by ONE launch after an optimizer step
 * (the reference has no counterpart: cuDNN reads torch-layout weights, networks.py convolutions).
 * A job is the argument list of ir2rgb_conv2d_pack_weight[_adjoint].  _table_bytes returns the size
 * of the job table (negative IR2RGB_E* on error); _build writes it into HOST memory (16-byte
 * aligned) and returns the entry count (<= 4 per job, one per weight class; negative on error) and
 * the launch's block count; the caller copies the table to 16-byte aligned device memory once -- it
 * stays valid while the job pointers do -- and calls _run every time the weights changed.  All jobs
 * of a table share one dtype. */
typedef struct ir2rgb_pack_job {
    ir2rgb_conv_desc desc;
    const float *w;
    void *wpacked;
    int adjoint;
    int reserved;
} ir2rgb_pack_job;
long ir2rgb_conv2d_pack_batch_table_bytes(const ir2rgb_pack_job *jobs, int njobs);
int ir2rgb_conv2d_pack_batch_build(const ir2rgb_pack_job *jobs, int njobs, void *table_host, long table_bytes,
                                   int *nblocks);
int ir2rgb_conv2d_pack_batch_run(const void *table_dev, int nentries, int nblocks, int dtype, void *stream);

/* y = act(conv(x) + bias); bias and stats_partial may be NULL.  x, wpacked, y 16-byte aligned. */
int ir2rgb_conv2d_fwd(const ir2rgb_conv_desc *d, const void *x, const void *wpacked, const float *bias, void *y,
                      float *stats_partial, void *stream);

/* The same with a caller-owned workspace, which lets layers with too few output tiles for the chip split their input
 * channels over two workgroups per tile (the 1024 -> 1024 3x3 convolutions of ResnetBlock, networks.py:556-580, at
 * 32 x 64 and their data gradients: 128 tiles of 128 px x 128 cout for 256 CUs).  ir2rgb_conv2d_fwd_workspace_bytes:
 * 0 = this descriptor has no use for one (pass NULL), < 0 = error.  The workspace is 16-byte aligned, must be ZERO
 * when first used and is left ready for the next launch by every launch (tickets advance by a fixed amount per
 * launch; the partial sums need no initialisation); launches that share one must be ordered (same stream).
 * Results are bit-identical from launch to launch: two partial sums are added, in either order. */
long ir2rgb_conv2d_fwd_workspace_bytes(const ir2rgb_conv_desc *d);
int ir2rgb_conv2d_fwd_ws(const ir2rgb_conv_desc *d, const void *x, const void *wpacked, const float *bias, void *y,
                         float *stats_partial, void *workspace, long workspace_bytes, void *stream);

/* Name of the device kernel ir2rgb_conv2d_fwd launches for `d` ("conv3x3_patch_kernel",
 * "conv_igemm_kernel", "conv_igemm_classes_kernel"; "" for an invalid descriptor): for profiles. */
const char *ir2rgb_conv2d_kernel_name(const ir2rgb_conv_desc *d);

/* Training-mode BatchNorm2d statistics (reference norm_layer = nn.BatchNorm2d, networks.py:41-48).
 * Reduces the [rows][2][C] partial sums written by ir2rgb_conv2d_fwd over `count` pixels into
 * scale = gamma*invstd and shift = beta - mean*scale, and updates running_mean / running_var
 * in place (momentum form of torch: new = (1-m)*old + m*batch, unbiased variance).  gamma, beta,
 * running_*, mean_out, invstd_out may be NULL.  stat_updates >= 1: the momentum update is applied
 * that many times (a forward that stands for several identical forwards of the reference, e.g.
 * compute_loss_D evaluating netD on the same real frames twice per step, discriminator.py:154-166). */
int ir2rgb_bn_finalize(const float *stats_partial, int rows, int C, long count, const float *gamma,
                       const float *beta, float *running_mean, float *running_var, float momentum, float eps,
                       float *scale, float *shift, float *mean_out, float *invstd_out, int stat_updates,
                       void *stream);

/* The same with two more inputs.
 * conv_bias (may be NULL): the statistics are those of the convolution output WITHOUT its bias.  BatchNorm subtracts
 *   the batch mean, so the bias of the convolution in front of it (nn.Conv2d(..., bias=True) + norm_layer,
 *   networks.py:141-171) cancels exactly; leaving it out of the half-precision activations keeps a constant input
 *   (e.g. the all-zero previous frames of no_first_img, generator.py:219-220) exactly constant, as it is in fp32.
 *   Only the running mean sees it: running_mean follows mean + conv_bias, as nn.BatchNorm2d's does.
 * frozen != 0: module.eval() -- no batch statistics (stats_partial / rows / count unused), scale and shift come
 *   from running_mean / running_var (both required):  scale = gamma*rsqrt(running_var+eps),
 *   shift = beta - (running_mean - conv_bias)*scale; mean_out = running_mean - conv_bias; nothing is updated. */
int ir2rgb_bn_finalize_ex(const float *stats_partial, int rows, int C, long count, const float *gamma,
                          const float *beta, const float *conv_bias, float *running_mean, float *running_var,
                          float momentum, float eps, float *scale, float *shift, float *mean_out,
                          float *invstd_out, int stat_updates, int frozen, void *stream);

/* ir2rgb_bn_finalize_ex (training mode, frozen = 0) and ir2rgb_bn_apply in ONE launch, for convolutions that wrote at
 * most IR2RGB_BN_FUSED_MAX_ROWS partial rows (the residual blocks).  Same arithmetic in the same order: results are
 * bit-identical to the two calls.  C % 64 == 0.  x / y / res1 / res2 as in ir2rgb_bn_apply (y may alias x). */
#define IR2RGB_BN_FUSED_MAX_ROWS 128
int ir2rgb_bn_finalize_apply(const float *stats_partial, int rows, int C, long count, const float *gamma,
                             const float *beta, const float *conv_bias, float *running_mean, float *running_var,
                             float momentum, float eps, float *scale, float *shift, float *mean_out,
                             float *invstd_out, int stat_updates, const void *x, const void *res1, const void *res2,
                             void *y, long npix, int act, int dtype, void *stream);

/* y = act(x*scale[c] + shift[c]) + res1 + res2 on NHWC half tensors of npix pixels x C channels
 * (C % 8 == 0).  act: 0 none, 1 ReLU, 2 LeakyReLU(0.2).  res1/res2 may be NULL; y may alias x.
 * Covers norm+activation (networks.py:141-171, :253-271, :678-699), the ResnetBlock skip
 * (:585) and the encoder sum (:192, :290-291). */
int ir2rgb_bn_apply(const void *x, const float *scale, const float *shift, const void *res1, const void *res2,
                    void *y, long npix, int C, int act, int dtype, void *stream);

/* Layout converters between the reference's NCHW fp32 tensors and NHWC half. */
int ir2rgb_nchw_f32_to_nhwc_half(const float *in, void *out, int N, int C, int H, int W, int dtype, void *stream);
int ir2rgb_nhwc_half_to_nchw_f32(const void *in, float *out, int N, int C, int H, int W, int dtype, void *stream);

/* x-direction im2col of a small-channel NCHW fp32 image into 64-channel NHWC half:
 *   out[n][y][ox][ci*KW + kx] = in[n][ci][y][pad(ox*stride_w + kx - pad_w)],  Cin*KW <= 64.
 * Turns the first layers (ReflectionPad2d(3)+Conv7x7 on 9/6 channels, networks.py:141,:150,
 * :253-255; Conv4x4 s2 p2 on 6/13 channels, :680) into KH x 1 convolutions for the MFMA kernel. */
int ir2rgb_xexpand(const float *in, void *out, int N, int Cin, int H, int W, int Wout, int KW, int stride_w,
                   int pad_w, int pad_mode, int dtype, void *stream);
/* Same with Cx = 64 or 128 output channels (Cin*KW <= Cx): the 12-channel 7x7 first layer of FlowNetS. */
int ir2rgb_xexpand_cx(const float *in, void *out, int N, int Cin, int H, int W, int Wout, int KW, int stride_w,
                      int pad_w, int pad_mode, int Cx, int dtype, void *stream);

/* NCHW fp32 [N,C,H,W] -> channels [c_off, c_off+C) of an NHWC half buffer with ld channels per pixel,
 * with an optional fused activation (0 none, 2 LeakyReLU(0.1): corr_activation, FlowNetC.py:32). */
int ir2rgb_nchw_f32_to_nhwc_half_slice(const float *in, void *out, int N, int C, int H, int W, int ld, int c_off,
                                       int act, int dtype, void *stream);

/* Finish of a separable head: T [N,H,W,CT] fp32 holds, in channel co*KH+ky, the horizontal
 * part of a KHxKW convolution; out[n][co][y][x] = f(sum_ky T[n][refl(y+ky-pad)][x][co*KH+ky] + bias[co]).
 * acts packs one nibble per output channel: 0 -> linear * mul, 1 -> tanh, 2 -> sigmoid.
 * (ReflectionPad2d(3)+Conv7x7 heads with tanh / *20 / sigmoid, networks.py:166,:170-171,:200-201) */
int ir2rgb_head_finish(const float *T, const float *bias, float *out, int N, int H, int W, int Cout, int KH,
                       int CT, int pad_h, unsigned acts, float mul, void *stream);

/* Temporal blend of the generator (networks.py:89-100, :207-209):
 *   warp = grid_sample(prev, grid + flow_normalised, bilinear, border)   [align_corners quirk kept]
 *   out  = raw * w + warp * (1 - w)
 * raw [N,3,H,W], prev [N,Cp,H,W] (its LAST 3 channels are warped), flow [N,2,H,W], w [N,1,H,W]. */
int ir2rgb_warp_blend_fwd(const float *raw, const float *prev, const float *flow, const float *w, float *out,
                          float *warp_out, int N, int Cp, int H, int W, void *stream);

/* ------------------------------------------------------------------------------------------
 * Backward companions (HBM-bound).  The data gradient of every convolution is itself an
 * ir2rgb_conv2d_fwd call (transposed <-> strided, flipped weights); these cover the rest of
 * what torch.autograd + cuDNN do for the reference's loss.backward() (train_vid2vid.py:166-169).
 * ------------------------------------------------------------------------------------------ */

/* Rows R of the scratch needed by ir2rgb_bn_bwd: `partial` must hold (R*2 + 3)*C floats
 * ([R][2][C] pixel-range partial sums followed by three coefficient vectors).  < 0: error; C must
 * be a power of two in [64, 2048]. */
int ir2rgb_bn_bwd_blocks(long npix, int C);

/* Backward of activation + training-mode BatchNorm2d on NHWC half tensors:
 *   g' = gz * act'(y*scale+shift);  dbeta = sum g';  dgamma = sum g'*yhat;
 *   gy = scale * (g' - dbeta/n - yhat*dgamma/n),  yhat = (y-mean)*invstd.
 * With scale == NULL (no norm layer): gy = gz * act'(y) and dbeta = sum gy (the bias gradient).
 * act: 0 none, 1 ReLU, 2 LeakyReLU(0.2); act | 16: the statistics were frozen (evaluation-mode BatchNorm,
 * ir2rgb_bn_finalize_ex(frozen)): gy = scale * g', dgamma / dbeta as above; act | 32: dgamma / dbeta are ADDED to what the
 * two vectors hold (the later sample groups of a layer whose batch is several independent forwards).  gy may alias gz. */
int ir2rgb_bn_bwd(const void *gz, const void *y, const float *scale, const float *shift, const float *mean,
                  const float *invstd, void *gy, float *dgamma, float *dbeta, float *partial, long npix, int C,
                  int act, int dtype, void *stream);

/* dst[i] = idx[i] >= 0 ? src[idx[i]] : 0 (fp32; n elements): rearranged copies of a weight tensor (x-im2col layout of
 * the first layers, zero-padded widths) from a precomputed index map, refreshed after every optimizer step. */
int ir2rgb_gather_f32(const float *src, const int *idx, float *dst, long n, void *stream);

/* AvgPool2d(3, stride 2, padding 1, count_include_pad=False) on fp32 planes: the image pyramids of the multi-scale
 * discriminators (reference networks.py:639, :658-666) and of the generator inputs (base_model.py:64-82).
 * backward 0: x [planes][H][W] -> y [planes][Ho][Wo], Ho = (H-1)/2 + 1;  backward 1: x = the gradient
 * [planes][Ho][Wo] -> y = the input gradient [planes][H][W]. */
int ir2rgb_avgpool3s2(const float *x, float *y, long planes, int H, int W, int backward, void *stream);

/* Gradient of a thin fp32 convolution output (the 1-channel PatchGAN logits, NLayerDiscriminator's last
 * layer, networks.py:676-677) prepared for the MFMA kernels: gz [N,Cout,H,W] fp32, Cout <= 8 ->
 * g64 [N,H,W,64] and g8 [N,H,W,8] NHWC half (channels >= Cout zero; data- / weight-gradient operands) and
 * dbias[Cout] = sum over N,H,W of gz (deterministic). */
int ir2rgb_thin_grad_expand(const float *gz, void *g64, void *g8, float *dbias, int N, int Cout, int H, int W,
                            int dtype, void *stream);

/* Adjoint of nn.ReflectionPad2d: dxpad [N,H+2*pad_h,W+2*pad_w,C] -> dx [N,H,W,C] (NHWC half). */
int ir2rgb_fold_reflect(const void *dxpad, void *dx, int N, int H, int W, int C, int pad_h, int pad_w, int dtype,
                        void *stream);

/* Backward of ir2rgb_head_finish: gout/out [N,Cout,H,W] fp32 (out = the forward result) ->
 * dT [N,H,W,CT] half (gradient w.r.t. the separable row responses, channels >= Cout*KH zeroed)
 * and dbias [Cout] fp32. */
int ir2rgb_head_finish_bwd(const float *gout, const float *out, void *dT, float *dbias, float *partial, int N, int H, int W,
                           int Cout, int KH, int CT, int pad_h, unsigned acts, float mul, int dtype, void *stream);
/* Rows of 8 floats `partial` must hold for ir2rgb_head_finish_bwd (one per workgroup: the bias gradient is summed in a
 * fixed order by a second, one-workgroup kernel -- no float atomics, bit-reproducible). */
int ir2rgb_head_finish_bwd_rows(int N, int H, int W);

/* Backward of ir2rgb_warp_blend_fwd w.r.t. raw, flow and w (prev is a detached input on the
 * training path, reference generator.py:153-154). */
int ir2rgb_warp_blend_bwd(const float *gout, const float *raw, const float *prev, const float *flow, const float *w,
                          float *graw, float *gflow, float *gw, int N, int Cp, int H, int W, void *stream);

/* Adjoint of ir2rgb_xexpand: dxe [N,H,Wout,64] half -> din [N,Cin,H,W] fp32. */
int ir2rgb_xexpand_bwd(const void *dxe, float *din, int N, int Cin, int H, int W, int Wout, int KW, int stride_w,
                       int pad_w, int pad_mode, int dtype, void *stream);

/* Weight gradient of a convolution described by `d` (same descriptor as the forward call) on the
 * matrix cores: x is the forward input [N,Hin,Win,Cin], gy the gradient w.r.t. the forward output
 * [N,Hout,Wout,Cout] (both NHWC half, channels % 8 == 0); dw receives the gradient in the torch
 * weight layout, fp32 ([Cout,Cin,kh,kw], or [Cin,Cout,kh,kw] when d->transposed).  `workspace`
 * holds ir2rgb_conv2d_wgrad_workspace_elems(d) floats.  Reflection padding is honoured by the
 * gather (no padded copy).  Replaces cuDNN's backward-filter in the reference's loss.backward(). */
long ir2rgb_conv2d_wgrad_workspace_elems(const ir2rgb_conv_desc *d);
int ir2rgb_conv2d_wgrad(const ir2rgb_conv_desc *d, const void *x, const void *gy, float *dw, float *workspace,
                        void *stream);

/* dw += the same weight gradient (a parameter that is used several times in one backward pass: the discriminators are
 * applied to two or three inputs per window, discriminator.py:154-166): the sum with the earlier contributions happens in
 * the kernel's own finish pass instead of in a separate add.  Workspace: ir2rgb_conv2d_wgrad_acc_workspace_elems. */
int ir2rgb_conv2d_wgrad_acc(const ir2rgb_conv_desc *d, const void *x, const void *gy, float *dw, float *workspace,
                            void *stream);
long ir2rgb_conv2d_wgrad_acc_workspace_elems(const ir2rgb_conv_desc *d);

/* FlowNet2 flow up-sampler ConvTranspose2d(2,2,4,2,1) (reference FlowNetC.py:47-50 etc.): in [N,2,h,w]
 * fp32 NCHW, weight [2,2,4,4], bias [2] or NULL -> channels [c_off, c_off+2) of an NHWC half buffer
 * [N,2h,2w,ld]. */
int ir2rgb_flow_upsample_slice(const float *in, const float *weight, const float *bias, void *out, int N, int h,
                               int w, int ld, int c_off, int dtype, void *stream);

/* ------------------------------------------------------------------------------------------
 * Scalar losses of the loop body, a group of terms per launch.  Replaces the chains of torch
 * elementwise + reduction kernels behind criterionFeat (nn.L1Loss on discriminator features,
 * reference discriminator.py:199-210), criterionGAN (least-squares GANLoss, models/networks.py
 * GANLoss / loss.py) and the confidence-masked L1 terms (criterionFlow, loss.py MaskedL1Loss).
 *   kind 0: weight * mean |a - b|          a, b half tensors of identical dense layout, n % 8 == 0
 *   kind 1: weight * mean (a - target)^2   a fp32
 *   kind 2: weight * mean |a*m - b*m|      a, b fp32 NCHW [N,C,H,W] (b NULL = zeros), mask fp32
 *                                          [N,1,H,W]; hw = H*W, chw = C*H*W
 * Every term adds into out[slot] (slot 0..3).  The sums are deterministic (per-block partials added
 * in block order).  ga, when not NULL, receives d out[slot] / d a scaled by gout[slot] (same dtype
 * and layout as a).
 * ------------------------------------------------------------------------------------------ */
#define IR2RGB_LOSS_MAX_ITEMS 32
typedef struct ir2rgb_loss_item {
    const void *a;
    const void *b;
    void *ga;
    const void *mask;
    long n;
    long hw;
    long chw;
    float weight;
    float target;
    int kind;
    int slot;
} ir2rgb_loss_item;

/* Floats of `partial` scratch that ir2rgb_loss_multi_fwd needs. */
int ir2rgb_loss_partial_elems(void);
/* out[0 .. max slot] = the summed terms (slots no term names are not written). */
int ir2rgb_loss_multi_fwd(const ir2rgb_loss_item *items, int count, int dtype, float *partial, float *out,
                          void *stream);
/* Writes items[i].ga for every item that has one; gout = gradient w.r.t. out (device, fp32). */
int ir2rgb_loss_multi_bwd(const ir2rgb_loss_item *items, int count, int dtype, const float *gout, void *stream);

/* ------------------------------------------------------------------------------------------
 * torch.optim.Adam step (weight_decay 0, amsgrad off) of one optimizer in one launch; replaces the
 * per-tensor torch kernels behind optimizer_G/D/D_T.step() (reference train_vid2vid.py:93-105).
 *   table  : device array of { float *p; const float *g; float *m; float *v; long n; } (40 bytes each)
 *   blocks : device array of int pairs (tensor index, chunk index): one workgroup updates elements
 *            [chunk*E, min(n, (chunk+1)*E)) of its tensor, E = ir2rgb_adam_chunk_elems(); the caller
 *            lists every chunk of every tensor exactly once
 *   step   : 1-based step count (bias corrections 1 - beta^step are evaluated on the host in double)
 * ------------------------------------------------------------------------------------------ */
int ir2rgb_adam_chunk_elems(void);
int ir2rgb_adam_step(const void *table, const void *blocks, int nblocks, float lr, float beta1, float beta2, float eps,
                     int step, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* IR2RGB_HIP_H */
