/*
 * reidgan_hip.h — C ABI of libreidgan_hip.so, the MI355X (gfx950) kernel library behind the
 * ReID-GAN training step (joint FD-GAN + cluster-contrast).
 *
 * The reference (daemon-219/ReID-GAN) has no FFI: its hot path is torch.nn modules calling
 * cuDNN/cuBLAS/ATen.  Each entry point below replaces the ATen/cuDNN work behind one family of
 * reference call sites (cited per function as FD/ = FD-GAN-master/, CC/ = cluster-contrast-reid-main/).
 * The host side that binds them with ctypes lives in reid-gan_amd/rg_hip/lib.py; INTEGRATION.md
 * shows the stub a reference maintainer would add.
 *
 * Contract (every function):
 *   - returns 0 (RG_OK) or a negative status; rg_last_error() gives the message (thread local);
 *   - plain pointers to DEVICE memory and sizes; tensors are contiguous fp32 NCHW, labels int64;
 *   - asynchronous on `stream`; never allocates, frees or synchronises (rg_profile_collect excepted; the first call of a
 *     convolution geometry outside a stream capture waits once for a timing of its candidate kernels on `stream`: rg_conv_tune_stats);
 *   - scratch comes from the caller: `workspace`/`workspace_bytes`, sized by the *_workspace query;
 *   - pointers must be 16-byte aligned (torch allocations are).
 */
#ifndef REIDGAN_HIP_H
#define REIDGAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* rg_stream_t; /* == hipStream_t */

#define RG_OK 0
#define RG_ERR_INVALID (-1)
#define RG_ERR_LAUNCH (-2)
#define RG_ERR_WORKSPACE (-3)

/* activation codes of the fused epilogues */
#define RG_ACT_NONE 0
#define RG_ACT_RELU 1
#define RG_ACT_LEAKY 2
#define RG_ACT_TANH 3

const char* rg_last_error(void);
int rg_version(void);

/* ---- per-family launch profiler (HIP events on the launch stream; used by bench.py) ---------- */
#define RG_FAMILY_COUNT 11 /* conv_fwd, conv_dgrad, conv_wgrad, norm, eltwise, pool, loss, cm, optim, misc, conv_f8 */
int rg_family_count(void);
int rg_profile_enable(int on);
int rg_profile_reset(void);
int rg_profile_collect(double* ms, double* flops, double* bytes, long long* calls); /* synchronises events */

/* ---- convolution: implicit GEMM, fp32 tensors, split-bf16 arithmetic on v_mfma_f32_32x32x16_bf16 ----
 * (every fp32 operand is split exactly into three bf16 pieces and each product evaluated as the six partial products of weight
 * >= 2^-16 with fp32 accumulation: the error model of an fp32 FMA chain, csrc/conv_igemm.hip; -DRG_MATH=1 builds the fp32-MFMA
 * v_mfma_f32_32x32x2_f32 arithmetic of rounds 1-2).  Overflow is NOT fp32's: an operand that is +-Inf, or finite but above the
 * largest bf16 (|x| >= 3.3961e38: its leading piece rounds to Inf), makes its products NaN (Inf - Inf in the split) where an fp32
 * FMA chain would give +-Inf; nothing in the reference's training range comes within 30 orders of magnitude of that)
 * Replaces nn.Conv2d / nn.ConvTranspose2d of the ResNet-50 trunk (CC/clustercontrast/models/
 * resnet_ibn_a.py:70-159, FD/reid/models/resnet.py:65-75), CustomPoseGenerator
 * (FD/fdgan/networks.py:86-138) and NLayerDiscriminator (FD/fdgan/networks.py:206-232), and
 * nn.Linear / Tensor.mm (FD/reid/models/embedding.py:21-24, CC/clustercontrast/models/cm.py:16,26)
 * as 1x1 geometry.
 * Geometry: x[N][C][H][W], w[K][C][KH][KW], y[N][K][P][Q]; y = act(conv(x,w)*scale[k] + shift[k] + residual).
 * scale/shift/residual may be NULL. dgrad computes dx from dy (== ConvTranspose2d forward with
 * dy as its input and dx[N][C][H][W] as its output; epilogue indexed by c). Stride <= 2 for dgrad.
 */
size_t rg_conv2d_fwd_workspace(int N, int C, int K, int KH, int KW, int P, int Q);
/* w_krsc (optional, may be NULL): the weights re-laid out as [K][KH*KW][C] by rg_weights_to_krsc.  With it the
 * kernels walk the reduction (r,s)-major, so the padding test of the pixel gather happens once per 16-deep k-tile
 * (fwd, C % 16 == 0) and the weight operand of the data gradient is contiguous (float4 loads; K % 16 == 0,
 * C % 4 == 0); NULL selects the generic loaders.  Workspace: split-K scratch for layers too small to fill the chip
 * (dgrad: stride 1 only); a too-small workspace just disables the split.  Every tensor must be < 2 GiB. */
int rg_conv2d_fwd(const float* x, const float* w, const float* w_krsc, float* y, int N, int C, int H, int W, int K,
                  int KH, int KW, int SH, int SW, int PH, int PW, int P, int Q, const float* scale, const float* shift,
                  const float* residual, int act, float slope, void* workspace, size_t workspace_bytes,
                  rg_stream_t stream);
size_t rg_conv2d_dgrad_workspace(int N, int C, int H, int W, int K, int KH, int KW, int SH, int SW);
/* relu_mask (dgrad only, may be NULL): a tensor shaped like dx; after scale/shift/residual/act the result is zeroed where
 * relu_mask <= 0.  Passing the convolution's own forward INPUT (the ReLU output of the layer below) makes this the
 * ReLU backward of that layer, applied to the sum of this data gradient and `residual` (the skip gradient). */
int rg_conv2d_dgrad(const float* dy, const float* w, const float* w_krsc, float* dx, int N, int C, int H, int W, int K,
                    int KH, int KW, int SH, int SW, int PH, int PW, int P, int Q, const float* scale,
                    const float* shift, const float* residual, int act, float slope, const float* relu_mask,
                    float* rowsum, int rowsum_cols, void* workspace, size_t workspace_bytes, rg_stream_t stream);
/* rowsum (dgrad only, may be NULL): [C][rowsum_cols] — per input channel, the sums of the FINAL dx values over blocks of
 * pixels (one column per class, pixel tile and wave column; fixed summation order).  Their sum over the columns is the
 * per-channel sum of dx: exactly the `partials` rg_bn_fold_wgrad of the layer below needs (dbeta / dgamma), so that layer does
 * not read dx again.  rowsum_cols must equal rg_conv2d_dgrad_rowsum_cols(...), which is 0 when the launch would use split-K
 * or the small-C kernel (no fused row sums there). */
int rg_conv2d_dgrad_rowsum_cols(int N, int C, int H, int W, int K, int KH, int KW, int SH, int SW, int PH, int PW, int P,
                                int Q);
int rg_weights_to_krsc(const float* w, float* w_krsc, int K, int C, int KH, int KW, rg_stream_t stream);
/* the same re-layout for every filter of a network in ONE launch: `table` (device memory) holds `count` entries of 6 int64 words
 * {w, w_krsc, K, C, KH*KW, first block}, filters are cut into blocks of rg_krsc_chunk() elements, total_blocks = sum of the blocks */
int rg_krsc_chunk(void);
int rg_weights_to_krsc_multi(const void* table, int count, int total_blocks, rg_stream_t stream);
/* development knob: pin the fwd/dgrad planner's tile (0: 128x128, 1: 64x128, 2: 64x64, 3: 32x256) and split-K count;
 * (-1, -1) releases it (same effect as the RG_CONV_FORCE="tile,splits" environment variable) */
int rg_conv_set_force(int tile, int splits);
/* development knob: bit mask of the conv kernel families that run on the bf16-plane operand path of csrc/conv_planes.h (operands
 * split into their bf16 pieces once, on the way into LDS) instead of the default kernels: 1 forward, 2 data gradient, 4 weight
 * gradient, 8: with a family bit, the eight-wave form of the 128 x 128 tile (same as RG_CONV_PL); identical results up to fp32
 * summation order.  Returns the previous mask. */
int rg_conv_set_planes(int mask);
/* The generic fwd / dgrad / wgrad kernels exist in two implementations (default kernels and the plane path above); unless one is
 * forced (rg_conv_set_planes, RG_CONV_TUNE=0) the first call of a geometry outside a stream capture times both on the launch
 * stream and the faster serves that geometry from then on (the first call's output already comes from it); the tile / split-K plan
 * of a forward geometry and the tap-reuse-vs-generic path of a 3x3 layer are chosen the same way.  Returns the number of choices
 * measured so far; out[0] / out[1] (int[2], may be NULL): how many kernel choices went to the default kernels / the plane path. */
int rg_conv_tune_stats(int* out);
/* Split-K without the finishing launch.  Registers (count > 0) or removes (counters == NULL) the arrival counters of `stream`:
 * `count` ZERO-INITIALISED 32-bit words of device memory that the caller keeps alive and never writes (the library does not
 * allocate; launches on one stream are ordered and every launch leaves its counters at zero, so one buffer per stream serves all of
 * them).  With counters registered, a split forward / unit-stride data-gradient launch on that stream with <= count output tiles
 * finishes inside the convolution kernel: the last split of a tile to arrive sums the tile's partials in split order and applies
 * the epilogue — the finishing kernel's values bit for bit, whichever split arrives last.  Without them, or inside a stream capture,
 * the finishing kernel follows as before.  (Measured equal to the finishing kernel within +-1 % on the FD-GAN step: rg_hip registers
 * counters only with RG_SPLITK_INKERNEL=1.) */
int rg_conv_splitk_arrivals(void* counters, int count, rg_stream_t stream);
/* test / development query: the number of split-K launches that finished inside the convolution kernel so far (this process) */
int rg_conv_splitk_inkernel_count(void);
size_t rg_conv2d_wgrad_workspace(int N, int C, int K, int KH, int KW, int P, int Q);
int rg_conv2d_wgrad(const float* x, const float* dy, float* dw, int N, int C, int H, int W, int K, int KH, int KW,
                    int SH, int SW, int PH, int PW, int P, int Q, void* workspace, size_t workspace_bytes,
                    rg_stream_t stream);

/* ---- BatchNorm 1d/2d on [N][C][HW] ------------------------------------------------------------
 * Replaces nn.BatchNorm2d/1d (FD/fdgan/networks.py:26-35; resnet_ibn_a.py:75-81; FD/reid/models/
 * embedding.py:16-19; CC/clustercontrast/models/resnet.py:58-66). `stat` is invstd (train, from
 * rg_bn_stats) or the running variance when stat_is_var != 0 (eval). The apply kernels fuse an
 * optional residual add and activation; the backward takes the forward OUTPUT for the mask. */
size_t rg_bn_workspace(int N, int C, int HW);
int rg_bn_stats(const float* x, float* mean, float* invstd, float* running_mean, float* running_var, int N, int C,
                int HW, float eps, float momentum, void* workspace, size_t workspace_bytes, rg_stream_t stream);
int rg_bn_apply_fwd(const float* x, const float* mean, const float* stat, const float* gamma, const float* beta,
                    const float* residual, float* y, int N, int C, int HW, int stat_is_var, float eps, int act,
                    float slope, rg_stream_t stream);
int rg_bn_bwd_reduce(const float* x, const float* dy, const float* y_act, const float* mean, const float* stat,
                     float* sum_dy, float* sum_dy_xhat, int N, int C, int HW, int stat_is_var, float eps, int act,
                     float slope, void* workspace, size_t workspace_bytes, rg_stream_t stream);
/* Train-mode torch.nn.BatchNorm2d / 1d (FD/fdgan/networks.py:28, CC/clustercontrast/models/resnet.py trunk) in one launch per
 * direction for small per-channel extents (rg_bn_train_fused_ok: N*HW <= 16384 and C >= 128): forward computes the batch statistics,
 * updates the running statistics, writes mean / invstd [C] and y = act(gamma*xhat + beta + residual); backward writes the channel sums
 * of g = dy*act'(y) and g*xhat (dbeta / dgamma) and dx / dres (either may be NULL). */
size_t rg_bn_train_fused_ok(int N, int C, int HW);
int rg_bn_train_fwd_fused(const float* x, const float* gamma, const float* beta, const float* residual, float* y, float* mean,
                          float* invstd, float* running_mean, float* running_var, int N, int C, int HW, float eps, float momentum,
                          int act, float slope, rg_stream_t stream);
int rg_bn_train_bwd_fused(const float* x, const float* dy, const float* y_act, const float* mean, const float* invstd,
                          const float* gamma, float* dx, float* dres, float* sum_dy, float* sum_dy_xhat, int N, int C, int HW,
                          int act, float slope, rg_stream_t stream);
/* torch.nn.InstanceNorm2d (affine or not; FD/fdgan/networks.py:30, CC/dual_gan/models/base_function.py:38-49) in one launch per
 * direction: forward writes y = act(gamma[c] * xhat + beta[c] + residual) and the per-instance mean / invstd [N*C]; backward
 * writes dx, dres (either may be NULL), the per-instance sums of g = dy*act'(y) and g*xhat [N*C] and, when sum_dx is given, the
 * per-instance sums of dx (the bias gradient of the convolution in front of the norm). */
int rg_instnorm_fwd(const float* x, const float* gamma, const float* beta, const float* residual, float* y, float* mean,
                    float* invstd, int N, int C, int HW, float eps, int act, float slope, rg_stream_t stream);
int rg_instnorm_bwd(const float* x, const float* dy, const float* y_act, const float* mean, const float* invstd,
                    const float* gamma, float* dx, float* dres, float* sum_dy, float* sum_dy_xhat, float* sum_dx, int N, int C,
                    int HW, int act, float slope, rg_stream_t stream);
/* out[c] = sum_{n, pixels} dy[n][c][.] in one launch (the bias gradient of torch's Conv2d / Linear backward, grad_output.sum over
 * every axis but the channel) when rg_channel_sum_ok says so — at most 32768 values per channel; larger maps use rg_bn_bwd_reduce */
size_t rg_channel_sum_ok(int N, int C, int HW);
int rg_channel_sum(const float* dy, float* out, int N, int C, int HW, rg_stream_t stream);
/* out_a[c] = sum_n a[n][c], out_b[c] = sum_n b[n][c]: dgamma / dbeta of an affine InstanceNorm2d (torch.nn.InstanceNorm2d
 * backward as used by CC/dual_gan/models/base_function.py:38-49) from the per-(n,c) sums of rg_bn_bwd_reduce. */
int rg_rows_sum_pair(const float* a, const float* b, float* out_a, float* out_b, int N, int C, rg_stream_t stream);
int rg_bn_bwd_apply(const float* x, const float* dy, const float* y_act, const float* mean, const float* stat,
                    const float* gamma, const float* sum_dy, const float* sum_dy_xhat, float* dx, float* dres, int N,
                    int C, int HW, int train, int stat_is_var, float eps, int act, float slope, rg_stream_t stream);

/* eval-mode backward in ONE pass (dx, dres and — when sum_dy / sum_dy_xhat are given — the affine-gradient sums) */
int rg_bn_eval_bwd(const float* x, const float* dy, const float* y_act, const float* running_mean,
                   const float* running_var, const float* gamma, float* dx, float* dres, float* sum_dy,
                   float* sum_dy_xhat, int N, int C, int HW, float eps, int act, float slope, void* workspace,
                   size_t workspace_bytes, rg_stream_t stream);

/* ---- element-wise --------------------------------------------------------------------------- */
int rg_act_fwd(const float* x, float* y, int64_t n, int act, float slope, rg_stream_t stream);
int rg_act_bwd(const float* dy, const float* y, float* dx, int64_t n, int act, float slope, rg_stream_t stream);
int rg_axpby(const float* a, const float* b, float* y, int64_t n, float alpha, float beta, rg_stream_t stream);
/* Pair batch of FDGANModel.set_input (FD/fdgan/model.py:127-150): out[0:B] = a, out[B:2B][i] = take_a[i] ? a[i] : b[i] — the
 * reference's `cat([x1, x1*mask + x2*(1-mask)])` with a 0/1 mask, and plain `cat([x1, x2])` for take_a == NULL.  per = floats
 * per sample. */
int rg_pair_cat(const float* a, const float* b, const int64_t* take_a, float* out, int B, int64_t per, rg_stream_t stream);
int rg_fill(float* y, int64_t n, float v, rg_stream_t stream);
/* one idle wave for `us` microseconds (1..100000) on `stream`: the probe rg_hip.ops.concurrent_stream uses to find streams that do
 * not share a hardware queue (no reference counterpart: the reference is single-stream) */
int rg_spin_us(int us, rg_stream_t stream);
/* (x1-x2)^2 of EltwiseSubEmbed, FD/reid/models/embedding.py:26-31 */
int rg_sub_square_fwd(const float* a, const float* b, float* y, int64_t n, rg_stream_t stream);
int rg_sub_square_bwd(const float* a, const float* b, const float* dy, float* da, float* db, int64_t n,
                      rg_stream_t stream);
/* nn.Dropout of the generator decoder, FD/fdgan/networks.py:105-109,149-156; counter-based mask */
int rg_dropout(const float* x, float* y, int64_t n, float p, unsigned long long seed, rg_stream_t stream);
int rg_dropout_clocked(const float* x, float* y, int64_t n, float p, unsigned long long seed, const unsigned long long* clock,
                       rg_stream_t stream);
/* F.normalize(x, dim=1) on [rows][D], CC/clustercontrast/models/cm.py:125, resnet.py:90-107 */
int rg_l2norm_rows_fwd(const float* x, float* y, float* norm, int rows, int D, float eps, rg_stream_t stream);
int rg_l2norm_rows_bwd(const float* y, const float* dy, const float* norm, float* dx, int rows, int D, float eps,
                       rg_stream_t stream);
/* F.normalize(x, dim=1) of a feature MAP [N][C][HW] (unit norm over the channels at every pixel; the `gan_x` output of the
 * cluster-contrast ResNet in train mode, CC/clustercontrast/models/resnet.py:100-107) and its backward; norm is [N][HW] */
int rg_l2norm_channels_fwd(const float* x, float* y, float* norm, int N, int C, int HW, float eps, rg_stream_t stream);
int rg_l2norm_channels_bwd(const float* y, const float* dy, const float* norm, float* dx, int N, int C, int HW, float eps,
                           rg_stream_t stream);
/* torch.cat along channels / its backward slices, FD/fdgan/model.py:160-161, networks.py:175 */
int rg_copy_channels(const float* src, float* dst, int N, int Cc, int HW, int Cs, int sc0, int Cd, int dc0,
                     int accumulate, rg_stream_t stream);

/* my_resize / my_normalize / my_transform, CC/clustercontrast/utils/data/diff_augs.py:6-16: bicubic (A = -0.75,
 * align_corners = False, clamped border taps) resize [N,C,H,W] -> [N,C,OH,OW] fused with (v - mean[c]) / std[c];
 * mean/std are device arrays of C floats or both NULL (resize only); OH == H and OW == W normalises only.
 * The backward is the exact adjoint, gathered per input pixel (deterministic). */
int rg_bicubic_normalize_fwd(const float* x, float* y, int N, int C, int H, int W, int OH, int OW, const float* mean,
                             const float* stdv, rg_stream_t stream);
int rg_bicubic_normalize_bwd(const float* dy, float* dx, int N, int C, int H, int W, int OH, int OW, const float* stdv,
                             rg_stream_t stream);

/* ClusterMemory_Gradient.update_clusters, CC/clustercontrast/models/cm.py:184-190: g[id][:] /= |g[id]| + eps for every id of
 * the int64 device list (each distinct id once) */
int rg_normalize_listed_rows(float* g, const void* ids, int n_ids, int rows, int D, float eps, rg_stream_t stream);

/* mean(f(a + b*x)), f = ReLU when clamp != 0 else identity: hinge / wgangp / generator branches of the dual_gan GANLoss,
 * CC/dual_gan/models/external_function.py:58-68; workspace as rg_loss_workspace() */
int rg_affine_relu_mean_fwd(const float* x, float* loss, int64_t n, float a, float b, int clamp, void* workspace,
                            size_t workspace_bytes, rg_stream_t stream);
int rg_affine_relu_mean_bwd(const float* x, const float* grad_out, float* dx, int64_t n, float a, float b, int clamp,
                            float grad_scale, rg_stream_t stream);

/* ---- FP8 convolution family (BASELINE config 5: "dual_gan two-generator path, fp8 MFMA convs") ------------------
 * Replaces, at reduced precision, the cuDNN work behind the nn.Conv2d / nn.ConvTranspose2d layers of the dual_gan
 * generators and discriminator (CC/dual_gan/models/base_function.py:236-443, networks.py:165-275,917-955), which the
 * reference runs in fp32.  OCP e4m3 for activations and filters, e5m2 for gradients, one scale per tensor, fp32
 * accumulation on v_mfma_f32_32x32x16_{fp8,bf8}_{fp8,bf8}; outputs are fp32 NCHW like every other entry point.
 *
 * Scaling state of one tensor: float[4] device memory {amax in use, amax being collected, dequantisation scale = amax in
 * use / format max, format max (448 e4m3 / 57344 e5m2; written once by the caller)}.  rg_f8_quantize scales by
 * format max / state[0], clamps, converts and collects max|x| into state[1]; rg_f8_roll_scales makes the collected
 * values current for `count` consecutive states (delayed scaling: once per step); rg_f8_amax followed by a roll gives
 * just-in-time scaling.  fmt: 0 = e4m3, 1 = e5m2. */
int rg_f8_amax(const float* x, int64_t n, float* state, rg_stream_t stream);
int rg_f8_roll_scales(float* states, int count, rg_stream_t stream);
/* out[b][l][r] (one byte each, r padded with zeros to a multiple of 16) = fp8(in[b*bs + r*rs + l]): the transposing
 * quantiser behind every operand layout; scale_out (one device float, may be NULL) receives the dequantisation scale this
 * call used, which is what the GEMMs below take as sx / sw / sdy (a layer's state may be re-scaled for another tensor
 * before the backward pass reads this one) —  activations [N][C][HW] -> [N][HW][Cp] (b = n, r = c) for the forward / data
 * gradient and -> [C][HW][Np] (b = c, r = n) for the weight gradient; filters [K][C][RS] -> [K][RS][Cp] and [C][RS][Kp] */
int rg_f8_quantize(const float* in, void* out, float* state, float* scale_out, int fmt, int B, int R, int L, int64_t bs,
                   int64_t rs, rg_stream_t stream);
/* both layouts of in[N][C][L] from ONE pass over the fp32 data: a [N][L][Cp] and b [C][L][Np] (either may be NULL) */
int rg_f8_quantize_dual(const float* in, void* a, void* b, float* state, float* scale_out, int fmt, int N, int C, int L,
                        rg_stream_t stream);
/* the output-gradient operand of a convolution backward in one pass over dy: g = dy * act'(yact) (act 0: g = dy, yact unused) is
 * quantised into a / b as rg_f8_quantize_dual does (either may be NULL) and its per-tile channel sums go to
 * part[rg_f8_grad_tiles(N, L)][C] (NULL: not wanted; the bias gradient is the column sum of `part`, rg_rows_sum_pair) — replaces
 * rg_act_bwd + rg_f8_quantize_dual + the channel-sum passes of the reference's autograd (torch Conv2d backward: grad_bias =
 * grad_output.sum((0, 2, 3))) on the fp8 path; g itself is never written */
int rg_f8_grad_tiles(int N, int L);
int rg_f8_quantize_grad(const float* dy, const float* yact, int act, float slope, void* a, void* b, float* part, float* state,
                        float* scale_out, int fmt, int N, int C, int L, rg_stream_t stream);
/* y = act(sx*sw * conv(xq, wq) + shift[k] + residual); xq [N][H*W][Cp], wq [K][KH*KW][Cp] (e4m3), sx / sw: the quantiser's scale_out of each operand */
int rg_conv2d_f8_fwd(const void* xq, const void* wq, const float* sx, const float* sw, int fmt_x, float* y, int N, int C,
                     int H, int W, int K, int KH, int KW, int SH, int SW, int PH, int PW, int P, int Q, const float* shift,
                     const float* residual, int act, float slope, rg_stream_t stream);
/* dx (or the forward of ConvTranspose2d) from dyq [N][P*Q][Kp] and the filters as [C][KH*KW][Kp]; stride <= 2 */
int rg_conv2d_f8_dgrad(const void* dyq, const void* wq_t, const float* sdy, const float* sw, int fmt_dy, float* dx, int N,
                       int C, int H, int W, int K, int KH, int KW, int SH, int SW, int PH, int PW, int P, int Q,
                       const float* shift, const float* residual, int act, float slope, rg_stream_t stream);
/* dw[K][C][KH][KW] from the batch-contiguous layouts xq [C][H*W][Np], dyq [K][P*Q][Np]; split over the pixels, partials in
 * the workspace, summed in fixed order */
size_t rg_conv2d_f8_wgrad_workspace(int N, int C, int K, int KH, int KW, int P, int Q);
int rg_conv2d_f8_wgrad(const void* xq, const void* dyq, const float* sx, const float* sdy, int fmt_x, int fmt_dy, float* dw,
                       int N, int C, int H, int W, int K, int KH, int KW, int SH, int SW, int PH, int PW, int P, int Q,
                       void* workspace, size_t workspace_bytes, rg_stream_t stream);

/* WGAN-GP penalty terms of cal_gradient_penalty, CC/dual_gan/models/external_function.py:100-101, per sample row r of
 * grads[rows][D] with g = grads[r] + 1e-16: pen[r] = scale * (|g|_2 - constant)^2, v[r][:] = d pen[r] / d grads[r] */
int rg_grad_penalty_rows(const float* grads, float* pen, float* v, int rows, int D, float constant, float scale,
                         rg_stream_t stream);

/* ---- pseudo-labelling front half / evaluation distances ------------------------------ */
/* faiss IndexFlatIP.search of CC/clustercontrast/utils/infomap_cluster.py:51-78 on a similarity block s[rows][cols]
 * (from the GEMM): the k best columns per row in (value descending, index ascending) order; idx int32 [rows][k] */
int rg_topk_rows(const float* s, int rows, int cols, int k, int* idx, float* val, rg_stream_t stream);
/* |x_r|^2 per row and m = alpha*m + a*rowv[r] + b*colv[c]: pairwise_distance, CC/clustercontrast/evaluators.py:71-88,
 * FD/reid/evaluators.py:76-98 (the -2 x.y^T term comes from the GEMM) */
int rg_row_sqsum(const float* x, float* out, int rows, int D, rg_stream_t stream);
int rg_add_outer_terms(float* m, const float* rowv, const float* colv, float alpha, float a, float b, int rows, int cols,
                       rg_stream_t stream);
/* generate_cluster_features, CC/examples/cluster_contrast_gan_train_usl_infomap.py:332-348: out[s] = mean of the rows
 * x[order[j]], j in [offsets[s], offsets[s+1]) (int64 device arrays; members in list order) */
int rg_segment_mean(const float* x, const void* order, const void* offsets, float* out, int segments, int D,
                    rg_stream_t stream);

/* ---- conv + frozen-statistics BatchNorm fold (E / D_id of FD-GAN: set_bn_fix, FD/fdgan/networks.py:57-60 with
 * trainable affine parameters, model.py:72-85).  Forward: rg_conv2d_fwd with scale/shift from rg_bn_fold — the
 * pre-normalisation tensor is never written.  Backward without it:
 *   g = dy * act'(y), sum_g = channel sums (dbeta)                     rg_act_bwd_sum (g may be NULL for act none)
 *   G = rg_conv2d_wgrad(x, g);  dgamma = invstd * (sum_m W[k][m] G[k][m] - mean * sum_g);  G *= scale (in place -> dW)
 *                                                                      rg_bn_fold_wgrad (dgamma may be NULL)
 *   dx = rg_conv2d_dgrad(g, rows of W scaled by scale)                 rg_scale_rows */
int rg_bn_fold(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps,
               float* scale, float* shift, float* invstd, int C, rg_stream_t stream);
int rg_act_bwd_sum(const float* dy, const float* y_act, float* g, float* sum_g, int N, int C, int HW, int act, float slope,
                   void* workspace, size_t workspace_bytes, rg_stream_t stream);
/* sums either as sum_g[K] (rg_act_bwd_sum) or as slice partials [K][n_slices] of rg_act_bwd_partial, which
 * rg_bn_fold_wgrad then adds up itself and also writes as dbeta (saves the finalize launch on the critical stream) */
int rg_bn_fold_wgrad(const float* w, float* g, const float* scale, const float* invstd, const float* running_mean,
                     const float* sum_g, const float* partials, int n_slices, float* dbeta, float* dgamma, int K, int M,
                     rg_stream_t stream);
/* rg_conv2d_wgrad + rg_bn_fold_wgrad in one call (same arguments): with split-K the fold finishes inside the reduction launch
 * (one launch, one pass over G less per folded layer) */
int rg_conv2d_wgrad_fold(const float* x, const float* dy, float* dw, int N, int C, int H, int W, int K, int KH, int KW,
                         int SH, int SW, int PH, int PW, int P, int Q, const float* w, const float* scale,
                         const float* invstd, const float* running_mean, const float* sum_g, const float* partials,
                         int n_slices, float* dbeta, float* dgamma, void* workspace, size_t workspace_bytes,
                         rg_stream_t stream);
int rg_bn_slices(int N, int C, int HW);
int rg_act_bwd_partial(const float* dy, const float* y_act, float* g, float* part, int N, int C, int HW, int act, float slope,
                       rg_stream_t stream);
int rg_scale_rows(const float* w, const float* scale, float* out, int K, int M, rg_stream_t stream);

/* All (conv, frozen BatchNorm) pairs of a network in ONE launch: `table` is a device array of 16 int64 words per pair
 * {w, gamma, beta, running_mean, running_var, w_scaled, w_scaled_krsc|0, scale, shift, invstd, K, C, KH*KW,
 *  eps (float bits), first workgroup of the pair, 0}; a pair takes ceil(K*C*KH*KW / rg_fold_chunk()) workgroups.
 * Writes scale/shift/invstd and the filters times scale[k] in [K][C][RS] and (optionally) [K][RS][C] layout, so that the
 * forward is conv(x, W*scale) + shift and the data gradient uses the same scaled filters. */
int rg_fold_filters_multi(const void* table, int n_pairs, int total_blocks, rg_stream_t stream);
int rg_fold_chunk(void);

/* ---- dual_gan blocks (CC/dual_gan/models/base_function.py, PTM.py) ---------------------------- */
/* nn.AvgPool2d(k, k) of the ResBlockEncoder shortcuts, base_function.py:372-420; P = H / k, Q = W / k */
int rg_avgpool2d_fwd(const float* x, float* y, int N, int C, int H, int W, int k, rg_stream_t stream);
int rg_avgpool2d_bwd(const float* dy, float* dx, int N, int C, int H, int W, int k, rg_stream_t stream);
/* nn.ReflectionPad2d(pad) of the Output block, base_function.py:423-443; y is [N,C,H+2pad,W+2pad] */
/* act (0 none, 1 relu, 2 leaky relu with slope > 0): the element-wise activation the Output block applies in front of the padding
 * (base_function.py:436-441 `nonlinearity, ReflectionPad2d, conv`), folded in: y = pad(act(x)), dx = act'(x_act) * pad^T(dy) with
 * x_act the tensor the forward padded (NULL when act == 0); the fused forms need W % 4 == 0, W >= 8, pad <= 3 */
int rg_reflection_pad2d_fwd(const float* x, float* y, int N, int C, int H, int W, int pad, int act, float slope, rg_stream_t stream);
int rg_reflection_pad2d_bwd(const float* dy, const float* x_act, float* dx, int N, int C, int H, int W, int pad, int act, float slope,
                            rg_stream_t stream);
/* torch.nn.utils.spectral_norm (base_function.py:121-126; every ResDiscriminator conv, networks.py:917-955) on the
 * filter viewed as W[K][M]: training != 0 runs one power iteration in place on u[K], v[M] (eps-clamped norms), then
 * sigma = u.(W v); writes w_sn = W / sigma and sigma[0] = sigma, sigma[1] = 1 / sigma (device, 2 floats); uv_saved (may be NULL,
 * K + M floats) receives the u, v this forward used — the constants of its backward, which the next forward overwrites.
 * K <= 1024, M <= 12288.  Backward: dw (+)= (dw_sn - (sum dw_sn * w_sn) u v^T) / sigma with the forward's u, v. */
int rg_spectral_norm_fwd(const float* w, float* u, float* v, float* w_sn, float* sigma, float* uv_saved, int K, int M,
                         int training, float eps, rg_stream_t stream);
/* The same for every spectral-normed filter of one network forward (ResDiscriminator: 13 filters) in two launches. */
#define RG_SN_MAX_BATCH 16
typedef struct rg_sn_desc {
    const float* w;
    float* u;
    float* v;
    float* w_sn;
    float* sigma;
    float* uv_saved;
    int K;
    int M;
} rg_sn_desc;
int rg_spectral_norm_fwd_multi(const rg_sn_desc* descs, int count, int training, float eps, rg_stream_t stream);
size_t rg_spectral_norm_bwd_workspace(int K, int M);
int rg_spectral_norm_bwd(const float* dw_sn, const float* w_sn, const float* u, const float* v, const float* sigma,
                         float* dw, int K, int M, int accumulate, void* workspace, size_t workspace_bytes,
                         rg_stream_t stream);
/* Batched strided fp32 GEMM (MFMA) for nn.MultiheadAttention inside CAB / TTB, PTM.py:162-247:
 * C[b0][b1][m][n] = alpha * sum_k A[b0][b1][m][k] B[b0][b1][k][n] + beta * C; all strides in elements, so Q^T K,
 * P V and the backward products run on [B][C][L] token maps without permutes. */
int rg_bgemm(const float* A, const float* B, float* C, int M, int N, int K, int64_t a_ms, int64_t a_ks, int64_t b_ks,
             int64_t b_ns, int64_t c_ms, int64_t c_ns, int batch0, int batch1, int64_t a_b0, int64_t a_b1, int64_t b_b0,
             int64_t b_b1, int64_t c_b0, int64_t c_b1, float alpha, float beta, rg_stream_t stream);
/* y[r][:] = softmax(scale * x[r][:]) and ds = scale * p * (dp - sum(dp * p)); in place allowed */
int rg_softmax_rows_fwd(const float* x, float* y, int rows, int cols, float scale, rg_stream_t stream);
int rg_softmax_rows_bwd(const float* p, const float* dp, float* ds, int rows, int cols, float scale, rg_stream_t stream);

/* AEModel.hard_mix, CC/dual_gan/models/AE_model.py:274-292: out[j] = lam * src[idx_a[j]] + (1 - lam) * src[idx_b[j]] over rows
 * of `len` floats; idx_* are int64 device arrays of rows_out entries.  Backward: dsrc[r] = sum over the rows that read r. */
int rg_mix_rows_fwd(const float* src, const void* idx_a, const void* idx_b, float lam, float* out, int rows_src,
                    int rows_out, int64_t len, rg_stream_t stream);
int rg_mix_rows_bwd(const float* g, const void* idx_a, const void* idx_b, float lam, float* dsrc, int rows_src, int rows_out,
                    int64_t len, rg_stream_t stream);

/* ---- pooling -------------------------------------------------------------------------------- */
int rg_maxpool2d_fwd(const float* x, float* y, unsigned char* argmax, int N, int C, int H, int W, int KH, int KW,
                     int SH, int SW, int PH, int PW, int P, int Q, rg_stream_t stream);
int rg_maxpool2d_bwd(const float* dy, const unsigned char* argmax, float* dx, int N, int C, int H, int W, int KH,
                     int KW, int SH, int SW, int PH, int PW, int P, int Q, rg_stream_t stream);
/* F.avg_pool2d(x, x.size()[2:]), FD/reid/models/resnet.py:71 */
int rg_global_avgpool_fwd(const float* x, float* y, int N, int C, int HW, rg_stream_t stream);
int rg_global_avgpool_bwd(const float* dy, float* dx, int N, int C, int HW, rg_stream_t stream);
/* GeneralizedMeanPoolingP, CC/clustercontrast/models/pooling.py:57-103 */
int rg_gem_pool_fwd(const float* x, const float* p, float* y, int N, int C, int HW, float eps, rg_stream_t stream);
int rg_gem_pool_bwd(const float* x, const float* p, const float* y, const float* dy, float* dx, float* dp, int N,
                    int C, int HW, float eps, void* workspace, size_t workspace_bytes, rg_stream_t stream);

/* ---- losses (scalar outputs live in device memory; grad_out is a 1-element device tensor or NULL) */
size_t rg_loss_workspace(void);
/* GANLoss: sigmoid + BCE vs constant target, FD/fdgan/losses.py:29-32 */
int rg_sigmoid_bce_fwd(const float* x, float* loss, int64_t n, float target, void* workspace, size_t workspace_bytes,
                       rg_stream_t stream);
int rg_sigmoid_bce_bwd(const float* x, const float* grad_out, float* dx, int64_t n, float target, float grad_scale,
                       rg_stream_t stream);
/* lsgan: MSE vs constant label, CC/dual_gan/models/external_function.py:53-57 */
int rg_mse_const_fwd(const float* x, float* loss, int64_t n, float target, void* workspace, size_t workspace_bytes,
                     rg_stream_t stream);
int rg_mse_const_bwd(const float* x, const float* grad_out, float* dx, int64_t n, float target, float grad_scale,
                     rg_stream_t stream);
/* F.l1_loss, optionally over the rows with row_labels == 1 (same-identity pairs), FD/fdgan/model.py:190-194.
 * out2[0] = loss, out2[1] = 1/(number of selected elements). */
int rg_l1_fwd(const float* a, const float* b, const int64_t* row_labels, float* out2, int rows, int64_t inner,
              void* workspace, size_t workspace_bytes, rg_stream_t stream);
int rg_l1_bwd(const float* a, const float* b, const int64_t* row_labels, const float* grad_out, const float* out2,
              float* da, float* db, int rows, int64_t inner, float grad_scale, rg_stream_t stream);
/* per-sample form: out[r] = mean_i |a[r][i] - b[r][i]| — nn.L1Loss(reduction='none')(a, b).flatten(1).mean(-1), the `loss_rec` of
 * AEModel.get_loss_G(need_cm=True), CC/dual_gan/models/AE_model.py:366 — and its backward from the per-row cotangent */
int rg_l1_rows_fwd(const float* a, const float* b, float* out, int rows, int64_t inner, rg_stream_t stream);
int rg_l1_rows_bwd(const float* a, const float* b, const float* grad_rows, float* da, float* db, int rows, int64_t inner,
                   rg_stream_t stream);
/* out[r] = mean_i (x[r][i] - c)^2: GANLoss('lsgan')(D(fake), label)'s map averaged per sample, AE_model.py:378-384 get_L1_loss(with_dis) */
int rg_mse_const_rows_fwd(const float* x, float c, float* out, int rows, int64_t inner, rg_stream_t stream);
int rg_mse_const_rows_bwd(const float* x, float c, const float* grad_rows, float* dx, int rows, int64_t inner, rg_stream_t stream);
/* F.cross_entropy(scale*logits, labels, reduction='none'), FD/fdgan/model.py:189, CC/.../cm.py:134-135 */
int rg_softmax_ce_fwd(const float* logits, const int64_t* labels, float* loss_rows, float* lse, int B, int K,
                      float scale, rg_stream_t stream);
int rg_softmax_ce_bwd(const float* logits, const int64_t* labels, const float* lse, const float* grad_rows,
                      float* dlogits, int B, int K, float scale, float grad_scale, rg_stream_t stream);
/* out[0] = scale * sum_i x[i]*w[i] (w may be NULL): .mean() / conf_mask weighting of per-sample losses */
int rg_weighted_sum_fwd(const float* x, const float* w, float* out, int64_t n, float scale, rg_stream_t stream);
int rg_weighted_sum_bwd(const float* grad_out, const float* w, float* dx, int64_t n, float scale, rg_stream_t stream);

/* ---- ClusterMemory momentum update, CC/clustercontrast/models/cm.py:29-31 (CM), :57-70 (CM_Hard),
 * :100-104 (CM_gan second bank: normalize_eps = 1).  In place on features[K][D], batch order kept. */
int rg_cm_update(const float* inputs, const int64_t* targets, float* features, int B, int D, int K, float momentum,
                 int normalize_eps, rg_stream_t stream);
int rg_cm_update_hard(const float* inputs, const int64_t* targets, float* features, int B, int D, int K,
                      float momentum, rg_stream_t stream);

/* ---- fused optimizer steps on contiguous ranges (FD/fdgan/model.py:100-125; CC/examples/
 * cluster_contrast_gan_train_usl_infomap.py:281-284). grad_scale multiplies g first (1/world for DDP). */
int rg_adam_step(float* p, const float* g, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                 float beta2, float eps, float weight_decay, int step, float grad_scale, rg_stream_t stream);
int rg_sgd_step(float* p, const float* g, float* momentum_buf, int64_t n, float lr, float momentum,
                float weight_decay, int first_step, float grad_scale, rg_stream_t stream);
/* Device-side clocks for hipGraph replay (kernel arguments are frozen at capture, so per-step values live in device memory):
 * Adam state double[4] = {step, beta1^step, beta2^step, -} (initialise {0, 1, 1, 0}); rg_adam_advance once per optimizer step,
 * then rg_adam_step_dev per parameter range; rg_u64_add advances a step counter that rg_dropout_clocked mixes into its seed. */
int rg_adam_advance(double* state, float beta1, float beta2, rg_stream_t stream);
int rg_adam_step_dev(float* p, const float* g, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                     float eps, float weight_decay, const double* state, float grad_scale, rg_stream_t stream);
int rg_u64_add(unsigned long long* p, unsigned long long v, rg_stream_t stream);

/* ---- on-device input synthesis: the reference's per-sample PIL / scipy steps as batch kernels ---- */
/* Pose heat maps out[N][J][H][W] from integer joint centres centers[N][J][2] = (row, col); a negative (or out-of-range)
 * coordinate marks a missing / erased joint -> zero map.  sigma[N]: one Gaussian width per sample.
 * mode 0: FD/reid/utils/data/preprocessor.py:114-131 (_generate_pose_map) — unit impulse -> scipy gaussian_filter(sigma,
 *         truncate 4, mode 'reflect') -> divided by the map maximum (float64 arithmetic, float32 result);
 * mode 1: CC/clustercontrast/utils/data/pose_utils.py:51-70 (cords_to_map) — exp(-((y-r)^2 + (x-c)^2) / (2 sigma^2)); centres
 *         outside the image are evaluated like any other (the reference does), only INT32_MIN marks a missing joint.
 * The random choices (erased joint, sigma) are drawn on the host in the reference's order and arrive as inputs. */
int rg_pose_maps(const int* centers, const float* sigma, float* out, int N, int J, int H, int W, int mode,
                 rg_stream_t stream);
/* out[n][c][y][x] = P[n][c][top + y][left + (flip ? W-1-x : x)] with P = x[N][C][Hs][Ws] padded by `pad` pixels of
 * pad_value[c] (NULL: 0) and params[N][3] = (flip, top, left): torchvision's Pad(pad) + RandomCrop((H, W)) +
 * RandomHorizontalFlip of CC/examples/cluster_contrast_gan_train_usl_infomap.py:110-119, and np.flip(maps, 2) of
 * FD/reid/utils/data/preprocessor.py:88-91 (pad 0, top = left = 0). */
int rg_flip_pad_crop(const float* x, const int* params, const float* pad_value, float* out, int N, int C, int Hs, int Ws,
                     int H, int W, int pad, rg_stream_t stream);
/* RandomErasing, CC/clustercontrast/utils/data/transforms.py:52-96: x[n][c][r0:r0+h][c0:c0+w] = fill[c] in place for
 * rects[N][4] = (r0, c0, h, w); h = 0 leaves sample n untouched; a NaN fill[c] leaves channel c untouched (the reference
 * erases only channel 0 of non-RGB inputs). */
int rg_erase_rects(float* x, const int* rects, const float* fill, int N, int C, int H, int W, rg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* REIDGAN_HIP_H */
