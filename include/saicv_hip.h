/*
 * libsaicv_hip.so -- C-ABI of the MI355X (gfx950) kernels behind the SimpleAICV DDP
 * forward/backward hot path.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer borrowed for the call (allocated by the caller, e.g.
 *    PyTorch's caching allocator); the library allocates nothing and frees nothing;
 *  - `stream` is a hipStream_t passed as void*; every launch goes on it, no implicit syncs,
 *    so calls are capturable in a hipGraph and safe on the communication side stream;
 *  - return 0 on success, negative on error; the message is in saicv_last_error_string()
 *    (thread local).  Nothing throws across the ABI;
 *  - dtype: SAICV_BF16 = perf mode (bf16 storage, fp32 accumulate / statistics),
 *           SAICV_F32  = parity mode (fp32 everywhere, exact-f32 MFMA);
 *  - activations are dense NHWC ([N,H,W,C], C fastest); conv weights are "KRSC"
 *    ([Cout][R][S][Cin], Cin fastest); C and Cout must be multiples of 8 (bf16) / 4 (f32).
 *
 * Each entry point names the reference call it replaces (paths relative to the reference
 * repository zgcr/SimpleAICV_pytorch_training_examples).
 */
#ifndef SAICV_HIP_H
#define SAICV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAICV_BF16 0
#define SAICV_F32 1

typedef struct saicv_conv_desc {
    int N, H, W, C;      /* input  [N,H,W,C]                       */
    int K, R, S;         /* weight [K][R][S][C]                    */
    int stride, pad;
    int OH, OW;          /* output [N,OH,OW,K]                     */
    int dtype;           /* SAICV_BF16 | SAICV_F32                 */
} saicv_conv_desc;

int saicv_version(void);
const char* saicv_last_error_string(void);

/* ---- deterministic mode ----------------------------------------------------------------
 * Replaces `torch.backends.cudnn.deterministic = True` of the reference's set_seed (tools/utils.py:95-107).
 * on != 0: every reduction that otherwise adds workgroup partials with fp32 atomics (weight / bias gradients of the
 * convolution, linear and depthwise kernels, BatchNorm / GroupNorm statistics of non-convolution inputs, the stem's fused
 * BatchNorm backward sums, relative-position table gradients, mask-loss sums, focal / SmoothL1 sums, the gradient norm)
 * parks its partials in a library-owned workspace and folds them in a fixed order: results are bit-identical from run to run
 * for the same shapes.  The workspace (96 MiB per stream, doubling on demand) is the one exception to "the library allocates
 * nothing"; saicv_deterministic_prepare(stream) allocates it up front so that no allocation falls into a hipGraph capture.
 * saicv_set_deterministic returns the previous setting.  Process-global, not thread-local. */
int saicv_set_deterministic(int on);
int saicv_get_deterministic(void);
int saicv_deterministic_prepare(void* stream);

/* ---- layout packing ------------------------------------------------------------------ */
/* NCHW-shaped fp32 batch with element strides (sN,sC,sH,sW) -> dense NHWC [N,H,W,Cp],
 * channels zero-padded.  Input contract: SimpleAICV/classification/common.py:645-665
 * (ClassificationCollater returns an NHWC-strided, NCHW-shaped fp32 tensor). */
int saicv_pack_input(int dtype, const float* src, long sN, long sC, long sH, long sW, void* dst,
                     int N, int C, int H, int W, int Cp, void* stream);
/* fp32 master weight [O,I,R,S] (element strides) -> Wf[O][R][S][Ip] and optionally the
 * data-gradient matrix Wd[I][R][S][Op] (NULL to skip; Op >= O is its leading dimension, rows
 * O..Op-1 of Wf / columns O..Op-1 of Wd are the caller's zero padding).  nn.Conv2d / nn.Linear
 * weights of resnet.py:33-39, :204. */
int saicv_pack_weight(int dtype, const float* w, long sO, long sI, long sR, long sS, int O, int I,
                      int R, int S, int Ip, int Op, void* wf, void* wd, void* stream);
/* fp32 dW[O][R][S][Ip] -> gradient tensor [O,I,R,S] with element strides. */
int saicv_unpack_wgrad(const float* dw, int O, int I, int R, int S, int Ip, float* grad, long sO,
                       long sI, long sR, long sS, int accumulate, void* stream);

/* ---- convolution / linear (implicit GEMM on MFMA) ------------------------------------- */
/* All weights of a model repacked in one launch (after the optimizer step): `descs` is a DEVICE array of n descriptors,
 * each as for saicv_pack_weight plus its first tile index; tiles of one weight = tiles_i * tiles_o * R * S with
 * tiles_i = ceil(Ip / T), tiles_o = ceil(Op / T), T = `tile` (32, or 64 for the vectorised form: requires sI == 1, sO / sR /
 * sS / I / Ip / Op multiples of 4 and 16-byte aligned w / wf / wd); total_tiles = the sum.  wd may be NULL per weight. */
typedef struct saicv_pack_desc {
    const float* w;
    long sO, sI, sR, sS;
    int O, I, R, S, Ip, Op;
    void* wf;
    void* wd;
    int tile_begin, tiles_i, tiles_o, tile;      /* tile: 0 / 32 or 64 */
} saicv_pack_desc;
int saicv_pack_weight_batched(int dtype, const saicv_pack_desc* descs, int n, int total_tiles, void* stream);

/* The stride-2 stem as a stride-1 convolution over a space-to-depth image (reference resnet.py:172-174, `conv1` = 7x7,
 * stride 2, padding 3 on 3 channels): dst[N][Hq][Wq][Cq], Hq = (H + 2 pad + 1) / 2, channel (a, b, c) of pixel (u, v) =
 * x[2u + a - pad][2v + b - pad][c] (zero outside the image, channels 4C..Cq zero); the weights regroup the same way into
 * Wf[O][(R+1)/2][(S+1)/2][Cq]; saicv_conv2d_fwd / _wgrad then run with H = Hq, W = Wq, C = Cq, R = (R+1)/2, stride 1, pad 0
 * (GEMM K 392 -> 256 for the ResNet stem) and saicv_unpack_wgrad_s2d scatters dW back to the [O, I, R, S] gradient. */
int saicv_pack_input_s2d(int dtype, const float* src, long sN, long sC, long sH, long sW, void* dst, int N, int C, int H,
                         int W, int pad, int Cq, void* stream);
int saicv_pack_weight_s2d(int dtype, const float* w, long sO, long sI, long sR, long sS, int O, int I, int R, int S, int Cq,
                          void* wf, void* stream);
int saicv_unpack_wgrad_s2d(const float* dw, int O, int I, int R, int S, int Cq, float* grad, long sO, long sI, long sR,
                           long sS, int accumulate, void* stream);

/* rows of BN partial statistics saicv_conv2d_fwd writes when stat_sum != NULL */
int saicv_conv2d_stat_rows(const saicv_conv_desc* d);
/* y = conv(x, wf) [+ bias]; optionally per-channel partial sum / sum-of-squares of y
 * ([rows][K] each) for the following BatchNorm.  ATen `convolution` under
 * ConvBnActBlock.forward, resnet.py:45-48.  out_f32: write y as fp32 (logits). */
int saicv_conv2d_fwd(const saicv_conv_desc* d, const void* x, const void* wf, const float* bias,
                     void* y, int out_f32, float* stat_sum, float* stat_sq, void* stream);
/* dx = conv_transpose(dy, w) via Wd[C][R][S][K]   (`convolution_backward`, grad_input) */
int saicv_conv2d_dgrad(const saicv_conv_desc* d, const void* dy, const void* wd, void* dx,
                       void* stream);
/* dw[K][R][S][C] (fp32) += dy^T * im2col(x)       (`convolution_backward`, grad_weight).
 * Accumulates with fp32 atomics: zero dw first for a plain gradient. */
int saicv_conv2d_wgrad(const saicv_conv_desc* d, const void* dy, const void* x, float* dw,
                       void* stream);
/* the same with the bias gradient: dbias[K] (fp32) += column sums of dy, taken from the dY tiles the kernel already holds
 * (the workgroups of the first weight-tile column add them): no separate pass over dy for a biased nn.Conv2d */
int saicv_conv2d_wgrad_bias(const saicv_conv_desc* d, const void* dy, const void* x, float* dw, float* dbias, void* stream);
/* nn.Linear on the same kernels: y[M][N] = addend + row_scale[m / rows_per_scale] * (x[M][K] wf[N][K]^T + bias)
 * (addend / row_scale optional: the residual add and drop-path of vit.py:159-163 fused into the
 * epilogue); dx[M][K] = dy[M][N] wd[K][N]^T (+ addend); dw[N][K] (fp32) += dy^T x and, when dbias
 * is given, dbias[N] (fp32) += column sums of dy, computed from the tiles the kernel streams anyway. */
int saicv_linear_fwd(int dtype, const void* x, const void* wf, const float* bias, void* y, int M, int K, int N,
                     int out_f32, const void* addend, const float* row_scale, int rows_per_scale, void* stream);
int saicv_linear_dgrad(int dtype, const void* dy, const void* wd, void* dx, int M, int K, int N, const void* addend,
                       void* stream);
/* MLP fusions (reference vit.py:91-99, segment_anything/image_encoder.py:187-198):
 * y_pre = x W^T + b and y_act = gelu(y_pre) from ONE GEMM (exact-erf GELU applied to the stored, rounded y_pre);
 * dx = (dy W) * gelu'(pre): the backward of the activation folded into the epilogue of the next layer's dgrad.
 * N (fwd) / K (dgrad) must keep rows 16-byte aligned. */
int saicv_linear_gelu_fwd(int dtype, const void* x, const void* wf, const float* bias, void* y_pre, void* y_act, int M,
                          int K, int N, void* stream);
int saicv_linear_dgrad_gelu(int dtype, const void* dy, const void* wd, const void* pre, void* dx, int M, int K, int N,
                            void* stream);
/* The same pair with the activation's DERIVATIVE stored instead of the pre-activation (which nothing else reads): the forward
 * computes gelu(y) and gelu'(y) from one exp / one rcp anyway, and the backward epilogue becomes one multiply per element instead
 * of ~14 vector operations -- what a one-workgroup-per-CU GEMM cannot hide (profiles/r03_lds_fill_and_kc8.md, section 6).
 * y_dact = gelu'(x W^T + b) in the compute dtype; dx = (dy W) * factor. */
int saicv_linear_gelu_fwd_aux(int dtype, const void* x, const void* wf, const float* bias, void* y_dact, void* y_act, int M,
                              int K, int N, void* stream);
int saicv_linear_dgrad_mul(int dtype, const void* dy, const void* wd, const void* factor, void* dx, int M, int K, int N,
                           void* stream);
int saicv_linear_wgrad(int dtype, const void* dy, const void* x, float* dw, float* dbias, int M, int K, int N,
                       void* stream);
/* conv data-gradient that adds an existing gradient (residual branch) in its epilogue */
int saicv_conv2d_dgrad_add(const saicv_conv_desc* d, const void* dy, const void* wd, const void* addend, void* dx,
                           void* stream);
/* Data gradient of a conv inside a residual network, with what the neighbouring BatchNorm / shortcut nodes need folded
 * into its epilogue (every field optional; all tensors share dx's [N*H*W][C] coordinates):
 *   dx = dgrad(dy, wd) + addend * [addend_gate bit]      -- the shortcut gradient behind the block's final ReLU
 *        (resnet.py:94-95,152-153: `x = x + inputs; x = self.relu(x)`); addend_gate is the mask saicv_bn_act_fwd wrote,
 *        so the masked copy `dres` of saicv_bn_act_bwd is never materialised;
 *   part_g[row][c] = sum g, part_gx[row][c] = sum g * (bn_y - bn_mean) * bn_invstd,  g = dx * [bn_mask bit]
 *        -- the reduction pass of the BatchNorm (+ReLU) that produced this conv's input, whose backward receives dx as
 *        its dz (`native_batch_norm_backward`); rows = saicv_conv2d_dgrad_stat_rows(d), fed to
 *        saicv_bn_act_bwd_from_partials.  bn_mask NULL = no ReLU between that BatchNorm and this conv. */
typedef struct saicv_dgrad_fuse {
    const void* addend;
    const void* addend_gate;
    const void* bn_y;
    const void* bn_mask;
    const float* bn_mean;
    const float* bn_invstd;
    float* part_g;
    float* part_gx;
    int part_rows;      /* 0: one row per tile row (saicv_conv2d_dgrad_stat_rows rows, for saicv_bn_act_bwd_from_partials);
                         * > 0: the sums are ADDED (fp32 atomics) into this many rows of a buffer the caller zeroed, for
                         * saicv_bn_act_bwd_inline */
    int reserved;
} saicv_dgrad_fuse;
int saicv_conv2d_dgrad_stat_rows(const saicv_conv_desc* d);
int saicv_conv2d_dgrad_fused(const saicv_conv_desc* d, const void* dy, const void* wd, const saicv_dgrad_fuse* f, void* dx,
                             void* stream);
/* out[r][:] = x[r][:] * scale[r / rows_per_scale] */
int saicv_row_scale(int dtype, const void* x, const float* scale, void* out, size_t rows, int row_len,
                    int rows_per_scale, void* stream);
/* dbias[N] (fp32) += column sums of dy[M][N] */
int saicv_colsum(int dtype, const void* dy, int M, int N, float* dbias, void* stream);

/* ---- BatchNorm2d + ReLU + residual ---------------------------------------------------- */
size_t saicv_bn_ws_floats(int C);
/* partial sums -> mean/invstd/scale/shift, running-stat update (momentum, unbiased var) and, when
 * num_batches_tracked (device int64) is given, its += 1.  `native_batch_norm` (training) under resnet.py:41. */
int saicv_bn_finalize_fwd(const float* sum, const float* sq, int rows, int C, double count,
                          const float* gamma, const float* beta, float* running_mean,
                          float* running_var, double momentum, double eps, float* mean,
                          float* invstd, float* scale, float* shift, float* ws, long long* num_batches_tracked,
                          void* stream);
/* eval mode: scale/shift from running statistics */
int saicv_bn_eval_coeffs(int C, const float* gamma, const float* beta, const float* running_mean,
                         const float* running_var, double eps, float* scale, float* shift,
                         void* stream);
/* z = [relu](y*scale + shift [+ res])   (resnet.py:41-42, :94-95, :152-153).
 * relu_mask (optional, M*C/8 bytes in bf16, M*C/4 in fp32): one byte per 16-byte chunk of z holding the
 * sign bits [pre-ReLU value > 0] -- everything the backward pass needs from z at 1/16 of its size. */
int saicv_bn_act_fwd(int dtype, const void* y, const void* res, void* z, const float* scale,
                     const float* shift, size_t M, int C, int relu, void* relu_mask, void* stream);
size_t saicv_bn_bwd_ws_floats(size_t M, int C, int dtype);
/* backward of the fused block: g = dz*[z>0]; dy, dres(=g, optional), dgamma, dbeta
 * (accumulate != 0: dgamma/dbeta are added to, e.g. straight into the gradient arena).
 * The ReLU gate comes from relu_mask when given (z may then be NULL), else from z.
 * `threshold_backward` + `native_batch_norm_backward` (+ residual `add` backward). */
int saicv_bn_act_bwd(int dtype, const void* dz, const void* z, const void* relu_mask, const void* y, const float* gamma,
                     const float* mean, const float* invstd, void* dy, void* dres, float* dgamma,
                     float* dbeta, size_t M, int C, int relu, int accumulate, float* ws,
                     void* stream);

/* BatchNorm statistics without the partial-reduce / finalize launches: the convolution ADDS its per-tile sums (fp32 atomics)
 * into stat_rows (1..64) rows of [stat_rows][K] buffers the caller zeroed (row = tile row mod stat_rows), and
 * saicv_bn_act_fwd_stats derives mean / invstd / scale / shift from them in every workgroup (K <= 2048), applies them, and --
 * workgroup 0 -- stores mean / invstd and updates running statistics and num_batches_tracked as saicv_bn_finalize_fwd does.
 * The summation order of the atomics is not fixed: statistics differ from run to run in the last bit or two. */
int saicv_conv2d_fwd_stats(const saicv_conv_desc* d, const void* x, const void* wf, void* y, float* stat_sum, float* stat_sq,
                           int stat_rows, void* stream);
int saicv_bn_act_fwd_stats(int dtype, const void* y, const void* res, void* z, const float* stat_sum, const float* stat_sq,
                           int stat_rows, double count, const float* gamma, const float* beta, float* running_mean,
                           float* running_var, double momentum, double eps, long long* num_batches_tracked, float* mean,
                           float* invstd, size_t M, int C, int relu, void* relu_mask, void* stream);
/* The residual join of a block whose shortcut is convolution + BatchNorm (reference resnet.py:90-95, :148-153: `identity =
 * self.downsample_conv(inputs); x = x + identity; x = self.relu(x)`): res is the shortcut convolution's RAW output and its
 * BatchNorm-apply (res_scale[c] * res + res_shift[c], from saicv_bn_finalize_fwd / saicv_bn_eval_coeffs) happens in this pass, the
 * normalised shortcut is never written.  stat_sum != NULL: the main branch's statistics as in saicv_bn_act_fwd_stats (scale /
 * shift unused); stat_sum == NULL: scale / shift given as in saicv_bn_act_fwd (the statistics arguments unused). */
int saicv_bn_act_fwd_join(int dtype, const void* y, const void* res, const float* res_scale, const float* res_shift, void* z,
                          const float* scale, const float* shift, const float* stat_sum, const float* stat_sq, int stat_rows,
                          double count, const float* gamma, const float* beta, float* running_mean, float* running_var,
                          double momentum, double eps, long long* num_batches_tracked, float* mean, float* invstd, size_t M, int C,
                          int relu, void* relu_mask, void* stream);
/* backward counterpart: part_g / part_gx are `rows` atomically accumulated rows (saicv_conv2d_dgrad_fused with part_rows > 0);
 * coefficients, dgamma and dbeta come out of the one streaming kernel */
int saicv_bn_act_bwd_inline(int dtype, const void* dz, const void* relu_mask, const void* y, const float* gamma,
                            const float* mean, const float* invstd, const float* part_g, const float* part_gx, int rows,
                            void* dy, void* dres, float* dgamma, float* dbeta, size_t M, int C, int relu, int accumulate,
                            void* stream);
/* the same with the reduction pass already done by saicv_conv2d_dgrad_fused (part_g / part_gx, `rows` rows of C);
 * ws as for saicv_bn_act_bwd */
int saicv_bn_act_bwd_from_partials(int dtype, const void* dz, const void* relu_mask, const void* y, const float* gamma,
                                   const float* mean, const float* invstd, const float* part_g, const float* part_gx,
                                   int rows, void* dy, void* dres, float* dgamma, float* dbeta, size_t M, int C, int relu,
                                   int accumulate, float* ws, void* stream);

/* ---- pooling --------------------------------------------------------------------------- */
/* nn.MaxPool2d(3,2,1), resnet.py:184; idx = window position of the first maximum */
int saicv_maxpool_fwd(int dtype, const void* x, void* out, uint8_t* idx, int N, int H, int W, int C,
                      int OH, int OW, int K, int stride, int pad, void* stream);
int saicv_maxpool_bwd(int dtype, const void* dout, const uint8_t* idx, void* dx, int N, int H, int W,
                      int C, int OH, int OW, int K, int stride, int pad, void* stream);
/* BatchNorm-apply + ReLU + MaxPool2d as ONE pass over the convolution output y (the ResNet stem: ConvBnActBlock followed by
 * nn.MaxPool2d(3, 2, 1), reference classification/backbones/resnet.py:172-184, 226-229; same in detection/models/backbones/
 * detr_resnet.py): out = maxpool(relu(scale[c] * y + shift[c])), idx = window position of the first maximum (as saicv_maxpool_fwd).
 * The full-resolution activation and its ReLU mask are never written.  C / (16 / element size) must divide 256. */
int saicv_bn_relu_maxpool_fwd(int dtype, const void* y, const float* scale, const float* shift, void* out, uint8_t* idx, int N, int H,
                              int W, int C, int OH, int OW, int K, int stride, int pad, void* stream);
/* Its backward, fused with the BatchNorm backward: from dout (gradient of the pooled tensor) and idx the gradient reaching the
 * BatchNorm output is rebuilt per input pixel (gather form, ReLU gate from scale * y + shift > 0), reduced per channel (ws: 2 * C
 * floats, zeroed here) and turned into dy = gamma * invstd * (g - mean(g) - xhat * mean(g * xhat)); dgamma / dbeta written
 * (accumulate = 0) or added to (1).  Training statistics only (mean / invstd of THIS batch); K <= 2 * stride + 1. */
size_t saicv_bn_relu_maxpool_bwd_ws_floats(int C);
int saicv_bn_relu_maxpool_bwd(int dtype, const void* dout, const uint8_t* idx, const void* y, const float* gamma, const float* mean,
                              const float* invstd, const float* scale, const float* shift, void* dy, float* dgamma, float* dbeta,
                              int accumulate, float* ws, int N, int H, int W, int C, int OH, int OW, int K, int stride, int pad,
                              void* stream);
/* nn.AdaptiveAvgPool2d((1,1)), resnet.py:203 */
int saicv_avgpool_fwd(int dtype, const void* x, void* out, int N, int HW, int C, void* stream);
int saicv_avgpool_bwd(int dtype, const void* dout, void* dx, int N, int HW, int C, void* stream);

/* ---- loss ------------------------------------------------------------------------------ */
/* CELoss (soft=0, label int64[B]) / OneHotLabelCELoss (soft=1, label fp32[B][C]);
 * SimpleAICV/classification/losses.py:21-28, :86-91.  loss[0] = mean; dlogits (optional) =
 * d loss / d logits for upstream gradient 1. */
int saicv_softmax_ce_fwd(const float* logits, const void* label, int soft, int B, int C,
                         float* row_loss, float* loss, float* dlogits, void* stream);
/* out = in * scale[0] (device scalar), out dtype selectable */
int saicv_scale_by_scalar(int out_dtype, const float* in, const float* scale, void* out, size_t n,
                          void* stream);

/* ---- flat-arena optimizers / GradScaler (tools/utils.py:292-679, :199-200) ------------- */
/* One launch over the flat arenas; every 1024-element block belongs to one parameter.
 *   block_group[b]  optimizer param-group of block b (-1: not optimized)
 *   hyper[g*8..]    lr, weight_decay, momentum|beta1, beta2, eps, 1-beta1, 1-beta2, nesterov flag
 *   found_inf       nullable device flag: != 0 skips the whole step (GradScaler semantics)
 *   has_grad        nullable, one byte per block: 0 = this parameter received no gradient this step and is
 *                   skipped like torch.optim skips `grad is None` (no decay, no moment update, no step count)
 *   step_blk        AdamW: per-block step counters (= torch's per-parameter state['step']), advanced on the
 *                   device only when the block is really updated; bias corrections are computed from them */
int saicv_sgd_flat(float* p, const float* g, float* mom, const int32_t* block_group,
                   const float* hyper, const float* inv_scale, const float* found_inf,
                   const uint8_t* has_grad, size_t n, void* stream);
int saicv_adamw_flat(float* p, const float* g, float* m, float* v, const int32_t* block_group,
                     const float* hyper, const float* inv_scale, const float* found_inf,
                     const uint8_t* has_grad, float* step_blk, size_t n, void* stream);
int saicv_grad_stats(const float* g, size_t n, float* found_inf, float* sumsq, void* stream);
int saicv_grad_clip_scale(float* g, size_t n, const float* sumsq, const float* inv_scale,
                          double max_norm, void* stream);
/* torch.nn.utils.clip_grad_value_ on the flat gradient arena with the GradScaler unscale folded in:
 * g = clamp(g * inv_scale, -value, value)  (reference tools/scripts.py:211-218). */
int saicv_grad_clip_value(float* g, size_t n, const float* inv_scale, double value, void* stream);
int saicv_scaler_update(float* state, const float* found_inf, double growth, double backoff,
                        int interval, void* stream);

/* ---- transformer blocks (SimpleAICV/classification/backbones/vit.py) -------------------- */
/* nn.LayerNorm over the last dim C of x[M][C]; saves mean / rstd (fp32 [M]).  vit.py:147,151,225 */
int saicv_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean,
                        float* rstd, int M, int C, double eps, void* stream);
size_t saicv_layernorm_bwd_ws_floats(int M, int C);
/* dx (= addend + LN backward; addend optional: the residual-stream gradient of a pre-LN block),
 * dgamma, dbeta (accumulate != 0: added to) */
int saicv_layernorm_bwd(int dtype, const void* dy, const void* x, const float* gamma, const float* mean,
                        const float* rstd, const void* addend, void* dx, float* dgamma, float* dbeta, float* ws,
                        int M, int C, int accumulate, void* stream);
/* The same with a second output dx_scaled[row] = out_scale[row / rows_per_scale] * dx[row] (the factor applied to the stored, rounded
 * gradient): the drop-path branch that produced this LayerNorm's input consumes exactly that (vit.py:160-161, x + drop_path(branch(x)):
 * d branch = factor * d out), so the separate saicv_row_scale pass over the residual-stream gradient disappears. */
int saicv_layernorm_bwd_scaled(int dtype, const void* dy, const void* x, const float* gamma, const float* mean,
                               const float* rstd, const void* addend, void* dx, float* dgamma, float* dbeta, float* ws,
                               int M, int C, int accumulate, const float* out_scale, int rows_per_scale, void* dx_scaled, void* stream);
/* DETR's post-norm residual norm(x + dropout(branch)) (reference detection/models/detr.py:89,92,114,118,122) as ONE pass each way.
 * Forward: sum_out = x + mask / (1 - p) * branch (stored, rounded), y = LayerNorm(sum_out), mean / rstd of its rows.  The mask is a
 * counter-based function of (seed + *seed_device, row, column) -- keep iff hash >= p * 2^32 -- regenerated by the backward, which
 * returns dsum (the gradient of x) and dbranch = mask / (1 - p) * dsum; dgamma / dbeta as saicv_layernorm_bwd.  seed_device: NULL or
 * a word in device memory (see saicv_attn_desc.seed_device). */
int saicv_dropout_add_layernorm_fwd(int dtype, const void* x, const void* branch, double p, unsigned int seed, const unsigned int* seed_device,
                                    const float* gamma, const float* beta, void* sum_out, void* y, float* mean, float* rstd, int M, int C,
                                    double eps, void* stream);
int saicv_dropout_add_layernorm_bwd(int dtype, const void* dy, const void* sum, const float* gamma, const float* mean, const float* rstd,
                                    double p, unsigned int seed, const unsigned int* seed_device, void* dsum, void* dbranch,
                                    float* dgamma, float* dbeta, float* ws, int M, int C, int accumulate, void* stream);
/* dropout(relu(x)) of DETR's feed-forward (reference detection/models/detr.py:90-91, 120-121: linear2(dropout(activation(linear1(x))))) as one
 * pass each way: y = keep ? relu(x) / (1 - p) : 0, keep a counter-based function of (seed + *seed_device, element) as in
 * saicv_dropout_add_layernorm_fwd; the backward reads y instead of a mask (y > 0 exactly where the gradient passes): dx = dy / (1 - p)
 * there, 0 elsewhere.  n: elements, a multiple of 8 (bf16) / 4 (fp32). */
int saicv_relu_dropout_fwd(int dtype, const void* x, void* y, size_t n, double p, unsigned int seed, const unsigned int* seed_device, void* stream);
int saicv_relu_dropout_bwd(int dtype, const void* dy, const void* y, void* dx, size_t n, double p, void* stream);
/* nn.GELU() (exact erf form), vit.py:87-99 */
int saicv_gelu_fwd(int dtype, const void* x, void* y, size_t n, void* stream);
int saicv_gelu_bwd(int dtype, const void* dy, const void* x, void* dx, size_t n, void* stream);
/* softmax(q k^T * scale) v per (batch, head) straight from the packed qkv GEMM output
 * qkv[B*N][3*H*D] (column = which*H*D + head*D + d, as vit.py:65-68 views it) into the head-merged
 * out[B*N][H*D]; lse[B][H][N] is kept for the backward.  MultiHeadAttention.forward, vit.py:61-80.
 * D = 64, N <= 256. */
int saicv_attention_fwd(int dtype, const void* qkv, void* out, float* lse, int B, int N, int H, int D,
                        double scale, void* stream);
int saicv_attention_bwd(int dtype, const void* qkv, const void* out, const void* dout, const float* lse,
                        void* dqkv, int B, int N, int H, int D, double scale, void* stream);

/* ---- SAM image-encoder layout / relative position (reference segment_anything/image_encoder.py) -------- */
/* window_partition (:32-55): x [B, H, W, C] -> out [B*nW, ws*ws, C], zero padded to multiples of ws */
int saicv_window_partition(int dtype, const void* x, void* out, int B, int H, int W, int C, int ws, void* stream);
/* window_unpartition (:58-79) fused with the residual add of Block.forward (:236): out = addend + unpartition(win) */
int saicv_window_unpartition(int dtype, const void* win, const void* addend, void* out, int B, int H, int W, int C, int ws,
                             void* stream);
/* get_rel_pos + the einsums of add_decomposed_rel_pos (:82-144) for q_size == k_size, head dim 64:
 * rel_h[b*heads + n, q, kh] = <q[b, q, n, :], tab_h[qh - kh + Sh - 1, :]>, rel_w likewise; q is addressed as
 * q + b*q_bs + qi*q_rs + n*64 (elements); tab_* are the fp32 parameters [2S-1][64]. */
int saicv_relpos_fwd(int dtype, const void* q, long q_rs, long q_bs, const float* tab_h, const float* tab_w, float* rel_h,
                     float* rel_w, int B, int heads, int Sh, int Sw, void* stream);
/* backward: dq (layout of q) += d_rel_* . tab_* ; dtab_* (fp32, both or neither) += d_rel_* . q, accumulated
 * through privatised copies in `ws` (saicv_relpos_bwd_ws_floats floats, contents irrelevant) */
size_t saicv_relpos_bwd_ws_floats(int Sh, int Sw);
int saicv_relpos_bwd(int dtype, const void* q, void* dq, long q_rs, long q_bs, const float* tab_h, const float* tab_w,
                     const float* d_rel_h, const float* d_rel_w, float* dtab_h, float* dtab_w, float* ws, int B, int heads,
                     int Sh, int Sw, void* stream);

/* SAM mask-loss statistics of logits [B, M, HW] against targets [B, HW] (fp32) in one pass:
 * stats[b, m, 0..5] = { sum focal, sum sigmoid*t, sum sigmoid, sum t, #(x>thr & t>thr), #(x>thr | t>thr) }.
 * Replaces SAMLoss.focal_loss / dice_loss / iou_predict_loss reductions
 * (reference interactive_segmentation/losses.py:136-198).  stats is zeroed by the call. */
int saicv_mask_loss_stats(int dtype, const void* logits, const float* targets, float* stats, int B, int M, size_t HW,
                          double alpha, double gamma, double thr, void* stream);
/* dlogits = coef[b,m,0] * dfocal/dx + (coef[b,m,1] * t + coef[b,m,2]) * sigmoid'(x) */
int saicv_mask_loss_grad(int dtype, const void* logits, const float* targets, const float* coef, void* dlogits, int B,
                         int M, size_t HW, double alpha, double gamma, void* stream);

/* ---- SAM mask-decoder / loss tail (segment_anything/mask_decoder.py:137-140, sam.py:155-158, losses.py:136-198) ---- */
/* masks[b][t][p] = <hyper[b][t][:], x[b][p][:]>, x [B][P][C = 32] (the upscaled embedding, NHWC), T <= 8 mask tokens */
int saicv_hyper_product_fwd(int dtype, const void* x, const void* hyper, void* out, int B, int T, int P, int C, void* stream);
/* dx [B][P][C] (compute dtype), dhyper [B][T][C] fp32 (zeroed here, then accumulated with atomics) */
int saicv_hyper_product_bwd(int dtype, const void* x, const void* hyper, const void* dout, void* dx, float* dhyper, int B,
                            int T, int P, int C, void* stream);
/* F.interpolate(scale 4, mode="bilinear", align_corners=False) over `planes` [h][w] planes, and its backward */
int saicv_upsample4_fwd(int dtype, const void* low, void* out, int planes, int h, int w, void* stream);
int saicv_upsample4_bwd(int dtype, const void* dhi, void* dlow, int planes, int h, int w, void* stream);
/* saicv_mask_loss_stats / _grad taken from the LOW-resolution logits [B][M][h][w] against full-resolution targets
 * [B][4h][4w]: the x4-interpolated logits exist only in registers; the gradient is with respect to the low-resolution logits */
int saicv_mask_loss_stats_up4(int dtype, const void* low, const float* targets, float* stats, int B, int M, int h, int w,
                              double alpha, double gamma, double thr, void* stream);
int saicv_mask_loss_grad_up4(int dtype, const void* low, const float* targets, const float* coef, void* dlow, int B, int M,
                             int h, int w, double alpha, double gamma, void* stream);

/* Streaming attention (any Nq / Nk, head dim 32 or 64, separate q / k / v with strides).
 * Replaces SAM Attention.forward + add_decomposed_rel_pos (reference interactive_segmentation/models/
 * segment_anything/image_encoder.py:116-184) and DETR's nn.MultiheadAttention calls with a float
 * key_padding_mask = additive bias (reference detection/models/detr.py:252-260, 309-327).
 * logits[q, k] = scale * <q, k> + key_bias[b, k] + rel_h[bh, q, k / Sw] + rel_w[bh, q, k % Sw]
 * (each bias term optional).  All strides in ELEMENTS and multiples of 16 bytes; head h of a row
 * starts at element h * D.  lse / dsum are [B*H, Nq] fp32; d_rel_h / d_rel_w are overwritten. */
typedef struct saicv_attn_desc {
    const void* q; const void* k; const void* v;
    long q_rs, k_rs, v_rs;          /* row strides */
    long q_bs, k_bs, v_bs;          /* batch strides */
    void* out; long o_rs, o_bs;     /* [B, Nq, H*D]-like, also the layout of dout */
    const void* dout;
    void* dq; void* dk; void* dv;   /* layouts of q / k / v */
    float* lse;
    float* dsum;
    const float* key_bias;          /* [B, Nk] or NULL */
    const float* rel_h; const float* rel_w;   /* [B*H, Nq, Sh] / [B*H, Nq, Sw] or NULL */
    float* d_rel_h; float* d_rel_w;
    int Sh, Sw;
    int B, H, Nq, Nk;
    float scale;
    float dropout_p;                /* attention-probability dropout (0 = off); same seed in fwd and bwd */
    unsigned int seed;
    const unsigned int* seed_device; /* NULL, or a word in device memory ADDED to `seed` (a step captured into a hipGraph freezes `seed`;
                                      * the host bumps this word before every replay so that replays draw different masks) */
} saicv_attn_desc;
int saicv_attention_stream_fwd(int dtype, int D, const saicv_attn_desc* desc, void* stream);
/* backward = dQ pass (also writes dsum and d_rel_*) followed by the dK/dV pass */
int saicv_attention_stream_bwd(int dtype, int D, const saicv_attn_desc* desc, void* stream);

/* ---- on-device batch preparation ---------------------------------
 * Mixup / CutMix of sample i with sample B-1-i (reference SimpleAICV/classification/mixupcutmixclassificationcollator.py:
 * 140-284) from a plan the HOST draws with the reference's numpy calls (one entry per sample), on an NHWC batch in device
 * memory: fp32, or uint8 with the dataset's per-channel normalisation v * scale[c] + shift[c] folded in.  mode 0: copy,
 * 1: x * lam + x' * one_minus_lam (two products and a sum, each rounded: bit-identical to the reference's tensor expression),
 * 2: the box rows [yl, yh) x columns [xl, xh) come from x'.  dst: fp32 [B][H][W][C], not aliasing src. */
typedef struct saicv_mix_plan {
    int mode, yl, yh, xl, xh;
    float lam, one_minus_lam;              /* image mixing weights */
    float label_lam, label_one_minus_lam;  /* label mixing weights (the box-corrected lambda for CutMix) */
} saicv_mix_plan;
int saicv_mixup_cutmix(int src_is_u8, const void* src, const saicv_mix_plan* plan, const float* scale, const float* shift,
                       float* dst, int B, int H, int W, int C, void* stream);
/* out[b][c] = y[b][c] * label_lam + y[B-1-b][c] * label_one_minus_lam, y = on_value at the label's class, off_value elsewhere
 * (the smoothed one-hot of the same collater, :262-284) */
int saicv_soft_labels(const long long* labels, const saicv_mix_plan* plan, float off_value, float on_value, float* out, int B,
                      int num_classes, void* stream);
/* The dataset normalisation on the uint8 batch the loader shipped (reference SimpleAICV/classification/common.py:228-248,
 * TorchMeanStdNormalize = torchvision ToTensor + Normalize, there per sample on a CPU worker): src u8 [n] in NHWC order (n =
 * B*H*W*C), dst fp32 [n]: ((float)v / 255 - mean[c]) / std[c], each step rounded as the tensor expressions round. */
int saicv_u8_normalize(const unsigned char* src, const float* mean, const float* stdv, float* dst, size_t n, int C, void* stream);
/* RandomErasing on the device batch (reference common.py:561-640): the host draws the boxes (and, for the 'const' / 'rand'
 * modes, the colour) with the reference's numpy calls; x fp32 [B][H][W][C] is filled in place.  mode 0: color[c]; mode 1
 * ('pixel'): every element its own N(0, 1) value from a counter-based generator keyed by (seed, box index, element).  Boxes of
 * one call must not overlap within an image (the reference applies a sample's boxes in sequence: one call per round). */
typedef struct saicv_erase_box {
    int b, top, left, h, w, mode;
    float color[4];
} saicv_erase_box;
int saicv_random_erase(float* x, const saicv_erase_box* boxes, int nboxes, int B, int H, int W, int C, unsigned int seed, void* stream);
/* One click per sample, uniform over the error region of a mask prediction (reference tools/interactive_segmentation_scripts.py:
 * 202-228): label 1 on a false-negative pixel, label 0 on a false-positive one -- or on a background pixel when the
 * prediction is exact.  gt: fp32 [B][H][W], a pixel is foreground where gt > gt_threshold; pred: [B][pred_channels][H][W]
 * logits (dtype 0 bf16 / 1 fp32; NULL = nothing predicted), channel pred_index[b] (NULL = 0), foreground where
 * > pred_threshold.  keys_ws: B * 4 u64 of scratch; points: fp32 [B][3] = (x, y, label).  The draw of every (pixel, label)
 * slot is a counter-based function of (seed, sample, slot): no noise tensor exists. */
int saicv_sam_sample_point(int pred_dtype, const float* gt, const void* pred, long pred_plane_stride, const long long* pred_index,
                           int pred_channels, float gt_threshold, float pred_threshold, unsigned int seed,
                           unsigned long long* keys_ws, float* points, int B, int H, int W, void* stream);
/* The same with the seed completed on the device: seed + *seed_device (NULL: seed alone).  A training step captured into a hipGraph
 * (tools/interactive_segmentation_scripts.py, config.use_step_graph) bakes `seed` in; the host bumps *seed_device between replays. */
int saicv_sam_sample_point_dseed(int pred_dtype, const float* gt, const void* pred, long pred_plane_stride, const long long* pred_index,
                                 int pred_channels, float gt_threshold, float pred_threshold, unsigned int seed, const unsigned int* seed_device,
                                 unsigned long long* keys_ws, float* points, int B, int H, int W, void* stream);

/* SAM sparse prompt tokens in one launch (reference segment_anything/prompt_encoder.py:150-190 embed_points / embed_boxes over
 * :28-49 PositionEmbeddingRandom): points fp32 [B][Np][3] = (x, y, label) or NULL, `pad` = 1 appends the reference's padding
 * click, boxes fp32 [B][4] or NULL, gauss fp32 [2][F], table fp32 [5][2F] = rows {negative click, positive click, box corner 1,
 * box corner 2, not-a-point}; tokens fp32 [B][T][2F], kinds int32 [B][T] (kept for the backward), T = Np + pad + 2 [boxes].
 * _bwd ACCUMULATES d table[5][C] from d tokens.  saicv_sam_grid_pe: the same encoding on the S x S grid centres, out [2F][S][S]. */
int saicv_sam_prompt_tokens(const float* points, int Np, int pad, const float* boxes, const float* gauss, int F, const float* table,
                            float image_size, float* tokens, int* kinds, int B, void* stream);
int saicv_sam_prompt_tokens_bwd(const float* dtokens, const int* kinds, float* dtable, int BT, int C, void* stream);
int saicv_sam_grid_pe(const float* gauss, int F, int S, float* out, void* stream);

/* DETR sine position embedding (reference detection/models/backbones/detr_resnet.py:28-64): mask u8 / bool [B][H][W] (non-zero =
 * padding) -> out fp32 [B][2F][H][W]; H * W <= 8192. */
int saicv_detr_sine_pe(const unsigned char* mask, float* out, int B, int H, int W, int F, float temperature, float eps, void* stream);

/* DETR box losses over the static-shape pair buffers (late r06; compute_batch_l1_iou_loss, reference SimpleAICV/detection/losses.py:938-954,
 * as DETRLoss.forward_static states it): reg fp32 [L][B][Q][4] raw predictions (cx cy w h; clamped to [lo, hi] inside), gt fp32 [B][T][5]
 * (rows with class < 0 are padding), src / tgt int64 [B][T] and w fp32 [B][T] from saicv_detr_assign.  out fp32 [2 L + 1]:
 * out[l] = sum over pairs of w * |p - t|_1 / n, out[L + l] = sum of w * (1 - GIoU(p, t)) / n, out[2 L] = n = number of ground-truth rows
 * (n = 0 gives nan, as the reference's 0 / 0).  Backward: dreg fp32 [L][B][Q][4] (every row written: zeros for unmatched queries) from
 * d_l1 / d_iou fp32 [L] (either may be NULL = zeros) and the forward's out (for n). */
int saicv_detr_box_loss_fwd(const float* reg, const float* gt, const long long* src, const long long* tgt, const float* w, int L, int B, int Q,
                            int T, double lo, double hi, float* out, void* stream);
int saicv_detr_box_loss_bwd(const float* reg, const float* gt, const long long* src, const long long* tgt, const float* w, const float* d_l1,
                            const float* d_iou, const float* out, int L, int B, int Q, int T, double lo, double hi, float* dreg, void* stream);

/* DETR Hungarian assignment on the device (r05; replaces the host-side scipy.optimize.linear_sum_assignment call of reference
 * SimpleAICV/detection/losses.py:1009-1090, which put a device -> host copy and a synchronisation between forward and loss of every
 * step): cost fp32 [B][Q][T] = matching cost of every query against every row of the padded ground truth, valid u8 [B][T] = rows that
 * are boxes.  Per image the minimum-cost assignment over the valid columns, scipy's algorithm and tie rules in double precision
 * (nan -> 1e5, one-signed infinities -> a finite value beyond any total, as the reference's wrapper).  Output: src / tgt int64 [B][T],
 * w fp32 [B][T]: slot k < min(n, Q) holds the pair (query src, ground-truth row tgt) with w = 1, the other slots 0.  Q, T <= 2048. */
int saicv_detr_assign(const float* cost, const unsigned char* valid, int B, int Q, int T, long long* src, long long* tgt, float* w, void* stream);

/* ---- depthwise convolution ---------------------------------
 * nn.Conv2d(C, C, K, stride, padding, dilation, groups=C) and its backward: reference
 * SimpleAICV/classification/backbones/van.py:30,68,75 (3x3 / 5x5 / dilated 7x7 of the LKA block) and convformer.py (7x7 of the
 * SepConv mixer).  NHWC activations; weights tap-major wt[K*K][C] in the compute dtype (the [C,1,K,K] parameter transposed),
 * weight gradient dwt[K*K][C] fp32.  dtype 0 bf16 / 1 fp32, fp32 accumulation, C a multiple of 8 (bf16) / 4 (fp32), K <= 8. */
int saicv_dwconv2d_fwd(int dtype, const void* x, const void* wt, const float* bias, void* y, int N, int H, int W, int C, int OH,
                       int OW, int K, int stride, int pad, int dil, void* stream);
int saicv_dwconv2d_dgrad(int dtype, const void* dy, const void* wt, void* dx, int N, int H, int W, int C, int OH, int OW, int K,
                         int stride, int pad, int dil, void* stream);
/* ACCUMULATES into dwt / dbias (fp32 atomics; dbias may be NULL) */
int saicv_dwconv2d_wgrad(int dtype, const void* dy, const void* x, float* dwt, float* dbias, int N, int H, int W, int C, int OH,
                         int OW, int K, int stride, int pad, int dil, void* stream);

/* ---- streaming glue of the convolutional backbones that reuse the hot-path blocks (csrc/elemwise.hip) ---- */
/* nn.ReLU / nn.LeakyReLU(slope) / nn.SiLU (kind 0 / 1 / 2) on a dense tensor of n elements (n % (16 / element size) == 0):
 * reference classification/backbones/darknet.py:16-33 (ActivationBlock), van.py:44,103, convformer.py:53,86.
 * The backward takes the forward INPUT x: dx = dy * act'(x). */
int saicv_act_fwd(int dtype, int kind, double slope, const void* x, void* y, size_t n, void* stream);
int saicv_act_bwd(int dtype, int kind, double slope, const void* dy, const void* x, void* dx, size_t n, void* stream);
/* out = a * b on tensors of one dense layout (van.py:91 `u * attn`); da = dy * b, db = dy * a (either may be NULL) */
int saicv_mul_fwd(int dtype, const void* a, const void* b, void* out, size_t n, void* stream);
int saicv_mul_bwd(int dtype, const void* dy, const void* a, const void* b, void* da, void* db, size_t n, void* stream);
/* DINOv3 blocks (reference SimpleAICV/detection/models/backbones/dinov3vit.py).  saicv_rope_apply: rotary position embedding on
 * the q and k thirds of a packed projection qkv [B][N][3][heads][D] -> out (same layout; v and the first `prefix` tokens copied):
 * y = x * cos + rot(x) * sin, rot(x) = [-x2, x1] over the halves of D (:262-276, :331-353), sin / cos fp32 [N - prefix][D];
 * transpose = 1 is the gradient map dx = g * cos + rot^T(g * sin).  fp32 arithmetic, one rounding.  D a multiple of 16 (bf16) / 8.
 * saicv_swiglu_*: hidden = silu(x1) * x2 (:137-140) and (dx1, dx2) from dy in one pass each. */
int saicv_rope_apply(int dtype, const void* qkv, const float* sin_t, const float* cos_t, void* out, int B, int N, int heads, int D,
                     int prefix, int transpose, void* stream);
int saicv_swiglu_fwd(int dtype, const void* x1, const void* x2, void* out, size_t n, void* stream);
int saicv_swiglu_bwd(int dtype, const void* dy, const void* x1, const void* x2, void* dx1, void* dx2, size_t n, void* stream);
/* out[m][c] = x[m][c] + s[c] * y[m][c] on [M][C] (NHWC) tensors, s fp32 [C]; x NULL: s * y; s NULL: x + y.
 * van.py:181-185 (x + layer_scale * branch), the residual joins of darknet.py (Darknet53Block) and convformer.py:157-163.
 * Backward of the branch: dy = s * dout (dy NULL: skipped; s NULL: ones), ds[c] += sum_m dout * y (ds NULL: skipped; fp32 atomics). */
int saicv_channel_scale_add_fwd(int dtype, const void* x, const void* y, const float* s, void* out, size_t M, int C, void* stream);
int saicv_channel_scale_add_bwd(int dtype, const void* dout, const void* y, const float* s, void* dy, float* ds, size_t M, int C,
                                void* stream);
/* per-channel sum / sum of squares of x[M][C] ADDED into sum[C] / sq[C] (fp32, zeroed by the caller): statistics of a
 * BatchNorm2d whose input is not a convolution output (van.py:176,178,260; convformer.py:34-35,143,149); feed them to
 * saicv_bn_finalize_fwd(rows = 1), then saicv_bn_act_fwd / saicv_bn_act_bwd as for the fused blocks. */
int saicv_bn_stats(int dtype, const void* x, size_t M, int C, float* sum, float* sq, void* stream);
/* The feature pyramid's top-down merge (reference SimpleAICV/detection/models/fpn.py:57-75): out[N,H,W,C] (fp32) =
 * F.interpolate(top[N,h,w,C], size=(H,W), mode='bilinear', align_corners=False) + lateral[N,H,W,C] (NULL: the resize alone), ATen's
 * fp32 tap arithmetic; and its gradient towards `top` as a fixed-order GATHER (ATen's backward scatters with atomics: not
 * reproducible run to run).  dtype_top / dtype_lat: SAICV_BF16 | SAICV_F32; C % 4 == 0. */
int saicv_resize_bilinear_add_fwd(int dtype_top, int dtype_lat, const void* top, const void* lateral, float* out, int N, int h, int w,
                                  int H, int W, int C, void* stream);
int saicv_resize_bilinear_bwd(int dtype_top, const float* dout, void* dtop, int N, int h, int w, int H, int W, int C, void* stream);

/* ---- nn.GroupNorm (+ ReLU) on NHWC activations (csrc/groupnorm.hip; reference SimpleAICV/detection/models/head.py:101-124) ---- */
size_t saicv_groupnorm_ws_floats(int N, int C);
/* y = [relu](GroupNorm_G(x)) on x [N][HW][C] (C % (16 / element size) == 0), fp32 arithmetic; gamma / beta fp32 [C] or NULL.
 * Saves mean_rstd [2][N][G] and the per-(sample, channel) affine coefficients ab [2][N][C] (y = x * a + b) for the backward. */
int saicv_groupnorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean_rstd, float* ab, float* ws,
                        int N, int HW, int C, int G, double eps, int relu, void* stream);
/* dx; dgamma[C] / dbeta[C] fp32 are ADDED to (NULL: not wanted); with relu the gate is recomputed from x * a + b */
int saicv_groupnorm_bwd(int dtype, const void* dy, const void* x, const float* gamma, const float* mean_rstd, const float* ab, void* dx,
                        float* dgamma, float* dbeta, float* ws, int N, int HW, int C, int G, int relu, void* stream);

/* ---- dense-detector training loss (csrc/detloss.hip; reference SimpleAICV/detection/losses.py RetinaLoss :123-433) ---- */
/* get_batch_anchors_annotations (:330-416): anchors [A][4] fp32 (one image's table, shared by the batch), annots [B][G][5] fp32
 * (x_min, y_min, x_max, y_max, class; class < 0 = padding row; G <= 1024) -> targets [B][A][5]: the box target of the best-IoU
 * ground-truth box ([tx, ty, tw, th] when smoothl1 != 0, the box itself otherwise) and the class target -1 (ignored: 0.4 <= IoU
 * < 0.5, or no ground truth in the image) / 0 (background: IoU < 0.4) / class + 1 (IoU >= 0.5); pos_count[0] += positives. */
int saicv_retina_assign(const float* anchors, const float* annots, float* targets, float* pos_count, int B, int A, int G,
                        int smoothl1, void* stream);
/* compute_batch_focal_loss (:222-262) of ONE pyramid level where the head wrote it: probs [B][Al][C] fp32 against
 * targets [B][At][5] (this level's anchors start at row `off` of every image).  loss_sum[0] += the sum before the division by
 * the positive count; dprobs (optional, same shape as probs) = its gradient (zero outside the clamp range [1e-4, 1 - 1e-4]). */
int saicv_focal_loss_level(const float* probs, const float* targets, float* dprobs, float* loss_sum, int B, int Al, int At, int off,
                           int C, double alpha, double gamma, void* stream);
/* compute_batch_smoothl1_loss (:305-328) of one level's positive anchors: reg [B][Al][4] fp32 (16-byte aligned) against
 * targets[..][0:4]; loss_sum[0] += the sum before the division; dreg (optional) = its gradient. */
int saicv_smoothl1_level(const float* reg, const float* targets, float* dreg, float* loss_sum, int B, int Al, int At, int off,
                         double beta, void* stream);
/* FCOSLoss.get_batch_position_annotations (:623-842): points [P][5] fp32 = (x, y, stride, regression range low, high) of one
 * image's pyramid, annots [B][G][5] -> targets [B][P][5] = (l, t, r, b, class + 1 | 0 for a negative point) -- the layout
 * saicv_focal_loss_level reads -- and centerness [B][P]; pos_count[0] += positives.  A ground-truth box is a candidate when the
 * point is strictly inside it, within radius * stride of its centre (center_sample != 0) and max(l, t, r, b) lies inside the
 * range; the smallest candidate wins. */
int saicv_fcos_assign(const float* points, const float* annots, float* targets, float* centerness, float* pos_count, int B, int P,
                      int G, double radius, int center_sample, void* stream);
/* Evaluation: first-maximum class and its probability per anchor / point of ONE level (reference SimpleAICV/detection/decode.py
 * RetinaDecoder :219-230; FCOSDecoder :331-343 with centerness [B][Al]: score = sqrt(probability * centre-ness)), written at the
 * level's offset of [B][At] score / class arrays: the [B][At][C] tensor never leaves the device. */
int saicv_det_best_class(const float* probs, const float* centerness, float* scores, int* classes, int B, int Al, int At, int off, int C,
                         void* stream);

/* ---- gradient all-reduce over RCCL / xGMI ------------------------------------------------
 * The reducer of nn.parallel.DistributedDataParallel (reference tools/train_classification_model.py:217-227 wraps the
 * model; tools/scripts.py:183-226 relies on gradients being averaged when backward() returns): contiguous ranges of the
 * flat fp32 gradient arena are averaged over the ranks on a communication stream while backward keeps running.
 * One process per GPU, one communicator per process.  Bootstrap: rank 0 calls saicv_comm_unique_id and hands the 128
 * bytes to the other ranks out of band (the host mirror uses the torch.distributed store); every rank then calls
 * saicv_comm_create (collective).  All other calls only enqueue work; RCCL is resolved with dlopen at first use. */
typedef struct saicv_comm saicv_comm;
/* 0 when RCCL could be resolved in this process (no communicator, no collective involved): the ranks agree on this BEFORE
 * anyone enters the collective saicv_comm_create, so that a rank-local failure sends everybody to the fallback instead of
 * leaving the others blocked inside ncclCommInitRank. */
int saicv_comm_available(void);
int saicv_comm_unique_id(void* id128);
int saicv_comm_create(const void* id128, int world, int rank, saicv_comm** out);
/* grads[0..n) (device, fp32) <- sum or mean over the ranks, in place, ordered after everything enqueued on producer_stream
 * so far (the stream whose kernels wrote this bucket): on the library's communication stream while producer_stream is being
 * captured into a hipGraph (or with SAICV_COMM_MODE=events), on producer_stream itself otherwise. */
int saicv_comm_allreduce_bucket(saicv_comm* c, float* grads, size_t n, int average, void* producer_stream);
/* The two halves of an all-reduce as separate collectives (over the point-to-point xGMI mesh a direct
 * reduce-scatter + all-gather keeps every link busy with 1/world of the bucket).  Stream choice and ordering as above.
 *   reduce_scatter: shard[0..n_per_rank) <- sum or mean over the ranks of grads[rank * n_per_rank ..), grads holds
 *                   world * n_per_rank floats; shard may alias grads + rank * n_per_rank (in place).
 *   all_gather    : buf[r * n_per_rank ..) <- rank r's shard, for every r; shard may alias buf + rank * n_per_rank. */
int saicv_comm_reduce_scatter(saicv_comm* c, const float* grads, float* shard, size_t n_per_rank, int average, void* producer_stream);
int saicv_comm_all_gather(saicv_comm* c, const float* shard, float* buf, size_t n_per_rank, void* producer_stream);
/* buf[0..bytes) <- root's copy (constructor-time parameter / per-forward buffer broadcast), ordered after what `stream` has
 * enqueued so far and before what it enqueues next (same stream choice as the all-reduce). */
int saicv_comm_broadcast(saicv_comm* c, void* buf, size_t bytes, int root, void* stream);
/* consumer_stream is ordered after every bucket enqueued so far (before the optimizer reads the gradients). */
int saicv_comm_join(saicv_comm* c, void* consumer_stream);
int saicv_comm_stats(const saicv_comm* c, int* world, int* rank, unsigned long long* buckets, unsigned long long* bytes);
int saicv_comm_destroy(saicv_comm* c);

#ifdef __cplusplus
}
#endif
#endif /* SAICV_HIP_H */
