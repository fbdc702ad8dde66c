// extern "C" surface of libsaicv_hip.so (declared in include/saicv_hip.h).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "common.h"
#include "saicv_internal.h"
#include "../../include/saicv_hip.h"

namespace saicv {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return -2;
    }
    return 0;
}

// declared in the other translation units
int pack_weight_batched(int, const saicv_pack_desc*, int, int, hipStream_t);
int pack_input_s2d(int, const float*, long, long, long, long, void*, int, int, int, int, int, int, hipStream_t);
int pack_weight_s2d(int, const float*, long, long, long, long, int, int, int, int, int, void*, hipStream_t);
int unpack_wgrad_s2d(const float*, int, int, int, int, int, float*, long, long, long, long, int, hipStream_t);
size_t bn_ws_floats(int C);
int bn_finalize_fwd(const float*, const float*, int, int, double, const float*, const float*, float*,
                    float*, double, double, float*, float*, float*, float*, float*, long long*, hipStream_t);
int bn_eval_coeffs(int, const float*, const float*, const float*, const float*, double, float*, float*,
                   hipStream_t);
int bn_act_fwd(int, const void*, const void*, void*, const float*, const float*, size_t, int, int, void*,
               hipStream_t);
size_t bn_bwd_ws_floats(size_t, int, int);
int bn_bwd(int, const void*, const void*, const void*, const void*, const float*, const float*, const float*, void*,
           void*, float*, float*, size_t, int, int, int, float*, hipStream_t);
int bn_act_fwd_stats(int, const void*, const void*, void*, const float*, const float*, int, double, const float*, const float*,
                     float*, float*, double, double, long long*, float*, float*, size_t, int, int, void*, hipStream_t);
int bn_act_fwd_join(int, const void*, const void*, const float*, const float*, void*, const float*, const float*, const float*,
                    const float*, int, double, const float*, const float*, float*, float*, double, double, long long*, float*,
                    float*, size_t, int, int, void*, hipStream_t);
int bn_bwd_inline(int, const void*, const void*, const void*, const float*, const float*, const float*, const float*,
                  const float*, int, void*, void*, float*, float*, size_t, int, int, int, hipStream_t);
int bn_bwd_from_partials(int, const void*, const void*, const void*, const float*, const float*, const float*, const float*,
                         const float*, int, void*, void*, float*, float*, size_t, int, int, int, float*, hipStream_t);
int maxpool_fwd(int, const void*, void*, uint8_t*, int, int, int, int, int, int, int, int, int, hipStream_t);
int maxpool_bwd(int, const void*, const uint8_t*, void*, int, int, int, int, int, int, int, int, int, hipStream_t);
int bn_relu_maxpool_fwd(int, const void*, const float*, const float*, void*, uint8_t*, int, int, int, int, int, int, int, int, int, hipStream_t);
int bn_relu_maxpool_bwd(int, const void*, const uint8_t*, const void*, const float*, const float*, const float*, const float*, const float*,
                        void*, float*, float*, int, float*, int, int, int, int, int, int, int, int, int, hipStream_t);
int avgpool_fwd(int, const void*, void*, int, int, int, hipStream_t);
int avgpool_bwd(int, const void*, void*, int, int, int, hipStream_t);
int softmax_ce_fwd(const float*, const void*, int, int, int, float*, float*, float*, hipStream_t);
int scale_by_scalar(int, const float*, const float*, void*, size_t, hipStream_t);
int pack_input(int, const float*, long, long, long, long, void*, int, int, int, int, int, hipStream_t);
int pack_weight(int, const float*, long, long, long, long, int, int, int, int, int, int, void*, void*, hipStream_t);
int unpack_wgrad(const float*, int, int, int, int, int, float*, long, long, long, long, int, hipStream_t);
int colsum(int, const void*, int, int, float*, hipStream_t);
int row_scale(int, const void*, const float*, void*, size_t, int, int, hipStream_t);
int sgd_flat(float*, const float*, float*, const int32_t*, const float*, const float*, const float*, const uint8_t*, size_t, hipStream_t);
int adamw_flat(float*, const float*, float*, float*, const int32_t*, const float*, const float*, const float*, const uint8_t*, float*, size_t, hipStream_t);
int grad_stats(const float*, size_t, float*, float*, hipStream_t);
int grad_clip_scale(float*, size_t, const float*, const float*, double, hipStream_t);
int grad_clip_value(float*, size_t, const float*, double, hipStream_t);
int scaler_update(float*, const float*, double, double, int, hipStream_t);
int layernorm_fwd(int, const void*, const float*, const float*, void*, float*, float*, int, int, double, hipStream_t);
size_t layernorm_bwd_ws_floats(int, int);
int dropout_add_layernorm_fwd(int, const void*, const void*, double, unsigned, const unsigned*, const float*, const float*, void*, void*,
                              float*, float*, int, int, double, hipStream_t);
int dropout_add_layernorm_bwd(int, const void*, const void*, const float*, const float*, const float*, double, unsigned, const unsigned*,
                              void*, void*, float*, float*, float*, int, int, int, hipStream_t);
int layernorm_bwd(int, const void*, const void*, const float*, const float*, const float*, const void*, void*,
                  float*, float*, float*, int, int, int, hipStream_t, const float* = nullptr, int = 1, void* = nullptr);
int gelu_fwd(int, const void*, void*, size_t, hipStream_t);
int gelu_bwd(int, const void*, const void*, void*, size_t, hipStream_t);
int relu_dropout_fwd(int, const void*, void*, size_t, double, unsigned, const unsigned*, hipStream_t);
int relu_dropout_bwd(int, const void*, const void*, void*, size_t, double, hipStream_t);
int attention_fwd(int, const void*, void*, float*, int, int, int, int, double, hipStream_t);
int attention_bwd(int, const void*, const void*, const void*, const float*, void*, int, int, int, int, double,
                  hipStream_t);

}  // namespace saicv

using namespace saicv;

static inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

static int check_desc(const saicv_conv_desc* d, const char* who) {
    SAICV_REQUIRE(d != nullptr, "%s: null descriptor", who);
    SAICV_REQUIRE(d->dtype == SAICV_BF16 || d->dtype == SAICV_F32, "%s: bad dtype %d", who, d->dtype);
    SAICV_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->K > 0 && d->R > 0 && d->S > 0,
                  "%s: non-positive dimension", who);
    SAICV_REQUIRE(d->stride >= 1 && d->pad >= 0, "%s: bad stride/pad", who);
    const int oh = (d->H + 2 * d->pad - d->R) / d->stride + 1;
    const int ow = (d->W + 2 * d->pad - d->S) / d->stride + 1;
    SAICV_REQUIRE(oh == d->OH && ow == d->OW, "%s: OH/OW (%d,%d) inconsistent with geometry (%d,%d)",
                  who, d->OH, d->OW, oh, ow);
    SAICV_REQUIRE((long long)d->N * d->H * d->W * (long long)d->C < (1ll << 31) &&
                      (long long)d->N * d->OH * d->OW * (long long)d->K < (1ll << 31),
                  "%s: tensor exceeds 2^31 elements", who);
    return 0;
}

extern "C" {

int saicv_version(void) { return 100; }
const char* saicv_last_error_string(void) { return g_err; }

int saicv_pack_input(int dtype, const float* src, long sN, long sC, long sH, long sW, void* dst,
                     int N, int C, int H, int W, int Cp, void* stream) {
    return pack_input(dtype, src, sN, sC, sH, sW, dst, N, C, H, W, Cp, S(stream));
}
int saicv_pack_weight(int dtype, const float* w, long sO, long sI, long sR, long sS, int O, int I,
                      int R, int Sx, int Ip, int Op, void* wf, void* wd, void* stream) {
    return pack_weight(dtype, w, sO, sI, sR, sS, O, I, R, Sx, Ip, Op, wf, wd, S(stream));
}
int saicv_unpack_wgrad(const float* dw, int O, int I, int R, int Sx, int Ip, float* grad, long sO,
                       long sI, long sR, long sS, int accumulate, void* stream) {
    return unpack_wgrad(dw, O, I, R, Sx, Ip, grad, sO, sI, sR, sS, accumulate, S(stream));
}

int saicv_pack_weight_batched(int dtype, const saicv_pack_desc* descs, int n, int total_tiles, void* stream) {
    return pack_weight_batched(dtype, descs, n, total_tiles, S(stream));
}
int saicv_pack_input_s2d(int dtype, const float* src, long sN, long sC, long sH, long sW, void* dst, int N, int C, int H,
                         int W, int pad, int Cq, void* stream) {
    return pack_input_s2d(dtype, src, sN, sC, sH, sW, dst, N, C, H, W, pad, Cq, S(stream));
}
int saicv_pack_weight_s2d(int dtype, const float* w, long sO, long sI, long sR, long sS, int O, int I, int R, int Sx,
                          int Cq, void* wf, void* stream) {
    return pack_weight_s2d(dtype, w, sO, sI, sR, sS, O, I, R, Sx, Cq, wf, S(stream));
}
int saicv_unpack_wgrad_s2d(const float* dw, int O, int I, int R, int Sx, int Cq, float* grad, long sO, long sI, long sR,
                           long sS, int accumulate, void* stream) {
    return unpack_wgrad_s2d(dw, O, I, R, Sx, Cq, grad, sO, sI, sR, sS, accumulate, S(stream));
}

// which streaming form (csrc/pwstream.hip) a convolution's geometry allows: 1 pointwise / stride 1 / no padding, 2 3 x 3 / stride 1 / padding 1
static int stream_form(const saicv_conv_desc* d) {
    if (d->R == 1 && d->S == 1 && d->pad == 0 && d->stride == 1) return 1;
    if (d->R == 3 && d->S == 3 && d->pad == 1 && d->stride == 1) return 2;
    return 0;
}

int saicv_conv2d_stat_rows(const saicv_conv_desc* d) {
    if (!d) return -1;
    return conv_stat_rows(d->N * d->OH * d->OW, d->K, d->R * d->S * d->C, d->dtype, stream_form(d));
}

int saicv_conv2d_fwd(const saicv_conv_desc* d, const void* x, const void* wf, const float* bias,
                     void* y, int out_f32, float* stat_sum, float* stat_sq, void* stream) {
    if (check_desc(d, "saicv_conv2d_fwd")) return -1;
    const int M = d->N * d->OH * d->OW;
    return igemm_nt(d->dtype, 0, x, wf, y, bias, stat_sum, stat_sq, d->H, d->W, d->C, d->OH, d->OW,
                    d->R, d->S, d->stride, d->pad, M, d->K, d->R * d->S * d->C, d->K, out_f32, S(stream));
}

int saicv_conv2d_dgrad(const saicv_conv_desc* d, const void* dy, const void* wd, void* dx,
                       void* stream) {
    if (check_desc(d, "saicv_conv2d_dgrad")) return -1;
    // rows = input pixels; gather source = dy [N,OH,OW,K]; k = (r,s,kout)
    const int M = d->N * d->H * d->W;
    return igemm_nt(d->dtype, 1, dy, wd, dx, nullptr, nullptr, nullptr, d->OH, d->OW, d->K, d->H, d->W,
                    d->R, d->S, d->stride, d->pad, M, d->C, d->R * d->S * d->K, d->C, 0, S(stream));
}

int saicv_conv2d_wgrad(const saicv_conv_desc* d, const void* dy, const void* x, float* dw,
                       void* stream) {
    if (check_desc(d, "saicv_conv2d_wgrad")) return -1;
    const int M = d->N * d->OH * d->OW;
    return igemm_tn(d->dtype, dy, x, dw, d->H, d->W, d->C, d->OH, d->OW, d->R, d->S, d->stride, d->pad,
                    M, d->K, d->R * d->S * d->C, S(stream));
}

int saicv_conv2d_wgrad_bias(const saicv_conv_desc* d, const void* dy, const void* x, float* dw, float* dbias,
                            void* stream) {
    if (check_desc(d, "saicv_conv2d_wgrad_bias")) return -1;
    const int M = d->N * d->OH * d->OW;
    return igemm_tn(d->dtype, dy, x, dw, d->H, d->W, d->C, d->OH, d->OW, d->R, d->S, d->stride, d->pad,
                    M, d->K, d->R * d->S * d->C, S(stream), dbias);
}

int saicv_linear_fwd(int dtype, const void* x, const void* wf, const float* bias, void* y, int M, int K, int N,
                     int out_f32, const void* addend, const float* row_scale, int rows_per_scale, void* stream) {
    EpiExtra ex;
    ex.addend = addend; ex.row_scale = row_scale; ex.rows_per_scale = rows_per_scale > 0 ? rows_per_scale : 1;
    return igemm_nt(dtype, 0, x, wf, y, bias, nullptr, nullptr, 1, 1, K, 1, 1, 1, 1, 1, 0, M, N, K, N, out_f32,
                    S(stream), (addend || row_scale) ? &ex : nullptr);
}
int saicv_linear_dgrad(int dtype, const void* dy, const void* wd, void* dx, int M, int K, int N, const void* addend,
                       void* stream) {
    EpiExtra ex;
    ex.addend = addend;
    return igemm_nt(dtype, 1, dy, wd, dx, nullptr, nullptr, nullptr, 1, 1, N, 1, 1, 1, 1, 1, 0, M, K, N, K, 0,
                    S(stream), addend ? &ex : nullptr);
}
int saicv_linear_gelu_fwd(int dtype, const void* x, const void* wf, const float* bias, void* y_pre, void* y_act, int M,
                          int K, int N, void* stream) {
    EpiExtra ex;
    ex.act_mode = 1; ex.out2 = y_act;
    return igemm_nt(dtype, 0, x, wf, y_pre, bias, nullptr, nullptr, 1, 1, K, 1, 1, 1, 1, 1, 0, M, N, K, N, 0, S(stream), &ex);
}
int saicv_linear_dgrad_gelu(int dtype, const void* dy, const void* wd, const void* pre, void* dx, int M, int K, int N,
                            void* stream) {
    EpiExtra ex;
    ex.act_mode = 2; ex.addend = pre;
    return igemm_nt(dtype, 1, dy, wd, dx, nullptr, nullptr, nullptr, 1, 1, N, 1, 1, 1, 1, 1, 0, M, K, N, K, 0, S(stream), &ex);
}
int saicv_linear_gelu_fwd_aux(int dtype, const void* x, const void* wf, const float* bias, void* y_dact, void* y_act, int M,
                              int K, int N, void* stream) {
    EpiExtra ex;
    ex.act_mode = 3; ex.out2 = y_act;
    return igemm_nt(dtype, 0, x, wf, y_dact, bias, nullptr, nullptr, 1, 1, K, 1, 1, 1, 1, 1, 0, M, N, K, N, 0, S(stream), &ex);
}
int saicv_linear_dgrad_mul(int dtype, const void* dy, const void* wd, const void* factor, void* dx, int M, int K, int N,
                           void* stream) {
    EpiExtra ex;
    ex.act_mode = 4; ex.addend = factor;
    return igemm_nt(dtype, 1, dy, wd, dx, nullptr, nullptr, nullptr, 1, 1, N, 1, 1, 1, 1, 1, 0, M, K, N, K, 0, S(stream), &ex);
}
int saicv_linear_wgrad(int dtype, const void* dy, const void* x, float* dw, float* dbias, int M, int K, int N,
                       void* stream) {
    return igemm_tn(dtype, dy, x, dw, 1, 1, K, 1, 1, 1, 1, 1, 0, M, N, K, S(stream), dbias);
}
int saicv_conv2d_dgrad_add(const saicv_conv_desc* d, const void* dy, const void* wd, const void* addend, void* dx,
                           void* stream) {
    if (check_desc(d, "saicv_conv2d_dgrad_add")) return -1;
    const int M = d->N * d->H * d->W;
    EpiExtra ex;
    ex.addend = addend;
    return igemm_nt(d->dtype, 1, dy, wd, dx, nullptr, nullptr, nullptr, d->OH, d->OW, d->K, d->H, d->W,
                    d->R, d->S, d->stride, d->pad, M, d->C, d->R * d->S * d->K, d->C, 0, S(stream),
                    addend ? &ex : nullptr);
}
int saicv_conv2d_dgrad_stat_rows(const saicv_conv_desc* d) {
    if (!d) return -1;
    return conv_bwd_stat_rows(d->N * d->H * d->W, d->H, d->W, d->C, d->R * d->S * d->K, d->stride, d->dtype, stream_form(d));
}
int saicv_conv2d_dgrad_fused(const saicv_conv_desc* d, const void* dy, const void* wd, const saicv_dgrad_fuse* f, void* dx,
                             void* stream) {
    if (check_desc(d, "saicv_conv2d_dgrad_fused")) return -1;
    if (!f) { set_error("saicv_conv2d_dgrad_fused: null fusion descriptor"); return -1; }
    const int M = d->N * d->H * d->W;
    EpiExtra ex;
    ex.addend = f->addend;
    ex.addend_gate = static_cast<const uint8_t*>(f->addend_gate);
    ex.bs_y = f->bn_y;
    ex.bs_mask = static_cast<const uint8_t*>(f->bn_mask);
    ex.bs_mean = f->bn_mean;
    ex.bs_invstd = f->bn_invstd;
    ex.bs_g = f->part_g;
    ex.bs_gx = f->part_gx;
    ex.stat_atomic_rows = f->part_rows;        // > 0: the caller zeroed part_rows rows and wants the sums added into them
    if (f->bn_y && d->stride > 1 && f->part_rows == 0) {
        // parity classes smaller than the largest one leave their last partial rows unwritten
        const size_t bytes = (size_t)saicv_conv2d_dgrad_stat_rows(d) * d->C * sizeof(float);
        if (f->part_g) hipMemsetAsync(f->part_g, 0, bytes, S(stream));
        if (f->part_gx) hipMemsetAsync(f->part_gx, 0, bytes, S(stream));
    }
    return igemm_nt(d->dtype, 1, dy, wd, dx, nullptr, nullptr, nullptr, d->OH, d->OW, d->K, d->H, d->W,
                    d->R, d->S, d->stride, d->pad, M, d->C, d->R * d->S * d->K, d->C, 0, S(stream),
                    (f->addend || f->bn_y) ? &ex : nullptr);
}
int saicv_row_scale(int dtype, const void* x, const float* scale, void* out, size_t rows, int row_len,
                    int rows_per_scale, void* stream) {
    return row_scale(dtype, x, scale, out, rows, row_len, rows_per_scale, S(stream));
}

int saicv_colsum(int dtype, const void* dy, int M, int N, float* dbias, void* stream) {
    return colsum(dtype, dy, M, N, dbias, S(stream));
}

size_t saicv_bn_ws_floats(int C) { return bn_ws_floats(C); }

int saicv_bn_finalize_fwd(const float* sum, const float* sq, int rows, int C, double count,
                          const float* gamma, const float* beta, float* running_mean,
                          float* running_var, double momentum, double eps, float* mean,
                          float* invstd, float* scale, float* shift, float* ws, long long* num_batches_tracked,
                          void* stream) {
    return bn_finalize_fwd(sum, sq, rows, C, count, gamma, beta, running_mean, running_var, momentum,
                           eps, mean, invstd, scale, shift, ws, num_batches_tracked, S(stream));
}
int saicv_bn_eval_coeffs(int C, const float* gamma, const float* beta, const float* running_mean,
                         const float* running_var, double eps, float* scale, float* shift,
                         void* stream) {
    return bn_eval_coeffs(C, gamma, beta, running_mean, running_var, eps, scale, shift, S(stream));
}
int saicv_bn_act_fwd(int dtype, const void* y, const void* res, void* z, const float* scale,
                     const float* shift, size_t M, int C, int relu, void* relu_mask, void* stream) {
    return bn_act_fwd(dtype, y, res, z, scale, shift, M, C, relu, relu_mask, S(stream));
}
size_t saicv_bn_bwd_ws_floats(size_t M, int C, int dtype) { return bn_bwd_ws_floats(M, C, dtype); }
int saicv_bn_act_bwd(int dtype, const void* dz, const void* z, const void* relu_mask, const void* y, const float* gamma,
                     const float* mean, const float* invstd, void* dy, void* dres, float* dgamma,
                     float* dbeta, size_t M, int C, int relu, int accumulate, float* ws, void* stream) {
    return bn_bwd(dtype, dz, z, relu_mask, y, gamma, mean, invstd, dy, dres, dgamma, dbeta, M, C, relu, accumulate, ws,
                  S(stream));
}

int saicv_conv2d_fwd_stats(const saicv_conv_desc* d, const void* x, const void* wf, void* y, float* stat_sum, float* stat_sq,
                           int stat_rows, void* stream) {
    if (check_desc(d, "saicv_conv2d_fwd_stats")) return -1;
    if (!stat_sum || !stat_sq || stat_rows < 1) { set_error("saicv_conv2d_fwd_stats: statistics rows missing"); return -1; }
    const int M = d->N * d->OH * d->OW;
    EpiExtra ex;
    ex.stat_atomic_rows = stat_rows;
    return igemm_nt(d->dtype, 0, x, wf, y, nullptr, stat_sum, stat_sq, d->H, d->W, d->C, d->OH, d->OW,
                    d->R, d->S, d->stride, d->pad, M, d->K, d->R * d->S * d->C, d->K, 0, S(stream), &ex);
}
int saicv_bn_act_fwd_stats(int dtype, const void* y, const void* res, void* z, const float* stat_sum, const float* stat_sq,
                           int stat_rows, double count, const float* gamma, const float* beta, float* running_mean,
                           float* running_var, double momentum, double eps, long long* num_batches_tracked, float* mean,
                           float* invstd, size_t M, int C, int relu, void* relu_mask, void* stream) {
    return bn_act_fwd_stats(dtype, y, res, z, stat_sum, stat_sq, stat_rows, count, gamma, beta, running_mean, running_var,
                            momentum, eps, num_batches_tracked, mean, invstd, M, C, relu, relu_mask, S(stream));
}
int saicv_bn_act_fwd_join(int dtype, const void* y, const void* res, const float* res_scale, const float* res_shift, void* z,
                          const float* scale, const float* shift, const float* stat_sum, const float* stat_sq, int stat_rows,
                          double count, const float* gamma, const float* beta, float* running_mean, float* running_var,
                          double momentum, double eps, long long* num_batches_tracked, float* mean, float* invstd, size_t M, int C,
                          int relu, void* relu_mask, void* stream) {
    return bn_act_fwd_join(dtype, y, res, res_scale, res_shift, z, scale, shift, stat_sum, stat_sq, stat_rows, count, gamma, beta,
                           running_mean, running_var, momentum, eps, num_batches_tracked, mean, invstd, M, C, relu, relu_mask,
                           S(stream));
}
int saicv_bn_act_bwd_inline(int dtype, const void* dz, const void* relu_mask, const void* y, const float* gamma,
                            const float* mean, const float* invstd, const float* part_g, const float* part_gx, int rows,
                            void* dy, void* dres, float* dgamma, float* dbeta, size_t M, int C, int relu, int accumulate,
                            void* stream) {
    return bn_bwd_inline(dtype, dz, relu_mask, y, gamma, mean, invstd, part_g, part_gx, rows, dy, dres, dgamma, dbeta, M, C,
                         relu, accumulate, S(stream));
}

int saicv_bn_act_bwd_from_partials(int dtype, const void* dz, const void* relu_mask, const void* y, const float* gamma,
                                   const float* mean, const float* invstd, const float* part_g, const float* part_gx,
                                   int rows, void* dy, void* dres, float* dgamma, float* dbeta, size_t M, int C, int relu,
                                   int accumulate, float* ws, void* stream) {
    return bn_bwd_from_partials(dtype, dz, relu_mask, y, gamma, mean, invstd, part_g, part_gx, rows, dy, dres, dgamma, dbeta,
                                M, C, relu, accumulate, ws, S(stream));
}

int saicv_maxpool_fwd(int dtype, const void* x, void* out, uint8_t* idx, int N, int H, int W, int C,
                      int OH, int OW, int K, int stride, int pad, void* stream) {
    return maxpool_fwd(dtype, x, out, idx, N, H, W, C, OH, OW, K, stride, pad, S(stream));
}
int saicv_maxpool_bwd(int dtype, const void* dout, const uint8_t* idx, void* dx, int N, int H, int W,
                      int C, int OH, int OW, int K, int stride, int pad, void* stream) {
    return maxpool_bwd(dtype, dout, idx, dx, N, H, W, C, OH, OW, K, stride, pad, S(stream));
}
int saicv_bn_relu_maxpool_fwd(int dtype, const void* y, const float* scale, const float* shift, void* out, uint8_t* idx, int N, int H,
                              int W, int C, int OH, int OW, int K, int stride, int pad, void* stream) {
    return bn_relu_maxpool_fwd(dtype, y, scale, shift, out, idx, N, H, W, C, OH, OW, K, stride, pad, S(stream));
}
size_t saicv_bn_relu_maxpool_bwd_ws_floats(int C) { return 2 * (size_t)C; }
int saicv_bn_relu_maxpool_bwd(int dtype, const void* dout, const uint8_t* idx, const void* y, const float* gamma, const float* mean,
                              const float* invstd, const float* scale, const float* shift, void* dy, float* dgamma, float* dbeta,
                              int accumulate, float* ws, int N, int H, int W, int C, int OH, int OW, int K, int stride, int pad,
                              void* stream) {
    return bn_relu_maxpool_bwd(dtype, dout, idx, y, gamma, mean, invstd, scale, shift, dy, dgamma, dbeta, accumulate, ws, N, H, W, C,
                               OH, OW, K, stride, pad, S(stream));
}

int saicv_avgpool_fwd(int dtype, const void* x, void* out, int N, int HW, int C, void* stream) {
    return avgpool_fwd(dtype, x, out, N, HW, C, S(stream));
}
int saicv_avgpool_bwd(int dtype, const void* dout, void* dx, int N, int HW, int C, void* stream) {
    return avgpool_bwd(dtype, dout, dx, N, HW, C, S(stream));
}

int saicv_softmax_ce_fwd(const float* logits, const void* label, int soft, int B, int C,
                         float* row_loss, float* loss, float* dlogits, void* stream) {
    return softmax_ce_fwd(logits, label, soft, B, C, row_loss, loss, dlogits, S(stream));
}
int saicv_scale_by_scalar(int out_dtype, const float* in, const float* scale, void* out, size_t n,
                          void* stream) {
    return scale_by_scalar(out_dtype, in, scale, out, n, S(stream));
}

int saicv_sgd_flat(float* p, const float* g, float* mom, const int32_t* block_group,
                   const float* hyper, const float* inv_scale, const float* found_inf,
                   const uint8_t* has_grad, size_t n, void* stream) {
    return sgd_flat(p, g, mom, block_group, hyper, inv_scale, found_inf, has_grad, n, S(stream));
}
int saicv_adamw_flat(float* p, const float* g, float* m, float* v, const int32_t* block_group,
                     const float* hyper, const float* inv_scale, const float* found_inf,
                     const uint8_t* has_grad, float* step_blk, size_t n, void* stream) {
    return adamw_flat(p, g, m, v, block_group, hyper, inv_scale, found_inf, has_grad, step_blk, n, S(stream));
}
int saicv_grad_stats(const float* g, size_t n, float* found_inf, float* sumsq, void* stream) {
    return grad_stats(g, n, found_inf, sumsq, S(stream));
}
int saicv_grad_clip_scale(float* g, size_t n, const float* sumsq, const float* inv_scale,
                          double max_norm, void* stream) {
    return grad_clip_scale(g, n, sumsq, inv_scale, max_norm, S(stream));
}
int saicv_grad_clip_value(float* g, size_t n, const float* inv_scale, double value, void* stream) {
    return grad_clip_value(g, n, inv_scale, value, S(stream));
}
int saicv_scaler_update(float* state, const float* found_inf, double growth, double backoff,
                        int interval, void* stream) {
    return scaler_update(state, found_inf, growth, backoff, interval, S(stream));
}

int saicv_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean,
                        float* rstd, int M, int C, double eps, void* stream) {
    return layernorm_fwd(dtype, x, gamma, beta, y, mean, rstd, M, C, eps, S(stream));
}
size_t saicv_layernorm_bwd_ws_floats(int M, int C) { return layernorm_bwd_ws_floats(M, C); }
int saicv_layernorm_bwd(int dtype, const void* dy, const void* x, const float* gamma, const float* mean,
                        const float* rstd, const void* addend, void* dx, float* dgamma, float* dbeta, float* ws,
                        int M, int C, int accumulate, void* stream) {
    return layernorm_bwd(dtype, dy, x, gamma, mean, rstd, addend, dx, dgamma, dbeta, ws, M, C, accumulate, S(stream));
}
int saicv_dropout_add_layernorm_fwd(int dtype, const void* x, const void* branch, double p, unsigned int seed, const unsigned int* seed_device,
                                    const float* gamma, const float* beta, void* sum_out, void* y, float* mean, float* rstd, int M, int C,
                                    double eps, void* stream) {
    return dropout_add_layernorm_fwd(dtype, x, branch, p, seed, seed_device, gamma, beta, sum_out, y, mean, rstd, M, C, eps, S(stream));
}
int saicv_dropout_add_layernorm_bwd(int dtype, const void* dy, const void* sum, const float* gamma, const float* mean, const float* rstd,
                                    double p, unsigned int seed, const unsigned int* seed_device, void* dsum, void* dbranch,
                                    float* dgamma, float* dbeta, float* ws, int M, int C, int accumulate, void* stream) {
    return dropout_add_layernorm_bwd(dtype, dy, sum, gamma, mean, rstd, p, seed, seed_device, dsum, dbranch, dgamma, dbeta, ws, M, C,
                                     accumulate, S(stream));
}
int saicv_layernorm_bwd_scaled(int dtype, const void* dy, const void* x, const float* gamma, const float* mean,
                               const float* rstd, const void* addend, void* dx, float* dgamma, float* dbeta, float* ws,
                               int M, int C, int accumulate, const float* out_scale, int rows_per_scale, void* dx_scaled, void* stream) {
    return layernorm_bwd(dtype, dy, x, gamma, mean, rstd, addend, dx, dgamma, dbeta, ws, M, C, accumulate, S(stream), out_scale,
                         rows_per_scale, dx_scaled);
}
int saicv_gelu_fwd(int dtype, const void* x, void* y, size_t n, void* stream) {
    return gelu_fwd(dtype, x, y, n, S(stream));
}
int saicv_gelu_bwd(int dtype, const void* dy, const void* x, void* dx, size_t n, void* stream) {
    return gelu_bwd(dtype, dy, x, dx, n, S(stream));
}
int saicv_relu_dropout_fwd(int dtype, const void* x, void* y, size_t n, double p, unsigned int seed, const unsigned int* seed_device, void* stream) {
    return relu_dropout_fwd(dtype, x, y, n, p, seed, seed_device, S(stream));
}
int saicv_relu_dropout_bwd(int dtype, const void* dy, const void* y, void* dx, size_t n, double p, void* stream) {
    return relu_dropout_bwd(dtype, dy, y, dx, n, p, S(stream));
}
int saicv_attention_fwd(int dtype, const void* qkv, void* out, float* lse, int B, int N, int H, int D,
                        double scale, void* stream) {
    return attention_fwd(dtype, qkv, out, lse, B, N, H, D, scale, S(stream));
}
int saicv_attention_bwd(int dtype, const void* qkv, const void* out, const void* dout, const float* lse,
                        void* dqkv, int B, int N, int H, int D, double scale, void* stream) {
    return attention_bwd(dtype, qkv, out, dout, lse, dqkv, B, N, H, D, scale, S(stream));
}
int saicv_window_partition(int dtype, const void* x, void* out, int B, int H, int W, int C, int ws, void* stream) {
    return window_partition(dtype, x, out, B, H, W, C, ws, S(stream));
}
int saicv_window_unpartition(int dtype, const void* win, const void* addend, void* out, int B, int H, int W, int C, int ws,
                             void* stream) {
    return window_unpartition(dtype, win, addend, out, B, H, W, C, ws, S(stream));
}
int saicv_relpos_fwd(int dtype, const void* q, long q_rs, long q_bs, const float* tab_h, const float* tab_w, float* rel_h,
                     float* rel_w, int B, int heads, int Sh, int Sw, void* stream) {
    return relpos_fwd(dtype, q, q_rs, q_bs, tab_h, tab_w, rel_h, rel_w, B, heads, Sh, Sw, S(stream));
}
int saicv_relpos_bwd(int dtype, const void* q, void* dq, long q_rs, long q_bs, const float* tab_h, const float* tab_w,
                     const float* d_rel_h, const float* d_rel_w, float* dtab_h, float* dtab_w, float* ws, int B, int heads,
                     int Sh, int Sw, void* stream) {
    return relpos_bwd(dtype, q, dq, q_rs, q_bs, tab_h, tab_w, d_rel_h, d_rel_w, dtab_h, dtab_w, ws, B, heads, Sh, Sw, S(stream));
}
size_t saicv_relpos_bwd_ws_floats(int Sh, int Sw) { return relpos_bwd_ws_floats(Sh, Sw); }
int saicv_mask_loss_stats(int dtype, const void* logits, const float* targets, float* stats, int B, int M, size_t HW,
                          double alpha, double gamma, double thr, void* stream) {
    return mask_loss_stats(dtype, logits, targets, stats, B, M, HW, alpha, gamma, thr, S(stream));
}
int saicv_mask_loss_grad(int dtype, const void* logits, const float* targets, const float* coef, void* dlogits, int B,
                         int M, size_t HW, double alpha, double gamma, void* stream) {
    return mask_loss_grad(dtype, logits, targets, coef, dlogits, B, M, HW, alpha, gamma, S(stream));
}
int saicv_hyper_product_fwd(int dtype, const void* x, const void* hyper, void* out, int B, int T, int P, int C, void* stream) {
    return hyper_product_fwd(dtype, x, hyper, out, B, T, P, C, S(stream));
}
int saicv_hyper_product_bwd(int dtype, const void* x, const void* hyper, const void* dout, void* dx, float* dhyper, int B,
                            int T, int P, int C, void* stream) {
    return hyper_product_bwd(dtype, x, hyper, dout, dx, dhyper, B, T, P, C, S(stream));
}
int saicv_upsample4_fwd(int dtype, const void* low, void* out, int planes, int h, int w, void* stream) {
    return upsample4_fwd(dtype, low, out, planes, h, w, S(stream));
}
int saicv_upsample4_bwd(int dtype, const void* dhi, void* dlow, int planes, int h, int w, void* stream) {
    return upsample4_bwd(dtype, dhi, dlow, planes, h, w, S(stream));
}
int saicv_mask_loss_stats_up4(int dtype, const void* low, const float* targets, float* stats, int B, int M, int h, int w,
                              double alpha, double gamma, double thr, void* stream) {
    return mask_loss_stats_up4(dtype, low, targets, stats, B, M, h, w, alpha, gamma, thr, S(stream));
}
int saicv_mask_loss_grad_up4(int dtype, const void* low, const float* targets, const float* coef, void* dlow, int B, int M,
                             int h, int w, double alpha, double gamma, void* stream) {
    return mask_loss_grad_up4(dtype, low, targets, coef, dlow, B, M, h, w, alpha, gamma, S(stream));
}
int saicv_attention_stream_fwd(int dtype, int D, const saicv_attn_desc* desc, void* stream) {
    if (!desc) { set_error("attention_stream: null descriptor"); return -1; }
    return attention_stream(dtype, D, 0, desc, S(stream));
}
int saicv_attention_stream_bwd(int dtype, int D, const saicv_attn_desc* desc, void* stream) {
    if (!desc) { set_error("attention_stream: null descriptor"); return -1; }
    if (!desc->dout || !desc->dq || !desc->dk || !desc->dv || !desc->dsum) {
        set_error("attention_stream_bwd: dout / dq / dk / dv / dsum are required");
        return -1;
    }
    if (desc->rel_h && (!desc->d_rel_h || !desc->d_rel_w)) {
        set_error("attention_stream_bwd: d_rel_h / d_rel_w are required with rel_h / rel_w");
        return -1;
    }
    const int rc = attention_stream(dtype, D, 1, desc, S(stream));
    return rc ? rc : attention_stream(dtype, D, 2, desc, S(stream));
}

}  // extern "C"
