// Internal C++ declarations shared between the .hip translation units and capi.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace saicv {

void set_error(const char* fmt, ...);
int check_launch(const char* what);

// igemm.hip
// form: 1 = the product reads whole consecutive rows of its source (pointwise, stride 1, no padding), 2 = 3 x 3 / stride 1 / padding 1
int conv_stat_rows(int M, int Nn, int Kd, int dtype, int form = 0);
// optional epilogue extras: out = addend + row_scale[m / rows_per_scale] * (acc + bias)
struct EpiExtra {
    const void* addend = nullptr;      // same dtype / layout as out
    const float* row_scale = nullptr;  // one factor per group of rows_per_scale rows (drop-path)
    int rows_per_scale = 1;
    int act_mode = 0;                  // 1: out2 = gelu(out) ; 2: out = (acc + bias) * gelu'(addend)
    void* out2 = nullptr;
    // data-gradient extras (igemm.hip NTParams): gate bits of the addend; BatchNorm-backward partial sums of the output
    const uint8_t* addend_gate = nullptr;
    const void* bs_y = nullptr;
    const uint8_t* bs_mask = nullptr;      // nullptr: no ReLU in front (every element counts)
    const float* bs_mean = nullptr;
    const float* bs_invstd = nullptr;
    float* bs_g = nullptr;
    float* bs_gx = nullptr;
    int stat_atomic_rows = 0;              // > 0: statistics added atomically into this many rows of a zeroed buffer
};
// partial rows the data gradient (mode 1; M rows on the OH x OW pixel grid) writes with bs_*
int conv_bwd_stat_rows(int M, int OH, int OW, int Nn, int Kd, int stride, int dtype, int form = 0);
int igemm_nt(int dtype, int mode, const void* src, const void* wgt, void* out, const float* bias,
             float* stat_sum, float* stat_sq, int H, int W, int C, int OH, int OW, int R, int S,
             int stride, int pad, int M, int Nn, int Kd, int ldo, int out_f32, hipStream_t st,
             const EpiExtra* ex = nullptr);
int igemm_tn(int dtype, const void* dy, const void* src, float* dw, int H, int W, int C, int OH,
             int OW, int R, int S, int stride, int pad, int M, int Cout, int Kd, hipStream_t st,
             float* dbias = nullptr);

// pwstream.hip: weight-resident streaming kernel for small pointwise products; blocks = rows of partial statistics (0: not eligible)
int pw_stream_blocks(int dtype, int M, int Nn, int Kd, bool fused_dgrad);
int pw_stream(int M, int Nn, int Kd, const void* src, const void* wgt, void* out, float* stat_sum, float* stat_sq,
              int stat_atomic_rows, const EpiExtra* ex, int stream_out, hipStream_t st);
int pw3_stream_blocks(int dtype, int M, int Nn, int Kd);
int pw3_stream(int mode, int M, int H, int W, const void* src, const void* wgt, void* out, float* stat_sum, float* stat_sq,
               int stat_atomic_rows, const EpiExtra* ex, int stream_out, hipStream_t st);

// sam.hip
int window_partition(int dtype, const void* x, void* out, int B, int H, int W, int C, int ws, hipStream_t st);
int window_unpartition(int dtype, const void* win, const void* addend, void* out, int B, int H, int W, int C, int ws,
                       hipStream_t st);
int relpos_fwd(int dtype, const void* q, long q_rs, long q_bs, const float* tab_h, const float* tab_w, float* rel_h,
               float* rel_w, int B, int heads, int Sh, int Sw, hipStream_t st);
int relpos_bwd(int dtype, const void* q, void* dq, long q_rs, long q_bs, const float* tab_h, const float* tab_w,
               const float* d_rel_h, const float* d_rel_w, float* dtab_h, float* dtab_w, float* ws, int B, int heads, int Sh,
               int Sw, hipStream_t st);
size_t relpos_bwd_ws_floats(int Sh, int Sw);
// maskloss.hip
int mask_loss_stats(int dtype, const void* logits, const float* targets, float* stats, int B, int M, size_t HW,
                    double alpha, double gamma, double thr, hipStream_t st);
int mask_loss_grad(int dtype, const void* logits, const float* targets, const float* coef, void* dlogits, int B, int M,
                   size_t HW, double alpha, double gamma, hipStream_t st);
// samtail.hip
int hyper_product_fwd(int dtype, const void* x, const void* hyper, void* out, int B, int Tm, int P, int C, hipStream_t st);
int hyper_product_bwd(int dtype, const void* x, const void* hyper, const void* dout, void* dx, float* dhyper, int B, int Tm,
                      int P, int C, hipStream_t st);
int upsample4_fwd(int dtype, const void* low, void* out, int planes, int h, int w, hipStream_t st);
int upsample4_bwd(int dtype, const void* dhi, void* dlow, int planes, int h, int w, hipStream_t st);
int mask_loss_stats_up4(int dtype, const void* low, const float* targets, float* stats, int B, int M, int h, int w,
                        double alpha, double gamma, double thr, hipStream_t st);
int mask_loss_grad_up4(int dtype, const void* low, const float* targets, const float* coef, void* dlow, int B, int M, int h,
                       int w, double alpha, double gamma, hipStream_t st);
// attn_stream.hip: which = 0 forward, 1 dQ pass, 2 dK/dV pass; desc = const saicv_attn_desc*
int attention_stream(int dtype, int D, int which, const void* desc, hipStream_t st);

}  // namespace saicv
