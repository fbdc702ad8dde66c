// Softmax cross-entropy (hard and soft labels), fp32, one wavefront per sample.
// Replaces `_log_softmax` + `nll_loss_forward` (+backward) of reference
// SimpleAICV/classification/losses.py:21-28 (CELoss) and :86-91 (OneHotLabelCELoss).
// Forward also emits the gradient w.r.t. the logits (scaled by 1/B; the caller multiplies by
// the upstream scalar), so backward is a single scale pass.
#include "common.h"
#include "saicv_internal.h"

namespace {

// logits [B][C] fp32, label int64 [B]  ->  row_loss[B], dlogits[B][C] = (softmax - onehot)/B
__global__ __launch_bounds__(256) void ce_hard_kernel(const float* __restrict__ logits,
                                                      const int64_t* __restrict__ label, int B, int C,
                                                      float* __restrict__ row_loss,
                                                      float* __restrict__ dlogits) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= B) return;
    const float* x = logits + (size_t)row * C;
    float mx = -INFINITY;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, x[c]);
    mx = wave_max(mx);
    float se = 0.f;
    for (int c = lane; c < C; c += 64) se += expf(x[c] - mx);
    se = wave_sum(se);
    const float lse = mx + logf(se);
    const int64_t t = label[row];
    const bool valid = t >= 0 && t < C;          // a label outside [0, C) contributes neither loss nor gradient
    if (lane == 0) row_loss[row] = valid ? (lse - x[t]) : 0.f;
    if (dlogits) {
        const float invB = valid ? 1.f / (float)B : 0.f;
        float* d = dlogits + (size_t)row * C;
        for (int c = lane; c < C; c += 64) {
            float pr = expf(x[c] - lse);
            if (c == t) pr -= 1.f;
            d[c] = pr * invB;
        }
    }
}

// soft labels y [B][C] fp32:  loss_row = sum(-y * log_softmax(x));  d = (softmax*sum(y) - y)/B
__global__ __launch_bounds__(256) void ce_soft_kernel(const float* __restrict__ logits,
                                                      const float* __restrict__ y, int B, int C,
                                                      float* __restrict__ row_loss,
                                                      float* __restrict__ dlogits) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= B) return;
    const float* x = logits + (size_t)row * C;
    const float* yy = y + (size_t)row * C;
    float mx = -INFINITY;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, x[c]);
    mx = wave_max(mx);
    float se = 0.f, sy = 0.f, sxy = 0.f;
    for (int c = lane; c < C; c += 64) {
        se += expf(x[c] - mx);
        sy += yy[c];
        sxy += yy[c] * x[c];
    }
    se = wave_sum(se);
    sy = wave_sum(sy);
    sxy = wave_sum(sxy);
    const float lse = mx + logf(se);
    if (lane == 0) row_loss[row] = lse * sy - sxy;
    if (dlogits) {
        const float invB = 1.f / (float)B;
        float* d = dlogits + (size_t)row * C;
        for (int c = lane; c < C; c += 64) d[c] = (expf(x[c] - lse) * sy - yy[c]) * invB;
    }
}

// mean over rows (single block, deterministic order)
__global__ __launch_bounds__(256) void mean_rows_kernel(const float* __restrict__ v, int B,
                                                        float* __restrict__ out) {
    __shared__ float part[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < B; i += 256) s += v[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (part[0] + part[1] + part[2] + part[3]) / (float)B;
}

// out[i] = in[i] * scale[0]   (in fp32; out T)
template <typename T>
__global__ __launch_bounds__(256) void scale_by_scalar_kernel(const float* __restrict__ in,
                                                              const float* __restrict__ scale,
                                                              T* __restrict__ out, size_t n) {
    const float s = scale[0];
    const size_t gstride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gstride)
        out[i] = from_f32<T>(in[i] * s);
}

}  // namespace

namespace saicv {

int softmax_ce_fwd(const float* logits, const void* label, int soft, int B, int C, float* row_loss,
                   float* loss, float* dlogits, hipStream_t st) {
    SAICV_REQUIRE(B > 0 && C > 0, "softmax_ce_fwd: empty problem");
    const int grid = (B + 3) / 4;
    if (soft)
        hipLaunchKernelGGL(ce_soft_kernel, dim3(grid), dim3(256), 0, st, logits, (const float*)label, B, C, row_loss, dlogits);
    else
        hipLaunchKernelGGL(ce_hard_kernel, dim3(grid), dim3(256), 0, st, logits, (const int64_t*)label, B, C, row_loss, dlogits);
    hipLaunchKernelGGL(mean_rows_kernel, dim3(1), dim3(256), 0, st, row_loss, B, loss);
    return check_launch("softmax_ce_fwd");
}

int scale_by_scalar(int out_dtype, const float* in, const float* scale, void* out, size_t n,
                    hipStream_t st) {
    size_t b = (n + 255) / 256;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    if (out_dtype == SAICV_DTYPE_BF16)
        hipLaunchKernelGGL(scale_by_scalar_kernel<bf16_t>, dim3((int)b), dim3(256), 0, st, in, scale, (bf16_t*)out, n);
    else
        hipLaunchKernelGGL(scale_by_scalar_kernel<float>, dim3((int)b), dim3(256), 0, st, in, scale, (float*)out, n);
    return check_launch("scale_by_scalar");
}

}  // namespace saicv
