// SAM mask-loss statistics in ONE streaming pass over the [B, M, H*W] logits.
//
// Replaces the ~15 full-resolution elementwise / reduction ops of SAMLoss.focal_loss / dice_loss /
// iou_predict_loss (reference SimpleAICV/interactive_segmentation/losses.py:136-198): per (sample b, mask m)
//   stats[0] = sum focal(x, t)         focal = (a t + (1-a)(1-t)) (1 - pt)^g bce(x, t),  pt = p t + (1-p)(1-t)
//   stats[1] = sum sigmoid(x) * t      (dice intersection)
//   stats[2] = sum sigmoid(x)
//   stats[3] = sum t
//   stats[4] = #{x > thr and t > thr}  (IoU intersection)
//   stats[5] = #{x > thr or  t > thr}  (IoU union)
// and, backward, d logits = c0 dfocal/dx + (c1 t + c2) p (1 - p) with per-(b, m) coefficients.
// HBM-bound: algorithmic bytes = logits + targets read once (forward), + d logits written (backward).
#include "common.h"
#include "saicv_internal.h"
#include "det.h"

namespace {

constexpr int ML_THREADS = 256;
constexpr int ML_CHUNKS = 8;            // 16-byte logit chunks per thread

struct MaskTerm {
    float p, bce, one_m_pt, af;
};

DEVINL MaskTerm mask_term(float x, float t, float alpha) {
    MaskTerm r;
    const float e = expf(-fabsf(x));
    r.p = x >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
    r.bce = fmaxf(x, 0.f) - x * t + log1pf(e);
    r.one_m_pt = 1.f - (r.p * t + (1.f - r.p) * (1.f - t));
    r.af = alpha * t + (1.f - alpha) * (1.f - t);
    return r;
}

DEVINL float pow_gamma(float v, float gamma) {
    return gamma == 2.f ? v * v : powf(fmaxf(v, 0.f), gamma);
}

template <typename T>
__global__ __launch_bounds__(ML_THREADS) void mask_loss_stats_kernel(const T* __restrict__ logits,
                                                                      const float* __restrict__ targets,
                                                                      float* __restrict__ stats, int M, size_t HW,
                                                                      float alpha, float gamma, float thr, const saicv::DetSink det) {
    constexpr int N = Chunk<T>::N;
    const int bm = blockIdx.y, b = bm / M;
    const T* lg = logits + (size_t)bm * HW;
    const float* tg = targets + (size_t)b * HW;
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const size_t nchunks = HW / N;
    const size_t base = (size_t)blockIdx.x * ML_THREADS * ML_CHUNKS + threadIdx.x;
#pragma unroll
    for (int j = 0; j < ML_CHUNKS; ++j) {
        const size_t c = base + (size_t)j * ML_THREADS;
        if (c < nchunks) {
            float x[N], t[N];
            Chunk<T>::unpack(ld_chunk(lg + c * N), x);
#pragma unroll
            for (int q = 0; q < N / 4; ++q) {
                const f32x4 tv = *reinterpret_cast<const f32x4*>(tg + c * N + q * 4);
                t[q * 4] = tv[0]; t[q * 4 + 1] = tv[1]; t[q * 4 + 2] = tv[2]; t[q * 4 + 3] = tv[3];
            }
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const MaskTerm m = mask_term(x[k], t[k], alpha);
                acc[0] += m.af * pow_gamma(m.one_m_pt, gamma) * m.bce;
                acc[1] += m.p * t[k];
                acc[2] += m.p;
                acc[3] += t[k];
                const bool pi = x[k] > thr, ti = t[k] > thr;
                acc[4] += (pi && ti) ? 1.f : 0.f;
                acc[5] += (pi || ti) ? 1.f : 0.f;
            }
        }
    }
    __shared__ float red[ML_THREADS / 64][6];
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[i] = wave_sum(acc[i]);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
#pragma unroll
        for (int i = 0; i < 6; ++i) red[wave][i] = acc[i];
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < ML_THREADS / 64; ++w) v += red[w][threadIdx.x];
        saicv::det_add(det, &stats[(size_t)bm * 6 + threadIdx.x], (size_t)bm * 6 + threadIdx.x, blockIdx.x, v);      // pixel slab = partial
    }
}

template <typename T>
__global__ __launch_bounds__(ML_THREADS) void mask_loss_grad_kernel(const T* __restrict__ logits,
                                                                     const float* __restrict__ targets,
                                                                     const float* __restrict__ coef,
                                                                     T* __restrict__ dlogits, int M, size_t HW,
                                                                     float alpha, float gamma) {
    constexpr int N = Chunk<T>::N;
    const int bm = blockIdx.y, b = bm / M;
    const T* lg = logits + (size_t)bm * HW;
    T* dg = dlogits + (size_t)bm * HW;
    const float* tg = targets + (size_t)b * HW;
    const float c0 = coef[bm * 3], c1 = coef[bm * 3 + 1], c2 = coef[bm * 3 + 2];
    const size_t nchunks = HW / N;
    const size_t base = (size_t)blockIdx.x * ML_THREADS * ML_CHUNKS + threadIdx.x;
#pragma unroll
    for (int j = 0; j < ML_CHUNKS; ++j) {
        const size_t c = base + (size_t)j * ML_THREADS;
        if (c < nchunks) {
            float x[N], t[N], g[N];
            Chunk<T>::unpack(ld_chunk(lg + c * N), x);
#pragma unroll
            for (int q = 0; q < N / 4; ++q) {
                const f32x4 tv = *reinterpret_cast<const f32x4*>(tg + c * N + q * 4);
                t[q * 4] = tv[0]; t[q * 4 + 1] = tv[1]; t[q * 4 + 2] = tv[2]; t[q * 4 + 3] = tv[3];
            }
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const MaskTerm m = mask_term(x[k], t[k], alpha);
                const float sp = m.p * (1.f - m.p);
                const float dpt = sp * (2.f * t[k] - 1.f);
                const float w_g = pow_gamma(m.one_m_pt, gamma);
                const float w_gm1 = gamma == 2.f ? m.one_m_pt : powf(fmaxf(m.one_m_pt, 0.f), gamma - 1.f);
                const float dfocal = m.af * (gamma * w_gm1 * (-dpt) * m.bce + w_g * (m.p - t[k]));
                g[k] = c0 * dfocal + (c1 * t[k] + c2) * sp;
            }
            st_chunk(dg + c * N, Chunk<T>::pack(g));
        }
    }
}

}  // namespace

namespace saicv {

int mask_loss_stats(int dtype, const void* logits, const float* targets, float* stats, int B, int M, size_t HW,
                    double alpha, double gamma, double thr, hipStream_t st) {
    const int n = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    SAICV_REQUIRE(B > 0 && M > 0 && HW > 0 && HW % n == 0, "mask_loss_stats: H*W=%zu must be a positive multiple of %d", HW, n);
    hipMemsetAsync(stats, 0, (size_t)B * M * 6 * sizeof(float), st);
    const size_t per_block = (size_t)ML_THREADS * ML_CHUNKS * n;
    dim3 grid((unsigned)((HW + per_block - 1) / per_block), B * M);
    DetParts det;
    if (det.begin(st, (int)grid.x, (size_t)B * M * 6, "mask_loss_stats")) return -1;
    if (dtype == SAICV_DTYPE_BF16)
        hipLaunchKernelGGL(mask_loss_stats_kernel<bf16_t>, grid, dim3(ML_THREADS), 0, st, (const bf16_t*)logits, targets,
                           stats, M, HW, (float)alpha, (float)gamma, (float)thr, det.sink());
    else
        hipLaunchKernelGGL(mask_loss_stats_kernel<float>, grid, dim3(ML_THREADS), 0, st, (const float*)logits, targets,
                           stats, M, HW, (float)alpha, (float)gamma, (float)thr, det.sink());
    if (check_launch("mask_loss_stats")) return -2;
    return det.fold(stats, 0, (size_t)B * M * 6);
}

int mask_loss_grad(int dtype, const void* logits, const float* targets, const float* coef, void* dlogits, int B, int M,
                   size_t HW, double alpha, double gamma, hipStream_t st) {
    const int n = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    SAICV_REQUIRE(B > 0 && M > 0 && HW > 0 && HW % n == 0, "mask_loss_grad: H*W=%zu must be a positive multiple of %d", HW, n);
    const size_t per_block = (size_t)ML_THREADS * ML_CHUNKS * n;
    dim3 grid((unsigned)((HW + per_block - 1) / per_block), B * M);
    if (dtype == SAICV_DTYPE_BF16)
        hipLaunchKernelGGL(mask_loss_grad_kernel<bf16_t>, grid, dim3(ML_THREADS), 0, st, (const bf16_t*)logits, targets,
                           coef, (bf16_t*)dlogits, M, HW, (float)alpha, (float)gamma);
    else
        hipLaunchKernelGGL(mask_loss_grad_kernel<float>, grid, dim3(ML_THREADS), 0, st, (const float*)logits, targets,
                           coef, (float*)dlogits, M, HW, (float)alpha, (float)gamma);
    return check_launch("mask_loss_grad");
}

}  // namespace saicv
