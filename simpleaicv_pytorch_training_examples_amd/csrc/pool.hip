// MaxPool2d(3,2,1) and global average pooling, NHWC, gfx950.
// Replaces `max_pool2d_with_indices` (+backward) and `mean.dim` of reference
// SimpleAICV/classification/backbones/resnet.py:184 (maxpool1) and :203 (avgpool).
// HBM-bound streaming kernels; one 16-byte chunk of channels per thread.
#include "common.h"
#include "saicv_internal.h"

namespace {

// Forward: out[n,oh,ow,c] = max over the KxK window; idx = window position (kh*K+kw) of the
// FIRST maximum in scan order (ATen CPU `max_pool2d_with_indices` tie rule: strict '>' or NaN).
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ out,
                                                          uint8_t* __restrict__ idx, int Nimg, int H,
                                                          int W, int C, int OH, int OW, int K,
                                                          int stride, int pad) {
    constexpr int N = Chunk<T>::N;
    const int cpr = C / N;
    const size_t total = (size_t)Nimg * OH * OW * cpr;
    const size_t gstride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gstride) {
        const int cb = (int)(i % cpr);
        size_t pix = i / cpr;
        const int ow = (int)(pix % OW); pix /= OW;
        const int oh = (int)(pix % OH);
        const int n = (int)(pix / OH);
        float best[N];
        int bi[N];
#pragma unroll
        for (int j = 0; j < N; ++j) { best[j] = -INFINITY; bi[j] = 0; }
        bool first = true;
        for (int kh = 0; kh < K; ++kh) {
            const int ih = oh * stride - pad + kh;
            if ((unsigned)ih >= (unsigned)H) continue;
            for (int kw = 0; kw < K; ++kw) {
                const int iw = ow * stride - pad + kw;
                if ((unsigned)iw >= (unsigned)W) continue;
                float v[N];
                Chunk<T>::unpack(ld_chunk(x + ((size_t)(n * H + ih) * W + iw) * C + cb * N), v);
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    if (first || v[j] > best[j] || v[j] != v[j]) { best[j] = v[j]; bi[j] = kh * K + kw; }
                }
                first = false;
            }
        }
        const size_t o = (((size_t)(n * OH + oh)) * OW + ow) * C + cb * N;
        st_chunk(out + o, Chunk<T>::pack(best));
#pragma unroll
        for (int j = 0; j < N; ++j) idx[o + j] = (uint8_t)bi[j];
    }
}

// Backward (gather form, no atomics): dx[n,h,w,c] = sum of dout over windows whose recorded
// argmax is (h,w).
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ dout,
                                                          const uint8_t* __restrict__ idx,
                                                          T* __restrict__ dx, int Nimg, int H, int W,
                                                          int C, int OH, int OW, int K, int stride,
                                                          int pad) {
    constexpr int N = Chunk<T>::N;
    const int cpr = C / N;
    const size_t total = (size_t)Nimg * H * W * cpr;
    const size_t gstride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gstride) {
        const int cb = (int)(i % cpr);
        size_t pix = i / cpr;
        const int w = (int)(pix % W); pix /= W;
        const int h = (int)(pix % H);
        const int n = (int)(pix / H);
        float acc[N];
#pragma unroll
        for (int j = 0; j < N; ++j) acc[j] = 0.f;
        // windows oh with oh*stride - pad <= h <= oh*stride - pad + K-1
        const int oh_lo = max(0, (h + pad - (K - 1) + stride - 1) / stride);
        const int oh_hi = min(OH - 1, (h + pad) / stride);
        const int ow_lo = max(0, (w + pad - (K - 1) + stride - 1) / stride);
        const int ow_hi = min(OW - 1, (w + pad) / stride);
        auto gather = [&](int oh, int ow, const u32x4& cg, const uint8_t* ib) {
            const int want = (h - (oh * stride - pad)) * K + (w - (ow * stride - pad));
            float g[N];
            Chunk<T>::unpack(cg, g);
#pragma unroll
            for (int j = 0; j < N; ++j) acc[j] += (ib[j] == want) ? g[j] : 0.f;
        };
        auto load_idx = [&](size_t o, uint8_t* ib) {                 // N index bytes, contiguous
            if (N == 8) {
                const uint2 raw = *reinterpret_cast<const uint2*>(idx + o);
                __builtin_memcpy(ib, &raw, 8);
            } else {
                const uint32_t raw = *reinterpret_cast<const uint32_t*>(idx + o);
                __builtin_memcpy(ib, &raw, 4);
            }
        };
        if (oh_hi < oh_lo || ow_hi < ow_lo) {
            // no window covers this pixel: the gradient is zero
        } else if (oh_hi - oh_lo <= 1 && ow_hi - ow_lo <= 1) {
            // the usual case (K <= 2 * stride + 1... at most 2 x 2 windows cover a pixel): all eight loads in flight before
            // the first use -- the one-window-at-a-time loop ran at 2.5 TB/s
            u32x4 cg[4];
            uint8_t ib[4][N];
            bool ok[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int oh = oh_lo + (q >> 1), ow = ow_lo + (q & 1);
                ok[q] = oh <= oh_hi && ow <= ow_hi;
                const size_t o = (((size_t)(n * OH + min(oh, oh_hi))) * OW + min(ow, ow_hi)) * C + cb * N;
                cg[q] = ld_chunk(dout + o);
                load_idx(o, ib[q]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (ok[q]) gather(oh_lo + (q >> 1), ow_lo + (q & 1), cg[q], ib[q]);
        } else {
            for (int oh = oh_lo; oh <= oh_hi; ++oh)
                for (int ow = ow_lo; ow <= ow_hi; ++ow) {
                    const size_t o = (((size_t)(n * OH + oh)) * OW + ow) * C + cb * N;
                    uint8_t ib[N];
                    load_idx(o, ib);
                    gather(oh, ow, ld_chunk(dout + o), ib);
                }
        }
        st_chunk(dx + ((size_t)(n * H + h) * W + w) * C + cb * N, Chunk<T>::pack(acc));
    }
}

// Global average pool: x [Nimg, HW, C] -> out [Nimg, C]  (out dtype T)
template <typename T>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ out,
                                                          int Nimg, int HW, int C) {
    constexpr int N = Chunk<T>::N;
    const int cpr = C / N;
    const size_t total = (size_t)Nimg * cpr;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int cb = (int)(i % cpr);
    const int n = (int)(i / cpr);
    float acc[N];
#pragma unroll
    for (int j = 0; j < N; ++j) acc[j] = 0.f;
    for (int p = 0; p < HW; ++p) {
        float v[N];
        Chunk<T>::unpack(ld_chunk(x + ((size_t)n * HW + p) * C + cb * N), v);
#pragma unroll
        for (int j = 0; j < N; ++j) acc[j] += v[j];
    }
    const float inv = 1.f / (float)HW;
#pragma unroll
    for (int j = 0; j < N; ++j) acc[j] *= inv;
    st_chunk(out + (size_t)n * C + cb * N, Chunk<T>::pack(acc));
}

template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const T* __restrict__ dout, T* __restrict__ dx,
                                                          int Nimg, int HW, int C) {
    constexpr int N = Chunk<T>::N;
    const int cpr = C / N;
    const size_t total = (size_t)Nimg * HW * cpr;
    const size_t gstride = (size_t)gridDim.x * blockDim.x;
    const float inv = 1.f / (float)HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gstride) {
        const int cb = (int)(i % cpr);
        const size_t pix = i / cpr;
        const int n = (int)(pix / HW);
        float g[N];
        Chunk<T>::unpack(ld_chunk(dout + (size_t)n * C + cb * N), g);
#pragma unroll
        for (int j = 0; j < N; ++j) g[j] *= inv;
        st_chunk(dx + pix * C + cb * N, Chunk<T>::pack(g));
    }
}

inline int sgrid(size_t total) {
    size_t b = (total + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

namespace saicv {

int maxpool_fwd(int dtype, const void* x, void* out, uint8_t* idx, int Nimg, int H, int W, int C,
                int OH, int OW, int K, int stride, int pad, hipStream_t st) {
    const int n = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    SAICV_REQUIRE(C % n == 0, "maxpool_fwd: C=%d must be a multiple of %d", C, n);
    SAICV_REQUIRE(K * K <= 255, "maxpool_fwd: window too large");
    const size_t total = (size_t)Nimg * OH * OW * (C / n);
    if (dtype == SAICV_DTYPE_BF16)
        hipLaunchKernelGGL(maxpool_fwd_kernel<bf16_t>, dim3(sgrid(total)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)out, idx, Nimg, H, W, C, OH, OW, K, stride, pad);
    else
        hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(sgrid(total)), dim3(256), 0, st, (const float*)x, (float*)out, idx, Nimg, H, W, C, OH, OW, K, stride, pad);
    return check_launch("maxpool_fwd");
}

int maxpool_bwd(int dtype, const void* dout, const uint8_t* idx, void* dx, int Nimg, int H, int W,
                int C, int OH, int OW, int K, int stride, int pad, hipStream_t st) {
    const int n = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    SAICV_REQUIRE(C % n == 0, "maxpool_bwd: C=%d must be a multiple of %d", C, n);
    const size_t total = (size_t)Nimg * H * W * (C / n);
    if (dtype == SAICV_DTYPE_BF16)
        hipLaunchKernelGGL(maxpool_bwd_kernel<bf16_t>, dim3(sgrid(total)), dim3(256), 0, st, (const bf16_t*)dout, idx, (bf16_t*)dx, Nimg, H, W, C, OH, OW, K, stride, pad);
    else
        hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(sgrid(total)), dim3(256), 0, st, (const float*)dout, idx, (float*)dx, Nimg, H, W, C, OH, OW, K, stride, pad);
    return check_launch("maxpool_bwd");
}

int avgpool_fwd(int dtype, const void* x, void* out, int Nimg, int HW, int C, hipStream_t st) {
    const int n = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    SAICV_REQUIRE(C % n == 0, "avgpool_fwd: C=%d must be a multiple of %d", C, n);
    const size_t total = (size_t)Nimg * (C / n);
    const int grid = (int)((total + 255) / 256);
    if (dtype == SAICV_DTYPE_BF16)
        hipLaunchKernelGGL(avgpool_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)out, Nimg, HW, C);
    else
        hipLaunchKernelGGL(avgpool_fwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x, (float*)out, Nimg, HW, C);
    return check_launch("avgpool_fwd");
}

int avgpool_bwd(int dtype, const void* dout, void* dx, int Nimg, int HW, int C, hipStream_t st) {
    const int n = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    SAICV_REQUIRE(C % n == 0, "avgpool_bwd: C=%d must be a multiple of %d", C, n);
    const size_t total = (size_t)Nimg * HW * (C / n);
    if (dtype == SAICV_DTYPE_BF16)
        hipLaunchKernelGGL(avgpool_bwd_kernel<bf16_t>, dim3(sgrid(total)), dim3(256), 0, st, (const bf16_t*)dout, (bf16_t*)dx, Nimg, HW, C);
    else
        hipLaunchKernelGGL(avgpool_bwd_kernel<float>, dim3(sgrid(total)), dim3(256), 0, st, (const float*)dout, (float*)dx, Nimg, HW, C);
    return check_launch("avgpool_bwd");
}

}  // namespace saicv
