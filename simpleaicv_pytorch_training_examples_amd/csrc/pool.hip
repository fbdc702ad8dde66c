// MaxPool2d(3,2,1) and global average pooling, NHWC, gfx950.
// Replaces `max_pool2d_with_indices` (+backward) and `mean.dim` of reference
// SimpleAICV/classification/backbones/resnet.py:184 (maxpool1) and :203 (avgpool).
// HBM-bound streaming kernels; one 16-byte chunk of channels per thread.
#include "common.h"
#include "saicv_internal.h"
#include "det.h"

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


// ---------------------------------------------------------------------------------------------------------------------------
// BatchNorm-apply + ReLU + MaxPool in one pass (r04): the ResNet stem, `x = self.conv1(x); x = self.maxpool1(x)` of reference
// SimpleAICV/classification/backbones/resnet.py:226-229 (ConvBnActBlock :19-48, then nn.MaxPool2d(3, 2, 1)).  The full-resolution
// activation z = relu(scale * y + shift) -- 411 MB at batch 256 -- is never written: the pooling windows read the convolution
// output y and apply the per-channel affine + ReLU on the fly (values rounded to T as bn_act_fwd would have stored them, so the
// maxima and the first-maximum tie rule are those of the unfused pair).  Backward: the gradient of the pooled tensor is gathered
// back to every input pixel through the recorded window positions (as maxpool_bwd does), gated by the ReLU (scale * y + shift
// > 0) and fed straight into the BatchNorm backward: one pass for the two per-channel sums, one that writes d y.  Neither the
// pooled gradient scattered to full resolution nor z's mask exist.  HBM per step at b256: forward 1.41 -> 0.57 GB, backward
// 2.67 -> ~1.5 GB.
// KT > 0: the window size as a compile-time constant (the stem's 3): the window's loads are issued back to back.  With a run-time K the
// kh / kw loops stay rolled and every load waits for the previous one -- nine L2 round trips per pooled element: 148 us forward and 211 us
// in the reduction pass against ~95 us of HBM time each (r06 trace).
template <typename T, int KT>
__global__ __launch_bounds__(256) void bn_relu_maxpool_fwd_kernel(const T* __restrict__ y, const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, T* __restrict__ out,
                                                                  uint8_t* __restrict__ idx, int Nimg, int H, int W, int C, int OH,
                                                                  int OW, int K, int stride, int pad) {
    constexpr int N = Chunk<T>::N;
    const int cpr = C / N;
    const size_t total = (size_t)Nimg * OH * OW * cpr;
    const size_t gstride = (size_t)gridDim.x * blockDim.x;      // a multiple of cpr (host): a thread keeps its channel chunk
    const int cb = (int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) % cpr);
    float sc[N], sh[N];
#pragma unroll
    for (int j = 0; j < N; ++j) { sc[j] = scale[cb * N + j]; sh[j] = shift[cb * N + j]; }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gstride) {
        size_t pix = i / cpr;
        const int ow = (int)(pix % OW); pix /= OW;
        const int oh = (int)(pix % OH);
        const int n = (int)(pix / OH);
        float best[N];
        int bi[N];
#pragma unroll
        for (int j = 0; j < N; ++j) { best[j] = -INFINITY; bi[j] = 0; }
        bool first = true;
        if constexpr (KT > 0) {
            u32x4 raw[KT * KT];
            bool ok[KT * KT];
#pragma unroll
            for (int kh = 0; kh < KT; ++kh) {
#pragma unroll
                for (int kw = 0; kw < KT; ++kw) {
                    const int ih = oh * stride - pad + kh, iw = ow * stride - pad + kw;
                    ok[kh * KT + kw] = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
                    const int ihc = min(max(ih, 0), H - 1), iwc = min(max(iw, 0), W - 1);
                    raw[kh * KT + kw] = ld_chunk(y + ((size_t)(n * H + ihc) * W + iwc) * C + cb * N);
                }
            }
#pragma unroll
            for (int pos = 0; pos < KT * KT; ++pos) {
                if (!ok[pos]) continue;
                float v[N];
                Chunk<T>::unpack(raw[pos], v);
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const float z = round_through<T>(fmaxf(fmaf(sc[j], v[j], sh[j]), 0.f));
                    if (first || z > best[j] || z != z) { best[j] = z; bi[j] = pos; }
                }
                first = false;
            }
        } else {
            for (int kh = 0; kh < K; ++kh) {
                const int ih = oh * stride - pad + kh;
                if ((unsigned)ih >= (unsigned)H) continue;
                for (int kw = 0; kw < K; ++kw) {
                    const int iw = ow * stride - pad + kw;
                    if ((unsigned)iw >= (unsigned)W) continue;
                    float v[N];
                    Chunk<T>::unpack(ld_chunk(y + ((size_t)(n * H + ih) * W + iw) * C + cb * N), v);
#pragma unroll
                    for (int j = 0; j < N; ++j) {
                        const float z = round_through<T>(fmaxf(fmaf(sc[j], v[j], sh[j]), 0.f));
                        if (first || z > best[j] || z != z) { best[j] = z; bi[j] = kh * K + kw; }
                    }
                    first = false;
                }
            }
        }
        const size_t o = (((size_t)(n * OH + oh)) * OW + ow) * C + cb * N;
        st_chunk(out + o, Chunk<T>::pack(best));
        uint8_t packed[N];                                  // one 8- / 4-byte store instead of N byte stores (o is a multiple of N)
#pragma unroll
        for (int j = 0; j < N; ++j) packed[j] = (uint8_t)bi[j];
        if (N == 8) {
            uint2 raw;
            __builtin_memcpy(&raw, packed, 8);
            *reinterpret_cast<uint2*>(idx + o) = raw;
        } else {
            uint32_t raw;
            __builtin_memcpy(&raw, packed, 4);
            *reinterpret_cast<uint32_t*>(idx + o) = raw;
        }
    }
}

// g[n,h,w,c] = [scale * y + shift > 0] * sum of dout over the windows whose recorded maximum sits at (h, w): the gradient that
// reaches the BatchNorm output.  (At most 2 x 2 windows cover a pixel for K <= 2 * stride + 1; the host checks that.)
template <typename T>
DEVINL void pooled_grad(float (&g)[Chunk<T>::N], const T* __restrict__ dout, const uint8_t* __restrict__ idx, const float* yv,
                        const float* sc, const float* sh, int n, int h, int w, int cb, int C, int OH, int OW, int K, int stride,
                        int pad) {
    constexpr int N = Chunk<T>::N;
#pragma unroll
    for (int j = 0; j < N; ++j) g[j] = 0.f;
    const int oh_lo = max(0, (h + pad - (K - 1) + stride - 1) / stride), oh_hi = min(OH - 1, (h + pad) / stride);
    const int ow_lo = max(0, (w + pad - (K - 1) + stride - 1) / stride), ow_hi = min(OW - 1, (w + pad) / stride);
    if (oh_hi < oh_lo || ow_hi < ow_lo) return;
    u32x4 cg[4];
    uint8_t ib[4][N];
    bool ok[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {                                // all eight loads in flight before the first use
        const int oh = oh_lo + (q >> 1), ow = ow_lo + (q & 1);
        ok[q] = oh <= oh_hi && ow <= ow_hi;
        const size_t o = (((size_t)(n * OH + min(oh, oh_hi))) * OW + min(ow, ow_hi)) * C + cb * N;
        cg[q] = ld_chunk(dout + o);
        if (N == 8) {
            const uint2 raw = *reinterpret_cast<const uint2*>(idx + o);
            __builtin_memcpy(ib[q], &raw, 8);
        } else {
            const uint32_t raw = *reinterpret_cast<const uint32_t*>(idx + o);
            __builtin_memcpy(ib[q], &raw, 4);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (!ok[q]) continue;
        const int oh = oh_lo + (q >> 1), ow = ow_lo + (q & 1);
        const int want = (h - (oh * stride - pad)) * K + (w - (ow * stride - pad));
        float d[N];
        Chunk<T>::unpack(cg[q], d);
#pragma unroll
        for (int j = 0; j < N; ++j) g[j] += (ib[q][j] == want) ? d[j] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < N; ++j) g[j] = fmaf(sc[j], yv[j], sh[j]) > 0.f ? g[j] : 0.f;
}

// pass 1: sums[0][c] += sum g, sums[1][c] += sum g * (y - mean) * invstd   (fp32 atomics, one pair per channel and workgroup).
// Walks the POOLED tensor (a quarter of the pixels): every pooled element sends its gradient to exactly one input position, the
// recorded maximum of its window, so the sums over input pixels are sums over pooled elements of d * [gate] * (1 | xhat) with y
// taken at that position -- the window's chunks are re-read as in the forward pass (L2 hits), the per-channel position selects.
template <typename T, int KT>
__global__ __launch_bounds__(256) void bn_relu_maxpool_bwd_reduce_kernel(const T* __restrict__ dout, const uint8_t* __restrict__ idx,
                                                                         const T* __restrict__ y, const float* __restrict__ mean,
                                                                         const float* __restrict__ invstd, const float* __restrict__ scale,
                                                                         const float* __restrict__ shift, float* __restrict__ sums,
                                                                         int Nimg, int H, int W, int C, int OH, int OW, int K, int stride,
                                                                         int pad, const saicv::DetSink det) {
    constexpr int N = Chunk<T>::N;
    const int cpr = C / N;
    const size_t total = (size_t)Nimg * OH * OW * cpr;
    const size_t gstride = (size_t)gridDim.x * blockDim.x;
    const int cb = (int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) % cpr);
    float sc[N], sh[N], mu[N], is[N], sg[N], sx[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        sc[j] = scale[cb * N + j]; sh[j] = shift[cb * N + j]; mu[j] = mean[cb * N + j]; is[j] = invstd[cb * N + j];
        sg[j] = 0.f; sx[j] = 0.f;
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gstride) {
        size_t pix = i / cpr;
        const int ow = (int)(pix % OW); pix /= OW;
        const int oh = (int)(pix % OH);
        const int n = (int)(pix / OH);
        const size_t o = (((size_t)(n * OH + oh)) * OW + ow) * C + cb * N;
        float d[N];
        Chunk<T>::unpack(ld_chunk(dout + o), d);
        uint8_t ib[N];
        if (N == 8) {
            const uint2 raw = *reinterpret_cast<const uint2*>(idx + o);
            __builtin_memcpy(ib, &raw, 8);
        } else {
            const uint32_t raw = *reinterpret_cast<const uint32_t*>(idx + o);
            __builtin_memcpy(ib, &raw, 4);
        }
        float ya[N];                                   // y at each channel's recorded maximum
#pragma unroll
        for (int j = 0; j < N; ++j) ya[j] = 0.f;
        if constexpr (KT > 0) {
            u32x4 raw[KT * KT];
#pragma unroll
            for (int kh = 0; kh < KT; ++kh) {
#pragma unroll
                for (int kw = 0; kw < KT; ++kw) {                 // (a position outside the image is never a recorded maximum: its clamped load is ignored)
                    const int ihc = min(max(oh * stride - pad + kh, 0), H - 1), iwc = min(max(ow * stride - pad + kw, 0), W - 1);
                    raw[kh * KT + kw] = ld_chunk(y + ((size_t)(n * H + ihc) * W + iwc) * C + cb * N);
                }
            }
#pragma unroll
            for (int pos = 0; pos < KT * KT; ++pos) {
                float v[N];
                Chunk<T>::unpack(raw[pos], v);
#pragma unroll
                for (int j = 0; j < N; ++j) ya[j] = (ib[j] == pos) ? v[j] : ya[j];
            }
        } else {
            for (int kh = 0; kh < K; ++kh) {
                const int ih = oh * stride - pad + kh;
                if ((unsigned)ih >= (unsigned)H) continue;
                for (int kw = 0; kw < K; ++kw) {
                    const int iw = ow * stride - pad + kw;
                    if ((unsigned)iw >= (unsigned)W) continue;
                    float v[N];
                    Chunk<T>::unpack(ld_chunk(y + ((size_t)(n * H + ih) * W + iw) * C + cb * N), v);
                    const int pos = kh * K + kw;
#pragma unroll
                    for (int j = 0; j < N; ++j) ya[j] = (ib[j] == pos) ? v[j] : ya[j];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const float g = fmaf(sc[j], ya[j], sh[j]) > 0.f ? d[j] : 0.f;
            sg[j] += g;
            sx[j] = fmaf(g, (ya[j] - mu[j]) * is[j], sx[j]);
        }
    }
    __shared__ float red[256 * 16];
#pragma unroll
    for (int j = 0; j < N; ++j) { red[threadIdx.x * 2 * N + j] = sg[j]; red[threadIdx.x * 2 * N + N + j] = sx[j]; }
    __syncthreads();
    // threads tid, tid + cpr, tid + 2 cpr, ... share a channel chunk (256 % cpr == 0)
    for (int o = threadIdx.x; o < cpr * 2 * N; o += 256) {
        const int c0 = o / (2 * N), e = o - c0 * 2 * N;
        float a = 0.f;
        for (int t = c0; t < 256; t += cpr) a += red[t * 2 * N + e];
        const int which = e / N, j = e - which * N;
        saicv::det_add(det, sums + (size_t)which * C + c0 * N + j, (size_t)which * C + c0 * N + j, blockIdx.x, a);      // workgroup = partial
    }
}

// pass 2: dy = gamma * invstd * (g - mean(g) - xhat * mean(g xhat)); workgroup 0 also leaves dgamma / dbeta
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_maxpool_bwd_apply_kernel(const T* __restrict__ dout, const uint8_t* __restrict__ idx,
                                                                        const T* __restrict__ y, const float* __restrict__ gamma,
                                                                        const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                                        const float* __restrict__ sums, T* __restrict__ dy,
                                                                        float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate,
                                                                        int Nimg, int H, int W, int C, int OH, int OW, int K, int stride,
                                                                        int pad) {
    constexpr int N = Chunk<T>::N;
    const int cpr = C / N;
    const size_t total = (size_t)Nimg * H * W * cpr;
    const size_t gstride = (size_t)gridDim.x * blockDim.x;
    const int cb = (int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) % cpr);
    const float inv_m = 1.f / ((float)Nimg * (float)H * (float)W);
    float sc[N], sh[N], mu[N], is[N], k0[N], mg[N], mx[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const int c = cb * N + j;
        sc[j] = scale[c]; sh[j] = shift[c]; mu[j] = mean[c]; is[j] = invstd[c];
        k0[j] = gamma[c] * is[j];
        mg[j] = sums[c] * inv_m;
        mx[j] = sums[C + c] * inv_m;
    }
    if (blockIdx.x == 0) {
        for (int c = threadIdx.x; c < C; c += 256) {
            if (accumulate) { dbeta[c] += sums[c]; dgamma[c] += sums[C + c]; }
            else { dbeta[c] = sums[c]; dgamma[c] = sums[C + c]; }
        }
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gstride) {
        size_t pix = i / cpr;
        const int w = (int)(pix % W); pix /= W;
        const int h = (int)(pix % H);
        const int n = (int)(pix / H);
        const size_t off = ((size_t)(n * H + h) * W + w) * C + cb * N;
        float yv[N], g[N], o[N];
        Chunk<T>::unpack(ld_chunk(y + off), yv);
        pooled_grad<T>(g, dout, idx, yv, sc, sh, n, h, w, cb, C, OH, OW, K, stride, pad);
#pragma unroll
        for (int j = 0; j < N; ++j) o[j] = k0[j] * (g[j] - mg[j] - (yv[j] - mu[j]) * is[j] * mx[j]);
        st_chunk(dy + off, Chunk<T>::pack(o));
    }
}

// pass 2 for the stem's own geometry (K = 3, stride 2, padding 1; late r06): one thread per 2 x 2 block of input pixels.  The four pixels
// (2a, 2b) ... (2a + 1, 2b + 1) are covered by the same four windows (a, b) ... (a + 1, b + 1) -- by 1, 2, 2 and 4 of them -- so the
// four gradient chunks and position words are fetched ONCE per block instead of four (clamped) times per pixel: 16 memory instructions
// per four output chunks instead of 40.  Same sums in the same order as pooled_grad (windows ascending in oh, then ow).
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_maxpool_bwd_apply_k3s2_kernel(const T* __restrict__ dout, const uint8_t* __restrict__ idx,
                                                                             const T* __restrict__ y, const float* __restrict__ gamma,
                                                                             const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                                             const float* __restrict__ sums, T* __restrict__ dy,
                                                                             float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate,
                                                                             int Nimg, int H, int W, int C, int OH, int OW) {
    constexpr int N = Chunk<T>::N;
    const int cpr = C / N;
    const int HB = (H + 1) / 2, WB = (W + 1) / 2;
    const size_t total = (size_t)Nimg * HB * WB * cpr;
    const size_t gstride = (size_t)gridDim.x * blockDim.x;
    const int cb = (int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) % cpr);
    const float inv_m = 1.f / ((float)Nimg * (float)H * (float)W);
    float sc[N], sh[N], mu[N], is[N], k0[N], mg[N], mx[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const int c = cb * N + j;
        sc[j] = scale[c]; sh[j] = shift[c]; mu[j] = mean[c]; is[j] = invstd[c];
        k0[j] = gamma[c] * is[j];
        mg[j] = sums[c] * inv_m;
        mx[j] = sums[C + c] * inv_m;
    }
    if (blockIdx.x == 0) {
        for (int c = threadIdx.x; c < C; c += 256) {
            if (accumulate) { dbeta[c] += sums[c]; dgamma[c] += sums[C + c]; }
            else { dbeta[c] = sums[c]; dgamma[c] = sums[C + c]; }
        }
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gstride) {
        size_t blk = i / cpr;
        const int b = (int)(blk % WB); blk /= WB;
        const int a = (int)(blk % HB);
        const int n = (int)(blk / HB);
        // the four windows (a + dy, b + dx); one that does not exist is fetched clamped and ignored
        u32x4 cg[4];
        uint8_t ib[4][N];
        bool wok[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int oh = a + (q >> 1), ow = b + (q & 1);
            wok[q] = oh < OH && ow < OW;
            const size_t o = (((size_t)(n * OH + min(oh, OH - 1))) * OW + min(ow, OW - 1)) * C + cb * N;
            cg[q] = ld_chunk(dout + o);
            if (N == 8) {
                const uint2 raw = *reinterpret_cast<const uint2*>(idx + o);
                __builtin_memcpy(ib[q], &raw, 8);
            } else {
                const uint32_t raw = *reinterpret_cast<const uint32_t*>(idx + o);
                __builtin_memcpy(ib[q], &raw, 4);
            }
        }
        u32x4 yr[4];
        bool pok[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int h = 2 * a + (p >> 1), w = 2 * b + (p & 1);
            pok[p] = h < H && w < W;
            yr[p] = ld_chunk(y + ((size_t)(n * H + min(h, H - 1)) * W + min(w, W - 1)) * C + cb * N);
        }
        float d[4][N];
#pragma unroll
        for (int q = 0; q < 4; ++q) Chunk<T>::unpack(cg[q], d[q]);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (!pok[p]) continue;
            const int r = p >> 1, c = p & 1;
            float yv[N], g[N], o[N];
            Chunk<T>::unpack(yr[p], yv);
#pragma unroll
            for (int j = 0; j < N; ++j) g[j] = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int dyq = q >> 1, dxq = q & 1;
                if ((r == 0 && dyq == 1) || (c == 0 && dxq == 1)) continue;        // an even row / column lies in window a (b) only
                if (!wok[q]) continue;
                const int kh = r == 0 ? 1 : (dyq == 0 ? 2 : 0), kw = c == 0 ? 1 : (dxq == 0 ? 2 : 0);
                const int want = kh * 3 + kw;
#pragma unroll
                for (int j = 0; j < N; ++j) g[j] += (ib[q][j] == want) ? d[q][j] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const float gj = fmaf(sc[j], yv[j], sh[j]) > 0.f ? g[j] : 0.f;
                o[j] = k0[j] * (gj - mg[j] - (yv[j] - mu[j]) * is[j] * mx[j]);
            }
            const int h = 2 * a + r, w = 2 * b + c;
            st_chunk(dy + ((size_t)(n * H + h) * W + w) * C + cb * N, Chunk<T>::pack(o));
        }
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

int bn_relu_maxpool_fwd(int dtype, const void* y, const float* scale, const float* shift, void* out, uint8_t* idx, int Nimg, int H,
                        int W, int C, int OH, int OW, int K, int stride, int pad, hipStream_t st) {
    const int n = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    SAICV_REQUIRE(C % n == 0 && 256 % (C / n) == 0, "bn_relu_maxpool_fwd: C=%d must be %d x a power of two <= 256", C, n);
    SAICV_REQUIRE(K * K <= 255, "bn_relu_maxpool_fwd: window too large");
    const size_t total = (size_t)Nimg * OH * OW * (C / n);
    if (dtype == SAICV_DTYPE_BF16) {
        if (K == 3) hipLaunchKernelGGL((bn_relu_maxpool_fwd_kernel<bf16_t, 3>), dim3(sgrid(total)), dim3(256), 0, st, (const bf16_t*)y, scale, shift, (bf16_t*)out, idx, Nimg, H, W, C, OH, OW, K, stride, pad);
        else hipLaunchKernelGGL((bn_relu_maxpool_fwd_kernel<bf16_t, 0>), dim3(sgrid(total)), dim3(256), 0, st, (const bf16_t*)y, scale, shift, (bf16_t*)out, idx, Nimg, H, W, C, OH, OW, K, stride, pad);
    } else {
        if (K == 3) hipLaunchKernelGGL((bn_relu_maxpool_fwd_kernel<float, 3>), dim3(sgrid(total)), dim3(256), 0, st, (const float*)y, scale, shift, (float*)out, idx, Nimg, H, W, C, OH, OW, K, stride, pad);
        else hipLaunchKernelGGL((bn_relu_maxpool_fwd_kernel<float, 0>), dim3(sgrid(total)), dim3(256), 0, st, (const float*)y, scale, shift, (float*)out, idx, Nimg, H, W, C, OH, OW, K, stride, pad);
    }
    return check_launch("bn_relu_maxpool_fwd");
}

int bn_relu_maxpool_bwd(int dtype, const void* dout, const uint8_t* idx, const void* y, const float* gamma, const float* mean,
                        const float* invstd, const float* scale, const float* shift, void* dy, float* dgamma, float* dbeta,
                        int accumulate, float* ws, int Nimg, int H, int W, int C, int OH, int OW, int K, int stride, int pad,
                        hipStream_t st) {
    const int n = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    SAICV_REQUIRE(C % n == 0 && 256 % (C / n) == 0, "bn_relu_maxpool_bwd: C=%d must be %d x a power of two <= 256", C, n);
    SAICV_REQUIRE(K <= 2 * stride + 1, "bn_relu_maxpool_bwd: K=%d, stride=%d: more than 2 x 2 windows cover a pixel", K, stride);
    if (hipMemsetAsync(ws, 0, 2 * (size_t)C * sizeof(float), st) != hipSuccess) { set_error("bn_relu_maxpool_bwd: memset failed"); return -1; }
    const size_t total = (size_t)Nimg * H * W * (C / n), ptotal = (size_t)Nimg * OH * OW * (C / n);
    // the stem's own geometry: one thread per 2 x 2 pixel block (SAICV_POOL_APPLY_QUAD=0: the per-pixel form, for the A/B)
    static const bool quad_ok = !(getenv("SAICV_POOL_APPLY_QUAD") && atoi(getenv("SAICV_POOL_APPLY_QUAD")) == 0);
    const bool k3s2 = quad_ok && K == 3 && stride == 2 && pad == 1 && OH == (H + 1) / 2 && OW == (W + 1) / 2;
    const size_t qtotal = (size_t)Nimg * ((H + 1) / 2) * ((W + 1) / 2) * (C / n);
    DetParts det;
    if (det.begin(st, sgrid(ptotal), (size_t)2 * C, "bn_relu_maxpool_bwd")) return -1;
    if (dtype == SAICV_DTYPE_BF16) {
        if (K == 3) hipLaunchKernelGGL((bn_relu_maxpool_bwd_reduce_kernel<bf16_t, 3>), dim3(sgrid(ptotal)), dim3(256), 0, st, (const bf16_t*)dout, idx, (const bf16_t*)y, mean, invstd, scale, shift, ws, Nimg, H, W, C, OH, OW, K, stride, pad, det.sink());
        else hipLaunchKernelGGL((bn_relu_maxpool_bwd_reduce_kernel<bf16_t, 0>), dim3(sgrid(ptotal)), dim3(256), 0, st, (const bf16_t*)dout, idx, (const bf16_t*)y, mean, invstd, scale, shift, ws, Nimg, H, W, C, OH, OW, K, stride, pad, det.sink());
        if (det.fold(ws, 0, (size_t)2 * C)) return -1;
        if (k3s2) hipLaunchKernelGGL(bn_relu_maxpool_bwd_apply_k3s2_kernel<bf16_t>, dim3(sgrid(qtotal)), dim3(256), 0, st, (const bf16_t*)dout, idx, (const bf16_t*)y, gamma, mean, invstd, scale, shift, ws, (bf16_t*)dy, dgamma, dbeta, accumulate, Nimg, H, W, C, OH, OW);
        else hipLaunchKernelGGL(bn_relu_maxpool_bwd_apply_kernel<bf16_t>, dim3(sgrid(total)), dim3(256), 0, st, (const bf16_t*)dout, idx, (const bf16_t*)y, gamma, mean, invstd, scale, shift, ws, (bf16_t*)dy, dgamma, dbeta, accumulate, Nimg, H, W, C, OH, OW, K, stride, pad);
    } else {
        if (K == 3) hipLaunchKernelGGL((bn_relu_maxpool_bwd_reduce_kernel<float, 3>), dim3(sgrid(ptotal)), dim3(256), 0, st, (const float*)dout, idx, (const float*)y, mean, invstd, scale, shift, ws, Nimg, H, W, C, OH, OW, K, stride, pad, det.sink());
        else hipLaunchKernelGGL((bn_relu_maxpool_bwd_reduce_kernel<float, 0>), dim3(sgrid(ptotal)), dim3(256), 0, st, (const float*)dout, idx, (const float*)y, mean, invstd, scale, shift, ws, Nimg, H, W, C, OH, OW, K, stride, pad, det.sink());
        if (det.fold(ws, 0, (size_t)2 * C)) return -1;
        if (k3s2) hipLaunchKernelGGL(bn_relu_maxpool_bwd_apply_k3s2_kernel<float>, dim3(sgrid(qtotal)), dim3(256), 0, st, (const float*)dout, idx, (const float*)y, gamma, mean, invstd, scale, shift, ws, (float*)dy, dgamma, dbeta, accumulate, Nimg, H, W, C, OH, OW);
        else hipLaunchKernelGGL(bn_relu_maxpool_bwd_apply_kernel<float>, dim3(sgrid(total)), dim3(256), 0, st, (const float*)dout, idx, (const float*)y, gamma, mean, invstd, scale, shift, ws, (float*)dy, dgamma, dbeta, accumulate, Nimg, H, W, C, OH, OW, K, stride, pad);
    }
    return check_launch("bn_relu_maxpool_bwd");
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
