// The HBM-heavy tail of SAM's mask decoder and loss on gfx950:
//
//   hyper_product_{fwd,bwd}   masks[b, t, p] = <hyper[b, t, :], x[b, p, :]>, C = 32 channels, T <= 8 mask tokens
//                             (reference segment_anything/mask_decoder.py:137-140: `hyper_in @ upscaled_embedding`)
//   upsample4_{fwd,bwd}       F.interpolate(low, 4x, mode="bilinear", align_corners=False)
//                             (reference segment_anything/sam.py:155-158, 256^2 -> 1024^2)
//   mask_loss_stats_up4       the six SAMLoss sums of maskloss.hip taken from the LOW-RESOLUTION logits: every
//                             full-resolution logit is interpolated in registers, never stored
//   mask_loss_grad_up4        d loss / d LOW-RESOLUTION logits in one pass: each low-resolution pixel gathers the
//                             gradient of the 8 x 8 full-resolution pixels whose bilinear footprint contains it
//                             (reference losses.py:136-198 + the backward of F.interpolate)
//
// All HBM-bound streaming kernels: per sample and decoder pass the reference moves a [4, 1024, 1024] fp32 tensor through
// interpolate, ~15 loss ops and their backwards (336 MB per tensor at batch 20); here the loss reads the 4 MB target and
// the 1 MB of low-resolution logits (L2-resident), forward and backward.
//
// Bilinear rule (ATen upsample_bilinear2d, align_corners = false, scale 1/4): src = max(0.25 (d + 0.5) - 0.5, 0),
// i0 = floor(src), i1 = min(i0 + 1, n - 1), l1 = src - i0, l0 = 1 - l1; out = l0h (l0w v00 + l1w v01) + l1h (l0w v10 + l1w v11).
#include <stdlib.h>
#include "common.h"
#include "saicv_internal.h"
#include "det.h"

namespace {

constexpr int ST_THREADS = 256;

struct Tap { int i0, i1; float l0, l1; };

DEVINL Tap tap4(int d, int n) {
    float src = 0.25f * ((float)d + 0.5f) - 0.5f;
    src = fmaxf(src, 0.f);
    Tap t;
    t.i0 = (int)src;
    t.i1 = min(t.i0 + 1, n - 1);
    t.l1 = src - (float)t.i0;
    t.l0 = 1.f - t.l1;
    return t;
}

// ---- maskloss.hip's per-element terms (kept identical)
struct MaskTerm { float p, bce, one_m_pt, af; };
DEVINL MaskTerm mask_term(float x, float t, float alpha) {
    MaskTerm r;
    const float e = expf(-fabsf(x));
    r.p = x >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
    r.bce = fmaxf(x, 0.f) - x * t + log1pf(e);
    r.one_m_pt = 1.f - (r.p * t + (1.f - r.p) * (1.f - t));
    r.af = alpha * t + (1.f - alpha) * (1.f - t);
    return r;
}
DEVINL float pow_gamma(float v, float gamma) { return gamma == 2.f ? v * v : powf(fmaxf(v, 0.f), gamma); }
DEVINL float loss_grad(float x, float t, float alpha, float gamma, float c0, float c1, float c2) {
    const MaskTerm m = mask_term(x, t, alpha);
    const float sp = m.p * (1.f - m.p);
    const float dpt = sp * (2.f * t - 1.f);
    const float w_g = pow_gamma(m.one_m_pt, gamma);
    const float w_gm1 = gamma == 2.f ? m.one_m_pt : powf(fmaxf(m.one_m_pt, 0.f), gamma - 1.f);
    const float dfocal = m.af * (gamma * w_gm1 * (-dpt) * m.bce + w_g * (m.p - t));
    return c0 * dfocal + (c1 * t + c2) * sp;
}

// ------------------------------------------------------------------------------------------ hyper-network product
// one thread per pixel: 32 channels = 64 (bf16) / 128 (fp32) contiguous bytes; hyper vectors broadcast from LDS
template <typename T, int C>
__global__ __launch_bounds__(ST_THREADS) void hyper_fwd_kernel(const T* __restrict__ x, const T* __restrict__ hyper,
                                                                T* __restrict__ out, int Tm, int P) {
    __shared__ float hs[8 * C];
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < Tm * C; i += ST_THREADS) hs[i] = to_f32(hyper[(size_t)b * Tm * C + i]);
    __syncthreads();
    const int p = blockIdx.x * ST_THREADS + threadIdx.x;
    if (p >= P) return;
    constexpr int N = Chunk<T>::N;
    float v[C];
    const T* xp = x + ((size_t)b * P + p) * C;
#pragma unroll
    for (int c = 0; c < C / N; ++c) Chunk<T>::unpack(ld_chunk(xp + c * N), v + c * N);
    for (int t = 0; t < Tm; ++t) {
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) a = fmaf(v[c], hs[t * C + c], a);
        out[((size_t)b * Tm + t) * P + p] = from_f32<T>(a);
    }
}

// dx[b, p, :] = sum_t dout[b, t, p] hyper[b, t, :];  dhyper[b, t, :] += sum_p dout[b, t, p] x[b, p, :]  (fp32, pre-zeroed)
template <typename T, int C>
__global__ __launch_bounds__(ST_THREADS) void hyper_bwd_kernel(const T* __restrict__ x, const T* __restrict__ hyper,
                                                                const T* __restrict__ dout, T* __restrict__ dx,
                                                                float* __restrict__ dhyper, int Tm, int P, const saicv::DetSink det) {
    __shared__ float hs[8 * C];
    __shared__ float xs[ST_THREADS][C + 1];
    __shared__ float ds[8][ST_THREADS];
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < Tm * C; i += ST_THREADS) hs[i] = to_f32(hyper[(size_t)b * Tm * C + i]);
    const int p = blockIdx.x * ST_THREADS + threadIdx.x;
    const bool ok = p < P;
    constexpr int N = Chunk<T>::N;
    float v[C];
#pragma unroll
    for (int c = 0; c < C; ++c) v[c] = 0.f;
    if (ok) {
        const T* xp = x + ((size_t)b * P + p) * C;
#pragma unroll
        for (int c = 0; c < C / N; ++c) Chunk<T>::unpack(ld_chunk(xp + c * N), v + c * N);
    }
#pragma unroll
    for (int c = 0; c < C; ++c) xs[threadIdx.x][c] = v[c];
    float g[8];
    for (int t = 0; t < Tm; ++t) {
        g[t] = ok ? to_f32(dout[((size_t)b * Tm + t) * P + p]) : 0.f;
        ds[t][threadIdx.x] = g[t];
    }
    __syncthreads();
    if (ok) {
        float o[C];
#pragma unroll
        for (int c = 0; c < C; ++c) o[c] = 0.f;
        for (int t = 0; t < Tm; ++t)
#pragma unroll
            for (int c = 0; c < C; ++c) o[c] = fmaf(g[t], hs[t * C + c], o[c]);
        T* dp = dx + ((size_t)b * P + p) * C;
#pragma unroll
        for (int c = 0; c < C / N; ++c) st_chunk(dp + c * N, Chunk<T>::pack(o + c * N));
    }
    // the block's share of dhyper: thread (t, c) sums its 256 pixels from LDS, one atomic per (t, c) per block
    for (int i = threadIdx.x; i < Tm * C; i += ST_THREADS) {
        const int t = i / C, c = i - t * C;
        float a = 0.f;
        for (int q = 0; q < ST_THREADS; ++q) a = fmaf(ds[t][q], xs[q][c], a);
        saicv::det_add(det, dhyper + (size_t)b * Tm * C + i, (size_t)b * Tm * C + i, blockIdx.x, a);      // pixel block = partial
    }
}

// ------------------------------------------------------------------------------------------ x4 bilinear
// thread = one low-resolution column group j of one output-row group q: the 4 x 4 outputs (rows 4q.., cols 4j..)
// need low rows {q-1, q, q+1} x cols {j-1, j, j+1} (clamped): 9 loads, 16 results, four 8- / 16-byte row stores
template <typename T>
DEVINL void load3x3(const T* __restrict__ low, int h, int w, int q, int j, float (&v)[3][3]) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int r = min(max(q - 1 + a, 0), h - 1);
#pragma unroll
        for (int c = 0; c < 3; ++c) v[a][c] = to_f32(low[(size_t)r * w + min(max(j - 1 + c, 0), w - 1)]);
    }
}
// interpolated value of output (4q + r, 4j + s) from the 3 x 3 neighbourhood, ATen's operation order
DEVINL float interp(const float (&v)[3][3], int q, int j, int r, int s, int h, int w) {
    const Tap th = tap4(4 * q + r, h), tw = tap4(4 * j + s, w);
    const int a0 = th.i0 - (q - 1), a1 = th.i1 - (q - 1), c0 = tw.i0 - (j - 1), c1 = tw.i1 - (j - 1);
    auto at = [&](int a, int c) -> float {           // indices are in 0..2 by construction; select without dynamic indexing
        float o = 0.f;
#pragma unroll
        for (int x = 0; x < 3; ++x)
#pragma unroll
            for (int y = 0; y < 3; ++y) o = (x == a && y == c) ? v[x][y] : o;
        return o;
    };
    return th.l0 * (tw.l0 * at(a0, c0) + tw.l1 * at(a0, c1)) + th.l1 * (tw.l0 * at(a1, c0) + tw.l1 * at(a1, c1));
}

template <typename T>
__global__ __launch_bounds__(ST_THREADS) void up4_fwd_kernel(const T* __restrict__ low, T* __restrict__ out, int h, int w) {
    const int plane = blockIdx.z, q = blockIdx.y;
    const int j = blockIdx.x * ST_THREADS + threadIdx.x;
    if (j >= w) return;
    float v[3][3];
    load3x3(low + (size_t)plane * h * w, h, w, q, j, v);
    T* o = out + (size_t)plane * 16 * h * w;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        T* row = o + (size_t)(4 * q + r) * 4 * w + 4 * j;
#pragma unroll
        for (int s = 0; s < 4; ++s) row[s] = from_f32<T>(interp(v, q, j, r, s, h, w));
    }
}

// contribution weights of low pixel i to output d along one axis (0 when d does not touch i)
DEVINL float wgt(int d, int i, int n) {
    const Tap t = tap4(d, n);
    return (t.i0 == i ? t.l0 : 0.f) + (t.i1 == i ? t.l1 : 0.f);
}

template <typename T>
__global__ __launch_bounds__(ST_THREADS) void up4_bwd_kernel(const T* __restrict__ dhi, T* __restrict__ dlow, int h, int w) {
    const int plane = blockIdx.z, i = blockIdx.y;
    const int j = blockIdx.x * ST_THREADS + threadIdx.x;
    if (j >= w) return;
    const T* g = dhi + (size_t)plane * 16 * h * w;
    float acc = 0.f;
    // a low pixel is touched by output rows 4i-2 .. 4i+5 (more at the clamped borders: rows 0, 1 belong to row 0 only,
    // the last two rows to row h-1 twice) -- wgt() carries the exact ATen weights including both clamps
    for (int d = max(4 * i - 2, 0); d <= min(4 * i + 5, 4 * h - 1); ++d) {
        const float wh = wgt(d, i, h);
        if (wh == 0.f) continue;
        for (int e = max(4 * j - 2, 0); e <= min(4 * j + 5, 4 * w - 1); ++e)
            acc = fmaf(wh * wgt(e, j, w), to_f32(g[(size_t)d * 4 * w + e]), acc);
    }
    dlow[((size_t)plane * h + i) * w + j] = from_f32<T>(acc);
}

// ------------------------------------------------------------------------------------------ loss from low-resolution logits
template <typename T>
__global__ __launch_bounds__(ST_THREADS) void stats_up4_kernel(const T* __restrict__ low, const float* __restrict__ targets,
                                                                float* __restrict__ stats, int M, int h, int w,
                                                                float alpha, float gamma, float thr, const saicv::DetSink det) {
    const int bm = blockIdx.z, b = bm / M, q = blockIdx.y;
    const int j = blockIdx.x * ST_THREADS + threadIdx.x;
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (j < w) {
        float v[3][3];
        load3x3(low + (size_t)bm * h * w, h, w, q, j, v);
        const float* tg = targets + (size_t)b * 16 * h * w;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const f32x4 tv = *reinterpret_cast<const f32x4*>(tg + (size_t)(4 * q + r) * 4 * w + 4 * j);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float x = interp(v, q, j, r, s, h, w), t = tv[s];
                const MaskTerm m = mask_term(x, t, alpha);
                acc[0] += m.af * pow_gamma(m.one_m_pt, gamma) * m.bce;
                acc[1] += m.p * t;
                acc[2] += m.p;
                acc[3] += t;
                const bool pi = x > thr, ti = t > thr;
                acc[4] += (pi && ti) ? 1.f : 0.f;
                acc[5] += (pi || ti) ? 1.f : 0.f;
            }
        }
    }
    __shared__ float red[ST_THREADS / 64][6];
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[i] = wave_sum(acc[i]);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
#pragma unroll
        for (int i = 0; i < 6; ++i) red[wave][i] = acc[i];
    __syncthreads();
    if (threadIdx.x < 6) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < ST_THREADS / 64; ++k) s += red[k][threadIdx.x];
        // (column block, low-resolution row) = partial
        saicv::det_add(det, &stats[(size_t)bm * 6 + threadIdx.x], (size_t)bm * 6 + threadIdx.x, blockIdx.y * gridDim.x + blockIdx.x, s);
    }
}

// d loss / d low[b, m, i, j]: the 8 x 8 output pixels around (4i, 4j) are re-interpolated from the 5 x 5 low-resolution
// neighbourhood (L1 / L2 resident), their loss gradient evaluated and weighted -- no full-resolution gradient tensor
template <typename T>
__global__ __launch_bounds__(ST_THREADS) void grad_up4_kernel(const T* __restrict__ low, const float* __restrict__ targets,
                                                               const float* __restrict__ coef, T* __restrict__ dlow, int M,
                                                               int h, int w, float alpha, float gamma) {
    const int bm = blockIdx.z, b = bm / M, i = blockIdx.y;
    const int j = blockIdx.x * ST_THREADS + threadIdx.x;
    if (j >= w) return;
    const float c0 = coef[bm * 3], c1 = coef[bm * 3 + 1], c2 = coef[bm * 3 + 2];
    const T* lp = low + (size_t)bm * h * w;
    const float* tg = targets + (size_t)b * 16 * h * w;
    float acc = 0.f;
    for (int d = max(4 * i - 2, 0); d <= min(4 * i + 5, 4 * h - 1); ++d) {
        const Tap th = tap4(d, h);
        const float wh = (th.i0 == i ? th.l0 : 0.f) + (th.i1 == i ? th.l1 : 0.f);
        if (wh == 0.f) continue;
        const T* r0 = lp + (size_t)th.i0 * w;
        const T* r1 = lp + (size_t)th.i1 * w;
        for (int e = max(4 * j - 2, 0); e <= min(4 * j + 5, 4 * w - 1); ++e) {
            const Tap tw = tap4(e, w);
            const float ww = (tw.i0 == j ? tw.l0 : 0.f) + (tw.i1 == j ? tw.l1 : 0.f);
            const float x = th.l0 * (tw.l0 * to_f32(r0[tw.i0]) + tw.l1 * to_f32(r0[tw.i1])) +
                            th.l1 * (tw.l0 * to_f32(r1[tw.i0]) + tw.l1 * to_f32(r1[tw.i1]));
            const float t = tg[(size_t)d * 4 * w + e];
            acc = fmaf(wh * ww, loss_grad(x, t, alpha, gamma, c0, c1, c2), acc);
        }
    }
    dlow[((size_t)bm * h + i) * w + j] = from_f32<T>(acc);
}

// The same gradient, tiled: a block owns 16 x 16 low-resolution pixels.  The one-thread-per-low-pixel kernel above evaluates
// the loss gradient of every output pixel in each of the (up to four) low pixels it feeds AND once more per row / column of
// the 8 x 8 window that only carries a zero weight -- 64 evaluations per low pixel, 4 per output pixel, each ~60 vector
// operations with three transcendentals (1.7 ms per call at 20 x 4 x 256 x 256).  Here the 72 x 72 output pixels that touch the
// tile are evaluated ONCE into LDS (1.27 per output pixel), from an 18 x 18 LDS copy of the low-resolution neighbourhood, and
// each thread then gathers its 8 x 8 window with the bilinear weights.  Same formulas, same summation order.
constexpr int GU_T = 16;                  // low-resolution tile edge
constexpr int GU_H = 4 * GU_T + 8;        // output pixels touching the tile, per axis (72)
template <typename T>
__global__ __launch_bounds__(GU_T * GU_T) void grad_up4_tiled_kernel(const T* __restrict__ low, const float* __restrict__ targets,
                                                                     const float* __restrict__ coef, T* __restrict__ dlow, int M,
                                                                     int h, int w, float alpha, float gamma) {
    __shared__ float lowt[GU_T + 2][GU_T + 2];
    __shared__ float gt[GU_H][GU_H + 1];
    const int bm = blockIdx.z, b = bm / M;
    const int i0 = blockIdx.y * GU_T, j0 = blockIdx.x * GU_T;
    const float c0 = coef[bm * 3], c1 = coef[bm * 3 + 1], c2 = coef[bm * 3 + 2];
    const T* lp = low + (size_t)bm * h * w;
    const float* tg = targets + (size_t)b * 16 * h * w;
    for (int q = threadIdx.x; q < (GU_T + 2) * (GU_T + 2); q += GU_T * GU_T) {
        const int a = q / (GU_T + 2), c = q - a * (GU_T + 2);
        const int r = min(max(i0 - 1 + a, 0), h - 1), cc = min(max(j0 - 1 + c, 0), w - 1);
        lowt[a][c] = to_f32(lp[(size_t)r * w + cc]);
    }
    __syncthreads();
    // output pixels (d, e) with d in [4 i0 - 2, 4 i0 + 4 GU_T + 5] (clipped to the image): their gradient, once
    const int D0 = 4 * i0 - 2, E0 = 4 * j0 - 2;
    for (int q = threadIdx.x; q < GU_H * GU_H; q += GU_T * GU_T) {
        const int dd = q / GU_H, ee = q - dd * GU_H;
        const int d = D0 + dd, e = E0 + ee;
        float g = 0.f;
        if (d >= 0 && d < 4 * h && e >= 0 && e < 4 * w) {
            const Tap th = tap4(d, h), tw = tap4(e, w);
            const int a0 = th.i0 - (i0 - 1), a1 = th.i1 - (i0 - 1), b0 = tw.i0 - (j0 - 1), b1 = tw.i1 - (j0 - 1);
            const float x = th.l0 * (tw.l0 * lowt[a0][b0] + tw.l1 * lowt[a0][b1]) + th.l1 * (tw.l0 * lowt[a1][b0] + tw.l1 * lowt[a1][b1]);
            g = loss_grad(x, tg[(size_t)d * 4 * w + e], alpha, gamma, c0, c1, c2);
        }
        gt[dd][ee] = g;
    }
    __syncthreads();
    const int ti = threadIdx.x / GU_T, tj = threadIdx.x - ti * GU_T;
    const int i = i0 + ti, j = j0 + tj;
    if (i >= h || j >= w) return;
    float acc = 0.f;
    for (int d = max(4 * i - 2, 0); d <= min(4 * i + 5, 4 * h - 1); ++d) {
        const float wh = wgt(d, i, h);
        if (wh == 0.f) continue;
        for (int e = max(4 * j - 2, 0); e <= min(4 * j + 5, 4 * w - 1); ++e)
            acc = fmaf(wh * wgt(e, j, w), gt[d - D0][e - E0], acc);
    }
    dlow[((size_t)bm * h + i) * w + j] = from_f32<T>(acc);
}

}  // namespace

namespace saicv {

int hyper_product_fwd(int dtype, const void* x, const void* hyper, void* out, int B, int Tm, int P, int C, hipStream_t st) {
    SAICV_REQUIRE(C == 32, "hyper_product: C=%d (the mask decoder's upscaled embedding has 32 channels)", C);
    SAICV_REQUIRE(B > 0 && P > 0 && Tm >= 1 && Tm <= 8, "hyper_product: bad sizes B=%d P=%d T=%d", B, P, Tm);
    dim3 grid((P + ST_THREADS - 1) / ST_THREADS, B);
    if (dtype == SAICV_DTYPE_BF16)
        hipLaunchKernelGGL((hyper_fwd_kernel<bf16_t, 32>), grid, dim3(ST_THREADS), 0, st, (const bf16_t*)x, (const bf16_t*)hyper, (bf16_t*)out, Tm, P);
    else
        hipLaunchKernelGGL((hyper_fwd_kernel<float, 32>), grid, dim3(ST_THREADS), 0, st, (const float*)x, (const float*)hyper, (float*)out, Tm, P);
    return check_launch("hyper_product_fwd");
}

int hyper_product_bwd(int dtype, const void* x, const void* hyper, const void* dout, void* dx, float* dhyper, int B, int Tm,
                      int P, int C, hipStream_t st) {
    SAICV_REQUIRE(C == 32, "hyper_product: C=%d (the mask decoder's upscaled embedding has 32 channels)", C);
    SAICV_REQUIRE(B > 0 && P > 0 && Tm >= 1 && Tm <= 8, "hyper_product: bad sizes B=%d P=%d T=%d", B, P, Tm);
    hipMemsetAsync(dhyper, 0, (size_t)B * Tm * C * sizeof(float), st);
    dim3 grid((P + ST_THREADS - 1) / ST_THREADS, B);
    DetParts det;
    if (det.begin(st, (int)grid.x, (size_t)B * Tm * C, "hyper_product_bwd")) return -1;
    if (dtype == SAICV_DTYPE_BF16)
        hipLaunchKernelGGL((hyper_bwd_kernel<bf16_t, 32>), grid, dim3(ST_THREADS), 0, st, (const bf16_t*)x, (const bf16_t*)hyper, (const bf16_t*)dout, (bf16_t*)dx, dhyper, Tm, P, det.sink());
    else
        hipLaunchKernelGGL((hyper_bwd_kernel<float, 32>), grid, dim3(ST_THREADS), 0, st, (const float*)x, (const float*)hyper, (const float*)dout, (float*)dx, dhyper, Tm, P, det.sink());
    if (check_launch("hyper_product_bwd")) return -2;
    return det.fold(dhyper, 0, (size_t)B * Tm * C);
}

#define UP4_CHECK(who)                                                                                                   \
    SAICV_REQUIRE(planes > 0 && h > 0 && w > 0 && h <= 65535 && planes <= 65535, who ": bad sizes planes=%d h=%d w=%d", planes, h, w)

int upsample4_fwd(int dtype, const void* low, void* out, int planes, int h, int w, hipStream_t st) {
    UP4_CHECK("upsample4_fwd");
    dim3 grid((w + ST_THREADS - 1) / ST_THREADS, h, planes);
    if (dtype == SAICV_DTYPE_BF16) hipLaunchKernelGGL(up4_fwd_kernel<bf16_t>, grid, dim3(ST_THREADS), 0, st, (const bf16_t*)low, (bf16_t*)out, h, w);
    else hipLaunchKernelGGL(up4_fwd_kernel<float>, grid, dim3(ST_THREADS), 0, st, (const float*)low, (float*)out, h, w);
    return check_launch("upsample4_fwd");
}

int upsample4_bwd(int dtype, const void* dhi, void* dlow, int planes, int h, int w, hipStream_t st) {
    UP4_CHECK("upsample4_bwd");
    dim3 grid((w + ST_THREADS - 1) / ST_THREADS, h, planes);
    if (dtype == SAICV_DTYPE_BF16) hipLaunchKernelGGL(up4_bwd_kernel<bf16_t>, grid, dim3(ST_THREADS), 0, st, (const bf16_t*)dhi, (bf16_t*)dlow, h, w);
    else hipLaunchKernelGGL(up4_bwd_kernel<float>, grid, dim3(ST_THREADS), 0, st, (const float*)dhi, (float*)dlow, h, w);
    return check_launch("upsample4_bwd");
}

int mask_loss_stats_up4(int dtype, const void* low, const float* targets, float* stats, int B, int M, int h, int w,
                        double alpha, double gamma, double thr, hipStream_t st) {
    const int planes = B * M;
    UP4_CHECK("mask_loss_stats_up4");
    hipMemsetAsync(stats, 0, (size_t)planes * 6 * sizeof(float), st);
    dim3 grid((w + ST_THREADS - 1) / ST_THREADS, h, planes);
    DetParts det;
    if (det.begin(st, (int)(grid.x * grid.y), (size_t)planes * 6, "mask_loss_stats_up4")) return -1;
    if (dtype == SAICV_DTYPE_BF16)
        hipLaunchKernelGGL(stats_up4_kernel<bf16_t>, grid, dim3(ST_THREADS), 0, st, (const bf16_t*)low, targets, stats, M, h, w, (float)alpha, (float)gamma, (float)thr, det.sink());
    else
        hipLaunchKernelGGL(stats_up4_kernel<float>, grid, dim3(ST_THREADS), 0, st, (const float*)low, targets, stats, M, h, w, (float)alpha, (float)gamma, (float)thr, det.sink());
    if (check_launch("mask_loss_stats_up4")) return -2;
    return det.fold(stats, 0, (size_t)planes * 6);
}

int mask_loss_grad_up4(int dtype, const void* low, const float* targets, const float* coef, void* dlow, int B, int M, int h,
                       int w, double alpha, double gamma, hipStream_t st) {
    const int planes = B * M;
    UP4_CHECK("mask_loss_grad_up4");
    static const int tiled = getenv("SAICV_GRAD_UP4_TILED") ? atoi(getenv("SAICV_GRAD_UP4_TILED")) : 1;
    if (tiled) {
        dim3 tgrid((w + GU_T - 1) / GU_T, (h + GU_T - 1) / GU_T, planes);
        if (dtype == SAICV_DTYPE_BF16)
            hipLaunchKernelGGL(grad_up4_tiled_kernel<bf16_t>, tgrid, dim3(GU_T * GU_T), 0, st, (const bf16_t*)low, targets, coef, (bf16_t*)dlow, M, h, w, (float)alpha, (float)gamma);
        else
            hipLaunchKernelGGL(grad_up4_tiled_kernel<float>, tgrid, dim3(GU_T * GU_T), 0, st, (const float*)low, targets, coef, (float*)dlow, M, h, w, (float)alpha, (float)gamma);
        return check_launch("mask_loss_grad_up4");
    }
    dim3 grid((w + ST_THREADS - 1) / ST_THREADS, h, planes);
    if (dtype == SAICV_DTYPE_BF16)
        hipLaunchKernelGGL(grad_up4_kernel<bf16_t>, grid, dim3(ST_THREADS), 0, st, (const bf16_t*)low, targets, coef, (bf16_t*)dlow, M, h, w, (float)alpha, (float)gamma);
    else
        hipLaunchKernelGGL(grad_up4_kernel<float>, grid, dim3(ST_THREADS), 0, st, (const float*)low, targets, coef, (float*)dlow, M, h, w, (float)alpha, (float)gamma);
    return check_launch("mask_loss_grad_up4");
}

}  // namespace saicv
