// Depthwise convolution (groups == channels), NHWC, gfx950 -- SURVEY.md section 8(f) rank 2: the one kernel family the
// remaining reference backbones need beyond the blocks of the hot path.
// Replaces nn.Conv2d(C, C, k, stride, padding, dilation, groups=C) (+ backward) of
//   reference SimpleAICV/classification/backbones/van.py:30,68,75 (3x3 / 5x5 / dilated 7x7 depthwise of the LKA block),
//   reference SimpleAICV/classification/backbones/convformer.py (7x7 depthwise of the SepConv token mixer).
// One multiply-add per (output element, tap): 2 k^2 flop against 4 bytes of activation traffic per element -- HBM-bound for
// every kernel size the reference uses (k <= 7: 24.5 flop/byte against a ridge of 312), so these are streaming kernels:
// 16-byte channel chunks per lane, fp32 accumulation, no LDS, no matrix cores (do NOT reshape this into a GEMM: a depthwise
// convolution as an im2col GEMM multiplies by a block-diagonal matrix that is 1/C dense).
//   forward / data gradient: gather form, one thread per (pixel, channel chunk), TW consecutive pixels of a row per thread so
//       that the k weight chunks of a kernel row are loaded once per TW outputs; taps outside the image are skipped
//       (zero padding), stride and dilation are general (the data gradient of a stride > 1 layer skips the taps whose
//       source coordinate is not a multiple of the stride);
//   stride 1 / dilation 1 (k = 3, 5, 7) forward and data gradient: a thread walks the TW + k - 1 source columns of its eight
//       outputs once per kernel row (each chunk loaded and unpacked once);
//   weight gradient: dW[tap][c] = sum over pixels of dy[p][c] * x[p @ tap][c]; a thread owns (channel chunk, kernel row, pixel
//       sub-range), a block leaves its partial sums with fp32 atomics (also the bias gradient).
// Measured (scripts/dwconv_bench.py, profiles/r03_dwconv_bench.jsonl): these kernels are VALU-bound, not HBM-bound, on this
// machine -- a bf16 chunk costs one unpack and one FMA per element and tap on the vector ALU (5 GFLOP + as many unpacks for a
// 7 x 7 layer that moves 0.1 GB): 0.5-1.9 TB/s of activation traffic forward.
// Weights travel TAP-MAJOR: Wt[k*k][C] in the compute dtype (forward / data gradient), dWt[k*k][C] fp32 (weight gradient);
// the Python wrapper converts from / to PyTorch's [C, 1, k, k].
#include "common.h"
#include "saicv_internal.h"
#include "det.h"
#include "../../include/saicv_hip.h"

namespace {

constexpr int DW_TW = 4;            // output pixels of one row per thread

// forward (FLIP = false): y[n,oh,ow,c] = b[c] + sum_{r,s} x[n, oh*st - pad + r*dil, ow*st - pad + s*dil, c] * w[r*K+s][c]
// data gradient (FLIP = true, tensors swapped by the caller): dx[n,ih,iw,c] = sum_{r,s} dy[n, (ih + pad - r*dil)/st,
// (iw + pad - s*dil)/st, c] * w[r*K+s][c] over the taps whose quotients are exact and inside the dy image.
template <typename T, bool FLIP>
__global__ __launch_bounds__(256) void dwconv_gather_kernel(const T* __restrict__ src, const T* __restrict__ wt,
                                                            const float* __restrict__ bias, T* __restrict__ dst, int Nimg,
                                                            int SH, int SW, int DH, int DW_, int C, int K, int stride, int pad,
                                                            int dil) {
    constexpr int N = Chunk<T>::N;
    const int cpr = C / N;
    const int wt_tiles = (DW_ + DW_TW - 1) / DW_TW;
    const size_t total = (size_t)Nimg * DH * wt_tiles * cpr;
    const size_t gstride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gstride) {
        const int cb = (int)(i % cpr);
        size_t t = i / cpr;
        const int wtile = (int)(t % wt_tiles); t /= wt_tiles;
        const int dh = (int)(t % DH);
        const int n = (int)(t / DH);
        const int dw0 = wtile * DW_TW;
        float acc[DW_TW][N];
#pragma unroll
        for (int j = 0; j < DW_TW; ++j)
#pragma unroll
            for (int e = 0; e < N; ++e) acc[j][e] = 0.f;
        if (!FLIP && bias != nullptr) {
#pragma unroll
            for (int e = 0; e < N; ++e) {
                const float b = bias[cb * N + e];
#pragma unroll
                for (int j = 0; j < DW_TW; ++j) acc[j][e] = b;
            }
        }
        for (int r = 0; r < K; ++r) {
            int sh;
            if (!FLIP) {
                sh = dh * stride - pad + r * dil;
            } else {
                const int num = dh + pad - r * dil;
                if (num < 0 || num % stride) continue;
                sh = num / stride;
            }
            if ((unsigned)sh >= (unsigned)SH) continue;
            const T* srow = src + ((size_t)(n * SH + sh) * SW) * C + cb * N;
            for (int s = 0; s < K; ++s) {
                float w[N];
                Chunk<T>::unpack(ld_chunk(wt + (size_t)(r * K + s) * C + cb * N), w);
#pragma unroll
                for (int j = 0; j < DW_TW; ++j) {
                    const int dw = dw0 + j;
                    int sw;
                    bool ok = dw < DW_;
                    if (!FLIP) {
                        sw = dw * stride - pad + s * dil;
                    } else {
                        const int num = dw + pad - s * dil;
                        ok = ok && num >= 0 && (num % stride) == 0;
                        sw = num / stride;
                    }
                    if (ok && (unsigned)sw < (unsigned)SW) {
                        float v[N];
                        Chunk<T>::unpack(ld_chunk(srow + (size_t)sw * C), v);
#pragma unroll
                        for (int e = 0; e < N; ++e) acc[j][e] = fmaf(v[e], w[e], acc[j][e]);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < DW_TW; ++j) {
            const int dw = dw0 + j;
            if (dw < DW_) st_chunk(dst + ((size_t)(n * DH + dh) * DW_ + dw) * C + cb * N, Chunk<T>::pack(acc[j]));
        }
    }
}

// stride 1, dilation 1 (K = 3, 5, 7): a thread walks the TW + K - 1 source columns of its TW outputs ONCE per kernel row --
// each loaded chunk is unpacked once and feeds every output it overlaps -- instead of K loads and unpacks per output.
// FLIPW = data gradient: the same correlation with the kernel mirrored and pad' = K - 1 - pad.
template <typename T, int K, bool FLIPW>
__global__ __launch_bounds__(256) void dwconv_s1_kernel(const T* __restrict__ src, const T* __restrict__ wt, const float* __restrict__ bias,
                                                        T* __restrict__ dst, int Nimg, int SH, int SW, int DH, int DW_, int C, int pad) {
    constexpr int N = Chunk<T>::N;
    constexpr int TW = 8;
    const int cpr = C / N;
    const int wt_tiles = (DW_ + TW - 1) / TW;
    const size_t total = (size_t)Nimg * DH * wt_tiles * cpr;
    const size_t gstride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gstride) {
        const int cb = (int)(i % cpr);
        size_t t = i / cpr;
        const int wtile = (int)(t % wt_tiles); t /= wt_tiles;
        const int dh = (int)(t % DH);
        const int n = (int)(t / DH);
        const int dw0 = wtile * TW;
        float acc[TW][N];
#pragma unroll
        for (int j = 0; j < TW; ++j)
#pragma unroll
            for (int e = 0; e < N; ++e) acc[j][e] = (!FLIPW && bias != nullptr) ? bias[cb * N + e] : 0.f;
        for (int r = 0; r < K; ++r) {
            const int sh = dh - pad + r;
            if ((unsigned)sh >= (unsigned)SH) continue;
            float w[K][N];
#pragma unroll
            for (int s = 0; s < K; ++s) {
                const int tap = FLIPW ? (K - 1 - r) * K + (K - 1 - s) : r * K + s;
                Chunk<T>::unpack(ld_chunk(wt + (size_t)tap * C + cb * N), w[s]);
            }
            const T* srow = src + ((size_t)(n * SH + sh) * SW) * C + cb * N;
#pragma unroll
            for (int ci = 0; ci < TW + K - 1; ++ci) {
                const int sw = dw0 - pad + ci;
                if ((unsigned)sw < (unsigned)SW) {
                    float v[N];
                    Chunk<T>::unpack(ld_chunk(srow + (size_t)sw * C), v);
#pragma unroll
                    for (int s = 0; s < K; ++s) {
                        const int j = ci - s;           // output dw0 + j reads source column dw0 + j - pad + s
                        if (j >= 0 && j < TW) {
#pragma unroll
                            for (int e = 0; e < N; ++e) acc[j][e] = fmaf(v[e], w[s][e], acc[j][e]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < TW; ++j) {
            const int dw = dw0 + j;
            if (dw < DW_) st_chunk(dst + ((size_t)(n * DH + dh) * DW_ + dw) * C + cb * N, Chunk<T>::pack(acc[j]));
        }
    }
}

// weight gradient.  A thread owns (channel chunk, kernel ROW r, pixel sub-range) and keeps the K column sums of that row in
// registers: per output pixel one dy chunk and K source chunks of input row oh*stride - pad + r*dil.  Threads run over
// CONSECUTIVE channel chunks first (coalesced 128-byte segments), kernel rows second, sub-ranges third; blockIdx.x walks chunk
// groups, blockIdx.y pixel ranges.  The sub-ranges of a block are summed through LDS, so a launch issues (number of pixel
// ranges) atomics per weight -- the host keeps that at a few hundred: a depthwise weight tensor is tiny (k*k*C values), and
// thousands of blocks adding into the same 576 addresses of a 3 x 3 x 64 layer serialise (measured: 8 ms).
template <typename T, int K>
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(const T* __restrict__ dy, const T* __restrict__ x, float* __restrict__ dwt,
                                                           float* __restrict__ dbias, int Nimg, int H, int W, int OH, int OW, int C,
                                                           int stride, int pad, int dil, int pix_per_block, int chunks_per_block,
                                                           const saicv::DetSink det) {
    constexpr int N = Chunk<T>::N;
    constexpr int ACC = (K + 1) * N;                        // K column sums + the bias sum, N channels each
    extern __shared__ float red[];                          // [256][ACC]
    const int cpr = C / N;
    const int cl = threadIdx.x % chunks_per_block;
    const int rr = threadIdx.x / chunks_per_block;
    const int r = rr % K;
    const int ps = rr / K;
    const int lanes_per_ps = chunks_per_block * K;
    const int nps = 256 / lanes_per_ps;
    const int cb = blockIdx.x * chunks_per_block + cl;
    const bool live = ps < nps && cb < cpr;
    const size_t npix = (size_t)Nimg * OH * OW;
    const size_t sub = ((size_t)pix_per_block + nps - 1) / nps;
    const size_t b0 = (size_t)blockIdx.y * pix_per_block;
    const size_t b1 = b0 + pix_per_block < npix ? b0 + pix_per_block : npix;
    const size_t p0 = b0 + (size_t)ps * sub;
    const size_t p1 = p0 + sub < b1 ? p0 + sub : b1;
    float acc[K][N], bacc[N];
#pragma unroll
    for (int s = 0; s < K; ++s)
#pragma unroll
        for (int e = 0; e < N; ++e) acc[s][e] = 0.f;
#pragma unroll
    for (int e = 0; e < N; ++e) bacc[e] = 0.f;
    if (live && p0 < p1) {
        size_t p = p0;
        int ow = (int)(p % OW);
        size_t q = p / OW;
        int oh = (int)(q % OH);
        int n = (int)(q / OH);
        for (; p < p1; ++p) {
            float g[N];
            Chunk<T>::unpack(ld_chunk(dy + p * C + cb * N), g);
            const int ih = oh * stride - pad + r * dil;
            if ((unsigned)ih < (unsigned)H) {
                const T* xrow = x + ((size_t)(n * H + ih) * W) * C + cb * N;
#pragma unroll
                for (int s = 0; s < K; ++s) {
                    const int iw = ow * stride - pad + s * dil;
                    if ((unsigned)iw < (unsigned)W) {
                        float v[N];
                        Chunk<T>::unpack(ld_chunk(xrow + (size_t)iw * C), v);
#pragma unroll
                        for (int e = 0; e < N; ++e) acc[s][e] = fmaf(g[e], v[e], acc[s][e]);
                    }
                }
            }
            if (r == 0) {
#pragma unroll
                for (int e = 0; e < N; ++e) bacc[e] += g[e];
            }
            if (++ow == OW) { ow = 0; if (++oh == OH) { oh = 0; ++n; } }
        }
    }
    // sum the sub-ranges: thread (cl, r, ps) parks its sums, the ps == 0 threads add the others and issue the atomics
    float* mine = red + (size_t)threadIdx.x * ACC;
#pragma unroll
    for (int s = 0; s < K; ++s)
#pragma unroll
        for (int e = 0; e < N; ++e) mine[s * N + e] = acc[s][e];
#pragma unroll
    for (int e = 0; e < N; ++e) mine[K * N + e] = bacc[e];
    __syncthreads();
    if (live && ps == 0) {
        for (int o = 1; o < nps; ++o) {
            const float* other = red + (size_t)(threadIdx.x + o * lanes_per_ps) * ACC;
#pragma unroll
            for (int s = 0; s < K; ++s)
#pragma unroll
                for (int e = 0; e < N; ++e) acc[s][e] += other[s * N + e];
#pragma unroll
            for (int e = 0; e < N; ++e) bacc[e] += other[K * N + e];
        }
#pragma unroll
        for (int s = 0; s < K; ++s)
#pragma unroll
            for (int e = 0; e < N; ++e)      // pixel range blockIdx.y = partial blockIdx.y of [K*K][C] weights (+ [C] bias sums behind them)
                saicv::det_add(det, dwt + (size_t)(r * K + s) * C + cb * N + e, (size_t)(r * K + s) * C + cb * N + e, blockIdx.y, acc[s][e]);
        if (r == 0 && dbias != nullptr) {
#pragma unroll
            for (int e = 0; e < N; ++e) saicv::det_add(det, dbias + cb * N + e, (size_t)K * K * C + cb * N + e, blockIdx.y, bacc[e]);
        }
    }
}

int dw_grid(size_t items) {
    size_t b = (items + 255) / 256;
    if (b > 16384) b = 16384;
    if (b < 1) b = 1;
    return (int)b;
}

int dw_check(const char* who, int dtype, int C, int K, int stride, int pad, int dil) {
    const int n = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    SAICV_REQUIRE(dtype == SAICV_DTYPE_BF16 || dtype == SAICV_DTYPE_F32, "%s: dtype %d", who, dtype);
    SAICV_REQUIRE(C > 0 && C % n == 0, "%s: C=%d must be a multiple of %d", who, C, n);
    SAICV_REQUIRE(K >= 1 && K <= 8 && stride >= 1 && pad >= 0 && dil >= 1, "%s: kernel %d stride %d pad %d dilation %d", who, K, stride, pad, dil);
    return 0;
}

}  // namespace

using saicv::check_launch;

extern "C" {

// reference van.py:30,68,75 / convformer.py depthwise nn.Conv2d forward.  x [N,H,W,C], wt [K*K][C] (tap-major, compute
// dtype), bias [C] fp32 or NULL, y [N,OH,OW,C].
int saicv_dwconv2d_fwd(int dtype, const void* x, const void* wt, const float* bias, void* y, int N, int H, int W, int C, int OH,
                       int OW, int K, int stride, int pad, int dil, void* stream) {
    if (dw_check("dwconv2d_fwd", dtype, C, K, stride, pad, dil)) return -1;
    SAICV_REQUIRE(OH == (H + 2 * pad - dil * (K - 1) - 1) / stride + 1 && OW == (W + 2 * pad - dil * (K - 1) - 1) / stride + 1,
                  "dwconv2d_fwd: output %d x %d does not match the geometry", OH, OW);
    hipStream_t st = (hipStream_t)stream;
    const int n = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    const size_t items = (size_t)N * OH * ((OW + DW_TW - 1) / DW_TW) * (C / n);
    if (stride == 1 && dil == 1 && (K == 3 || K == 5 || K == 7)) {
        const size_t it8 = (size_t)N * OH * ((OW + 7) / 8) * (C / n);
#define DW_S1(TT, KK) hipLaunchKernelGGL((dwconv_s1_kernel<TT, KK, false>), dim3(dw_grid(it8)), dim3(256), 0, st, (const TT*)x, (const TT*)wt, bias, (TT*)y, N, H, W, OH, OW, C, pad)
        if (dtype == SAICV_DTYPE_BF16) { if (K == 3) DW_S1(bf16_t, 3); else if (K == 5) DW_S1(bf16_t, 5); else DW_S1(bf16_t, 7); }
        else { if (K == 3) DW_S1(float, 3); else if (K == 5) DW_S1(float, 5); else DW_S1(float, 7); }
#undef DW_S1
        return check_launch("dwconv2d_fwd");
    }
    if (dtype == SAICV_DTYPE_BF16)
        hipLaunchKernelGGL((dwconv_gather_kernel<bf16_t, false>), dim3(dw_grid(items)), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)wt,
                           bias, (bf16_t*)y, N, H, W, OH, OW, C, K, stride, pad, dil);
    else
        hipLaunchKernelGGL((dwconv_gather_kernel<float, false>), dim3(dw_grid(items)), dim3(256), 0, st, (const float*)x, (const float*)wt,
                           bias, (float*)y, N, H, W, OH, OW, C, K, stride, pad, dil);
    return check_launch("dwconv2d_fwd");
}

// its input gradient: dy [N,OH,OW,C] -> dx [N,H,W,C]
int saicv_dwconv2d_dgrad(int dtype, const void* dy, const void* wt, void* dx, int N, int H, int W, int C, int OH, int OW, int K,
                         int stride, int pad, int dil, void* stream) {
    if (dw_check("dwconv2d_dgrad", dtype, C, K, stride, pad, dil)) return -1;
    hipStream_t st = (hipStream_t)stream;
    const int n = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    const size_t items = (size_t)N * H * ((W + DW_TW - 1) / DW_TW) * (C / n);
    if (stride == 1 && dil == 1 && (K == 3 || K == 5 || K == 7) && K - 1 - pad >= 0) {
        // dx = dy correlated with the mirrored kernel at pad' = K - 1 - pad
        const size_t it8 = (size_t)N * H * ((W + 7) / 8) * (C / n);
#define DW_S1(TT, KK) hipLaunchKernelGGL((dwconv_s1_kernel<TT, KK, true>), dim3(dw_grid(it8)), dim3(256), 0, st, (const TT*)dy, (const TT*)wt, (const float*)nullptr, (TT*)dx, N, OH, OW, H, W, C, K - 1 - pad)
        if (dtype == SAICV_DTYPE_BF16) { if (K == 3) DW_S1(bf16_t, 3); else if (K == 5) DW_S1(bf16_t, 5); else DW_S1(bf16_t, 7); }
        else { if (K == 3) DW_S1(float, 3); else if (K == 5) DW_S1(float, 5); else DW_S1(float, 7); }
#undef DW_S1
        return check_launch("dwconv2d_dgrad");
    }
    if (dtype == SAICV_DTYPE_BF16)
        hipLaunchKernelGGL((dwconv_gather_kernel<bf16_t, true>), dim3(dw_grid(items)), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)wt,
                           (const float*)nullptr, (bf16_t*)dx, N, OH, OW, H, W, C, K, stride, pad, dil);
    else
        hipLaunchKernelGGL((dwconv_gather_kernel<float, true>), dim3(dw_grid(items)), dim3(256), 0, st, (const float*)dy, (const float*)wt,
                           (const float*)nullptr, (float*)dx, N, OH, OW, H, W, C, K, stride, pad, dil);
    return check_launch("dwconv2d_dgrad");
}

// its weight (and bias) gradient, ACCUMULATED into dwt [K*K][C] fp32 / dbias [C] fp32 (NULL: none) with atomics
int saicv_dwconv2d_wgrad(int dtype, const void* dy, const void* x, float* dwt, float* dbias, int N, int H, int W, int C, int OH,
                         int OW, int K, int stride, int pad, int dil, void* stream) {
    if (dw_check("dwconv2d_wgrad", dtype, C, K, stride, pad, dil)) return -1;
    hipStream_t st = (hipStream_t)stream;
    const int n = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    // lanes = chunks (fastest) x kernel rows: the largest power-of-two chunk count with chunks * K <= 256
    int cpb = 1;
    while (cpb * 2 * K <= 256 && cpb * 2 <= C / n) cpb *= 2;
    const int groups = (C / n + cpb - 1) / cpb;
    const size_t npix = (size_t)N * OH * OW;
    // pixel ranges = atomics per weight: at most 512, at least 256 pixels per block, ~4 blocks per CU when the layer allows
    size_t ranges = (1024 + groups - 1) / groups;
    if (ranges > 512) ranges = 512;
    if (ranges < 1) ranges = 1;
    size_t ppb = (npix + ranges - 1) / ranges;
    if (ppb < 256) ppb = 256;
    ranges = (npix + ppb - 1) / ppb;
    SAICV_REQUIRE(ranges <= 65535, "dwconv2d_wgrad: %zu pixel ranges", ranges);
    dim3 grid(groups, (unsigned)ranges);
    saicv::DetParts det;
    if (det.begin(st, (int)ranges, (size_t)K * K * C + (dbias ? C : 0), "dwconv2d_wgrad")) return -1;
#define DW_WG(TT, KK) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dwconv_wgrad_kernel<TT, KK>), hipFuncAttributeMaxDynamicSharedMemorySize, 256 * (KK + 1) * 8 * 4); \
    hipLaunchKernelGGL((dwconv_wgrad_kernel<TT, KK>), grid, dim3(256), 256 * (KK + 1) * n * sizeof(float), st, (const TT*)dy, (const TT*)x, dwt, dbias, N, H, W, OH, OW, C, stride, pad, dil, (int)ppb, cpb, det.sink())
#define DW_WGK(TT) switch (K) { case 1: { DW_WG(TT, 1); } break; case 2: { DW_WG(TT, 2); } break; case 3: { DW_WG(TT, 3); } break; case 4: { DW_WG(TT, 4); } break; \
                                 case 5: { DW_WG(TT, 5); } break; case 6: { DW_WG(TT, 6); } break; case 7: { DW_WG(TT, 7); } break; default: { DW_WG(TT, 8); } break; }
    if (dtype == SAICV_DTYPE_BF16) { DW_WGK(bf16_t) } else { DW_WGK(float) }
#undef DW_WGK
#undef DW_WG
    if (check_launch("dwconv2d_wgrad")) return -2;
    if (det.fold(dwt, 0, (size_t)K * K * C)) return -1;
    return dbias ? det.fold(dbias, (size_t)K * K * C, (size_t)C) : 0;
}

}  // extern "C"
