// Layout / dtype packing kernels (HBM-bound, tiny next to the activations).
//  - pack_input : NCHW-shaped image batch (any strides, fp32) -> dense NHWC [N,H,W,Cp] in the
//                 compute dtype, channels zero-padded to Cp (the 3-channel stem becomes C=8
//                 so every implicit-GEMM gather is a 16-byte chunk).  The reference collater
//                 hands over NHWC-strided fp32 (SimpleAICV/classification/common.py:645-665).
//  - pack_weight: fp32 master weight [O,I,R,S] (any strides) -> compute-dtype forward matrix
//                 Wf[O][R][S][Ip] and, optionally, the data-gradient matrix Wd[I][R][S][O].
//  - unpack_wgrad: fp32 dW[O][R][S][Ip] -> gradient tensor [O,I,R,S] with arbitrary strides.
#include "common.h"
#include "saicv_internal.h"
#include "det.h"
#include "../../include/saicv_hip.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void pack_input_kernel(const float* __restrict__ src, long sN,
                                                         long sC, long sH, long sW, T* __restrict__ dst,
                                                         int Nimg, int C, int H, int W, int Cp) {
    const size_t total = (size_t)Nimg * H * W;
    const size_t gstride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gstride) {
        size_t pix = i;
        const int w = (int)(pix % W); pix /= W;
        const int h = (int)(pix % H);
        const int n = (int)(pix / H);
        const float* s = src + (size_t)n * sN + (size_t)h * sH + (size_t)w * sW;
        T* d = dst + i * Cp;
        for (int c = 0; c < Cp; ++c) d[c] = from_f32<T>(c < C ? s[(size_t)c * sC] : 0.f);
    }
}

// One [32 o][32 i] tile of one tap (r, s) per block, transposed through LDS: the master weight is read once along its
// contiguous axis, Wf rows (i fastest) and Wd rows (o fastest) are both written 64 contiguous bytes per 32 threads.
// (The element-per-thread version spent four integer divisions per element and scattered 2-byte Wd writes: 18 us for
// a ViT fc1 matrix that is 4 us of traffic; it ran 50-54 times per step.)
template <typename T>
DEVINL void pack_weight_tile(float (*tile)[33], const float* __restrict__ w, long sO, long sI, long sR, long sS, int O, int I,
                             int R, int S, int Ip, int Op, T* __restrict__ wf, T* __restrict__ wd, int bi, int bo, int tap) {
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;        // 32 x 8
    const int i0 = bi * 32, o0 = bo * 32;
    const int r = tap / S, s = tap - r * S;
#pragma unroll
    for (int oo = ty; oo < 32; oo += 8) {
        const int o = o0 + oo, i = i0 + tx;
        const float v = (o < O && i < I) ? w[(size_t)o * sO + (size_t)i * sI + (size_t)r * sR + (size_t)s * sS] : 0.f;
        tile[oo][tx] = v;
        if (wf && o < Op && i < Ip) wf[(((size_t)o * R + r) * S + s) * Ip + i] = from_f32<T>(v);
    }
    __syncthreads();
    if (wd) {
#pragma unroll
        for (int ii = ty; ii < 32; ii += 8) {
            const int i = i0 + ii, o = o0 + tx;
            if (i < I && o < Op) wd[(((size_t)i * R + r) * S + s) * Op + o] = from_f32<T>(tile[tx][ii]);
        }
    }
}

// The same on a [64 o][64 i] tile with 16-byte reads and 8-byte (4 x bf16) / 16-byte (4 x f32) writes, for weights whose
// input-channel axis is contiguous with every row start 16-byte aligned (conv weights in channels_last, nn.Linear): the
// 2-byte scalar writes of the 32 x 32 form moved the 0.7 GB of a ViT-B repack at 0.7 TB/s (r02: 0.99 ms per step).
template <typename T>
DEVINL void pack_weight_tile64(float (*tile)[65], const float* __restrict__ w, long sO, long sR, long sS, int O, int I,
                               int R, int S, int Ip, int Op, T* __restrict__ wf, T* __restrict__ wd, int bi, int bo, int tap) {
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;        // 16 groups of four x 16 rows
    const int i0 = bi * 64, o0 = bo * 64;
    const int r = tap / S, s = tap - r * S;
    typedef __attribute__((ext_vector_type(4))) T vec4;
#pragma unroll
    for (int oo = ty; oo < 64; oo += 16) {
        const int o = o0 + oo, i = i0 + 4 * tx;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (o < O && i < I) v = *reinterpret_cast<const f32x4*>(w + (size_t)o * sO + i + (size_t)r * sR + (size_t)s * sS);
#pragma unroll
        for (int k = 0; k < 4; ++k) tile[oo][4 * tx + k] = v[k];
        if (wf && o < Op && i < Ip) {
            vec4 q;
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = from_f32<T>(v[k]);
            *reinterpret_cast<vec4*>(wf + (((size_t)o * R + r) * S + s) * Ip + i) = q;
        }
    }
    __syncthreads();
    if (wd) {
#pragma unroll
        for (int ii = ty; ii < 64; ii += 16) {
            const int i = i0 + ii, o = o0 + 4 * tx;
            if (i < I && o < Op) {
                vec4 q;
#pragma unroll
                for (int k = 0; k < 4; ++k) q[k] = from_f32<T>(tile[4 * tx + k][ii]);
                *reinterpret_cast<vec4*>(wd + (((size_t)i * R + r) * S + s) * Op + o) = q;
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w, long sO, long sI,
                                                          long sR, long sS, int O, int I, int R, int S,
                                                          int Ip, int Op, T* __restrict__ wf, T* __restrict__ wd) {
    __shared__ float tile[32][33];
    pack_weight_tile<T>(tile, w, sO, sI, sR, sS, O, I, R, S, Ip, Op, wf, wd, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Every weight of a model in ONE launch (the per-step repack after the optimizer: 54 launches of ~5 us for ResNet-50):
// a block finds its weight by bisection over the tile prefix, then handles one 32 x 32 tile of one tap as above.
template <typename T>
__global__ __launch_bounds__(256) void pack_weight_batched_kernel(const saicv_pack_desc* __restrict__ descs, int n) {
    __shared__ float tile64[64][65];
    float (*tile)[33] = reinterpret_cast<float (*)[33]>(&tile64[0][0]);
    const int t = blockIdx.x;
    int lo = 0, hi = n - 1;
    while (lo < hi) {                                   // last descriptor with tile_begin <= t (uniform per block)
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].tile_begin <= t) lo = mid; else hi = mid - 1;
    }
    const saicv_pack_desc d = descs[lo];
    int local = t - d.tile_begin;
    const int bi = local % d.tiles_i; local /= d.tiles_i;
    const int bo = local % d.tiles_o;
    const int tap = local / d.tiles_o;
    if (d.tile == 64)            // (uniform) descriptor built for 64 x 64 tiles: contiguous, aligned input-channel axis
        pack_weight_tile64<T>(tile64, d.w, d.sO, d.sR, d.sS, d.O, d.I, d.R, d.S, d.Ip, d.Op, static_cast<T*>(d.wf),
                              static_cast<T*>(d.wd), bi, bo, tap);
    else
        pack_weight_tile<T>(tile, d.w, d.sO, d.sI, d.sR, d.sS, d.O, d.I, d.R, d.S, d.Ip, d.Op, static_cast<T*>(d.wf),
                            static_cast<T*>(d.wd), bi, bo, tap);
}

__global__ __launch_bounds__(256) void unpack_wgrad_kernel(const float* __restrict__ dw, int O, int I,
                                                           int R, int S, int Ip, float* __restrict__ g,
                                                           long sO, long sI, long sR, long sS,
                                                           int accumulate) {
    const size_t total = (size_t)O * R * S * I;
    const size_t gstride = (size_t)gridDim.x * blockDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gstride) {
        size_t t = e;
        const int i = (int)(t % I); t /= I;
        const int s = (int)(t % S); t /= S;
        const int r = (int)(t % R);
        const int o = (int)(t / R);
        const float v = dw[(((size_t)o * R + r) * S + s) * Ip + i];
        float* d = g + (size_t)o * sO + (size_t)i * sI + (size_t)r * sR + (size_t)s * sS;
        *d = accumulate ? (*d + v) : v;
    }
}

// ---- stride-2 stem as a stride-1 convolution on a space-to-depth image -------------------------------------------------
// A K x K stride-2 convolution with padding p over x[H][W][C] equals a ceil(K/2) x ceil(K/2) stride-1, unpadded convolution
// over Q[u][v][(a, b, c)] = P[2u + a][2v + b][c], P = x shifted by p with a zero border, with the weights regrouped as
// W'[o][R][S][(a, b, c)] = W[o][2R + a][2S + b][c] (taps beyond K are zero).  For the 7 x 7 x 3 ResNet stem the GEMM's K
// dimension drops from 7*7*8 = 392 (3 channels padded to a 16-byte chunk) to 4*4*16 = 256, and the packed input shrinks
// from 8 to 4 bf16 per pixel.
template <typename T>
__global__ __launch_bounds__(256) void pack_input_s2d_kernel(const float* __restrict__ src, long sN, long sC, long sH,
                                                             long sW, T* __restrict__ dst, int Nimg, int C, int H, int W,
                                                             int pad, int Hq, int Wq, int Cq) {
    const size_t total = (size_t)Nimg * Hq * Wq * 4;              // one (pixel of Q, a, b) group of C channels per thread
    const size_t gstride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gstride) {
        size_t t = i;
        const int ab = (int)(t & 3); t >>= 2;
        const int v = (int)(t % Wq); t /= Wq;
        const int u = (int)(t % Hq);
        const int n = (int)(t / Hq);
        const int h = 2 * u + (ab >> 1) - pad, w = 2 * v + (ab & 1) - pad;
        const bool in = (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
        const float* sp = src + (size_t)n * sN + (size_t)(in ? h : 0) * sH + (size_t)(in ? w : 0) * sW;
        T* d = dst + (((size_t)n * Hq + u) * Wq + v) * Cq + ab * C;
        for (int c = 0; c < C; ++c) d[c] = from_f32<T>(in ? sp[(size_t)c * sC] : 0.f);
        if (ab == 3)
            for (int c = 4 * C; c < Cq; ++c) dst[(((size_t)n * Hq + u) * Wq + v) * Cq + c] = from_f32<T>(0.f);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void pack_weight_s2d_kernel(const float* __restrict__ w, long sO, long sI, long sR,
                                                              long sS, int O, int I, int R, int S, int R2, int S2, int Cq,
                                                              T* __restrict__ wf) {
    const size_t total = (size_t)O * R2 * S2 * Cq;
    const size_t gstride = (size_t)gridDim.x * blockDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gstride) {
        size_t t = e;
        const int q = (int)(t % Cq); t /= Cq;
        const int s2 = (int)(t % S2); t /= S2;
        const int r2 = (int)(t % R2);
        const int o = (int)(t / R2);
        float v = 0.f;
        if (q < 4 * I) {
            const int ab = q / I, c = q - ab * I;
            const int r = 2 * r2 + (ab >> 1), sx = 2 * s2 + (ab & 1);
            if (r < R && sx < S) v = w[(size_t)o * sO + (size_t)c * sI + (size_t)r * sR + (size_t)sx * sS];
        }
        wf[e] = from_f32<T>(v);
    }
}

__global__ __launch_bounds__(256) void unpack_wgrad_s2d_kernel(const float* __restrict__ dw, int O, int I, int R, int S,
                                                               int R2, int S2, int Cq, float* __restrict__ g, long sO,
                                                               long sI, long sR, long sS, int accumulate) {
    const size_t total = (size_t)O * R * S * I;
    const size_t gstride = (size_t)gridDim.x * blockDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gstride) {
        size_t t = e;
        const int i = (int)(t % I); t /= I;
        const int s = (int)(t % S); t /= S;
        const int r = (int)(t % R);
        const int o = (int)(t / R);
        const int ab = ((r & 1) << 1) | (s & 1);
        const float v = dw[(((size_t)o * R2 + (r >> 1)) * S2 + (s >> 1)) * Cq + ab * I + i];
        float* d = g + (size_t)o * sO + (size_t)i * sI + (size_t)r * sR + (size_t)s * sS;
        *d = accumulate ? (*d + v) : v;
    }
}

// column sums of a [M][N] matrix (bias gradient), T in, fp32 out (accumulating atomics).
// A workgroup streams a slab of rows with 16-byte loads: a thread owns one chunk column and
// strides over rows; the row lanes are combined through LDS; one atomic per column per slab.
// r04: 1024-thread workgroups and at most 128 slabs.  The 256-thread form used up to 2048 slabs -- 2048 atomic adds on each of
// the N output addresses, which serialise in L2 (~190 ns each): the 77 MB patch-embedding bias gradient of ViT-B took 395 us in
// the model (rocprofv3 kernel stats) against 19 us of HBM time.  128 slabs of 1024 threads keep ~8 MB of loads in flight (four
// 16-byte loads per thread) and leave 128 adds per address.
constexpr int CS_THREADS = 1024;
template <typename T>
__global__ __launch_bounds__(CS_THREADS) void colsum_kernel(const T* __restrict__ x, int M, int N,
                                                            int rows_per, float* __restrict__ out, const saicv::DetSink det) {
    constexpr int E = Chunk<T>::N;
    const int cpr = N / E;
    const int cols = cpr < CS_THREADS ? cpr : CS_THREADS;
    const int rpp = CS_THREADS / cols;
    const int tx = threadIdx.x % cols, ty = threadIdx.x / cols;
    const int r0 = blockIdx.x * rows_per;
    const int r1 = min(M, r0 + rows_per);
    __shared__ float red[CS_THREADS * E];
    for (int base = 0; base < cpr; base += cols) {
        const int cb = base + tx;
        const bool active = (cb < cpr) && (ty < rpp);
        float acc[E];
#pragma unroll
        for (int k = 0; k < E; ++k) acc[k] = 0.f;
        if (active) {
            // four independent 16-byte loads in flight per thread (one at a time left the 77 MB bias-gradient pass of the ViT patch
            // embedding latency-bound: 207 us against 15 us of HBM time)
            int r = r0 + ty;
            for (; r + 3 * rpp < r1; r += 4 * rpp) {
                u32x4 c[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) c[u] = ld_chunk(x + (size_t)(r + u * rpp) * N + cb * E);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float v[E];
                    Chunk<T>::unpack(c[u], v);
#pragma unroll
                    for (int k = 0; k < E; ++k) acc[k] += v[k];
                }
            }
            for (; r < r1; r += rpp) {
                float v[E];
                Chunk<T>::unpack(ld_chunk(x + (size_t)r * N + cb * E), v);
#pragma unroll
                for (int k = 0; k < E; ++k) acc[k] += v[k];
            }
        }
#pragma unroll
        for (int k = 0; k < E; ++k) red[threadIdx.x * E + k] = acc[k];
        __syncthreads();
        if (active && ty == 0) {
#pragma unroll
            for (int k = 0; k < E; ++k) {
                float sgm = 0.f;
                for (int t = 0; t < rpp; ++t) sgm += red[(t * cols + tx) * E + k];
                saicv::det_add(det, out + cb * E + k, (size_t)cb * E + k, blockIdx.x, sgm);      // slab blockIdx.x = partial blockIdx.x
            }
        }
        __syncthreads();
    }
}

// out[r][:] = x[r][:] * scale[r / rows_per_scale]   (drop-path backward, per-sample factors)
template <typename T>
__global__ __launch_bounds__(256) void row_scale_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                        T* __restrict__ out, size_t nchunks, int chunks_per_group) {
    constexpr int N = Chunk<T>::N;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += stride) {
        const float s = scale[i / chunks_per_group];
        float v[N];
        Chunk<T>::unpack(ld_chunk(x + i * N), v);
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] *= s;
        st_chunk(out + i * N, Chunk<T>::pack(v));
    }
}

inline int sgrid(size_t total) {
    size_t b = (total + 255) / 256;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

namespace saicv {

int pack_input(int dtype, const float* src, long sN, long sC, long sH, long sW, void* dst, int Nimg,
               int C, int H, int W, int Cp, hipStream_t st) {
    SAICV_REQUIRE(Cp >= C, "pack_input: Cp=%d < C=%d", Cp, C);
    const size_t total = (size_t)Nimg * H * W;
    if (dtype == SAICV_DTYPE_BF16)
        hipLaunchKernelGGL(pack_input_kernel<bf16_t>, dim3(sgrid(total)), dim3(256), 0, st, src, sN, sC, sH, sW, (bf16_t*)dst, Nimg, C, H, W, Cp);
    else
        hipLaunchKernelGGL(pack_input_kernel<float>, dim3(sgrid(total)), dim3(256), 0, st, src, sN, sC, sH, sW, (float*)dst, Nimg, C, H, W, Cp);
    return check_launch("pack_input");
}

int pack_weight(int dtype, const float* w, long sO, long sI, long sR, long sS, int O, int I, int R,
                int S, int Ip, int Op, void* wf, void* wd, hipStream_t st) {
    SAICV_REQUIRE(Ip >= I, "pack_weight: Ip=%d < I=%d", Ip, I);
    SAICV_REQUIRE(wd == nullptr || Ip == I, "pack_weight: data-gradient matrix needs unpadded I");
    SAICV_REQUIRE(Op >= O, "pack_weight: Op=%d < O=%d", Op, O);
    SAICV_REQUIRE(R * S <= 65535 && (Op + 31) / 32 <= 65535, "pack_weight: tap count / output channels beyond the launch grid");
    const dim3 grid((Ip + 31) / 32, (Op + 31) / 32, R * S);     // rows o in [O, Op) and columns i in [I, Ip) are written as zeros
    if (dtype == SAICV_DTYPE_BF16)
        hipLaunchKernelGGL(pack_weight_kernel<bf16_t>, grid, dim3(256), 0, st, w, sO, sI, sR, sS, O, I, R, S, Ip, Op, (bf16_t*)wf, (bf16_t*)wd);
    else
        hipLaunchKernelGGL(pack_weight_kernel<float>, grid, dim3(256), 0, st, w, sO, sI, sR, sS, O, I, R, S, Ip, Op, (float*)wf, (float*)wd);
    return check_launch("pack_weight");
}

int pack_weight_batched(int dtype, const saicv_pack_desc* descs, int n, int total_tiles, hipStream_t st) {
    SAICV_REQUIRE(descs != nullptr && n > 0 && total_tiles > 0, "pack_weight_batched: empty descriptor table");
    if (dtype == SAICV_DTYPE_BF16)
        hipLaunchKernelGGL(pack_weight_batched_kernel<bf16_t>, dim3(total_tiles), dim3(256), 0, st, descs, n);
    else
        hipLaunchKernelGGL(pack_weight_batched_kernel<float>, dim3(total_tiles), dim3(256), 0, st, descs, n);
    return check_launch("pack_weight_batched");
}

int unpack_wgrad(const float* dw, int O, int I, int R, int S, int Ip, float* g, long sO, long sI,
                 long sR, long sS, int accumulate, hipStream_t st) {
    const size_t total = (size_t)O * R * S * I;
    hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(sgrid(total)), dim3(256), 0, st, dw, O, I, R, S, Ip, g, sO, sI, sR, sS, accumulate);
    return check_launch("unpack_wgrad");
}

int pack_input_s2d(int dtype, const float* src, long sN, long sC, long sH, long sW, void* dst, int Nimg, int C, int H,
                   int W, int pad, int Cq, hipStream_t st) {
    SAICV_REQUIRE(Cq >= 4 * C && pad >= 0, "pack_input_s2d: Cq=%d < 4*C=%d", Cq, 4 * C);
    const int Hq = (H + 2 * pad + 1) / 2, Wq = (W + 2 * pad + 1) / 2;
    const size_t total = (size_t)Nimg * Hq * Wq * 4;
    if (dtype == SAICV_DTYPE_BF16)
        hipLaunchKernelGGL(pack_input_s2d_kernel<bf16_t>, dim3(sgrid(total)), dim3(256), 0, st, src, sN, sC, sH, sW, (bf16_t*)dst, Nimg, C, H, W, pad, Hq, Wq, Cq);
    else
        hipLaunchKernelGGL(pack_input_s2d_kernel<float>, dim3(sgrid(total)), dim3(256), 0, st, src, sN, sC, sH, sW, (float*)dst, Nimg, C, H, W, pad, Hq, Wq, Cq);
    return check_launch("pack_input_s2d");
}

int pack_weight_s2d(int dtype, const float* w, long sO, long sI, long sR, long sS, int O, int I, int R, int S, int Cq,
                    void* wf, hipStream_t st) {
    SAICV_REQUIRE(Cq >= 4 * I, "pack_weight_s2d: Cq=%d < 4*I=%d", Cq, 4 * I);
    const int R2 = (R + 1) / 2, S2 = (S + 1) / 2;
    const size_t total = (size_t)O * R2 * S2 * Cq;
    if (dtype == SAICV_DTYPE_BF16)
        hipLaunchKernelGGL(pack_weight_s2d_kernel<bf16_t>, dim3(sgrid(total)), dim3(256), 0, st, w, sO, sI, sR, sS, O, I, R, S, R2, S2, Cq, (bf16_t*)wf);
    else
        hipLaunchKernelGGL(pack_weight_s2d_kernel<float>, dim3(sgrid(total)), dim3(256), 0, st, w, sO, sI, sR, sS, O, I, R, S, R2, S2, Cq, (float*)wf);
    return check_launch("pack_weight_s2d");
}

int unpack_wgrad_s2d(const float* dw, int O, int I, int R, int S, int Cq, float* g, long sO, long sI, long sR, long sS,
                     int accumulate, hipStream_t st) {
    const size_t total = (size_t)O * R * S * I;
    hipLaunchKernelGGL(unpack_wgrad_s2d_kernel, dim3(sgrid(total)), dim3(256), 0, st, dw, O, I, R, S, (R + 1) / 2, (S + 1) / 2, Cq, g,
                       sO, sI, sR, sS, accumulate);
    return check_launch("unpack_wgrad_s2d");
}

int row_scale(int dtype, const void* x, const float* scale, void* out, size_t rows, int row_len,
              int rows_per_scale, hipStream_t st) {
    const int n = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    SAICV_REQUIRE(row_len % n == 0 && rows_per_scale >= 1, "row_scale: row length %d must be a multiple of %d", row_len, n);
    const size_t nchunks = rows * (size_t)(row_len / n);
    const int cpg = rows_per_scale * (row_len / n);
    if (dtype == SAICV_DTYPE_BF16)
        hipLaunchKernelGGL(row_scale_kernel<bf16_t>, dim3(sgrid(nchunks)), dim3(256), 0, st, (const bf16_t*)x, scale, (bf16_t*)out, nchunks, cpg);
    else
        hipLaunchKernelGGL(row_scale_kernel<float>, dim3(sgrid(nchunks)), dim3(256), 0, st, (const float*)x, scale, (float*)out, nchunks, cpg);
    return check_launch("row_scale");
}

int colsum(int dtype, const void* x, int M, int N, float* out, hipStream_t st) {
    const int e = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    SAICV_REQUIRE(N % e == 0, "colsum: N=%d must be a multiple of %d", N, e);
    const int cpr = N / e;
    const int rpp = cpr >= CS_THREADS ? 1 : CS_THREADS / cpr;
    int slabs = M / (rpp * 8);                 // >= 8 passes per slab
    if (slabs > 128) slabs = 128;              // atomic adds per output address (see the kernel)
    if (slabs < 1) slabs = 1;
    const int rows_per = (M + slabs - 1) / slabs;
    slabs = (M + rows_per - 1) / rows_per;
    DetParts det;
    if (det.begin(st, slabs, (size_t)N, "colsum")) return -1;
    if (dtype == SAICV_DTYPE_BF16)
        hipLaunchKernelGGL(colsum_kernel<bf16_t>, dim3(slabs), dim3(CS_THREADS), 0, st, (const bf16_t*)x, M, N, rows_per, out, det.sink());
    else
        hipLaunchKernelGGL(colsum_kernel<float>, dim3(slabs), dim3(CS_THREADS), 0, st, (const float*)x, M, N, rows_per, out, det.sink());
    if (check_launch("colsum")) return -2;
    return det.fold(out, 0, (size_t)N);
}

}  // namespace saicv
