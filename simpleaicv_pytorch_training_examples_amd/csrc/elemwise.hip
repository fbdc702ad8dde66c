// Streaming glue of the convolutional backbones that reuse the hot-path blocks: the activations,
// gates and per-channel scales between the GEMM / depthwise / BatchNorm kernels of
//   reference SimpleAICV/classification/backbones/darknet.py:16-33  (nn.LeakyReLU(0.1) / nn.SiLU / nn.ReLU after BatchNorm),
//   van.py:86-93 (u * attn), :181-185 (x + layer_scale * branch), :176 (BatchNorm2d on a block input),
//   convformer.py:40-78 (ReLU between the pointwise linears), :157-163 (x + branch).
// All HBM-bound: 16-byte chunks (8 bf16 / 4 f32), one pass per tensor, fp32 arithmetic, grid-stride loops sized for
// ~8 workgroups per CU.  Reductions over pixels (per-channel statistics, the gradient of a per-channel scale) keep a
// chunk column per lane group, reduce across the workgroup in LDS and leave one fp32 atomic per channel and workgroup.
#include "common.h"
#include "saicv_internal.h"
#include "det.h"

namespace {

constexpr int EW_THREADS = 256;

inline int ew_grid(size_t items) {
    size_t g = (items + EW_THREADS - 1) / EW_THREADS;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

enum { ACT_RELU = 0, ACT_LEAKY = 1, ACT_SILU = 2 };

template <int KIND> DEVINL float act_val(float x, float slope) {
    if (KIND == ACT_RELU) return x > 0.f ? x : 0.f;
    if (KIND == ACT_LEAKY) return x > 0.f ? x : x * slope;
    const float s = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
    return x * s;
}
template <int KIND> DEVINL float act_der(float x, float slope) {
    if (KIND == ACT_RELU) return x > 0.f ? 1.f : 0.f;
    if (KIND == ACT_LEAKY) return x > 0.f ? 1.f : slope;
    const float s = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
    return s * fmaf(x, 1.f - s, 1.f);
}

// y = act(x)  |  dx = dy * act'(x)
template <typename T, int KIND, bool BWD>
__global__ __launch_bounds__(EW_THREADS) void act_kernel(const T* __restrict__ a, const T* __restrict__ x, T* __restrict__ out,
                                                         size_t chunks, float slope) {
    constexpr int N = Chunk<T>::N;
    for (size_t i = (size_t)blockIdx.x * EW_THREADS + threadIdx.x; i < chunks; i += (size_t)gridDim.x * EW_THREADS) {
        float xv[N], o[N];
        Chunk<T>::unpack(ld_chunk(x + i * N), xv);
        if (BWD) {
            float g[N];
            Chunk<T>::unpack(ld_chunk(a + i * N), g);
#pragma unroll
            for (int j = 0; j < N; ++j) o[j] = g[j] * act_der<KIND>(xv[j], slope);
        } else {
#pragma unroll
            for (int j = 0; j < N; ++j) o[j] = act_val<KIND>(xv[j], slope);
        }
        st_chunk(out + i * N, Chunk<T>::pack(o));
    }
}

// out = a * b  |  da = dy * b, db = dy * a
template <typename T, bool BWD>
__global__ __launch_bounds__(EW_THREADS) void mul_kernel(const T* __restrict__ dy, const T* __restrict__ a, const T* __restrict__ b,
                                                         T* __restrict__ o0, T* __restrict__ o1, size_t chunks) {
    constexpr int N = Chunk<T>::N;
    for (size_t i = (size_t)blockIdx.x * EW_THREADS + threadIdx.x; i < chunks; i += (size_t)gridDim.x * EW_THREADS) {
        float av[N], bv[N], r0[N];
        Chunk<T>::unpack(ld_chunk(a + i * N), av);
        Chunk<T>::unpack(ld_chunk(b + i * N), bv);
        if (BWD) {
            float g[N], r1[N];
            Chunk<T>::unpack(ld_chunk(dy + i * N), g);
#pragma unroll
            for (int j = 0; j < N; ++j) { r0[j] = g[j] * bv[j]; r1[j] = g[j] * av[j]; }
            if (o0) st_chunk(o0 + i * N, Chunk<T>::pack(r0));
            if (o1) st_chunk(o1 + i * N, Chunk<T>::pack(r1));
        } else {
#pragma unroll
            for (int j = 0; j < N; ++j) r0[j] = av[j] * bv[j];
            st_chunk(o0 + i * N, Chunk<T>::pack(r0));
        }
    }
}

// out[m][c] = x[m][c] + s[c] * y[m][c]      (x NULL: s * y;  s NULL: x + y)
template <typename T>
__global__ __launch_bounds__(EW_THREADS) void scale_add_kernel(const T* __restrict__ x, const T* __restrict__ y, const float* __restrict__ s,
                                                               T* __restrict__ out, size_t chunks, int cpr) {
    constexpr int N = Chunk<T>::N;
    for (size_t i = (size_t)blockIdx.x * EW_THREADS + threadIdx.x; i < chunks; i += (size_t)gridDim.x * EW_THREADS) {
        float yv[N], o[N];
        Chunk<T>::unpack(ld_chunk(y + i * N), yv);
        if (s) {
            const int c0 = (int)(i % (size_t)cpr) * N;
#pragma unroll
            for (int j = 0; j < N; ++j) yv[j] *= s[c0 + j];
        }
        if (x) {
            float xv[N];
            Chunk<T>::unpack(ld_chunk(x + i * N), xv);
#pragma unroll
            for (int j = 0; j < N; ++j) o[j] = xv[j] + yv[j];
        } else {
#pragma unroll
            for (int j = 0; j < N; ++j) o[j] = yv[j];
        }
        st_chunk(out + i * N, Chunk<T>::pack(o));
    }
}

// Column reductions over the rows of a [M][C] tensor.  A workgroup owns `cw` chunk columns (a power of two <= 64) and a range of
// rows; lane = column (fastest) x row lane.  MODE 0: sum[c] += x, sq[c] += x^2 (BatchNorm statistics of a block input);
// MODE 1: dy = s * dout (optional), ds[c] += dout * y (gradient of the per-channel scale, van.py:181).
template <typename T, int MODE>
__global__ __launch_bounds__(EW_THREADS) void colred_kernel(const T* __restrict__ p0, const T* __restrict__ p1, const float* __restrict__ s,
                                                            T* __restrict__ dy, float* __restrict__ r0, float* __restrict__ r1,
                                                            size_t M, int C, int cw, size_t rows_per_block, const saicv::DetSink det) {
    constexpr int N = Chunk<T>::N;
    __shared__ float red[EW_THREADS * N * (MODE == 0 ? 2 : 1)];
    const int cpr = C / N;
    const int col = blockIdx.x * cw + (threadIdx.x % cw);
    const int rl = threadIdx.x / cw, nrl = EW_THREADS / cw;
    const size_t m0 = (size_t)blockIdx.y * rows_per_block;
    size_t m1 = m0 + rows_per_block;
    if (m1 > M) m1 = M;
    float a0[N], a1[N], sv[N];
#pragma unroll
    for (int j = 0; j < N; ++j) { a0[j] = 0.f; a1[j] = 0.f; sv[j] = 1.f; }
    if (col < cpr) {
        if (MODE == 1 && s) {
#pragma unroll
            for (int j = 0; j < N; ++j) sv[j] = s[col * N + j];
        }
        for (size_t m = m0 + rl; m < m1; m += nrl) {
            const size_t off = m * C + (size_t)col * N;
            float v[N];
            Chunk<T>::unpack(ld_chunk(p0 + off), v);
            if (MODE == 0) {
#pragma unroll
                for (int j = 0; j < N; ++j) { a0[j] += v[j]; a1[j] = fmaf(v[j], v[j], a1[j]); }
            } else {
                if (r0) {
                    float yv[N];
                    Chunk<T>::unpack(ld_chunk(p1 + off), yv);
#pragma unroll
                    for (int j = 0; j < N; ++j) a0[j] = fmaf(v[j], yv[j], a0[j]);
                }
                if (dy) {
                    float o[N];
#pragma unroll
                    for (int j = 0; j < N; ++j) o[j] = v[j] * sv[j];
                    st_chunk(dy + off, Chunk<T>::pack(o));
                }
            }
        }
    }
    if (MODE == 1 && !r0) return;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        red[threadIdx.x * N + j] = a0[j];
        if (MODE == 0) red[EW_THREADS * N + threadIdx.x * N + j] = a1[j];
    }
    __syncthreads();
    if (rl == 0 && col < cpr) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            float t0 = 0.f, t1 = 0.f;
            for (int r = 0; r < nrl; ++r) {
                t0 += red[(r * cw + threadIdx.x) * N + j];
                if (MODE == 0) t1 += red[EW_THREADS * N + (r * cw + threadIdx.x) * N + j];
            }
            saicv::det_add(det, r0 + col * N + j, (size_t)col * N + j, blockIdx.y, t0);      // row range blockIdx.y = partial blockIdx.y
            if (MODE == 0) saicv::det_add(det, r1 + col * N + j, (size_t)C + col * N + j, blockIdx.y, t1);
        }
    }
}


// ---------------------------------------------------------------------------------------------- bilinear resize (feature pyramid)
// F.interpolate(top, size=(H, W), mode='bilinear') (align_corners = False) + lateral of the pyramid's top-down path (reference
// SimpleAICV/detection/models/fpn.py:57-75), NHWC.  ATen's arithmetic in fp32: scale = in / out, src = max(scale * (dst + 0.5) - 0.5, 0),
// i0 = int(src), i1 = i0 + (i0 < in - 1), weights (1 - frac, frac).  The output is fp32 whatever the inputs are (torch.autocast
// runs interpolate in fp32 and the add promotes).  ATen's backward scatters with atomics (not reproducible run to run); here the
// gradient is a GATHER: a source pixel walks the few destination rows / columns whose taps touch it, in a fixed order.
struct ResizeTap { int i0, i1; float w0, w1; };
DEVINL ResizeTap resize_tap(int d, float scale, int n_in) {
    const float src = fmaxf(scale * ((float)d + 0.5f) - 0.5f, 0.f);
    ResizeTap t;
    t.i0 = (int)src;
    t.i1 = t.i0 + (t.i0 < n_in - 1 ? 1 : 0);
    t.w1 = src - (float)t.i0;
    t.w0 = 1.f - t.w1;
    return t;
}
template <typename T> DEVINL f32x4 ld4(const T* p);
template <> DEVINL f32x4 ld4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <> DEVINL f32x4 ld4<bf16_t>(const bf16_t* p) {
    const u32x2 r = *reinterpret_cast<const u32x2*>(p);
    return f32x4{__uint_as_float(r[0] << 16), __uint_as_float(r[0] & 0xffff0000u), __uint_as_float(r[1] << 16), __uint_as_float(r[1] & 0xffff0000u)};
}
template <typename T> DEVINL void st4(T* p, f32x4 v);
template <> DEVINL void st4<float>(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
template <> DEVINL void st4<bf16_t>(bf16_t* p, f32x4 v) {
    bf16x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (bf16_t)v[j];
    *reinterpret_cast<bf16x4*>(p) = o;
}

template <typename TT, typename TL>
__global__ __launch_bounds__(EW_THREADS) void resize_add_fwd_kernel(const TT* __restrict__ top, const TL* __restrict__ lat, float* __restrict__ out,
                                                                    int Nn, int h, int w, int H, int W, int C, float sh, float sw) {
    const int c4 = C / 4;
    const size_t items = (size_t)Nn * H * W * c4;
    for (size_t i = (size_t)blockIdx.x * EW_THREADS + threadIdx.x; i < items; i += (size_t)gridDim.x * EW_THREADS) {
        const int c = (int)(i % c4);
        size_t t = i / c4;
        const int ox = (int)(t % W); t /= W;
        const int oy = (int)(t % H);
        const int n = (int)(t / H);
        const ResizeTap ty = resize_tap(oy, sh, h), tx = resize_tap(ox, sw, w);
        const TT* b = top + (size_t)n * h * w * C + (size_t)c * 4;
        const f32x4 v00 = ld4(b + ((size_t)ty.i0 * w + tx.i0) * C), v01 = ld4(b + ((size_t)ty.i0 * w + tx.i1) * C);
        const f32x4 v10 = ld4(b + ((size_t)ty.i1 * w + tx.i0) * C), v11 = ld4(b + ((size_t)ty.i1 * w + tx.i1) * C);
        f32x4 o = ty.w0 * (tx.w0 * v00 + tx.w1 * v01) + ty.w1 * (tx.w0 * v10 + tx.w1 * v11);
        if (lat) o += ld4(lat + i * 4);
        *reinterpret_cast<f32x4*>(out + i * 4) = o;
    }
}

// dtop[n, y, x, :] = sum over destination pixels (oy, ox) of wy(oy -> y) * wx(ox -> x) * dout[n, oy, ox, :]
template <typename TT>
__global__ __launch_bounds__(EW_THREADS) void resize_bwd_kernel(const float* __restrict__ dout, TT* __restrict__ dtop, int Nn, int h, int w, int H,
                                                                int W, int C, float sh, float sw) {
    const int c4 = C / 4;
    const size_t items = (size_t)Nn * h * w * c4;
    for (size_t i = (size_t)blockIdx.x * EW_THREADS + threadIdx.x; i < items; i += (size_t)gridDim.x * EW_THREADS) {
        const int c = (int)(i % c4);
        size_t t = i / c4;
        const int x = (int)(t % w); t /= w;
        const int y = (int)(t % h);
        const int n = (int)(t / h);
        // destination rows whose source position lies in (y - 1, y + 1): a generous integer window, the taps decide
        const int oy0 = max((int)(((float)y - 1.f + 0.5f) / sh - 0.5f) - 1, 0), oy1 = min((int)(((float)y + 1.f + 0.5f) / sh - 0.5f) + 2, H - 1);
        const int ox0 = max((int)(((float)x - 1.f + 0.5f) / sw - 0.5f) - 1, 0), ox1 = min((int)(((float)x + 1.f + 0.5f) / sw - 0.5f) + 2, W - 1);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* g = dout + (size_t)n * H * W * C + (size_t)c * 4;
        for (int oy = oy0; oy <= oy1; ++oy) {
            const ResizeTap ty = resize_tap(oy, sh, h);
            const float wy = (ty.i0 == y ? ty.w0 : 0.f) + (ty.i1 == y ? ty.w1 : 0.f);
            if (wy == 0.f) continue;
            for (int ox = ox0; ox <= ox1; ++ox) {
                const ResizeTap tx = resize_tap(ox, sw, w);
                const float wx = (tx.i0 == x ? tx.w0 : 0.f) + (tx.i1 == x ? tx.w1 : 0.f);
                if (wx == 0.f) continue;
                acc += (wy * wx) * *reinterpret_cast<const f32x4*>(g + ((size_t)oy * W + ox) * C);
            }
        }
        st4(dtop + i * 4, acc);
    }
}

struct ColGeom { int cw; dim3 grid; size_t rpb; };
inline ColGeom col_geom(size_t M, int cpr) {
    ColGeom g;
    g.cw = 1;
    while (g.cw * 2 <= cpr && g.cw * 2 <= 64) g.cw *= 2;
    const int groups = (cpr + g.cw - 1) / g.cw;
    // ~2048 workgroups over the chip, at most 1024 row ranges (= atomics per channel), at least 64 rows per range
    size_t ranges = (2048 + groups - 1) / groups;
    if (ranges > 1024) ranges = 1024;
    size_t rpb = (M + ranges - 1) / ranges;
    if (rpb < 64) rpb = 64;
    ranges = (M + rpb - 1) / rpb;
    g.rpb = rpb;
    g.grid = dim3(groups, (unsigned)ranges);
    return g;
}

inline int ew_check(const char* what, int dtype, size_t n) {
    SAICV_REQUIRE(dtype == SAICV_DTYPE_BF16 || dtype == SAICV_DTYPE_F32, "%s: dtype %d", what, dtype);
    const int e = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    SAICV_REQUIRE(n % e == 0, "%s: %zu elements are not whole 16-byte chunks", what, n);
    return 0;
}

// DINOv3 blocks (reference SimpleAICV/detection/models/backbones/dinov3vit.py).
// Rotary position embedding on the q and k thirds of a packed [B][N][3][heads][D] projection (:262-276 rope_rotate_half /
// rope_apply, :331-353 apply_rope): for tokens n >= prefix, y = x * cos + rot(x) * sin with rot(x) = [-x2, x1] over the two
// halves of the head dimension, computed in fp32 and rounded once (the reference casts q / k to the fp32 of sin / cos and back);
// v and the prefix tokens are copied.  TRANSPOSE: the gradient, dx = g * cos + rot^T(g * sin), rot^T(u) = [u2, -u1].
// One thread per 16-byte chunk of the LOWER half and its partner chunk in the upper half.
template <typename T, bool TRANSPOSE>
__global__ __launch_bounds__(EW_THREADS) void rope_kernel(const T* __restrict__ in, const float* __restrict__ sn, const float* __restrict__ cs,
                                                          T* __restrict__ out, size_t rows, int N, int heads, int D, int prefix) {
    constexpr int NE = Chunk<T>::N;
    const int half = D / 2, cph = half / NE;               // chunks per half head
    const size_t C3 = (size_t)3 * heads * D;
    const size_t items = rows * (size_t)3 * heads * cph;    // rows = B * N
    for (size_t i = (size_t)blockIdx.x * EW_THREADS + threadIdx.x; i < items; i += (size_t)gridDim.x * EW_THREADS) {
        const int c = (int)(i % cph);
        size_t t = i / cph;
        const int h = (int)(t % heads); t /= heads;
        const int part = (int)(t % 3);
        const size_t row = t / 3;
        const int n = (int)(row % N);
        const size_t off = row * C3 + ((size_t)part * heads + h) * D + (size_t)c * NE;
        const u32x4 lo = ld_chunk(in + off), hi = ld_chunk(in + off + half);
        if (part == 2 || n < prefix) {
            st_chunk(out + off, lo);
            st_chunk(out + off + half, hi);
            continue;
        }
        float a[NE], b[NE], s1[NE], s2[NE], c1[NE], c2[NE], o1[NE], o2[NE];
        Chunk<T>::unpack(lo, a);
        Chunk<T>::unpack(hi, b);
        const float* sp = sn + (size_t)(n - prefix) * D + c * NE;
        const float* cp = cs + (size_t)(n - prefix) * D + c * NE;
#pragma unroll
        for (int j = 0; j < NE; ++j) { s1[j] = sp[j]; s2[j] = sp[j + half]; c1[j] = cp[j]; c2[j] = cp[j + half]; }
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            if (TRANSPOSE) {
                o1[j] = a[j] * c1[j] + b[j] * s2[j];
                o2[j] = b[j] * c2[j] - a[j] * s1[j];
            } else {
                o1[j] = a[j] * c1[j] - b[j] * s1[j];
                o2[j] = b[j] * c2[j] + a[j] * s2[j];
            }
        }
        st_chunk(out + off, Chunk<T>::pack(o1));
        st_chunk(out + off + half, Chunk<T>::pack(o2));
    }
}

// SwiGLU gate (dinov3vit.py:137-140: hidden = silu(x1) * x2) and its backward in one pass each: the silu output is never written
template <typename T, bool BWD>
__global__ __launch_bounds__(EW_THREADS) void swiglu_kernel(const T* __restrict__ dy, const T* __restrict__ x1, const T* __restrict__ x2,
                                                            T* __restrict__ o1, T* __restrict__ o2, size_t chunks) {
    constexpr int N = Chunk<T>::N;
    for (size_t i = (size_t)blockIdx.x * EW_THREADS + threadIdx.x; i < chunks; i += (size_t)gridDim.x * EW_THREADS) {
        float a[N], b[N], r1[N], r2[N];
        Chunk<T>::unpack(ld_chunk(x1 + i * N), a);
        Chunk<T>::unpack(ld_chunk(x2 + i * N), b);
        if (BWD) {
            float g[N];
            Chunk<T>::unpack(ld_chunk(dy + i * N), g);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                r1[j] = g[j] * b[j] * act_der<ACT_SILU>(a[j], 0.f);
                r2[j] = g[j] * act_val<ACT_SILU>(a[j], 0.f);
            }
            st_chunk(o1 + i * N, Chunk<T>::pack(r1));
            st_chunk(o2 + i * N, Chunk<T>::pack(r2));
        } else {
#pragma unroll
            for (int j = 0; j < N; ++j) r1[j] = act_val<ACT_SILU>(a[j], 0.f) * b[j];
            st_chunk(o1 + i * N, Chunk<T>::pack(r1));
        }
    }
}

}  // namespace

extern "C" {

// nn.ReLU / nn.LeakyReLU(slope) / nn.SiLU on a dense tensor of n elements (kind 0 / 1 / 2): darknet.py:16-33, van.py:44, :103
int saicv_act_fwd(int dtype, int kind, double slope, const void* x, void* y, size_t n, void* stream) {
    if (ew_check("act_fwd", dtype, n)) return -1;
    SAICV_REQUIRE(kind >= 0 && kind <= 2, "act_fwd: kind %d", kind);
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const size_t chunks = n / (dtype == SAICV_DTYPE_BF16 ? 8 : 4);
#define ACT(TT, KK) hipLaunchKernelGGL((act_kernel<TT, KK, false>), dim3(ew_grid(chunks)), dim3(EW_THREADS), 0, st, (const TT*)nullptr, (const TT*)x, (TT*)y, chunks, (float)slope)
    if (dtype == SAICV_DTYPE_BF16) { if (kind == 0) ACT(bf16_t, 0); else if (kind == 1) ACT(bf16_t, 1); else ACT(bf16_t, 2); }
    else { if (kind == 0) ACT(float, 0); else if (kind == 1) ACT(float, 1); else ACT(float, 2); }
#undef ACT
    return saicv::check_launch("act_fwd");
}

// dx = dy * act'(x)   (x = the forward INPUT)
int saicv_act_bwd(int dtype, int kind, double slope, const void* dy, const void* x, void* dx, size_t n, void* stream) {
    if (ew_check("act_bwd", dtype, n)) return -1;
    SAICV_REQUIRE(kind >= 0 && kind <= 2, "act_bwd: kind %d", kind);
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const size_t chunks = n / (dtype == SAICV_DTYPE_BF16 ? 8 : 4);
#define ACT(TT, KK) hipLaunchKernelGGL((act_kernel<TT, KK, true>), dim3(ew_grid(chunks)), dim3(EW_THREADS), 0, st, (const TT*)dy, (const TT*)x, (TT*)dx, chunks, (float)slope)
    if (dtype == SAICV_DTYPE_BF16) { if (kind == 0) ACT(bf16_t, 0); else if (kind == 1) ACT(bf16_t, 1); else ACT(bf16_t, 2); }
    else { if (kind == 0) ACT(float, 0); else if (kind == 1) ACT(float, 1); else ACT(float, 2); }
#undef ACT
    return saicv::check_launch("act_bwd");
}

// out = a * b, same dense layout (van.py:91  `u * attn`)
int saicv_mul_fwd(int dtype, const void* a, const void* b, void* out, size_t n, void* stream) {
    if (ew_check("mul_fwd", dtype, n)) return -1;
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const size_t chunks = n / (dtype == SAICV_DTYPE_BF16 ? 8 : 4);
    if (dtype == SAICV_DTYPE_BF16)
        hipLaunchKernelGGL((mul_kernel<bf16_t, false>), dim3(ew_grid(chunks)), dim3(EW_THREADS), 0, st, (const bf16_t*)nullptr, (const bf16_t*)a,
                           (const bf16_t*)b, (bf16_t*)out, (bf16_t*)nullptr, chunks);
    else
        hipLaunchKernelGGL((mul_kernel<float, false>), dim3(ew_grid(chunks)), dim3(EW_THREADS), 0, st, (const float*)nullptr, (const float*)a,
                           (const float*)b, (float*)out, (float*)nullptr, chunks);
    return saicv::check_launch("mul_fwd");
}

// da = dy * b, db = dy * a (either may be NULL)
int saicv_mul_bwd(int dtype, const void* dy, const void* a, const void* b, void* da, void* db, size_t n, void* stream) {
    if (ew_check("mul_bwd", dtype, n)) return -1;
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const size_t chunks = n / (dtype == SAICV_DTYPE_BF16 ? 8 : 4);
    if (dtype == SAICV_DTYPE_BF16)
        hipLaunchKernelGGL((mul_kernel<bf16_t, true>), dim3(ew_grid(chunks)), dim3(EW_THREADS), 0, st, (const bf16_t*)dy, (const bf16_t*)a,
                           (const bf16_t*)b, (bf16_t*)da, (bf16_t*)db, chunks);
    else
        hipLaunchKernelGGL((mul_kernel<float, true>), dim3(ew_grid(chunks)), dim3(EW_THREADS), 0, st, (const float*)dy, (const float*)a,
                           (const float*)b, (float*)da, (float*)db, chunks);
    return saicv::check_launch("mul_bwd");
}

// out[m][c] = x[m][c] + s[c] * y[m][c] over a [M][C] (NHWC) tensor; x NULL: s * y; s NULL: x + y (residual joins of
// darknet.py Darknet53Block, convformer.py:157-163; layer scale of van.py:181-185)
int saicv_channel_scale_add_fwd(int dtype, const void* x, const void* y, const float* s, void* out, size_t M, int C, void* stream) {
    if (ew_check("channel_scale_add_fwd", dtype, (size_t)C)) return -1;
    SAICV_REQUIRE(x != nullptr || s != nullptr, "channel_scale_add_fwd: neither an addend nor a scale");
    if (M == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int e = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    const size_t chunks = M * (size_t)(C / e);
    if (dtype == SAICV_DTYPE_BF16)
        hipLaunchKernelGGL((scale_add_kernel<bf16_t>), dim3(ew_grid(chunks)), dim3(EW_THREADS), 0, st, (const bf16_t*)x, (const bf16_t*)y, s,
                           (bf16_t*)out, chunks, C / e);
    else
        hipLaunchKernelGGL((scale_add_kernel<float>), dim3(ew_grid(chunks)), dim3(EW_THREADS), 0, st, (const float*)x, (const float*)y, s,
                           (float*)out, chunks, C / e);
    return saicv::check_launch("channel_scale_add_fwd");
}

// backward of the scaled branch: dy = s * dout (dy NULL: not wanted; s NULL: ones), ds[c] += sum_m dout * y (ds NULL: not wanted;
// fp32 atomics into a buffer the caller zeroed or accumulates in)
int saicv_channel_scale_add_bwd(int dtype, const void* dout, const void* y, const float* s, void* dy, float* ds, size_t M, int C,
                                void* stream) {
    if (ew_check("channel_scale_add_bwd", dtype, (size_t)C)) return -1;
    SAICV_REQUIRE(ds == nullptr || y != nullptr, "channel_scale_add_bwd: the scale gradient needs the branch output");
    if (M == 0 || (dy == nullptr && ds == nullptr)) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int e = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    const ColGeom g = col_geom(M, C / e);
    saicv::DetParts det;
    if (det.begin(st, ds ? (int)g.grid.y : 0, (size_t)C, "channel_scale_add_bwd")) return -1;
    if (dtype == SAICV_DTYPE_BF16)
        hipLaunchKernelGGL((colred_kernel<bf16_t, 1>), g.grid, dim3(EW_THREADS), 0, st, (const bf16_t*)dout, (const bf16_t*)y, s, (bf16_t*)dy, ds,
                           (float*)nullptr, M, C, g.cw, g.rpb, det.sink());
    else
        hipLaunchKernelGGL((colred_kernel<float, 1>), g.grid, dim3(EW_THREADS), 0, st, (const float*)dout, (const float*)y, s, (float*)dy, ds,
                           (float*)nullptr, M, C, g.cw, g.rpb, det.sink());
    if (saicv::check_launch("channel_scale_add_bwd")) return -2;
    return det.fold(ds, 0, (size_t)C);
}

// per-channel sum and sum of squares of x[M][C], ADDED into sum[C] / sq[C] (fp32, zeroed by the caller): the statistics of a
// BatchNorm2d whose input is not a convolution output (van.py:176,178 norm1 / norm2, :260 stage norm; convformer.py:143,149);
// saicv_bn_finalize_fwd(rows = 1) turns them into mean / invstd / scale / shift as for the convolution epilogues' rows
int saicv_rope_apply(int dtype, const void* qkv, const float* sin_t, const float* cos_t, void* out, int B, int N, int heads, int D,
                     int prefix, int transpose, void* stream) {
    SAICV_REQUIRE(qkv && sin_t && cos_t && out && B > 0 && N > 0 && heads > 0 && prefix >= 0 && prefix <= N, "saicv_rope_apply: bad arguments");
    const int ne = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    SAICV_REQUIRE(dtype == SAICV_DTYPE_BF16 || dtype == SAICV_DTYPE_F32, "saicv_rope_apply: dtype %d", dtype);
    SAICV_REQUIRE(D % (2 * ne) == 0, "saicv_rope_apply: head dim %d must be a multiple of %d", D, 2 * ne);
    hipStream_t st = (hipStream_t)stream;
    const size_t rows = (size_t)B * N, items = rows * 3 * heads * (D / 2 / ne);
#define ROPE(T, TR) hipLaunchKernelGGL((rope_kernel<T, TR>), dim3(ew_grid(items)), dim3(EW_THREADS), 0, st, (const T*)qkv, sin_t, cos_t, (T*)out, rows, N, heads, D, prefix)
    if (dtype == SAICV_DTYPE_BF16) { if (transpose) ROPE(bf16_t, true); else ROPE(bf16_t, false); }
    else { if (transpose) ROPE(float, true); else ROPE(float, false); }
#undef ROPE
    return saicv::check_launch("rope_apply");
}

int saicv_swiglu_fwd(int dtype, const void* x1, const void* x2, void* out, size_t n, void* stream) {
    if (ew_check("swiglu_fwd", dtype, n)) return -1;
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const size_t chunks = n / (dtype == SAICV_DTYPE_BF16 ? 8 : 4);
    if (dtype == SAICV_DTYPE_BF16)
        hipLaunchKernelGGL((swiglu_kernel<bf16_t, false>), dim3(ew_grid(chunks)), dim3(EW_THREADS), 0, st, (const bf16_t*)nullptr, (const bf16_t*)x1,
                           (const bf16_t*)x2, (bf16_t*)out, (bf16_t*)nullptr, chunks);
    else
        hipLaunchKernelGGL((swiglu_kernel<float, false>), dim3(ew_grid(chunks)), dim3(EW_THREADS), 0, st, (const float*)nullptr, (const float*)x1,
                           (const float*)x2, (float*)out, (float*)nullptr, chunks);
    return saicv::check_launch("swiglu_fwd");
}

int saicv_swiglu_bwd(int dtype, const void* dy, const void* x1, const void* x2, void* dx1, void* dx2, size_t n, void* stream) {
    if (ew_check("swiglu_bwd", dtype, n)) return -1;
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const size_t chunks = n / (dtype == SAICV_DTYPE_BF16 ? 8 : 4);
    if (dtype == SAICV_DTYPE_BF16)
        hipLaunchKernelGGL((swiglu_kernel<bf16_t, true>), dim3(ew_grid(chunks)), dim3(EW_THREADS), 0, st, (const bf16_t*)dy, (const bf16_t*)x1,
                           (const bf16_t*)x2, (bf16_t*)dx1, (bf16_t*)dx2, chunks);
    else
        hipLaunchKernelGGL((swiglu_kernel<float, true>), dim3(ew_grid(chunks)), dim3(EW_THREADS), 0, st, (const float*)dy, (const float*)x1,
                           (const float*)x2, (float*)dx1, (float*)dx2, chunks);
    return saicv::check_launch("swiglu_bwd");
}

int saicv_bn_stats(int dtype, const void* x, size_t M, int C, float* sum, float* sq, void* stream) {
    if (ew_check("bn_stats", dtype, (size_t)C)) return -1;
    if (M == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int e = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    const ColGeom g = col_geom(M, C / e);
    saicv::DetParts det;
    if (det.begin(st, (int)g.grid.y, (size_t)2 * C, "bn_stats")) return -1;
    if (dtype == SAICV_DTYPE_BF16)
        hipLaunchKernelGGL((colred_kernel<bf16_t, 0>), g.grid, dim3(EW_THREADS), 0, st, (const bf16_t*)x, (const bf16_t*)nullptr, (const float*)nullptr,
                           (bf16_t*)nullptr, sum, sq, M, C, g.cw, g.rpb, det.sink());
    else
        hipLaunchKernelGGL((colred_kernel<float, 0>), g.grid, dim3(EW_THREADS), 0, st, (const float*)x, (const float*)nullptr, (const float*)nullptr,
                           (float*)nullptr, sum, sq, M, C, g.cw, g.rpb, det.sink());
    if (saicv::check_launch("bn_stats")) return -2;
    if (det.fold(sum, 0, (size_t)C)) return -1;
    return det.fold(sq, (size_t)C, (size_t)C);
}

// out[N, H, W, C] (fp32) = bilinear_resize(top[N, h, w, C]) + lateral[N, H, W, C] (NULL: none); dtype_top / dtype_lat: SAICV_BF16 | SAICV_F32
int saicv_resize_bilinear_add_fwd(int dtype_top, int dtype_lat, const void* top, const void* lateral, float* out, int N, int h, int w, int H,
                                  int W, int C, void* stream) {
    SAICV_REQUIRE(N > 0 && h > 0 && w > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "resize_bilinear_add_fwd: N=%d %dx%d -> %dx%d C=%d (C %% 4)", N, h, w, H, W, C);
    hipStream_t st = (hipStream_t)stream;
    const float sh = (float)h / (float)H, sw = (float)w / (float)W;
    const size_t items = (size_t)N * H * W * (C / 4);
    const bool tb = dtype_top == SAICV_DTYPE_BF16, lb = dtype_lat == SAICV_DTYPE_BF16;
#define RS_FWD(TT, TL) hipLaunchKernelGGL((resize_add_fwd_kernel<TT, TL>), dim3(ew_grid(items)), dim3(EW_THREADS), 0, st, (const TT*)top, (const TL*)lateral, out, N, h, w, H, W, C, sh, sw)
    if (tb && lb) RS_FWD(bf16_t, bf16_t); else if (tb) RS_FWD(bf16_t, float); else if (lb) RS_FWD(float, bf16_t); else RS_FWD(float, float);
#undef RS_FWD
    return saicv::check_launch("resize_bilinear_add_fwd");
}

// dtop[N, h, w, C] (dtype_top) = the transposed map applied to dout[N, H, W, C] (fp32): a gather in a fixed order
int saicv_resize_bilinear_bwd(int dtype_top, const float* dout, void* dtop, int N, int h, int w, int H, int W, int C, void* stream) {
    SAICV_REQUIRE(N > 0 && h > 0 && w > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "resize_bilinear_bwd: N=%d %dx%d -> %dx%d C=%d (C %% 4)", N, h, w, H, W, C);
    hipStream_t st = (hipStream_t)stream;
    const float sh = (float)h / (float)H, sw = (float)w / (float)W;
    const size_t items = (size_t)N * h * w * (C / 4);
    if (dtype_top == SAICV_DTYPE_BF16)
        hipLaunchKernelGGL((resize_bwd_kernel<bf16_t>), dim3(ew_grid(items)), dim3(EW_THREADS), 0, st, dout, (bf16_t*)dtop, N, h, w, H, W, C, sh, sw);
    else
        hipLaunchKernelGGL((resize_bwd_kernel<float>), dim3(ew_grid(items)), dim3(EW_THREADS), 0, st, dout, (float*)dtop, N, h, w, H, W, C, sh, sw);
    return saicv::check_launch("resize_bilinear_bwd");
}

}  // extern "C"
