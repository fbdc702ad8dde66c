// nn.GroupNorm (+ the ReLU behind it) on NHWC activations: the normalisation of the FCOS head towers (SURVEY.md §8(f) rank 2;
// reference SimpleAICV/detection/models/head.py:101-124, GroupNorm(32, 256) after every tower convolution, on five pyramid levels).
// ATen runs it in fp32 on NCHW data -- for an NHWC bf16 activation that is a layout copy, two dtype casts and five kernels each
// way.  Here: x [N][HW][C] in the compute dtype, fp32 arithmetic, 16-byte chunks, two streaming passes each way:
//   forward   gn_reduce<0>   per (sample, channel) sum and sum of squares            (fp32 atomics into a zeroed [2][N][C])
//             gn_coeffs      per (sample, group) mean / rstd -> per (sample, channel) a = rstd * gamma, b = beta - mean * a
//             gn_apply       y = x * a + b, optional ReLU
//   backward  gn_reduce<1>   per (sample, channel) A = sum dy', B = sum dy' * x, dy' = dy gated by [x * a + b > 0] when ReLU is fused
//             gn_bwd_coeffs  dgamma, dbeta, and per (sample, channel) p, q, r with dx = p * dy' + q * x + r
//             gn_apply_bwd   dx
#include "common.h"
#include "saicv_internal.h"
#include "det.h"

namespace {

constexpr int GN_THREADS = 256;

// blockIdx.x: group of `cw` chunk columns, blockIdx.y: row range of the sample, blockIdx.z: sample
template <typename T, int MODE>
__global__ __launch_bounds__(GN_THREADS) void gn_reduce_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ ab,
                                                               float* __restrict__ out0, float* __restrict__ out1, int HW, int C, int cw,
                                                               int rows_per_block, int relu, const saicv::DetSink det) {
    constexpr int N = Chunk<T>::N;
    __shared__ float red[GN_THREADS * N * 2];
    const int cpr = C / N;
    const int col = blockIdx.x * cw + (threadIdx.x % cw);
    const int rl = threadIdx.x / cw, nrl = GN_THREADS / cw;
    const int n = blockIdx.z;
    const int m0 = blockIdx.y * rows_per_block;
    const int m1 = min(m0 + rows_per_block, HW);
    float a0[N], a1[N], ca[N], cb[N];
#pragma unroll
    for (int j = 0; j < N; ++j) { a0[j] = 0.f; a1[j] = 0.f; ca[j] = 0.f; cb[j] = 0.f; }
    if (col < cpr) {
        if (MODE == 1 && relu) {
            const size_t NC = (size_t)gridDim.z * C;
#pragma unroll
            for (int j = 0; j < N; ++j) { ca[j] = ab[(size_t)n * C + col * N + j]; cb[j] = ab[NC + (size_t)n * C + col * N + j]; }
        }
        for (int m = m0 + rl; m < m1; m += nrl) {
            const size_t off = ((size_t)n * HW + m) * C + (size_t)col * N;
            float v[N];
            Chunk<T>::unpack(ld_chunk(x + off), v);
            if (MODE == 0) {
#pragma unroll
                for (int j = 0; j < N; ++j) { a0[j] += v[j]; a1[j] = fmaf(v[j], v[j], a1[j]); }
            } else {
                float g[N];
                Chunk<T>::unpack(ld_chunk(dy + off), g);
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const float gg = (relu && !(fmaf(v[j], ca[j], cb[j]) > 0.f)) ? 0.f : g[j];
                    a0[j] += gg;
                    a1[j] = fmaf(gg, v[j], a1[j]);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < N; ++j) {
        red[threadIdx.x * N + j] = a0[j];
        red[GN_THREADS * N + threadIdx.x * N + j] = a1[j];
    }
    __syncthreads();
    if (rl == 0 && col < cpr) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            float t0 = 0.f, t1 = 0.f;
            for (int r = 0; r < nrl; ++r) {
                t0 += red[(r * cw + threadIdx.x) * N + j];
                t1 += red[GN_THREADS * N + (r * cw + threadIdx.x) * N + j];
            }
            // row range blockIdx.y = partial blockIdx.y of the [2][N][C] sums
            saicv::det_add(det, out0 + (size_t)n * C + col * N + j, (size_t)n * C + col * N + j, blockIdx.y, t0);
            saicv::det_add(det, out1 + (size_t)n * C + col * N + j, (size_t)gridDim.z * C + (size_t)n * C + col * N + j, blockIdx.y, t1);
        }
    }
}

// thread = (sample, channel): statistics of its group from the channel sums, then the channel's affine coefficients
__global__ void gn_coeffs_kernel(const float* __restrict__ sums, const float* __restrict__ gamma, const float* __restrict__ beta,
                                 float* __restrict__ mean_rstd, float* __restrict__ ab, int Nn, int C, int cpg, float count, float eps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Nn * C) return;
    const int n = i / C, c = i - n * C, g0 = (c / cpg) * cpg;
    const float* S = sums + (size_t)n * C;
    const float* Q = sums + (size_t)Nn * C + (size_t)n * C;
    float s = 0.f, q = 0.f;
    for (int k = 0; k < cpg; ++k) { s += S[g0 + k]; q += Q[g0 + k]; }
    const float mean = s / count;
    const float var = fmaxf(q / count - mean * mean, 0.f);
    const float rstd = rsqrtf(var + eps);
    const float a = rstd * (gamma ? gamma[c] : 1.f);
    ab[i] = a;
    ab[(size_t)Nn * C + i] = (beta ? beta[c] : 0.f) - mean * a;
    if (c == g0) {
        const int G = C / cpg;
        mean_rstd[n * G + c / cpg] = mean;
        mean_rstd[Nn * G + n * G + c / cpg] = rstd;
    }
}

template <typename T>
__global__ __launch_bounds__(GN_THREADS) void gn_apply_kernel(const T* __restrict__ x, const float* __restrict__ ab, T* __restrict__ y, size_t chunks,
                                                              int HW, int C, size_t NC, int relu) {
    constexpr int N = Chunk<T>::N;
    const int cpr = C / N;
    for (size_t i = (size_t)blockIdx.x * GN_THREADS + threadIdx.x; i < chunks; i += (size_t)gridDim.x * GN_THREADS) {
        const size_t row = i / cpr;
        const int col = (int)(i - row * cpr);
        const size_t n = row / HW;
        const float* a = ab + n * C + col * N;
        float v[N], o[N];
        Chunk<T>::unpack(ld_chunk(x + i * N), v);
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const float t = fmaf(v[j], a[j], a[NC + j]);
            o[j] = relu ? fmaxf(t, 0.f) : t;
        }
        st_chunk(y + i * N, Chunk<T>::pack(o));
    }
}

// thread = channel c: loops over the samples; dgamma[c], dbeta[c] (+=) and the samples' p, q, r
__global__ void gn_bwd_coeffs_kernel(const float* __restrict__ AB, const float* __restrict__ mean_rstd, const float* __restrict__ gamma,
                                     float* __restrict__ pqr, float* __restrict__ dgamma, float* __restrict__ dbeta, int Nn, int C, int cpg,
                                     float count) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const int G = C / cpg, g = c / cpg, g0 = g * cpg;
    const size_t NC = (size_t)Nn * C;
    float dg = 0.f, db = 0.f;
    for (int n = 0; n < Nn; ++n) {
        const float* A = AB + (size_t)n * C;
        const float* B = AB + NC + (size_t)n * C;
        const float mean = mean_rstd[n * G + g], rstd = mean_rstd[Nn * G + n * G + g];
        dg += rstd * (B[c] - mean * A[c]);
        db += A[c];
        float s1 = 0.f, s2 = 0.f;
        for (int k = 0; k < cpg; ++k) {
            const float gm = gamma ? gamma[g0 + k] : 1.f;
            s1 += gm * A[g0 + k];
            s2 += gm * rstd * (B[g0 + k] - mean * A[g0 + k]);
        }
        const float q = -rstd * rstd * s2 / count;
        pqr[(size_t)n * C + c] = rstd * (gamma ? gamma[c] : 1.f);
        pqr[NC + (size_t)n * C + c] = q;
        pqr[2 * NC + (size_t)n * C + c] = -rstd * s1 / count - q * mean;
    }
    if (dgamma) atomicAdd(dgamma + c, dg);
    if (dbeta) atomicAdd(dbeta + c, db);
}

template <typename T>
__global__ __launch_bounds__(GN_THREADS) void gn_apply_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ ab,
                                                                  const float* __restrict__ pqr, T* __restrict__ dx, size_t chunks, int HW, int C,
                                                                  size_t NC, int relu) {
    constexpr int N = Chunk<T>::N;
    const int cpr = C / N;
    for (size_t i = (size_t)blockIdx.x * GN_THREADS + threadIdx.x; i < chunks; i += (size_t)gridDim.x * GN_THREADS) {
        const size_t row = i / cpr;
        const int col = (int)(i - row * cpr);
        const size_t n = row / HW;
        const float* ca = ab + n * C + col * N;                   // a, b of this sample's channels (y = x * a + b)
        const float* cp = pqr + n * C + col * N;                  // p, q, r
        float v[N], g[N], o[N];
        Chunk<T>::unpack(ld_chunk(x + i * N), v);
        Chunk<T>::unpack(ld_chunk(dy + i * N), g);
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const float pre = fmaf(v[j], ca[j], ca[NC + j]);
            const float keep = (relu == 0 || pre > 0.f) ? 1.f : 0.f;
            o[j] = fmaf(cp[j], g[j] * keep, fmaf(cp[NC + j], v[j], cp[2 * NC + j]));
        }
        st_chunk(dx + i * N, Chunk<T>::pack(o));
    }
}

struct GnGeom { int cw; dim3 grid; int rpb; };
inline GnGeom gn_geom(int Nn, int HW, int cpr) {
    GnGeom g;
    g.cw = 1;
    while (g.cw * 2 <= cpr && g.cw * 2 <= 64) g.cw *= 2;
    const int groups = (cpr + g.cw - 1) / g.cw;
    int ranges = (1024 + groups * Nn - 1) / (groups * Nn);
    if (ranges > 256) ranges = 256;
    int rpb = (HW + ranges - 1) / ranges;
    if (rpb < 32) rpb = 32;
    ranges = (HW + rpb - 1) / rpb;
    g.rpb = rpb;
    g.grid = dim3(groups, ranges, Nn);
    return g;
}

inline int gn_grid(size_t items) {
    size_t g = (items + GN_THREADS - 1) / GN_THREADS;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

inline int gn_check(const char* what, int dtype, int Nn, int HW, int C, int G) {
    SAICV_REQUIRE(dtype == SAICV_DTYPE_BF16 || dtype == SAICV_DTYPE_F32, "%s: dtype %d", what, dtype);
    const int e = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    SAICV_REQUIRE(Nn > 0 && Nn <= 65535 && HW > 0 && C > 0 && G > 0 && C % G == 0, "%s: N=%d HW=%d C=%d groups=%d", what, Nn, HW, C, G);
    SAICV_REQUIRE(C % e == 0, "%s: C=%d must be a multiple of %d", what, C, e);
    return 0;
}

}  // namespace

extern "C" {

size_t saicv_groupnorm_ws_floats(int N, int C) { return (size_t)N * C * 5; }

// y = [relu](GroupNorm(x)) on x [N][HW][C].  gamma / beta fp32 [C] (NULL: no affine).  Saves mean_rstd [2][N][G] and the
// per-(sample, channel) coefficients ab [2][N][C] for the backward.  ws: saicv_groupnorm_ws_floats(N, C) floats.
// Reference: nn.GroupNorm(32, inplanes) + nn.ReLU of SimpleAICV/detection/models/head.py:101-124.
int saicv_groupnorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean_rstd, float* ab, float* ws,
                        int N, int HW, int C, int G, double eps, int relu, void* stream) {
    if (gn_check("groupnorm_fwd", dtype, N, HW, C, G)) return -1;
    hipStream_t st = (hipStream_t)stream;
    const int e = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    float* sums = ws;                                   // [2][N][C]
    if (hipMemsetAsync(sums, 0, (size_t)2 * N * C * sizeof(float), st) != hipSuccess) { saicv::set_error("groupnorm_fwd: memset failed"); return -1; }
    const GnGeom g = gn_geom(N, HW, C / e);
    saicv::DetParts det;
    if (det.begin(st, (int)g.grid.y, (size_t)2 * N * C, "groupnorm_fwd")) return -1;
    if (dtype == SAICV_DTYPE_BF16)
        hipLaunchKernelGGL((gn_reduce_kernel<bf16_t, 0>), g.grid, dim3(GN_THREADS), 0, st, (const bf16_t*)x, (const bf16_t*)nullptr, (const float*)nullptr,
                           sums, sums + (size_t)N * C, HW, C, g.cw, g.rpb, 0, det.sink());
    else
        hipLaunchKernelGGL((gn_reduce_kernel<float, 0>), g.grid, dim3(GN_THREADS), 0, st, (const float*)x, (const float*)nullptr, (const float*)nullptr,
                           sums, sums + (size_t)N * C, HW, C, g.cw, g.rpb, 0, det.sink());
    if (det.fold(sums, 0, (size_t)2 * N * C)) return -1;
    hipLaunchKernelGGL(gn_coeffs_kernel, dim3((N * C + 255) / 256), dim3(256), 0, st, sums, gamma, beta, mean_rstd, ab, N, C, C / G,
                       (float)((double)HW * (C / G)), (float)eps);
    const size_t chunks = (size_t)N * HW * (C / e);
    if (dtype == SAICV_DTYPE_BF16)
        hipLaunchKernelGGL((gn_apply_kernel<bf16_t>), dim3(gn_grid(chunks)), dim3(GN_THREADS), 0, st, (const bf16_t*)x, ab, (bf16_t*)y, chunks, HW, C,
                           (size_t)N * C, relu);
    else
        hipLaunchKernelGGL((gn_apply_kernel<float>), dim3(gn_grid(chunks)), dim3(GN_THREADS), 0, st, (const float*)x, ab, (float*)y, chunks, HW, C,
                           (size_t)N * C, relu);
    return saicv::check_launch("groupnorm_fwd");
}

// dx, dgamma[C] / dbeta[C] (fp32, ADDED to; NULL: not wanted) from dy, x and what the forward saved (relu: the gate is
// recomputed from x * a + b)
int saicv_groupnorm_bwd(int dtype, const void* dy, const void* x, const float* gamma, const float* mean_rstd, const float* ab, void* dx,
                        float* dgamma, float* dbeta, float* ws, int N, int HW, int C, int G, int relu, void* stream) {
    if (gn_check("groupnorm_bwd", dtype, N, HW, C, G)) return -1;
    hipStream_t st = (hipStream_t)stream;
    const int e = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    float* AB = ws;                                     // [2][N][C]
    float* pqr = ws + (size_t)2 * N * C;                // [3][N][C]
    if (hipMemsetAsync(AB, 0, (size_t)2 * N * C * sizeof(float), st) != hipSuccess) { saicv::set_error("groupnorm_bwd: memset failed"); return -1; }
    const GnGeom g = gn_geom(N, HW, C / e);
    saicv::DetParts det;
    if (det.begin(st, (int)g.grid.y, (size_t)2 * N * C, "groupnorm_bwd")) return -1;
    if (dtype == SAICV_DTYPE_BF16)
        hipLaunchKernelGGL((gn_reduce_kernel<bf16_t, 1>), g.grid, dim3(GN_THREADS), 0, st, (const bf16_t*)x, (const bf16_t*)dy, ab, AB, AB + (size_t)N * C,
                           HW, C, g.cw, g.rpb, relu, det.sink());
    else
        hipLaunchKernelGGL((gn_reduce_kernel<float, 1>), g.grid, dim3(GN_THREADS), 0, st, (const float*)x, (const float*)dy, ab, AB, AB + (size_t)N * C,
                           HW, C, g.cw, g.rpb, relu, det.sink());
    if (det.fold(AB, 0, (size_t)2 * N * C)) return -1;
    hipLaunchKernelGGL(gn_bwd_coeffs_kernel, dim3((C + 63) / 64), dim3(64), 0, st, AB, mean_rstd, gamma, pqr, dgamma, dbeta, N, C, C / G,
                       (float)((double)HW * (C / G)));
    const size_t chunks = (size_t)N * HW * (C / e);
    if (dtype == SAICV_DTYPE_BF16)
        hipLaunchKernelGGL((gn_apply_bwd_kernel<bf16_t>), dim3(gn_grid(chunks)), dim3(GN_THREADS), 0, st, (const bf16_t*)dy, (const bf16_t*)x, ab, pqr,
                           (bf16_t*)dx, chunks, HW, C, (size_t)N * C, relu);
    else
        hipLaunchKernelGGL((gn_apply_bwd_kernel<float>), dim3(gn_grid(chunks)), dim3(GN_THREADS), 0, st, (const float*)dy, (const float*)x, ab, pqr,
                           (float*)dx, chunks, HW, C, (size_t)N * C, relu);
    return saicv::check_launch("groupnorm_bwd");
}

}  // extern "C"
