// BatchNorm2d (training + eval) fused with ReLU and the residual add, NHWC, gfx950.
//
// Replaces `native_batch_norm` + `relu_` (+ `add`) and their backward ATen dispatches of
// reference SimpleAICV/classification/backbones/resnet.py:41-42 (ConvBnActBlock),
// :94-95 (BasicBlock tail), :152-153 (Bottleneck tail).
//
// Forward:  per-channel sum / sum-of-squares partials come from the conv epilogue
//           (igemm.hip); `bn_reduce_partials` + `bn_finalize_fwd` turn them into
//           mean / invstd / (scale, shift) and update running stats; `bn_act_fwd`
//           streams y -> z = relu(y*scale + shift [+ res]) with 16-byte accesses.
// Backward: `bn_bwd_reduce` streams (dz, z, y) once for sum(g), sum(g*xhat);
//           `bn_finalize_bwd` produces dgamma/dbeta and the per-channel coefficients;
//           `bn_bwd_apply` streams (dz, z, y) again and writes dy (and dres = g).
// All statistics are fp32; activations are bf16 (perf mode) or fp32 (parity mode).
// These kernels are HBM-bound: algorithmic bytes are listed in DESIGN.md.
#include "common.h"
#include "saicv_internal.h"

namespace {

// r05 -- cache policy of the two big streaming kernels.  Their INPUTS (the convolution's raw output y and the residual in the forward
// pass; dz, y and the mask in the backward pass) are read exactly once by these kernels: streaming loads ("nt") keep them from
// displacing what the neighbouring kernels re-read.  Their OUTPUTS (z; dy, dres) are read by the very next kernel and keep the default
// policy.  Same box, library A/B on the ResNet-50 step (profiles/r05_nt_experiments.md): streaming loads 21.39 -> 21.22 ms (forward) and
// -> 21.21 ms (backward); streaming STORES lose (21.55 / 21.50 ms).  -DSAICV_BN_FWD_LD_PLAIN / _BWD_LD_PLAIN / _FWD_ST_NT / _BWD_ST_NT
// build the other policies (scripts/build_variant_lib.py).
#ifdef SAICV_BN_FWD_LD_PLAIN
#define BNF_LD ld_chunk
#else
#define BNF_LD ld_chunk_nt
#endif
#ifdef SAICV_BN_FWD_ST_NT
#define BNF_ST st_chunk_nt
#else
#define BNF_ST st_chunk
#endif
#ifdef SAICV_BN_BWD_LD_PLAIN
#define BNB_LD ld_chunk
#else
#define BNB_LD ld_chunk_nt
#endif
#ifdef SAICV_BN_BWD_ST_NT
#define BNB_ST st_chunk_nt
#else
#define BNB_ST st_chunk
#endif

constexpr int kMaxBlocks = 1024;        // four 256-thread blocks per CU (sweep 512..16384: flat within 1 %, 2048 the slowest)

// ---------------------------------------------------------------- partial reduce [P][C] -> [Y][C]
__global__ __launch_bounds__(256) void bn_reduce_partials_kernel(const float* __restrict__ a,
                                                                 const float* __restrict__ b, int P,
                                                                 int C, float* __restrict__ oa,
                                                                 float* __restrict__ ob, int rows_per) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int ty = threadIdx.x >> 6;
    const int p0 = blockIdx.y * rows_per;
    const int p1 = min(P, p0 + rows_per);
    float sa = 0.f, sb = 0.f;
    if (c < C) {
        // latency-bound without it: four rows (eight loads) in flight per thread
        float a4[4] = {0.f, 0.f, 0.f, 0.f}, b4[4] = {0.f, 0.f, 0.f, 0.f};
        int p = p0 + ty;
        for (; p + 12 < p1; p += 16) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a4[u] += a[(size_t)(p + 4 * u) * C + c];
                b4[u] += b[(size_t)(p + 4 * u) * C + c];
            }
        }
        for (; p < p1; p += 4) {
            a4[0] += a[(size_t)p * C + c];
            b4[0] += b[(size_t)p * C + c];
        }
        sa = (a4[0] + a4[1]) + (a4[2] + a4[3]);
        sb = (b4[0] + b4[1]) + (b4[2] + b4[3]);
    }
    __shared__ float la[4][64], lb[4][64];
    la[ty][threadIdx.x & 63] = sa;
    lb[ty][threadIdx.x & 63] = sb;
    __syncthreads();
    if (ty == 0 && c < C) {
        const int x = threadIdx.x;
        oa[(size_t)blockIdx.y * C + c] = la[0][x] + la[1][x] + la[2][x] + la[3][x];
        ob[(size_t)blockIdx.y * C + c] = lb[0][x] + lb[1][x] + lb[2][x] + lb[3][x];
    }
}

// column sums of the partial rows by one block of NW wavefronts per 64 channels: wave ty takes rows ty, ty + NW, ...
// eight at a time with all of their loads in flight, LDS combines the waves.  NW = 4 serves P <= 32 (one batch); NW = 16
// serves P <= 1024 in at most eight batches (~5 us), which is cheaper than a separate partial-reduction launch plus a
// P <= 32 finalize (5 + 5 us): only the 56 x 56 layers (P = 3136..6272) still take the two-kernel route.
// (A one-wave serial loop over 32 rows cost ~10 us of pure load latency per BatchNorm, twice per step.)
template <int NW>
DEVINL void finalize_colsum(const float* __restrict__ a, const float* __restrict__ b, int P, int C, int c, int ty,
                            float& sa, float& sb) {
    sa = 0.f;
    sb = 0.f;
    for (int p0 = ty; p0 < P; p0 += NW * 8) {
        float va[8], vb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = p0 + NW * u;
            const bool ok = c < C && p < P;
            va[u] = ok ? a[(size_t)p * C + c] : 0.f;
            vb[u] = ok ? b[(size_t)p * C + c] : 0.f;
        }
        sa += ((va[0] + va[1]) + (va[2] + va[3])) + ((va[4] + va[5]) + (va[6] + va[7]));
        sb += ((vb[0] + vb[1]) + (vb[2] + vb[3])) + ((vb[4] + vb[5]) + (vb[6] + vb[7]));
    }
    __shared__ float la[NW][64], lb[NW][64];
    const int x = threadIdx.x & 63;
    la[ty][x] = sa;
    lb[ty][x] = sb;
    __syncthreads();
    sa = 0.f;
    sb = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) { sa += la[w][x]; sb += lb[w][x]; }
}
constexpr int kFinalizeDirect = 1024;    // partial rows a finalize kernel takes without a reduction launch

// ---------------------------------------------------------------- forward finalize
// Follows torch.nn.BatchNorm2d training semantics (biased var for normalisation, unbiased
// var into running_var, momentum 0.1) as used by reference resnet.py:41.
template <int NW>
__global__ __launch_bounds__(64 * NW) void bn_finalize_fwd_kernel(const float* __restrict__ sum, const float* __restrict__ sq,
                                       int P, int C, float count, const float* __restrict__ gamma,
                                       const float* __restrict__ beta, float* running_mean,
                                       float* running_var, float momentum, float eps,
                                       float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                       float* __restrict__ scale, float* __restrict__ shift,
                                       long long* __restrict__ num_batches_tracked) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int ty = threadIdx.x >> 6;
    if (blockIdx.x == 0 && threadIdx.x == 0 && num_batches_tracked != nullptr) *num_batches_tracked += 1;     // BatchNorm2d bookkeeping, no extra launch
    float s, q;
    finalize_colsum<NW>(sum, sq, P, C, c, ty, s, q);
    if (c >= C || ty != 0) return;
    const float mean = s / count;
    float var = q / count - mean * mean;
    var = fmaxf(var, 0.f);
    const float invstd = rsqrtf(var + eps);
    mean_out[c] = mean;
    invstd_out[c] = invstd;
    const float g = gamma ? gamma[c] : 1.f;
    const float bt = beta ? beta[c] : 0.f;
    scale[c] = g * invstd;
    shift[c] = bt - mean * g * invstd;
    if (running_mean) {
        const float unbiased = count > 1.f ? var * count / (count - 1.f) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
}

// eval mode: scale/shift from running statistics
__global__ void bn_eval_coeffs_kernel(int C, const float* __restrict__ gamma,
                                      const float* __restrict__ beta,
                                      const float* __restrict__ running_mean,
                                      const float* __restrict__ running_var, float eps,
                                      float* __restrict__ scale, float* __restrict__ shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float invstd = rsqrtf(running_var[c] + eps);
    const float g = gamma ? gamma[c] : 1.f;
    const float bt = beta ? beta[c] : 0.f;
    scale[c] = g * invstd;
    shift[c] = bt - running_mean[c] * g * invstd;
}

// ---------------------------------------------------------------- forward apply
// HOIST: the launch guarantees (gridDim.x*256) % (C/N) == 0, so a thread sees the same N
// channels on every grid-stride iteration and keeps scale/shift in registers.
// SELF: the statistics arrive as a few rows of atomically accumulated sums (igemm.hip, stat_atomic_rows) and EVERY block
// finalises all channels into LDS first (C <= kInlineMaxC; block 0 also stores mean / invstd and updates the running
// statistics): the partial-reduce and finalize launches between the convolution and this kernel disappear.
constexpr int kInlineMaxC = 2048;
struct BnFwdInline {
    const float* sum; const float* sq; int rows; float count;
    const float* gamma; const float* beta; float eps, momentum;
    float* running_mean; float* running_var; long long* num_batches_tracked;
    float* mean_out; float* invstd_out;
    // r04: the residual is a RAW convolution output whose BatchNorm-apply happens here (the downsample branch of a residual
    // block: res = res_scale[c] * res + res_shift[c]); NULL = the residual is a finished activation
    const float* res_scale; const float* res_shift;
};

template <typename T, bool RELU, bool RES, bool HOIST, bool SELF>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const T* __restrict__ y,
                                                         const T* __restrict__ res, T* __restrict__ z,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ shift,
                                                         size_t nchunks, int C, uint8_t* __restrict__ mask,
                                                         const BnFwdInline st) {
    constexpr int N = Chunk<T>::N;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t first = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    __shared__ __attribute__((aligned(16))) float lsc[SELF ? kInlineMaxC : 4], lsh[SELF ? kInlineMaxC : 4];
    if (SELF) {
        if (blockIdx.x == 0 && threadIdx.x == 0 && st.num_batches_tracked != nullptr) *st.num_batches_tracked += 1;
        for (int c = threadIdx.x; c < C; c += 256) {
            float s = 0.f, q = 0.f;
            for (int r = 0; r < st.rows; ++r) { s += st.sum[(size_t)r * C + c]; q += st.sq[(size_t)r * C + c]; }
            const float mean = s / st.count;
            const float var = fmaxf(q / st.count - mean * mean, 0.f);
            const float invstd = rsqrtf(var + st.eps);
            const float g = st.gamma ? st.gamma[c] : 1.f, bt = st.beta ? st.beta[c] : 0.f;
            lsc[c] = g * invstd;
            lsh[c] = bt - mean * g * invstd;
            if (blockIdx.x == 0) {
                st.mean_out[c] = mean;
                st.invstd_out[c] = invstd;
                if (st.running_mean) {
                    const float unbiased = st.count > 1.f ? var * st.count / (st.count - 1.f) : var;
                    st.running_mean[c] = (1.f - st.momentum) * st.running_mean[c] + st.momentum * mean;
                    st.running_var[c] = (1.f - st.momentum) * st.running_var[c] + st.momentum * unbiased;
                }
            }
        }
        __syncthreads();
        scale = lsc;
        shift = lsh;
    }
    float sc[N], sh[N], rsc[N], rsh[N];
    const bool res_affine = RES && st.res_scale != nullptr;                  // uniform
    auto load_coeffs = [&](int c0) {
#pragma unroll
        for (int j = 0; j < N; j += 4) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(scale + c0 + j);
            const f32x4 b = *reinterpret_cast<const f32x4*>(shift + c0 + j);
#pragma unroll
            for (int k = 0; k < 4; ++k) { sc[j + k] = a[k]; sh[j + k] = b[k]; }
            if (res_affine) {
                const f32x4 ra = *reinterpret_cast<const f32x4*>(st.res_scale + c0 + j);
                const f32x4 rb = *reinterpret_cast<const f32x4*>(st.res_shift + c0 + j);
#pragma unroll
                for (int k = 0; k < 4; ++k) { rsc[j + k] = ra[k]; rsh[j + k] = rb[k]; }
            }
        }
    };
    if (HOIST) load_coeffs((int)((first * N) % (size_t)C));
    for (size_t i = first; i < nchunks; i += stride) {
        if (!HOIST) load_coeffs((int)((i * N) % (size_t)C));
        float v[N];
        Chunk<T>::unpack(BNF_LD(y + i * N), v);
        float rr[N];
        if (RES) Chunk<T>::unpack(BNF_LD(res + i * N), rr);
        unsigned bits = 0;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            float o = fmaf(v[j], sc[j], sh[j]);
            if (RES) o += res_affine ? fmaf(rr[j], rsc[j], rsh[j]) : rr[j];
            if (RELU) {
                bits |= (o > 0.f ? 1u : 0u) << j;
                o = fmaxf(o, 0.f);
            }
            v[j] = o;
        }
        BNF_ST(z + i * N, Chunk<T>::pack(v));
        if (RELU && mask != nullptr) mask[i] = (uint8_t)bits;     // one bit per element: all backward needs of z
    }
}

// ---------------------------------------------------------------- backward reduce
// Block handles a slab of rows; thread owns one chunk column (N channels) and strides over
// rows; per-block partials [slab][C].  Any C that is a multiple of N works.
template <typename T, bool RELU>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dz,
                                                            const T* __restrict__ z,
                                                            const uint8_t* __restrict__ mask,
                                                            const T* __restrict__ y,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, int M,
                                                            int C, int rows_per,
                                                            float* __restrict__ part_g,
                                                            float* __restrict__ part_gx) {
    constexpr int N = Chunk<T>::N;
    const int cpr = C / N;                       // chunk columns per row
    const int cols = cpr < 256 ? cpr : 256;      // columns handled per pass by this block
    const int rpp = 256 / cols;                  // row lanes per pass
    const int tx = threadIdx.x % cols;
    const int ty = threadIdx.x / cols;           // may be >= rpp for the few left-over threads
    const int r0 = blockIdx.x * rows_per;
    const int r1 = min(M, r0 + rows_per);
    __shared__ float lg_[256 * 8], lx_[256 * 8];
    for (int base = 0; base < cpr; base += cols) {           // uniform trip count
        const int cb = base + tx;
        const bool active = (cb < cpr) && (ty < rpp);
        const int c0 = active ? cb * N : 0;
        float mu[N], is[N], ag[N], ax[N];
#pragma unroll
        for (int j = 0; j < N; ++j) { mu[j] = mean[c0 + j]; is[j] = invstd[c0 + j]; ag[j] = 0.f; ax[j] = 0.f; }
        if (active) {
            // rows are taken four at a time: all loads of the group are issued before the first use, which keeps
            // ~12 16-byte requests per thread in flight (the one-row loop ran at 3.3 TB/s)
            auto row_bits = [&](size_t e) -> unsigned {
                if (!RELU) return 0xffu;
                if (mask != nullptr) return mask[e / N];
                float zz[N];
                Chunk<T>::unpack(ld_chunk(z + e), zz);
                unsigned b = 0;
#pragma unroll
                for (int j = 0; j < N; ++j) b |= (zz[j] > 0.f ? 1u : 0u) << j;
                return b;
            };
            auto accumulate = [&](const u32x4& cg, const u32x4& cy, unsigned bits) {
                float g[N], yy[N];
                Chunk<T>::unpack(cg, g);
                Chunk<T>::unpack(cy, yy);
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const float gj = ((bits >> j) & 1u) ? g[j] : 0.f;
                    ag[j] += gj;
                    ax[j] += gj * (yy[j] - mu[j]) * is[j];
                }
            };
            int r = r0 + ty;
            for (; r + 3 * rpp < r1; r += 4 * rpp) {
                u32x4 cg[4], cy[4];
                unsigned bits[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const size_t e = (size_t)(r + u * rpp) * C + c0;
                    cg[u] = ld_chunk(dz + e);
                    cy[u] = ld_chunk(y + e);
                    bits[u] = row_bits(e);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) accumulate(cg[u], cy[u], bits[u]);
            }
            for (; r < r1; r += rpp) {
                const size_t e = (size_t)r * C + c0;
                const u32x4 cg = ld_chunk(dz + e), cy = ld_chunk(y + e);
                accumulate(cg, cy, row_bits(e));
            }
        }
        // combine the rpp row-lanes of this column through LDS
#pragma unroll
        for (int j = 0; j < N; ++j) { lg_[threadIdx.x * N + j] = ag[j]; lx_[threadIdx.x * N + j] = ax[j]; }
        __syncthreads();
        if (active && ty == 0) {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                float sg = 0.f, sx = 0.f;
                for (int t = 0; t < rpp; ++t) {
                    sg += lg_[(t * cols + tx) * N + j];
                    sx += lx_[(t * cols + tx) * N + j];
                }
                part_g[(size_t)blockIdx.x * C + c0 + j] = sg;
                part_gx[(size_t)blockIdx.x * C + c0 + j] = sx;
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------- backward finalize
// dy = A*(g - mg) - A*xhat*mgx  with A = gamma*invstd, mg = sum(g)/M, mgx = sum(g*xhat)/M
// written as dy = ca*g + cb*y + cc per channel.
template <int NW>
__global__ __launch_bounds__(64 * NW) void bn_finalize_bwd_kernel(const float* __restrict__ pg, const float* __restrict__ pgx,
                                       int P, int C, float count, const float* __restrict__ gamma,
                                       const float* __restrict__ mean,
                                       const float* __restrict__ invstd, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta, float* __restrict__ ca,
                                       float* __restrict__ cb, float* __restrict__ cc, int accumulate) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int ty = threadIdx.x >> 6;
    float sg, sx;
    finalize_colsum<NW>(pg, pgx, P, C, c, ty, sg, sx);
    if (c >= C || ty != 0) return;
    if (dgamma) dgamma[c] = accumulate ? dgamma[c] + sx : sx;
    if (dbeta) dbeta[c] = accumulate ? dbeta[c] + sg : sg;
    const float g = gamma ? gamma[c] : 1.f;
    const float A = g * invstd[c];
    const float mg = sg / count, mgx = sx / count;
    const float B = -A * invstd[c] * mgx;      // coefficient on y
    ca[c] = A;
    cb[c] = B;
    cc[c] = -A * mg - B * mean[c];
}

// ---------------------------------------------------------------- backward apply
// SELF: as in the forward kernel -- the sums arrive as a few atomically accumulated rows, every block derives the
// per-channel coefficients into LDS, block 0 also writes dgamma / dbeta.
struct BnBwdInline {
    const float* pg; const float* pgx; int rows; float count;
    const float* gamma; const float* mean; const float* invstd;
    float* dgamma; float* dbeta; int accumulate;
};

template <typename T, bool RELU, bool RES, bool HOIST, bool SELF>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dz,
                                                           const T* __restrict__ z,
                                                           const T* __restrict__ y,
                                                           const float* __restrict__ ca,
                                                           const float* __restrict__ cb,
                                                           const float* __restrict__ cc,
                                                           T* __restrict__ dy, T* __restrict__ dres,
                                                           size_t nchunks, int C, const uint8_t* __restrict__ mask,
                                                           const BnBwdInline st) {
    constexpr int N = Chunk<T>::N;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t first = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    __shared__ __attribute__((aligned(16))) float la[SELF ? kInlineMaxC : 4], lb[SELF ? kInlineMaxC : 4], lc[SELF ? kInlineMaxC : 4];
    if (SELF) {
        for (int c = threadIdx.x; c < C; c += 256) {
            float sg = 0.f, sx = 0.f;
            for (int r = 0; r < st.rows; ++r) { sg += st.pg[(size_t)r * C + c]; sx += st.pgx[(size_t)r * C + c]; }
            if (blockIdx.x == 0) {
                if (st.dgamma) st.dgamma[c] = st.accumulate ? st.dgamma[c] + sx : sx;
                if (st.dbeta) st.dbeta[c] = st.accumulate ? st.dbeta[c] + sg : sg;
            }
            const float g = st.gamma ? st.gamma[c] : 1.f;
            const float A = g * st.invstd[c];
            const float mg = sg / st.count, mgx = sx / st.count;
            const float B = -A * st.invstd[c] * mgx;
            la[c] = A;
            lb[c] = B;
            lc[c] = -A * mg - B * st.mean[c];
        }
        __syncthreads();
        ca = la; cb = lb; cc = lc;
    }
    float ka[N], kb[N], kc[N];
    auto load_coeffs = [&](int c0) {
#pragma unroll
        for (int j = 0; j < N; j += 4) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(ca + c0 + j);
            const f32x4 b = *reinterpret_cast<const f32x4*>(cb + c0 + j);
            const f32x4 c = *reinterpret_cast<const f32x4*>(cc + c0 + j);
#pragma unroll
            for (int k = 0; k < 4; ++k) { ka[j + k] = a[k]; kb[j + k] = b[k]; kc[j + k] = c[k]; }
        }
    };
    if (HOIST) load_coeffs((int)((first * N) % (size_t)C));
    for (size_t i = first; i < nchunks; i += stride) {
        if (!HOIST) load_coeffs((int)((i * N) % (size_t)C));
        float g[N], yy[N], zz[N], o[N];
        Chunk<T>::unpack(BNB_LD(dz + i * N), g);
        Chunk<T>::unpack(BNB_LD(y + i * N), yy);
        unsigned bits = 0xffu;
        if (RELU) {
            if (mask != nullptr) {
                bits = mask[i];
            } else {
                Chunk<T>::unpack(ld_chunk(z + i * N), zz);
                bits = 0;
#pragma unroll
                for (int j = 0; j < N; ++j) bits |= (zz[j] > 0.f ? 1u : 0u) << j;
            }
        }
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const float gj = ((bits >> j) & 1u) ? g[j] : 0.f;
            g[j] = gj;
            o[j] = fmaf(ka[j], gj, fmaf(kb[j], yy[j], kc[j]));
        }
        BNB_ST(dy + i * N, Chunk<T>::pack(o));
        if (RES) BNB_ST(dres + i * N, Chunk<T>::pack(g));
    }
}

inline int stream_grid(size_t nchunks) {
    size_t b = (nchunks + 255) / 256;
    static const int cap = getenv("SAICV_BN_BLOCKS") ? atoi(getenv("SAICV_BN_BLOCKS")) : kMaxBlocks;     // tuning aid
    if (b > (size_t)cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

// reduce [P][C] partial pairs down to at most 32 rows (in place into ws), returns new P
int reduce_partials(const float*& a, const float*& b, int P, int C, float* ws, hipStream_t st) {
    static const int direct = getenv("SAICV_BN_DIRECT") ? atoi(getenv("SAICV_BN_DIRECT")) : kFinalizeDirect;   // tuning aid
    if (P <= direct) return P;
    const int Y = 32;
    const int rows_per = (P + Y - 1) / Y;
    const int y_used = (P + rows_per - 1) / rows_per;
    float* oa = ws;
    float* ob = ws + (size_t)Y * C;
    dim3 grid((C + 63) / 64, y_used);
    hipLaunchKernelGGL(bn_reduce_partials_kernel, grid, dim3(256), 0, st, a, b, P, C, oa, ob, rows_per);
    a = oa;
    b = ob;
    return y_used;
}

}  // namespace

namespace saicv {

// workspace floats needed by bn_finalize (two [32][C] slabs)
size_t bn_ws_floats(int C) { return (size_t)64 * C; }

int bn_finalize_fwd(const float* sum, const float* sq, int P, int C, double count, const float* gamma,
                    const float* beta, float* running_mean, float* running_var, double momentum,
                    double eps, float* mean, float* invstd, float* scale, float* shift, float* ws,
                    long long* num_batches_tracked, hipStream_t st) {
    P = reduce_partials(sum, sq, P, C, ws, st);
    hipLaunchKernelGGL(P <= 32 ? bn_finalize_fwd_kernel<4> : bn_finalize_fwd_kernel<16>, dim3((C + 63) / 64), dim3(P <= 32 ? 256 : 1024), 0, st, sum, sq, P, C,
                       (float)count, gamma, beta, running_mean, running_var, (float)momentum, (float)eps,
                       mean, invstd, scale, shift, num_batches_tracked);
    return check_launch("bn_finalize_fwd");
}

int bn_eval_coeffs(int C, const float* gamma, const float* beta, const float* running_mean,
                   const float* running_var, double eps, float* scale, float* shift, hipStream_t st) {
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((C + 63) / 64), dim3(64), 0, st, C, gamma, beta,
                       running_mean, running_var, (float)eps, scale, shift);
    return check_launch("bn_eval_coeffs");
}

template <typename T>
static int bn_act_fwd_t(const void* y, const void* res, void* z, const float* scale,
                        const float* shift, size_t M, int C, int relu, uint8_t* mask, hipStream_t st,
                        const BnFwdInline* inl = nullptr, const float* res_scale = nullptr, const float* res_shift = nullptr) {
    constexpr int N = Chunk<T>::N;
    const size_t nchunks = M * (size_t)C / N;
    const int grid = stream_grid(nchunks);
    const T* yy = (const T*)y; const T* rr = (const T*)res; T* zz = (T*)z;
    const bool hoist = ((size_t)grid * 256) % (size_t)(C / N) == 0;
    BnFwdInline none = {};
    none.res_scale = res_scale;
    none.res_shift = res_shift;
#define LAUNCH2(R, S, H) do { if (inl) hipLaunchKernelGGL((bn_act_fwd_kernel<T, R, S, H, true>), dim3(grid), dim3(256), 0, st, yy, rr, zz, scale, shift, nchunks, C, mask, *inl); \
                              else hipLaunchKernelGGL((bn_act_fwd_kernel<T, R, S, H, false>), dim3(grid), dim3(256), 0, st, yy, rr, zz, scale, shift, nchunks, C, mask, none); } while (0)
#define LAUNCH(R, S) do { if (hoist) LAUNCH2(R, S, true); else LAUNCH2(R, S, false); } while (0)
    if (relu) { if (res) LAUNCH(true, true); else LAUNCH(true, false); }
    else      { if (res) LAUNCH(false, true); else LAUNCH(false, false); }
#undef LAUNCH
#undef LAUNCH2
    return check_launch("bn_act_fwd");
}

int bn_act_fwd(int dtype, const void* y, const void* res, void* z, const float* scale,
               const float* shift, size_t M, int C, int relu, void* relu_mask, hipStream_t st) {
    uint8_t* mask = (uint8_t*)relu_mask;
    const int n = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    SAICV_REQUIRE(C % n == 0, "bn_act_fwd: C=%d must be a multiple of %d", C, n);
    if (dtype == SAICV_DTYPE_BF16) return bn_act_fwd_t<bf16_t>(y, res, z, scale, shift, M, C, relu, mask, st);
    return bn_act_fwd_t<float>(y, res, z, scale, shift, M, C, relu, mask, st);
}

int bn_act_fwd_stats(int dtype, const void* y, const void* res, void* z, const float* sum, const float* sq, int rows,
                     double count, const float* gamma, const float* beta, float* running_mean, float* running_var,
                     double momentum, double eps, long long* num_batches_tracked, float* mean_out, float* invstd_out, size_t M,
                     int C, int relu, void* relu_mask, hipStream_t st) {
    const int n = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    SAICV_REQUIRE(C % n == 0 && C <= kInlineMaxC, "bn_act_fwd_stats: C=%d must be a multiple of %d and <= %d", C, n, kInlineMaxC);
    SAICV_REQUIRE(sum && sq && rows >= 1 && rows <= 64 && mean_out && invstd_out, "bn_act_fwd_stats: statistics rows / outputs missing");
    BnFwdInline inl = {sum, sq, rows, (float)count, gamma, beta, (float)eps, (float)momentum, running_mean, running_var,
                       num_batches_tracked, mean_out, invstd_out};
    if (dtype == SAICV_DTYPE_BF16) return bn_act_fwd_t<bf16_t>(y, res, z, nullptr, nullptr, M, C, relu, (uint8_t*)relu_mask, st, &inl);
    return bn_act_fwd_t<float>(y, res, z, nullptr, nullptr, M, C, relu, (uint8_t*)relu_mask, st, &inl);
}

// the residual join of a block whose shortcut is a convolution + BatchNorm (downsample): that BatchNorm's apply happens in this
// pass (res_scale / res_shift), its output is never written.  sum != NULL: the main branch's statistics arrive as atomically
// accumulated rows (as bn_act_fwd_stats), else scale / shift are given (as bn_act_fwd).
int bn_act_fwd_join(int dtype, const void* y, const void* res, const float* res_scale, const float* res_shift, void* z,
                    const float* scale, const float* shift, const float* sum, const float* sq, int rows, double count,
                    const float* gamma, const float* beta, float* running_mean, float* running_var, double momentum, double eps,
                    long long* num_batches_tracked, float* mean_out, float* invstd_out, size_t M, int C, int relu, void* relu_mask,
                    hipStream_t st) {
    const int n = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    SAICV_REQUIRE(C % n == 0 && C % 4 == 0, "bn_act_fwd_join: C=%d must be a multiple of %d", C, n);
    SAICV_REQUIRE(res && res_scale && res_shift, "bn_act_fwd_join: residual and its coefficients are required");
    if (sum != nullptr) {
        SAICV_REQUIRE(C <= kInlineMaxC && sq && rows >= 1 && rows <= 64 && mean_out && invstd_out, "bn_act_fwd_join: statistics rows / outputs missing (C=%d)", C);
        BnFwdInline inl = {sum, sq, rows, (float)count, gamma, beta, (float)eps, (float)momentum, running_mean, running_var,
                           num_batches_tracked, mean_out, invstd_out, res_scale, res_shift};
        if (dtype == SAICV_DTYPE_BF16) return bn_act_fwd_t<bf16_t>(y, res, z, nullptr, nullptr, M, C, relu, (uint8_t*)relu_mask, st, &inl);
        return bn_act_fwd_t<float>(y, res, z, nullptr, nullptr, M, C, relu, (uint8_t*)relu_mask, st, &inl);
    }
    SAICV_REQUIRE(scale && shift, "bn_act_fwd_join: scale / shift missing");
    if (dtype == SAICV_DTYPE_BF16) return bn_act_fwd_t<bf16_t>(y, res, z, scale, shift, M, C, relu, (uint8_t*)relu_mask, st, nullptr, res_scale, res_shift);
    return bn_act_fwd_t<float>(y, res, z, scale, shift, M, C, relu, (uint8_t*)relu_mask, st, nullptr, res_scale, res_shift);
}

// rows of partials produced by bn_bwd (so the caller can size the workspace)
int bn_bwd_slabs(size_t M, int C, int dtype) {
    const int n = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    const int cpr = C / n;
    const int rpp = cpr >= 256 ? 1 : 256 / cpr;
    static const int cap = getenv("SAICV_BN_SLABS") ? atoi(getenv("SAICV_BN_SLABS")) : 512;     // tuning aid (sweep: 256..2048)
    static const int passes = getenv("SAICV_BN_PASSES") ? atoi(getenv("SAICV_BN_PASSES")) : 16;
    // >= 16 passes per slab and at most two blocks per CU: the [slabs][C] partials of a 7 x 7 x 2048 layer were a
    // third of its tensor traffic at 1024 slabs
    size_t s = M / ((size_t)rpp * passes);
    if (s > (size_t)cap) s = cap;
    if (s < 1) s = 1;
    return (int)s;
}

// workspace: [2][slabs][C] partials + [2][32][C] second stage + 3*C coefficients
size_t bn_bwd_ws_floats(size_t M, int C, int dtype) {
    return (size_t)2 * bn_bwd_slabs(M, C, dtype) * C + (size_t)64 * C + (size_t)3 * C;
}

template <typename T>
static int bn_bwd_t(const void* dz, const void* z, const uint8_t* mask, const void* y, const float* gamma,
                    const float* mean, const float* invstd, void* dy, void* dres, float* dgamma,
                    float* dbeta, size_t M, int C, int relu, int accumulate, float* ws, hipStream_t st,
                    const float* ext_g = nullptr, const float* ext_gx = nullptr, int ext_rows = 0) {
    constexpr int N = Chunk<T>::N;
    const int slabs = bn_bwd_slabs(M, C, sizeof(T) == 2 ? SAICV_DTYPE_BF16 : SAICV_DTYPE_F32);
    const int rows_per = (int)((M + slabs - 1) / slabs);
    const int used = (int)((M + rows_per - 1) / rows_per);
    float* pg = ws;
    float* pgx = ws + (size_t)slabs * C;
    float* ws2 = ws + (size_t)2 * slabs * C;
    float* coef = ws2 + (size_t)64 * C;
    const T* dzz = (const T*)dz; const T* zz = (const T*)z; const T* yy = (const T*)y;
    const float* a = pg; const float* b = pgx;
    int P0 = used;
    if (ext_g != nullptr) {          // the partial sums came out of the producing data-gradient's epilogue: no reduce pass
        a = ext_g; b = ext_gx; P0 = ext_rows;
    } else if (relu)
        hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, true>), dim3(used), dim3(256), 0, st, dzz, zz, mask, yy, mean, invstd, (int)M, C, rows_per, pg, pgx);
    else
        hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, false>), dim3(used), dim3(256), 0, st, dzz, zz, mask, yy, mean, invstd, (int)M, C, rows_per, pg, pgx);
    const int P = reduce_partials(a, b, P0, C, ws2, st);
    hipLaunchKernelGGL(P <= 32 ? bn_finalize_bwd_kernel<4> : bn_finalize_bwd_kernel<16>, dim3((C + 63) / 64), dim3(P <= 32 ? 256 : 1024), 0, st, a, b, P, C,
                       (float)M, gamma, mean, invstd, dgamma, dbeta, coef, coef + C, coef + 2 * C, accumulate);
    const size_t nchunks = M * (size_t)C / N;
    const int grid = stream_grid(nchunks);
    T* dyy = (T*)dy; T* drr = (T*)dres;
    const bool hoist = ((size_t)grid * 256) % (size_t)(C / N) == 0;
    const BnBwdInline none = {};
#define LAUNCH(R, S) do { if (hoist) hipLaunchKernelGGL((bn_bwd_apply_kernel<T, R, S, true, false>), dim3(grid), dim3(256), 0, st, dzz, zz, yy, coef, coef + C, coef + 2 * C, dyy, drr, nchunks, C, mask, none); \
                          else hipLaunchKernelGGL((bn_bwd_apply_kernel<T, R, S, false, false>), dim3(grid), dim3(256), 0, st, dzz, zz, yy, coef, coef + C, coef + 2 * C, dyy, drr, nchunks, C, mask, none); } while (0)
    if (relu) { if (dres) LAUNCH(true, true); else LAUNCH(true, false); }
    else      { if (dres) LAUNCH(false, true); else LAUNCH(false, false); }
#undef LAUNCH
    return check_launch("bn_bwd");
}

template <typename T>
static int bn_bwd_inline_t(const void* dz, const uint8_t* mask, const void* y, void* dy, void* dres, size_t M, int C, int relu,
                           const BnBwdInline& inl, hipStream_t st) {
    constexpr int N = Chunk<T>::N;
    const size_t nchunks = M * (size_t)C / N;
    const int grid = stream_grid(nchunks);
    const bool hoist = ((size_t)grid * 256) % (size_t)(C / N) == 0;
    const T* dzz = (const T*)dz; const T* yy = (const T*)y; T* dyy = (T*)dy; T* drr = (T*)dres;
#define LAUNCH2(R, S, H) hipLaunchKernelGGL((bn_bwd_apply_kernel<T, R, S, H, true>), dim3(grid), dim3(256), 0, st, dzz, (const T*)nullptr, yy, \
                                            (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, dyy, drr, nchunks, C, mask, inl)
#define LAUNCH(R, S) do { if (hoist) LAUNCH2(R, S, true); else LAUNCH2(R, S, false); } while (0)
    if (relu) { if (dres) LAUNCH(true, true); else LAUNCH(true, false); }
    else      { if (dres) LAUNCH(false, true); else LAUNCH(false, false); }
#undef LAUNCH
#undef LAUNCH2
    return check_launch("bn_bwd_inline");
}

// backward with the reduction already done into a few atomically accumulated rows: ONE launch (coefficients in-kernel)
int bn_bwd_inline(int dtype, const void* dz, const void* relu_mask, const void* y, const float* gamma, const float* mean,
                  const float* invstd, const float* part_g, const float* part_gx, int rows, void* dy, void* dres, float* dgamma,
                  float* dbeta, size_t M, int C, int relu, int accumulate, hipStream_t st) {
    const int n = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    SAICV_REQUIRE(C % n == 0 && C <= kInlineMaxC, "bn_bwd_inline: C=%d must be a multiple of %d and <= %d", C, n, kInlineMaxC);
    SAICV_REQUIRE(part_g && part_gx && rows >= 1 && rows <= 64, "bn_bwd_inline: partial sums missing");
    SAICV_REQUIRE(!relu || relu_mask != nullptr, "bn_bwd_inline: relu needs the sign mask");
    const BnBwdInline inl = {part_g, part_gx, rows, (float)M, gamma, mean, invstd, dgamma, dbeta, accumulate};
    if (dtype == SAICV_DTYPE_BF16) return bn_bwd_inline_t<bf16_t>(dz, (const uint8_t*)relu_mask, y, dy, dres, M, C, relu, inl, st);
    return bn_bwd_inline_t<float>(dz, (const uint8_t*)relu_mask, y, dy, dres, M, C, relu, inl, st);
}

int bn_bwd_from_partials(int dtype, const void* dz, const void* relu_mask, const void* y, const float* gamma,
                         const float* mean, const float* invstd, const float* part_g, const float* part_gx, int rows,
                         void* dy, void* dres, float* dgamma, float* dbeta, size_t M, int C, int relu, int accumulate,
                         float* ws, hipStream_t st) {
    const uint8_t* mask = (const uint8_t*)relu_mask;
    const int n = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    SAICV_REQUIRE(C % n == 0, "bn_bwd_from_partials: C=%d must be a multiple of %d", C, n);
    SAICV_REQUIRE(part_g != nullptr && part_gx != nullptr && rows > 0, "bn_bwd_from_partials: partial sums missing");
    SAICV_REQUIRE(!relu || mask != nullptr, "bn_bwd_from_partials: relu needs the sign mask");
    if (dtype == SAICV_DTYPE_BF16)
        return bn_bwd_t<bf16_t>(dz, nullptr, mask, y, gamma, mean, invstd, dy, dres, dgamma, dbeta, M, C, relu, accumulate, ws, st,
                                part_g, part_gx, rows);
    return bn_bwd_t<float>(dz, nullptr, mask, y, gamma, mean, invstd, dy, dres, dgamma, dbeta, M, C, relu, accumulate, ws, st,
                           part_g, part_gx, rows);
}

int bn_bwd(int dtype, const void* dz, const void* z, const void* relu_mask, const void* y, const float* gamma,
           const float* mean, const float* invstd, void* dy, void* dres, float* dgamma, float* dbeta,
           size_t M, int C, int relu, int accumulate, float* ws, hipStream_t st) {
    const uint8_t* mask = (const uint8_t*)relu_mask;
    const int n = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    SAICV_REQUIRE(C % n == 0, "bn_bwd: C=%d must be a multiple of %d", C, n);
    SAICV_REQUIRE(!relu || z != nullptr || mask != nullptr, "bn_bwd: relu needs the forward output z or its sign mask");
    if (dtype == SAICV_DTYPE_BF16)
        return bn_bwd_t<bf16_t>(dz, z, mask, y, gamma, mean, invstd, dy, dres, dgamma, dbeta, M, C, relu, accumulate, ws, st);
    return bn_bwd_t<float>(dz, z, mask, y, gamma, mean, invstd, dy, dres, dgamma, dbeta, M, C, relu, accumulate, ws, st);
}

}  // namespace saicv
