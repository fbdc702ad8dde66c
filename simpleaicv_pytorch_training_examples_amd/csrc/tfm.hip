// Transformer-block kernels for gfx950: LayerNorm, GELU, fused multi-head attention.
//
// Replaces the ATen dispatches of reference SimpleAICV/classification/backbones/vit.py:
//   native_layer_norm (+backward)            TransformerEncoderLayer.norm1/norm2 :147,:151, ViT.norm :225
//   gelu (+backward)                         FeedForward.gelu :87-99
//   bmm, mul, _softmax, bmm (+backwards)     MultiHeadAttention.forward :61-80
// LayerNorm / GELU are HBM-bound streaming kernels (wavefront reductions, 16-byte accesses).
// Attention never materialises the [N,N] matrix in HBM: one workgroup owns one (batch, head),
// keeps K and V (backward: two operands per phase) in LDS and walks 16-query tiles per
// wavefront with MFMA 16x16x32 bf16 / 16x16x4 f32.  It reads q,k,v straight out of the packed
// qkv GEMM output [B*N, 3*C] and writes the head-merged [B*N, C] layout, so none of the
// reference's permute / contiguous copies exist.  Sequence length N <= 256 (ViT-B: 197; SAM
// windows: 196); the 4096-token global attention of SAM needs the streaming variant (next).
#include <stdlib.h>
#include "common.h"
#include "saicv_internal.h"

// cache policy of the LayerNorm kernels' loads (library variant for A/B runs: -DSAICV_LN_LD_NT = streaming loads)
#ifdef SAICV_LN_LD_NT
#define LN_LD ld_chunk_nt
#else
#define LN_LD ld_chunk
#endif

namespace {

// ======================================================================== LayerNorm
// one wavefront per row; a lane holds NCH chunks (columns lane, lane+64, ...) in registers
// The keep decision of the fused residual dropout (reference detection/models/detr.py:89,92,114,118,122: norm(x + dropout(branch))): a
// counter-based hash of (seed, row, column), the same in the forward and the backward kernel.  keep iff hash >= p * 2^32.
DEVINL bool ln_keep(unsigned seed, unsigned row, unsigned col, unsigned thresh) {
    unsigned h = seed ^ (row * 0x9E3779B1u) ^ (col * 0x85EBCA77u);
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    return h >= thresh;
}
struct LnDrop {                 // branch == nullptr: a plain LayerNorm
    const void* branch;         // [M][C], added to x behind the dropout
    void* sum_out;              // x + dropout(branch): what the LayerNorm normalises, kept for the backward
    unsigned seed;
    const unsigned* seed_device;
    unsigned thresh;
    float inv_keep;
};

template <typename T, int NCH>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* __restrict__ x,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            T* __restrict__ y, float* __restrict__ mean,
                                                            float* __restrict__ rstd, int M, int C, float eps, const LnDrop dr) {
    constexpr int N = Chunk<T>::N;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    const int cpr = C / N;
    const unsigned dseed = dr.branch != nullptr && dr.seed_device != nullptr ? dr.seed + *dr.seed_device : dr.seed;
    float v[NCH][N];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = lane + 64 * j;
        if (c < cpr) {
            Chunk<T>::unpack(LN_LD(x + (size_t)row * C + c * N), v[j]);
            if (dr.branch != nullptr) {         // s = x + dropout(branch), rounded to T as a separate add kernel would store it
                float b[N];
                Chunk<T>::unpack(LN_LD(reinterpret_cast<const T*>(dr.branch) + (size_t)row * C + c * N), b);
#pragma unroll
                for (int k = 0; k < N; ++k) v[j][k] += ln_keep(dseed, (unsigned)row, (unsigned)(c * N + k), dr.thresh) ? b[k] * dr.inv_keep : 0.f;
                const auto packed = Chunk<T>::pack(v[j]);
                st_chunk(reinterpret_cast<T*>(dr.sum_out) + (size_t)row * C + c * N, packed);
                Chunk<T>::unpack(packed, v[j]);
            }
#pragma unroll
            for (int k = 0; k < N; ++k) s += v[j][k];
        } else {
#pragma unroll
            for (int k = 0; k < N; ++k) v[j][k] = 0.f;
        }
    }
    const float mu = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        if (lane + 64 * j < cpr) {
#pragma unroll
            for (int k = 0; k < N; ++k) { const float d = v[j][k] - mu; q += d * d; }
        }
    }
    const float rs = rsqrtf(wave_sum(q) / (float)C + eps);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = lane + 64 * j;
        if (c < cpr) {
            float o[N];
#pragma unroll
            for (int k = 0; k < N; ++k) o[k] = (v[j][k] - mu) * rs * gamma[c * N + k] + beta[c * N + k];
            st_chunk(y + (size_t)row * C + c * N, Chunk<T>::pack(o));
        }
    }
}

// backward: dx per row; per-block partial dgamma / dbeta (a wavefront walks rows_per rows)
constexpr int LNB_WAVES = 8;       // wavefronts per block of the backward kernel

template <typename T, int NCH>
__global__ __launch_bounds__(64 * LNB_WAVES) void layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd,
                                                            const T* __restrict__ addend, T* __restrict__ dx,
                                                            float* __restrict__ part_g, float* __restrict__ part_b,
                                                            int M, int C, int rows_per, const float* __restrict__ out_scale,
                                                            int rows_per_scale, T* __restrict__ dxs, const LnDrop dr) {
    constexpr int N = Chunk<T>::N;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int cpr = C / N;
    const unsigned dseed = dr.branch != nullptr && dr.seed_device != nullptr ? dr.seed + *dr.seed_device : dr.seed;
    float gam[NCH][N], ag[NCH][N], ab[NCH][N];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = lane + 64 * j;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            gam[j][k] = c < cpr ? gamma[c * N + k] : 0.f;
            ag[j][k] = 0.f;
            ab[j][k] = 0.f;
        }
    }
    const int r0 = blockIdx.x * rows_per * LNB_WAVES;
    // software pipeline: the 16-byte chunks of row i+1 are in flight while row i is reduced and written
    u32x4 nd[NCH], nx[NCH], na[NCH];
    float nmu = 0.f, nrs = 0.f;
    auto fetch = [&](int row) {
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = lane + 64 * j;
            if (c < cpr) {
                nd[j] = LN_LD(dy + (size_t)row * C + c * N);
                nx[j] = LN_LD(x + (size_t)row * C + c * N);
                if (addend != nullptr) na[j] = LN_LD(addend + (size_t)row * C + c * N);
            }
        }
        nmu = mean[row];
        nrs = rstd[row];
    };
    if (r0 + wave < M) fetch(r0 + wave);
    for (int i = 0; i < rows_per; ++i) {
        const int row = r0 + i * LNB_WAVES + wave;
        if (row >= M) break;
        const float mu = nmu, rs = nrs;
        u32x4 cd[NCH], cx[NCH], ca[NCH];
#pragma unroll
        for (int j = 0; j < NCH; ++j) { cd[j] = nd[j]; cx[j] = nx[j]; ca[j] = na[j]; }
        if (i + 1 < rows_per && row + LNB_WAVES < M) fetch(row + LNB_WAVES);
        float g[NCH][N], xh[NCH][N];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = lane + 64 * j;
            if (c < cpr) {
                float d[N], xv[N];
                Chunk<T>::unpack(cd[j], d);
                Chunk<T>::unpack(cx[j], xv);
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    xh[j][k] = (xv[k] - mu) * rs;
                    g[j][k] = d[k] * gam[j][k];
                    s1 += g[j][k];
                    s2 += g[j][k] * xh[j][k];
                    ag[j][k] += d[k] * xh[j][k];
                    ab[j][k] += d[k];
                }
            }
        }
        s1 = wave_sum(s1) / (float)C;
        s2 = wave_sum(s2) / (float)C;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = lane + 64 * j;
            if (c < cpr) {
                float o[N];
#pragma unroll
                for (int k = 0; k < N; ++k) o[k] = rs * (g[j][k] - s1 - xh[j][k] * s2);
                if (addend != nullptr) {        // residual-stream gradient joins here (pre-LN blocks)
                    float a[N];
                    Chunk<T>::unpack(ca[j], a);
#pragma unroll
                    for (int k = 0; k < N; ++k) o[k] += a[k];
                }
                const auto packed = Chunk<T>::pack(o);
                st_chunk(dx + (size_t)row * C + c * N, packed);
                if (dxs != nullptr) {           // the drop-path twin: row factor x the STORED (rounded) gradient, as a separate row_scale pass would give
                    float r[N];
                    Chunk<T>::unpack(packed, r);
                    if (dr.branch != nullptr) {  // ... or the gradient of the branch behind the fused residual dropout: mask / keep, element by element
#pragma unroll
                        for (int k = 0; k < N; ++k) r[k] = ln_keep(dseed, (unsigned)row, (unsigned)(c * N + k), dr.thresh) ? r[k] * dr.inv_keep : 0.f;
                    } else {
                        const float sc = out_scale[row / rows_per_scale];
#pragma unroll
                        for (int k = 0; k < N; ++k) r[k] *= sc;
                    }
                    st_chunk(dxs + (size_t)row * C + c * N, Chunk<T>::pack(r));
                }
            }
        }
    }
    // combine the wavefronts of the block through LDS, one accumulator set at a time
    __shared__ float red[LNB_WAVES][64 * 8];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = lane + 64 * j;
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int k = 0; k < N; ++k) red[wave][lane * N + k] = pass == 0 ? ag[j][k] : ab[j][k];
            __syncthreads();
            if (wave == 0 && c < cpr) {
                float* dst = (pass == 0 ? part_g : part_b) + (size_t)blockIdx.x * C + c * N;
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    float v = 0.f;
#pragma unroll
                    for (int w = 0; w < LNB_WAVES; ++w) v += red[w][lane * N + k];
                    dst[k] = v;
                }
            }
            __syncthreads();
        }
    }
}

// ---- two rows per wavefront, 32 lanes each (r05 default for rows of 96 chunks; SAICV_LN_HALF=0 selects the one-row kernels)
// C = 768 in bf16 is 96 chunks of 16 bytes: one row per wavefront leaves 32 of 128 chunk slots empty (NCH = 2), and the SQ counters
// of the backward kernel say it is bound by instruction issue, not by bandwidth (profiles/r04_layernorm_sq_counters.json) -- so idle
// lanes cost time.  With 32 lanes per row and NCH = 3 every lane works, and the row reductions lose their last cross-half step.
DEVINL float half32_sum(float v) {          // sum over the 32 lanes of this lane's half of the wavefront
    v = row16_sum(v);
    v += __shfl_xor(v, 16, 64);
    return v;
}

template <typename T, int NCH>
__global__ __launch_bounds__(256) void layernorm_fwd_half_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, T* __restrict__ y,
                                                                 float* __restrict__ mean, float* __restrict__ rstd, int M, int C, float eps) {
    constexpr int N = Chunk<T>::N;
    const int lane = threadIdx.x & 63, sl = lane & 31;
    const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + (lane >> 5);
    const bool live = row < M;                       // (the other half of the wavefront may still hold a row: no early return)
    const int cpr = C / N;                           // == 32 * NCH (launcher)
    float v[NCH][N];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = sl + 32 * j;
        if (live) Chunk<T>::unpack(LN_LD(x + (size_t)row * C + c * N), v[j]);
        else {
#pragma unroll
            for (int k = 0; k < N; ++k) v[j][k] = 0.f;
        }
#pragma unroll
        for (int k = 0; k < N; ++k) s += v[j][k];
    }
    const float mu = half32_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
        for (int k = 0; k < N; ++k) { const float d = v[j][k] - mu; q += d * d; }
    const float rs = rsqrtf(half32_sum(q) / (float)C + eps);
    if (!live) return;
    if (sl == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = sl + 32 * j;
        float o[N];
#pragma unroll
        for (int k = 0; k < N; ++k) o[k] = (v[j][k] - mu) * rs * gamma[c * N + k] + beta[c * N + k];
        st_chunk(y + (size_t)row * C + c * N, Chunk<T>::pack(o));
    }
    (void)cpr;
}

template <typename T, int NCH>
__global__ __launch_bounds__(64 * LNB_WAVES) void layernorm_bwd_half_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                                 const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, const T* __restrict__ addend,
                                                                 T* __restrict__ dx, float* __restrict__ part_g,
                                                                 float* __restrict__ part_b, int M, int C, int rows_per,
                                                                 const float* __restrict__ out_scale, int rows_per_scale, T* __restrict__ dxs,
                                                                 const LnDrop dr) {
    const unsigned dseed = dr.branch != nullptr && dr.seed_device != nullptr ? dr.seed + *dr.seed_device : dr.seed;
    // rows_per counts ROW PAIRS per wavefront here: a block covers rows_per * LNB_WAVES * 2 rows
    constexpr int N = Chunk<T>::N;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, sl = lane & 31, sub = lane >> 5;
    float gam[NCH][N], ag[NCH][N], ab[NCH][N];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = sl + 32 * j;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            gam[j][k] = gamma[c * N + k];
            ag[j][k] = 0.f;
            ab[j][k] = 0.f;
        }
    }
    const int r0 = blockIdx.x * rows_per * LNB_WAVES * 2;
    u32x4 nd[NCH], nx[NCH], na[NCH];
    float nmu = 0.f, nrs = 0.f;
    auto fetch = [&](int row) {
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = sl + 32 * j;
            nd[j] = LN_LD(dy + (size_t)row * C + c * N);
            nx[j] = LN_LD(x + (size_t)row * C + c * N);
            if (addend != nullptr) na[j] = LN_LD(addend + (size_t)row * C + c * N);
        }
        nmu = mean[row];
        nrs = rstd[row];
    };
    auto zero_next = [&]() {                   // a half without a row contributes zeros (d = 0 => nothing reaches the sums)
#pragma unroll
        for (int j = 0; j < NCH; ++j) { nd[j] = u32x4{0u, 0u, 0u, 0u}; nx[j] = u32x4{0u, 0u, 0u, 0u}; na[j] = u32x4{0u, 0u, 0u, 0u}; }
        nmu = 0.f;
        nrs = 0.f;
    };
    const int first = r0 + wave * 2 + sub;
    if (first < M) fetch(first); else zero_next();
    for (int i = 0; i < rows_per; ++i) {
        const int pair = r0 + (i * LNB_WAVES + wave) * 2;         // wave-uniform
        if (pair >= M) break;
        const int row = pair + sub;
        const bool live = row < M;
        const float mu = nmu, rs = nrs;
        u32x4 cd[NCH], cx[NCH], ca[NCH];
#pragma unroll
        for (int j = 0; j < NCH; ++j) { cd[j] = nd[j]; cx[j] = nx[j]; ca[j] = na[j]; }
        if (i + 1 < rows_per) {
            const int nrow = row + LNB_WAVES * 2;
            if (nrow < M) fetch(nrow); else zero_next();
        }
        float g[NCH][N], xh[NCH][N];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            float d[N], xv[N];
            Chunk<T>::unpack(cd[j], d);
            Chunk<T>::unpack(cx[j], xv);
#pragma unroll
            for (int k = 0; k < N; ++k) {
                xh[j][k] = (xv[k] - mu) * rs;
                g[j][k] = d[k] * gam[j][k];
                s1 += g[j][k];
                s2 += g[j][k] * xh[j][k];
                ag[j][k] += d[k] * xh[j][k];
                ab[j][k] += d[k];
            }
        }
        s1 = half32_sum(s1) / (float)C;
        s2 = half32_sum(s2) / (float)C;
        if (live) {
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int c = sl + 32 * j;
                float o[N];
#pragma unroll
                for (int k = 0; k < N; ++k) o[k] = rs * (g[j][k] - s1 - xh[j][k] * s2);
                if (addend != nullptr) {
                    float a[N];
                    Chunk<T>::unpack(ca[j], a);
#pragma unroll
                    for (int k = 0; k < N; ++k) o[k] += a[k];
                }
                const auto packed = Chunk<T>::pack(o);
                st_chunk(dx + (size_t)row * C + c * N, packed);
                if (dxs != nullptr) {           // the drop-path twin: row factor x the STORED (rounded) gradient, as a separate row_scale pass would give
                    float r[N];
                    Chunk<T>::unpack(packed, r);
                    if (dr.branch != nullptr) {  // ... or the gradient of the branch behind the fused residual dropout: mask / keep, element by element
#pragma unroll
                        for (int k = 0; k < N; ++k) r[k] = ln_keep(dseed, (unsigned)row, (unsigned)(c * N + k), dr.thresh) ? r[k] * dr.inv_keep : 0.f;
                    } else {
                        const float sc = out_scale[row / rows_per_scale];
#pragma unroll
                        for (int k = 0; k < N; ++k) r[k] *= sc;
                    }
                    st_chunk(dxs + (size_t)row * C + c * N, Chunk<T>::pack(r));
                }
            }
        }
    }
    // combine the two halves of every wavefront and the wavefronts of the block through LDS, one accumulator set at a time
    __shared__ float red[LNB_WAVES][64 * 8];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = sl + 32 * j;
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int k = 0; k < N; ++k) red[wave][lane * N + k] = pass == 0 ? ag[j][k] : ab[j][k];
            __syncthreads();
            if (wave == 0 && sub == 0) {
                float* dst = (pass == 0 ? part_g : part_b) + (size_t)blockIdx.x * C + c * N;
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    float v = 0.f;
#pragma unroll
                    for (int w = 0; w < LNB_WAVES; ++w) v += red[w][sl * N + k] + red[w][(sl + 32) * N + k];
                    dst[k] = v;
                }
            }
            __syncthreads();
        }
    }
}

// out[c] (=|+=) sum_p part[p][c]; blockIdx.y selects one of two (partials, output) pairs, so dgamma and dbeta of a LayerNorm
// backward leave in ONE launch (ViT-B: 25 launches of ~7 us per step less)
__global__ __launch_bounds__(1024) void colreduce_kernel(const float* __restrict__ part0, const float* __restrict__ part1, int P, int C,
                                                         float* __restrict__ out0, float* __restrict__ out1, int accumulate) {
    const float* __restrict__ part = blockIdx.y ? part1 : part0;
    float* __restrict__ out = blockIdx.y ? out1 : out0;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int ty = threadIdx.x >> 6;              // 16 row lanes
    float s = 0.f;
    if (c < C)
        for (int p = ty; p < P; p += 16) s += part[(size_t)p * C + c];
    __shared__ float l[16][64];
    l[ty][threadIdx.x & 63] = s;
    __syncthreads();
    if (ty == 0 && c < C) {
        const int t = threadIdx.x;
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) v += l[k][t];
        out[c] = accumulate ? out[c] + v : v;
    }
}

// ======================================================================== GELU (exact erf form)
template <typename T>
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, size_t nchunks) {
    constexpr int N = Chunk<T>::N;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += stride) {
        float v[N];
        Chunk<T>::unpack(ld_chunk(x + i * N), v);
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = gelu_fwd_f(v[k]);
        st_chunk(y + i * N, Chunk<T>::pack(v));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                       T* __restrict__ dx, size_t nchunks) {
    constexpr int N = Chunk<T>::N;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += stride) {
        float v[N], g[N];
        Chunk<T>::unpack(ld_chunk(x + i * N), v);
        Chunk<T>::unpack(ld_chunk(dy + i * N), g);
#pragma unroll
        for (int k = 0; k < N; ++k) g[k] *= gelu_grad_f(v[k]);
        st_chunk(dx + i * N, Chunk<T>::pack(g));
    }
}

// ======================================================================== dropout(relu(x)) in one pass
// DETR's feed-forward `linear2(dropout(activation(linear1(x))))` (reference detection/models/detr.py:90-91, 120-121) spent a clamp, a dropout,
// a threshold-backward and a masked-scale launch per layer on the [B * L, 4 C] hidden tensor.  y = keep ? relu(x) / (1 - p) : 0 with the
// counter-based keep decision of the fused residual dropout (chunk index, lane in chunk); the backward needs no mask: y > 0 exactly where
// the gradient passes (relu open AND kept), dx = dy / (1 - p) there.
template <typename T>
__global__ __launch_bounds__(256) void relu_dropout_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, size_t nchunks, unsigned seed,
                                                               const unsigned* __restrict__ seed_device, unsigned thresh, float inv_keep) {
    constexpr int N = Chunk<T>::N;
    const unsigned dseed = seed_device ? seed + *seed_device : seed;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += stride) {
        float v[N];
        Chunk<T>::unpack(ld_chunk(x + i * N), v);
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = (v[k] > 0.f && ln_keep(dseed, (unsigned)i, (unsigned)k, thresh)) ? v[k] * inv_keep : 0.f;
        st_chunk(y + i * N, Chunk<T>::pack(v));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void relu_dropout_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dx,
                                                               size_t nchunks, float inv_keep) {
    constexpr int N = Chunk<T>::N;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += stride) {
        float v[N], g[N];
        Chunk<T>::unpack(ld_chunk(y + i * N), v);
        Chunk<T>::unpack(ld_chunk(dy + i * N), g);
#pragma unroll
        for (int k = 0; k < N; ++k) g[k] = v[k] > 0.f ? g[k] * inv_keep : 0.f;
        st_chunk(dx + i * N, Chunk<T>::pack(g));
    }
}

// ======================================================================== attention
// LDS image of a [rows][64] operand: 16-byte chunks, chunk c of row r at slot c ^ swz(r), which
// keeps both the row-fragment reads (ds_read_b128, 16 rows x one chunk) and the transposing
// reads (ds_read_b64_tr_b16, 4 rows x 32 bytes) conflict-light.
template <typename T> struct AttnLds;
template <> struct AttnLds<bf16_t> {
    static constexpr int ROWB = 128;                        // 64 d * 2 B
    static constexpr int DCH = 8;                           // chunks per row
    static DEVINL int off(int row, int chunk) { return row * ROWB + (((chunk ^ (row >> 1)) & 7) << 4); }
};
template <> struct AttnLds<float> {
    static constexpr int ROWB = 256;
    static constexpr int DCH = 16;
    static DEVINL int off(int row, int chunk) { return row * ROWB + (((chunk ^ row) & 15) << 4); }
};

// cooperative copy of `rows` x 64 elements (row stride `rs` elements in HBM) into the LDS image;
// rows >= nvalid are zero filled
template <typename T>
DEVINL void attn_stage(char* lds, const T* __restrict__ g, int rs, int nvalid, int rows) {
    constexpr int EPC = ElemTraits<T>::EPC;
    constexpr int DCH = AttnLds<T>::DCH;
    for (int i = threadIdx.x; i < rows * DCH; i += blockDim.x) {
        const int r = i / DCH, c = i - r * DCH;
        const u32x4 v = r < nvalid ? ld_chunk(g + (size_t)r * rs + c * EPC) : zero_chunk();
        st_chunk(lds + AttnLds<T>::off(r, c), v);
    }
}

// B-operand fragment for D = A(16 x k) * B(k x 16) where B[k][col] = M[kbase + kperm][cbase + col]
// and M is an LDS image with rows = k.  bf16: two transposing reads (k = 8 rows per lane group,
// taken as rows g*4..g*4+3 of the first 16-row tile and of the second); f32: four plain reads.
template <typename T> struct TrFrag;
template <> struct TrFrag<bf16_t> {
    // rows kbase + g*4 + j (j<4) and kbase + 16 + g*4 + j
    static DEVINL u32x4 load(const char* lds, int kbase, int cbase, int l15, int lg) {
        typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
        const int col = cbase + (l15 & 3) * 4;
        const int r0 = kbase + lg * 4 + (l15 >> 2);
        const int r1 = r0 + 16;
        const char* q0 = lds + r0 * 128 + ((((col >> 3) ^ (r0 >> 1)) & 7) << 4) + ((col & 4) << 1);
        const char* q1 = lds + r1 * 128 + ((((col >> 3) ^ (r1 >> 1)) & 7) << 4) + ((col & 4) << 1);
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)q0);
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)q1);
        const u32x2 a = __builtin_bit_cast(u32x2, lo), b = __builtin_bit_cast(u32x2, hi);
        return u32x4{a[0], a[1], b[0], b[1]};
    }
};

// pack probabilities of two adjacent 16-key tiles (fragment rows g*4+r) into a bf16 A operand
DEVINL u32x4 pack_p(const f32x4& t0, const f32x4& t1) {
    float f[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
    return Chunk<bf16_t>::pack(f);
}

// O(16 x 64) += P(16 x 32 keys) * M(32 keys x 64): P given as two adjacent C-layout tiles t0, t1
// (lane: row-of-A = l15, k = kbase + {0,16} + g*4 + r) and M an LDS image with rows = keys.
template <typename T>
DEVINL void pv_pair(f32x4 (&o)[4], const f32x4& t0, const f32x4& t1, const char* lds, int kbase, int l15, int lg) {
    if constexpr (sizeof(T) == 2) {
        const u32x4 pa = pack_p(t0, t1);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const u32x4 vb = TrFrag<bf16_t>::load(lds, kbase, dt * 16, l15, lg);
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, pa),
                                                           __builtin_bit_cast(bf16x8, vb), o[dt], 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = kbase + half * 16 + lg * 4 + r;
                const float a = half == 0 ? t0[r] : t1[r];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const int d = dt * 16 + l15;
                    const float b = *reinterpret_cast<const float*>(lds + AttnLds<float>::off(row, d >> 2) + (d & 3) * 4);
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, o[dt], 0, 0, 0);
                }
            }
        }
    }
}

// row fragments (A or B operand with k = d) of 16 rows starting at `row0` of an LDS image
template <typename T>
DEVINL void lds_row_frags(u32x4 (&f)[4], const char* lds, int row0, int l15, int lg) {
    constexpr int STEPS = AttnLds<T>::DCH / 4;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) f[s] = ld_chunk(lds + AttnLds<T>::off(row0 + l15, s * 4 + lg));
}
// the same straight from HBM (row stride rs elements); rows >= nvalid give zeros
template <typename T>
DEVINL void gmem_row_frags(u32x4 (&f)[4], const T* __restrict__ g, int rs, int row0, int nvalid, int l15, int lg) {
    constexpr int EPC = ElemTraits<T>::EPC;
    constexpr int STEPS = AttnLds<T>::DCH / 4;
    const int r = row0 + l15;
#pragma unroll
    for (int s = 0; s < STEPS; ++s)
        f[s] = r < nvalid ? ld_chunk(g + (size_t)r * rs + (s * 4 + lg) * EPC) : zero_chunk();
}
template <typename T>
DEVINL f32x4 tile_mma(const u32x4 (&a)[4], const u32x4 (&b)[4]) {
    constexpr int STEPS = AttnLds<T>::DCH / 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < STEPS; ++s) Mma<T>::run(acc, a[s], b[s]);
    return acc;
}

constexpr int MAXKT = 16;          // up to 256 keys
constexpr int ATT_THREADS = 512;   // 8 wavefronts: 2 per SIMD hide each other's LDS / HBM latency

// ------------------------------------------------------------------------ forward
template <typename T>
__global__ __launch_bounds__(ATT_THREADS) void attention_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ out,
                                                                    float* __restrict__ lse, int B, int N, int H,
                                                                    float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int bh = blockIdx.x;
    const int b = bh / H, h = bh - b * H;
    const int C = H * 64, RS = 3 * C;
    const int Np = (N + 31) & ~31;
    const int nkt = Np / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    char* Ks = smem;
    char* Vs = smem + Np * AttnLds<T>::ROWB;
    const T* base = qkv + (size_t)b * N * RS + h * 64;
    attn_stage<T>(Ks, base + C, RS, N, Np);
    attn_stage<T>(Vs, base + 2 * C, RS, N, Np);
    __syncthreads();
    const int nqt = (N + 15) / 16;
    for (int qt = wave; qt < nqt; qt += ATT_THREADS / 64) {
        u32x4 qf[4];
        gmem_row_frags<T>(qf, base, RS, qt * 16, N, l15, lg);
        // S^T tiles: rows = keys (kt*16 + lg*4 + r), col = query l15
        f32x4 st[MAXKT];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < MAXKT; ++kt) {
            if (kt < nkt) {
                u32x4 kf[4];
                lds_row_frags<T>(kf, Ks, kt * 16, l15, lg);
                st[kt] = tile_mma<T>(kf, qf);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt * 16 + lg * 4 + r;
                    st[kt][r] = key < N ? st[kt][r] * (scale * 1.4426950408889634f) : -INFINITY;        // log2 domain: v_exp_f32 below
                    mx = fmaxf(mx, st[kt][r]);
                }
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int blk = 0; blk < MAXKT / 2; ++blk) {
            if (blk * 2 < nkt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    st[2 * blk][r] = __builtin_amdgcn_exp2f(st[2 * blk][r] - mx);
                    st[2 * blk + 1][r] = __builtin_amdgcn_exp2f(st[2 * blk + 1][r] - mx);
                    sum += st[2 * blk][r] + st[2 * blk + 1][r];
                }
                pv_pair<T>(o, st[2 * blk], st[2 * blk + 1], Vs, blk * 32, l15, lg);
            }
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        if (lg == 0 && qt * 16 + l15 < N) lse[((size_t)b * H + h) * N + qt * 16 + l15] = (mx + log2f(sum)) * 0.6931471805599453f;
        // O tile: col d = dt*16 + l15, rows = queries lg*4 + r; 1/sum lives on lanes with l15 == query
        const float inv = 1.f / sum;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float iq = __shfl(inv, lg * 4 + r, 64);
            const int q = qt * 16 + lg * 4 + r;
            if (q < N) {
                T* dst = out + ((size_t)b * N + q) * C + h * 64 + l15;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) dst[dt * 16] = from_f32<T>(o[dt][r] * iq);
            }
        }
    }
}

// ------------------------------------------------------------------------ backward
// phase 0: D[q] = sum_d dO[q][d] * O[q][d] -> LDS;  LSE -> LDS
// phase A: LDS = {K, V}:  per 16-query tile  dQ  = scale * dS K        (dS in "S^T" layout)
// phase B: LDS = {Q, dO}: per 16-key tile    dV  = P^T dO ,  dK = scale * dS^T Q
// Both phases stream over pairs of 16-row tiles and feed the MFMAs at once (the log-sum-exp is
// known), so only two probability tiles are live per wavefront.
template <typename T>
__global__ __launch_bounds__(ATT_THREADS) void attention_bwd_kernel(const T* __restrict__ qkv, const T* __restrict__ out,
                                                                    const T* __restrict__ dout, const float* __restrict__ lse,
                                                                    T* __restrict__ dqkv, int B, int N, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int bh = blockIdx.x;
    const int b = bh / H, h = bh - b * H;
    const int C = H * 64, RS = 3 * C;
    const int Np = (N + 31) & ~31;
    const int nblk = Np / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    char* M0 = smem;
    char* M1 = smem + Np * AttnLds<T>::ROWB;
    float* Dq = reinterpret_cast<float*>(smem + 2 * Np * AttnLds<T>::ROWB);
    float* Ls = Dq + Np;
    const T* qb = qkv + (size_t)b * N * RS + h * 64;
    const T* ob = out + (size_t)b * N * C + h * 64;
    const T* dob = dout + (size_t)b * N * C + h * 64;
    T* dqb = dqkv + (size_t)b * N * RS + h * 64;
    constexpr int EPC = ElemTraits<T>::EPC;
    constexpr int DCH = AttnLds<T>::DCH;
    constexpr int NW = ATT_THREADS / 64;

    // ---- phase 0: two threads per query row (half of the d range each)
    for (int i = threadIdx.x; i < Np * 2; i += ATT_THREADS) {
        const int q = i >> 1, hf = i & 1;
        float d = 0.f;
        if (q < N) {
#pragma unroll
            for (int c = 0; c < DCH / 2; ++c) {
                float a[Chunk<T>::N], o[Chunk<T>::N];
                Chunk<T>::unpack(ld_chunk(dob + (size_t)q * C + (hf * (DCH / 2) + c) * EPC), a);
                Chunk<T>::unpack(ld_chunk(ob + (size_t)q * C + (hf * (DCH / 2) + c) * EPC), o);
#pragma unroll
                for (int k = 0; k < Chunk<T>::N; ++k) d += a[k] * o[k];
            }
        }
        d += __shfl_xor(d, 1, 64);
        if (hf == 0) {
            Dq[q] = d;
            Ls[q] = q < N ? lse[((size_t)b * H + h) * N + q] * 1.4426950408889634f : 0.f;      // log2 units
        }
    }
    attn_stage<T>(M0, qb + C, RS, N, Np);          // K
    attn_stage<T>(M1, qb + 2 * C, RS, N, Np);      // V
    __syncthreads();

    // ---- phase A: dQ
    const int nqt = (N + 15) / 16;
    for (int qt = wave; qt < nqt; qt += NW) {
        u32x4 qf[4], dof[4];
        gmem_row_frags<T>(qf, qb, RS, qt * 16, N, l15, lg);
        gmem_row_frags<T>(dof, dob, C, qt * 16, N, l15, lg);
        const int q = qt * 16 + l15;
        const float lq = Ls[q], dq_ = Dq[q];
        const bool qok = q < N;
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int blk = 0; blk < nblk; ++blk) {
            f32x4 ds[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int kt = blk * 2 + t;
                u32x4 kf[4], vf[4];
                lds_row_frags<T>(kf, M0, kt * 16, l15, lg);
                lds_row_frags<T>(vf, M1, kt * 16, l15, lg);
                const f32x4 sv = tile_mma<T>(kf, qf);      // rows keys, col query
                const f32x4 dp = tile_mma<T>(vf, dof);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt * 16 + lg * 4 + r;
                    const float p = (key < N && qok) ? __builtin_amdgcn_exp2f(sv[r] * (scale * 1.4426950408889634f) - lq) : 0.f;
                    ds[t][r] = p * (dp[r] - dq_) * scale;
                }
            }
            pv_pair<T>(o, ds[0], ds[1], M0, blk * 32, l15, lg);      // dQ += dS * K
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qq = qt * 16 + lg * 4 + r;
            if (qq < N) {
                T* dst = dqb + (size_t)qq * RS + l15;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) dst[dt * 16] = from_f32<T>(o[dt][r]);
            }
        }
    }
    __syncthreads();
    attn_stage<T>(M0, qb, RS, N, Np);              // Q
    attn_stage<T>(M1, dob, C, N, Np);              // dO
    __syncthreads();

    // ---- phase B: dK, dV per key tile; P / dS tiles in the un-swapped layout
    // (col = key l15, rows = queries lg*4 + r), accumulated over all query tiles
    for (int kt = wave; kt < nqt; kt += NW) {
        u32x4 kf[4], vf[4];
        gmem_row_frags<T>(kf, qb + C, RS, kt * 16, N, l15, lg);
        gmem_row_frags<T>(vf, qb + 2 * C, RS, kt * 16, N, l15, lg);
        const bool kok = kt * 16 + l15 < N;
        f32x4 dv[4], dk[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        for (int blk = 0; blk < nblk; ++blk) {
            f32x4 pt[2], dst_[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int qt = blk * 2 + t;
                u32x4 qf[4], dof[4];
                lds_row_frags<T>(qf, M0, qt * 16, l15, lg);
                lds_row_frags<T>(dof, M1, qt * 16, l15, lg);
                const f32x4 sv = tile_mma<T>(qf, kf);      // rows queries, col key
                const f32x4 dp = tile_mma<T>(dof, vf);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qq = qt * 16 + lg * 4 + r;
                    const float p = (qq < N && kok) ? __builtin_amdgcn_exp2f(sv[r] * (scale * 1.4426950408889634f) - Ls[qq]) : 0.f;
                    pt[t][r] = p;
                    dst_[t][r] = p * (dp[r] - Dq[qq]) * scale;
                }
            }
            // the A operand wants row = key; the tiles hold key on l15 and queries on (lg, r): the
            // layout pv_pair expects with "keys" := queries
            pv_pair<T>(dv, pt[0], pt[1], M1, blk * 32, l15, lg);       // dV += P^T dO
            pv_pair<T>(dk, dst_[0], dst_[1], M0, blk * 32, l15, lg);   // dK += dS^T Q
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kk = kt * 16 + lg * 4 + r;
            if (kk < N) {
                T* dkp = dqb + (size_t)kk * RS + C + l15;
                T* dvp = dqb + (size_t)kk * RS + 2 * C + l15;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    dkp[dt * 16] = from_f32<T>(dk[dt][r]);
                    dvp[dt * 16] = from_f32<T>(dv[dt][r]);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------ backward, bf16, two tiles per wavefront
// Same three phases as above with the instruction mix turned around: the one-tile kernel issues ~104 VALU and 16 LDS
// instructions per 12 MFMAs (address swizzles, masks, scalar stores) and runs at the VALU / LDS rate.  Here
//  * a wavefront owns TWO 16-row tiles (32 queries in phase A, 32 keys in phase B) that share every LDS fragment read;
//  * every LDS address is a lane constant plus a block offset (a 32-row block is 4096 bytes and the swizzle only looks
//    at row bits below that);
//  * no masks: padded K / V / Q / dO rows are zero in LDS, so whatever the softmax recomputation produces for them
//    multiplies zeros (their own output rows are never stored);
//  * the 1/sqrt(d) factor is applied once to the finished dQ / dK tiles;
//  * outputs are computed TRANSPOSED (A = operand^T fragment, B = probabilities), so a lane ends up with four consecutive
//    d of one row and stores 8 bytes instead of four scattered 2-byte elements.
// 4 wavefronts per workgroup, two workgroups per CU (LDS), pairs of tiles dealt round-robin to the wavefronts.
// TPW = 16-row tiles per wavefront: 2 (four wavefronts per workgroup, fragment reads shared by the pair) or 1 (eight wavefronts,
// half the registers: four wavefronts per SIMD stay resident)
template <int TPW> struct Att2 { static constexpr int THREADS = TPW == 2 ? 256 : 512; };

DEVINL u32x4 tr_pair(const char* q) {
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)q);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(q + 2048));
    const u32x2 a = __builtin_bit_cast(u32x2, lo), b = __builtin_bit_cast(u32x2, hi);
    return u32x4{a[0], a[1], b[0], b[1]};
}
DEVINL f32x4 mma2(const u32x4 (&a)[2], const u32x4 (&b)[2]) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    Mma<bf16_t>::run(acc, a[0], b[0]);
    Mma<bf16_t>::run(acc, a[1], b[1]);
    return acc;
}
DEVINL void st_bf16x4(bf16_t* dst, const f32x4& v, float mul) {
    bf16x4 pk;
#pragma unroll
    for (int r = 0; r < 4; ++r) pk[r] = (bf16_t)(v[r] * mul);
    *reinterpret_cast<bf16x4*>(dst) = pk;
}

template <int TPW>
__global__ __launch_bounds__(Att2<TPW>::THREADS, TPW == 2 ? 2 : 4) void attention_bwd2_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                                                         const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                                         bf16_t* __restrict__ dqkv, int B, int N, int H, float scale) {
    typedef bf16_t T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int bh = blockIdx.x;
    const int b = bh / H, h = bh - b * H;
    const int C = H * 64, RS = 3 * C;
    const int Np = (N + 31) & ~31;
    const int nblk = Np / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    char* M0 = smem;
    char* M1 = smem + Np * 128;
    float* Dq = reinterpret_cast<float*>(smem + 2 * Np * 128);
    float* Ls = Dq + Np;
    const T* qb = qkv + (size_t)b * N * RS + h * 64;
    const T* ob = out + (size_t)b * N * C + h * 64;
    const T* dob = dout + (size_t)b * N * C + h * 64;
    T* dqb = dqkv + (size_t)b * N * RS + h * 64;
    constexpr int ATT2_THREADS = Att2<TPW>::THREADS;
    constexpr int NW = ATT2_THREADS / 64;
    constexpr int RPW = 16 * TPW;                  // rows owned by a wavefront per item

    // ---- phase 0: D[q] = dO[q] . O[q], two threads per query row; log-sum-exp in log2 units
    for (int i = threadIdx.x; i < Np * 2; i += ATT2_THREADS) {
        const int q = i >> 1, hf = i & 1;
        float d = 0.f;
        if (q < N) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float a[8], o[8];
                Chunk<T>::unpack(ld_chunk(dob + (size_t)q * C + (hf * 4 + c) * 8), a);
                Chunk<T>::unpack(ld_chunk(ob + (size_t)q * C + (hf * 4 + c) * 8), o);
#pragma unroll
                for (int k = 0; k < 8; ++k) d += a[k] * o[k];
            }
        }
        d += __shfl_xor(d, 1, 64);
        if (hf == 0) {
            Dq[q] = d;
            Ls[q] = q < N ? lse[((size_t)b * H + h) * N + q] * 1.4426950408889634f : 0.f;
        }
    }
    attn_stage<T>(M0, qb + C, RS, N, Np);          // K
    attn_stage<T>(M1, qb + 2 * C, RS, N, Np);      // V
    __syncthreads();

    // lane constants: row-fragment chunk (row l15 of a tile, chunks s*4 + lg) and transposed-read address (rows
    // lg*4 + (l15>>2) and +16 of a 32-row block, 16 columns dt*16..)
    int rf[2], tro[4];
#pragma unroll
    for (int s = 0; s < 2; ++s) rf[s] = l15 * 128 + ((((s * 4 + lg) ^ (l15 >> 1)) & 7) << 4);
    {
        const int trow = lg * 4 + (l15 >> 2);
        const int tsw = (trow >> 1) & 7;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) tro[dt] = trow * 128 + ((((dt * 2 + ((l15 & 3) >> 1)) ^ tsw) & 7) << 4) + ((l15 & 1) << 3);
    }
    const float c2 = scale * 1.4426950408889634f;
    const int npair = (N + RPW - 1) / RPW;         // items: groups of TPW 16-row tiles

    // ---- phase A: dQ for the 32 queries of a pair; S^T tiles (rows keys, column = this lane's query)
    for (int pr = wave; pr < npair; pr += NW) {
        u32x4 qf[TPW][2], dof[TPW][2];
        float lq[TPW], dqv[TPW];
#pragma unroll
        for (int qi = 0; qi < TPW; ++qi) {
            const int row = pr * RPW + qi * 16 + l15;
            const bool ok = row < N;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                qf[qi][s] = ok ? ld_chunk(qb + (size_t)row * RS + (s * 4 + lg) * 8) : zero_chunk();
                dof[qi][s] = ok ? ld_chunk(dob + (size_t)row * C + (s * 4 + lg) * 8) : zero_chunk();
            }
            lq[qi] = Ls[row];
            dqv[qi] = Dq[row];
        }
        f32x4 o[TPW][4];
#pragma unroll
        for (int qi = 0; qi < TPW; ++qi)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[qi][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int blk = 0; blk < nblk; ++blk) {
            const char* kb = M0 + blk * 4096;
            const char* vb = M1 + blk * 4096;
            u32x4 kf[2][2], vf[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    kf[t][s] = ld_chunk(kb + t * 2048 + rf[s]);
                    vf[t][s] = ld_chunk(vb + t * 2048 + rf[s]);
                }
            u32x4 kT[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) kT[dt] = tr_pair(kb + tro[dt]);
#pragma unroll
            for (int qi = 0; qi < TPW; ++qi) {
                f32x4 ds[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const f32x4 sv = mma2(kf[t], qf[qi]);
                    const f32x4 dp = mma2(vf[t], dof[qi]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) ds[t][r] = __builtin_amdgcn_exp2f(fmaf(sv[r], c2, -lq[qi])) * (dp[r] - dqv[qi]);
                }
                const u32x4 pa = pack_p(ds[0], ds[1]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) Mma<T>::run(o[qi][dt], kT[dt], pa);      // dQ^T: rows d, column query
            }
        }
#pragma unroll
        for (int qi = 0; qi < TPW; ++qi) {
            const int row = pr * RPW + qi * 16 + l15;
            if (row < N) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) st_bf16x4(dqb + (size_t)row * RS + dt * 16 + lg * 4, o[qi][dt], scale);
            }
        }
    }
    __syncthreads();
    attn_stage<T>(M0, qb, RS, N, Np);              // Q
    attn_stage<T>(M1, dob, C, N, Np);              // dO
    __syncthreads();

    // ---- phase B: dK, dV for the 32 keys of a pair; tiles with rows = queries, column = this lane's key
    for (int pr = wave; pr < npair; pr += NW) {
        u32x4 kf[TPW][2], vf[TPW][2];
#pragma unroll
        for (int ki = 0; ki < TPW; ++ki) {
            const int row = pr * RPW + ki * 16 + l15;
            const bool ok = row < N;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                kf[ki][s] = ok ? ld_chunk(qb + C + (size_t)row * RS + (s * 4 + lg) * 8) : zero_chunk();
                vf[ki][s] = ok ? ld_chunk(qb + 2 * C + (size_t)row * RS + (s * 4 + lg) * 8) : zero_chunk();
            }
        }
        f32x4 dv[TPW][4], dk[TPW][4];
#pragma unroll
        for (int ki = 0; ki < TPW; ++ki)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) { dv[ki][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dk[ki][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        for (int blk = 0; blk < nblk; ++blk) {
            const char* qp = M0 + blk * 4096;
            const char* dp_ = M1 + blk * 4096;
            u32x4 qf[2][2], dof[2][2];
            f32x4 Lv[2], Dv[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    qf[t][s] = ld_chunk(qp + t * 2048 + rf[s]);
                    dof[t][s] = ld_chunk(dp_ + t * 2048 + rf[s]);
                }
                Lv[t] = *reinterpret_cast<const f32x4*>(Ls + blk * 32 + t * 16 + lg * 4);
                Dv[t] = *reinterpret_cast<const f32x4*>(Dq + blk * 32 + t * 16 + lg * 4);
            }
            u32x4 doT[4], qT[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                doT[dt] = tr_pair(dp_ + tro[dt]);
                qT[dt] = tr_pair(qp + tro[dt]);
            }
#pragma unroll
            for (int ki = 0; ki < TPW; ++ki) {
                f32x4 pt[2], dst_[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const f32x4 sv = mma2(qf[t], kf[ki]);
                    const f32x4 dpp = mma2(dof[t], vf[ki]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float p = __builtin_amdgcn_exp2f(fmaf(sv[r], c2, -Lv[t][r]));
                        pt[t][r] = p;
                        dst_[t][r] = p * (dpp[r] - Dv[t][r]);
                    }
                }
                const u32x4 pa = pack_p(pt[0], pt[1]);
                const u32x4 da = pack_p(dst_[0], dst_[1]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    Mma<T>::run(dv[ki][dt], doT[dt], pa);      // dV^T: rows d, column key
                    Mma<T>::run(dk[ki][dt], qT[dt], da);       // dK^T
                }
            }
        }
#pragma unroll
        for (int ki = 0; ki < TPW; ++ki) {
            const int row = pr * RPW + ki * 16 + l15;
            if (row < N) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    st_bf16x4(dqb + (size_t)row * RS + C + dt * 16 + lg * 4, dk[ki][dt], scale);
                    st_bf16x4(dqb + (size_t)row * RS + 2 * C + dt * 16 + lg * 4, dv[ki][dt], 1.f);
                }
            }
        }
    }
}

inline int sgrid(size_t n) {
    size_t b = (n + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

template <typename K>
void allow_lds(K k, size_t bytes) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace

namespace saicv {

// rows of exactly 96 chunks (C = 768 in bf16, C = 384 in fp32) take the two-rows-per-wavefront kernels: measured on the ViT-B step
// (same box, r05): LayerNorm backward 1.755 -> 1.674 ms, forward 0.826 -> 0.798 ms per step.  SAICV_LN_HALF=0 selects the one-row kernels.
static bool ln_half_rows(int cpr) {
    const char* e = getenv("SAICV_LN_HALF");
    return cpr == 96 && !(e != nullptr && atoi(e) == 0);
}

template <typename T>
static int layernorm_fwd_t(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                           float* rstd, int M, int C, double eps, hipStream_t st, const LnDrop dr = LnDrop{}) {
    constexpr int N = Chunk<T>::N;
    const int nch = (C / N + 63) / 64;
    if (dr.branch == nullptr && ln_half_rows(C / N)) {              // 32 lanes per row, no idle chunk slots at 96 chunks
        hipLaunchKernelGGL((layernorm_fwd_half_kernel<T, 3>), dim3((M + 7) / 8), dim3(256), 0, st, (const T*)x, gamma, beta, (T*)y, mean,
                           rstd, M, C, (float)eps);
        return check_launch("layernorm_fwd");
    }
    dim3 grid((M + 3) / 4), block(256);
#define LN_LAUNCH(NCH) hipLaunchKernelGGL((layernorm_fwd_kernel<T, NCH>), grid, block, 0, st, (const T*)x, gamma, beta, (T*)y, mean, rstd, M, C, (float)eps, dr)
    if (nch <= 1) LN_LAUNCH(1);
    else if (nch <= 2) LN_LAUNCH(2);
    else if (nch <= 4) LN_LAUNCH(4);
    else if (nch <= 8) LN_LAUNCH(8);
    else { set_error("layernorm_fwd: C=%d too wide", C); return -1; }
#undef LN_LAUNCH
    return check_launch("layernorm_fwd");
}

int layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean,
                  float* rstd, int M, int C, double eps, hipStream_t st) {
    const int n = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    SAICV_REQUIRE(C % n == 0 && M > 0, "layernorm_fwd: C=%d must be a multiple of %d", C, n);
    if (dtype == SAICV_DTYPE_BF16) return layernorm_fwd_t<bf16_t>(x, gamma, beta, y, mean, rstd, M, C, eps, st);
    return layernorm_fwd_t<float>(x, gamma, beta, y, mean, rstd, M, C, eps, st);
}

static LnDrop ln_drop(const void* branch, void* sum_out, double p, unsigned seed, const unsigned* seed_device) {
    LnDrop d;
    d.branch = branch; d.sum_out = sum_out; d.seed = seed; d.seed_device = seed_device;
    d.thresh = (unsigned)fmin(p * 4294967296.0, 4294967040.0);
    d.inv_keep = (float)(1.0 / (1.0 - p));
    return d;
}

// out = LayerNorm(sum), sum = x + dropout_p(branch)   (DETR's post-norm residual, reference detection/models/detr.py:89,92,114,118,122)
int dropout_add_layernorm_fwd(int dtype, const void* x, const void* branch, double p, unsigned seed, const unsigned* seed_device,
                              const float* gamma, const float* beta, void* sum_out, void* y, float* mean, float* rstd, int M, int C,
                              double eps, hipStream_t st) {
    const int n = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    SAICV_REQUIRE(C % n == 0 && M > 0 && branch && sum_out && p >= 0.0 && p < 1.0, "dropout_add_layernorm_fwd: C=%d (multiple of %d), p=%g in [0, 1)", C, n, p);
    const LnDrop dr = ln_drop(branch, sum_out, p, seed, seed_device);
    if (dtype == SAICV_DTYPE_BF16) return layernorm_fwd_t<bf16_t>(x, gamma, beta, y, mean, rstd, M, C, eps, st, dr);
    return layernorm_fwd_t<float>(x, gamma, beta, y, mean, rstd, M, C, eps, st, dr);
}

static int ln_bwd_blocks(int M, int* rows_per) {
    // one block of 8 wavefronts per CU, each wavefront streaming its rows with a one-row prefetch: more, smaller
    // blocks were measured slower (start-up latency and dgamma / dbeta partial rows per block)
    static const int target = getenv("SAICV_LN_BWD_BLOCKS") ? atoi(getenv("SAICV_LN_BWD_BLOCKS")) : 256;
    int rp = (M + LNB_WAVES * target - 1) / (LNB_WAVES * target);
    if (rp < 1) rp = 1;
    *rows_per = rp;
    return (M + LNB_WAVES * rp - 1) / (LNB_WAVES * rp);
}

size_t layernorm_bwd_ws_floats(int M, int C) {
    int rp;
    return (size_t)2 * ln_bwd_blocks(M, &rp) * C;
}

template <typename T>
static int layernorm_bwd_t(const void* dy, const void* x, const float* gamma, const float* mean,
                           const float* rstd, const void* addend, void* dx, float* dgamma, float* dbeta, float* ws,
                           int M, int C, int accumulate, hipStream_t st, const float* out_scale, int rows_per_scale, void* dxs,
                           const LnDrop dr = LnDrop{}) {
    constexpr int N = Chunk<T>::N;
    const int nch = (C / N + 63) / 64;
    int rp;
    const int nb = ln_bwd_blocks(M, &rp);
    float* pg = ws;
    float* pb = ws + (size_t)nb * C;
    dim3 grid(nb), block(64 * LNB_WAVES);
#define LN_LAUNCH(NCH) hipLaunchKernelGGL((layernorm_bwd_kernel<T, NCH>), grid, block, 0, st, (const T*)dy, (const T*)x, gamma, mean, rstd, (const T*)addend, (T*)dx, pg, pb, M, C, rp, out_scale, rows_per_scale, (T*)dxs, dr)
    if (ln_half_rows(C / N)) {
        // the same nb blocks (the workspace is sized for them), each covering 2 * rp2 * LNB_WAVES rows with rp2 row PAIRS per wavefront
        const int rp2 = (rp + 1) / 2;
        hipLaunchKernelGGL((layernorm_bwd_half_kernel<T, 3>), grid, block, 0, st, (const T*)dy, (const T*)x, gamma, mean, rstd,
                           (const T*)addend, (T*)dx, pg, pb, M, C, rp2, out_scale, rows_per_scale, (T*)dxs, dr);
    } else
    if (nch <= 1) LN_LAUNCH(1);
    else if (nch <= 2) LN_LAUNCH(2);
    else if (nch <= 4) LN_LAUNCH(4);
    else { set_error("layernorm_bwd: C=%d too wide", C); return -1; }
#undef LN_LAUNCH
    hipLaunchKernelGGL(colreduce_kernel, dim3((C + 63) / 64, 2), dim3(1024), 0, st, pg, pb, nb, C, dgamma, dbeta, accumulate);
    return check_launch("layernorm_bwd");
}

// The backward of dropout_add_layernorm_fwd: dsum (= the gradient of x) and dbranch = mask / (1 - p) * dsum in one pass
int dropout_add_layernorm_bwd(int dtype, const void* dy, const void* sum, const float* gamma, const float* mean, const float* rstd,
                              double p, unsigned seed, const unsigned* seed_device, void* dsum, void* dbranch, float* dgamma,
                              float* dbeta, float* ws, int M, int C, int accumulate, hipStream_t st) {
    const int n = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    SAICV_REQUIRE(C % n == 0 && M > 0 && dbranch && p >= 0.0 && p < 1.0, "dropout_add_layernorm_bwd: C=%d (multiple of %d), p=%g in [0, 1)", C, n, p);
    const LnDrop dr = ln_drop(sum, nullptr, p, seed, seed_device);        // (branch != nullptr marks the dropout form of the second output)
    if (dtype == SAICV_DTYPE_BF16)
        return layernorm_bwd_t<bf16_t>(dy, sum, gamma, mean, rstd, nullptr, dsum, dgamma, dbeta, ws, M, C, accumulate, st, nullptr, 1, dbranch, dr);
    return layernorm_bwd_t<float>(dy, sum, gamma, mean, rstd, nullptr, dsum, dgamma, dbeta, ws, M, C, accumulate, st, nullptr, 1, dbranch, dr);
}

// out_scale / rows_per_scale / dxs (optional): also write dxs[row] = out_scale[row / rows_per_scale] * dx[row] -- the gradient the
// drop-path branch below this LayerNorm's input consumes (reference vit.py:160-161: x + drop_path(branch(x))), saving its own pass
int layernorm_bwd(int dtype, const void* dy, const void* x, const float* gamma, const float* mean,
                  const float* rstd, const void* addend, void* dx, float* dgamma, float* dbeta, float* ws, int M,
                  int C, int accumulate, hipStream_t st, const float* out_scale, int rows_per_scale, void* dxs) {
    const int n = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    SAICV_REQUIRE(C % n == 0 && M > 0, "layernorm_bwd: C=%d must be a multiple of %d", C, n);
    SAICV_REQUIRE(dxs == nullptr || (out_scale != nullptr && rows_per_scale >= 1), "layernorm_bwd: a scaled output needs its row factors");
    if (dtype == SAICV_DTYPE_BF16)
        return layernorm_bwd_t<bf16_t>(dy, x, gamma, mean, rstd, addend, dx, dgamma, dbeta, ws, M, C, accumulate, st, out_scale, rows_per_scale, dxs);
    return layernorm_bwd_t<float>(dy, x, gamma, mean, rstd, addend, dx, dgamma, dbeta, ws, M, C, accumulate, st, out_scale, rows_per_scale, dxs);
}

int gelu_fwd(int dtype, const void* x, void* y, size_t n, hipStream_t st) {
    const int e = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    SAICV_REQUIRE(n % e == 0, "gelu_fwd: length must be a multiple of %d", e);
    if (dtype == SAICV_DTYPE_BF16)
        hipLaunchKernelGGL(gelu_fwd_kernel<bf16_t>, dim3(sgrid(n / e)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, n / e);
    else
        hipLaunchKernelGGL(gelu_fwd_kernel<float>, dim3(sgrid(n / e)), dim3(256), 0, st, (const float*)x, (float*)y, n / e);
    return check_launch("gelu_fwd");
}

int gelu_bwd(int dtype, const void* dy, const void* x, void* dx, size_t n, hipStream_t st) {
    const int e = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    SAICV_REQUIRE(n % e == 0, "gelu_bwd: length must be a multiple of %d", e);
    if (dtype == SAICV_DTYPE_BF16)
        hipLaunchKernelGGL(gelu_bwd_kernel<bf16_t>, dim3(sgrid(n / e)), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)x, (bf16_t*)dx, n / e);
    else
        hipLaunchKernelGGL(gelu_bwd_kernel<float>, dim3(sgrid(n / e)), dim3(256), 0, st, (const float*)dy, (const float*)x, (float*)dx, n / e);
    return check_launch("gelu_bwd");
}

static int relu_dropout_check(const char* who, int dtype, size_t n, double p) {
    SAICV_REQUIRE(dtype == SAICV_DTYPE_BF16 || dtype == SAICV_DTYPE_F32, "%s: dtype %d", who, dtype);
    SAICV_REQUIRE(n % (dtype == SAICV_DTYPE_BF16 ? 8 : 4) == 0, "%s: length must be a multiple of %d", who, dtype == SAICV_DTYPE_BF16 ? 8 : 4);
    SAICV_REQUIRE(n / (dtype == SAICV_DTYPE_BF16 ? 8 : 4) <= 0xffffffffull, "%s: more than 2^32 chunks", who);
    SAICV_REQUIRE(p >= 0.0 && p < 1.0, "%s: dropout probability %g outside [0, 1)", who, p);
    return 0;
}

int relu_dropout_fwd(int dtype, const void* x, void* y, size_t n, double p, unsigned seed, const unsigned* seed_device, hipStream_t st) {
    if (relu_dropout_check("relu_dropout_fwd", dtype, n, p)) return -1;
    if (n == 0) return 0;
    const int e = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    const unsigned thresh = (unsigned)(p * 4294967296.0);
    const float inv_keep = (float)(1.0 / (1.0 - p));
    if (dtype == SAICV_DTYPE_BF16)
        hipLaunchKernelGGL(relu_dropout_fwd_kernel<bf16_t>, dim3(sgrid(n / e)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, n / e, seed, seed_device, thresh, inv_keep);
    else
        hipLaunchKernelGGL(relu_dropout_fwd_kernel<float>, dim3(sgrid(n / e)), dim3(256), 0, st, (const float*)x, (float*)y, n / e, seed, seed_device, thresh, inv_keep);
    return check_launch("relu_dropout_fwd");
}

int relu_dropout_bwd(int dtype, const void* dy, const void* y, void* dx, size_t n, double p, hipStream_t st) {
    if (relu_dropout_check("relu_dropout_bwd", dtype, n, p)) return -1;
    if (n == 0) return 0;
    const int e = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    const float inv_keep = (float)(1.0 / (1.0 - p));
    if (dtype == SAICV_DTYPE_BF16)
        hipLaunchKernelGGL(relu_dropout_bwd_kernel<bf16_t>, dim3(sgrid(n / e)), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)y, (bf16_t*)dx, n / e, inv_keep);
    else
        hipLaunchKernelGGL(relu_dropout_bwd_kernel<float>, dim3(sgrid(n / e)), dim3(256), 0, st, (const float*)dy, (const float*)y, (float*)dx, n / e, inv_keep);
    return check_launch("relu_dropout_bwd");
}

static int attn_check(const char* who, int B, int N, int H, int D) {
    SAICV_REQUIRE(D == 64, "%s: head dim %d (only 64 is implemented)", who, D);
    SAICV_REQUIRE(N >= 1 && N <= 256, "%s: sequence length %d outside [1,256] (streaming variant not built yet)", who, N);
    SAICV_REQUIRE(B >= 1 && H >= 1, "%s: empty problem", who);
    return 0;
}

int attention_fwd(int dtype, const void* qkv, void* out, float* lse, int B, int N, int H, int D, double scale,
                  hipStream_t st) {
    if (attn_check("attention_fwd", B, N, H, D)) return -1;
    const int Np = (N + 31) & ~31;
    if (dtype == SAICV_DTYPE_BF16) {
        const size_t smem = (size_t)2 * Np * 128;
        auto k = attention_fwd_kernel<bf16_t>;
        static bool once = (allow_lds(k, 2 * 256 * 128), true);
        (void)once;
        hipLaunchKernelGGL(k, dim3(B * H), dim3(ATT_THREADS), smem, st, (const bf16_t*)qkv, (bf16_t*)out, lse, B, N, H, (float)scale);
    } else {
        const size_t smem = (size_t)2 * Np * 256;
        auto k = attention_fwd_kernel<float>;
        static bool once = (allow_lds(k, 2 * 256 * 256), true);
        (void)once;
        hipLaunchKernelGGL(k, dim3(B * H), dim3(ATT_THREADS), smem, st, (const float*)qkv, (float*)out, lse, B, N, H, (float)scale);
    }
    return check_launch("attention_fwd");
}

int attention_bwd(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                  int B, int N, int H, int D, double scale, hipStream_t st) {
    if (attn_check("attention_bwd", B, N, H, D)) return -1;
    const int Np = (N + 31) & ~31;
    // bf16: SAICV_ATTN_BWD2=1 selects the two-tiles-per-wavefront kernel.  Measured equal on the ViT-B step (42.58 vs 42.64 ms,
    // profiles/r03_tn_dma_and_attention.md): half the VALU / LDS instructions per MFMA, but 221 registers leave two
    // wavefronts per SIMD to hide the three staging round trips of a workgroup, where the one-tile kernel has four.
    // =2: the same instruction mix with one tile per wavefront (128 registers, four wavefronts per SIMD).  Read per call.
    const char* env2 = getenv("SAICV_ATTN_BWD2");
    const int two = env2 ? atoi(env2) : 0;
    if (dtype == SAICV_DTYPE_BF16 && two) {
        const size_t smem = (size_t)2 * Np * 128 + 2 * Np * sizeof(float);
        if (two == 2) {                 // the lean instruction mix with ONE tile per wavefront (eight wavefronts, 4 per SIMD)
            auto k = attention_bwd2_kernel<1>;
            static bool once = (allow_lds(k, 2 * 256 * 128 + 2048), true);
            (void)once;
            hipLaunchKernelGGL(k, dim3(B * H), dim3(512), smem, st, (const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, lse, (bf16_t*)dqkv, B, N, H, (float)scale);
        } else {
            auto k = attention_bwd2_kernel<2>;
            static bool once = (allow_lds(k, 2 * 256 * 128 + 2048), true);
            (void)once;
            hipLaunchKernelGGL(k, dim3(B * H), dim3(256), smem, st, (const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, lse, (bf16_t*)dqkv, B, N, H, (float)scale);
        }
        return check_launch("attention_bwd");
    }
    if (dtype == SAICV_DTYPE_BF16) {
        const size_t smem = (size_t)2 * Np * 128 + 2 * Np * sizeof(float);
        auto k = attention_bwd_kernel<bf16_t>;
        static bool once = (allow_lds(k, 2 * 256 * 128 + 2048), true);
        (void)once;
        hipLaunchKernelGGL(k, dim3(B * H), dim3(ATT_THREADS), smem, st, (const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, lse, (bf16_t*)dqkv, B, N, H, (float)scale);
    } else {
        const size_t smem = (size_t)2 * Np * 256 + 2 * Np * sizeof(float);
        auto k = attention_bwd_kernel<float>;
        static bool once = (allow_lds(k, 2 * 256 * 256 + 2048), true);
        (void)once;
        hipLaunchKernelGGL(k, dim3(B * H), dim3(ATT_THREADS), smem, st, (const float*)qkv, (const float*)out, (const float*)dout, lse, (float*)dqkv, B, N, H, (float)scale);
    }
    return check_launch("attention_bwd");
}

}  // namespace saicv
