// SAM image-encoder layout and relative-position kernels for gfx950.
//
//  * window_partition / window_unpartition (reference interactive_segmentation/models/segment_anything/
//    image_encoder.py:32-79): zero-pad the token grid to a multiple of the window, regroup into windows, and
//    back -- ONE streaming pass each (the reference needs pad + permute-copy, then permute-copy + slice-copy),
//    the un-partition optionally fused with the residual add of Block.forward (:236).
//  * relpos_fwd / relpos_bwd (get_rel_pos + add_decomposed_rel_pos, :82-144): the decomposed relative-position
//    logits rel_h[b*heads + n, q, kh] = <q[b, q, n, :], rel_pos_h[qh - kh + S - 1, :]> (and rel_w), their
//    gradient into q and into the two tables.  0.8 % of the attention flops: plain fp32 FMA kernels with the
//    table rows in LDS, replacing four strided batched GEMM dispatches, two gathers and two index_adds.
// All are HBM / latency bound; activations are bf16 (perf mode) or fp32 (parity mode).
#include <stdlib.h>
#include "common.h"
#include "saicv_internal.h"
#include "det.h"

namespace {

// ------------------------------------------------------------------------------------------------ windows
template <typename T>
__global__ __launch_bounds__(256) void window_partition_kernel(const T* __restrict__ x, T* __restrict__ out, int B,
                                                               int H, int W, int C, int ws, int nwh, int nww) {
    constexpr int N = Chunk<T>::N;
    const int cpr = C / N;
    const size_t total = (size_t)B * nwh * nww * ws * ws * cpr;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int c = (int)(i % cpr);
        size_t r = i / cpr;                       // row of out: ((b * nwh + wy) * nww + wx) * ws*ws + iy * ws + ix
        const int ix = (int)(r % ws); r /= ws;
        const int iy = (int)(r % ws); r /= ws;
        const int wx = (int)(r % nww); r /= nww;
        const int wy = (int)(r % nwh);
        const int b = (int)(r / nwh);
        const int y = wy * ws + iy, xx = wx * ws + ix;
        u32x4 v = zero_chunk();
        if (y < H && xx < W) v = ld_chunk(x + (((size_t)b * H + y) * W + xx) * C + (size_t)c * N);
        st_chunk(out + i * N, v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void window_unpartition_kernel(const T* __restrict__ win, const T* __restrict__ addend,
                                                                 T* __restrict__ out, int B, int H, int W, int C,
                                                                 int ws, int nwh, int nww) {
    constexpr int N = Chunk<T>::N;
    const int cpr = C / N;
    const size_t total = (size_t)B * H * W * cpr;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int c = (int)(i % cpr);
        size_t r = i / cpr;
        const int xx = (int)(r % W); r /= W;
        const int y = (int)(r % H);
        const int b = (int)(r / H);
        const int wy = y / ws, iy = y - wy * ws, wx = xx / ws, ix = xx - wx * ws;
        const size_t src = ((((size_t)b * nwh + wy) * nww + wx) * ws * ws + (size_t)iy * ws + ix) * C + (size_t)c * N;
        u32x4 v = ld_chunk(win + src);
        if (addend != nullptr) {
            float f[N], a[N];
            Chunk<T>::unpack(v, f);
            Chunk<T>::unpack(ld_chunk(addend + i * N), a);
#pragma unroll
            for (int k = 0; k < N; ++k) f[k] += a[k];
            v = Chunk<T>::pack(f);
        }
        st_chunk(out + i * N, v);
    }
}

inline int sgrid(size_t total) {
    size_t b = (total + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

// ------------------------------------------------------------------------------------------------ rel-pos
constexpr int RP_D = 64;                  // head dim
constexpr int RP_PITCH = RP_D + 4;        // floats per LDS table row: 16-byte aligned, rows 4 banks apart
constexpr int RP_THREADS = 256;
constexpr int RP_COPIES = 32;             // privatised table-gradient copies (atomic contention / 32), summed by a last kernel

struct RelPosParams {
    const void* q; long q_rs, q_bs;       // q[b, n, head*64 + c]: element strides of a row / a batch entry
    void* dq;                             // same layout as q (backward: dq += extra)
    const float* tab_h; const float* tab_w;       // [2*Sh-1][64], [2*Sw-1][64]
    float* rel_h; float* rel_w;           // [B*heads][N][Sh], [B*heads][N][Sw]   (backward: their gradients)
    float* dtab_h; float* dtab_w;         // table gradients: RP_COPIES privatised copies, `copy_stride` floats apart
    long copy_stride;
    int B, heads, Sh, Sw;
    int copies;                           // RP_COPIES; deterministic mode: one copy per workgroup (its adds then have no second contributor)
};

// stage the table rows this block needs: H rows for query row qh are tab_h[qh - kh + Sh - 1], kh = 0..Sh-1
// (stored at LDS row kh); all 2*Sw-1 rows of tab_w
DEVINL void rp_stage_tables(float* th, float* tw, const RelPosParams& p, int qh) {
    for (int i = threadIdx.x; i < p.Sh * (RP_D / 4); i += RP_THREADS) {
        const int kh = i / (RP_D / 4), c4 = i - kh * (RP_D / 4);
        *reinterpret_cast<f32x4*>(th + kh * RP_PITCH + c4 * 4) =
            *reinterpret_cast<const f32x4*>(p.tab_h + (size_t)(qh - kh + p.Sh - 1) * RP_D + c4 * 4);
    }
    for (int i = threadIdx.x; i < (2 * p.Sw - 1) * (RP_D / 4); i += RP_THREADS) {
        const int j = i / (RP_D / 4), c4 = i - j * (RP_D / 4);
        *reinterpret_cast<f32x4*>(tw + j * RP_PITCH + c4 * 4) = *reinterpret_cast<const f32x4*>(p.tab_w + (size_t)j * RP_D + c4 * 4);
    }
}

template <typename T>
DEVINL void rp_load_q(float (&qv)[RP_D], const T* __restrict__ qrow) {
    constexpr int N = Chunk<T>::N;
#pragma unroll
    for (int j = 0; j < RP_D / N; ++j) {
        float f[N];
        Chunk<T>::unpack(ld_chunk(qrow + j * N), f);
#pragma unroll
        for (int k = 0; k < N; ++k) qv[j * N + k] = f[k];
    }
}

// grid (Sh, B): block (b, qh); items = (qw, head)
template <typename T>
__global__ __launch_bounds__(RP_THREADS) void relpos_fwd_kernel(const RelPosParams p) {
    extern __shared__ __attribute__((aligned(16))) float rp_smem[];
    float* th = rp_smem;
    float* tw = rp_smem + p.Sh * RP_PITCH;
    const int qh = blockIdx.x, b = blockIdx.y;
    rp_stage_tables(th, tw, p, qh);
    __syncthreads();
    const int N = p.Sh * p.Sw;
    for (int item = threadIdx.x; item < p.Sw * p.heads; item += RP_THREADS) {
        const int qw = item / p.heads, head = item - qw * p.heads;
        const int qi = qh * p.Sw + qw;
        float qv[RP_D];
        rp_load_q<T>(qv, (const T*)p.q + (size_t)b * p.q_bs + (size_t)qi * p.q_rs + head * RP_D);
        auto dot = [&](const float* row) {
            float acc = 0.f;
#pragma unroll
            for (int c4 = 0; c4 < RP_D / 4; ++c4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(row + c4 * 4);
                acc = fmaf(qv[c4 * 4], t[0], fmaf(qv[c4 * 4 + 1], t[1], fmaf(qv[c4 * 4 + 2], t[2], fmaf(qv[c4 * 4 + 3], t[3], acc))));
            }
            return acc;
        };
        // 16-byte stores when the rows of rel_h / rel_w are 16-byte aligned (S % 4 == 0), else scalar
        float* oh = p.rel_h + (((size_t)b * p.heads + head) * N + qi) * p.Sh;
        if ((p.Sh & 3) == 0) {
            for (int kh = 0; kh < p.Sh; kh += 4) {
                f32x4 v;
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = dot(th + (kh + k) * RP_PITCH);
                *reinterpret_cast<f32x4*>(oh + kh) = v;
            }
        } else {
            for (int kh = 0; kh < p.Sh; ++kh) oh[kh] = dot(th + kh * RP_PITCH);
        }
        float* ow = p.rel_w + (((size_t)b * p.heads + head) * N + qi) * p.Sw;
        if ((p.Sw & 3) == 0) {
            for (int kw = 0; kw < p.Sw; kw += 4) {
                f32x4 v;
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = dot(tw + (qw - (kw + k) + p.Sw - 1) * RP_PITCH);
                *reinterpret_cast<f32x4*>(ow + kw) = v;
            }
        } else {
            for (int kw = 0; kw < p.Sw; ++kw) ow[kw] = dot(tw + (qw - kw + p.Sw - 1) * RP_PITCH);
        }
    }
}

// The same logits on the matrix cores (bf16 q): for one query row qh the Sw queries x 64 channels of a head times
//   Th^T  (Sh table rows qh - kh + Sh - 1, kh = 0..Sh-1)            -> rel_h[q][kh]
//   Tw'^T (ALL 2 Sw - 1 rows, stored REVERSED: Tw'[j] = Tw[2 Sw - 2 - j])  -> P[q][j],  rel_w[q][kw] = P[q][kw + Sw - 1 - qw]
// are two small GEMMs (K = 64): 96 MFMAs per head at Sh = Sw = 64, against 8 192 fp32 FMAs per query in the kernel above
// (which runs at the vector-ALU rate: 1.05 ms per 4096-token block where its 0.63 GB of traffic would take 0.13).  Tables are
// rounded to bf16 in LDS (what the reference's autocast einsum does), fp32 accumulation, fp32 outputs.  Work items are (head,
// 16-query tile) pairs dealt to the four wavefronts; a D tile (rows = queries lg*4 + r, column = l15) is stored as 16 consecutive
// floats of four query rows per instruction.
DEVINL int rpm_off(int row, int chunk) { return row * 128 + (((chunk ^ (row >> 1)) & 7) << 4); }

__global__ __launch_bounds__(RP_THREADS) void relpos_fwd_mfma_kernel(const RelPosParams p) {
    extern __shared__ __attribute__((aligned(16))) char rpm_smem[];
    const int qh = blockIdx.x, b = blockIdx.y;
    const int Sh = p.Sh, Sw = p.Sw, J = 2 * Sw - 1;
    const int ShP = (Sh + 15) & ~15, JP = (J + 15) & ~15;
    char* th = rpm_smem;                         // [ShP][64] bf16, swizzled rows of 128 bytes
    char* tw = rpm_smem + ShP * 128;             // [JP][64] bf16
    for (int i = threadIdx.x; i < (ShP + JP) * 8; i += RP_THREADS) {
        const int row = i >> 3, c = i & 7;
        const bool ish = row < ShP;
        const int r = ish ? row : row - ShP;
        const bool ok = ish ? r < Sh : r < J;
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = 0.f;
        if (ok) {
            const float* src = ish ? p.tab_h + (size_t)(qh - r + Sh - 1) * RP_D : p.tab_w + (size_t)(J - 1 - r) * RP_D;
            const f32x4 a = *reinterpret_cast<const f32x4*>(src + c * 8), bq = *reinterpret_cast<const f32x4*>(src + c * 8 + 4);
            f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3]; f[4] = bq[0]; f[5] = bq[1]; f[6] = bq[2]; f[7] = bq[3];
        }
        st_chunk((ish ? th : tw) + rpm_off(r, c), Chunk<bf16_t>::pack(f));
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int MT = (Sw + 15) >> 4, NTH = ShP >> 4, NTW = JP >> 4;
    const int N = Sh * Sw;
    for (int item = wave; item < p.heads * MT; item += RP_THREADS / 64) {
        const int head = item / MT, mt = item - head * MT;
        const int qw_a = mt * 16 + l15;                                  // the query this lane feeds as an A row
        u32x4 a[2];
        {
            const bf16_t* qrow = (const bf16_t*)p.q + (size_t)b * p.q_bs + (size_t)(qh * Sw + qw_a) * p.q_rs + head * RP_D;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) a[ks] = qw_a < Sw ? ld_chunk(qrow + (ks * 4 + lg) * 8) : zero_chunk();
        }
        const size_t plane = ((size_t)b * p.heads + head) * N + (size_t)qh * Sw;
        for (int nt = 0; nt < NTH; ++nt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) Mma<bf16_t>::run(acc, a[ks], ld_chunk(th + rpm_off(nt * 16 + l15, ks * 4 + lg)));
            const int kh = nt * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qw = mt * 16 + lg * 4 + r;
                if (qw < Sw && kh < Sh) p.rel_h[(plane + qw) * Sh + kh] = acc[r];
            }
        }
        for (int nt = 0; nt < NTW; ++nt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) Mma<bf16_t>::run(acc, a[ks], ld_chunk(tw + rpm_off(nt * 16 + l15, ks * 4 + lg)));
            const int j = nt * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qw = mt * 16 + lg * 4 + r;
                const int kw = j - (Sw - 1 - qw);
                if (qw < Sw && kw >= 0 && kw < Sw) p.rel_w[(plane + qw) * Sw + kw] = acc[r];
            }
        }
    }
}

// dq of the 64 x 64 global blocks on the matrix cores (bf16): dQ^T[c][q] = Th^T[c][kh] dRelH^T[kh][q] + Tw'^T[c][j] dP^T[j][q],
// dP[q][j] = dRelW[q][j - (Sw - 1 - qw)] (zero outside the row).  Tables are staged TRANSPOSED in LDS (rows = channels), the
// gradients of the logits are rounded to bf16 as they are loaded; a lane ends up with four consecutive channels of one query
// and adds them to dq with one 8-byte read-modify-write.
// LDS image with CH 16-byte chunks per row (CH = 4, 8, 16), XOR-swizzled by the row
template <int CH> DEVINL int rpm_offc(int row, int chunk) {
    return row * CH * 16 + (((chunk ^ (row >> (CH == 8 ? 1 : CH == 16 ? 0 : 2))) & (CH - 1)) << 4);
}
// eight consecutive floats row[w0 .. w0+7] of a row of S floats (zero outside [0, S)), rounded to bf16
template <bool ALIGNED = false>
DEVINL u32x4 rp_load8(const float* __restrict__ row, int w0, int S) {
    float f[8];
    if constexpr (ALIGNED) {                          // caller guarantees: 16-byte aligned and wholly inside the row
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(row + w0), x1 = *reinterpret_cast<const f32x4*>(row + w0 + 4);
        const float g[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
        return Chunk<bf16_t>::pack(g);
    }
    if (w0 >= 0 && w0 + 8 <= S) {                     // wholly inside: two 4-byte-aligned 16-byte loads
        typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
        const f32x4_u x0 = *reinterpret_cast<const f32x4_u*>(row + w0), x1 = *reinterpret_cast<const f32x4_u*>(row + w0 + 4);
        f[0] = x0[0]; f[1] = x0[1]; f[2] = x0[2]; f[3] = x0[3]; f[4] = x1[0]; f[5] = x1[1]; f[6] = x1[2]; f[7] = x1[3];
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = w0 + e;
            f[e] = (k >= 0 && k < S) ? row[k] : 0.f;
        }
    }
    return Chunk<bf16_t>::pack(f);
}

// KH / KW: 32-wide k-steps of the kh / j contractions (2 / 4 for the 64 x 64 global blocks, 1 / 1 for windows up to 16 x 16)
template <int KH, int KW>
__global__ __launch_bounds__(RP_THREADS) void relpos_bwd_dq_mfma_kernel(const RelPosParams p) {
    constexpr int CHH = KH * 4, CHW = KW * 4;                          // chunks per LDS row
    __shared__ __attribute__((aligned(16))) char thT[RP_D * CHH * 16];  // [c][kh] bf16, zero beyond Sh
    __shared__ __attribute__((aligned(16))) char twT[RP_D * CHW * 16];  // [c][j]  bf16, zero beyond 2 Sw - 2
    const int qh = blockIdx.x, b = blockIdx.y;
    const int Sh = KH == 2 ? 64 : p.Sh, Sw = KH == 2 ? 64 : p.Sw, J = 2 * Sw - 1;       // the global-block instantiation is 64 x 64 (constants fold)
    for (int i = threadIdx.x; i < KH * 32 * RP_D; i += RP_THREADS) {
        const int kh = i / RP_D, c = i - kh * RP_D;
        const bf16_t v = kh < Sh ? (bf16_t)p.tab_h[(size_t)(qh - kh + Sh - 1) * RP_D + c] : (bf16_t)0.f;
        *reinterpret_cast<bf16_t*>(thT + rpm_offc<CHH>(c, kh >> 3) + (kh & 7) * 2) = v;
    }
    for (int i = threadIdx.x; i < KW * 32 * RP_D; i += RP_THREADS) {
        const int j = i / RP_D, c = i - j * RP_D;
        const bf16_t v = j < J ? (bf16_t)p.tab_w[(size_t)(J - 1 - j) * RP_D + c] : (bf16_t)0.f;
        *reinterpret_cast<bf16_t*>(twT + rpm_offc<CHW>(c, j >> 3) + (j & 7) * 2) = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int N = Sh * Sw, MT = (Sw + 15) >> 4;
    for (int item = wave; item < p.heads * MT; item += RP_THREADS / 64) {
        const int head = item / MT, mt = item - head * MT;
        const int qw = mt * 16 + l15;                                 // this lane's query (a column of the product)
        const bool qok = qw < Sw;
        const size_t row = ((size_t)b * p.heads + head) * N + (size_t)qh * Sw + (qok ? qw : 0);
        u32x4 bh[KH], bw[KW];
#pragma unroll
        for (int ks = 0; ks < KH; ++ks) bh[ks] = qok ? rp_load8<KH == 2>(p.rel_h + row * Sh, ks * 32 + lg * 8, Sh) : zero_chunk();      // KH == 2: Sh == 64
#pragma unroll
        for (int ks = 0; ks < KW; ++ks) bw[ks] = qok ? rp_load8(p.rel_w + row * Sw, ks * 32 + lg * 8 - (Sw - 1 - qw), Sw) : zero_chunk();
        bf16_t* dq = (bf16_t*)p.dq + (size_t)b * p.q_bs + (size_t)(qh * Sw + (qok ? qw : 0)) * p.q_rs + head * RP_D;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KH; ++ks) Mma<bf16_t>::run(acc, ld_chunk(thT + rpm_offc<CHH>(ct * 16 + l15, ks * 4 + lg)), bh[ks]);
#pragma unroll
            for (int ks = 0; ks < KW; ++ks) Mma<bf16_t>::run(acc, ld_chunk(twT + rpm_offc<CHW>(ct * 16 + l15, ks * 4 + lg)), bw[ks]);
            // D^T tile: rows = channels ct*16 + lg*4 + r, column = query l15
            if (qok) {
                bf16x4* dst = reinterpret_cast<bf16x4*>(dq + ct * 16 + lg * 4);
                const bf16x4 old = *dst;
                bf16x4 nw;
#pragma unroll
                for (int r = 0; r < 4; ++r) nw[r] = (bf16_t)((float)old[r] + acc[r]);
                *dst = nw;
            }
        }
    }
}

// backward, part 1 (same grid): dq[b, q, head, :] += sum_kh drh * Th[kh] + sum_kw drw * Tw[qw - kw + Sw - 1]
template <typename T>
__global__ __launch_bounds__(RP_THREADS) void relpos_bwd_dq_kernel(const RelPosParams p) {
    constexpr int NC = Chunk<T>::N;
    extern __shared__ __attribute__((aligned(16))) float rp_smem[];
    float* th = rp_smem;
    float* tw = rp_smem + p.Sh * RP_PITCH;
    const int qh = blockIdx.x, b = blockIdx.y;
    rp_stage_tables(th, tw, p, qh);
    __syncthreads();
    const int N = p.Sh * p.Sw;
    for (int item = threadIdx.x; item < p.Sw * p.heads; item += RP_THREADS) {
        const int qw = item / p.heads, head = item - qw * p.heads;
        const int qi = qh * p.Sw + qw;
        float acc[RP_D];
#pragma unroll
        for (int c = 0; c < RP_D; ++c) acc[c] = 0.f;
        auto axpy = [&](float g, const float* row) {
#pragma unroll
            for (int c4 = 0; c4 < RP_D / 4; ++c4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(row + c4 * 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[c4 * 4 + k] = fmaf(g, t[k], acc[c4 * 4 + k]);
            }
        };
        const float* gh = p.rel_h + (((size_t)b * p.heads + head) * N + qi) * p.Sh;
        if ((p.Sh & 3) == 0) {
            for (int kh = 0; kh < p.Sh; kh += 4) {
                const f32x4 g = *reinterpret_cast<const f32x4*>(gh + kh);
#pragma unroll
                for (int k = 0; k < 4; ++k) axpy(g[k], th + (kh + k) * RP_PITCH);
            }
        } else {
            for (int kh = 0; kh < p.Sh; ++kh) axpy(gh[kh], th + kh * RP_PITCH);
        }
        const float* gw = p.rel_w + (((size_t)b * p.heads + head) * N + qi) * p.Sw;
        if ((p.Sw & 3) == 0) {
            for (int kw = 0; kw < p.Sw; kw += 4) {
                const f32x4 g = *reinterpret_cast<const f32x4*>(gw + kw);
#pragma unroll
                for (int k = 0; k < 4; ++k) axpy(g[k], tw + (qw - (kw + k) + p.Sw - 1) * RP_PITCH);
            }
        } else {
            for (int kw = 0; kw < p.Sw; ++kw) axpy(gw[kw], tw + (qw - kw + p.Sw - 1) * RP_PITCH);
        }
        T* dq = (T*)p.dq + (size_t)b * p.q_bs + (size_t)qi * p.q_rs + head * RP_D;
#pragma unroll
        for (int j = 0; j < RP_D / NC; ++j) {
            float f[NC];
            Chunk<T>::unpack(ld_chunk(dq + j * NC), f);
#pragma unroll
            for (int k = 0; k < NC; ++k) f[k] += acc[j * NC + k];
            st_chunk(dq + j * NC, Chunk<T>::pack(f));
        }
    }
}

// backward, part 2 (same grid): table gradients.  Items (qw, head) of the block go through LDS in tiles of 64:
// q tile [64][64] and the two gradient tiles; thread t owns output column group c4 = t & 15 and rows
// r0 + 16 i of dTh (i < NH) and of dTw (i < NW), r0 = t >> 4.  The gradient tiles are zero padded so that every
// accumulation is unconditional: gh to NH*16 columns; gw to Sw + NW*16 columns with the real values at offset
// NW*16 - Sw, which maps table row j to column qw - j + NW*16 - 1 for any j in [0, NW*16).
constexpr int RP_TILE = 64;
template <typename T, int NH, int NW>
__global__ __launch_bounds__(RP_THREADS) void relpos_bwd_tab_kernel(const RelPosParams p) {
    extern __shared__ __attribute__((aligned(16))) float rp_smem[];
    constexpr int GH_W = NH * 16;
    constexpr int GW_W = 64 + NW * 16;                     // constexpr pitch (Sw <= 64): no runtime divisions
    const int off_w = NW * 16 - p.Sw;
    float* qs = rp_smem;                                   // [RP_TILE][RP_PITCH]
    float* gh = qs + RP_TILE * RP_PITCH;                   // [RP_TILE][GH_W]
    float* gw = gh + RP_TILE * GH_W;                       // [RP_TILE][GW_W]
    int* qws = reinterpret_cast<int*>(gw + RP_TILE * GW_W);    // [RP_TILE] qw of each item
    int* rows = qws + RP_TILE;                             // [RP_TILE] row (b*heads + head) * N + q of each item, -1 = none
    int* qoff = rows + RP_TILE;                            // [RP_TILE] element offset of the item's q vector
    const int qh = blockIdx.x, b = blockIdx.y;
    const int N = p.Sh * p.Sw;
    const int c4 = threadIdx.x & 15, r0 = threadIdx.x >> 4;        // 16 row lanes
    const int nitems = p.Sw * p.heads;
    f32x4 ah[NH], aw[NW];
#pragma unroll
    for (int i = 0; i < NH; ++i) ah[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NW; ++i) aw[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the zero padding of the two gradient tiles is written once; tiles only overwrite their data columns
    for (int i = threadIdx.x; i < RP_TILE * (GH_W + GW_W); i += RP_THREADS) gh[i] = 0.f;
    const bool vec = ((p.Sh | p.Sw) & 3) == 0 && (off_w & 3) == 0;
    for (int base = 0; base < nitems; base += RP_TILE) {
        __syncthreads();
        if (threadIdx.x < RP_TILE) {
            const int item = base + threadIdx.x;
            const int it_c = min(item, nitems - 1);
            const int qw = it_c / p.heads, head = it_c - qw * p.heads;
            qws[threadIdx.x] = qw;
            rows[threadIdx.x] = item < nitems ? (b * p.heads + head) * N + qh * p.Sw + qw : -1;
            qoff[threadIdx.x] = (qh * p.Sw + qw) * (int)p.q_rs + head * RP_D;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < RP_TILE * (RP_D / Chunk<T>::N); i += RP_THREADS) {
            constexpr int NC = Chunk<T>::N;
            const int it = i / (RP_D / NC), j = i - it * (RP_D / NC);
            float f[NC];
#pragma unroll
            for (int k = 0; k < NC; ++k) f[k] = 0.f;
            if (rows[it] >= 0) Chunk<T>::unpack(ld_chunk((const T*)p.q + (size_t)b * p.q_bs + qoff[it] + j * NC), f);
#pragma unroll
            for (int k = 0; k < NC; ++k) qs[it * RP_PITCH + j * NC + k] = f[k];
        }
        if (vec) {        // 16-byte loads, all of a thread's loads in flight together (the tile is latency bound)
            const int h4 = p.Sh >> 2, w4 = p.Sw >> 2;
            for (int i = threadIdx.x; i < RP_TILE * h4; i += RP_THREADS) {
                const int it = i / h4, k4 = i - it * h4;
                const int row = rows[it];
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (row >= 0) v = *reinterpret_cast<const f32x4*>(p.rel_h + (size_t)row * p.Sh + k4 * 4);
                *reinterpret_cast<f32x4*>(gh + it * GH_W + k4 * 4) = v;
            }
            for (int i = threadIdx.x; i < RP_TILE * w4; i += RP_THREADS) {
                const int it = i / w4, k4 = i - it * w4;
                const int row = rows[it];
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (row >= 0) v = *reinterpret_cast<const f32x4*>(p.rel_w + (size_t)row * p.Sw + k4 * 4);
                *reinterpret_cast<f32x4*>(gw + it * GW_W + off_w + k4 * 4) = v;
            }
        } else {
#pragma unroll 4
            for (int i = threadIdx.x; i < RP_TILE * p.Sh; i += RP_THREADS) {
                const int it = i / p.Sh, k = i - it * p.Sh;
                const int row = rows[it];
                gh[it * GH_W + k] = row >= 0 ? p.rel_h[(size_t)row * p.Sh + k] : 0.f;
            }
#pragma unroll 4
            for (int i = threadIdx.x; i < RP_TILE * p.Sw; i += RP_THREADS) {
                const int it = i / p.Sw, k = i - it * p.Sw;
                const int row = rows[it];
                gw[it * GW_W + off_w + k] = row >= 0 ? p.rel_w[(size_t)row * p.Sw + k] : 0.f;
            }
        }
        __syncthreads();
        for (int it = 0; it < RP_TILE; ++it) {             // rows past the last item are zero: no tail branch
            const f32x4 qv = *reinterpret_cast<const f32x4*>(qs + it * RP_PITCH + c4 * 4);
            const float* ghr = gh + it * GH_W + r0;
            const float* gwr = gw + it * GW_W + qws[it] + NW * 16 - 1 - r0;
#pragma unroll
            for (int i = 0; i < NH; ++i) ah[i] += ghr[16 * i] * qv;
#pragma unroll
            for (int i = 0; i < NW; ++i) aw[i] += gwr[-16 * i] * qv;
        }
    }
    const int copy = (blockIdx.y * gridDim.x + blockIdx.x) % p.copies;
#pragma unroll
    for (int i = 0; i < NH; ++i) {
        const int kh = r0 + 16 * i;
        if (kh < p.Sh) {
            float* d = p.dtab_h + (size_t)copy * p.copy_stride + (size_t)(qh - kh + p.Sh - 1) * RP_D + c4 * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) unsafeAtomicAdd(d + k, ah[i][k]);
        }
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int j = r0 + 16 * i;
        if (j < 2 * p.Sw - 1) {
            float* d = p.dtab_w + (size_t)copy * p.copy_stride + (size_t)j * RP_D + c4 * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) unsafeAtomicAdd(d + k, aw[i][k]);
        }
    }
}

// Table gradients of the 64 x 64 global blocks on the matrix cores (bf16 q):
//   dTh[kh][c]  = sum over (head, qw) of dRelH[q][kh] q[q][c]             (table row qh - kh + 63)
//   dTw'[j][c]  = sum over (head, qw) of dP[q][j]     q[q][c],  dP[q][j] = dRelW[q][j - (63 - qw)]   (table row 126 - j)
// for the 64 queries of row qh: contraction over q, so both operands are reduction-major -- staged per head into LDS as bf16
// ([q][kh], [q][j] with the skew applied while staging, [q][c]) and read with transposing LDS loads (the weight-gradient
// kernel's fragment scheme and pair swizzle).  Wavefront w owns channel tile w and keeps its 4 + 8 accumulator tiles over all
// heads; the block adds them into its privatised copy with fp32 atomics.  The FMA kernel above takes 1.33 ms per block.
DEVINL int rpt_key(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }
template <int CPR> DEVINL int rpt_g(int r) { return CPR >= 16 ? rpt_key(r) : CPR == 8 ? (rpt_key(r) >> 1) : CPR == 4 ? (rpt_key(r) >> 2) : 0; }
// byte offset of 16-byte chunk c8 of row r in an LDS image with CPR chunks per row
template <int CPR> DEVINL int rpt_off(int r, int c8) { return r * CPR * 16 + ((((c8 >> 1) ^ rpt_g<CPR>(r)) << 5) | ((c8 & 1) << 4)); }
// transposed fragment: k = rows kbase + lg*8 + {0..7}, m (or n) = columns tile*16 + l15
template <int CPR> DEVINL u32x4 rpt_frag(const char* img, int kbase, int tile, int l15, int lg) {
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    const int row0 = kbase + lg * 8 + (l15 >> 2);
    const char* q = img + row0 * CPR * 16 + ((tile ^ rpt_g<CPR>(row0)) << 5) + (l15 & 3) * 8;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)q);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(q + 4 * CPR * 16));
    const u32x2 a = __builtin_bit_cast(u32x2, lo), b = __builtin_bit_cast(u32x2, hi);
    return u32x4{a[0], a[1], b[0], b[1]};
}

// KQ: 32-wide k-steps over the queries of a row; MH / MW: 16-row tiles of kh / j (2, 4, 8 for 64 x 64; 1, 1, 2 up to 16 x 16)
template <int KQ, int MH, int MW>
__global__ __launch_bounds__(RP_THREADS) void relpos_bwd_tab_mfma_kernel(const RelPosParams p) {
    constexpr int QR = KQ * 32, CH = MH * 2, CW = MW * 2;
    __shared__ __attribute__((aligned(16))) char qimg[QR * 128];          // [q][c]  bf16, zero rows beyond Sw
    __shared__ __attribute__((aligned(16))) char himg[QR * CH * 16];      // [q][kh] bf16
    __shared__ __attribute__((aligned(16))) char wimg[QR * CW * 16];      // [q][j]  bf16, j = kw + Sw - 1 - qw
    const int qh = blockIdx.x, b = blockIdx.y;
    const int Sh = KQ == 2 ? 64 : p.Sh, Sw = KQ == 2 ? 64 : p.Sw, J = 2 * Sw - 1;       // the global-block instantiation is 64 x 64 (constants fold)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;        // wave = channel tile
    const int l15 = lane & 15, lg = lane >> 4;
    const int N = Sh * Sw;
    f32x4 ah[MH], aw[MW];
#pragma unroll
    for (int i = 0; i < MH; ++i) ah[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < MW; ++i) aw[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int head = 0; head < p.heads; ++head) {
        const size_t row0 = ((size_t)b * p.heads + head) * N + (size_t)qh * Sw;
        __syncthreads();                                               // the previous head's fragments are consumed
        for (int i = threadIdx.x; i < QR * 8; i += RP_THREADS) {
            const int q = i >> 3, c8 = i & 7;
            st_chunk(qimg + rpt_off<8>(q, c8), q < Sw ? ld_chunk((const bf16_t*)p.q + (size_t)b * p.q_bs + (size_t)(qh * Sw + q) * p.q_rs + head * RP_D + c8 * 8)
                                                      : zero_chunk());
        }
        for (int i = threadIdx.x; i < QR * CH; i += RP_THREADS) {
            const int q = i / CH, c8 = i - q * CH;
            st_chunk(himg + rpt_off<CH>(q, c8), q < Sw ? rp_load8<KQ == 2>(p.rel_h + (row0 + q) * Sh, c8 * 8, Sh) : zero_chunk());      // KQ == 2: 64 x 64
        }
        for (int i = threadIdx.x; i < QR * CW; i += RP_THREADS) {      // dP: skewed by Sw - 1 - qw
            const int q = i / CW, c8 = i - q * CW;
            st_chunk(wimg + rpt_off<CW>(q, c8), q < Sw ? rp_load8(p.rel_w + (row0 + q) * Sw, c8 * 8 - (Sw - 1 - q), Sw) : zero_chunk());
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < KQ; ++ks) {
            const u32x4 bq = rpt_frag<8>(qimg, ks * 32, wave, l15, lg);            // B: k = q, n = channels of this wavefront's tile
#pragma unroll
            for (int mt = 0; mt < MH; ++mt) Mma<bf16_t>::run(ah[mt], rpt_frag<CH>(himg, ks * 32, mt, l15, lg), bq);
#pragma unroll
            for (int mt = 0; mt < MW; ++mt) Mma<bf16_t>::run(aw[mt], rpt_frag<CW>(wimg, ks * 32, mt, l15, lg), bq);
        }
    }
    // D tile: rows m = mt*16 + lg*4 + r (kh or j), column = channel wave*16 + l15
    const int copy = (blockIdx.y * gridDim.x + blockIdx.x) % p.copies;
    float* dh = p.dtab_h + (size_t)copy * p.copy_stride;
    float* dw = p.dtab_w + (size_t)copy * p.copy_stride;
    const int c = wave * 16 + l15;
#pragma unroll
    for (int mt = 0; mt < MH; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kh = mt * 16 + lg * 4 + r;
            if (kh < Sh) unsafeAtomicAdd(dh + (size_t)(qh - kh + Sh - 1) * RP_D + c, ah[mt][r]);
        }
#pragma unroll
    for (int mt = 0; mt < MW; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = mt * 16 + lg * 4 + r;
            if (j < J) unsafeAtomicAdd(dw + (size_t)(J - 1 - j) * RP_D + c, aw[mt][r]);
        }
}

// dst[i] += sum over the privatised copies
__global__ __launch_bounds__(256) void relpos_tab_reduce_kernel(const float* __restrict__ ws, long copy_stride, int n,
                                                                float* __restrict__ dst_h, int n_h, float* __restrict__ dst_w) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v = 0.f;
#pragma unroll 8
    for (int c = 0; c < RP_COPIES; ++c) v += ws[(size_t)c * copy_stride + i];
    if (i < n_h) dst_h[i] += v;
    else dst_w[i - n_h] += v;
}

template <typename K>
void rp_allow_lds(K k) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

}  // namespace

namespace saicv {

static int window_check(const char* who, int dtype, int B, int H, int W, int C, int ws) {
    const int n = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    SAICV_REQUIRE(B > 0 && H > 0 && W > 0 && ws > 0 && C % n == 0, "%s: B=%d H=%d W=%d C=%d ws=%d (C must be a multiple of %d)",
                  who, B, H, W, C, ws, n);
    return 0;
}

int window_partition(int dtype, const void* x, void* out, int B, int H, int W, int C, int ws, hipStream_t st) {
    if (window_check("window_partition", dtype, B, H, W, C, ws)) return -1;
    const int nwh = (H + ws - 1) / ws, nww = (W + ws - 1) / ws;
    const int n = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    const size_t total = (size_t)B * nwh * nww * ws * ws * (C / n);
    if (dtype == SAICV_DTYPE_BF16)
        hipLaunchKernelGGL(window_partition_kernel<bf16_t>, dim3(sgrid(total)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)out, B, H, W, C, ws, nwh, nww);
    else
        hipLaunchKernelGGL(window_partition_kernel<float>, dim3(sgrid(total)), dim3(256), 0, st, (const float*)x, (float*)out, B, H, W, C, ws, nwh, nww);
    return check_launch("window_partition");
}

int window_unpartition(int dtype, const void* win, const void* addend, void* out, int B, int H, int W, int C, int ws,
                       hipStream_t st) {
    if (window_check("window_unpartition", dtype, B, H, W, C, ws)) return -1;
    const int nwh = (H + ws - 1) / ws, nww = (W + ws - 1) / ws;
    const int n = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    const size_t total = (size_t)B * H * W * (C / n);
    if (dtype == SAICV_DTYPE_BF16)
        hipLaunchKernelGGL(window_unpartition_kernel<bf16_t>, dim3(sgrid(total)), dim3(256), 0, st, (const bf16_t*)win, (const bf16_t*)addend, (bf16_t*)out, B, H, W, C, ws, nwh, nww);
    else
        hipLaunchKernelGGL(window_unpartition_kernel<float>, dim3(sgrid(total)), dim3(256), 0, st, (const float*)win, (const float*)addend, (float*)out, B, H, W, C, ws, nwh, nww);
    return check_launch("window_unpartition");
}

static int relpos_fill(RelPosParams& p, const char* who, int dtype, const void* q, long q_rs, long q_bs, const float* tab_h,
                       const float* tab_w, int B, int heads, int Sh, int Sw) {
    const int n = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    SAICV_REQUIRE(B > 0 && heads > 0 && Sh > 0 && Sw > 0 && Sh <= 128 && 2 * Sw - 1 <= 128, "%s: B=%d heads=%d Sh=%d Sw=%d (sizes up to 64 x 64)", who, B, heads, Sh, Sw);
    SAICV_REQUIRE(q_rs % n == 0 && q_bs % n == 0, "%s: q strides must keep rows 16-byte aligned", who);
    p.q = q; p.q_rs = q_rs; p.q_bs = q_bs; p.tab_h = tab_h; p.tab_w = tab_w; p.B = B; p.heads = heads; p.Sh = Sh; p.Sw = Sw;
    return 0;
}

int relpos_fwd(int dtype, const void* q, long q_rs, long q_bs, const float* tab_h, const float* tab_w, float* rel_h,
               float* rel_w, int B, int heads, int Sh, int Sw, hipStream_t st) {
    RelPosParams p = {};
    if (relpos_fill(p, "relpos_fwd", dtype, q, q_rs, q_bs, tab_h, tab_w, B, heads, Sh, Sw)) return -1;
    p.rel_h = rel_h; p.rel_w = rel_w;
    const size_t smem = (size_t)(Sh + 2 * Sw - 1) * RP_PITCH * sizeof(float);
    static const int use_mfma = getenv("SAICV_RELPOS_MFMA") ? atoi(getenv("SAICV_RELPOS_MFMA")) : 1;
    if (dtype == SAICV_DTYPE_BF16 && use_mfma) {
        const size_t smem_m = (size_t)(((Sh + 15) & ~15) + ((2 * Sw - 1 + 15) & ~15)) * 128;
        hipLaunchKernelGGL(relpos_fwd_mfma_kernel, dim3(Sh, B), dim3(RP_THREADS), smem_m, st, p);
    } else if (dtype == SAICV_DTYPE_BF16) {
        auto k = relpos_fwd_kernel<bf16_t>;
        static bool once = (rp_allow_lds(k), true); (void)once;
        hipLaunchKernelGGL(k, dim3(Sh, B), dim3(RP_THREADS), smem, st, p);
    } else {
        auto k = relpos_fwd_kernel<float>;
        static bool once = (rp_allow_lds(k), true); (void)once;
        hipLaunchKernelGGL(k, dim3(Sh, B), dim3(RP_THREADS), smem, st, p);
    }
    return check_launch("relpos_fwd");
}

size_t relpos_bwd_ws_floats(int Sh, int Sw) { return (size_t)RP_COPIES * (2 * Sh - 1 + 2 * Sw - 1) * RP_D; }

int relpos_bwd(int dtype, const void* q, void* dq, long q_rs, long q_bs, const float* tab_h, const float* tab_w,
               const float* d_rel_h, const float* d_rel_w, float* dtab_h, float* dtab_w, float* ws, int B, int heads, int Sh,
               int Sw, hipStream_t st) {
    RelPosParams p = {};
    if (relpos_fill(p, "relpos_bwd", dtype, q, q_rs, q_bs, tab_h, tab_w, B, heads, Sh, Sw)) return -1;
    SAICV_REQUIRE((dtab_h == nullptr) == (dtab_w == nullptr) && (dtab_h == nullptr || ws != nullptr),
                  "relpos_bwd: table gradients come together and need the workspace");
    const int n_h = (2 * Sh - 1) * RP_D, n_w = (2 * Sw - 1) * RP_D;
    p.dq = dq; p.rel_h = const_cast<float*>(d_rel_h); p.rel_w = const_cast<float*>(d_rel_w);
    p.copy_stride = n_h + n_w;
    p.dtab_h = ws; p.dtab_w = ws ? ws + n_h : nullptr;       // kernels write the privatised copies
    p.copies = RP_COPIES;
    // deterministic mode (det.h): a copy per workgroup in the library's workspace, folded in workgroup order
    DetParts det;
    if (dtab_h && det.begin(st, Sh * B, (size_t)(n_h + n_w), "relpos_bwd")) return -1;
    if (det.on()) {
        p.dtab_h = det.sink().part; p.dtab_w = det.sink().part + n_h;
        p.copies = Sh * B;
    } else if (dtab_h) {
        hipMemsetAsync(ws, 0, relpos_bwd_ws_floats(Sh, Sw) * sizeof(float), st);
    }
    const size_t smem1 = (size_t)(Sh + 2 * Sw - 1) * RP_PITCH * sizeof(float);
    const int nh = (Sh + 15) / 16, nw = (2 * Sw - 1 + 15) / 16;
    const int NHc = nh <= 1 ? 1 : nh <= 4 ? 4 : 8, NWc = nw <= 2 ? 2 : 8;
    SAICV_REQUIRE(Sw <= 64, "relpos_bwd: Sw=%d (table-gradient tiles are laid out for Sw <= 64)", Sw);
    const size_t smem2 = (size_t)(RP_TILE * RP_PITCH + RP_TILE * NHc * 16 + RP_TILE * (64 + NWc * 16)) * sizeof(float) +
                         3 * RP_TILE * sizeof(int);
#define RP_TAB(TT)                                                                                           \
    do {                                                                                                     \
        if (NHc == 1 && NWc == 2) { auto k2 = relpos_bwd_tab_kernel<TT, 1, 2>; static bool o1 = (rp_allow_lds(k2), true); (void)o1;             \
            hipLaunchKernelGGL(k2, dim3(Sh, B), dim3(RP_THREADS), smem2, st, p); }                            \
        else if (NHc == 4 && NWc == 8) { auto k2 = relpos_bwd_tab_kernel<TT, 4, 8>; static bool o2 = (rp_allow_lds(k2), true); (void)o2;      \
            hipLaunchKernelGGL(k2, dim3(Sh, B), dim3(RP_THREADS), smem2, st, p); }                            \
        else { auto k2 = relpos_bwd_tab_kernel<TT, 8, 8>; static bool o3 = (rp_allow_lds(k2), true); (void)o3;                                \
            hipLaunchKernelGGL(k2, dim3(Sh, B), dim3(RP_THREADS), (size_t)(RP_TILE * RP_PITCH + RP_TILE * 128 + RP_TILE * (64 + 128)) * sizeof(float) + 3 * RP_TILE * sizeof(int), st, p); } \
    } while (0)
    static const int use_mfma = getenv("SAICV_RELPOS_MFMA") ? atoi(getenv("SAICV_RELPOS_MFMA")) : 1;
    if (dtype == SAICV_DTYPE_BF16) {
        if (use_mfma && Sh == 64 && Sw == 64) {
            hipLaunchKernelGGL((relpos_bwd_dq_mfma_kernel<2, 4>), dim3(Sh, B), dim3(RP_THREADS), 0, st, p);
        } else if (use_mfma && Sh <= 16 && Sw <= 16) {
            hipLaunchKernelGGL((relpos_bwd_dq_mfma_kernel<1, 1>), dim3(Sh, B), dim3(RP_THREADS), 0, st, p);
        } else {
            auto k1 = relpos_bwd_dq_kernel<bf16_t>;
            static bool once = (rp_allow_lds(k1), true); (void)once;
            hipLaunchKernelGGL(k1, dim3(Sh, B), dim3(RP_THREADS), smem1, st, p);
        }
        if (dtab_h) {
            if (use_mfma && Sh == 64 && Sw == 64) hipLaunchKernelGGL((relpos_bwd_tab_mfma_kernel<2, 4, 8>), dim3(Sh, B), dim3(RP_THREADS), 0, st, p);
            else if (use_mfma && Sh <= 16 && Sw <= 16) hipLaunchKernelGGL((relpos_bwd_tab_mfma_kernel<1, 1, 2>), dim3(Sh, B), dim3(RP_THREADS), 0, st, p);
            else RP_TAB(bf16_t);
        }
    } else {
        auto k1 = relpos_bwd_dq_kernel<float>;
        static bool once = (rp_allow_lds(k1), true); (void)once;
        hipLaunchKernelGGL(k1, dim3(Sh, B), dim3(RP_THREADS), smem1, st, p);
        if (dtab_h) RP_TAB(float);
    }
#undef RP_TAB
    if (det.on()) {
        if (check_launch("relpos_bwd")) return -2;
        if (det.fold(dtab_h, 0, (size_t)n_h)) return -1;
        return det.fold(dtab_w, (size_t)n_h, (size_t)n_w);
    }
    if (dtab_h)
        hipLaunchKernelGGL(relpos_tab_reduce_kernel, dim3((n_h + n_w + 255) / 256), dim3(256), 0, st, ws, p.copy_stride,
                           n_h + n_w, dtab_h, n_h, dtab_w);
    return check_launch("relpos_bwd");
}

}  // namespace saicv
