// Streaming (flash-style) multi-head attention for gfx950: any query / key length, head dim 32 or
// 64, separate q / k / v tensors with strides (self- and cross-attention), an optional additive
// per-key bias and an optional decomposed relative-position bias with gradients.
//
// Replaces, without ever writing the [Nq, Nk] matrix to HBM:
//   * SAM  Attention.forward + add_decomposed_rel_pos
//          (reference interactive_segmentation/models/segment_anything/image_encoder.py:116-184):
//          logits = (q*scale) k^T + rel_h[q, kh] + rel_w[q, kw]   (4096-token global blocks and
//          14x14 windows);
//   * DETR nn.MultiheadAttention self / cross attention with a FLOAT key_padding_mask, which
//          PyTorch applies as an ADDITIVE bias (+1.0 on padded keys, reference detection/models/
//          detr.py:252-260) -> `key_bias`, head dim 32.
// Structure: a workgroup = 8 wavefronts = 128 queries (forward / dQ) or 128 keys (dK,dV); the
// other side streams through LDS in 64-row chunks; scores live in the S^T accumulator layout
// (query on the lane, keys on lane-group/register) so softmax row reductions are two shuffles;
// P.V and dS.K go through transposing LDS reads.  Online softmax with running max / sum.
#include "common.h"
#include "saicv_internal.h"
#include "../../include/saicv_hip.h"

namespace {

constexpr int SA_THREADS = 512;
constexpr int SA_WAVES = 8;
constexpr int SA_CHUNK = 64;            // rows of the streamed operand per LDS chunk

typedef saicv_attn_desc SAParams;   // public descriptor (include/saicv_hip.h) is the kernel argument

// ---------------------------------------------------------------- LDS image of a [rows][D] operand
template <int ROWB> DEVINL int sa_off(int row, int chunk);
template <> DEVINL int sa_off<64>(int row, int chunk) {
    const int q = (row >> 2) & 3;
    return row * 64 + ((chunk ^ (((q & 1) << 1) ^ ((q >> 1) * 3))) << 4);
}
template <> DEVINL int sa_off<128>(int row, int chunk) { return row * 128 + (((chunk ^ (row >> 1)) & 7) << 4); }
template <> DEVINL int sa_off<256>(int row, int chunk) { return row * 256 + (((chunk ^ row) & 15) << 4); }

template <typename T, int D>
struct SA {
    static constexpr int EPC = ElemTraits<T>::EPC;
    static constexpr int ROWB = D * (int)sizeof(T);
    static constexpr int DCH = ROWB / 16;
    static constexpr int STEPS = DCH / 4;              // MFMA k-steps over d
    static constexpr int DT = D / 16;                  // 16-wide output tiles over d
    static constexpr int CHUNK_BYTES = SA_CHUNK * ROWB;

    // stage `rows` (<= SA_CHUNK) rows starting at global row r0; rows >= nvalid are zero
    static DEVINL void stage(char* lds, const T* __restrict__ g, long rs, int r0, int nvalid) {
        for (int i = threadIdx.x; i < SA_CHUNK * DCH; i += SA_THREADS) {
            const int r = i / DCH, c = i - r * DCH;
            const u32x4 v = (r0 + r) < nvalid ? ld_chunk(g + (size_t)(r0 + r) * rs + c * EPC) : zero_chunk();
            st_chunk(lds + sa_off<ROWB>(r, c), v);
        }
    }
    static DEVINL void lds_frags(u32x4 (&f)[STEPS], const char* lds, int row0, int l15, int lg) {
#pragma unroll
        for (int s = 0; s < STEPS; ++s) f[s] = ld_chunk(lds + sa_off<ROWB>(row0 + l15, s * 4 + lg));
    }
    static DEVINL void gmem_frags(u32x4 (&f)[STEPS], const T* __restrict__ g, long rs, int row0, int nvalid, int l15, int lg) {
        const int r = row0 + l15;
#pragma unroll
        for (int s = 0; s < STEPS; ++s)
            f[s] = r < nvalid ? ld_chunk(g + (size_t)r * rs + (s * 4 + lg) * EPC) : zero_chunk();
    }
    static DEVINL f32x4 tile(const u32x4 (&a)[STEPS], const u32x4 (&b)[STEPS]) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < STEPS; ++s) Mma<T>::run(acc, a[s], b[s]);
        return acc;
    }
    // o(16 x D) += P(16 x 32) * M(32 x D), P as two C-layout tiles (rows kbase + {0,16} + lg*4 + r of M)
    static DEVINL void pv(f32x4 (&o)[DT], const f32x4& t0, const f32x4& t1, const char* lds, int kbase, int l15, int lg) {
        if constexpr (sizeof(T) == 2) {
            typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
            float f[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
            const u32x4 pa = Chunk<bf16_t>::pack(f);
            const int r0 = kbase + lg * 4 + (l15 >> 2), r1 = r0 + 16;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int col = dt * 16 + (l15 & 3) * 4;
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(lds + sa_off<ROWB>(r0, col >> 3) + ((col & 4) << 1)));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(lds + sa_off<ROWB>(r1, col >> 3) + ((col & 4) << 1)));
                const u32x2 a = __builtin_bit_cast(u32x2, lo), b = __builtin_bit_cast(u32x2, hi);
                const u32x4 vb = {a[0], a[1], b[0], b[1]};
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, pa), __builtin_bit_cast(bf16x8, vb), o[dt], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int half = 0; half < 2; ++half)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = kbase + half * 16 + lg * 4 + r;
                    const float a = half == 0 ? t0[r] : t1[r];
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) {
                        const int d = dt * 16 + l15;
                        const float b = *reinterpret_cast<const float*>(lds + sa_off<ROWB>(row, d >> 2) + (d & 3) * 4);
                        o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, o[dt], 0, 0, 0);
                    }
                }
        }
    }
};

// additive bias of (query q, key) in the S^T layout: the lane owns one query (its rel rows are
// staged in LDS as rh[SA_Q16][Sh+1], rw[..][Sw+1]), keys vary
DEVINL float sa_bias(const float* rh, const float* rw, int Sh, int Sw, int l15, int key, const float* kb) {
    float b = kb ? kb[key] : 0.f;
    if (rh) {
        const int kh = key / Sw, kw = key - kh * Sw;
        b += rh[l15 * (Sh + 1) + kh] + rw[l15 * (Sw + 1) + kw];
    }
    return b;
}

// ------------------------------------------------------------------------------------ forward
template <typename T, int D>
__global__ __launch_bounds__(SA_THREADS) void sa_fwd_kernel(const SAParams p) {
    using S = SA<T, D>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, lg = lane >> 4;
    char* Ks = smem;
    char* Vs = smem + S::CHUNK_BYTES;
    float* relbuf = reinterpret_cast<float*>(smem + 2 * S::CHUNK_BYTES) + wave * 16 * (p.Sh + p.Sw + 2);
    const T* qg = (const T*)p.q + (size_t)b * p.q_bs + h * D;
    const T* kg = (const T*)p.k + (size_t)b * p.k_bs + h * D;
    const T* vg = (const T*)p.v + (size_t)b * p.v_bs + h * D;
    const float* kb = p.key_bias ? p.key_bias + (size_t)b * p.Nk : nullptr;
    const int q0 = blockIdx.x * 128 + wave * 16;
    const bool has_rel = p.rel_h != nullptr;
    float* rh = has_rel ? relbuf : nullptr;
    float* rw = has_rel ? relbuf + 16 * (p.Sh + 1) : nullptr;
    if (has_rel) {      // this wave's 16 query rows of rel_h / rel_w -> LDS (pitch S+1)
        for (int i = lane; i < 16 * p.Sh; i += 64) {
            const int r = i / p.Sh, c = i - r * p.Sh;
            rh[r * (p.Sh + 1) + c] = (q0 + r) < p.Nq ? p.rel_h[((size_t)bh * p.Nq + q0 + r) * p.Sh + c] : 0.f;
        }
        for (int i = lane; i < 16 * p.Sw; i += 64) {
            const int r = i / p.Sw, c = i - r * p.Sw;
            rw[r * (p.Sw + 1) + c] = (q0 + r) < p.Nq ? p.rel_w[((size_t)bh * p.Nq + q0 + r) * p.Sw + c] : 0.f;
        }
    }
    u32x4 qf[S::STEPS];
    S::gmem_frags(qf, qg, p.q_rs, q0, p.Nq, l15, lg);
    float m_run = -INFINITY, l_run = 0.f;
    f32x4 o[S::DT];
#pragma unroll
    for (int dt = 0; dt < S::DT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int k0 = 0; k0 < p.Nk; k0 += SA_CHUNK) {
        __syncthreads();                                  // previous chunk fully consumed
        S::stage(Ks, kg, p.k_rs, k0, p.Nk);
        S::stage(Vs, vg, p.v_rs, k0, p.Nk);
        __syncthreads();
        f32x4 st[4];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            u32x4 kf[S::STEPS];
            S::lds_frags(kf, Ks, kt * 16, l15, lg);
            st[kt] = S::tile(kf, qf);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = k0 + kt * 16 + lg * 4 + r;
                st[kt][r] = key < p.Nk ? st[kt][r] * p.scale + sa_bias(rh, rw, p.Sh, p.Sw, l15, key, kb) : -INFINITY;
                mx = fmaxf(mx, st[kt][r]);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = expf(m_run - m_new);          // first chunk: exp(-inf) = 0
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                st[kt][r] = expf(st[kt][r] - m_new);
                psum += st[kt][r];
            }
        l_run = l_run * alpha + psum;                      // per-lane partial sum (same alpha on all lane groups)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float aq = __shfl(alpha, lg * 4 + r, 64);   // O rows are queries lg*4 + r
#pragma unroll
            for (int dt = 0; dt < S::DT; ++dt) o[dt][r] *= aq;
        }
        S::pv(o, st[0], st[1], Vs, 0, l15, lg);
        S::pv(o, st[2], st[3], Vs, 32, l15, lg);
    }
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    if (lg == 0 && q0 + l15 < p.Nq) p.lse[(size_t)bh * p.Nq + q0 + l15] = m_run + logf(l_run);
    const float inv = 1.f / l_run;
    T* og = (T*)p.out + (size_t)b * p.o_bs + h * D;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float iq = __shfl(inv, lg * 4 + r, 64);
        const int q = q0 + lg * 4 + r;
        if (q < p.Nq) {
#pragma unroll
            for (int dt = 0; dt < S::DT; ++dt) og[(size_t)q * p.o_rs + dt * 16 + l15] = from_f32<T>(o[dt][r] * iq);
        }
    }
}

// ------------------------------------------------------------------------------------ backward: dQ (+ D, d rel-pos)
template <typename T, int D>
__global__ __launch_bounds__(SA_THREADS) void sa_bwd_dq_kernel(const SAParams p) {
    using S = SA<T, D>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, lg = lane >> 4;
    char* Ks = smem;
    char* Vs = smem + S::CHUNK_BYTES;
    const int relpitch = 16 * (p.Sh + p.Sw + 2);
    float* relbuf = reinterpret_cast<float*>(smem + 2 * S::CHUNK_BYTES) + wave * 2 * relpitch;
    const T* qg = (const T*)p.q + (size_t)b * p.q_bs + h * D;
    const T* kg = (const T*)p.k + (size_t)b * p.k_bs + h * D;
    const T* vg = (const T*)p.v + (size_t)b * p.v_bs + h * D;
    const T* og = (const T*)p.out + (size_t)b * p.o_bs + h * D;
    const T* dog = (const T*)p.dout + (size_t)b * p.o_bs + h * D;
    const float* kb = p.key_bias ? p.key_bias + (size_t)b * p.Nk : nullptr;
    const int q0 = blockIdx.x * 128 + wave * 16;
    const bool has_rel = p.rel_h != nullptr;
    float* rh = has_rel ? relbuf : nullptr;
    float* rw = has_rel ? relbuf + 16 * (p.Sh + 1) : nullptr;
    float* gh = has_rel ? relbuf + relpitch : nullptr;         // gradient accumulators, same shape
    float* gw = has_rel ? gh + 16 * (p.Sh + 1) : nullptr;
    if (has_rel) {
        for (int i = lane; i < 16 * p.Sh; i += 64) {
            const int r = i / p.Sh, c = i - r * p.Sh;
            rh[r * (p.Sh + 1) + c] = (q0 + r) < p.Nq ? p.rel_h[((size_t)bh * p.Nq + q0 + r) * p.Sh + c] : 0.f;
            gh[r * (p.Sh + 1) + c] = 0.f;
        }
        for (int i = lane; i < 16 * p.Sw; i += 64) {
            const int r = i / p.Sw, c = i - r * p.Sw;
            rw[r * (p.Sw + 1) + c] = (q0 + r) < p.Nq ? p.rel_w[((size_t)bh * p.Nq + q0 + r) * p.Sw + c] : 0.f;
            gw[r * (p.Sw + 1) + c] = 0.f;
        }
    }
    u32x4 qf[S::STEPS], dof[S::STEPS];
    S::gmem_frags(qf, qg, p.q_rs, q0, p.Nq, l15, lg);
    S::gmem_frags(dof, dog, p.o_rs, q0, p.Nq, l15, lg);
    const int q = q0 + l15;
    const bool qok = q < p.Nq;
    // D[q] = sum_d dO*O: this lane covers chunks s*4+lg of its query row
    float dsum = 0.f;
    if (qok) {
#pragma unroll
        for (int s = 0; s < S::STEPS; ++s) {
            float a[Chunk<T>::N], c[Chunk<T>::N];
            Chunk<T>::unpack(dof[s], a);
            Chunk<T>::unpack(ld_chunk(og + (size_t)q * p.o_rs + (s * 4 + lg) * S::EPC), c);
#pragma unroll
            for (int k = 0; k < Chunk<T>::N; ++k) dsum += a[k] * c[k];
        }
    }
    dsum += __shfl_xor(dsum, 16, 64);
    dsum += __shfl_xor(dsum, 32, 64);
    if (lg == 0 && qok) p.dsum[(size_t)bh * p.Nq + q] = dsum;
    const float lq = qok ? p.lse[(size_t)bh * p.Nq + q] : 0.f;
    f32x4 o[S::DT];
#pragma unroll
    for (int dt = 0; dt < S::DT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int k0 = 0; k0 < p.Nk; k0 += SA_CHUNK) {
        __syncthreads();
        S::stage(Ks, kg, p.k_rs, k0, p.Nk);
        S::stage(Vs, vg, p.v_rs, k0, p.Nk);
        __syncthreads();
        f32x4 ds[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            u32x4 kf[S::STEPS], vf[S::STEPS];
            S::lds_frags(kf, Ks, kt * 16, l15, lg);
            S::lds_frags(vf, Vs, kt * 16, l15, lg);
            const f32x4 sv = S::tile(kf, qf);
            const f32x4 dp = S::tile(vf, dof);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = k0 + kt * 16 + lg * 4 + r;
                float g = 0.f;
                if (key < p.Nk && qok) {
                    const float pr = expf(sv[r] * p.scale + sa_bias(rh, rw, p.Sh, p.Sw, l15, key, kb) - lq);
                    g = pr * (dp[r] - dsum);               // d logits
                    if (has_rel) {
                        const int kh = key / p.Sw, kw = key - kh * p.Sw;
                        atomicAdd(&gh[l15 * (p.Sh + 1) + kh], g);
                        atomicAdd(&gw[l15 * (p.Sw + 1) + kw], g);
                    }
                }
                ds[kt][r] = g * p.scale;
            }
        }
        S::pv(o, ds[0], ds[1], Ks, 0, l15, lg);            // dQ += dS K
        S::pv(o, ds[2], ds[3], Ks, 32, l15, lg);
    }
    T* dqg = (T*)p.dq + (size_t)b * p.q_bs + h * D;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int qq = q0 + lg * 4 + r;
        if (qq < p.Nq) {
#pragma unroll
            for (int dt = 0; dt < S::DT; ++dt) dqg[(size_t)qq * p.q_rs + dt * 16 + l15] = from_f32<T>(o[dt][r]);
        }
    }
    if (has_rel) {      // LDS atomics of this wave are visible to the wave after its own waits
        __builtin_amdgcn_s_waitcnt(0);
        for (int i = lane; i < 16 * p.Sh; i += 64) {
            const int r = i / p.Sh, c = i - r * p.Sh;
            if (q0 + r < p.Nq) p.d_rel_h[((size_t)bh * p.Nq + q0 + r) * p.Sh + c] = gh[r * (p.Sh + 1) + c];
        }
        for (int i = lane; i < 16 * p.Sw; i += 64) {
            const int r = i / p.Sw, c = i - r * p.Sw;
            if (q0 + r < p.Nq) p.d_rel_w[((size_t)bh * p.Nq + q0 + r) * p.Sw + c] = gw[r * (p.Sw + 1) + c];
        }
    }
}

// ------------------------------------------------------------------------------------ backward: dK, dV
// P / dS tiles in the un-swapped layout (col = key l15, rows = queries lg*4 + r)
template <typename T, int D>
__global__ __launch_bounds__(SA_THREADS) void sa_bwd_dkv_kernel(const SAParams p) {
    using S = SA<T, D>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, lg = lane >> 4;
    char* Qs = smem;
    char* Os = smem + S::CHUNK_BYTES;                      // dO chunk
    float* Dq = reinterpret_cast<float*>(smem + 2 * S::CHUNK_BYTES);
    float* Ls = Dq + SA_CHUNK;
    const T* qg = (const T*)p.q + (size_t)b * p.q_bs + h * D;
    const T* kg = (const T*)p.k + (size_t)b * p.k_bs + h * D;
    const T* vg = (const T*)p.v + (size_t)b * p.v_bs + h * D;
    const T* dog = (const T*)p.dout + (size_t)b * p.o_bs + h * D;
    const int key0 = blockIdx.x * 128 + wave * 16;
    const int key = key0 + l15;
    const bool kok = key < p.Nk;
    const float kbias = (p.key_bias && kok) ? p.key_bias[(size_t)b * p.Nk + key] : 0.f;
    const int kh = (p.rel_h && kok) ? key / p.Sw : 0;
    const int kw = (p.rel_h && kok) ? key - kh * p.Sw : 0;
    u32x4 kf[S::STEPS], vf[S::STEPS];
    S::gmem_frags(kf, kg, p.k_rs, key0, p.Nk, l15, lg);
    S::gmem_frags(vf, vg, p.v_rs, key0, p.Nk, l15, lg);
    f32x4 dv[S::DT], dk[S::DT];
#pragma unroll
    for (int dt = 0; dt < S::DT; ++dt) { dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    for (int q0 = 0; q0 < p.Nq; q0 += SA_CHUNK) {
        __syncthreads();
        S::stage(Qs, qg, p.q_rs, q0, p.Nq);
        S::stage(Os, dog, p.o_rs, q0, p.Nq);
        for (int i = threadIdx.x; i < SA_CHUNK; i += SA_THREADS) {
            const bool ok = q0 + i < p.Nq;
            Dq[i] = ok ? p.dsum[(size_t)bh * p.Nq + q0 + i] : 0.f;
            Ls[i] = ok ? p.lse[(size_t)bh * p.Nq + q0 + i] : 0.f;
        }
        __syncthreads();
        f32x4 pt[4], dst[4];
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            u32x4 qf[S::STEPS], dof[S::STEPS];
            S::lds_frags(qf, Qs, qt * 16, l15, lg);
            S::lds_frags(dof, Os, qt * 16, l15, lg);
            const f32x4 sv = S::tile(qf, kf);              // rows queries, col key
            const f32x4 dp = S::tile(dof, vf);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ql = qt * 16 + lg * 4 + r, qq = q0 + ql;
                float pr = 0.f;
                if (qq < p.Nq && kok) {
                    float bias = kbias;
                    if (p.rel_h)
                        bias += p.rel_h[((size_t)bh * p.Nq + qq) * p.Sh + kh] + p.rel_w[((size_t)bh * p.Nq + qq) * p.Sw + kw];
                    pr = expf(sv[r] * p.scale + bias - Ls[ql]);
                }
                pt[qt][r] = pr;
                dst[qt][r] = pr * (dp[r] - Dq[ql]) * p.scale;
            }
        }
        S::pv(dv, pt[0], pt[1], Os, 0, l15, lg);           // dV += P^T dO
        S::pv(dv, pt[2], pt[3], Os, 32, l15, lg);
        S::pv(dk, dst[0], dst[1], Qs, 0, l15, lg);         // dK += dS^T Q
        S::pv(dk, dst[2], dst[3], Qs, 32, l15, lg);
    }
    T* dkg = (T*)p.dk + (size_t)b * p.k_bs + h * D;
    T* dvg = (T*)p.dv + (size_t)b * p.v_bs + h * D;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int kk = key0 + lg * 4 + r;
        if (kk < p.Nk) {
#pragma unroll
            for (int dt = 0; dt < S::DT; ++dt) {
                dkg[(size_t)kk * p.k_rs + dt * 16 + l15] = from_f32<T>(dk[dt][r]);
                dvg[(size_t)kk * p.v_rs + dt * 16 + l15] = from_f32<T>(dv[dt][r]);
            }
        }
    }
}

template <typename K>
void sa_allow_lds(K k) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

template <typename T, int D>
int sa_launch(const SAParams& p, int which, hipStream_t st) {
    const size_t chunk = (size_t)SA_CHUNK * D * sizeof(T);
    const size_t rel = p.rel_h ? (size_t)SA_WAVES * 16 * (p.Sh + p.Sw + 2) * sizeof(float) : 0;
    if (which == 0) {
        auto k = sa_fwd_kernel<T, D>;
        static bool once = (sa_allow_lds(k), true);
        (void)once;
        hipLaunchKernelGGL(k, dim3((p.Nq + 127) / 128, p.B * p.H), dim3(SA_THREADS), 2 * chunk + rel, st, p);
    } else if (which == 1) {
        auto k = sa_bwd_dq_kernel<T, D>;
        static bool once = (sa_allow_lds(k), true);
        (void)once;
        hipLaunchKernelGGL(k, dim3((p.Nq + 127) / 128, p.B * p.H), dim3(SA_THREADS), 2 * chunk + 2 * rel, st, p);
    } else {
        auto k = sa_bwd_dkv_kernel<T, D>;
        static bool once = (sa_allow_lds(k), true);
        (void)once;
        hipLaunchKernelGGL(k, dim3((p.Nk + 127) / 128, p.B * p.H), dim3(SA_THREADS), 2 * chunk + 2 * SA_CHUNK * sizeof(float), st, p);
    }
    return saicv::check_launch("attention_stream");
}

}  // namespace

namespace saicv {

int attention_stream(int dtype, int D, int which, const void* desc_ptr, hipStream_t st) {
    const SAParams& p = *reinterpret_cast<const SAParams*>(desc_ptr);
    SAICV_REQUIRE(D == 32 || D == 64, "attention_stream: head dim %d (32 or 64)", D);
    SAICV_REQUIRE(p.B >= 1 && p.H >= 1 && p.Nq >= 1 && p.Nk >= 1, "attention_stream: empty problem");
    const int e = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    SAICV_REQUIRE(p.q_rs % e == 0 && p.k_rs % e == 0 && p.v_rs % e == 0 && p.o_rs % e == 0 &&
                      p.q_bs % e == 0 && p.k_bs % e == 0 && p.v_bs % e == 0 && p.o_bs % e == 0,
                  "attention_stream: strides must be multiples of %d elements (16-byte rows)", e);
    SAICV_REQUIRE((p.rel_h == nullptr) == (p.rel_w == nullptr), "attention_stream: rel_h and rel_w come together");
    if (p.rel_h) {
        SAICV_REQUIRE(p.Sh >= 1 && p.Sw >= 1 && p.Sh * p.Sw == p.Nk, "attention_stream: Sh*Sw must equal Nk");
        SAICV_REQUIRE(p.Sh + p.Sw <= 160, "attention_stream: relative-position tables too wide for LDS");
    }
    if (dtype == SAICV_DTYPE_BF16) return D == 64 ? sa_launch<bf16_t, 64>(p, which, st) : sa_launch<bf16_t, 32>(p, which, st);
    return D == 64 ? sa_launch<float, 64>(p, which, st) : sa_launch<float, 32>(p, which, st);
}

}  // namespace saicv
