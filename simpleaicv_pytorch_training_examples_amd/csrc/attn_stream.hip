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
//
// Structure.  A workgroup is 4 wavefronts; each wavefront owns 32 rows (queries in the forward
// and dQ kernels, keys in the dK/dV kernel) as two 16-row MFMA tiles that share every LDS
// fragment read; the other operand streams through LDS in 64-row chunks, moved by the DMA engine
// (`buffer_load ... lds` as an assembly statement, sa_dma16) one or two chunks ahead of the one
// being consumed.  Scores live in the S^T accumulator layout (query on the lane, keys on
// lane-group / register), so softmax row reductions are two lane swaps; P.V and dS.K use
// transposing LDS reads.  Softmax runs in the log2 domain (v_exp_f32).
//
// What bounds these kernels (r04, rocprofv3 SQ counters + the instruction streams): the vector
// issue port -- an MFMA holds it four slots, every other vector instruction one -- and any s_waitcnt vmcnt the compiler places
// inside a chunk (it drains the DMA pieces in flight).  Hence: packed fp32 elementwise math with the row constants folded into
// one fma / the MFMA C operand, masks only in the partial chunk, no compiler-visible VMEM operation between the kernels' own
// waits at the top of a chunk (sa_settle), registers instead of LDS read-modify-write for the rel-pos gradients.
// Relative-position modes (template REL):
//   0 none;
//   1 small tables (Sh + Sw <= 32, Nk <= 256: SAM windows): rows of rel_h / rel_w in LDS, gradients
//     as one more MFMA against a 0/1 key -> (kh, kw) indicator matrix;
//   2 Sw == 64 (SAM global blocks): a 64-key chunk is exactly one kh row, so rel_w lives in 16
//     registers per tile and rel_h is one value per chunk; gradients accumulate in registers;
//   3 anything else: LDS tables + LDS atomics (slow, kept for generality).
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "saicv_internal.h"
#include "../../include/saicv_hip.h"

namespace {

constexpr int SA_THREADS = 256;
constexpr int SA_WAVES = 4;
constexpr int SA_CHUNK = 64;            // rows of the streamed operand per LDS chunk
constexpr int SA_WROWS = 32;            // rows owned by one wavefront (two MFMA tiles)
constexpr int SA_BROWS = SA_WAVES * SA_WROWS;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr int SA_KB_LDS = 8192;         // longest key-bias row staged in LDS (32 KB)

typedef saicv_attn_desc SAParams;   // public descriptor (include/saicv_hip.h) is the kernel argument

// ---------------------------------------------------------------- LDS image of a [rows][cols] operand
template <int ROWB> DEVINL int sa_off(int row, int chunk);
template <> DEVINL int sa_off<64>(int row, int chunk) {
    const int q = (row >> 2) & 3;
    return row * 64 + ((chunk ^ (((q & 1) << 1) ^ ((q >> 1) * 3))) << 4);
}
template <> DEVINL int sa_off<128>(int row, int chunk) { return row * 128 + (((chunk ^ (row >> 1)) & 7) << 4); }
template <> DEVINL int sa_off<256>(int row, int chunk) { return row * 256 + (((chunk ^ row) & 15) << 4); }

DEVINL float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// attention-probability dropout (nn.MultiheadAttention(dropout=p) in DETR): a counter-based hash of
// (seed, batch*head, query, key) decides, identically in the forward and both backward kernels,
// whether P[q, key] survives; survivors are scaled by 1 / (1 - p).  The softmax normaliser and the
// saved log-sum-exp are those of the undropped probabilities.
DEVINL bool sa_keep(unsigned seed, int bh, int q, int key, int Nk, unsigned thresh) {
    unsigned h = (unsigned)q * (unsigned)Nk + (unsigned)key;
    h ^= seed;
    h += (unsigned)bh * 0x9E3779B9u;
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h >= thresh;
}
DEVINL unsigned sa_thresh(float p) { return (unsigned)fminf(p * 4294967296.f, 4294967040.f); }

// acc[s][dt] += A_s(16 x 32) * M(32 x 16*DT): A_s given as two C-layout tiles t0[s], t1[s] (lane: A row l15,
// k = kbase + {0,16} + lg*4 + r), M an LDS image with ROWB-byte rows = k.  The B fragments are read once
// and shared by the NS row sets.
template <typename T, int ROWB, int DT, int NS>
DEVINL void pvN(f32x4 (&acc)[NS][DT], const f32x4 (&t0)[NS], const f32x4 (&t1)[NS], const char* lds, int kbase,
                int l15, int lg) {
    if constexpr (sizeof(T) == 2) {
        typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
        u32x4 pa[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            float f[8] = {t0[s][0], t0[s][1], t0[s][2], t0[s][3], t1[s][0], t1[s][1], t1[s][2], t1[s][3]};
            pa[s] = Chunk<bf16_t>::pack(f);
        }
        const int r0 = kbase + lg * 4 + (l15 >> 2), r1 = r0 + 16;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int col = dt * 16 + (l15 & 3) * 4;
            const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(lds + sa_off<ROWB>(r0, col >> 3) + ((col & 4) << 1)));
            const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(lds + sa_off<ROWB>(r1, col >> 3) + ((col & 4) << 1)));
            const u32x2 a = __builtin_bit_cast(u32x2, lo), b = __builtin_bit_cast(u32x2, hi);
            const u32x4 vb = {a[0], a[1], b[0], b[1]};
#pragma unroll
            for (int s = 0; s < NS; ++s)
                acc[s][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, pa[s]),
                                                                    __builtin_bit_cast(bf16x8, vb), acc[s][dt], 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = kbase + half * 16 + lg * 4 + r;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const int d = dt * 16 + l15;
                    const float b = *reinterpret_cast<const float*>(lds + sa_off<ROWB>(row, d >> 2) + (d & 3) * 4);
#pragma unroll
                    for (int s = 0; s < NS; ++s)
                        acc[s][dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(half == 0 ? t0[s][r] : t1[s][r], b, acc[s][dt], 0, 0, 0);
                }
            }
    }
}

// XOR term of sa_off: the chunk stored at 16-byte position `pos` of row `row` is chunk pos ^ sa_swz(row)
template <int ROWB> DEVINL int sa_swz(int row);
template <> DEVINL int sa_swz<64>(int row) { const int q = (row >> 2) & 3; return ((q & 1) << 1) ^ ((q >> 1) * 3); }
template <> DEVINL int sa_swz<128>(int row) { return (row >> 1) & 7; }
template <> DEVINL int sa_swz<256>(int row) { return row & 15; }

// LDS-DMA as an assembly statement.  Through the builtin the compiler tracks the transfer as a pending LDS write and puts a
// vmcnt wait for it in front of every later LDS access it cannot tell apart from the destination -- the ds_read_tr intrinsics
// of the P.V / dS.K steps in particular: the r04 instruction streams showed s_waitcnt vmcnt(0..2) between the two MFMA groups
// of EVERY chunk, i.e. the next chunk's L2 round trip exposed in the middle of the current one (SQ_WAIT_ANY 27-61 % of the
// wavefront cycles).  The statement hides the LDS side; completion is what the kernels' own counted vmcnt + barrier at the top
// of a chunk establish (both carry a memory clobber).  The compiler's own vmcnt arithmetic stays safe: counters retire in
// order, an unknown older transfer only makes its waits longer than it thinks.  m0 = destination of the wavefront's 1 KiB
// piece; one wait state between writing m0 and the DMA reading it (LDS-DMA m0 hazard, gfx9).
typedef u32x4 sa_rsrc_t;
DEVINL sa_rsrc_t sa_make_rsrc(const void* g, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(g);
    return sa_rsrc_t{(unsigned)a, (unsigned)(a >> 32) & 0xffffu, bytes, 0x00020000u};
}
DEVINL unsigned sa_lds_addr(const void* lds) {
    typedef __attribute__((address_space(3))) const char lds_char;
    return (unsigned)reinterpret_cast<unsigned long long>((lds_char*)lds);
}
DEVINL void sa_dma16(const sa_rsrc_t& rs, unsigned lds_dst, unsigned voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_dst), "v"(voff), "s"(rs) : "memory", "m0");
}

// "These registers are final": an empty statement that takes them as operands.  The compiler finishes its own pending loads
// into them HERE -- before the chunk loop -- instead of at their first use inside it, where its s_waitcnt vmcnt(n) (n = the
// loads IT knows to be younger) would also wait for the DMA pieces it knows nothing about, in every chunk.
template <typename V> DEVINL void sa_settle(V& v) { asm volatile("" : "+v"(v)); }
template <typename V, int N> DEVINL void sa_settle(V (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) sa_settle(v[i]);
}

// A wavefront's [32 rows][D] result tile (C layout: a[t][dt][r] = row t*16 + lg*4 + r, column dt*16 + l15) to global rows of
// stride rs.  Straight from the accumulators that is 8 * D/16 stores of ONE element per lane (16 lanes = 32 contiguous
// bytes): store-issue bound, and for the 3-4 chunk problems (ViT, SAM windows, DETR decoder) a visible share of a workgroup's
// life.  Through a wave-private LDS tile (pitch D * sizeof(T) + 16 bytes: the four row groups of a write land on different
// banks) every lane stores 16 bytes, a row's bytes contiguous: D * sizeof(T) / 32 store instructions per wavefront.
template <typename T, int D>
DEVINL void sa_store_tile(const f32x4 (&a)[2][D / 16], char* stage, T* g, long rs, int row0, int nvalid, int lane) {
    constexpr int ROWB = D * (int)sizeof(T), PITCH = ROWB + 16, CPR = ROWB / 16;      // 16-byte chunks per row
    const int l15 = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int dt = 0; dt < D / 16; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                *reinterpret_cast<T*>(stage + (t * 16 + lg * 4 + r) * PITCH + (dt * 16 + l15) * (int)sizeof(T)) = from_f32<T>(a[t][dt][r]);
    __builtin_amdgcn_s_waitcnt(0xc07f);                        // lgkmcnt(0): the wavefront's own writes have landed
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 32 * CPR / 64; ++it) {
        const int i = it * 64 + lane, row = i / CPR, c = i - row * CPR;
        const u32x4 v = ld_chunk(stage + row * PITCH + c * 16);
        if (row0 + row < nvalid) st_chunk(reinterpret_cast<char*>(g + (size_t)(row0 + row) * rs) + c * 16, v);
    }
}
template <typename T, int D> constexpr int sa_stage_bytes() { return 32 * (D * (int)sizeof(T) + 16); }      // per wavefront; 4 of them fit every kernel's K / V (Q / dO) buffers

// the 4-byte form: one dword per lane (256 bytes per wavefront piece) -- a strided column of an fp32 table into LDS
DEVINL void sa_dma4(const sa_rsrc_t& rs, unsigned lds_dst, unsigned voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" ::"s"(lds_dst), "v"(voff), "s"(rs) : "memory", "m0");
}

template <typename T, int D>
struct SA {
    static constexpr int EPC = ElemTraits<T>::EPC;
    static constexpr int ROWB = D * (int)sizeof(T);
    static constexpr int DCH = ROWB / 16;
    static constexpr int STEPS = DCH / 4;              // MFMA k-steps over d
    static constexpr int DT = D / 16;                  // 16-wide output tiles over d
    static constexpr int CHUNK_BYTES = SA_CHUNK * ROWB;
    static constexpr int NLD = SA_CHUNK * DCH / SA_THREADS;   // 16-byte loads per thread per chunk

    // one 64-row chunk in flight: global -> registers (load), registers -> LDS image (store)
    struct Stager {
        u32x4 r[NLD];
        DEVINL void load(const T* __restrict__ g, long rs, int r0, int nvalid) {
#pragma unroll
            for (int j = 0; j < NLD; ++j) {
                const int i = threadIdx.x + j * SA_THREADS;
                const int row = i / DCH, c = i - row * DCH;
                r[j] = (r0 + row) < nvalid ? ld_chunk(g + (size_t)(r0 + row) * rs + c * EPC) : zero_chunk();
            }
        }
        DEVINL void store(char* lds) const {
#pragma unroll
            for (int j = 0; j < NLD; ++j) {
                const int i = threadIdx.x + j * SA_THREADS;
                const int row = i / DCH, c = i - row * DCH;
                st_chunk(lds + sa_off<ROWB>(row, c), r[j]);
            }
        }
    };
    // the same chunk global -> LDS by DMA (no staging registers): the DMA fills wave-linear 16-byte slots, so the
    // image's XOR swizzle is applied to the SOURCE address.  Rows >= nvalid come back as zeros through the buffer bounds
    // alone: the descriptor ends behind row nvalid - 1 (rs >= D), every later row starts past it -- no select per piece
    // (the host checks that the whole operand stays below 2 GiB, so the 32-bit offsets cannot wrap)
    static DEVINL sa_rsrc_t rsrc(const T* g, long rs, int nvalid) {
        return sa_make_rsrc(g, (unsigned)(((size_t)(nvalid - 1) * rs + D) * sizeof(T)));
    }
    // per-lane byte offset of piece j inside a chunk that starts at row 0 (constant over the kernel: hoisted by the caller's loop)
    static DEVINL unsigned dma_lane_off(long rs, int j, int wave, int lane) {
        const int slot = (j * SA_WAVES + wave) * 64 + lane;
        const int row = slot / DCH, pos = slot - row * DCH;
        const int c = pos ^ sa_swz<ROWB>(row);
        return (unsigned)(((size_t)row * rs + c * EPC) * sizeof(T));
    }
    static DEVINL void dma(const sa_rsrc_t& r, char* lds, long rs, int r0, int nvalid, int wave, int lane) {
        const unsigned base = (unsigned)((size_t)r0 * rs * sizeof(T));          // wave-uniform
        const unsigned dst = sa_lds_addr(lds) + wave * 1024;
#pragma unroll
        for (int j = 0; j < NLD; ++j)
            sa_dma16(r, dst + j * SA_WAVES * 1024, dma_lane_off(rs, j, wave, lane) + base);      // (the vector offset is what the bounds check sees)
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
    // the same chain on top of a C operand the caller keeps in registers (a row constant such as -D costs nothing this way)
    static DEVINL f32x4 tile_c(const u32x4 (&a)[STEPS], const u32x4 (&b)[STEPS], f32x4 acc) {
#pragma unroll
        for (int s = 0; s < STEPS; ++s) Mma<T>::run(acc, a[s], b[s]);
        return acc;
    }
};

// per-wavefront LDS tables of the rows of rel_h / rel_w it owns, pre-multiplied by log2(e)
DEVINL void sa_load_tables(float* rh, float* rw, const SAParams& p, int bh, int q0, int lane) {
    for (int i = lane; i < SA_WROWS * p.Sh; i += 64) {
        const int r = i / p.Sh, c = i - r * p.Sh;
        rh[r * (p.Sh + 1) + c] = (q0 + r) < p.Nq ? p.rel_h[((size_t)bh * p.Nq + q0 + r) * p.Sh + c] * LOG2E : 0.f;
    }
    for (int i = lane; i < SA_WROWS * p.Sw; i += 64) {
        const int r = i / p.Sw, c = i - r * p.Sw;
        rw[r * (p.Sw + 1) + c] = (q0 + r) < p.Nq ? p.rel_w[((size_t)bh * p.Nq + q0 + r) * p.Sw + c] * LOG2E : 0.f;
    }
}

// key -> (kh, kw) without an integer division: (key + 0.5) / Sw is never within 0.5 / Sw of an integer,
// far outside the fp32 error for any key count this kernel can meet
DEVINL void sa_split_key(int key, float inv_sw, int Sw, int& kh, int& kw) {
    kh = (int)(((float)key + 0.5f) * inv_sw);
    kw = key - kh * Sw;
}

// REL 1 (SAM windows: Sh + Sw <= 32, Nk <= 256): the decomposed relative-position bias rides on the MFMA.
//   rel_h[q][kh(k)] + rel_w[q][kw(k)] = sum_e R[q][e] * E[k][e],   R[q] = [rel_h[q][0..Sh) | rel_w[q][0..Sw) | 0],
//   E[k][e] = 1 for e == kh(k) and e == Sh + kw(k), else 0
// i.e. 32 more columns of the q k^T contraction (one 16x16x32 bf16 MFMA per tile) instead of two LDS table reads
// and two adds per logit.  R is pre-divided by the softmax scale so the sum lands in the same accumulator as q k^T.
// bf16 mode rounds R to bf16 exactly as the reference's autocast einsum does; fp32 mode uses the exact-f32 MFMA.
template <typename T> struct SARel {
    static constexpr int EROWB = 32 * (int)sizeof(T);       // one row of E / R: 32 columns
    static constexpr int CPR = EROWB / 16;                  // 16-byte chunks per row
    static constexpr int ECH = CPR / 4;                     // MFMA k-steps (1 bf16, 2 f32)
    static constexpr int EBYTES = 256 * EROWB;
    // indicator matrix E[256 keys][32] into LDS (swizzled like every operand image)
    static DEVINL void build_E(char* Es, const SAParams& p) {
        const float inv_sw = 1.f / (float)p.Sw;
        for (int i = threadIdx.x; i < 256 * CPR; i += SA_THREADS) {
            const int row = i / CPR, ch = i - row * CPR;
            int kh, kw;
            sa_split_key(row, inv_sw, p.Sw, kh, kw);
            float f[Chunk<T>::N];
#pragma unroll
            for (int j = 0; j < Chunk<T>::N; ++j) {
                const int col = ch * Chunk<T>::N + j;
                f[j] = (row < p.Nk && (col == kh || col == p.Sh + kw)) ? 1.f : 0.f;
            }
            st_chunk(Es + sa_off<EROWB>(row, ch), Chunk<T>::pack(f));
        }
    }
    // the same rows for ONE key, straight into an operand fragment (chunk index ch)
    static DEVINL u32x4 key_frag(const SAParams& p, int key, int ch) {
        int kh, kw;
        sa_split_key(key, 1.f / (float)p.Sw, p.Sw, kh, kw);
        float f[Chunk<T>::N];
#pragma unroll
        for (int j = 0; j < Chunk<T>::N; ++j) {
            const int col = ch * Chunk<T>::N + j;
            f[j] = (key < p.Nk && (col == kh || col == p.Sh + kw)) ? 1.f : 0.f;
        }
        return Chunk<T>::pack(f);
    }
    // chunk ch of R[q] (scaled) from the fp32 rel_h / rel_w tensors
    static DEVINL u32x4 row_frag(const SAParams& p, int bh, int q, int ch, float mul) {
        float f[Chunk<T>::N];
#pragma unroll
        for (int j = 0; j < Chunk<T>::N; ++j) {
            const int col = ch * Chunk<T>::N + j;
            float v = 0.f;
            if (q < p.Nq) {
                if (col < p.Sh) v = p.rel_h[((size_t)bh * p.Nq + q) * p.Sh + col];
                else if (col < p.Sh + p.Sw) v = p.rel_w[((size_t)bh * p.Nq + q) * p.Sw + col - p.Sh];
            }
            f[j] = v * mul;
        }
        return Chunk<T>::pack(f);
    }
};

// ------------------------------------------------------------------------------------ forward
template <typename T, int D, int REL, bool DROP, bool KB>
// (256, 2): two workgroups per CU; the backward kernels spill under that bound and are faster at one
__global__ __launch_bounds__(SA_THREADS, 2) void sa_fwd_kernel(const SAParams p) {
    using S = SA<T, D>;
    constexpr bool TAB = REL == 3;         // per-wave LDS tables + VALU adds (generic table widths)
    constexpr bool EMM = REL == 1;         // small tables: the bias rides on the MFMA (SARel)
    using RL = SARel<T>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // XCD-aware mapping (workgroups are dealt to the 8 XCDs round-robin in launch order, x fastest): every row block
    // of one (batch, head) reads the same K / V (Q / dO), so each XCD gets a contiguous (head, block) range
    const int sa_lid = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int bh = sa_lid / (int)gridDim.x, blk = sa_lid - bh * (int)gridDim.x;
    const int b = bh / p.H, h = bh - b * p.H;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l15 = lane & 15, lg = lane >> 4;
    char* KV = smem;                                      // [2 buffers][K chunk | V chunk], filled by DMA
    float* rh = reinterpret_cast<float*>(smem + 4 * S::CHUNK_BYTES) + wave * SA_WROWS * (p.Sh + p.Sw + 2);
    float* rw = rh + SA_WROWS * (p.Sh + 1);
    const T* qg = (const T*)p.q + (size_t)b * p.q_bs + h * D;
    const T* kg = (const T*)p.k + (size_t)b * p.k_bs + h * D;
    const T* vg = (const T*)p.v + (size_t)b * p.v_bs + h * D;
    // DETR: the key-bias row (* log2 e, zero-padded to whole chunks) goes to LDS once -- a global load inside the chunk loop
    // would be waited for behind the next chunk's DMA (vmcnt is in order); published by the first barrier of the loop
    const float* kbs = reinterpret_cast<const float*>(smem + 4 * S::CHUNK_BYTES);
    if constexpr (KB) {
        static_assert(REL == 0 || REL == 2, "key bias: plain and Sw == 64 forms");
        float* w = reinterpret_cast<float*>(smem + 4 * S::CHUNK_BYTES);
        const float* kbg = p.key_bias + (size_t)b * p.Nk;
        const int padded = (p.Nk + SA_CHUNK - 1) / SA_CHUNK * SA_CHUNK;
        for (int i = threadIdx.x; i < padded; i += SA_THREADS) w[i] = i < p.Nk ? kbg[i] * LOG2E : 0.f;
    }
    const int q0 = blk * SA_BROWS + wave * SA_WROWS;
    const bool live = q0 < p.Nq;                              // wave-uniform: a wavefront without a valid row only feeds the stream
    if constexpr (TAB) sa_load_tables(rh, rw, p, bh, q0, lane);
    char* Es = smem + 4 * S::CHUNK_BYTES;                 // EMM: indicator matrix, published by the loop's first barrier
    u32x4 rf[2][RL::ECH];
    if constexpr (EMM) {
        RL::build_E(Es, p);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int e = 0; e < RL::ECH; ++e) rf[qt][e] = RL::row_frag(p, bh, q0 + qt * 16 + l15, e * 4 + lg, 1.f / p.scale);
    }
    f32x4 rwq[2][4];                                          // REL 2: rel_w[q][kt*16 + lg*4 + 0..3] * log2(e)
    if constexpr (REL == 2) {
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int q = q0 + qt * 16 + l15;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (q < p.Nq) v = *reinterpret_cast<const f32x4*>(p.rel_w + ((size_t)bh * p.Nq + q) * 64 + kt * 16 + lg * 4);
                rwq[qt][kt] = v * LOG2E;
            }
        }
    }
    u32x4 qf[2][S::STEPS];
    S::gmem_frags(qf[0], qg, p.q_rs, q0, p.Nq, l15, lg);
    S::gmem_frags(qf[1], qg, p.q_rs, q0 + 16, p.Nq, l15, lg);
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
    f32x4 o[2][S::DT];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int dt = 0; dt < S::DT; ++dt) o[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float c2 = p.scale * LOG2E;
    constexpr bool drop = DROP;      // compiled out of the relative-position instantiations (register budget)
    const unsigned dthresh = sa_thresh(p.dropout_p);
    // (r06) a captured step freezes p.seed; the varying part lives in device memory and is bumped by the host before every replay
    const unsigned dseed = (DROP && p.seed_device != nullptr) ? p.seed + *p.seed_device : p.seed;
    const float inv_keep = drop ? 1.f / (1.f - p.dropout_p) : 1.f;
    const float inv_sw = TAB ? 1.f / (float)p.Sw : 0.f;
    const sa_rsrc_t k_rsrc = S::rsrc(kg, p.k_rs, p.Nk), v_rsrc = S::rsrc(vg, p.v_rs, p.Nk);
    S::dma(k_rsrc, KV, p.k_rs, 0, p.Nk, wave, lane);
    S::dma(v_rsrc, KV + S::CHUNK_BYTES, (D == 32 ? p.v_rs : p.k_rs), 0, p.Nk, wave, lane);
    float rhn[2] = {0.f, 0.f};
    if constexpr (REL == 2) {
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int q = q0 + qt * 16 + l15;
            if (q < p.Nq) rhn[qt] = p.rel_h[((size_t)bh * p.Nq + q) * p.Sh];
        }
    }
    sa_settle(qf);
    if constexpr (EMM) sa_settle(rf);
    if constexpr (REL == 2) sa_settle(rwq);

    for (int k0 = 0; k0 < p.Nk; k0 += SA_CHUNK) {
        const int buf = (k0 / SA_CHUNK) & 1;
        const char* Ks = KV + buf * 2 * S::CHUNK_BYTES;
        const char* Vs = Ks + S::CHUNK_BYTES;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this chunk has landed (own DMA) ...
        if constexpr (REL == 2) sa_settle(rhn);
        __syncthreads();                                      // ... for every wavefront; the other buffer is free again
        if (k0 + SA_CHUNK < p.Nk) {                           // next chunk streams in under this one's math
            char* nxt = KV + (buf ^ 1) * 2 * S::CHUNK_BYTES;
            S::dma(k_rsrc, nxt, p.k_rs, k0 + SA_CHUNK, p.Nk, wave, lane);
            S::dma(v_rsrc, nxt + S::CHUNK_BYTES, (D == 32 ? p.v_rs : p.k_rs), k0 + SA_CHUNK, p.Nk, wave, lane);
        }
        // REL 2: this chunk's rel_h value was fetched during the previous chunk (a load issued here and used
        // right away would also wait for the DMA of the NEXT chunk: vmcnt retires in order)
        float rhc[2] = {rhn[0] * LOG2E, rhn[1] * LOG2E};
        if constexpr (REL == 2) {
            if (k0 + SA_CHUNK < p.Nk) {
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) {
                    const int q = q0 + qt * 16 + l15;
                    if (q < p.Nq) rhn[qt] = p.rel_h[((size_t)bh * p.Nq + q) * p.Sh + (k0 >> 6) + 1];
                }
            }
        }
        if (live) {
        f32x4 st[2][4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            u32x4 kf[S::STEPS];
            S::lds_frags(kf, Ks, kt * 16, l15, lg);
            st[0][kt] = S::tile(kf, qf[0]);
            st[1][kt] = S::tile(kf, qf[1]);
            if constexpr (EMM) {
#pragma unroll
                for (int e = 0; e < RL::ECH; ++e) {
                    const u32x4 ef = ld_chunk(Es + sa_off<RL::EROWB>(k0 + kt * 16 + l15, e * 4 + lg));
                    Mma<T>::run(st[0][kt], ef, rf[0][e]);
                    Mma<T>::run(st[1][kt], ef, rf[1][e]);
                }
            }
        }
        const bool tail = k0 + SA_CHUNK > p.Nk;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = k0 + kt * 16 + lg * 4 + r;
                float kbv = 0.f;
                if constexpr (KB) kbv = kbs[key];
                int kh = 0, kw = 0;
                if constexpr (TAB) sa_split_key(key, inv_sw, p.Sw, kh, kw);
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) {
                    float bias = kbv;
                    if constexpr (REL == 2) bias += rhc[qt] + rwq[qt][kt][r];
                    if constexpr (TAB) bias += rh[(qt * 16 + l15) * (p.Sh + 1) + kh] + rw[(qt * 16 + l15) * (p.Sw + 1) + kw];
                    float s = st[qt][kt][r] * c2 + bias;
                    if (tail && key >= p.Nk) s = -INFINITY;
                    st[qt][kt][r] = s;
                }
            }
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[qt][kt][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run[qt], mx);
            const float alpha = fast_exp2(m_run[qt] - m_new);       // first chunk: exp2(-inf) = 0
            m_run[qt] = m_new;
            float psum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    st[qt][kt][r] = fast_exp2(st[qt][kt][r] - m_new);
                    psum += st[qt][kt][r];
                }
            l_run[qt] = l_run[qt] * alpha + psum;         // per-lane partial (alpha is equal on all lane groups)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float aq = __shfl(alpha, lg * 4 + r, 64);     // O rows are queries lg*4 + r
#pragma unroll
                for (int dt = 0; dt < S::DT; ++dt) o[qt][dt][r] *= aq;
            }
        }
        if (drop) {
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        st[qt][kt][r] = sa_keep(dseed, bh, q0 + qt * 16 + l15, k0 + kt * 16 + lg * 4 + r, p.Nk, dthresh)
                                            ? st[qt][kt][r] * inv_keep : 0.f;
        }
        {
            const f32x4 a0[2] = {st[0][0], st[1][0]}, a1[2] = {st[0][1], st[1][1]};
            pvN<T, S::ROWB, S::DT, 2>(o, a0, a1, Vs, 0, l15, lg);
            const f32x4 b0[2] = {st[0][2], st[1][2]}, b1[2] = {st[0][3], st[1][3]};
            pvN<T, S::ROWB, S::DT, 2>(o, b0, b1, Vs, 32, l15, lg);
        }
        }
    }
    T* og = (T*)p.out + (size_t)b * p.o_bs + h * D;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        float l = l_run[qt];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const int qrow = q0 + qt * 16 + l15;
        if (lg == 0 && qrow < p.Nq) p.lse[(size_t)bh * p.Nq + qrow] = (m_run[qt] + log2f(l)) * LN2;
        const float inv = 1.f / l;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float iq = __shfl(inv, lg * 4 + r, 64);
            const int q = q0 + qt * 16 + lg * 4 + r;
            if (q < p.Nq) {
#pragma unroll
                for (int dt = 0; dt < S::DT; ++dt) og[(size_t)q * p.o_rs + dt * 16 + l15] = from_f32<T>(o[qt][dt][r] * iq);
            }
        }
    }
}

// ------------------------------------------------------------------------------------ forward, r04 structure (REL 0 / 2, no dropout)
// Same tiling as sa_fwd_kernel (four wavefronts x 32 queries, 64-key chunks, S^T accumulators) with what the r04 profile said
// the r03 kernel waits for taken off the chunk's critical path:
//  * the K / V stream is a THREE-slot LDS-DMA ring with a counted vmcnt: two chunks are in flight while one is consumed (one
//    chunk of lead did not cover an L2 round trip under load: 0.14-0.16 of the MFMA peak with the matrix cores idle most of a
//    chunk) -- and no ordinary global load is left inside the loop (the compiler waits vmcnt(0) for those, which drains the
//    ring): the rel_h rows of the block's 128 queries are staged in LDS once (REL 2), the key-bias row as before (REL 0);
//  * the reductions over the four lane groups (row maximum, final normaliser) are v_permlane16_swap / v_permlane32_swap
//    (two VALU instructions each) instead of ds_bpermute round trips through the LDS crossbar;
//  * the running maximum moves only when a row's maximum grew by more than 2^RESCALE_LOG2 (guide T13): P stays <= 256, and
//    the per-chunk broadcast of alpha to the O layout (eight more ds_bpermute) plus 32 multiplies happen on the first chunks
//    only -- a wave-uniform branch the other chunks skip.
// Numerics: P = exp2(s - m) with the same m in the normaliser and in P.V, so the result is the softmax for any m <= max + 8;
// bf16 P carries 8 bits either way.
constexpr float SA_RESCALE_LOG2 = 8.f;
constexpr int SA_RING = 3;

// op over lanes l, l^16, l^32, l^48 in registers (gfx950 permlane swaps; both halves of a swap hold the pair's two values)
DEVINL float sa_lg_max(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
DEVINL float sa_lg_sum(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

template <typename T, int D, int REL, bool KB>
__global__ __launch_bounds__(SA_THREADS, 3) void sa_fwd2_kernel(const SAParams p) {      // (<= 168 registers: three workgroups per CU)
    using S = SA<T, D>;
    static_assert(REL == 0 || (REL <= 2 && !KB), "r04 forward: no bias / key bias (REL 0), window tables on the MFMA (REL 1) or "
                                                 "the 64-wide decomposed bias (REL 2)");
    using RL = SARel<T>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int sa_lid = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int bh = sa_lid / (int)gridDim.x, blk = sa_lid - bh * (int)gridDim.x;
    const int b = bh / p.H, h = bh - b * p.H;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l15 = lane & 15, lg = lane >> 4;
    // ring stage: K chunk | V chunk | REL 2: the chunk's rel_h column of the block's 128 queries (1 KiB: every wavefront moves one
    // 256-byte piece so that all four count the same number of transfers; wavefronts 2, 3 deposit duplicates nobody reads)
    constexpr int RHB = REL == 2 ? 4 * 256 : 0;
    constexpr int STAGE = 2 * S::CHUNK_BYTES + RHB;
    constexpr int PIECES = 2 * S::NLD + (REL == 2 ? 1 : 0);   // transfers per wavefront and chunk
    char* KV = smem;                                          // [SA_RING][K | V | rel_h column]
    float* aux = reinterpret_cast<float*>(smem + SA_RING * STAGE);      // REL 0: key bias
    char* Es = smem + SA_RING * STAGE;                        // REL 1: key -> (kh, kw) indicator matrix [256][32] (SARel)
    const T* qg = (const T*)p.q + (size_t)b * p.q_bs + h * D;
    const T* kg = (const T*)p.k + (size_t)b * p.k_bs + h * D;
    const T* vg = (const T*)p.v + (size_t)b * p.v_bs + h * D;
    const int q0 = blk * SA_BROWS + wave * SA_WROWS;
    const bool live = q0 < p.Nq;                              // wave-uniform
    const sa_rsrc_t k_rsrc = S::rsrc(kg, p.k_rs, p.Nk), v_rsrc = S::rsrc(vg, p.v_rs, p.Nk);
    const int nchunk = (p.Nk + SA_CHUNK - 1) / SA_CHUNK;
    // REL 2: rel_h[q][kh] travels with the ring -- column ci of the block's rows, one dword per lane (rows past Nq: out of bounds
    // -> zeros).  r04 first staged all 64 columns of the 128 rows in LDS (32 KiB: 80 KiB per workgroup, two per CU); with the column
    // in the ring the workgroup needs 51 KiB and a third one fits (the kernel holds 167 registers)
    const sa_rsrc_t rh_rsrc = sa_make_rsrc(REL == 2 ? (const void*)(p.rel_h + (size_t)bh * p.Nq * p.Sh) : (const void*)p.lse,
                                           REL == 2 ? (unsigned)((size_t)p.Nq * p.Sh * sizeof(float)) : 0u);
    const unsigned rh_lane = (unsigned)(((size_t)(blk * SA_BROWS + (wave & 1) * 64 + lane) * p.Sh) * sizeof(float));
    auto dma_rh = [&](int ci, int s) {
        if constexpr (REL == 2) sa_dma4(rh_rsrc, sa_lds_addr(KV + s * STAGE + 2 * S::CHUNK_BYTES) + wave * 256, rh_lane + (unsigned)ci * 4u);
    };
    // ring prologue first: the two chunks stream in under the rest of the set-up
    S::dma(k_rsrc, KV, p.k_rs, 0, p.Nk, wave, lane);
    S::dma(v_rsrc, KV + S::CHUNK_BYTES, (D == 32 ? p.v_rs : p.k_rs), 0, p.Nk, wave, lane);
    dma_rh(0, 0);
    if (nchunk > 1) {
        S::dma(k_rsrc, KV + STAGE, p.k_rs, SA_CHUNK, p.Nk, wave, lane);
        S::dma(v_rsrc, KV + STAGE + S::CHUNK_BYTES, (D == 32 ? p.v_rs : p.k_rs), SA_CHUNK, p.Nk, wave, lane);
        dma_rh(1, 1);
    }
    const float* kb = aux;
    if constexpr (KB) {                                       // host: Nk <= SA_KB_LDS on this path
        const float* kbg = p.key_bias + (size_t)b * p.Nk;
        for (int i = threadIdx.x; i < nchunk * SA_CHUNK; i += SA_THREADS) aux[i] = i < p.Nk ? kbg[i] * LOG2E : 0.f;
    }
    u32x4 rf[2][RL::ECH];
    if constexpr (REL == 1) {
        RL::build_E(Es, p);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int e = 0; e < RL::ECH; ++e) rf[qt][e] = RL::row_frag(p, bh, q0 + qt * 16 + l15, e * 4 + lg, 1.f / p.scale);
    }
    f32x4 rwq[2][4];                                          // REL 2: rel_w[q][kt*16 + lg*4 + 0..3] * log2(e)
    if constexpr (REL == 2) {
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int q = q0 + qt * 16 + l15;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (q < p.Nq) v = *reinterpret_cast<const f32x4*>(p.rel_w + ((size_t)bh * p.Nq + q) * 64 + kt * 16 + lg * 4);
                rwq[qt][kt] = v * LOG2E;
            }
        }
    }
    u32x4 qf[2][S::STEPS];
    S::gmem_frags(qf[0], qg, p.q_rs, q0, p.Nq, l15, lg);
    S::gmem_frags(qf[1], qg, p.q_rs, q0 + 16, p.Nq, l15, lg);
    float m_run[2] = {-INFINITY, -INFINITY};
    f32x4 l_run[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    f32x4 o[2][S::DT];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int dt = 0; dt < S::DT; ++dt) o[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float c2 = p.scale * LOG2E;
    // every ordinary load of the prologue has been consumed into registers / LDS before the first counted wait below
    sa_settle(qf);
    if constexpr (REL == 1) sa_settle(rf);
    if constexpr (REL == 2) sa_settle(rwq);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (REL == 1 || KB) __syncthreads();            // the LDS tables written above are complete for every wavefront

    int slot = 0;                                             // ring slot of the chunk being consumed
    for (int ci = 0; ci < nchunk; ++ci) {
        const int k0 = ci * SA_CHUNK;
        const char* Ks = KV + slot * STAGE;
        const char* Vs = Ks + S::CHUNK_BYTES;
        // chunk ci has landed: behind it at most chunk ci + 1 (2 * NLD loads per thread) is still in flight
        if (ci + 1 < nchunk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                         // ... for every wavefront; chunk ci - 1's slot is free again
        if (ci + 2 < nchunk) {
            const int ns = slot == 0 ? 2 : slot - 1;          // (slot + 2) % 3
            S::dma(k_rsrc, KV + ns * STAGE, p.k_rs, k0 + 2 * SA_CHUNK, p.Nk, wave, lane);
            S::dma(v_rsrc, KV + ns * STAGE + S::CHUNK_BYTES, (D == 32 ? p.v_rs : p.k_rs), k0 + 2 * SA_CHUNK, p.Nk, wave, lane);
            dma_rh(ci + 2, ns);
        }
        if (live) {                                           // (a wavefront whose 32 rows all lie past Nq only feeds the ring)
        f32x4 st[2][4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            u32x4 kf[S::STEPS];
            S::lds_frags(kf, Ks, kt * 16, l15, lg);
            st[0][kt] = S::tile(kf, qf[0]);
            st[1][kt] = S::tile(kf, qf[1]);
            if constexpr (REL == 1) {                         // + R[q] . E[key]: the table values join the same accumulator
#pragma unroll
                for (int e = 0; e < RL::ECH; ++e) {
                    const u32x4 ef = ld_chunk(Es + sa_off<RL::EROWB>(k0 + kt * 16 + l15, e * 4 + lg));
                    Mma<T>::run(st[0][kt], ef, rf[0][e]);
                    Mma<T>::run(st[1][kt], ef, rf[1][e]);
                }
            }
        }
        const bool tail = REL != 2 && k0 + SA_CHUNK > p.Nk;   // wave-uniform: only the last chunk pays for the key mask (REL 2: Nk = Sh * 64)
        float rowb[2] = {0.f, 0.f};                           // REL 2: the chunk is one kh row -> one bias value per query
        if constexpr (REL == 2) {
            const float* rhs = reinterpret_cast<const float*>(Ks + 2 * S::CHUNK_BYTES) + wave * SA_WROWS + l15;
            rowb[0] = rhs[0] * LOG2E;
            rowb[1] = rhs[16] * LOG2E;
        }
        // The elementwise work between the two MFMA groups is what the kernel is bound by (one VALU instruction is four
        // cycles per wavefront, an MFMA sixteen): it is written on float4 values so that the multiply-adds, the running sum
        // and the exponent offsets issue as packed fp32 instructions (two elements each), and the log2-domain scale rides
        // on the instruction that adds the bias / the offset instead of being a pass of its own.
        constexpr bool plain = REL != 2 && !KB;               // logits stay unscaled until the exponent's fma
        const f32x4 c2v = {c2, c2, c2, c2};
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            if (!plain) {
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    f32x4 bq;
                    if constexpr (REL == 2) bq = rwq[qt][kt];
                    else bq = *reinterpret_cast<const f32x4*>(kb + k0 + kt * 16 + lg * 4);      // (aux is padded to the chunk)
                    st[qt][kt] = st[qt][kt] * c2v + bq;
                }
            }
            if (tail) {                                       // a real branch: the other chunks pay neither selects nor indices
                int kb0 = k0 + lg * 4;
                asm volatile("" : "+v"(kb0));                 // (opaque: keeps the index arithmetic inside the branch)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kb0 + kt * 16 + r >= p.Nk) st[qt][kt][r] = -INFINITY;
            }
            float mx = fmaxf(fmaxf(st[qt][0][0], st[qt][0][1]), st[qt][0][2]);       // v_max3_f32 chain
            mx = fmaxf(fmaxf(mx, st[qt][0][3]), st[qt][1][0]);
            mx = fmaxf(fmaxf(mx, st[qt][1][1]), st[qt][1][2]);
            mx = fmaxf(fmaxf(mx, st[qt][1][3]), st[qt][2][0]);
            mx = fmaxf(fmaxf(mx, st[qt][2][1]), st[qt][2][2]);
            mx = fmaxf(fmaxf(mx, st[qt][2][3]), st[qt][3][0]);
            mx = fmaxf(fmaxf(mx, st[qt][3][1]), st[qt][3][2]);
            mx = fmaxf(mx, st[qt][3][3]);
            if (plain) mx *= c2;                              // c2 > 0: the maximum commutes with the scale
            mx = sa_lg_max(mx) + rowb[qt];
            // deferred rescale: the reference point moves only when this row's maximum outgrew it by 2^8
            const bool grow = mx > m_run[qt] + SA_RESCALE_LOG2;
            if (__builtin_amdgcn_ballot_w64(grow) != 0) {    // wave-uniform: rare after the first chunks
                const float m_new = grow ? mx : m_run[qt];
                const float alpha = fast_exp2(m_run[qt] - m_new);           // first chunk: exp2(-inf) = 0
                m_run[qt] = m_new;
                l_run[qt] *= alpha;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float aq = __shfl(alpha, lg * 4 + r, 64);         // O rows are queries lg*4 + r
#pragma unroll
                    for (int dt = 0; dt < S::DT; ++dt) o[qt][dt][r] *= aq;
                }
            }
            const float off = rowb[qt] - m_run[qt];
            const f32x4 offv = {off, off, off, off};
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const f32x4 x = plain ? st[qt][kt] * c2v + offv : st[qt][kt] + offv;
                const f32x4 e = {fast_exp2(x[0]), fast_exp2(x[1]), fast_exp2(x[2]), fast_exp2(x[3])};
                st[qt][kt] = e;
                l_run[qt] += e;                               // per-lane partials (four of them); combined at the end
            }
        }
        {
            const f32x4 a0[2] = {st[0][0], st[1][0]}, a1[2] = {st[0][1], st[1][1]};
            pvN<T, S::ROWB, S::DT, 2>(o, a0, a1, Vs, 0, l15, lg);
            const f32x4 b0[2] = {st[0][2], st[1][2]}, b1[2] = {st[0][3], st[1][3]};
            pvN<T, S::ROWB, S::DT, 2>(o, b0, b1, Vs, 32, l15, lg);
        }
        }
        slot = slot == 2 ? 0 : slot + 1;
    }
    T* og = (T*)p.out + (size_t)b * p.o_bs + h * D;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const float l = sa_lg_sum((l_run[qt][0] + l_run[qt][1]) + (l_run[qt][2] + l_run[qt][3]));
        const int qrow = q0 + qt * 16 + l15;
        if (lg == 0 && qrow < p.Nq) p.lse[(size_t)bh * p.Nq + qrow] = (m_run[qt] + log2f(l)) * LN2;
        const float inv = 1.f / l;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float iq = __shfl(inv, lg * 4 + r, 64);
#pragma unroll
            for (int dt = 0; dt < S::DT; ++dt) o[qt][dt][r] *= iq;
        }
    }
    __syncthreads();                                          // every wavefront is done with the ring: it becomes the output stage
    if (live) sa_store_tile<T, D>(o, smem + wave * sa_stage_bytes<T, D>(), og, p.o_rs, q0, p.Nq, lane);
}

// ------------------------------------------------------------------------------------ backward: dQ (+ D, d rel-pos)
// LDS: K chunk | V chunk | REL 1: indicator [256][32] | REL 1/3: per-wave tables | REL 3: per-wave gradient tables
//
// r04: the chunk's elementwise work (32 logits per lane) was what the kernel waited for -- 417 vector instructions per chunk
// against 48 MFMAs (4 vs 16 cycles each).  Now: -D rides in the C operand of the dO.V^T MFMAs; the log2-domain scale, the
// row's -lse and the bias are ONE packed fma per two logits; the key mask is a branch only the last chunk takes; rows past
// Nq need no mask at all (their Q / dO fragments, D and lse are zeros: P = exp2(bias) is finite and dS = P * (0 - 0) = 0);
// the key bias is a template switch instead of a pointer test per logit.
template <typename T, int D, int REL, bool DROP, bool KB>
// every instantiation fits 256 VGPRs -- two workgroups per CU and VGPR-form MFMAs (no AGPR <-> VGPR copies around the
// short-lived S / dP tiles); the plain form is held to 168 so that THREE wavefronts share a SIMD: the loop is bound by the
// vector issue port (an MFMA holds it four slots, anything else one), and a third wavefront is what keeps the port fed
__global__ __launch_bounds__(SA_THREADS, (REL == 0 && !DROP && !KB && sizeof(T) == 2 && D == 64) ? 3 : 2) void sa_bwd_dq_kernel(const SAParams p) {
    using S = SA<T, D>;
    constexpr bool TAB = REL == 3;         // per-wave LDS tables + VALU adds
    constexpr bool EMM = REL == 1;         // bias on the MFMA (SARel), gradients through the same indicator matrix
    using RL = SARel<T>;
    constexpr int EROWB = RL::EROWB;
    constexpr int EBYTES = REL == 1 ? RL::EBYTES : 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // XCD-aware mapping (workgroups are dealt to the 8 XCDs round-robin in launch order, x fastest): every row block
    // of one (batch, head) reads the same K / V (Q / dO), so each XCD gets a contiguous (head, block) range
    const int sa_lid = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int bh = sa_lid / (int)gridDim.x, blk = sa_lid - bh * (int)gridDim.x;
    const int b = bh / p.H, h = bh - b * p.H;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l15 = lane & 15, lg = lane >> 4;
    char* KV = smem;                                      // [2 buffers][K chunk | V chunk], filled by DMA
    char* Es = smem + 4 * S::CHUNK_BYTES;
    const int tabw = SA_WROWS * (p.Sh + p.Sw + 2);
    float* tabs = reinterpret_cast<float*>(smem + 4 * S::CHUNK_BYTES + EBYTES);
    float* rh = tabs + wave * tabw * (REL == 3 ? 2 : 1);
    float* rw = rh + SA_WROWS * (p.Sh + 1);
    float* gh = rh + tabw;                                // REL 3 only
    float* gw = gh + SA_WROWS * (p.Sh + 1);
    // REL 2: d rel_w accumulates in 32 registers per lane (the kernel sits at two wavefronts per SIMD with or without them;
    // the r03 LDS read-modify-write of the same tile doubled the kernel's LDS traffic -- SQ_LDS_IDX_ACTIVE 203 vs 101 M)
    const T* qg = (const T*)p.q + (size_t)b * p.q_bs + h * D;
    const T* kg = (const T*)p.k + (size_t)b * p.k_bs + h * D;
    const T* vg = (const T*)p.v + (size_t)b * p.v_bs + h * D;
    const T* og = (const T*)p.out + (size_t)b * p.o_bs + h * D;
    const T* dog = (const T*)p.dout + (size_t)b * p.o_bs + h * D;
    const float* kbs = reinterpret_cast<const float*>(smem + 4 * S::CHUNK_BYTES);
    if constexpr (KB) {              // key-bias row (* log2 e, zero-padded to whole chunks) in LDS (see the forward kernel)
        static_assert(REL == 0 || REL == 2, "key bias: plain and Sw == 64 forms");
        float* w = reinterpret_cast<float*>(smem + 4 * S::CHUNK_BYTES);
        const float* kbg = p.key_bias + (size_t)b * p.Nk;
        const int padded = (p.Nk + SA_CHUNK - 1) / SA_CHUNK * SA_CHUNK;
        for (int i = threadIdx.x; i < padded; i += SA_THREADS) w[i] = i < p.Nk ? kbg[i] * LOG2E : 0.f;
    }
    const int q0 = blk * SA_BROWS + wave * SA_WROWS;
    const bool live = q0 < p.Nq;                              // wave-uniform: a wavefront without a valid row only feeds the stream
    const float inv_sw = TAB ? 1.f / (float)p.Sw : 0.f;
    if constexpr (TAB) sa_load_tables(rh, rw, p, bh, q0, lane);
    if constexpr (REL == 3) {
        for (int i = lane; i < tabw; i += 64) gh[i] = 0.f;
    }
    u32x4 rf[2][RL::ECH];
    if constexpr (EMM) {
        RL::build_E(Es, p);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int e = 0; e < RL::ECH; ++e) rf[qt][e] = RL::row_frag(p, bh, q0 + qt * 16 + l15, e * 4 + lg, 1.f / p.scale);
    }
    f32x4 rwq[2][4];                                          // REL 2: rel_w[q][kt*16 + lg*4 + 0..3] * log2(e)
    f32x4 gwq[2][4];                                          // REL 2: d rel_w of the same positions
    f32x2 ghq[2];                                             // REL 2: d rel_h of 8 chunks: lane group c >> 1 keeps chunk c in [c & 1]
    if constexpr (REL == 2) {
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            ghq[qt] = f32x2{0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) gwq[qt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int q = q0 + qt * 16 + l15;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (q < p.Nq) v = *reinterpret_cast<const f32x4*>(p.rel_w + ((size_t)bh * p.Nq + q) * 64 + kt * 16 + lg * 4);
                rwq[qt][kt] = v * LOG2E;
            }
        }
    }
    u32x4 qf[2][S::STEPS], dof[2][S::STEPS];
    float dsum[2], lq2[2];
    bool qok[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        S::gmem_frags(qf[qt], qg, p.q_rs, q0 + qt * 16, p.Nq, l15, lg);
        S::gmem_frags(dof[qt], dog, p.o_rs, q0 + qt * 16, p.Nq, l15, lg);
        const int q = q0 + qt * 16 + l15;
        qok[qt] = q < p.Nq;
        float ds = 0.f;                                    // D[q] = sum_d dO * O
        if (qok[qt]) {
#pragma unroll
            for (int s = 0; s < S::STEPS; ++s) {
                float a[Chunk<T>::N], c[Chunk<T>::N];
                Chunk<T>::unpack(dof[qt][s], a);
                Chunk<T>::unpack(ld_chunk(og + (size_t)q * p.o_rs + (s * 4 + lg) * S::EPC), c);
#pragma unroll
                for (int k = 0; k < Chunk<T>::N; ++k) ds += a[k] * c[k];
            }
        }
        ds += __shfl_xor(ds, 16, 64);
        ds += __shfl_xor(ds, 32, 64);
        dsum[qt] = ds;
        if (lg == 0 && qok[qt]) p.dsum[(size_t)bh * p.Nq + q] = ds;
        lq2[qt] = qok[qt] ? p.lse[(size_t)bh * p.Nq + q] * LOG2E : 0.f;
    }
    f32x4 o[2][S::DT];
    f32x4 ge[2][2];                                        // REL 1: d rel tables, 32 columns
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
        for (int dt = 0; dt < S::DT; ++dt) o[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        ge[qt][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        ge[qt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float c2 = p.scale * LOG2E;
    const f32x4 c2v = {c2, c2, c2, c2};
    constexpr bool drop = DROP;      // compiled out of the relative-position instantiations (register budget)
    const unsigned dthresh = sa_thresh(p.dropout_p);
    // (r06) a captured step freezes p.seed; the varying part lives in device memory and is bumped by the host before every replay
    const unsigned dseed = (DROP && p.seed_device != nullptr) ? p.seed + *p.seed_device : p.seed;
    const float inv_keep = drop ? 1.f / (1.f - p.dropout_p) : 1.f;
    // C operand of the dO.V^T tiles: -D (without dropout, which masks dP before D is subtracted)
    f32x4 dinit[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const float v = drop ? 0.f : -dsum[qt];
        dinit[qt] = f32x4{v, v, v, v};
    }
    const sa_rsrc_t k_rsrc = S::rsrc(kg, p.k_rs, p.Nk), v_rsrc = S::rsrc(vg, p.v_rs, p.Nk);
    S::dma(k_rsrc, KV, p.k_rs, 0, p.Nk, wave, lane);
    S::dma(v_rsrc, KV + S::CHUNK_BYTES, (D == 32 ? p.v_rs : p.k_rs), 0, p.Nk, wave, lane);
    // REL 2: rel_h[q][kh] of the two rows this lane owns, read one chunk ahead; rows past Nq read row Nq - 1 (any finite bias
    // does: their dS is zero, see above) so that the load needs no branch
    const float* rhp[2] = {nullptr, nullptr};
    float rhn[2] = {0.f, 0.f};
    if constexpr (REL == 2) {
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            rhp[qt] = p.rel_h + ((size_t)bh * p.Nq + min(q0 + qt * 16 + l15, p.Nq - 1)) * p.Sh;
            rhn[qt] = rhp[qt][0];
        }
    }
    const int nchunk = (p.Nk + SA_CHUNK - 1) / SA_CHUNK;
    sa_settle(qf);
    sa_settle(dof);
    sa_settle(dinit);
    sa_settle(lq2);
    if constexpr (EMM) sa_settle(rf);
    if constexpr (REL == 2) { sa_settle(rwq); sa_settle(rhn); }

    for (int ci = 0; ci < nchunk; ++ci) {
        const int k0 = ci * SA_CHUNK;
        const int buf = ci & 1;
        const char* Ks = KV + buf * 2 * S::CHUNK_BYTES;
        const char* Vs = Ks + S::CHUNK_BYTES;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this chunk has landed (own DMA) ...
        if constexpr (REL == 2) sa_settle(rhn);               // (the compiler's wait for last chunk's rel_h loads belongs HERE, not behind the DMA issue)
        __syncthreads();                                      // ... for every wavefront; the other buffer is free again
        if (ci + 1 < nchunk) {                                // next chunk streams in under this one's math
            char* nxt = KV + (buf ^ 1) * 2 * S::CHUNK_BYTES;
            S::dma(k_rsrc, nxt, p.k_rs, k0 + SA_CHUNK, p.Nk, wave, lane);
            S::dma(v_rsrc, nxt + S::CHUNK_BYTES, (D == 32 ? p.v_rs : p.k_rs), k0 + SA_CHUNK, p.Nk, wave, lane);
        }
        // exponent offset of the chunk: -lse (+ the chunk's rel_h value, REL 2), log2 domain
        f32x4 offv[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const float v = (REL == 2 ? rhn[qt] * LOG2E : 0.f) - lq2[qt];
            offv[qt] = f32x4{v, v, v, v};
        }
        if constexpr (REL == 2) {                             // fetched one chunk ahead (see forward)
            const int nx = min(ci + 1, nchunk - 1);
            rhn[0] = rhp[0][nx];
            rhn[1] = rhp[1][nx];
        }
        if (live) {
        const bool tail = REL != 2 && k0 + SA_CHUNK > p.Nk;   // wave-uniform (REL 2: Nk = Sh * 64, no partial chunk)
        // The chunk's math exists twice: the copy for the (one) partial chunk skips the 16-key tiles that hold no valid key at all --
        // for the short sequences (ViT N = 197, SAM windows N = 196: four chunks, the last with 4-5 keys) that is 3/16 of the S / dP
        // work and 1/8 of dS.K -- and pays for the key mask; the copy every other chunk runs has neither branch nor select.
        auto compute = [&](auto tail_c) {
        constexpr bool TAIL = decltype(tail_c)::value;
        const int nkt = TAIL ? (p.Nk - k0 + 15) >> 4 : 4;     // key tiles of this chunk with a valid key
        f32x4 g[2][4];
        f32x2 ghc[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};    // REL 2: partial row sums of the chunk's d logits
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            if (TAIL && kt >= nkt) {
                g[0][kt] = g[1][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
                continue;
            }
            u32x4 kf[S::STEPS], vf[S::STEPS];
            S::lds_frags(kf, Ks, kt * 16, l15, lg);
            S::lds_frags(vf, Vs, kt * 16, l15, lg);
            f32x4 sv2[2], dp[2];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                sv2[qt] = S::tile(kf, qf[qt]);
                dp[qt] = S::tile_c(vf, dof[qt], dinit[qt]);
            }
            if constexpr (EMM) {
#pragma unroll
                for (int e = 0; e < RL::ECH; ++e) {
                    const u32x4 ef = ld_chunk(Es + sa_off<EROWB>(k0 + kt * 16 + l15, e * 4 + lg));
                    Mma<T>::run(sv2[0], ef, rf[0][e]);
                    Mma<T>::run(sv2[1], ef, rf[1][e]);
                }
            }
            f32x4 kbq = {0.f, 0.f, 0.f, 0.f};                 // key bias of keys k0 + kt*16 + lg*4 + 0..3
            if constexpr (KB) kbq = *reinterpret_cast<const f32x4*>(kbs + k0 + kt * 16 + lg * 4);
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                f32x4 bq = offv[qt];
                if constexpr (REL == 2) bq += rwq[qt][kt];
                if constexpr (KB) bq += kbq;
                if constexpr (TAB) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        int kh, kw;
                        sa_split_key(min(k0 + kt * 16 + lg * 4 + r, p.Nk - 1), inv_sw, p.Sw, kh, kw);
                        bq[r] += rh[(qt * 16 + l15) * (p.Sh + 1) + kh] + rw[(qt * 16 + l15) * (p.Sw + 1) + kw];
                    }
                }
                const f32x4 x = sv2[qt] * c2v + bq;
                const f32x4 pr = {fast_exp2(x[0]), fast_exp2(x[1]), fast_exp2(x[2]), fast_exp2(x[3])};
                f32x4 gv;                                     // d logits
                if constexpr (drop) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool keep = sa_keep(dseed, bh, q0 + qt * 16 + l15, k0 + kt * 16 + lg * 4 + r, p.Nk, dthresh);
                        gv[r] = pr[r] * ((keep ? dp[qt][r] * inv_keep : 0.f) - dsum[qt]);
                    }
                } else {
                    gv = pr * dp[qt];
                }
                g[qt][kt] = gv;
                if constexpr (REL == 2) ghc[qt] += f32x2{gv[0], gv[1]} + f32x2{gv[2], gv[3]};
                if constexpr (REL == 3) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = k0 + kt * 16 + lg * 4 + r;
                        if (key < p.Nk && qok[qt]) {
                            int kh, kw;
                            sa_split_key(key, inv_sw, p.Sw, kh, kw);
                            atomicAdd(&gh[(qt * 16 + l15) * (p.Sh + 1) + kh], gv[r]);
                            atomicAdd(&gw[(qt * 16 + l15) * (p.Sw + 1) + kw], gv[r]);
                        }
                    }
                }
            }
        }
        if constexpr (TAIL) {
            const int kb0 = k0 + lg * 4;
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kb0 + kt * 16 + r >= p.Nk) g[qt][kt][r] = 0.f;
        }
        if constexpr (REL == 2) {
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) gwq[qt][kt] += g[qt][kt];
            // the whole chunk is one kh row: d rel_h[q][ci] = the row sum.  Stores inside the loop would be waited for by the
            // next chunk's vmcnt(0) (stores count there on gfx9): eight chunks are collected in registers -- every lane group
            // gets the sum, group c >> 1 keeps it -- and leave as one 8-byte store per lane (32 contiguous bytes per row)
            const int c8 = ci & 7;
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                const float v = sa_lg_sum(ghc[qt][0] + ghc[qt][1]);
                ghq[qt][0] = (c8 == lg * 2) ? v : ghq[qt][0];
                ghq[qt][1] = (c8 == lg * 2 + 1) ? v : ghq[qt][1];
            }
            if (c8 == 7 || ci == nchunk - 1) {
                asm volatile("" ::: "memory");
                const int base = ci - c8 + lg * 2;            // first kh column of this lane's two
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) {
                    const int q = q0 + qt * 16 + l15;
                    if (q < p.Nq) {
                        float* dst = p.d_rel_h + ((size_t)bh * p.Nq + q) * p.Sh + base;
                        if ((p.Sh & 1) == 0 && base + 1 < p.Sh) *reinterpret_cast<f32x2*>(dst) = ghq[qt];
                        else {
                            if (base < p.Sh) dst[0] = ghq[qt][0];
                            if (base + 1 < p.Sh) dst[1] = ghq[qt][1];
                        }
                    }
                }
            }
        }
        {
            const f32x4 a0[2] = {g[0][0], g[1][0]}, a1[2] = {g[0][1], g[1][1]};
            const f32x4 b0[2] = {g[0][2], g[1][2]}, b1[2] = {g[0][3], g[1][3]};
            pvN<T, S::ROWB, S::DT, 2>(o, a0, a1, Ks, 0, l15, lg);          // dQ += dS K   (scale applied at the end)
            if (!TAIL || nkt > 2) pvN<T, S::ROWB, S::DT, 2>(o, b0, b1, Ks, 32, l15, lg);
            if constexpr (REL == 1) {
                pvN<T, EROWB, 2, 2>(ge, a0, a1, Es, k0, l15, lg);          // d rel += dS E
                if (!TAIL || nkt > 2) pvN<T, EROWB, 2, 2>(ge, b0, b1, Es, k0 + 32, l15, lg);
            }
        }
        };
        if (tail) compute(std::true_type{});
        else compute(std::false_type{});
        }
    }
    T* dqg = (T*)p.dq + (size_t)b * p.q_bs + h * D;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qq = q0 + qt * 16 + lg * 4 + r;
            if (qq < p.Nq) {
#pragma unroll
                for (int dt = 0; dt < S::DT; ++dt) dqg[(size_t)qq * p.q_rs + dt * 16 + l15] = from_f32<T>(o[qt][dt][r] * p.scale);
                if constexpr (REL == 1) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int col = t * 16 + l15;
                        if (col < p.Sh) p.d_rel_h[((size_t)bh * p.Nq + qq) * p.Sh + col] = ge[qt][t][r];
                        else if (col < p.Sh + p.Sw) p.d_rel_w[((size_t)bh * p.Nq + qq) * p.Sw + col - p.Sh] = ge[qt][t][r];
                    }
                }
            }
        }
    if constexpr (REL == 2) {       // 16 bytes per lane, the four lane groups x four key tiles of a row are its 256 bytes
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int q = q0 + qt * 16 + l15;
            if (q < p.Nq) {
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
                    *reinterpret_cast<f32x4*>(p.d_rel_w + ((size_t)bh * p.Nq + q) * 64 + kt * 16 + lg * 4) = gwq[qt][kt];
            }
        }
    }
    if constexpr (REL == 3) {       // this wavefront's LDS atomics are complete once its own counters drain
        __builtin_amdgcn_s_waitcnt(0);
        for (int i = lane; i < SA_WROWS * p.Sh; i += 64) {
            const int r = i / p.Sh, c = i - r * p.Sh;
            if (q0 + r < p.Nq) p.d_rel_h[((size_t)bh * p.Nq + q0 + r) * p.Sh + c] = gh[r * (p.Sh + 1) + c];
        }
        for (int i = lane; i < SA_WROWS * p.Sw; i += 64) {
            const int r = i / p.Sw, c = i - r * p.Sw;
            if (q0 + r < p.Nq) p.d_rel_w[((size_t)bh * p.Nq + q0 + r) * p.Sw + c] = gw[r * (p.Sw + 1) + c];
        }
    }
}

// ------------------------------------------------------------------------------------ backward: dK, dV
// A wavefront owns 32 keys; queries stream in 64-row chunks (Q, dO, D, lse and, with a relative-position
// bias, the chunk's rows of rel_h / rel_w).  P / dS tiles: rows = queries lg*4 + r, column = key l15.
// REL 2 (Sw == 64): the block's 128 keys are two kh rows, so only two columns of rel_h are needed per
// chunk; rel_w rows go through LDS as float4 with an XOR swizzle on (row >> 2) that keeps the four lane
// groups (rows lg*4 + r) on disjoint banks.  Everything for chunk i+1 is fetched into registers during chunk i.
DEVINL int sa_rw_off(int row, int kw) { return row * 64 + ((((kw >> 2) ^ (((row >> 2) & 3) << 2))) << 2) + (kw & 3); }

// r04: as in the dQ kernel the chunk's elementwise work is packed: -lse (REL 2: + the key row's rel_h value) and -D come out
// of LDS as float4 row constants -- the first as the addend of the one fma that also applies the log2-domain scale, the
// second as the C operand of the dO.V^T MFMAs -- and nothing is masked: rows of the chunk past Nq are zero rows of Q / dO with
// D = lse = 0 (finite P, zero dS, zero dO), keys past Nk only ever touch their own (unstored) output rows.
template <typename T, int D, int REL, bool DROP, bool KB>
// (two wavefronts per SIMD: held to 168 registers for three, the plain form ran 841 -> 974 us -- more wavefronts parked)
__global__ __launch_bounds__(SA_THREADS, 2) void sa_bwd_dkv_kernel(const SAParams p) {
    using S = SA<T, D>;
    constexpr bool TAB = REL == 3;         // per-chunk LDS tables + VALU adds
    constexpr bool EMM = REL == 1;         // bias on the MFMA (SARel): R rows of the query chunk in LDS, E of the own keys in registers
    using RL = SARel<T>;
    constexpr int NPF = RL::CPR * SA_CHUNK / SA_THREADS;      // 16-byte chunks of R per thread per query chunk
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // XCD-aware mapping (workgroups are dealt to the 8 XCDs round-robin in launch order, x fastest): every row block
    // of one (batch, head) reads the same K / V (Q / dO), so each XCD gets a contiguous (head, block) range
    const int sa_lid = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int bh = sa_lid / (int)gridDim.x, blk = sa_lid - bh * (int)gridDim.x;
    const int b = bh / p.H, h = bh - b * p.H;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, lg = lane >> 4;
    char* QO = smem;                                       // [2 buffers][Q chunk | dO chunk], filled by DMA
    float* Dq = reinterpret_cast<float*>(smem + 4 * S::CHUNK_BYTES);
    float* Ls = Dq + SA_CHUNK;
    float* rhs = Ls + SA_CHUNK;                            // TAB: [64][Sh + 1]      REL 2: [2][64]
    char* Rs = reinterpret_cast<char*>(Ls + SA_CHUNK);     // EMM: R[64 queries][32] operand image
    float* rws = TAB ? rhs + SA_CHUNK * (p.Sh + 1) : rhs + 2 * SA_CHUNK;   // TAB: [64][Sw + 1]   REL 2: [64][64] swizzled
    const T* qg = (const T*)p.q + (size_t)b * p.q_bs + h * D;
    const T* kg = (const T*)p.k + (size_t)b * p.k_bs + h * D;
    const T* vg = (const T*)p.v + (size_t)b * p.v_bs + h * D;
    const T* dog = (const T*)p.dout + (size_t)b * p.o_bs + h * D;
    const float* dsg = p.dsum + (size_t)bh * p.Nq;
    const float* lsg = p.lse + (size_t)bh * p.Nq;
    const int key0 = blk * SA_BROWS + wave * SA_WROWS;
    const bool live = key0 < p.Nk;                            // wave-uniform: a wavefront without a valid key only feeds the stream
    u32x4 kf[2][S::STEPS], vf[2][S::STEPS];
    float kbias[2];
    int khl[2], kwl[2];
    bool kok[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int key = key0 + t * 16 + l15;
        kok[t] = key < p.Nk;
        kbias[t] = (KB && kok[t]) ? p.key_bias[(size_t)b * p.Nk + key] * LOG2E : 0.f;
        khl[t] = kwl[t] = 0;
        if constexpr (TAB) {
            if (kok[t]) sa_split_key(key, 1.f / (float)p.Sw, p.Sw, khl[t], kwl[t]);
        }
        if constexpr (REL == 2) kwl[t] = key & 63;
        S::gmem_frags(kf[t], kg, p.k_rs, key0 + t * 16, p.Nk, l15, lg);
        S::gmem_frags(vf[t], vg, p.v_rs, key0 + t * 16, p.Nk, l15, lg);
    }
    u32x4 ek[2][RL::ECH];
    if constexpr (EMM) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < RL::ECH; ++e) ek[t][e] = RL::key_frag(p, key0 + t * 16 + l15, e * 4 + lg);
    }
    float pf_rel[NPF][Chunk<T>::N];                          // raw rel_h / rel_w values of the next chunk's R rows
    f32x4 dv[2][S::DT], dk[2][S::DT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int dt = 0; dt < S::DT; ++dt) { dv[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dk[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const float c2 = p.scale * LOG2E;
    const f32x4 c2v = {c2, c2, c2, c2};
    constexpr bool drop = DROP;      // compiled out of the relative-position instantiations (register budget)
    const unsigned dthresh = sa_thresh(p.dropout_p);
    // (r06) a captured step freezes p.seed; the varying part lives in device memory and is bumped by the host before every replay
    const unsigned dseed = (DROP && p.seed_device != nullptr) ? p.seed + *p.seed_device : p.seed;
    const float inv_keep = drop ? 1.f / (1.f - p.dropout_p) : 1.f;
    const sa_rsrc_t q_rsrc = S::rsrc(qg, p.q_rs, p.Nq), o_rsrc = S::rsrc(dog, p.o_rs, p.Nq);
    float pf_d = 0.f, pf_l = 0.f, pf_rh = 0.f;             // D (tid < 64), lse * log2(e) and the rel_h column tid >> 6 (tid < 128)
    // REL 2: the [64 queries][64 kw] fp32 tile of rel_w goes global -> LDS by DMA (no registers, double buffered);
    // the DMA writes wave-linear 16-byte slots, so the XOR swizzle is applied to the SOURCE address
    const float* rwg = REL == 2 ? p.rel_w + (size_t)bh * p.Nq * 64 : nullptr;
    const sa_rsrc_t rw_rs = sa_make_rsrc(REL == 2 ? (const void*)rwg : (const void*)p.lse,
                                         REL == 2 ? (unsigned)((size_t)p.Nq * 64 * sizeof(float)) : 0u);      // rows past Nq: out of bounds -> zeros
    auto dma_rw = [&](int q0, int buf) {
        const unsigned dst = sa_lds_addr(rws + buf * SA_CHUNK * 64) + wave * 1024;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int slot = (j * SA_WAVES + wave) * 64 + lane, row = slot >> 4, pos = slot & 15;
            const int c4 = pos ^ (((row >> 2) & 3) << 2);
            sa_dma16(rw_rs, dst + j * SA_WAVES * 1024, (unsigned)(((q0 + row) * 64 + c4 * 4) * (int)sizeof(float)));
        }
    };
    const float* rhg = REL == 2 ? p.rel_h + (size_t)bh * p.Nq * p.Sh + 2 * blk : nullptr;
    auto prefetch = [&](int q0) {
        char* nxt = QO + ((q0 / SA_CHUNK) & 1) * 2 * S::CHUNK_BYTES;
        S::dma(q_rsrc, nxt, p.q_rs, q0, p.Nq, wave, lane);
        S::dma(o_rsrc, nxt + S::CHUNK_BYTES, p.o_rs, q0, p.Nq, wave, lane);
        {                                                  // (every wavefront: no branch, no merge of old and new values)
            const int r = q0 + (tid & 63);
            // RAW loads only (rows past Nq re-read row Nq - 1 and are zeroed when the chunk becomes current): any arithmetic on
            // the values here makes the compiler wait vmcnt(0) on the spot -- behind the DMA pieces issued just above, i.e. a
            // full L2 round trip exposed in every chunk (r04 trace: 54 % of the wavefront cycles parked)
            const int rc = min(r, p.Nq - 1);
            pf_l = lsg[rc];
            pf_d = dsg[rc];
            if constexpr (REL == 2) pf_rh = rhg[(size_t)rc * p.Sh + ((tid >> 6) & 1)];
        }
        if constexpr (EMM) {                               // converted and stored when the chunk becomes current
#pragma unroll
            for (int j = 0; j < NPF; ++j) {
                const int i = tid + j * SA_THREADS, row = i / RL::CPR, ch = i - row * RL::CPR, q = q0 + row;
#pragma unroll
                for (int n = 0; n < Chunk<T>::N; ++n) {
                    const int col = ch * Chunk<T>::N + n;
                    float v = 0.f;
                    if (q < p.Nq) {
                        if (col < p.Sh) v = p.rel_h[((size_t)bh * p.Nq + q) * p.Sh + col];
                        else if (col < p.Sh + p.Sw) v = p.rel_w[((size_t)bh * p.Nq + q) * p.Sw + col - p.Sh];
                    }
                    pf_rel[j][n] = v;
                }
            }
        }
    };
    sa_settle(kf);
    sa_settle(vf);
    sa_settle(kbias);
    if constexpr (EMM) sa_settle(ek);
    prefetch(0);
    if constexpr (REL == 2) dma_rw(0, 0);

    for (int q0 = 0; q0 < p.Nq; q0 += SA_CHUNK) {
        const char* Qs = QO + ((q0 / SA_CHUNK) & 1) * 2 * S::CHUNK_BYTES;
        const char* Os = Qs + S::CHUNK_BYTES;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this chunk's Q / dO (and rel_w) tiles have landed
        sa_settle(pf_l);
        sa_settle(pf_d);
        if constexpr (REL == 2) sa_settle(pf_rh);
        __syncthreads();
        if (tid < 2 * SA_CHUNK) {                          // the row constants, negated / combined once per row here
            const bool rok = q0 + (tid & 63) < p.Nq;
            const float l2 = rok ? pf_l * LOG2E : 0.f;
            if (tid < SA_CHUNK) { Dq[tid] = rok ? -pf_d : 0.f; Ls[tid] = -l2; }
            if constexpr (REL == 2) rhs[tid] = (rok ? pf_rh * LOG2E : 0.f) - l2;   // [2 kh rows of the block][64 queries]
        }
        if constexpr (EMM) {
            const float mul = 1.f / p.scale;
#pragma unroll
            for (int j = 0; j < NPF; ++j) {
                const int i = tid + j * SA_THREADS, row = i / RL::CPR, ch = i - row * RL::CPR;
                float f[Chunk<T>::N];
#pragma unroll
                for (int n = 0; n < Chunk<T>::N; ++n) f[n] = pf_rel[j][n] * mul;
                st_chunk(Rs + sa_off<RL::EROWB>(row, ch), Chunk<T>::pack(f));
            }
        }
        if constexpr (TAB) {
            for (int i = tid; i < SA_CHUNK * p.Sh; i += SA_THREADS) {
                const int r = i / p.Sh, c = i - r * p.Sh;
                rhs[r * (p.Sh + 1) + c] = (q0 + r) < p.Nq ? p.rel_h[((size_t)bh * p.Nq + q0 + r) * p.Sh + c] * LOG2E : 0.f;
            }
            for (int i = tid; i < SA_CHUNK * p.Sw; i += SA_THREADS) {
                const int r = i / p.Sw, c = i - r * p.Sw;
                rws[r * (p.Sw + 1) + c] = (q0 + r) < p.Nq ? p.rel_w[((size_t)bh * p.Nq + q0 + r) * p.Sw + c] * LOG2E : 0.f;
            }
        }
        __syncthreads();
        if (q0 + SA_CHUNK < p.Nq) {
            prefetch(q0 + SA_CHUNK);
            if constexpr (REL == 2) dma_rw(q0 + SA_CHUNK, ((q0 >> 6) + 1) & 1);
        }
        const float* rwc = rws + ((q0 >> 6) & 1) * SA_CHUNK * 64;      // REL 2: this chunk's tile
        if (live) {
        // the partial query chunk (N = 196 / 197: the last chunk has 4-5 rows) skips the 32-row pair and the 16-row tile that hold no
        // valid query: two wave-uniform compares per chunk (a second copy of the chunk's math, as in the dQ kernel, costs this kernel
        // its registers: 256 + spills)
        const int nvq = p.Nq - q0;                         // valid rows of this chunk (>= 64 except in the last one)
#pragma unroll
        for (int pair = 0; pair < 2; ++pair) {             // 32 queries at a time
            if (pair * 32 >= nvq) continue;
            f32x4 pt[2][2], dst[2][2];                     // [key tile][query tile of the pair]
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const int qt = pair * 2 + qq;
                if (qt * 16 >= nvq) {
                    pt[0][qq] = pt[1][qq] = dst[0][qq] = dst[1][qq] = f32x4{0.f, 0.f, 0.f, 0.f};
                    continue;
                }
                u32x4 qf[S::STEPS], dof[S::STEPS];
                S::lds_frags(qf, Qs, qt * 16, l15, lg);
                S::lds_frags(dof, Os, qt * 16, l15, lg);
                const int ql0 = qt * 16 + lg * 4;          // this lane's four query rows of the tile
                const f32x4 nd = *reinterpret_cast<const f32x4*>(Dq + ql0);
                f32x4 rowc;                                // -lse (+ rel_h), log2 domain
                if constexpr (REL == 2) rowc = *reinterpret_cast<const f32x4*>(rhs + (wave >> 1) * SA_CHUNK + ql0);
                else rowc = *reinterpret_cast<const f32x4*>(Ls + ql0);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f32x4 sv2 = S::tile(qf, kf[t]);        // rows queries, col key
                    f32x4 dp;
                    if constexpr (drop) dp = S::tile(dof, vf[t]);
                    else dp = S::tile_c(dof, vf[t], nd);   // dP - D
                    if constexpr (EMM) {
#pragma unroll
                        for (int e = 0; e < RL::ECH; ++e)
                            Mma<T>::run(sv2, ld_chunk(Rs + sa_off<RL::EROWB>(qt * 16 + l15, e * 4 + lg)), ek[t][e]);
                    }
                    f32x4 bq = rowc;
                    if constexpr (KB) bq += f32x4{kbias[t], kbias[t], kbias[t], kbias[t]};
                    if constexpr (TAB) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) bq[r] += rhs[(ql0 + r) * (p.Sh + 1) + khl[t]] + rws[(ql0 + r) * (p.Sw + 1) + kwl[t]];
                    }
                    if constexpr (REL == 2) {
                        const f32x4 rwv = {rwc[sa_rw_off(ql0, kwl[t])], rwc[sa_rw_off(ql0 + 1, kwl[t])],
                                           rwc[sa_rw_off(ql0 + 2, kwl[t])], rwc[sa_rw_off(ql0 + 3, kwl[t])]};
                        bq = rwv * f32x4{LOG2E, LOG2E, LOG2E, LOG2E} + bq;
                    }
                    const f32x4 x = sv2 * c2v + bq;
                    const f32x4 pr = {fast_exp2(x[0]), fast_exp2(x[1]), fast_exp2(x[2]), fast_exp2(x[3])};
                    if constexpr (drop) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float keepf = sa_keep(dseed, bh, q0 + ql0 + r, key0 + t * 16 + l15, p.Nk, dthresh) ? inv_keep : 0.f;
                            pt[t][qq][r] = pr[r] * keepf;
                            dst[t][qq][r] = pr[r] * (dp[r] * keepf + nd[r]);
                        }
                    } else {
                        pt[t][qq] = pr;
                        dst[t][qq] = pr * dp;
                    }
                }
            }
            const f32x4 p0[2] = {pt[0][0], pt[1][0]}, p1[2] = {pt[0][1], pt[1][1]};
            const f32x4 d0[2] = {dst[0][0], dst[1][0]}, d1[2] = {dst[0][1], dst[1][1]};
            pvN<T, S::ROWB, S::DT, 2>(dv, p0, p1, Os, pair * 32, l15, lg);      // dV += P^T dO
            pvN<T, S::ROWB, S::DT, 2>(dk, d0, d1, Qs, pair * 32, l15, lg);      // dK += dS^T Q  (scale at the end)
        }
        }
    }
    T* dkg = (T*)p.dk + (size_t)b * p.k_bs + h * D;
    T* dvg = (T*)p.dv + (size_t)b * p.v_bs + h * D;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kk = key0 + t * 16 + lg * 4 + r;
            if (kk < p.Nk) {
#pragma unroll
                for (int dt = 0; dt < S::DT; ++dt) {
                    dkg[(size_t)kk * p.k_rs + dt * 16 + l15] = from_f32<T>(dk[t][dt][r] * p.scale);
                    dvg[(size_t)kk * p.v_rs + dt * 16 + l15] = from_f32<T>(dv[t][dt][r]);
                }
            }
        }
}

template <typename K>
void sa_allow_lds(K k) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

template <typename T, int D, int REL, bool DROP = false, bool KB = false>
int sa_launch(const SAParams& p, int which, hipStream_t st) {
    const size_t chunk = (size_t)SA_CHUNK * D * sizeof(T);
    const size_t tab = (REL == 1 || REL == 3) ? (size_t)SA_WAVES * SA_WROWS * (p.Sh + p.Sw + 2) * sizeof(float) : 0;
    const size_t kbl = KB ? (size_t)((p.Nk + SA_CHUNK - 1) / SA_CHUNK) * SA_CHUNK * sizeof(float) : 0;      // key-bias row, whole chunks
    if (which == 0) {
        if constexpr (!DROP && REL <= 2 && sizeof(T) == 2) {
            // r04 forward (three-slot ring, in-register reductions, deferred rescale); SAICV_SA_FWD2=0 selects the r03 kernel
            // measured (r04e, one box, us r03 -> r04): SAM global N = 4096 with rel-pos 985 -> 838; plain d64 N = 4096 709 -> 689;
            // ViT N = 197 level; DETR d32 N = 1764 100 -> 107 -- so by default the decomposed-bias launches take it, SAICV_SA_FWD2=2
            // sends every eligible launch there, =0 none
            static const int fwd2_env = getenv("SAICV_SA_FWD2") ? atoi(getenv("SAICV_SA_FWD2")) : 1;
            const bool fwd2 = fwd2_env == 2 || (fwd2_env == 1 && REL != 1);
            const size_t aux = REL == 2 ? (size_t)SA_RING * 4 * 256               // the ring's rel_h column pieces
                             : REL == 1 ? (size_t)SARel<T>::EBYTES
                                        : (p.key_bias ? (size_t)((p.Nk + SA_CHUNK - 1) / SA_CHUNK) * SA_CHUNK * sizeof(float) : 0);   // padded to whole chunks
            if (fwd2 && (REL == 0 || !KB)) {
                const dim3 grid((p.Nq + SA_BROWS - 1) / SA_BROWS, p.B * p.H);
                if constexpr (REL == 0 && KB) {
                    auto k2 = sa_fwd2_kernel<T, D, 0, true>;
                    static bool once2 = (sa_allow_lds(k2), true);
                    (void)once2;
                    hipLaunchKernelGGL(k2, grid, dim3(SA_THREADS), SA_RING * 2 * chunk + aux, st, p);
                } else {
                    auto k2 = sa_fwd2_kernel<T, D, REL, false>;
                    static bool once2 = (sa_allow_lds(k2), true);
                    (void)once2;
                    hipLaunchKernelGGL(k2, grid, dim3(SA_THREADS), SA_RING * 2 * chunk + aux, st, p);
                }
                return saicv::check_launch("attention_stream");
            }
        }
        auto k = sa_fwd_kernel<T, D, REL, DROP, KB>;
        static bool once = (sa_allow_lds(k), true);
        (void)once;
        hipLaunchKernelGGL(k, dim3((p.Nq + SA_BROWS - 1) / SA_BROWS, p.B * p.H), dim3(SA_THREADS),
                           4 * chunk + (REL == 1 ? (size_t)SARel<T>::EBYTES : tab) + kbl, st, p);
    } else if (which == 1) {
        auto k = sa_bwd_dq_kernel<T, D, REL, DROP, KB>;
        static bool once = (sa_allow_lds(k), true);
        (void)once;
        const size_t e = REL == 1 ? (size_t)256 * 32 * sizeof(T) : 0;
        const size_t g2 = 0;          // (REL 2 kept d rel_w in LDS until r04)
        hipLaunchKernelGGL(k, dim3((p.Nq + SA_BROWS - 1) / SA_BROWS, p.B * p.H), dim3(SA_THREADS),
                           4 * chunk + e + g2 + kbl + (REL == 3 ? 2 * tab : 0), st, p);
    } else {
        auto k = sa_bwd_dkv_kernel<T, D, REL, DROP, KB>;
        static bool once = (sa_allow_lds(k), true);
        (void)once;
        const size_t rel = REL == 2 ? (size_t)(2 * SA_CHUNK + 2 * SA_CHUNK * 64) * sizeof(float)
                         : REL == 1 ? (size_t)SA_CHUNK * SARel<T>::EROWB
                         : REL ? (size_t)SA_CHUNK * (p.Sh + p.Sw + 2) * sizeof(float) : 0;
        hipLaunchKernelGGL(k, dim3((p.Nk + SA_BROWS - 1) / SA_BROWS, p.B * p.H), dim3(SA_THREADS),
                           4 * chunk + 2 * SA_CHUNK * sizeof(float) + rel, st, p);
    }
    return saicv::check_launch("attention_stream");
}

template <typename T>
int sa_dispatch(int D, const SAParams& p, int which, hipStream_t st) {
    // the additive key bias is a template switch of the backward kernels (a pointer test per logit otherwise): instantiated
    // for the plain kernels (DETR) and for the Sw == 64 relative-position form
    const bool kb = p.key_bias != nullptr;
    if (p.dropout_p > 0.f) {
        if (D == 32) return kb ? sa_launch<T, 32, 0, true, true>(p, which, st) : sa_launch<T, 32, 0, true, false>(p, which, st);
        return kb ? sa_launch<T, 64, 0, true, true>(p, which, st) : sa_launch<T, 64, 0, true, false>(p, which, st);
    }
    if (D == 32) return kb ? sa_launch<T, 32, 0, false, true>(p, which, st) : sa_launch<T, 32, 0>(p, which, st);
    if (!p.rel_h) return kb ? sa_launch<T, 64, 0, false, true>(p, which, st) : sa_launch<T, 64, 0>(p, which, st);
    if (p.Sw == 64) return kb ? sa_launch<T, 64, 2, false, true>(p, which, st) : sa_launch<T, 64, 2>(p, which, st);
    SAICV_REQUIRE(!kb, "attention_stream: a key bias together with relative-position tables is instantiated for Sw == 64 only");
    if (p.Sh + p.Sw <= 32 && p.Nk <= 256) return sa_launch<T, 64, 1>(p, which, st);
    return sa_launch<T, 64, 3>(p, which, st);
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
    {
        const size_t es = dtype == SAICV_DTYPE_BF16 ? 2 : 4, lim = (size_t)1 << 31;      // 32-bit buffer offsets inside one (batch, head) view
        SAICV_REQUIRE(((size_t)p.Nq * p.q_rs + D) * es < lim && ((size_t)p.Nk * p.k_rs + D) * es < lim &&
                          ((size_t)p.Nk * p.v_rs + D) * es < lim && ((size_t)p.Nq * p.o_rs + D) * es < lim,
                      "attention_stream: one batch element of an operand spans 2 GiB or more");
    }
    SAICV_REQUIRE(p.key_bias == nullptr || p.Nk <= SA_KB_LDS, "attention_stream: a key bias row of %d entries does not fit its LDS stage (%d)", p.Nk, SA_KB_LDS);
    // head dim 64 (packed qkv projections: ViT, SAM): the K and V pieces of a chunk share their per-lane offsets (two registers
    // the 168-register forms do not have); head dim 32 (DETR: k from the [q | k] projection, v from its own) keeps both strides
    SAICV_REQUIRE(D == 32 || p.k_rs == p.v_rs, "attention_stream: head dim 64 needs k and v with one row stride (%ld vs %ld)", (long)p.k_rs, (long)p.v_rs);
    SAICV_REQUIRE(p.dropout_p >= 0.f && p.dropout_p < 1.f, "attention_stream: dropout_p=%f outside [0, 1)", (double)p.dropout_p);
    SAICV_REQUIRE((p.rel_h == nullptr) == (p.rel_w == nullptr), "attention_stream: rel_h and rel_w come together");
    if (p.rel_h) {
        SAICV_REQUIRE(p.dropout_p == 0.f, "attention_stream: dropout is not instantiated together with a relative-position bias");
        SAICV_REQUIRE(D == 64, "attention_stream: the relative-position bias is instantiated for head dim 64");
        SAICV_REQUIRE(p.Sh >= 1 && p.Sw >= 1 && p.Sh * p.Sw == p.Nk, "attention_stream: Sh*Sw must equal Nk");
        SAICV_REQUIRE(p.Sh + p.Sw <= 128, "attention_stream: relative-position tables too wide for LDS");
    }
    if (dtype == SAICV_DTYPE_BF16) return sa_dispatch<bf16_t>(D, p, which, st);
    return sa_dispatch<float>(D, p, which, st);
}

}  // namespace saicv
