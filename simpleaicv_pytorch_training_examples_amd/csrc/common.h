// Device-side helpers shared by every gfx950 kernel in this library.
// CDNA4 only: 64-wide wavefronts, MFMA 16x16x32 bf16 / 16x16x4 f32, 160 KiB LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define SAICV_DTYPE_BF16 0
#define SAICV_DTYPE_F32 1

#define DEVINL __device__ __forceinline__

// ---------------------------------------------------------------- error state
namespace saicv {
void set_error(const char* fmt, ...);
int check_launch(const char* what);
}  // namespace saicv

#define SAICV_REQUIRE(cond, ...)            \
    do {                                    \
        if (!(cond)) {                      \
            saicv::set_error(__VA_ARGS__);  \
            return -1;                      \
        }                                   \
    } while (0)

// ---------------------------------------------------------------- 16-byte chunks
// Every tiled kernel moves data in 16-byte chunks: 8 bf16 or 4 f32.
template <typename T> struct ElemTraits;
template <> struct ElemTraits<bf16_t> { static constexpr int EPC = 8; };
template <> struct ElemTraits<float>  { static constexpr int EPC = 4; };

DEVINL u32x4 zero_chunk() { u32x4 z = {0u, 0u, 0u, 0u}; return z; }

DEVINL u32x4 ld_chunk(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
DEVINL void st_chunk(void* p, u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }
// streaming store: the line is not kept in L2 for a reader that will not come before it is evicted anyway (the vendor
// library's GEMMs store their output this way, "NTD" in its kernel names)
DEVINL void st_chunk_nt(void* p, u32x4 v) { __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p)); }
// streaming load: an operand read exactly once does not displace the lines other workgroups are about to re-read
DEVINL u32x4 ld_chunk_nt(const void* p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)); }

DEVINL float bf16_bits_to_f32(uint32_t hi16) { return __uint_as_float(hi16 << 16); }

DEVINL float to_f32(bf16_t v) { return (float)v; }
DEVINL float to_f32(float v) { return v; }

template <typename T> DEVINL T from_f32(float v);
template <> DEVINL bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }
template <> DEVINL float from_f32<float>(float v) { return v; }

// round a float through T (used so BN statistics see what is stored in HBM)
template <typename T> DEVINL float round_through(float v) { return to_f32(from_f32<T>(v)); }

// unpack a chunk into EPC floats / pack back
template <typename T> struct Chunk;
template <> struct Chunk<bf16_t> {
    static constexpr int N = 8;
    static DEVINL void unpack(u32x4 c, float* f) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i]     = __uint_as_float(c[i] << 16);
            f[2 * i + 1] = __uint_as_float(c[i] & 0xffff0000u);
        }
    }
    static DEVINL u32x4 pack(const float* f) {
        u32x4 c;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bf16x2 p;
            p[0] = (bf16_t)f[2 * i];
            p[1] = (bf16_t)f[2 * i + 1];
            c[i] = __builtin_bit_cast(uint32_t, p);
        }
        return c;
    }
};
template <> struct Chunk<float> {
    static constexpr int N = 4;
    static DEVINL void unpack(u32x4 c, float* f) {
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(c[i]);
    }
    static DEVINL u32x4 pack(const float* f) {
        u32x4 c;
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = __float_as_uint(f[i]);
        return c;
    }
};

// ---------------------------------------------------------------- MFMA wrappers
// D(16x16) += A(16xk) * B(kx16).  Lane l supplies row (l&15) of A and column (l&15) of B,
// k-group (l>>4).  C/D: col = l&15, row = (l>>4)*4 + reg   (guide §3).
// One call consumes one 16-byte chunk per operand: 32 k for bf16, 16 k for f32
// (the f32 form issues four 16x16x4 MFMAs, one per float of the chunk; the k order
//  is a permutation shared by A and B, which a dot product does not care about).
template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    static DEVINL void run(f32x4& acc, const u32x4& a, const u32x4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                     __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    static DEVINL void run(f32x4& acc, const u32x4& a, const u32x4& b) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[j]), __uint_as_float(b[j]),
                                                       acc, 0, 0, 0);
    }
};

// ---------------------------------------------------------------- exact-erf GELU pieces
// Phi(x) = 0.5 (1 + erf(x / sqrt 2)) and phi(x) = exp(-x^2 / 2) / sqrt(2 pi) from ONE v_exp_f32 and ONE v_rcp_f32:
// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below bf16 and below the 1e-3 fp32 parity bound);
// ~14 VALU operations against ~60 for erff + expf, which matters inside GEMM epilogues.
DEVINL void gelu_cdf_pdf(float x, float& cdf, float& pdf) {
    const float e = __builtin_amdgcn_exp2f(-0.72134752044448170f * x * x);      // exp(-x^2 / 2)
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752f, fabsf(x), 1.f));
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    const float erf_abs = fmaf(-poly, e, 1.f);
    cdf = 0.5f + copysignf(0.5f * erf_abs, x);
    pdf = 0.39894228040143268f * e;
}
// the same on eight values at once, written on float pairs so that the multiplies / fmas issue as packed fp32 instructions (one per two
// elements: inside a GEMM epilogue every vector instruction competes with the other workgroup's MFMAs for the issue port).  Same
// operations in the same order as gelu_cdf_pdf -- results are bit-identical.  -> gl = x * Phi(x), gr = Phi(x) + x * phi(x)
DEVINL void gelu_and_grad8(const float (&x)[8], float (&gl)[8], float (&gr)[8]) {
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const f32x2 v = {x[i], x[i + 1]};
        const f32x2 ea = (v * f32x2{-0.72134752044448170f, -0.72134752044448170f}) * v;
        const f32x2 e = {__builtin_amdgcn_exp2f(ea[0]), __builtin_amdgcn_exp2f(ea[1])};
        const f32x2 av = {fabsf(v[0]), fabsf(v[1])};
        const f32x2 ta = av * f32x2{0.3275911f * 0.70710678118654752f, 0.3275911f * 0.70710678118654752f} + f32x2{1.f, 1.f};
        const f32x2 t = {__builtin_amdgcn_rcpf(ta[0]), __builtin_amdgcn_rcpf(ta[1])};
        f32x2 q = t * f32x2{1.061405429f, 1.061405429f} + f32x2{-1.453152027f, -1.453152027f};
        q = t * q + f32x2{1.421413741f, 1.421413741f};
        q = t * q + f32x2{-0.284496736f, -0.284496736f};
        q = t * q + f32x2{0.254829592f, 0.254829592f};
        const f32x2 poly = t * q;
        const f32x2 erf_abs = f32x2{1.f, 1.f} - poly * e;                      // (contracted: fma(-poly, e, 1))
        const f32x2 sh = {copysignf(0.5f, v[0]), copysignf(0.5f, v[1])};
        const f32x2 cdf = sh * erf_abs + f32x2{0.5f, 0.5f};                    // 0.5 + copysign(0.5 * erf_abs, x): the scale by 0.5 is exact
        const f32x2 pdf = e * f32x2{0.39894228040143268f, 0.39894228040143268f};
        const f32x2 l = v * cdf, r = v * pdf + cdf;
        gl[i] = l[0]; gl[i + 1] = l[1];
        gr[i] = r[0]; gr[i + 1] = r[1];
    }
}
DEVINL void gelu_and_grad8(const float (&x)[4], float (&gl)[4], float (&gr)[4]) {      // fp32 chunks hold four values
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float c, d;
        gelu_cdf_pdf(x[i], c, d);
        gl[i] = x[i] * c;
        gr[i] = fmaf(x[i], d, c);
    }
}
DEVINL float gelu_fwd_f(float x) {
    float c, d;
    gelu_cdf_pdf(x, c, d);
    return x * c;
}
DEVINL float gelu_grad_f(float x) {
    float c, d;
    gelu_cdf_pdf(x, c, d);
    return fmaf(x, d, c);
}

// ---------------------------------------------------------------- reductions
// DPP lane permutations inside a 16-lane row: one VALU instruction each, no LDS crossbar (ds_bpermute)
template <int CTRL>
DEVINL float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// sum / max over the 16 lanes of a row, result in every lane: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror
DEVINL float row16_sum(float v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    return v;
}
DEVINL float row16_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0x140>(v));
    return v;
}
DEVINL float wave_sum(float v) {
    v = row16_sum(v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
DEVINL float wave_max(float v) {
    v = row16_max(v);
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}

// XCD-aware bijective block remap (guide T1): block b runs on XCD b%8; give each XCD a
// contiguous range of logical tiles so neighbouring tiles share that XCD's L2.
DEVINL int xcd_remap(int b, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = b & 7, idx = b >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// Ranks of the largest RCCL communicator this process has created through saicv_comm_create (comm.hip); 1 = this GPU runs the
// step's kernels only.  The weight-gradient kernel sizes its one resident round for all CU slots then, and for 85 % of them when
// all-reduce kernels share the GPU (igemm.hip, SAICV_TN_SLOTS_PCT overrides).
extern int g_saicv_comm_world;

