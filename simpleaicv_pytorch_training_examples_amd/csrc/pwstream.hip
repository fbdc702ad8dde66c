// Streaming pointwise convolution for the small-K x small-N layers of a residual stage (r06).
//
// ResNet-50's first stage at batch 256 runs 1 x 1 convolutions over M = 802 816 pixels with K, N in {64, 256}
// (reference SimpleAICV/classification/backbones/resnet.py:100-155, nn.Conv2d(inplanes, planes, 1) inside ConvBnActBlock :33-43)
// and their data gradients.  Per 16 pixels that is 2-8 KiB in, 2-8 KiB out and 0.13-0.5 MFLOP: HBM-bound by a factor of three to
// ten.  The tiled kernel (igemm.hip, igemm_nt1_kernel) spends 6.5-8 us of fixed cost per 128- or 256-row tile around a K loop of
// 1 us (profiles/r05_nt_timeline.md) and reaches 0.25-0.55 of the HBM bound on these shapes.
//
// Here there is no tile seam at all:
//   * the grid is ONE resident round of workgroups; every wavefront is an independent stream over 16-pixel row groups
//     (group g, g + G, g + 2G, ...), it never meets a workgroup barrier before the last statistics row;
//   * the WEIGHTS LIVE IN REGISTERS for the whole launch: a wavefront owns 64 output channels, i.e. 4 MFMA row tiles x K / 32
//     k-steps of 16-byte fragments = 32 (K = 64) to 128 (K = 256) VGPRs, loaded once; wider layers split the channels over the
//     NSPLIT wavefronts of a group, which then read the same (small) input rows -- L1 / L2 hits.  (r06, measured and removed:
//     the K = 256 weights in LDS with four row groups in flight instead of two -- forward 137 -> 129 us, data gradient 143 ->
//     157 us, profiles/r06_nt_experiments.md; read-heavy K = 256 products without fused operands stay on the tiled kernel,
//     whose LDS-DMA loads reach 4.9 TB/s on them against 4.0 here);
//   * the activations need no LDS either: with the weights as the FIRST MFMA operand, lane (pixel = lane & 15, k-group = lane >> 4)
//     of the second operand is 16 contiguous bytes of that pixel's row -- one global load per k-step, DEPTH row groups in flight
//     in registers ahead of the one being multiplied;
//   * the accumulators (channel rows x pixel columns) are staged through a WAVE-PRIVATE 2 KiB strip of LDS and read back as
//     16-byte row chunks, so that HBM sees whole 128-byte lines per pixel and every fused operand (shortcut gradient, its ReLU
//     gate, the pre-BatchNorm activation and its mask) is addressed exactly like the output; LDS operations of one wavefront
//     execute in order, so the strip needs no barrier;
//   * BatchNorm statistics (forward) and BatchNorm-backward sums (data gradient) accumulate in 16 registers per lane over the
//     wavefront's WHOLE stream and are combined once, at the end: one row of partials per workgroup (or atomics into the few
//     pooled rows of SAICV_BN_INLINE).
// Same arithmetic as the tiled kernel: bf16 operands, fp32 accumulation, statistics of the values as stored (rounded to bf16).
//
// TAPS = 9 (r06): the 3 x 3 / stride 1 / padding 1 convolution 64 -> 64 of the same stage (resnet.py:112, 84 % of stage 1's
// flops) and its data gradient as the SAME stream: the reduction is 9 taps x 64 channels, tap (r, s) of pixel (y, x) is the 128-byte
// row of pixel (y + r - 1, x + s - 1) -- one more address per tap, a buffer offset beyond the tensor (hardware zeros) where the tap
// leaves the image.  The 72 KiB of weights do not fit registers: they sit in LDS (64 rows of 9 * 128 + 32 bytes, conflict-free
// 16-byte fragment reads), one workgroup of 8 streams per CU shares them.  The activations go through a WAVE-PRIVATE LDS window:
// the three image rows around a row group, 18 pixels each, are fetched once per row group (9 loads of 16 bytes per lane, prefetched
// one row group ahead) and the nine taps read their fragments from there -- the first version fetched every tap from L1 / L2
// (18 loads per row group, 0.9 GB through the L2s per launch: 110 us, bound by exactly that, profiles/r06_nt_experiments.md).  One
// wavefront's LDS operations execute in order, so window writes, fragment reads and the staged output strip (which reuses the
// window) need no barrier.  A stream walks CONSECUTIVE row groups: the rows a neighbouring group needs again come from the XCD's L2.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "saicv_internal.h"

namespace {

struct PWParams {
    const bf16_t* src;          // [M][KD]
    const bf16_t* wgt;          // [ND][KD]
    bf16_t* out;                // [M][ND]
    float* stat_sum;            // forward BatchNorm statistics [rows][ND] (nullptr: none)
    float* stat_sq;
    int stat_atomic_rows;       // > 0: both kinds of sums are ADDED into this many zeroed rows; 0: row = workgroup
    const bf16_t* addend;       // out += addend (where its gate bit is set)
    const uint8_t* addend_gate; // one byte per 16-byte chunk of out
    const bf16_t* bs_y;         // BatchNorm-backward sums of the stored output: g = out * [mask], bs_g = sum g,
    const uint8_t* bs_mask;     //   bs_gx = sum g * (y - mean) * invstd
    const float* bs_mean;
    const float* bs_invstd;
    float* bs_g;
    float* bs_gx;
    uint32_t src_bytes;
    int M, mtiles, units;       // rows, 16-row groups, wavefront groups of the launch
    int stream_out;
    int H, W, tap_sign;         // TAPS = 9: image size; +1: source pixel (y + r - 1, x + s - 1) (forward), -1: (y + 1 - r, x + 1 - s) (data gradient)
    int tiles_per_unit;         // TAPS = 9: a stream walks this many CONSECUTIVE row groups (vertical taps meet in L1 / the XCD's L2)
};

// Weight fragments from LDS (TAPS = 9) as assembly statements in a fixed order with COUNTED waits, as igemm.hip's K loop does: left
// to the compiler every MFMA waits for a read issued one instruction earlier (the LDS round trip, 72 times per row group: 115 us per
// launch measured).  The reads of k-step ks + 1 are in flight while the four MFMAs of k-step ks run; the wait names the registers it
// releases, so the MFMAs cannot be scheduled above it.
typedef __attribute__((address_space(3))) const char lds_cchar_t;
DEVINL uint32_t lds_addr32(const char* q) { return (uint32_t)reinterpret_cast<uintptr_t>((lds_cchar_t*)q); }
template <int IMM> DEVINL void lds_rd128(u32x4& d, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(IMM) : "memory");
}
template <int N> DEVINL void lgkm_release(u32x4& a, u32x4& b, u32x4& c, u32x4& d, u32x4& e) {
    asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e) : "n"(N) : "memory");
}
template <int B, int E, typename F> DEVINL void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// Output stores are assembly statements in BOTH forms (streaming "nt" and plain): gfx9-family code has one counter (vmcnt) for loads
// and stores, and with a store pending next to loads the compiler treats the counter as out of order and waits vmcnt(0) before the next
// use of a loaded register -- at the head of every pass of the stream loop, which drains the row groups prefetched DEPTH ahead (r06:
// seen in the disassembly; the K = 128 / 256 forms with two groups in flight lost most of their prefetch to it).  Stores the compiler
// does not see leave it counting loads only, in order.  Nothing in the kernel reads `out` back; the wavefront's stores complete before
// the kernel ends.
DEVINL void st_stream(void* q, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(q), "v"(v) : "memory");
}
DEVINL void st_plain(void* q, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(q), "v"(v) : "memory");
}

constexpr int pw_nwaves(int nsplit, int taps) { return taps > 1 ? 8 : nsplit > 4 ? nsplit : 4; }

// CK input channels per tap (TAPS = 1: pointwise, 9: 3 x 3 / stride 1 / padding 1), ND output channels; NSPLIT wavefronts share a row
// group, 64 channels each (ND = 64 * NSPLIT)
template <int CK, int ND, int NSPLIT, bool STATS, bool EXTRAS, int TAPS = 1>
__global__ __launch_bounds__(64 * pw_nwaves(NSPLIT, TAPS)) void pw_stream_kernel(const PWParams p) {
    constexpr int KD = CK * TAPS;
    constexpr int NWAVES = pw_nwaves(NSPLIT, TAPS);
    constexpr int GPB = NWAVES / NSPLIT;          // row-group streams per workgroup
    constexpr int NT = 4, KS = KD / 32;           // MFMA row tiles (channels) and k-steps per wavefront
    constexpr int CPR = 8, RPP = 8, NPASS = 2;    // staged strip: 16 rows x 128 bytes, copied out 8 rows per pass
    constexpr int PITCH = 128 + 16;
    constexpr int DEPTH = TAPS > 1 ? 1 : KD <= 64 ? 4 : 2;       // row groups in flight in registers
    constexpr bool WLDS = TAPS > 1;               // the weights live in LDS
    // LDS pitches of the nine-tap form, conflict-free for ds_read_b128's lane groups ({0-3, 12-15, 20-27}, ... -- MI355X_MICROARCH.md, LDS):
    // 16 rows (pixels) at this pitch with the 16-byte k-group offsets of a fragment read meet 64 distinct banks in every group
    constexpr int WPITCH = KD * 2 + 32;           //   bytes per weight row
    constexpr int PW = CK * 2 + 32;               //   bytes per pixel of the wavefront's input window
    constexpr int WIN_PX = 18;                    //   pixels per window row: the 16 of the row group + one on each side
    constexpr int WIN_BYTES = 3 * WIN_PX * PW;    //   three image rows; + 128 zero bytes a tap outside the image reads instead
    constexpr int WAVE_LDS = WIN_BYTES + 128;
    constexpr uint32_t OOB = 0xfffffff0u;
    static_assert(ND == 64 * NSPLIT && KD % 32 == 0 && (TAPS == 1 || (TAPS == 9 && CK == 64)), "64 channels per wavefront");
    extern __shared__ __attribute__((aligned(16))) char wlds[];
    __shared__ __attribute__((aligned(16))) char strip[TAPS > 1 ? 1 : NWAVES][16 * PITCH];     // (nine taps: the strip reuses the window)
    static_assert(16 * PITCH <= WIN_BYTES, "the staged strip fits the window");
    __shared__ float red[(STATS || EXTRAS) ? NWAVES * 2 * 64 : 1];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int slice = wave % NSPLIT;              // which 64 channels
    const int nc0 = slice * 64;
    const int unit = blockIdx.x * GPB + wave / NSPLIT;
    const int U = p.units;

    // ---- weights: fragment (nt, ks) = rows nc0 + nt*16 + l15, k = ks*32 + lg*8 .. +7
    u32x4 wf[WLDS ? 1 : NT][WLDS ? 1 : KS];
    if constexpr (WLDS) {
        constexpr int CPROW = KD / 8;             // 16-byte chunks per weight row
        for (int c = threadIdx.x; c < ND * CPROW; c += 64 * NWAVES) {
            const int row = c / CPROW, ch = c - row * CPROW;
            *reinterpret_cast<u32x4*>(wlds + row * WPITCH + ch * 16) = ld_chunk(p.wgt + (size_t)row * KD + ch * 8);
        }
        __syncthreads();
    } else {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                wf[nt][ks] = ld_chunk(p.wgt + (size_t)(nc0 + nt * 16 + l15) * KD + ks * 32 + lg * 8);
    }
    const uint32_t waddr = lds_addr32(wlds + (nc0 + l15) * WPITCH + lg * 16);      // WLDS: this lane's fragment (nt, ks) at + nt * 16 rows + ks * 64 bytes

    const __amdgpu_buffer_rsrc_t src_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.src), 0, p.src_bytes, 0x00020000);
    u32x4 bfr[DEPTH][TAPS > 1 ? 1 : KS];
    // the stream's i-th row group: strided over the launch (pointwise: the chip reads one advancing window) or consecutive (TAPS = 9)
    const int count = TAPS == 1 ? (unit < p.mtiles ? (p.mtiles - unit + U - 1) / U : 0)
                                : max(0, min(p.tiles_per_unit, p.mtiles - unit * p.tiles_per_unit));
    auto tile_of = [&](int i) { return i >= count ? p.mtiles : TAPS == 1 ? unit + i * U : unit * p.tiles_per_unit + i; };   // p.mtiles: no row group
    const uint32_t lane_off = (uint32_t)l15 * (CK * 2) + (uint32_t)lg * 16;
    auto load_tile = [&](u32x4 (&dst)[TAPS > 1 ? 1 : KS], int tile) __attribute__((always_inline)) {
        if constexpr (TAPS == 1) {
            const uint32_t base = tile < p.mtiles ? (uint32_t)tile * (16 * KD * 2) + lane_off : OOB;       // rows past M read zeros (buffer bounds)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                dst[ks] = __builtin_amdgcn_raw_buffer_load_b128(src_rs, (int)(tile < p.mtiles ? base + ks * 64 : OOB), 0, 0);
        }
    };
    // TAPS = 9: the three image rows around a row group, 18 pixels each (pixel index q0 + px, q0 = first pixel + (row - 1) * W - 1), as
    // 16-byte chunks: lane -> (pixel (lane >> 3) + 8 j, chunk lane & 7), j = 0 .. 2.  Pixels before the tensor / past M read zeros; a
    // pixel of the neighbouring image row or image is loaded as it is -- the taps that would use it are redirected to zeros below.
    char* const win = WLDS ? wlds + ND * WPITCH + wave * WAVE_LDS : nullptr;
    u32x4 pf[WLDS ? 9 : 1];
    auto load_rows = [&](int tile) __attribute__((always_inline)) {
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int px = (lane >> 3) + 8 * j;
                const int q = tile * 16 + (rr - 1) * p.W - 1 + px;
                const bool ok = tile < p.mtiles && px < WIN_PX && (unsigned)q < (unsigned)p.M;
                uint32_t off = ok ? (uint32_t)q * (CK * 2) + (uint32_t)(lane & 7) * 16 : OOB;
                asm volatile("" : "+v"(off));
                pf[rr * 3 + j] = __builtin_amdgcn_raw_buffer_load_b128(src_rs, (int)off, 0, 0);
            }
    };
    if constexpr (WLDS) {
        if (lane < 8) *reinterpret_cast<u32x4*>(win + WIN_BYTES + lane * 16) = u32x4{0u, 0u, 0u, 0u};
        load_rows(tile_of(0));
    } else {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) load_tile(bfr[d], tile_of(d));
    }

    float ssum[8], ssq[8];        // STATS: sum / sum of squares; EXTRAS: sum g / sum g * y   of this lane's 8 channels
#pragma unroll
    for (int e = 0; e < 8; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; }
    const bool addp = EXTRAS && p.addend != nullptr, gatep = EXTRAS && p.addend_gate != nullptr;
    const bool bsp = EXTRAS && p.bs_y != nullptr, maskp = EXTRAS && p.bs_mask != nullptr;
    const bool stream_out = p.stream_out != 0;
    char* const my = WLDS ? win : strip[WLDS ? 0 : wave];
    const int crow = lane / CPR, cchunk = lane % CPR;             // copy-out coordinates: row of the pass, 16-byte chunk of the 128-byte row

    // One row group of slot D (a compile-time index: the slots are registers).  The loop below has NO path that skips a slot: with a
    // `break` between the slots the compiler's counter analysis sees a way round the loop on which slot 0's refill is the newest load
    // and waits for the younger slots' refills as well (vmcnt(17 - k) instead of vmcnt(35 - k) in the disassembly) -- the prefetch
    // distance collapses to one row group.  Whole passes first, the count % DEPTH remaining groups afterwards.
    auto step = [&](auto D, int i) __attribute__((always_inline)) {
        constexpr int d = decltype(D)::value;
        const int tile = tile_of(i);
        // fused operands of the two passes: requested before the products, consumed behind the staging round trip
        u32x4 av[NPASS], yv[NPASS];
        unsigned gb[NPASS], mb[NPASS];
        size_t ooff[NPASS];
        bool rok[NPASS];
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int row = tile * 16 + ps * RPP + crow;
            rok[ps] = row < p.M;
            ooff[ps] = (size_t)row * ND + nc0 + cchunk * 8;
            av[ps] = u32x4{0u, 0u, 0u, 0u};
            yv[ps] = u32x4{0u, 0u, 0u, 0u};
            gb[ps] = 0xffu;
            mb[ps] = 0xffu;
            if (EXTRAS && rok[ps]) {
                if (addp) av[ps] = ld_chunk(p.addend + ooff[ps]);
                if (bsp) yv[ps] = ld_chunk(p.bs_y + ooff[ps]);
                if (gatep) gb[ps] = p.addend_gate[ooff[ps] >> 3];
                if (maskp) mb[ps] = p.bs_mask[ooff[ps] >> 3];
            }
        }
        f32x4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (WLDS) {
            static_assert(NT == 4, "four fragments per k-step");
            // the prefetched rows of THIS row group -> window (chunks of one pixel are 128 contiguous bytes: conflict-free stores)
#pragma unroll
            for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int px = (lane >> 3) + 8 * j;
                    if (px < WIN_PX) *reinterpret_cast<u32x4*>(win + (rr * WIN_PX + px) * PW + (lane & 7) * 16) = pf[rr * 3 + j];
                }
            // this lane's pixel and, per tap, where its 16-byte k-group sits in the window (outside the image / past M: the zero chunk).
            // (the coordinates are computed for every lane and pinned: left free, the compiler moves the divisions under a divergent branch)
            const int pix = tile * 16 + l15;
            const bool live = pix < p.M;
            const int row = pix / p.W;
            int x = pix - row * p.W;
            int y = row % p.H;
            asm volatile("" : "+v"(x), "+v"(y));
            const uint32_t win32 = lds_addr32(win);
            uint32_t baddr[9];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const int dy = p.tap_sign * (r - 1), dx = p.tap_sign * (t - 1);
                    const bool ok = live && (unsigned)(y + dy) < (unsigned)p.H && (unsigned)(x + dx) < (unsigned)p.W;
                    baddr[r * 3 + t] = win32 + (ok ? (uint32_t)(((dy + 1) * WIN_PX + l15 + dx + 1) * PW) : (uint32_t)WIN_BYTES) + (uint32_t)lg * 16;
                }
            load_rows(tile_of(i + 1));                            // the next row group's rows travel during the products
            // fragment reads of k-step k + 1 (four of the weights, one of the window) in flight during the MFMAs of k-step k
            u32x4 wa[2][NT], wb[2];
            auto issue = [&](auto KSI) __attribute__((always_inline)) {
                constexpr int k = decltype(KSI)::value;
                static_for<0, NT>([&](auto NI) { lds_rd128<decltype(NI)::value * 16 * WPITCH + k * 64>(wa[k & 1][decltype(NI)::value], waddr); });
                lds_rd128<(k % (CK / 32)) * 64>(wb[k & 1], baddr[k / (CK / 32)]);
            };
            issue(std::integral_constant<int, 0>{});
            static_for<0, KS>([&](auto KSI) {
                constexpr int k = decltype(KSI)::value;
                if constexpr (k + 1 < KS) issue(std::integral_constant<int, k + 1>{});
                lgkm_release<(k + 1 < KS) ? NT + 1 : 0>(wa[k & 1][0], wa[k & 1][1], wa[k & 1][2], wa[k & 1][3], wb[k & 1]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) Mma<bf16_t>::run(acc[nt], wa[k & 1][nt], wb[k & 1]);
                __builtin_amdgcn_sched_barrier(0);
            });
        } else {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) Mma<bf16_t>::run(acc[nt], wf[nt][ks], bfr[d][ks]);
            load_tile(bfr[d], tile_of(i + DEPTH));           // the slot is free again: DEPTH groups ahead
        }
        // accumulators -> strip: D row lg*4 + r of tile nt = channel nt*16 + lg*4 + r, D column = pixel l15
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            bf16x4 pk;
#pragma unroll
            for (int r = 0; r < 4; ++r) pk[r] = (bf16_t)acc[nt][r];
            *reinterpret_cast<bf16x4*>(my + l15 * PITCH + (nt * 16 + lg * 4) * 2) = pk;
        }
        __builtin_amdgcn_wave_barrier();                      // (compiler ordering only: one wavefront's LDS operations execute in order)
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            u32x4 v = ld_chunk(my + (ps * RPP + crow) * PITCH + cchunk * 16);
            float f[8];
            if (EXTRAS && (addp || bsp)) {
                Chunk<bf16_t>::unpack(v, f);
                if (addp) {
                    float a[8];
                    Chunk<bf16_t>::unpack(av[ps], a);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] += ((gb[ps] >> e) & 1u) ? a[e] : 0.f;
                    v = Chunk<bf16_t>::pack(f);
                    if (bsp) Chunk<bf16_t>::unpack(v, f);     // the sums are over what is stored
                }
                if (bsp) {
                    float yy[8];
                    Chunk<bf16_t>::unpack(yv[ps], yy);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float ge = ((mb[ps] >> e) & 1u) ? f[e] : 0.f;
                        ssum[e] += ge;
                        ssq[e] = fmaf(ge, yy[e], ssq[e]);
                    }
                }
            }
            if (STATS) {
                Chunk<bf16_t>::unpack(v, f);
#pragma unroll
                for (int e = 0; e < 8; ++e) { ssum[e] += f[e]; ssq[e] = fmaf(f[e], f[e], ssq[e]); }
            }
            if (rok[ps]) {
                if (stream_out) st_stream(p.out + ooff[ps], v);
                else st_plain(p.out + ooff[ps], v);
            }
        }
        __builtin_amdgcn_wave_barrier();
    };
    int i0 = 0;
    for (; i0 + DEPTH <= count; i0 += DEPTH) static_for<0, DEPTH>([&](auto D) { step(D, i0 + decltype(D)::value); });
    static_for<0, DEPTH - 1>([&](auto D) {
        if (i0 + decltype(D)::value < count) step(D, i0 + decltype(D)::value);      // wave-uniform
    });

    if constexpr (STATS || EXTRAS) {
        const bool want = STATS ? p.stat_sum != nullptr : bsp;     // uniform
        if (want) {
            if (EXTRAS) {
                // sum g * (y - mean) * invstd = invstd * (sum g y - mean * sum g): the mean leaves while the sums are this lane's few
                // dozen rows -- not after the whole column (cancellation)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int c = nc0 + cchunk * 8 + e;
                    ssq[e] = p.bs_invstd[c] * fmaf(-p.bs_mean[c], ssum[e], ssq[e]);
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {                          // the 8 row lanes of a chunk column: lanes cchunk + 8 j
                ssum[e] += __shfl_xor(ssum[e], 8, 64);  ssq[e] += __shfl_xor(ssq[e], 8, 64);
                ssum[e] += __shfl_xor(ssum[e], 16, 64); ssq[e] += __shfl_xor(ssq[e], 16, 64);
                ssum[e] += __shfl_xor(ssum[e], 32, 64); ssq[e] += __shfl_xor(ssq[e], 32, 64);
            }
            if (lane < CPR) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    red[(wave * 2 + 0) * 64 + lane * 8 + e] = ssum[e];
                    red[(wave * 2 + 1) * 64 + lane * 8 + e] = ssq[e];
                }
            }
        }
        __syncthreads();
        if (want) {
            float* const d0 = STATS ? p.stat_sum : p.bs_g;
            float* const d1 = STATS ? p.stat_sq : p.bs_gx;
            const size_t row = p.stat_atomic_rows ? (size_t)(blockIdx.x % p.stat_atomic_rows) : (size_t)blockIdx.x;
            for (int c = threadIdx.x; c < 2 * ND; c += 64 * NWAVES) {
                const int which = c / ND, col = c - which * ND;
                const int sl = col / 64, cc = col - sl * 64;
                float a = 0.f;
#pragma unroll
                for (int g = 0; g < GPB; ++g) a += red[((g * NSPLIT + sl) * 2 + which) * 64 + cc];       // fixed order over the streams
                float* dst = (which ? d1 : d0) + row * ND + col;
                if (p.stat_atomic_rows) unsafeAtomicAdd(dst, a); else *dst = a;
            }
        }
    }
}

struct PWShape { int kd, nd, nsplit; };
constexpr PWShape kShapes[] = {{64, 64, 1}, {64, 256, 4}, {256, 64, 1}, {128, 128, 2}, {128, 512, 8}, {128, 256, 4}};

int blocks_per_cu() {
    static const int v = getenv("SAICV_PW_BPC") ? atoi(getenv("SAICV_PW_BPC")) : 2;
    return v < 1 ? 1 : v > 8 ? 8 : v;
}

template <int KD, int ND, int NSPLIT>
int launch(const PWParams& p, int blocks, bool stats, bool extras, hipStream_t st) {
    constexpr int NWAVES = NSPLIT > 4 ? NSPLIT : 4;
    dim3 grid(blocks), block(64 * NWAVES);
    if (stats) hipLaunchKernelGGL((pw_stream_kernel<KD, ND, NSPLIT, true, false>), grid, block, 0, st, p);
    else if (extras) hipLaunchKernelGGL((pw_stream_kernel<KD, ND, NSPLIT, false, true>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((pw_stream_kernel<KD, ND, NSPLIT, false, false>), grid, block, 0, st, p);
    return saicv::check_launch("pw_stream");
}

// the 3 x 3 form: 64 -> 64 channels, one workgroup of 8 streams per CU; dynamic LDS = weights (64 rows of 1 184 bytes) + 8 windows
int launch_taps9(const PWParams& p, int blocks, bool stats, bool extras, hipStream_t st) {
    constexpr size_t smem = (size_t)64 * (576 * 2 + 32) + 8 * (size_t)(3 * 18 * (64 * 2 + 32) + 128);      // (kernel: ND * WPITCH + NWAVES * WAVE_LDS)
    dim3 grid(blocks), block(64 * pw_nwaves(1, 9));
#define PW3_LAUNCH(ST, EX)                                                                                                        \
    {                                                                                                                             \
        auto k = pw_stream_kernel<64, 64, 1, ST, EX, 9>;                                                                          \
        static bool once = (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem), true); \
        (void)once;                                                                                                               \
        hipLaunchKernelGGL(k, grid, block, smem, st, p);                                                                          \
    }
    if (stats) PW3_LAUNCH(true, false)
    else if (extras) PW3_LAUNCH(false, true)
    else PW3_LAUNCH(false, false)
#undef PW3_LAUNCH
    return saicv::check_launch("pw_stream (3 x 3)");
}

}  // namespace

namespace saicv {

// Workgroups (= rows of partial statistics) of the streaming launch for a pointwise bf16 product [M][Kd] x [Nn][Kd]^T, 0 if this
// product stays on the tiled kernel (fused_dgrad: a data gradient with a shortcut addend or BatchNorm-backward sums).  A pure function of its arguments and of SAICV_PW_STREAM / SAICV_PW_MIN_ROWS / SAICV_PW_BPC.
int pw_stream_blocks(int dtype, int M, int Nn, int Kd, bool fused_dgrad) {
    const char* es = getenv("SAICV_PW_STREAM");           // (read per call: tests and tuning sweeps flip them in-process)
    const char* er = getenv("SAICV_PW_MIN_ROWS");
    const int on = es ? atoi(es) : 1;
    const int min_rows = er ? atoi(er) : 65536;
    if (!on || dtype != SAICV_DTYPE_BF16 || M < min_rows) return 0;
    for (const PWShape& s : kShapes) {
        if (s.kd != Kd || s.nd != Nn) continue;
        // K = 256: only the data gradient with fused operands (143 us against 159 tiled); forward + statistics and the plain data
        // gradient measured 129-141 us here against 104 us on the tiled kernel (profiles/r06_nt_experiments.md)
        if (Kd >= 256 && !fused_dgrad) return 0;
        const int nwaves = s.nsplit > 4 ? s.nsplit : 4;
        const int gpb = nwaves / s.nsplit;
        const int mtiles = (M + 15) / 16;
        const int want = (mtiles + gpb - 1) / gpb;
        const int cap = 256 * blocks_per_cu() * 4 / nwaves;
        return want < cap ? want : cap;
    }
    return 0;
}

// -> 1 launched, 0 not eligible (the caller takes the tiled kernel), < 0 error
int pw_stream(int M, int Nn, int Kd, const void* src, const void* wgt, void* out, float* stat_sum, float* stat_sq,
              int stat_atomic_rows, const EpiExtra* ex, int stream_out, hipStream_t st) {
    const bool extras = ex && (ex->addend || ex->bs_y);
    const int blocks = pw_stream_blocks(SAICV_DTYPE_BF16, M, Nn, Kd, extras && !stat_sum);
    if (blocks == 0) return 0;
    if (stat_sum && extras) return 0;
    PWParams p = {};
    p.src = (const bf16_t*)src; p.wgt = (const bf16_t*)wgt; p.out = (bf16_t*)out;
    p.stat_sum = stat_sum; p.stat_sq = stat_sq;
    p.stat_atomic_rows = stat_atomic_rows;
    if (ex) {
        p.addend = (const bf16_t*)ex->addend; p.addend_gate = ex->addend_gate;
        p.bs_y = (const bf16_t*)ex->bs_y; p.bs_mask = ex->bs_mask; p.bs_mean = ex->bs_mean; p.bs_invstd = ex->bs_invstd;
        p.bs_g = ex->bs_g; p.bs_gx = ex->bs_gx;
    }
    const size_t src_bytes = (size_t)M * Kd * 2;
    if (src_bytes >= 0xfffffff0ull) return 0;
    p.src_bytes = (uint32_t)src_bytes;
    p.M = M;
    p.mtiles = (M + 15) / 16;
    p.stream_out = stream_out;
    const bool stats = stat_sum != nullptr;
#define PW_CASE(KD, ND, NS)                                                        \
    if (Kd == KD && Nn == ND) {                                                    \
        constexpr int NW = NS > 4 ? NS : 4;                                        \
        p.units = blocks * (NW / NS);                                              \
        const int rc = launch<KD, ND, NS>(p, blocks, stats, extras, st);           \
        return rc ? rc : 1;                                                        \
    }
    PW_CASE(64, 64, 1)
    PW_CASE(64, 256, 4)
    PW_CASE(256, 64, 1)
    PW_CASE(128, 128, 2)
    PW_CASE(128, 512, 8)
    PW_CASE(128, 256, 4)
#undef PW_CASE
    return 0;
}

// The 3 x 3 / stride 1 / padding 1 form (64 -> 64 channels): workgroups (= rows of partial statistics), 0 if the tiled kernel keeps it.
int pw3_stream_blocks(int dtype, int M, int Nn, int Kd) {
    const char* es = getenv("SAICV_PW_STREAM3");           // (read per call)
    const char* er = getenv("SAICV_PW_MIN_ROWS");
    const int on = es ? atoi(es) : 1;
    const int min_rows = er ? atoi(er) : 65536;
    if (!on || dtype != SAICV_DTYPE_BF16 || M < min_rows || Nn != 64 || Kd != 576) return 0;
    const int mtiles = (M + 15) / 16;
    const int want = (mtiles + 7) / 8;
    return want < 256 ? want : 256;
}

// src [N][H][W][64] (M = N * H * W rows), wgt [64][(r, s, c)]; mode 0: out(y, x) = sum wgt(r, s) . src(y + r - 1, x + s - 1), mode 1 (the
// data gradient over the packed transposed weights): src(y + 1 - r, x + 1 - s).  -> 1 launched, 0 not eligible, < 0 error
int pw3_stream(int mode, int M, int H, int W, const void* src, const void* wgt, void* out, float* stat_sum, float* stat_sq,
               int stat_atomic_rows, const EpiExtra* ex, int stream_out, hipStream_t st) {
    const int blocks = pw3_stream_blocks(SAICV_DTYPE_BF16, M, 64, 576);
    if (blocks == 0) return 0;
    const bool extras = ex && (ex->addend || ex->bs_y);
    if (stat_sum && extras) return 0;
    PWParams p = {};
    p.src = (const bf16_t*)src; p.wgt = (const bf16_t*)wgt; p.out = (bf16_t*)out;
    p.stat_sum = stat_sum; p.stat_sq = stat_sq;
    p.stat_atomic_rows = stat_atomic_rows;
    if (ex) {
        p.addend = (const bf16_t*)ex->addend; p.addend_gate = ex->addend_gate;
        p.bs_y = (const bf16_t*)ex->bs_y; p.bs_mask = ex->bs_mask; p.bs_mean = ex->bs_mean; p.bs_invstd = ex->bs_invstd;
        p.bs_g = ex->bs_g; p.bs_gx = ex->bs_gx;
    }
    const size_t src_bytes = (size_t)M * 64 * 2;
    if (src_bytes >= 0xffffff00ull - 256 || H < 1 || W < 1 || M % (H * W) != 0) return 0;
    p.src_bytes = (uint32_t)src_bytes;
    p.M = M;
    p.mtiles = (M + 15) / 16;
    p.units = blocks * 8;
    p.tiles_per_unit = (p.mtiles + p.units - 1) / p.units;
    p.H = H; p.W = W;
    p.tap_sign = mode == 0 ? 1 : -1;
    p.stream_out = stream_out;
    const int rc = launch_taps9(p, blocks, stat_sum != nullptr, extras, st);
    return rc ? rc : 1;
}

}  // namespace saicv
