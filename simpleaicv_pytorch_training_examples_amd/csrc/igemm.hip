// Implicit-GEMM convolution / linear kernels for gfx950 (MI355X), NHWC.
//
//   igemm_nt : OUT[m][n] = sum_k GATHER(src)[m][k] * WGT[n][k]
//              conv forward (rows = output pixels, k = (r,s,c)), conv data-gradient
//              (rows = input pixels, k = (r,s,cout)) and nn.Linear forward / input-grad.
//   igemm_tn : DW[n][kk] += sum_m DY[m][n] * GATHER(src)[m][kk]
//              conv / linear weight-gradient; both operands are reduction-major in HBM,
//              so bf16 fragments are built with the LDS transpose read
//              (ds_read_b64_tr_b16) and the reduction over pixels is split across
//              workgroups with fp32 hardware atomics into a pre-zeroed fp32 gradient.
//
// Replaces the ATen `convolution` / `convolution_backward` / `addmm` / `mm` dispatches of
// reference SimpleAICV/classification/backbones/resnet.py:33-43 (ConvBnActBlock) and
// resnet.py:204 (fc).  NT: four tile geometries (256x256 / 8 wavefronts, 256x128 / 8, 128x128 / 4, 128x64 / 4), 64-byte
// K slices streamed global -> LDS by DMA (buffer_load ... lds) into a 3- or 4-stage ring, counted vmcnt and one raw
// barrier per K step, MFMA 16x16x32 bf16 (perf mode) or 16x16x4 f32 (parity mode), an XOR swizzle that makes the
// ds_read_b128 fragment reads conflict free, LDS-staged epilogue with the fused modes listed at NTParams.
// TN: 64x64 .. 256x256 tiles, register-staged loads, ds_read_b64_tr_b16 fragments, one resident round of workgroups.
#include <type_traits>

#include <stdlib.h>
#include "common.h"
#include "saicv_internal.h"
#include "det.h"

namespace {

struct FastDiv {           // exact unsigned division by a runtime constant (Granlund-Montgomery)
    uint32_t mul, s1, s2;
};
DEVINL uint32_t fdiv(uint32_t n, FastDiv d) {
    const uint32_t t = __umulhi(d.mul, n);
    return (t + ((n - t) >> d.s1)) >> d.s2;
}

// LDS-DMA ring depth per geometry: 4 stages, except 256 x 128 where 3 stages (72 KiB) let TWO workgroups share
// a CU so that one's epilogue overlaps the other's main loop
constexpr int nt_stages(int bm, int bn) { return (bm == 256 && bn == 128) ? 3 : 4; }
constexpr int nt_stages_kc8(int bm, int bn) { return (bm + bn) * 128 * 3 <= 150 * 1024 ? 3 : 2; }      // 128-byte K slices
// resident workgroups per CU by LDS (the ring is all a workgroup holds: the epilogue stages through a vacated slot)
// Which instantiations may run persistently (ticket draws hidden from the compiler, see draw_ticket): bf16 in and out,
// and a register budget in which the compiler spills nothing -- a spilled ticket register would be saved before the atomic
// has landed.  The 8-wavefront 256 x 128 geometry lives at 128 registers per lane (two workgroups per CU) and does spill
// in its epilogue; it stays one tile per workgroup.  scripts/check_ticket_regs.py verifies the generated code of the rest.
constexpr bool nt_can_persist(int elem_bytes, bool out_f32, int bm, int bn, int nwaves) {
    return elem_bytes == 2 && !out_f32 && !(nwaves == 8 && bm * bn <= 256 * 128);
}
constexpr int nt_blocks_per_cu(int bm, int bn) { return 160 * 1024 / (nt_stages(bm, bn) * (bm + bn) * 64 + 16); }

struct NTParams {
    const void* src;
    const void* wgt;
    void* out;
    const float* bias;
    float* stat_sum;
    float* stat_sq;
    const void* addend;         // epilogue: out = addend + row_scale * (acc + bias)   (act_mode 2: the GELU pre-activation)
    void* out2;                 // act_mode 1: second output, gelu(out)
    int lean_gelu;              // 1: act modes 1 / 3 without addend / row scale take the lean epilogue row loop (SAICV_GELU_EPI=0: the general one)
    int act_mode;               // 0 none | 1 out2 = gelu(out) | 2 out = (acc + bias) * gelu'(addend) | 3 out = gelu'(pre), out2 = gelu(pre) | 4 out = (acc + bias) * addend
    const float* row_scale;
    int rows_per_scale;
    // data-gradient epilogue extras of a residual network (all tensors share out's [M][ldo] coordinates):
    const uint8_t* addend_gate; // addend counts only where its bit is set (the shortcut gradient behind a ReLU: one bit per
                                // element, one byte per 16-byte chunk -- the mask bn_act_fwd wrote)
    const void* bs_y;           // BatchNorm-backward partial sums of THIS output, for the BatchNorm whose dz it is:
    const uint8_t* bs_mask;     //   g = out * [mask bit];  bs_g[row][c] = sum g,  bs_gx[row][c] = sum g * (y - mean) * invstd
    const float* bs_mean;       //   over the rows of one tile; row = parity class * bs_rows + tile_m
    const float* bs_invstd;
    float* bs_g;
    float* bs_gx;
    int bs_rows;
    // > 0: the statistics (forward sum / sum of squares, backward bs_g / bs_gx) are ADDED with fp32 atomics into this many
    // rows (row = tile row mod count) of a zeroed buffer instead of written one row per tile row: few enough rows that the
    // consuming BatchNorm kernel finalises them itself (no partial-reduce / finalize launches)
    int stat_atomic_rows;
    unsigned int* tickets;      // persistent launch: 8 per-XCD ticket counters (+ a departure counter at [32]), zero between launches
    int stagger_phases;         // persistent launch: workgroups start in this many phase groups ...
    int stagger_sleeps;         // ... `s_sleep 16` periods (~0.5 us each) apart, so that their epilogue store bursts interleave
    uint32_t src_bytes, wgt_bytes;
    int H, W, C;        // gather-source spatial dims / channels
    int OH, OW;         // pixel grid that indexes the GEMM rows
    int R, S, stride, pad;
    int M, Nn, Kd;
    int ldo;
    int tiles_n;
    int nblk;
    int grid_x;         // resident workgroups (256 CUs x workgroups per CU)
    int kc8;            // host: launch the 128-byte-K-slice instantiation (pointwise bf16, 256-row tiles)
    int stream_out;     // host: the output is written with streaming stores (see NT_OUT_ST)
    FastDiv fd_ohw, fd_ow;   // for OH*OW and OW (unit-stride row decomposition)
#ifdef SAICV_NT_TIMELINE
    unsigned long long* timeline;       // debug build only (scripts/nt_timeline.py): 8 shader-clock stamps per workgroup
#endif
};
#ifdef SAICV_NT_TIMELINE
// Debug build: per-workgroup phase stamps of igemm_nt1_kernel.  Stamps live in scalar registers and are written once, by one lane,
// behind the last store of the workgroup -- nothing is added to the vector-memory queue the K loop counts.
#define NT_STAMP(i) do { if (tl_on) tl_t[i] = __builtin_readcyclecounter(); } while (0)
static unsigned long long* g_nt_timeline = nullptr;
#else
#define NT_STAMP(i) do { } while (0)
#endif
// r05 -- output stores of igemm_nt1_kernel's epilogue are STREAMING stores ("nt": the line is not kept in L2 / MALL).  A training step's
// outputs are read by a LATER kernel and are 25-411 MB each against 32 MB of L2 and 256 MB of MALL; written with the default policy
// they evict what the neighbouring kernels re-read (the operand panels other tiles of this launch share, the dy a weight-gradient
// kernel reads right after the data-gradient kernel did).  Same box, library A/B (profiles/r05_nt_experiments.md): ResNet-50 21.85 ->
// 21.27 ms per step (igemm_nt 10.82 -> 10.60, igemm_tn 4.51 -> 4.30, bn_act_fwd 2.63 -> 2.55), ViT-B 40.04 -> 39.34 ms (igemm_nt 21.39 -> 20.43).
// -DSAICV_NT_PLAIN_STORES builds the default-policy variant (scripts/build_variant_lib.py).  (A run-time choice between the two store
// forms does not survive the compiler: `if (flag) nontemporal_store else store` is merged into one plain store.)
// Which launches stream: outputs of at least SAICV_NT_STREAM_MIN_MB MiB (p.stream_out).  The streaming form is an assembly statement --
// a run-time choice between __builtin_nontemporal_store and a plain store does not survive the compiler (both arms are merged into ONE
// plain store; the first build of this switch measured exactly like plain stores).  `s_nop 1`: the statement's data registers may be
// rewritten right behind it (guide section 5.7).
DEVINL void st_chunk_stream(void* q, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(q), "v"(v) : "memory");
}
#ifdef SAICV_NT_PLAIN_STORES
#define NT_OUT_ST st_chunk
#else
#define NT_OUT_ST(ptr, v) do { if (stream_out) st_chunk_stream(ptr, v); else st_chunk(ptr, v); } while (0)
#endif
// the data gradient's fused epilogue operands (shortcut gradient, pre-BatchNorm output, two mask bytes per chunk) are read exactly once
#ifdef SAICV_DGRAD_EPI_LD_NT
#define NT_EPI_LD ld_chunk_nt
#else
#define NT_EPI_LD ld_chunk
#endif

// NT kernel LDS image: one K tile = 64-byte rows = 4 chunks; chunk c of row r lives at slot
// c ^ f((r>>2)&3), f = {0,2,3,1}.  ds_read_b128 is served in 16-lane groups that mix rows 0-3 /
// 12-15 of chunk c with rows 4-11 of chunk c^1 (MI355X_MICROARCH.md, LDS table): with this f
// the 16 lanes of every group hit 16 distinct 16-byte bank groups.
DEVINL int lds_swz(int q) { return ((q & 1) << 1) ^ ((q >> 1) * 3); }
DEVINL int lds_off(int row, int chunk) { return row * 64 + ((chunk ^ lds_swz((row >> 2) & 3)) << 4); }
// 128-byte rows (KC = 8): chunk c of row r at slot c ^ ((r >> 1) & 7).  Two rows share a 256-byte bank window, and the 16-lane
// groups of ds_read_b128 mix rows {0-3, 12-15} of chunk c with rows {4-11} of chunk c + 1: per row parity that is eight (row,
// chunk) pairs whose slots c ^ {0, 1, 6, 7} and (c + 1) ^ {2, 3, 4, 5} are all different.
DEVINL int lds_off128(int row, int chunk) { return row * 128 + (((chunk ^ (row >> 1)) & 7) << 4); }

DEVINL u32x4 buf_ld(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, 0);
}

// compile-time loop: f(std::integral_constant<int, I>) for I in [I0, N) -- immediates of assembly statements must be constants
template <int I, int N, typename F> DEVINL void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
// r05 -- fragment reads of the bf16 K loops as ASSEMBLY statements.  With the reads as plain loads hipcc's scheduler, aiming at the
// launch bound's register budget, sinks the loads of W fragments 1.. BETWEEN the MFMA groups and reuses one register quad for
// all of them: `ds_read_b128; s_waitcnt lgkmcnt(0); 4 (8) MFMAs` four times per K step, i.e. four exposed LDS round trips per
// step and wavefront (a lone 256 x 128 workgroup ran 1 150 cycles per step for 544 cycles of MFMA work -- profiles/r05_nt_timeline.md).
// Here every read of a step is issued up front in a fixed order (W fragment 0, all A fragments, the other W fragments) and each MFMA
// group is released by a COUNTED lgkmcnt wait that names the registers it releases (LDS returns in order), as igemm_tn_dma_kernel does.
DEVINL uint32_t lds_addr32(const void* q) {
    typedef __attribute__((address_space(3))) const char lds_cchar;
    return (uint32_t)reinterpret_cast<uintptr_t>((lds_cchar*)q);
}
template <int IMM> DEVINL void lds_rd128(u32x4& d, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(IMM) : "memory");
}
template <int N> DEVINL void lgkm_release(u32x4& a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N) : "memory"); }
template <int N> DEVINL void lgkm_release(u32x4& a, u32x4& b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory"); }

// r05 -- MFMA shape of igemm_nt1_kernel: the kernel is written over a fragment geometry (FR rows per fragment, NV accumulator values
// per lane, KSUB MFMAs per 64-byte LDS row), so that v_mfma_f32_32x32x16_bf16 could be tried against 16x16x32.  Why it was tried:
// the GEMM kernels run AT the 1 400 W socket limit on random operands (scripts/power_probe.py: 1 397 W at 2.22 GHz), and an isolated
// 64 x 64 wavefront tile fed from LDS sustains 1 622 TFLOP/s at that limit on the 32 x 32 shape against 1 418 on 16x16x32
// (scripts/probes/mfma_power_probe.hip).  What the models said, same box, library A/B (profiles/r05_nt_experiments.md): ViT-B 40.37 ->
// 41.19 ms, ResNet-50 22.0 -> 22.2 ms with the 32 x 32 shape -- it moves twice the accumulator registers per flop (16 read + 16
// written per K = 16) and the real loop, unlike the probe's, was never issue-bound.  16x16x32 stays; -DSAICV_NT_MFMA32 builds the other
// (scripts/build_variant_lib.py).  32 x 32 fragments are the two 16-byte chunks (lane >> 5) of one 32-byte half of a 64-byte LDS row: the DMA
// image and its XOR swizzle are unchanged and stay conflict free (the 16-lane groups of ds_read_b128 still meet all four swizzle
// classes); C/D value r of a lane is column lane & 31 (second operand), row 8 * (r / 4) + 4 * (lane >> 5) + r % 4 (first operand).
// fp32 (parity mode): 16x16x4, four per 16-byte chunk.
template <typename T> struct NtMma;
#ifdef SAICV_NT_MFMA32
template <> struct NtMma<bf16_t> {
    static constexpr int FR = 32, NV = 16, KSUB = 2;        // fragment rows, accumulator values per lane, MFMAs per 64-byte row
    typedef f32x16 Acc;
    static DEVINL void run(Acc& acc, const u32x4& a, const u32x4& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
};
#else
template <> struct NtMma<bf16_t> {
    static constexpr int FR = 16, NV = 4, KSUB = 1;
    typedef f32x4 Acc;
    static DEVINL void run(Acc& acc, const u32x4& a, const u32x4& b) { Mma<bf16_t>::run(acc, a, b); }
};
#endif
template <> struct NtMma<float> {
    static constexpr int FR = 16, NV = 4, KSUB = 1;
    typedef f32x4 Acc;
    static DEVINL void run(Acc& acc, const u32x4& a, const u32x4& b) { Mma<float>::run(acc, a, b); }
};

// Main loop = LDS-DMA ring: `buffer_load ... lds` writes global chunks straight into an NSTAGE-slot LDS ring (no staging
// registers, no ds_write), NSTAGE-1 K tiles are in flight across the single raw s_barrier of each K step, and the wait is
// a COUNTED s_waitcnt vmcnt(N) that only retires the tile about to be consumed (guide section 5, T3/T4).  The DMA
// destination is wave-linear (base + lane*16), so the XOR swizzle is applied to the SOURCE: lane l fetches the chunk that
// belongs in the slot it will fill.
// The loop is bound by MFMA issue only if the integer work per K tile is tiny, so:
//  * all gathers are raw buffer loads: an out-of-range lane (padding halo, M/N/K tails) gets
//    an offset beyond the buffer and the hardware writes zeros -- no branches, no exec masking;
//  * the (r, s, c) position of a thread's K chunk advances incrementally (no divisions);
//  * per-row constants fold image base and top-left corner, so a gather address is one add;
//  * LDS fragment addresses (with the XOR swizzle) are computed once.
// PLAIN: 1x1 taps without padding (pointwise conv, nn.Linear and their data-gradients): the K
// index IS the channel offset, no tap walker and no halo tests in the loop.
// Tile geometry (BM_T x BN_T output tile, WM_ x WN_ wavefronts, each owning a (BM_T/WM_) x (BN_T/WN_) sub-tile).
//
// r03 -- ONE OPERAND STREAM PER WORKGROUP, ACROSS TILES.  Round 2 measured a fixed cost of 10-12 us per output tile next
// to ~1 us per K step (kernel exit, dispatch of the next workgroup, kernel-argument loads, index set-up, the first DMA's
// HBM latency, a workgroup-wide LDS-staged epilogue): 30 % of a ViT-B K = 768 tile, 36 % of a 64-channel 3 x 3 one.  Now:
//  * a launch with more tiles than resident workgroups is PERSISTENT: a workgroup draws its next tile from a per-XCD
//    ticket counter (the tiles of one XCD stay a contiguous, L2-sharing range; late or slow workgroups simply draw fewer
//    tickets, so nothing depends on all of them being resident -- RCCL kernels may hold CU slots);
//  * the DMA stream does not drain at a tile boundary: the K steps of the NEXT tile are issued into the ring while the
//    last steps of the current one are computed (the per-row gather state lives in the same registers -- it is
//    recomputed at the switch, NSTAGE-1 steps before the compute side follows);
//  * the epilogue is PER WAVEFRONT and needs no LDS of its own: a wavefront stages 16 rows of its sub-tile at a time
//    through its share of the ring slot the last K step just vacated, reads them back as 16-byte row chunks (whole 128-byte
//    lines per row in HBM) and stores them; no workgroup barrier, the next tile's operands are landing meanwhile;
//  * BatchNorm statistics of bf16 outputs still come from the staged rows on the matrix cores (16x16x16: one transposed
//    LDS read per 16 rows x 16 columns; sum = 1 . Y, sum of squares = diag(Y^T . Y)), per wavefront.
// vmcnt accounting across the seam: gfx9 returns vector-memory operations in issue order (loads, stores and atomics share
// the counter), so "at most LPT x min(2, steps issued after this one) operations outstanding" retires the K step about to
// be consumed whatever epilogue stores or side loads were issued in between: they only make the wait conservative.
// FUSEDK: the instantiation carries the fused epilogue modes (residual / drop-path / GELU / gated shortcut / BatchNorm-backward
// sums, unaligned rows) -- or only the plain copy-out with BN statistics and bias.  Two kernels instead of two paths in one:
// the plain one stays clear of the register cap (no spill reloads, i.e. no compiler vmcnt(0), around the tile seam) and can
// count its own stores exactly.
template <typename T, int BM_T, int BN_T, int WM_, int WN_, int MODE, bool OUT_F32, bool PLAIN, bool FUSEDK>
__global__ __launch_bounds__(64 * WM_ * WN_, nt_blocks_per_cu(BM_T, BN_T) * WM_ * WN_ / 4)
void igemm_nt_kernel(const NTParams p) {
    constexpr int EPC = ElemTraits<T>::EPC;
    constexpr int BK = 4 * EPC;                  // one 64-byte row per K tile
    constexpr int NWAVES = WM_ * WN_;
    constexpr int WMR = BM_T / WM_;              // rows of the output tile owned by one wavefront
    constexpr int WN = BN_T / WN_;
    constexpr int NT_ = WN / 16;
    constexpr int MT_ = WMR / 16;
    constexpr int AROWS = BM_T / 16 / NWAVES;    // A-tile DMA instructions per thread
    constexpr int WROWS = BN_T / 16 / NWAVES;    // weight-tile DMA instructions per thread
    constexpr int LPT = AROWS + WROWS;           // loads per thread per K tile
    constexpr int NSTAGE = nt_stages(BM_T, BN_T);
    constexpr int A_BYTES = BM_T * 64;
    constexpr int W_BYTES = BN_T * 64;
    constexpr int STAGE = A_BYTES + W_BYTES;
    constexpr int MBOX = NSTAGE * STAGE;         // one dword behind the ring: the tile after the one being streamed
    constexpr uint32_t OOB = 0xfffffff0u;        // 16-byte aligned, beyond any operand (host checks sizes)
    static_assert(AROWS >= 1 && WROWS >= 1, "every wavefront issues at least one DMA row block per operand");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform (LDS-DMA base)
    const int wm = wave % WM_;
    const int wn = wave / WM_;

    // ---- data-gradient with stride s > 1: the input pixels split into s*s parity classes
    // (h % s, w % s); a class only ever meets the taps r == (h + pad) mod s, so each class is a
    // dense GEMM over its own tap subset (no multiply-by-zero work).  blockIdx.y = class.
    // In class-local terms the source pixel of (row, tap) is (A0 - tr, B0 - ts).
    int cs = 1, ph = 0, pw = 0, r0 = 0, s0 = 0, q0h = p.pad, q0w = p.pad;
    int Sc = p.S, Hc = p.OH, Wc = p.OW, Mc = p.M, Kc = p.Kd;
    int nblk = p.nblk;
    if (MODE == 1 && p.stride > 1) {
        cs = p.stride;
        ph = blockIdx.y / cs;
        pw = blockIdx.y - ph * cs;
        Hc = (p.OH - ph + cs - 1) / cs;
        Wc = (p.OW - pw + cs - 1) / cs;
        Mc = (p.M / (p.OH * p.OW)) * Hc * Wc;
        r0 = (ph + p.pad) % cs;
        s0 = (pw + p.pad) % cs;
        q0h = (ph + p.pad - r0) / cs;
        q0w = (pw + p.pad - s0) / cs;
        const int Rc = r0 < p.R ? (p.R - r0 + cs - 1) / cs : 0;
        Sc = s0 < p.S ? (p.S - s0 + cs - 1) / cs : 0;
        Kc = Rc * Sc * p.C;
        nblk = ((Mc + BM_T - 1) / BM_T) * p.tiles_n;
    }
    const int ohw = Hc * Wc;
    const int nkt = (Kc + BK - 1) / BK;          // 0 for a class without taps: the output is zero

    // ---- which tiles: block b runs on XCD b % 8 (observed dispatch order; speed only) and XCD x owns the contiguous tile
    // range [xbase, xbase + xcount): its workgroups take the first gridDim.x / 8 of them by position, the rest by ticket
    constexpr bool CAN_PERSIST = nt_can_persist((int)sizeof(T), OUT_F32, BM_T, BN_T, NWAVES);
    const bool persistent = CAN_PERSIST && p.tickets != nullptr;      // host: only with more tiles than workgroups and nkt > NSTAGE
    const int xcd = blockIdx.x & 7;
    const int xq = nblk >> 3, xr = nblk & 7;
    const int xbase = (xcd < xr) ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq;
    const int xcount = xq + (xcd < xr ? 1 : 0);
    const int xwg = (int)(gridDim.x >> 3) + (((int)(gridDim.x & 7) > xcd) ? 1 : 0);     // workgroups of this launch on this XCD
    unsigned int* const ticket = persistent ? p.tickets + blockIdx.y * 8 + xcd : nullptr;
    const int pos0 = (int)(blockIdx.x >> 3);
    const int first = pos0 < xcount ? xbase + pos0 : -1;

    const __amdgpu_buffer_rsrc_t src_rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src), 0, p.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wgt_rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wgt), 0, p.wgt_bytes, 0x00020000);

    // ---- per-thread DMA state of the tile being streamed.  Wave w, instruction i, lane l fills LDS bytes
    // [(i*NWAVES + w)*1024 + l*16, +16) of the A region: row (i*NWAVES+w)*16 + (l>>2), slot l&3, i.e. the
    // logical chunk (l&3) ^ f((row>>2)&3) = (l&3) ^ f((l>>4)&3) -- one K-chunk column per thread.
    const int cc = (lane & 3) ^ lds_swz((lane >> 4) & 3);
    int rowc[AROWS], a0[AROWS], b0[AROWS];   // rowc = (image base + A0*W + B0) * C  [elements]
    int wrow[WROWS];                         // weight row base [elements], or -1
    int kpos = 0, tc0 = 0, ttr = 0, tts = 0; // K-chunk walker: k = kt*BK + cc*EPC  ->  (tr, ts, c0), advanced by BK per tile
#pragma unroll
    for (int i = 0; i < AROWS; ++i) { rowc[i] = 0; a0[i] = 0; b0[i] = 0; }      // (defined on every path: the arrays stay in registers)
#pragma unroll
    for (int j = 0; j < WROWS; ++j) wrow[j] = -1;
    auto stream_tile = [&](int bid) __attribute__((always_inline)) {        // point the DMA state at the first K step of tile `bid`
        const int tile_n = bid % p.tiles_n;
        const int tile_m = bid / p.tiles_n;
#pragma unroll
        for (int i = 0; i < AROWS; ++i) {
            const int mq = tile_m * BM_T + (i * NWAVES + wave) * 16 + (lane >> 2);
            const bool in = mq < Mc;
            const int m = in ? mq : 0;          // branch-free: rows past the end decompose row 0 and are then marked out of range
            int img, oh, ow;
            if (cs == 1) {
                img = (int)fdiv((uint32_t)m, p.fd_ohw);
                const int rem = m - img * ohw;
                oh = (int)fdiv((uint32_t)rem, p.fd_ow);
                ow = rem - oh * Wc;
            } else {
                img = m / ohw;
                const int rem = m - img * ohw;
                oh = rem / Wc;
                ow = rem - oh * Wc;
            }
            const int av = MODE == 0 ? oh * p.stride - p.pad : oh + q0h;
            const int bv = MODE == 0 ? ow * p.stride - p.pad : ow + q0w;
            a0[i] = in ? av : -(1 << 24);
            b0[i] = in ? bv : -(1 << 24);
            rowc[i] = in ? ((img * p.H + av) * p.W + bv) * p.C : 0;
        }
#pragma unroll
        for (int j = 0; j < WROWS; ++j) {
            const int n = tile_n * BN_T + (j * NWAVES + wave) * 16 + (lane >> 2);
            wrow[j] = n < p.Nn ? n * p.Kd : -1;
        }
        kpos = cc * EPC;
        const int tap = kpos / p.C;
        tc0 = kpos - tap * p.C;
        ttr = Sc > 0 ? tap / Sc : 0;
        tts = tap - ttr * Sc;
    };

    typedef __attribute__((address_space(3))) void lds_void;
    // issue the DMA of the next K tile (walker position) into ring slot `stage`
    auto issue_tile = [&](int stage) __attribute__((always_inline)) {
        char* base = smem + stage * STAGE + wave * 1024;
        const bool kvalid = kpos < Kc;
        int tapoff, kw;
        if (PLAIN) {
            tapoff = kpos;
            kw = kpos;
        } else if (MODE == 0) {
            tapoff = (ttr * p.W + tts) * p.C + tc0;
            kw = kpos;
        } else {
            tapoff = tc0 - (ttr * p.W + tts) * p.C;
            kw = ((r0 + ttr * cs) * p.S + (s0 + tts * cs)) * p.C + tc0;
        }
#pragma unroll
        for (int i = 0; i < AROWS; ++i) {
            const int ih = PLAIN ? a0[i] : (MODE == 0) ? a0[i] + ttr : a0[i] - ttr;
            const int iw = PLAIN ? b0[i] : (MODE == 0) ? b0[i] + tts : b0[i] - tts;
            // bitwise (not short-circuit) logic keeps this branch-free
            const bool ok = PLAIN ? (kvalid & (a0[i] >= 0))
                                  : (kvalid & ((unsigned)ih < (unsigned)p.H) & ((unsigned)iw < (unsigned)p.W));
            const uint32_t off = ok ? (uint32_t)(rowc[i] + tapoff) * (uint32_t)sizeof(T) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(src_rs, (lds_void*)(base + i * NWAVES * 1024), 16, (int)off, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < WROWS; ++j) {
            const bool ok = kvalid & (wrow[j] >= 0);
            const uint32_t off = ok ? (uint32_t)(wrow[j] + kw) * (uint32_t)sizeof(T) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wgt_rs, (lds_void*)(base + A_BYTES + j * NWAVES * 1024), 16, (int)off, 0, 0, 0);
        }
        // advance to the next K tile
        kpos += BK;
        if (PLAIN) return;
        tc0 += BK;
        if (p.C >= BK) {                 // at most one tap boundary per tile (uniform branch)
            const bool wrap = tc0 >= p.C;
            tc0 -= wrap ? p.C : 0;
            tts += wrap ? 1 : 0;
            const bool wrap2 = tts == Sc;
            tts = wrap2 ? 0 : tts;
            ttr += wrap2 ? 1 : 0;
        } else {
            while (tc0 >= p.C) {
                tc0 -= p.C;
                if (++tts == Sc) { tts = 0; ++ttr; }
            }
        }
    };

    f32x4 acc[NT_][MT_];
#pragma unroll
    for (int ni = 0; ni < NT_; ++ni)
#pragma unroll
        for (int mi = 0; mi < MT_; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int l15 = lane & 15;
    const int lg = lane >> 4;
    int fa[MT_], fw[NT_];               // LDS fragment offsets within a stage, hoisted out of the K loop (not out of the tile
                                        // loop: they are recomputed per tile so that they are not live across the epilogue)
    auto fragment_offsets = [&]() __attribute__((always_inline)) {
        int l15o = l15, lgo = lg;
        asm volatile("" : "+v"(l15o), "+v"(lgo));       // opaque: keeps the compiler from hoisting the offsets over the epilogue
#pragma unroll
        for (int mi = 0; mi < MT_; ++mi) fa[mi] = lds_off(wm * WMR + mi * 16 + l15o, lgo);
#pragma unroll
        for (int ni = 0; ni < NT_; ++ni) fw[ni] = A_BYTES + lds_off(wn * WN + ni * 16 + l15o, lgo);
    };

    auto compute = [&](int stage) __attribute__((always_inline)) {
        const char* base = smem + stage * STAGE;
        u32x4 af[MT_], wf[NT_];
#pragma unroll
        for (int mi = 0; mi < MT_; ++mi) af[mi] = ld_chunk(base + fa[mi]);
#pragma unroll
        for (int ni = 0; ni < NT_; ++ni) wf[ni] = ld_chunk(base + fw[ni]);
#pragma unroll
        for (int ni = 0; ni < NT_; ++ni)
#pragma unroll
            for (int mi = 0; mi < MT_; ++mi) Mma<T>::run(acc[ni][mi], wf[ni], af[mi]);
    };

    // ---- the operand stream.  bid_i / ki: tile and K step the DMA side is at; the mailbox holds the tile after bid_i
    // (written by wavefront 0 two K steps after it drew the ticket, read by everyone at the switch, >= 1 barrier later).
    int bid_c = first;                 // tile being computed
    int bid_i = first, ki = 0;
    int issued = 0, consumed = 0;      // K steps issued / retired by this workgroup, over all its tiles
    int st_i = 0, st_c = 0;            // ring slots of the next DMA / the step being consumed
    unsigned int tk = 0u;              // wavefront 0, lane 0: the ticket in flight
    bool mb_pending = false;
    int mb_wait = 0;
    // The ticket is drawn ASYNCHRONOUSLY: a returning atomic hidden from the compiler (atomicAdd() is followed by an
    // immediate s_waitcnt vmcnt(0) -- the whole DMA ring plus an L2 round trip, once per tile), whose result lands in `tk`
    // some time later.  It is read two K steps on, behind the counted vmcnt wait of that step: vector-memory operations
    // return in issue order and that wait leaves at most 2 x LPT of them outstanding, all issued after the atomic.
    // The compiler does not know `tk` is in flight: tests/test_kernel_asm.py checks in the generated code of every
    // instantiation that nothing reads or copies the destination register before the mailbox write.
    const uint32_t mbox_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(smem + MBOX);
    auto draw_ticket = [&]() __attribute__((always_inline)) {
        if constexpr (CAN_PERSIST) {
            if (wave == 0 && lane == 0) {
                const unsigned int one = 1u;
                asm volatile("global_atomic_add %0, %1, %2, off sc0 ; saicv ticket" : "=v"(tk) : "v"(ticket), "v"(one) : "memory");
            }
            mb_pending = true;
        }
    };
    auto issue_next = [&]() __attribute__((always_inline)) {
        if (bid_i < 0) return;
        if (ki == nkt) {               // the stream moves on to the next tile of this workgroup
            int nb = -1;
            if (CAN_PERSIST && persistent) {
                int v;              // LDS access spelled out: a volatile generic access becomes a FLAT load behind vmcnt(0)
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(mbox_addr) : "memory");
                nb = __builtin_amdgcn_readfirstlane(v);
            }
            bid_i = nb;
            ki = 0;
            if (bid_i < 0) return;
            stream_tile(bid_i);
            draw_ticket();
            mb_wait = 1;
        }
        issue_tile(st_i);
        ++ki;
        ++issued;
        st_i = (st_i + 1 == NSTAGE) ? 0 : st_i + 1;
    };
    // ---- staggered start.  Tiles of one launch take the same time, so workgroups that start together reach their epilogues
    // together: 256 x 128 KiB of output stores hit HBM at once (33 MB, ~7 us at 4.7 TB/s) while every matrix core idles --
    // and gfx9 returns vector-memory operations in issue order, so a wavefront cannot confirm the operand loads it issued
    // BEHIND its stores before those stores are acknowledged (r03 measurement: a ViT-B K = 768 tile computes in ~22 us and
    // then waits ~7 us).  A persistent workgroup keeps the phase it starts with; spreading the starts over a tile period in
    // a few groups turns the bursts into a steady write stream that the next tile's first K steps cover.
    if (persistent && p.stagger_phases > 1) {
        const int slot = (int)(blockIdx.x >> 3) + (int)(blockIdx.x >> 8) * (p.stagger_phases >> 1);   // b and b + 256 share a CU
        const int naps = (slot % p.stagger_phases) * p.stagger_sleeps;
        for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(16);
    }
    if (bid_c >= 0) {
        stream_tile(bid_i);
        if (persistent) draw_ticket();
        if (nkt > 0)
            for (int s = 0; s < NSTAGE - 1; ++s) issue_next();
    }

    typedef typename std::conditional<OUT_F32, float, T>::type TO;
    // ---- epilogue geometry (per wavefront): a "piece" = 16 rows x WNS columns of the sub-tile, staged in LDS with a
    // padded pitch, read back as 16-byte row chunks: CPR chunks per row, CPL per lane, lane -> (row = j*RPJ + lane/CPR,
    // chunk = lane % CPR) -- a lane's column chunk is the same for every row it touches
    constexpr int SLOT_SHARE = STAGE / NWAVES;
    constexpr int NSPLIT = (16 * (WN * (int)sizeof(TO) + 16) <= SLOT_SHARE) ? 1 : 2;     // column halves per piece (fp32 outputs)
    constexpr int NTH = NT_ / NSPLIT;
    constexpr int WNS = WN / NSPLIT;
    constexpr int OPITCH = WNS * (int)sizeof(TO) + 16;          // bytes; +16 staggers banks
    static_assert(16 * OPITCH <= SLOT_SHARE && NT_ % NSPLIT == 0, "the staged piece must fit this wavefront's share of a ring slot");
    constexpr int OEPC = 16 / (int)sizeof(TO);
    constexpr int CPR = WNS / OEPC;
    constexpr int CPL = (16 * CPR) / 64;
    constexpr int RPJ = 64 / CPR;                               // rows covered by one chunk pass of the wavefront
    static_assert(16 * CPR >= 64 && CPL * 64 == 16 * CPR, "a staged piece is a whole number of chunk passes of the wavefront");
    constexpr bool MSTAT = !OUT_F32 && sizeof(T) == 2 && MODE == 0;      // BN statistics from the staged bf16 rows (matrix cores)
    constexpr bool VSTAT = MODE == 0 && !MSTAT;                          // ... from the accumulators (fp32 outputs)
    constexpr bool DGRAD_EXTRAS = MODE == 1 && sizeof(TO) == sizeof(T);  // gated shortcut / BatchNorm-backward sums: data gradient only
    TO* const outp = reinterpret_cast<TO*>(p.out);
    const bool aligned = ((p.ldo * (int)sizeof(TO)) & 15) == 0;
    const bool remap = MODE == 1 && cs > 1;
    const bool do_stats = MODE == 0 && p.stat_sum != nullptr;
    const bool bstats = DGRAD_EXTRAS && p.bs_y != nullptr;
    const bool addp = p.addend != nullptr, scalep = p.row_scale != nullptr;
    const bool gatep = DGRAD_EXTRAS && p.addend_gate != nullptr, maskp = bstats && p.bs_mask != nullptr;
    const bool fused = p.act_mode != 0 || addp || scalep || bstats;
    const bool fast = aligned && !remap && !fused;             // plain copy-out of whole chunks (N tail checked per chunk)

    auto epilogue = [&](int bid, char* stg) __attribute__((always_inline)) {
        // lane-derived indices of the epilogue are recomputed per tile from an opaque copy of the lane id: hoisted out of the
        // tile loop they would sit in registers all through the main loop (the 256-wide geometries run at the 256-register cap)
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const int l15 = lane_o & 15;
        const int lg = lane_o >> 4;
        const int crow = lane_o / CPR;                                  // row of this lane's chunk within a pass
        const int ccol = lane_o % CPR;                                  // its chunk column
        const int tile_n = bid % p.tiles_n;
        const int tile_m = bid / p.tiles_n;
        const int m_base = tile_m * BM_T + wm * WMR;
        const int n_base = tile_n * BN_T + wn * WN;
        float ssum[NT_][4], ssq[NT_][4];                   // VSTAT: column sums from the accumulators
        float msum[NT_], msq[NT_];                          // MSTAT: this lane's column (sum: lanes lg == 0; squares: lg == l15 >> 2)
        float bag[NSPLIT][OEPC], bax[NSPLIT][OEPC];         // BatchNorm-backward sums of this lane's column chunk
#pragma unroll
        for (int ni = 0; ni < NT_; ++ni) {
            msum[ni] = 0.f; msq[ni] = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) { ssum[ni][r] = 0.f; ssq[ni][r] = 0.f; }
        }
#pragma unroll
        for (int h = 0; h < NSPLIT; ++h)
#pragma unroll
            for (int e = 0; e < OEPC; ++e) { bag[h][e] = 0.f; bax[h][e] = 0.f; }
        if (MODE == 0 && p.bias != nullptr) {      // (uniform) the bias joins the accumulators before they are staged: no bias
#pragma unroll                                     // registers live across the piece loop
            for (int ni = 0; ni < NT_; ++ni) {
                const int n = n_base + ni * 16 + lg * 4;
                const f32x4 bv = {n < p.Nn ? p.bias[n] : 0.f, n + 1 < p.Nn ? p.bias[n + 1] : 0.f,
                                  n + 2 < p.Nn ? p.bias[n + 2] : 0.f, n + 3 < p.Nn ? p.bias[n + 3] : 0.f};
#pragma unroll
                for (int mi = 0; mi < MT_; ++mi) acc[ni][mi] += bv;
            }
        }
        // accumulators of row block MI (static index) + bias -> this wavefront's staging rows, columns of half H
        auto stage_rows = [&](auto MI, auto H) __attribute__((always_inline)) {
            constexpr int mi = decltype(MI)::value, h = decltype(H)::value;
#pragma unroll
            for (int i = 0; i < NTH; ++i) {
                const int ni = h * NTH + i;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[ni][mi][r];
                if (VSTAT) {
                    if (do_stats) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float vr = OUT_F32 ? v[r] : round_through<T>(v[r]);
                            ssum[ni][r] += vr;
                            ssq[ni][r] = fmaf(vr, vr, ssq[ni][r]);
                        }
                    }
                }
                char* q = stg + l15 * OPITCH + (i * 16 + lg * 4) * (int)sizeof(TO);
                if (sizeof(TO) == 2) {
                    bf16x4 pk;
#pragma unroll
                    for (int r = 0; r < 4; ++r) pk[r] = (bf16_t)v[r];
                    *reinterpret_cast<bf16x4*>(q) = pk;
                } else {
                    *reinterpret_cast<f32x4*>(q) = f32x4{v[0], v[1], v[2], v[3]};
                }
            }
        };
        // The piece loop exists twice: FUSED = false is the plain copy-out (no global load anywhere in it, so the compiler has
        // no load destinations to protect with s_waitcnt vmcnt(0) -- which, with stores in flight, would be a store round
        // trip per piece: gfx9 returns loads and stores in issue order), FUSED = true carries every fused mode.
        auto pieces = [&](auto FUSED_T) __attribute__((always_inline)) {
            constexpr bool FUSED = decltype(FUSED_T)::value;
#pragma unroll 1
            for (int mi = 0; mi < MT_; ++mi) {
#pragma unroll
                for (int h = 0; h < NSPLIT; ++h) {
                    int mrow[CPL];
                    u32x4 av[CPL], yv[CPL];
                    unsigned gb[CPL], mb[CPL];
                    float sc[CPL];
                    const int ncol = n_base + h * WNS + ccol * OEPC;
                    const bool col_in = ncol < p.Nn;
                    const bool whole = aligned && ncol + OEPC <= p.Nn;
#pragma unroll
                    for (int j = 0; j < CPL; ++j) {
                        const int mr = m_base + mi * 16 + j * RPJ + crow;           // row in tile-order (class-local) numbering
                        int mo = (mr < Mc && col_in) ? mr : -1;
                        if (FUSED && remap && mo >= 0) {
                            const int img = mr / ohw;
                            const int rem = mr - img * ohw;
                            const int hc = rem / Wc;
                            mo = (img * p.OH + hc * cs + ph) * p.OW + (rem - hc * Wc) * cs + pw;
                        }
                        mrow[j] = mo;
                        av[j] = u32x4{0u, 0u, 0u, 0u};
                        yv[j] = u32x4{0u, 0u, 0u, 0u};
                        gb[j] = 0xffu;
                        mb[j] = 0xffu;
                        sc[j] = 1.f;
                        if constexpr (FUSED) {
                            // side operands of this piece's chunks: every global load in flight before the staging round trip
                            if (fused && mo >= 0) {
                                const size_t off = (size_t)mo * p.ldo + ncol;
                                if (scalep) sc[j] = p.row_scale[mo / p.rows_per_scale];
                                if (addp && whole) av[j] = ld_chunk(reinterpret_cast<const TO*>(p.addend) + off);
                                if (bstats) yv[j] = ld_chunk(reinterpret_cast<const TO*>(p.bs_y) + off);
                                if (gatep) gb[j] = p.addend_gate[off / OEPC];
                                if (maskp) mb[j] = p.bs_mask[off / OEPC];
                            }
                        }
                    }
                    // ---- accumulators -> LDS (the only part that needs static register indices: one small case per row block)
#define STAGE_CASE(K)                                                                                                       \
                    if constexpr (MT_ > K) {                                                                                \
                        if (mi == K) {                                                                                      \
                            if (h == 0) stage_rows(std::integral_constant<int, K>{}, std::integral_constant<int, 0>{});     \
                            if constexpr (NSPLIT == 2) {                                                                    \
                                if (h == 1) stage_rows(std::integral_constant<int, K>{}, std::integral_constant<int, 1>{}); \
                            }                                                                                               \
                        }                                                                                                   \
                    }
                    STAGE_CASE(0) STAGE_CASE(1) STAGE_CASE(2) STAGE_CASE(3) STAGE_CASE(4) STAGE_CASE(5) STAGE_CASE(6) STAGE_CASE(7)
#undef STAGE_CASE
                    // ---- BatchNorm statistics of the staged bf16 rows on the matrix cores
                    if constexpr (MSTAT) {
                        if (do_stats) {
                            typedef __attribute__((ext_vector_type(4))) short s16x4;
                            const s16x4 ones = {(short)0x3f80, (short)0x3f80, (short)0x3f80, (short)0x3f80};
                            // lane (column t = l15, row group lg) of a 16-lane group supplies the address of row lg*4 + (t>>2),
                            // columns (t&3)*4.. and receives rows lg*4 .. lg*4+3 of column t: the B fragment of a 16x16x16 MFMA.
                            // Spelled as asm: behind the BUILTIN transposed read the compiler waits vmcnt(0) (an LDS access it
                            // can see, while LDS-DMA writes are pending) -- with the previous piece's stores in flight that is a
                            // store round trip per piece.  LDS operations of one wavefront execute in issue order, so the staged
                            // rows written just above are what these reads return.
                            const uint32_t qa = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(
                                stg + (lg * 4 + (l15 >> 2)) * OPITCH + (l15 & 3) * 8);
                            s16x4 ys[NTH];
                            static_assert(NTH == 4 || NTH == 2, "column groups of a staged piece");
                            if constexpr (NTH == 4) {
                                asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %4 offset:32\n\t"
                                             "ds_read_b64_tr_b16 %2, %4 offset:64\n\tds_read_b64_tr_b16 %3, %4 offset:96\n\t"
                                             "s_waitcnt lgkmcnt(0)"
                                             : "=&v"(ys[0]), "=&v"(ys[1]), "=&v"(ys[2]), "=&v"(ys[3]) : "v"(qa) : "memory");
                            } else {
                                asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:32\n\t"
                                             "s_waitcnt lgkmcnt(0)"
                                             : "=&v"(ys[0]), "=&v"(ys[1]) : "v"(qa) : "memory");
                            }
#pragma unroll
                            for (int i = 0; i < NTH; ++i) {
                                const s16x4 y = ys[i];
                                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                                const f32x4 s1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ones, y, z, 0, 0, 0);
                                const f32x4 s2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(y, y, z, 0, 0, 0);
                                // the diagonal element of this lane's column: row l15 = lg*4 + r lives in lanes lg == l15 >> 2.
                                // (opaque copies: a select chain on vector elements is otherwise turned into a dynamically
                                // indexed extract, which the backend lowers through scratch memory)
                                float e0 = s2[0], e1 = s2[1], e2 = s2[2], e3 = s2[3];
                                asm volatile("" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3));
                                const int r = l15 & 3;
                                msum[h * NTH + i] += s1[0];
                                msq[h * NTH + i] += (r & 2) ? ((r & 1) ? e3 : e2) : ((r & 1) ? e1 : e0);
                            }
                        }
                    }
                    // ---- staged rows -> 16-byte row chunks -> (fused modes) -> HBM
#pragma unroll
                    for (int j = 0; j < CPL; ++j) {
                        const int mo = mrow[j];
                        u32x4 v = ld_chunk(stg + (j * RPJ + crow) * OPITCH + ccol * 16);
                        if (mo < 0) continue;
                        TO* o = outp + (size_t)mo * p.ldo + ncol;
                        if constexpr (!FUSED) {
                            if (whole) {
                                st_chunk(o, v);
                            } else {
                                float e[OEPC];
                                Chunk<TO>::unpack(v, e);
#pragma unroll
                                for (int k = 0; k < OEPC; ++k)
                                    if (ncol + k < p.Nn) o[k] = from_f32<TO>(e[k]);
                            }
                        } else {
                            float f[OEPC];
                            Chunk<TO>::unpack(v, f);
                            if (p.act_mode == 1 || p.act_mode == 3) {   // fc1 of an MLP: emit gelu() beside the pre-activation (1) or gelu'() (3)
                                float gl[OEPC], gr[OEPC];
                                gelu_and_grad8(f, gl, gr);
                                st_chunk(reinterpret_cast<TO*>(p.out2) + (size_t)mo * p.ldo + ncol, Chunk<TO>::pack(gl));
                                if (p.act_mode == 3) v = Chunk<TO>::pack(gr);
                            } else if (p.act_mode == 2 || p.act_mode == 4) {     // dgrad of fc2: times gelu'(pre) (2) or times the stored derivative (4)
                                float a[OEPC];
                                Chunk<TO>::unpack(av[j], a);
                                if (p.act_mode == 2) {
#pragma unroll
                                    for (int k = 0; k < OEPC; ++k) f[k] *= gelu_grad_f(a[k]);
                                } else {
#pragma unroll
                                    for (int k = 0; k < OEPC; ++k) f[k] *= a[k];
                                }
                                v = Chunk<TO>::pack(f);
                            } else if (addp || scalep) {
                                float a[OEPC];
                                if (whole) {
                                    Chunk<TO>::unpack(av[j], a);
                                } else {
#pragma unroll
                                    for (int k = 0; k < OEPC; ++k)
                                        a[k] = (addp && ncol + k < p.Nn)
                                                   ? to_f32(reinterpret_cast<const TO*>(p.addend)[(size_t)mo * p.ldo + ncol + k]) : 0.f;
                                }
#pragma unroll
                                for (int k = 0; k < OEPC; ++k) f[k] = fmaf(sc[j], f[k], ((gb[j] >> k) & 1u) ? a[k] : 0.f);
                                v = Chunk<TO>::pack(f);
                            }
                            if (bstats) {                     // BatchNorm-backward sums of what is stored (rounded to TO), behind its ReLU gate
                                float yy[OEPC];
                                Chunk<TO>::unpack(v, f);
                                Chunk<TO>::unpack(yv[j], yy);
#pragma unroll
                                for (int k = 0; k < OEPC; ++k) {
                                    const float ge = ((mb[j] >> k) & 1u) ? f[k] : 0.f;
                                    bag[h][k] += ge;
                                    bax[h][k] = fmaf(ge, yy[k], bax[h][k]);
                                }
                            }
                            if (whole) {
                                st_chunk(o, v);
                            } else {                          // N tail, or a leading dimension without 16-byte alignment
                                float e[OEPC];
                                Chunk<TO>::unpack(v, e);
#pragma unroll
                                for (int k = 0; k < OEPC; ++k)
                                    if (ncol + k < p.Nn) o[k] = from_f32<TO>(e[k]);
                            }
                        }
                    }
                }
            }
        };
        if constexpr (FUSEDK) pieces(std::true_type{});
        else pieces(std::false_type{});           // host: launched only when `fast` holds
        // ---- statistics of this wavefront's rows: one partial row per (tile row, wavefront row), or atomics into a few rows
        if (do_stats) {
            const size_t srow = p.stat_atomic_rows ? (size_t)((tile_m * WM_ + wm) % p.stat_atomic_rows) : (size_t)(tile_m * WM_ + wm);
            if constexpr (MSTAT) {
#pragma unroll
                for (int ni = 0; ni < NT_; ++ni) {
                    const int n = n_base + ni * 16 + l15;              // D: row lg*4 + r, column l15
                    if (n < p.Nn) {
                        if (lg == 0) {
                            if (p.stat_atomic_rows) unsafeAtomicAdd(&p.stat_sum[srow * (size_t)p.Nn + n], msum[ni]);
                            else p.stat_sum[srow * (size_t)p.Nn + n] = msum[ni];
                        }
                        if (lg == (l15 >> 2)) {
                            if (p.stat_atomic_rows) unsafeAtomicAdd(&p.stat_sq[srow * (size_t)p.Nn + n], msq[ni]);
                            else p.stat_sq[srow * (size_t)p.Nn + n] = msq[ni];
                        }
                    }
                }
            }
            if constexpr (VSTAT) {
                // rows m >= M were gathered as zeros (no bias when stats are requested) -> add 0.
#pragma unroll
                for (int ni = 0; ni < NT_; ++ni)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {        // over the 16 pixel lanes: four DPP adds each
                        const float a = row16_sum(ssum[ni][r]), b = row16_sum(ssq[ni][r]);
                        const int n = n_base + ni * 16 + lg * 4 + r;
                        if (l15 == 0 && n < p.Nn) {
                            if (p.stat_atomic_rows) {
                                unsafeAtomicAdd(&p.stat_sum[srow * (size_t)p.Nn + n], a);
                                unsafeAtomicAdd(&p.stat_sq[srow * (size_t)p.Nn + n], b);
                            } else {
                                p.stat_sum[srow * (size_t)p.Nn + n] = a;
                                p.stat_sq[srow * (size_t)p.Nn + n] = b;
                            }
                        }
                    }
            }
        }
        if (FUSEDK && bstats) {
            // sum g * (y - mean) * invstd = invstd * (sum g y - mean * sum g): the mean leaves after this lane's few rows,
            // while the sums are still small -- not after the whole column.  Lanes sharing a chunk column differ in crow.
            const size_t prow0 = (size_t)((MODE == 1 && cs > 1) ? blockIdx.y : 0) * p.bs_rows + tile_m;
            const size_t prow = prow0 * WM_ + wm;
            const size_t drow = p.stat_atomic_rows ? prow % p.stat_atomic_rows : prow;
#pragma unroll
            for (int h = 0; h < NSPLIT; ++h) {
                const int ncol = n_base + h * WNS + ccol * OEPC;
                const bool have = ncol + OEPC <= p.Nn;
#pragma unroll
                for (int e = 0; e < OEPC; ++e) {
                    const float mu = have ? p.bs_mean[ncol + e] : 0.f, is = have ? p.bs_invstd[ncol + e] : 0.f;
                    float g = bag[h][e];
                    float gx = is * fmaf(-mu, bag[h][e], bax[h][e]);
#pragma unroll
                    for (int d = CPR; d < 64; d <<= 1) {
                        g += __shfl_xor(g, d, 64);
                        gx += __shfl_xor(gx, d, 64);
                    }
                    if (lane_o < CPR && have) {
                        float* dg = p.bs_g + drow * (size_t)p.Nn + ncol + e;
                        float* dx = p.bs_gx + drow * (size_t)p.Nn + ncol + e;
                        if (p.stat_atomic_rows) { unsafeAtomicAdd(dg, g); unsafeAtomicAdd(dx, gx); }
                        else { *dg = g; *dx = gx; }
                    }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this wavefront's staging traffic is over: the slot may be refilled
    };

    // Stores a plain epilogue of a full tile issues per wavefront (one per chunk pass, every pass taken): known exactly, so
    // the first NSTAGE-1 K steps behind the seam -- whose DMAs were issued BEFORE those stores -- wait with the stores
    // allowed to stay in flight.  (vmcnt is a 6-bit count.)
    constexpr int S_PLAIN = MT_ * NSPLIT * CPL;
    static_assert(2 * LPT + S_PLAIN <= 63, "vmcnt immediate");
    int seam_steps = 0;                // first K steps of this tile whose DMA precedes S_PLAIN in-flight stores of the last epilogue
    while (bid_c >= 0) {
        fragment_offsets();
#pragma clang loop unroll(disable)
        for (int kc = 0; kc < nkt; ++kc) {
            // retire step `consumed` only: the steps issued after it stay in flight across the barrier
            const int ahead = issued - consumed - 1;      // wave-uniform
            // (the choice between the two counts lives INSIDE one asm statement: as separate statements behind a C++ branch
            // the same logic costs the 256-wide geometries 12-50 spilled registers -- another block structure for the
            // scheduler -- and a spilled register is what the asynchronous ticket cannot afford)
            const int behind_seam = __builtin_amdgcn_readfirstlane(seam_steps > kc ? 1 : 0);
            if (ahead >= 2)
                asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_waitcnt vmcnt(%2)\n\ts_branch 2f\n"
                             "1:\n\ts_waitcnt vmcnt(%1)\n2:" ::"s"(behind_seam), "n"(2 * LPT), "n"(2 * LPT + S_PLAIN) : "memory", "scc");
            else if (ahead == 1)
                asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_waitcnt vmcnt(%2)\n\ts_branch 2f\n"
                             "1:\n\ts_waitcnt vmcnt(%1)\n2:" ::"s"(behind_seam), "n"(LPT), "n"(LPT + S_PLAIN) : "memory", "scc");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();          // everyone's part of this step landed; the previous step is fully consumed
            if (CAN_PERSIST && mb_pending) {       // the ticket drawn two steps ago names the tile after bid_i
                if (mb_wait == 0) {
                    if (wave == 0 && lane == 0) {
                        const int pos = xwg + (int)tk;
                        const int nb = pos < xcount ? xbase + pos : -1;
                        asm volatile("ds_write_b32 %0, %1 ; saicv mailbox" :: "v"(mbox_addr), "v"(nb) : "memory");
                    }
                    mb_pending = false;
                } else {
                    --mb_wait;
                }
            }
            issue_next();                          // refill the slot the previous step just vacated
            compute(st_c);
            st_c = (st_c + 1 == NSTAGE) ? 0 : st_c + 1;
            ++consumed;
        }
        // ---- tile seam: the slot of the last K step is free once every wavefront has read its fragments; it is not
        // refilled before the barrier of the NEXT tile's first step, which every wavefront reaches after its epilogue
        const int st_last = st_c == 0 ? NSTAGE - 1 : st_c - 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        {
            // exact store count: plain path, no statistics traffic, the tile entirely inside the output (every lane of
            // every chunk pass stores, so every store instruction is issued)
            const int tm = bid_c / p.tiles_n, tn = bid_c - tm * p.tiles_n;
            // (bias loads of the epilogue are waited for by the compiler before the stores are issued: more operations behind
            // the DMAs than counted, never fewer)
            const bool exact = !FUSEDK && !do_stats && (tm + 1) * BM_T <= Mc && (tn + 1) * BN_T <= p.Nn;
            int ex = __builtin_amdgcn_readfirstlane(exact ? NSTAGE - 1 : 0);
            asm volatile("" : "+s"(ex));       // opaque: keeps the optimiser from specialising the K loop per epilogue path
            epilogue(bid_c, smem + st_last * STAGE + wave * SLOT_SHARE);
            seam_steps = ex;
        }
#pragma unroll
        for (int ni = 0; ni < NT_; ++ni)
#pragma unroll
            for (int mi = 0; mi < MT_; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
        // the DMA side switched tiles NSTAGE-1 steps ago (host: nkt > NSTAGE in persistent launches), or the stream ended
        bid_c = persistent ? bid_i : -1;
    }
    // ---- persistent launches leave their ticket counters zeroed: the last workgroup to leave resets them
    if (persistent && wave == 0 && lane == 0) {
        unsigned int* const done = p.tickets + 32;
        const unsigned int total = gridDim.x * gridDim.y;
        if (atomicAdd(done, 1u) == total - 1u) {
            for (int i = 0; i < 33; ++i) __hip_atomic_store(p.tickets + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ------------------------------------------------------------------------------------ NT, one tile per workgroup
// The r02 kernel, kept for every launch that is NOT persistent (at most one round of resident workgroups, short K loops,
// stride > 1 data-gradient classes, fp32 parity mode, fp32 outputs): its workgroup-wide epilogue (whole tile staged in LDS,
// statistics over all rows of the tile, eight independent row chunks per thread in flight) is the faster one when nothing
// follows the tile in the same workgroup, and it writes ONE partial-statistics row per tile row.
// Main loop = LDS-DMA ring: `buffer_load ... lds` writes global chunks straight into a 4-stage
// LDS ring (no staging registers, no ds_write), three K tiles are in flight across the single
// raw s_barrier of each tile, and the wait is a COUNTED s_waitcnt vmcnt(N) that only retires the
// tile about to be consumed (guide section 5, T3/T4).  The DMA destination is wave-linear
// (base + lane*16), so the XOR swizzle is applied to the SOURCE: lane l fetches the chunk that
// belongs in the slot it will fill.
// The loop is bound by MFMA issue only if the integer work per K tile is tiny, so:
//  * all gathers are raw buffer loads: an out-of-range lane (padding halo, M/N/K tails) gets
//    an offset beyond the buffer and the hardware writes zeros -- no branches, no exec masking;
//  * the (r, s, c) position of a thread's K chunk advances incrementally (no divisions);
//  * per-row constants fold image base and top-left corner, so a gather address is one add;
//  * LDS fragment addresses (with the XOR swizzle) are computed once.
// PLAIN: 1x1 taps without padding (pointwise conv, nn.Linear and their data-gradients): the K
// index IS the channel offset, no tap walker and no halo tests in the loop.
// Tile geometry (BM_T x BN_T output tile, WM_ x WN_ wavefronts, each owning a
// (BM_T/WM_) x (BN_T/WN_) sub-tile): 256x256 / 2x4, 256x128 / 4x2, 128x128 / 2x2, 128x64 / 2x2.
// Wider tiles raise flop per LDS-fill byte and the MFMAs issued per barrier and per DMA.
// KC = 16-byte chunks per LDS row = width of the K slice a step loads: 4 (64 bytes, 32 bf16) or 8 (128 bytes).  With 64-byte
// slices every DMA request is half a cache line; scripts/probes/fill_probe.hip measures what that costs on the fill path
// alone (profiles/r03_lds_fill_probe.jsonl: 8.7-10 TB/s against 11.5-12.3 TB/s for whole lines when four workgroups share
// a stream) and the 64-byte kernels sit exactly at that figure.  KC = 8 rows take twice the LDS per step: 3 slots of
// 48 KiB for the 256 x 128 tile, one workgroup per CU.
template <typename T, int BM_T, int BN_T, int WM_, int WN_, int MODE, bool OUT_F32, bool PLAIN, int KC = 4>
__global__ __launch_bounds__(64 * WM_ * WN_, (BM_T == 256 && BN_T == 128 && !OUT_F32 && KC == 4) ? 4 : 1) void igemm_nt1_kernel(const NTParams p) {
    constexpr int EPC = ElemTraits<T>::EPC;
    constexpr int BK = KC * EPC;                 // one 64- or 128-byte row per K tile
    constexpr int ROWB = KC * 16;                // bytes of an LDS row
    constexpr int RPI = 1024 / ROWB;             // tile rows filled by one DMA wave-instruction
    constexpr int NWAVES = WM_ * WN_;
    constexpr int NTHREADS = 64 * NWAVES;
    constexpr int WMR = BM_T / WM_;              // rows of the output tile owned by one wavefront
    constexpr int WN = BN_T / WN_;
    typedef NtMma<T> MM;
    constexpr int FR = MM::FR;                   // rows of a fragment (both operands): 32 (bf16, 32x32x16) or 16 (fp32, 16x16x4)
    constexpr int NG = MM::NV / 4;               // groups of four consecutive output columns per lane and accumulator tile
    constexpr int KSUB = MM::KSUB;
    constexpr int NT_ = WN / FR;
    constexpr int MT_ = WMR / FR;
    static_assert(WN % FR == 0 && WMR % FR == 0, "wavefront sub-tile in whole MFMA tiles");
    constexpr int AROWS = BM_T / RPI / NWAVES;   // A-tile DMA instructions per thread
    constexpr int WROWS = BN_T / RPI / NWAVES;   // weight-tile DMA instructions per thread
    constexpr int LPT = AROWS + WROWS;           // loads per thread per K tile
    constexpr int NSTAGE = KC == 8 ? nt_stages_kc8(BM_T, BN_T) : nt_stages(BM_T, BN_T);
    constexpr int A_BYTES = BM_T * ROWB;
    constexpr int W_BYTES = BN_T * ROWB;
    static_assert(KC == 4 || (KC == 8 && NWAVES % 2 == 0 && sizeof(T) == 2), "128-byte K slices: bf16, even wavefront count");
    constexpr int STAGE = A_BYTES + W_BYTES;
    constexpr uint32_t OOB = 0xfffffff0u;   // 16-byte aligned, beyond any operand (host checks sizes)

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform (LDS-DMA base)
    const int wm = wave % WM_;
    const int wn = wave / WM_;
#ifdef SAICV_NT_TIMELINE
    const bool tl_on = p.timeline != nullptr;
    unsigned long long tl_t[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};       // 8..10: inside the set-up (tile known | rows decomposed | walker ready)
    const unsigned long long tl_rt0 = tl_on ? __builtin_amdgcn_s_memrealtime() : 0ull;      // 100 MHz, the same on every XCD
    NT_STAMP(0);
#endif

    // ---- data-gradient with stride s > 1: the input pixels split into s*s parity classes
    // (h % s, w % s); a class only ever meets the taps r == (h + pad) mod s, so each class is a
    // dense GEMM over its own tap subset (no multiply-by-zero work).  blockIdx.y = class.
    // In class-local terms the source pixel of (row, tap) is (A0 - tr, B0 - ts).
    int cs = 1, ph = 0, pw = 0, r0 = 0, s0 = 0, q0h = p.pad, q0w = p.pad;
    int Sc = p.S, Hc = p.OH, Wc = p.OW, Mc = p.M, Kc = p.Kd;
    int nblk = p.nblk;
    if (MODE == 1 && p.stride > 1) {
        cs = p.stride;
        ph = blockIdx.y / cs;
        pw = blockIdx.y - ph * cs;
        Hc = (p.OH - ph + cs - 1) / cs;
        Wc = (p.OW - pw + cs - 1) / cs;
        Mc = (p.M / (p.OH * p.OW)) * Hc * Wc;
        r0 = (ph + p.pad) % cs;
        s0 = (pw + p.pad) % cs;
        q0h = (ph + p.pad - r0) / cs;
        q0w = (pw + p.pad - s0) / cs;
        const int Rc = r0 < p.R ? (p.R - r0 + cs - 1) / cs : 0;
        Sc = s0 < p.S ? (p.S - s0 + cs - 1) / cs : 0;
        Kc = Rc * Sc * p.C;
        nblk = ((Mc + BM_T - 1) / BM_T) * p.tiles_n;
    }
    // ---- tile loop: one tile per workgroup by default (the hardware dispatcher balances the load); with
    // SAICV_NT_PERSIST=1 the grid is one resident round of workgroups, each walking tiles tix, tix + grid, ...
    // Spreading the workgroups' start times over a tile period -- so that epilogue write bursts and MFMA loops of
    // different CUs interleave -- was measured and bought nothing: the ramp costs what the steady state gains.
    for (int tix = blockIdx.x; tix < nblk; tix += gridDim.x) {
    if (tix != (int)blockIdx.x) {                  // the previous tile's LDS reads are done (its stores may still drain)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    const int bid = xcd_remap(tix, nblk);
    const int tile_n = bid % p.tiles_n;
    const int tile_m = bid / p.tiles_n;
#ifdef SAICV_NT_TIMELINE
    if (tl_on) { int t_ = tile_m; asm volatile("" : "+s"(t_)); NT_STAMP(8); }
#endif

    const __amdgpu_buffer_rsrc_t src_rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src), 0, p.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wgt_rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wgt), 0, p.wgt_bytes, 0x00020000);

    typename MM::Acc acc[NT_][MT_];
#pragma unroll
    for (int ni = 0; ni < NT_; ++ni)
#pragma unroll
        for (int mi = 0; mi < MT_; ++mi)
#pragma unroll
            for (int r = 0; r < MM::NV; ++r) acc[ni][mi][r] = 0.f;

    const int ohw = Hc * Wc;
    const int lr = lane & (FR - 1);              // fragment row (first operand) / column (second operand) this lane addresses
    const int lk = lane / FR;                    // its 16-byte K chunk: 0..3 of the 64-byte row (fp32), 0..1 of each 32-byte half (bf16)
    // accumulator value r of tile (ni, mi): row m_base + mi*FR + lr, column n_base + ni*FR + fcol(r / 4) + r % 4
    auto fcol = [&](int g) { return FR == 32 ? 8 * g + 4 * lk : 4 * lk; };
    // r05 -- the bias starts the accumulators (see the prologue of the K loop): requested here, columns past N read zeros (buffer
    // bounds); N % 4 != 0 keeps the epilogue path
    const bool bias_early = p.bias != nullptr && (p.Nn & 3) == 0;      // uniform
    u32x4 bias_v[NT_][NG];
#pragma unroll
    for (int ni = 0; ni < NT_; ++ni)
#pragma unroll
        for (int g = 0; g < NG; ++g) bias_v[ni][g] = u32x4{0u, 0u, 0u, 0u};
    if (bias_early) {
        const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.Nn * 4, 0x00020000);
#pragma unroll
        for (int ni = 0; ni < NT_; ++ni)
#pragma unroll
            for (int g = 0; g < NG; ++g)
                bias_v[ni][g] = buf_ld(brs, (uint32_t)(tile_n * BN_T + wn * WN + ni * FR + fcol(g)) * 4u);
    }
    {
    // ---- per-thread DMA state.  Wave w, instruction i, lane l fills LDS bytes
    // [(i*NWAVES + w)*1024 + l*16, +16) of the A region: row (i*NWAVES+w)*16 + (l>>2), slot l&3, i.e. the
    // logical chunk (l&3) ^ f((row>>2)&3) = (l&3) ^ f((l>>4)&3) -- one K-chunk column per thread.
    // KC = 8: row (i*NWAVES + wave)*8 + (lane>>3), slot lane&7, logical chunk slot ^ ((row >> 1) & 7); (row >> 1) & 7 =
    // ((lane >> 4) + 4 * (wave & 1)) & 7 because NWAVES is even -- still one K-chunk column per thread
    const int cc = KC == 4 ? ((lane & 3) ^ lds_swz((lane >> 4) & 3)) : ((lane & 7) ^ (((lane >> 4) + 4 * (wave & 1)) & 7));
    int rowc[AROWS], a0[AROWS], b0[AROWS];   // rowc = (image base + A0*W + B0) * C  [elements]
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
        const int m = tile_m * BM_T + (i * NWAVES + wave) * RPI + lane / KC;
        if (m < Mc) {
            int img, oh, ow;
            if (cs == 1) {
                img = (int)fdiv((uint32_t)m, p.fd_ohw);
                const int rem = m - img * ohw;
                oh = (int)fdiv((uint32_t)rem, p.fd_ow);
                ow = rem - oh * Wc;
            } else {
                img = m / ohw;
                const int rem = m - img * ohw;
                oh = rem / Wc;
                ow = rem - oh * Wc;
            }
            if (MODE == 0) {
                a0[i] = oh * p.stride - p.pad;
                b0[i] = ow * p.stride - p.pad;
            } else {
                a0[i] = oh + q0h;
                b0[i] = ow + q0w;
            }
            rowc[i] = ((img * p.H + a0[i]) * p.W + b0[i]) * p.C;
        } else {
            rowc[i] = 0;
            a0[i] = -(1 << 24);
            b0[i] = -(1 << 24);
        }
    }
#ifdef SAICV_NT_TIMELINE
    if (tl_on) { int t_ = rowc[0]; asm volatile("" : "+v"(t_)); NT_STAMP(9); }
#endif
    int wrow[WROWS];                    // weight row base [elements], or -1
#pragma unroll
    for (int j = 0; j < WROWS; ++j) {
        const int n = tile_n * BN_T + (j * NWAVES + wave) * RPI + lane / KC;
        wrow[j] = n < p.Nn ? n * p.Kd : -1;
    }

    // K-chunk walker: k = kt*BK + cc*EPC  ->  (tr, ts, c0), advanced by BK per tile
    int kpos = cc * EPC;
    int tc0, ttr, tts;
    {
        const int tap = kpos / p.C;
        tc0 = kpos - tap * p.C;
        ttr = Sc > 0 ? tap / Sc : 0;
        tts = tap - ttr * Sc;
    }

    typedef __attribute__((address_space(3))) void lds_void;
    // issue the DMA of the next K tile (walker position) into ring slot `stage`
    auto issue_tile = [&](int stage) {
        char* base = smem + stage * STAGE + wave * 1024;
        const bool kvalid = kpos < Kc;
        int tapoff, kw;
        if (PLAIN) {
            tapoff = kpos;
            kw = kpos;
        } else if (MODE == 0) {
            tapoff = (ttr * p.W + tts) * p.C + tc0;
            kw = kpos;
        } else {
            tapoff = tc0 - (ttr * p.W + tts) * p.C;
            kw = ((r0 + ttr * cs) * p.S + (s0 + tts * cs)) * p.C + tc0;
        }
#pragma unroll
        for (int i = 0; i < AROWS; ++i) {
            const int ih = PLAIN ? a0[i] : (MODE == 0) ? a0[i] + ttr : a0[i] - ttr;
            const int iw = PLAIN ? b0[i] : (MODE == 0) ? b0[i] + tts : b0[i] - tts;
            // bitwise (not short-circuit) logic keeps this branch-free
            const bool ok = PLAIN ? (kvalid & (a0[i] >= 0))
                                  : (kvalid & ((unsigned)ih < (unsigned)p.H) & ((unsigned)iw < (unsigned)p.W));
            const uint32_t off = ok ? (uint32_t)(rowc[i] + tapoff) * (uint32_t)sizeof(T) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(src_rs, (lds_void*)(base + i * NWAVES * 1024), 16, (int)off, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < WROWS; ++j) {
            const bool ok = kvalid & (wrow[j] >= 0);
            const uint32_t off = ok ? (uint32_t)(wrow[j] + kw) * (uint32_t)sizeof(T) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wgt_rs, (lds_void*)(base + A_BYTES + j * NWAVES * 1024), 16, (int)off, 0, 0, 0);
        }
        // advance to the next K tile
        kpos += BK;
        if (PLAIN) return;
        tc0 += BK;
        if (p.C >= BK) {                 // at most one tap boundary per tile (uniform branch)
            const bool wrap = tc0 >= p.C;
            tc0 -= wrap ? p.C : 0;
            tts += wrap ? 1 : 0;
            const bool wrap2 = tts == Sc;
            tts = wrap2 ? 0 : tts;
            ttr += wrap2 ? 1 : 0;
        } else {
            while (tc0 >= p.C) {
                tc0 -= p.C;
                if (++tts == Sc) { tts = 0; ++ttr; }
            }
        }
    };

    // LDS fragment offsets within a stage (K sub-step 0 of the first 64-byte half).  Fragment mi / ni sits mi (ni) * FR rows below
    // fragment 0 at the same swizzle (FR rows change neither (row >> 2) & 3 nor (row >> 1) & 7): one register per operand + immediates.
    // bf16 sub-step s reads chunk 2 s + lk: an XOR of the offset with 32 s; the second 64-byte half of a 128-byte row (KC = 8): XOR 64.
    const int fa0 = KC == 4 ? lds_off(wm * WMR + lr, lk) : lds_off128(wm * WMR + lr, lk);
    const int fw0 = A_BYTES + (KC == 4 ? lds_off(wn * WN + lr, lk) : lds_off128(wn * WN + lr, lk));

    // bf16: fragment reads as assembly statements in a fixed order, MFMA groups released by counted waits (see lds_rd128 above).  Per
    // 64-byte half: sub-steps s = 0, 1 (K = 16 each), per sub-step W fragment 0, every A fragment, the other W fragments -- all
    // KSUB * (MT_ + NT_) reads of the half in flight before the first MFMA.  fp32 (parity mode) keeps the compiler's schedule: its
    // 16x16x4 MFMAs take 32 cycles each, an LDS round trip hides behind any group of them.
    constexpr bool ASM_FRAGS = sizeof(T) == 2;
    constexpr int RPH = KSUB * (MT_ + NT_);      // reads per 64-byte half
    // (16 reads: the counter holds 15, the sixteenth issues once the first -- they return in order -- has landed, which is what lgkmcnt(15) then asks for)
    static_assert(RPH <= 16, "lgkmcnt counts 15 operations");
    const uint32_t lds0 = lds_addr32(smem);
    auto frag_reads = [&](int stage, auto KS, u32x4 (&af)[KSUB][MT_], u32x4 (&wf)[KSUB][NT_]) __attribute__((always_inline)) {
        constexpr int ks = decltype(KS)::value;
        static_for<0, KSUB>([&](auto S) {
            constexpr int sb = decltype(S)::value;
            const uint32_t a_addr = lds0 + (uint32_t)(stage * STAGE) + (uint32_t)(fa0 ^ (ks * 64 + sb * 32));
            const uint32_t w_addr = lds0 + (uint32_t)(stage * STAGE) + (uint32_t)(fw0 ^ (ks * 64 + sb * 32));
            lds_rd128<0>(wf[sb][0], w_addr);
            static_for<0, MT_>([&](auto MI) { lds_rd128<decltype(MI)::value * FR * ROWB>(af[sb][decltype(MI)::value], a_addr); });
            static_for<1, NT_>([&](auto NI) { lds_rd128<decltype(NI)::value * FR * ROWB>(wf[sb][decltype(NI)::value], w_addr); });
        });
    };
    auto frag_mma = [&](u32x4 (&af)[KSUB][MT_], u32x4 (&wf)[KSUB][NT_]) __attribute__((always_inline)) {
        // read number P (issue order) has landed once lgkmcnt <= RPH - 1 - P.  The scheduling barriers keep each MFMA group above
        // the next wait: left alone, the MFMAs -- which touch no memory -- sink below the later assembly statements and the whole
        // step waits for its last read.
        static_for<0, KSUB>([&](auto S) {
            constexpr int sb = decltype(S)::value;
            constexpr int P0 = sb * (MT_ + NT_);
            static_for<0, MT_>([&](auto MI) {
                constexpr int mi = decltype(MI)::value;
                lgkm_release<RPH - 1 - (P0 + 1 + mi)>(wf[sb][0], af[sb][mi]);
                MM::run(acc[0][mi], wf[sb][0], af[sb][mi]);
                __builtin_amdgcn_sched_barrier(0);
            });
            static_for<1, NT_>([&](auto NI) {
                constexpr int ni = decltype(NI)::value;
                lgkm_release<RPH - 1 - (P0 + MT_ + ni)>(wf[sb][ni]);
#pragma unroll
                for (int mi = 0; mi < MT_; ++mi) MM::run(acc[ni][mi], wf[sb][ni], af[sb][mi]);
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    };
    auto compute = [&](int stage) {
        const char* base = smem + stage * STAGE;
#pragma unroll
        for (int ks = 0; ks < KC / 4; ++ks) {        // the swizzle is an XOR, so the second 64-byte half is offset ^ 64
#pragma unroll
            for (int sb = 0; sb < KSUB; ++sb) {
                u32x4 af[MT_], wf[NT_];
#pragma unroll
                for (int mi = 0; mi < MT_; ++mi) af[mi] = ld_chunk(base + (fa0 ^ (ks * 64 + sb * 32)) + mi * FR * ROWB);
#pragma unroll
                for (int ni = 0; ni < NT_; ++ni) wf[ni] = ld_chunk(base + (fw0 ^ (ks * 64 + sb * 32)) + ni * FR * ROWB);
#pragma unroll
                for (int ni = 0; ni < NT_; ++ni)
#pragma unroll
                    for (int mi = 0; mi < MT_; ++mi) MM::run(acc[ni][mi], wf[ni], af[mi]);
            }
        }
    };

#ifdef SAICV_NT_TIMELINE
    if (tl_on) { int t_ = kpos + fa0; asm volatile("" : "+v"(t_)); NT_STAMP(10); }
#endif
    const int nkt = (Kc + BK - 1) / BK;        // 0 for a class without taps: the output is zero
    // r05 -- the bias starts the accumulators: out = bias + sum, and the epilogue's 16 masked loads + adds per thread behind the K loop
    // (1.5 us of a 27 us ViT-B tile, profiles/r05_nt_timeline.md) are gone.  The four values per 16-column fragment were requested
    // at the top of the tile (bias_v, a compiler-visible load: an assembly load would be spilled before it has landed) and are consumed
    // HERE, before the first DMA is issued, so that the wait the compiler places does not drain the DMA ring.
    if (bias_early) {
#pragma unroll
        for (int ni = 0; ni < NT_; ++ni)
#pragma unroll
            for (int mi = 0; mi < MT_; ++mi)
#pragma unroll
                for (int r = 0; r < MM::NV; ++r) acc[ni][mi][r] = __uint_as_float(bias_v[ni][r / 4][r % 4]);
    }
    // prologue: NSTAGE-1 tiles in flight
    int issued = 0;
    for (; issued < NSTAGE - 1 && issued < nkt; ++issued) issue_tile(issued);
    NT_STAMP(1);
    int st_c = 0;                              // ring slot of the tile being consumed
    int st_i = issued % NSTAGE;                // ring slot the next DMA fills
    for (int kt = 0; kt < nkt; ++kt) {
        // retire tile kt only: the tiles issued after it stay in flight across the barrier
        const int ahead = issued - kt - 1;     // wave-uniform
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPT) : "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // everyone's part of tile kt landed; tile kt-1 fully consumed
#ifdef SAICV_NT_TIMELINE
        if (kt == 0) NT_STAMP(2);
#endif
        if constexpr (ASM_FRAGS) {
            u32x4 af[KSUB][MT_], wf[KSUB][NT_];
            frag_reads(st_c, std::integral_constant<int, 0>{}, af, wf);      // the first fragments travel while the DMA below is issued
            if (issued < nkt) {                // refill the slot tile kt-1 just vacated
                issue_tile(st_i);
                ++issued;
                st_i = (st_i + 1 == NSTAGE) ? 0 : st_i + 1;
            }
            frag_mma(af, wf);
            if constexpr (KC == 8) {
                frag_reads(st_c, std::integral_constant<int, 1>{}, af, wf);
                frag_mma(af, wf);
            }
        } else {
            if (issued < nkt) {                // refill the slot tile kt-1 just vacated
                issue_tile(st_i);
                ++issued;
                st_i = (st_i + 1 == NSTAGE) ? 0 : st_i + 1;
            }
            compute(st_c);
        }
        st_c = (st_c + 1 == NSTAGE) ? 0 : st_c + 1;
    }
    }
    __syncthreads();                           // LDS is reused by the epilogue
    NT_STAMP(3);

    // ---- epilogue.  acc[ni][mi][r]: n = n_base + ni*FR + fcol(r / 4) + r % 4 ; m = m_base + mi*FR + lr
    // BN statistics come straight from the accumulators; the output tile is staged through LDS
    // so that HBM sees whole rows written 16 bytes per lane (a lane's fragment is only 4 values
    // of one row: storing it directly gives 32- or 64-byte row segments and half the write bandwidth -- measured).
    // CODE SIZE is what this section is tuned for: with one workgroup per CU nothing overlaps the epilogue, and a
    // fully unrolled epilogue that carries every fused mode in every unrolled copy (6 k instructions, 48 KiB)
    // spent most of its 6 us per tile waiting for instruction fetch.  So: the unrolled part (accumulator -> LDS)
    // carries no alternatives, the common copy-out is 16 loads + 16 stores, and everything with a fused operand,
    // a row remap, an N tail or an unaligned leading dimension runs in ROLLED loops (one copy of the mode code).
    // the thread coordinates of the epilogue are re-derived from an opaque copy of the thread index: nothing of the epilogue's
    // address arithmetic can be hoisted above the K loop and held in registers (or spilled) across it
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    {
    const int tid = tid_e;
    const int lane = tid & 63;
    const int l15 = lane & 15;
    const int lg = lane >> 4;
    const int lr = lane & (FR - 1);
    const int lk = lane / FR;
    auto fcol = [&](int g) { return FR == 32 ? 8 * g + 4 * lk : 4 * lk; };
    typedef typename std::conditional<OUT_F32, float, T>::type TO;
    constexpr int OPITCH = BN_T * (int)sizeof(TO) + 16;         // bytes; +16 staggers banks
    constexpr int OCPR = BN_T * (int)sizeof(TO) / 16;           // 16-byte chunks per tile row
    constexpr int OEPC = 16 / (int)sizeof(TO);
    const int m_base = tile_m * BM_T + wm * WMR;
    const int n_base = tile_n * BN_T + wn * WN;
    const bool do_stats = p.stat_sum != nullptr;
    // bf16 outputs: the BN statistics are taken from the STAGED tile by the matrix cores (below), not here
    constexpr bool MSTAT = !OUT_F32 && sizeof(T) == 2;
    const bool valu_stats = do_stats && !MSTAT;
#pragma unroll
    for (int ni = 0; ni < NT_; ++ni) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int n0 = n_base + ni * FR + fcol(g);
            float bs[4] = {0.f, 0.f, 0.f, 0.f};
            if (p.bias != nullptr && !bias_early) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n0 + r < p.Nn) bs[r] = p.bias[n0 + r];
            }
            float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
            char* q0 = smem + (wm * WMR + lr) * OPITCH + (wn * WN + ni * FR + fcol(g)) * (int)sizeof(TO);
#pragma unroll
            for (int mi = 0; mi < MT_; ++mi) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[ni][mi][4 * g + r] + bs[r];
                if (valu_stats) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float vr = OUT_F32 ? v[r] : round_through<T>(v[r]);
                        ssum[r] += vr;
                        ssq[r] += vr * vr;
                    }
                }
                char* q = q0 + mi * FR * OPITCH;
                if (sizeof(TO) == 2) {
                    bf16x4 pk;
#pragma unroll
                    for (int r = 0; r < 4; ++r) pk[r] = (bf16_t)v[r];
                    *reinterpret_cast<bf16x4*>(q) = pk;
                } else {
                    *reinterpret_cast<f32x4*>(q) = f32x4{v[0], v[1], v[2], v[3]};
                }
            }
            if (valu_stats) {
                // rows m >= M were gathered as zeros (no bias when stats are requested) -> add 0.
#pragma unroll
                for (int r = 0; r < 4; ++r) {        // over the FR pixel lanes that share these columns
                    ssum[r] = row16_sum(ssum[r]);
                    ssq[r] = row16_sum(ssq[r]);
                    if (FR == 32) {
                        ssum[r] += __shfl_xor(ssum[r], 16, 64);
                        ssq[r] += __shfl_xor(ssq[r], 16, 64);
                    }
                }
                if (lr == 0) {                       // per-wavefront column sums -> LDS [2][WM_][BN_T] behind the staged tile
                    float* ws = reinterpret_cast<float*>(smem + BM_T * OPITCH) + wm * BN_T + wn * WN + ni * FR + fcol(g);
                    *reinterpret_cast<f32x4*>(ws) = f32x4{ssum[0], ssum[1], ssum[2], ssum[3]};
                    *reinterpret_cast<f32x4*>(ws + WM_ * BN_T) = f32x4{ssq[0], ssq[1], ssq[2], ssq[3]};
                }
            }
        }
    }
    __syncthreads();
    NT_STAMP(4);
    if constexpr (MSTAT) {
        if (do_stats) {
            // Column sums of the staged bf16 tile on the matrix cores: with Y the [32 rows][16 cols] block read
            // TRANSPOSED from LDS (ds_read_b64_tr_b16: lane (col, k-group) gets 8 consecutive rows of its column),
            //   sum_m y      = (1 . Y)[any row][col]          -- A = ones
            //   sum_m y * y  = (Y^T . Y)[col][col]            -- A = B = the same fragment, the diagonal
            // exact products, fp32 accumulation, and exactly the values the next kernel reads (bf16-rounded).
            // One wavefront owns 16 columns over ALL rows of the tile: no cross-wave combine, and the 14 VALU
            // operations per output element + 128 DPP adds of the accumulator version are gone from the epilogue.
            typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
            const u32x4 ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
            for (int cg = wave; cg < BN_T / 16; cg += NWAVES) {
                f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
                const char* q = smem + (lg * 8 + (l15 >> 2)) * OPITCH + (cg * 16 + (l15 & 3) * 4) * 2;
#pragma unroll 4
                for (int rb = 0; rb < BM_T / 32; ++rb, q += 32 * OPITCH) {
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(q));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(q + 4 * OPITCH));
                    const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
                    const u32x4 y = {l2[0], l2[1], h2[0], h2[1]};
                    Mma<T>::run(s1, ones, y);
                    Mma<T>::run(s2, y, y);
                }
                const int n = tile_n * BN_T + cg * 16 + l15;          // D: row lg*4 + r, column l15
                if (n < p.Nn) {
                    const size_t srow = p.stat_atomic_rows ? (size_t)(tile_m % p.stat_atomic_rows) : (size_t)tile_m;
                    if (lg == 0) {
                        if (p.stat_atomic_rows) unsafeAtomicAdd(&p.stat_sum[srow * (size_t)p.Nn + n], s1[0]);
                        else p.stat_sum[srow * (size_t)p.Nn + n] = s1[0];
                    }
                    if (lg == (l15 >> 2)) {
                        const int r = l15 & 3;
                        const float qv = r == 0 ? s2[0] : r == 1 ? s2[1] : r == 2 ? s2[2] : s2[3];
                        if (p.stat_atomic_rows) unsafeAtomicAdd(&p.stat_sq[srow * (size_t)p.Nn + n], qv);
                        else p.stat_sq[srow * (size_t)p.Nn + n] = qv;
                    }
                }
            }
        }
    } else
    if (do_stats) {
        // one row of partial statistics per WORKGROUP (fixed summation order over its wavefront rows): four times
        // fewer partial rows for the finalize kernels to read than one row per wavefront row
        for (int c = tid; c < 2 * BN_T; c += NTHREADS) {
            const int which = c / BN_T, col = c - which * BN_T;
            const float* ws = reinterpret_cast<const float*>(smem + BM_T * OPITCH) + which * WM_ * BN_T + col;
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < WM_; ++w) a += ws[w * BN_T];
            const int n = tile_n * BN_T + col;
            if (n < p.Nn) {
                float* dst = (which ? p.stat_sq : p.stat_sum) +
                             (size_t)(p.stat_atomic_rows ? tile_m % p.stat_atomic_rows : tile_m) * (size_t)p.Nn + n;
                if (p.stat_atomic_rows) unsafeAtomicAdd(dst, a); else *dst = a;
            }
        }
    }
    NT_STAMP(5);
    const bool stream_out = p.stream_out != 0;      // uniform
    const int oc = tid % OCPR;               // chunk within the tile row
    const int orow0 = tid / OCPR;
    constexpr int RPP = NTHREADS / OCPR;     // tile rows per pass
    constexpr int NIT = BM_T / RPP;
    const int ncol = tile_n * BN_T + oc * OEPC;
    TO* const outp = reinterpret_cast<TO*>(p.out);
    const bool aligned = ((p.ldo * (int)sizeof(TO)) & 15) == 0;
    const bool remap = MODE == 1 && cs > 1;
    constexpr bool DGRAD_EXTRAS = MODE == 1 && sizeof(TO) == sizeof(T);       // gated shortcut / BatchNorm-backward sums: data gradient only
    const bool bstats = DGRAD_EXTRAS && p.bs_y != nullptr;                                                        // uniform
    const bool plain = aligned && !remap && p.act_mode == 0 && p.addend == nullptr && p.row_scale == nullptr && !bstats;   // uniform
    float bag[OEPC], bax[OEPC];                                // this thread's sum g, sum g * y over its rows of the tile
#pragma unroll
    for (int j = 0; j < OEPC; ++j) { bag[j] = 0.f; bax[j] = 0.f; }
    if (plain && ncol + OEPC <= p.Nn) {
        // the common case: a branch-free copy, the LDS reads of eight rows in flight before the first store
        constexpr int GRP = NIT < 8 ? NIT : 8;
        const int mrow0 = tile_m * BM_T + orow0;
        const char* ls = smem + orow0 * OPITCH + oc * 16;
        TO* o = outp + (size_t)mrow0 * p.ldo + ncol;
        const size_t ostep = (size_t)RPP * p.ldo;
        const bool full = (tile_m + 1) * BM_T <= Mc;                                         // uniform
#pragma unroll
        for (int g = 0; g < NIT; g += GRP) {
            u32x4 v[GRP];
#pragma unroll
            for (int j = 0; j < GRP; ++j) v[j] = ld_chunk(ls + (g + j) * RPP * OPITCH);
            if (full) {
#pragma unroll
                for (int j = 0; j < GRP; ++j) NT_OUT_ST(o + (g + j) * ostep, v[j]);
            } else {
#pragma unroll
                for (int j = 0; j < GRP; ++j)
                    if (mrow0 + (g + j) * RPP < Mc) NT_OUT_ST(o + (g + j) * ostep, v[j]);
            }
        }
    } else if (aligned && !remap && (p.act_mode == 0 || p.act_mode == 4) && (p.Nn % OEPC) == 0) {   // uniform
        // residual add / drop-path scale (forward and data gradient) and the data gradient's gated shortcut and
        // BatchNorm-backward sums, on dense rows of whole chunks: four rows at a time, every global load of the group
        // (addend, y, the two mask bytes) in flight before the first use
        if (ncol < p.Nn) {
            constexpr int GRP = NIT < 4 ? NIT : 4;
            const int mrow0 = tile_m * BM_T + orow0;
            const char* ls = smem + orow0 * OPITCH + oc * 16;
            const bool addp = p.addend != nullptr, scalep = p.row_scale != nullptr;
            const bool mulp = p.act_mode == 4;          // out = (acc + bias) * addend: the stored gelu'() of an MLP (grouped loads instead of the rolled loop)
            const bool gatep = DGRAD_EXTRAS && p.addend_gate != nullptr, maskp = DGRAD_EXTRAS && p.bs_mask != nullptr;
            const TO* const addend = reinterpret_cast<const TO*>(p.addend);
            const TO* const ybn = reinterpret_cast<const TO*>(p.bs_y);
#pragma unroll 1
            for (int g = 0; g < NIT; g += GRP) {
                u32x4 v[GRP], av[GRP], yv[GRP];
                unsigned gb[GRP], mb[GRP];
                float sc[GRP];
#pragma unroll
                for (int j = 0; j < GRP; ++j) {
                    const int mrow = mrow0 + (g + j) * RPP;
                    const bool ok = mrow < Mc;
                    const size_t off = (size_t)mrow * p.ldo + ncol;
                    v[j] = ld_chunk(ls + (g + j) * RPP * OPITCH);
                    av[j] = u32x4{0u, 0u, 0u, 0u};
                    yv[j] = u32x4{0u, 0u, 0u, 0u};
                    gb[j] = 0xffu;
                    mb[j] = 0xffu;
                    sc[j] = 1.f;
                    if (ok) {
                        if (scalep) sc[j] = p.row_scale[mrow / p.rows_per_scale];
                        if (addp) av[j] = NT_EPI_LD(addend + off);
                        if (bstats) yv[j] = NT_EPI_LD(ybn + off);
                        if (gatep) gb[j] = p.addend_gate[off / OEPC];
                        if (bstats && maskp) mb[j] = p.bs_mask[off / OEPC];
                    }
                }
#pragma unroll
                for (int j = 0; j < GRP; ++j) {
                    const int mrow = mrow0 + (g + j) * RPP;
                    if (mrow < Mc) {
                        float f[OEPC];
                        Chunk<TO>::unpack(v[j], f);
                        if (addp || scalep) {
                            float a[OEPC];
                            Chunk<TO>::unpack(av[j], a);
                            if (mulp) {
#pragma unroll
                                for (int e = 0; e < OEPC; ++e) f[e] *= a[e];
                            } else {
#pragma unroll
                                for (int e = 0; e < OEPC; ++e) f[e] = fmaf(sc[j], f[e], ((gb[j] >> e) & 1u) ? a[e] : 0.f);
                            }
                            v[j] = Chunk<TO>::pack(f);
                            if (bstats) Chunk<TO>::unpack(v[j], f);      // the sums are over what is stored
                        }
                        if (bstats) {
                            float yy[OEPC];
                            Chunk<TO>::unpack(yv[j], yy);
#pragma unroll
                            for (int e = 0; e < OEPC; ++e) {
                                const float ge = ((mb[j] >> e) & 1u) ? f[e] : 0.f;
                                bag[e] += ge;
                                bax[e] = fmaf(ge, yy[e], bax[e]);
                            }
                        }
                        NT_OUT_ST(outp + (size_t)mrow * p.ldo + ncol, v[j]);
                    }
                }
            }
        }
    } else if (p.lean_gelu && aligned && !remap && (p.act_mode == 1 || p.act_mode == 3) && p.addend == nullptr && p.row_scale == nullptr && !bstats &&
               (p.Nn % OEPC) == 0) {   // uniform
        // fc1 of an MLP (r04): out = the pre-activation (1) or gelu'(pre) (3), out2 = gelu(pre).  Nothing but the staged tile is read,
        // so the row loop is just unpack -> packed-fp32 GELU pieces -> two packs -> two 16-byte stores: 322 -> ~150 vector
        // instructions per row of eight against the general rolled loop below (an epilogue instruction competes with the other
        // workgroup's MFMAs for the issue port)
        if (ncol < p.Nn) {
            const int mrow0 = tile_m * BM_T + orow0;
            const char* ls = smem + orow0 * OPITCH + oc * 16;
            TO* const o2 = reinterpret_cast<TO*>(p.out2);
            const bool emit_grad = p.act_mode == 3;
#pragma unroll 2
            for (int g = 0; g < NIT; ++g) {
                const int mrow = mrow0 + g * RPP;
                if (mrow < Mc) {
                    const u32x4 v = ld_chunk(ls + g * RPP * OPITCH);
                    float f[OEPC], gl[OEPC], gr[OEPC];
                    Chunk<TO>::unpack(v, f);
                    gelu_and_grad8(f, gl, gr);
                    const size_t off = (size_t)mrow * p.ldo + ncol;
                    NT_OUT_ST(o2 + off, Chunk<TO>::pack(gl));
                    NT_OUT_ST(outp + off, emit_grad ? Chunk<TO>::pack(gr) : v);
                }
            }
        }
    } else if (ncol < p.Nn) {
        // one copy of the fused-mode code in a ROLLED loop; the staged chunk and the addend chunk of the NEXT row are
        // fetched before the current row is processed, so a row's global-load latency hides under its predecessor
        const bool whole = aligned && ncol + OEPC <= p.Nn;
        const bool pre_add = whole && p.addend != nullptr;
        const bool gated = DGRAD_EXTRAS && pre_add && p.addend_gate != nullptr;   // host: the gate needs whole, aligned chunks
        const bool bst = bstats && whole;
        // per row: gate byte of the addend (bits 0-7) | ReLU-mask byte of the statistics (bits 8-15)
        auto side_bits = [&](int mrow) -> unsigned {
            const size_t ch = ((size_t)mrow * p.ldo + ncol) / OEPC;
            unsigned b = 0xffffu;
            if (gated) b = (b & 0xff00u) | p.addend_gate[ch];
            if (bst && p.bs_mask != nullptr) b = (b & 0x00ffu) | ((unsigned)p.bs_mask[ch] << 8);
            return b;
        };
        auto out_row = [&](int rr) -> int {                // output row of tile row rr, -1 past the end
            const int mrow = tile_m * BM_T + rr;
            if (rr >= BM_T || mrow >= Mc) return -1;
            if (!remap) return mrow;
            const int img = mrow / ohw;
            const int rem = mrow - img * ohw;
            const int hc = rem / Wc;
            return (img * p.OH + hc * cs + ph) * p.OW + (rem - hc * Wc) * cs + pw;
        };
        int rr = orow0;
        int m = out_row(rr);
        u32x4 v = {0u, 0u, 0u, 0u}, av = {0u, 0u, 0u, 0u}, yv = {0u, 0u, 0u, 0u};
        unsigned sb = 0xffffu;
        if (m >= 0) {
            v = ld_chunk(smem + rr * OPITCH + oc * 16);
            if (pre_add) av = ld_chunk(reinterpret_cast<const TO*>(p.addend) + (size_t)m * p.ldo + ncol);
            if (bst) yv = ld_chunk(reinterpret_cast<const TO*>(p.bs_y) + (size_t)m * p.ldo + ncol);
            if (gated || bst) sb = side_bits(m);
        }
#pragma unroll 1
        while (m >= 0) {
            const int rn = rr + RPP;
            const int mn = out_row(rn);
            u32x4 vn = {0u, 0u, 0u, 0u}, an = {0u, 0u, 0u, 0u}, yn = {0u, 0u, 0u, 0u};
            unsigned sbn = 0xffffu;
            if (mn >= 0) {
                vn = ld_chunk(smem + rn * OPITCH + oc * 16);
                if (pre_add) an = ld_chunk(reinterpret_cast<const TO*>(p.addend) + (size_t)mn * p.ldo + ncol);
                if (bst) yn = ld_chunk(reinterpret_cast<const TO*>(p.bs_y) + (size_t)mn * p.ldo + ncol);
                if (gated || bst) sbn = side_bits(mn);
            }
            TO* o = outp + (size_t)m * p.ldo + ncol;
            if (p.act_mode == 1 || p.act_mode == 3) {   // fc1 of an MLP: emit gelu() beside the pre-activation (1) or beside gelu'() (3)
                float f[OEPC], gl[OEPC], gr[OEPC];
                Chunk<TO>::unpack(v, f);
                gelu_and_grad8(f, gl, gr);
                NT_OUT_ST(reinterpret_cast<TO*>(p.out2) + (size_t)m * p.ldo + ncol, Chunk<TO>::pack(gl));
                if (p.act_mode == 3) v = Chunk<TO>::pack(gr);
            } else if (p.act_mode == 2 || p.act_mode == 4) {     // dgrad of fc2: d pre-activation = d act * gelu'(pre) (2) or * the stored derivative (4)
                float f[OEPC], a[OEPC];
                Chunk<TO>::unpack(v, f);
                Chunk<TO>::unpack(av, a);
                if (p.act_mode == 2) {
#pragma unroll
                    for (int j = 0; j < OEPC; ++j) f[j] *= gelu_grad_f(a[j]);
                } else {
#pragma unroll
                    for (int j = 0; j < OEPC; ++j) f[j] *= a[j];
                }
                v = Chunk<TO>::pack(f);
            } else if (p.addend != nullptr || p.row_scale != nullptr) {
                float f[OEPC], a[OEPC];
                Chunk<TO>::unpack(v, f);
                const float sc = p.row_scale ? p.row_scale[m / p.rows_per_scale] : 1.f;
                if (pre_add) {
                    Chunk<TO>::unpack(av, a);
                    if (gated) {
#pragma unroll
                        for (int j = 0; j < OEPC; ++j) a[j] = ((sb >> j) & 1u) ? a[j] : 0.f;
                    }
                } else {
                    for (int j = 0; j < OEPC; ++j)
                        a[j] = (p.addend != nullptr && ncol + j < p.Nn)
                                   ? to_f32(reinterpret_cast<const TO*>(p.addend)[(size_t)m * p.ldo + ncol + j]) : 0.f;
                }
#pragma unroll
                for (int j = 0; j < OEPC; ++j) f[j] = fmaf(sc, f[j], a[j]);
                v = Chunk<TO>::pack(f);
            }
            if (bst) {                        // BatchNorm-backward sums of what is stored (rounded to TO), behind its ReLU gate
                float f[OEPC], yy[OEPC];
                Chunk<TO>::unpack(v, f);
                Chunk<TO>::unpack(yv, yy);
#pragma unroll
                for (int j = 0; j < OEPC; ++j) {
                    const float gj = ((sb >> (8 + j)) & 1u) ? f[j] : 0.f;
                    bag[j] += gj;
                    bax[j] = fmaf(gj, yy[j], bax[j]);
                }
            }
            if (whole) {
                NT_OUT_ST(o, v);
            } else {                          // N tail, or a leading dimension without 16-byte alignment
                const TO* e = reinterpret_cast<const TO*>(&v);
                for (int j = 0; j < OEPC; ++j)
                    if (ncol + j < p.Nn) o[j] = e[j];
            }
            rr = rn; m = mn; v = vn; av = an; yv = yn; sb = sbn;
        }
    }
    if (bstats) {       // uniform: combine the row lanes of every column through the (now consumed) staging area
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);                      // [RPP][2][BN_T]
        const bool have = ncol + OEPC <= p.Nn;
#pragma unroll
        for (int j = 0; j < OEPC; ++j) {
            // sum g * (y - mean) * invstd = invstd * (sum g y - mean * sum g): the mean leaves after this thread's few
            // rows (BM_T / RPP of them), while the sums are still small -- not after the whole column
            const float mu = have ? p.bs_mean[ncol + j] : 0.f, is = have ? p.bs_invstd[ncol + j] : 0.f;
            red[(orow0 * 2 + 0) * BN_T + oc * OEPC + j] = bag[j];
            red[(orow0 * 2 + 1) * BN_T + oc * OEPC + j] = is * fmaf(-mu, bag[j], bax[j]);
        }
        __syncthreads();
        const size_t prow = (size_t)((MODE == 1 && cs > 1) ? blockIdx.y : 0) * p.bs_rows + tile_m;
        for (int c = tid; c < 2 * BN_T; c += NTHREADS) {
            const int which = c / BN_T, col = c - which * BN_T;
            float a = 0.f;
            for (int r = 0; r < RPP; ++r) a += red[(r * 2 + which) * BN_T + col];
            const int n = tile_n * BN_T + col;
            if (n < p.Nn) {
                float* dst = (which ? p.bs_gx : p.bs_g) + (p.stat_atomic_rows ? prow % p.stat_atomic_rows : prow) * (size_t)p.Nn + n;
                if (p.stat_atomic_rows) unsafeAtomicAdd(dst, a); else *dst = a;
            }
        }
    }
    }   // epilogue scope
#ifdef SAICV_NT_TIMELINE
    if (tl_on && tix == (int)blockIdx.x) {
        NT_STAMP(6);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the workgroup's stores are acknowledged
        NT_STAMP(7);
        if (tid == 0) {
            unsigned long long* rec = p.timeline + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16;
            unsigned int hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
#pragma unroll
            for (int i = 0; i < 8; ++i) rec[i] = tl_t[i];
            rec[8] = ((unsigned long long)xcc << 32) | hw;
            rec[9] = __builtin_amdgcn_s_memrealtime();
            rec[10] = tl_rt0;
            rec[11] = tl_t[8]; rec[12] = tl_t[9]; rec[13] = tl_t[10];
        }
    }
#endif
    }   // tile loop
}


// ------------------------------------------------------------------------------------ TN
struct TNParams {
    const void* dy;     // [M][Cout]
    const void* src;    // [Nimg,H,W,C] gather source (layer input)
    float* dw;          // [Cout][Kd] fp32, accumulated with atomics
    float* dbias;       // optional [Cout] fp32: column sums of dy (bias gradient), accumulated
    uint32_t dy_bytes, src_bytes;
    int H, W, C;
    int OH, OW;
    int R, S, stride, pad;
    int M, Cout, Kd;
    int tiles_a, tiles_b;       // tiles over Cout / over Kd
    int m_per_split;            // multiple of BR
    int d_img, d_oh, d_ow;      // mixed-radix digits of BR in (img, oh, ow)
    FastDiv fd_ohw, fd_ow;
    saicv::DetSink det;         // deterministic mode: split s parks its tile in part[s][n * Kd + kk], its bias sums in part[s][Cout * Kd + n]
};

template <typename T, int BA, int BB, int NWA, int NWB>
__global__ __launch_bounds__(64 * NWA * NWB) void igemm_tn_kernel(const TNParams p) {
    constexpr int THREADS = 64 * NWA * NWB;
    constexpr int EPC = ElemTraits<T>::EPC;
    constexpr int BR = 8 * EPC;                  // reduction rows per tile (64 bf16 / 32 f32)
    constexpr int PITCH_A = (BA + 16) * (int)sizeof(T);
    constexpr int PITCH_B = (BB + 16) * (int)sizeof(T);
    constexpr int A_BYTES = BR * PITCH_A;
    constexpr int B_BYTES = BR * PITCH_B;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int CPR_A = BA / EPC, CPR_B = BB / EPC;     // chunks per row
    constexpr int RPP_A = THREADS / CPR_A, RPP_B = THREADS / CPR_B;
    constexpr int NA = BR / RPP_A, NB = BR / RPP_B;       // chunks per thread
    constexpr int WA = BA / NWA, WB = BB / NWB;
    constexpr int AT = WA / 16, BT = WB / 16;
    constexpr uint32_t OOB = 0xfffffff0u;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wa = wave % NWA, wb = wave / NWA;
    const int l15 = lane & 15, lg = lane >> 4;

    // XCD-aware mapping: workgroups are dealt to the 8 XCDs round-robin in launch order (x fastest).  All output
    // tiles of one reduction split read the SAME rows of dY and of the gather source, so each XCD gets a contiguous
    // range of (split, tile) pairs in split-major order: those rows are fetched into one L2 instead of eight
    // (PMC: the weight-gradient kernels read 25 GB per ResNet-50 step against 11 GB of operands).
    const int ntile = gridDim.x;
    const int lid = xcd_remap(blockIdx.y * ntile + blockIdx.x, ntile * gridDim.y);
    const int split = lid / ntile;
    const int tile = lid - split * ntile;
    const int tile_a = tile % p.tiles_a;
    const int tile_b = tile / p.tiles_a;
    const int m_begin = split * p.m_per_split;
    const int m_end = min(p.M, m_begin + p.m_per_split);
    if (m_begin >= m_end) return;

    const __amdgpu_buffer_rsrc_t dy_rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dy), 0, p.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t src_rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src), 0, p.src_bytes, 0x00020000);

    // dY loader state: byte offset of this thread's chunk in row (m_begin + rba), advanced per tile
    const int ca = tid % CPR_A, rba = tid / CPR_A;
    const int n_ld = tile_a * BA + ca * EPC;
    const bool n_ok = n_ld < p.Cout;
    uint32_t dy_off = (uint32_t)((m_begin + rba) * p.Cout + n_ld) * (uint32_t)sizeof(T);
    const uint32_t dy_row_step = (uint32_t)(RPP_A * p.Cout) * (uint32_t)sizeof(T);
    const uint32_t dy_tile_step = (uint32_t)(BR * p.Cout) * (uint32_t)sizeof(T);
    int m_a = m_begin + rba;
    // gather loader state
    const int cb = tid % CPR_B, rbb = tid / CPR_B;
    const int kk_ld = tile_b * BB + cb * EPC;
    const bool kk_ok = kk_ld < p.Kd;
    const int tap = kk_ld / p.C;
    const int c0 = kk_ld - tap * p.C;
    const int fr = tap / p.S - p.pad;            // tap offsets with the padding folded in
    const int fs = tap - (tap / p.S) * p.S - p.pad;
    int g_img[NB], g_oh[NB], g_ow[NB];
    const int ohw = p.OH * p.OW;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int m = m_begin + rbb + RPP_B * i;
        const int img = (int)fdiv((uint32_t)m, p.fd_ohw);
        const int rem = m - img * ohw;
        g_img[i] = img;
        g_oh[i] = (int)fdiv((uint32_t)rem, p.fd_ow);
        g_ow[i] = rem - g_oh[i] * p.OW;
    }
    int m_b = m_begin + rbb;

    u32x4 ra[NA], rbv[NB];
    // bias gradient for free: the workgroups of the first kk tile also sum the dY chunks they stream
    const bool do_bias = (p.dbias != nullptr) && (tile_b == 0);
    float bsum[EPC];
#pragma unroll
    for (int k = 0; k < EPC; ++k) bsum[k] = 0.f;
    auto load_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const bool ok = n_ok & (m_a + RPP_A * i < m_end);
            ra[i] = buf_ld(dy_rs, ok ? dy_off + (uint32_t)i * dy_row_step : OOB);
        }
        dy_off += dy_tile_step;
        m_a += BR;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int ih = g_oh[i] * p.stride + fr;
            const int iw = g_ow[i] * p.stride + fs;
            const bool ok = kk_ok & (m_b + RPP_B * i < m_end) & ((unsigned)ih < (unsigned)p.H) &
                            ((unsigned)iw < (unsigned)p.W);
            const uint32_t off = (uint32_t)(((g_img[i] * p.H + ih) * p.W + iw) * p.C + c0) * (uint32_t)sizeof(T);
            rbv[i] = buf_ld(src_rs, ok ? off : OOB);
            // advance this row by BR pixels (mixed radix add with carries)
            int ow = g_ow[i] + p.d_ow;
            int cy = ow >= p.OW;
            ow -= cy ? p.OW : 0;
            int oh = g_oh[i] + p.d_oh + cy;
            cy = oh >= p.OH;
            oh -= cy ? p.OH : 0;
            g_ow[i] = ow;
            g_oh[i] = oh;
            g_img[i] += p.d_img + cy;
        }
        m_b += BR;
    };
    int st_a[NA], st_b[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) st_a[i] = (rba + RPP_A * i) * PITCH_A + ca * 16;
#pragma unroll
    for (int i = 0; i < NB; ++i) st_b[i] = A_BYTES + (rbb + RPP_B * i) * PITCH_B + cb * 16;
    auto store_tile = [&](int stage) {
        char* base = smem + stage * STAGE;
        if (do_bias) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                float f[EPC];
                Chunk<T>::unpack(ra[i], f);
#pragma unroll
                for (int k = 0; k < EPC; ++k) bsum[k] += f[k];
            }
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) st_chunk(base + st_a[i], ra[i]);
#pragma unroll
        for (int i = 0; i < NB; ++i) st_chunk(base + st_b[i], rbv[i]);
    };

    f32x4 acc[AT][BT];
#pragma unroll
    for (int ai = 0; ai < AT; ++ai)
#pragma unroll
        for (int bi = 0; bi < BT; ++bi) acc[ai][bi] = f32x4{0.f, 0.f, 0.f, 0.f};

    // LDS fragment offsets (stage 0), hoisted out of the reduction loop
    int fa_off[AT], fb_off[BT];
    if constexpr (sizeof(T) == 2) {
        // ds_read_b64_tr_b16: a 16-lane group reads a [4 rows][16 cols] block; lane t supplies the
        // 8-byte address of row (t>>2), cols (t&3)*4.. and receives column t of the four rows.
        const int row0 = lg * 8 + (l15 >> 2);
#pragma unroll
        for (int ai = 0; ai < AT; ++ai) fa_off[ai] = row0 * PITCH_A + (wa * WA + ai * 16 + (l15 & 3) * 4) * 2;
#pragma unroll
        for (int bi = 0; bi < BT; ++bi) fb_off[bi] = A_BYTES + row0 * PITCH_B + (wb * WB + bi * 16 + (l15 & 3) * 4) * 2;
    } else {
#pragma unroll
        for (int ai = 0; ai < AT; ++ai) fa_off[ai] = lg * PITCH_A + (wa * WA + ai * 16 + l15) * 4;
#pragma unroll
        for (int bi = 0; bi < BT; ++bi) fb_off[bi] = A_BYTES + lg * PITCH_B + (wb * WB + bi * 16 + l15) * 4;
    }

    auto compute = [&](int stage) {
        const char* base = smem + stage * STAGE;
        if constexpr (sizeof(T) == 2) {
            // BR = 64 rows -> two 32-deep k-steps
            typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4 af[AT], bf[BT];
#pragma unroll
                for (int ai = 0; ai < AT; ++ai) {
                    const char* q = base + fa_off[ai] + ks * 32 * PITCH_A;
                    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(q));
                    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(q + 4 * PITCH_A));
                    u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
                    af[ai] = u32x4{l2[0], l2[1], h2[0], h2[1]};
                }
#pragma unroll
                for (int bi = 0; bi < BT; ++bi) {
                    const char* q = base + fb_off[bi] + ks * 32 * PITCH_B;
                    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(q));
                    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(q + 4 * PITCH_B));
                    u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
                    bf[bi] = u32x4{l2[0], l2[1], h2[0], h2[1]};
                }
#pragma unroll
                for (int ai = 0; ai < AT; ++ai)
#pragma unroll
                    for (int bi = 0; bi < BT; ++bi)
                        acc[ai][bi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, af[ai]), __builtin_bit_cast(bf16x8, bf[bi]),
                            acc[ai][bi], 0, 0, 0);
            }
        } else {
            // f32: BR = 32 rows -> eight 4-deep k-steps of mfma 16x16x4 (lane: row/col l15, k = lg)
#pragma unroll
            for (int ks = 0; ks < BR / 4; ++ks) {
                float af[AT], bf[BT];
#pragma unroll
                for (int ai = 0; ai < AT; ++ai)
                    af[ai] = *reinterpret_cast<const float*>(base + fa_off[ai] + ks * 4 * PITCH_A);
#pragma unroll
                for (int bi = 0; bi < BT; ++bi)
                    bf[bi] = *reinterpret_cast<const float*>(base + fb_off[bi] + ks * 4 * PITCH_B);
#pragma unroll
                for (int ai = 0; ai < AT; ++ai)
#pragma unroll
                    for (int bi = 0; bi < BT; ++bi)
                        acc[ai][bi] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ai], bf[bi],
                                                                          acc[ai][bi], 0, 0, 0);
            }
        }
    };

    const int nt = (m_end - m_begin + BR - 1) / BR;
    load_tile();
    store_tile(0);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        const bool more = (t + 1) < nt;
        if (more) load_tile();
        compute(cur);
        if (more) store_tile(cur ^ 1);
        __syncthreads();
    }

    if (do_bias) {          // combine the RPP_A row lanes of each chunk column through LDS (now free)
        float* red = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int k = 0; k < EPC; ++k) red[tid * EPC + k] = bsum[k];
        __syncthreads();
        if (rba == 0 && n_ok) {
#pragma unroll
            for (int k = 0; k < EPC; ++k) {
                float sgm = 0.f;
                for (int t = 0; t < RPP_A; ++t) sgm += red[(t * CPR_A + ca) * EPC + k];
                saicv::det_add(p.det, p.dbias + n_ld + k, (size_t)p.Cout * p.Kd + n_ld + k, split, sgm);
            }
        }
    }
    // epilogue: D row -> n = .. + lg*4 + r ; D col -> kk = .. + l15
#pragma unroll
    for (int ai = 0; ai < AT; ++ai) {
        const int n0 = tile_a * BA + wa * WA + ai * 16 + lg * 4;
#pragma unroll
        for (int bi = 0; bi < BT; ++bi) {
            const int kk = tile_b * BB + wb * WB + bi * 16 + l15;
            if (kk < p.Kd) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n0 + r < p.Cout)
                        saicv::det_add(p.det, p.dw + (size_t)(n0 + r) * (size_t)p.Kd + kk, (size_t)(n0 + r) * (size_t)p.Kd + kk, split, acc[ai][bi][r]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------ TN, LDS-DMA ring (bf16)
// The weight-gradient product with the operand stream of the forward kernel: `buffer_load ... lds` writes the dY and
// gather-source rows of a 32-row reduction step straight into a 4-slot LDS ring (no staging registers, no ds_write),
// three steps stay in flight across the single raw s_barrier of a step and the wait is a counted s_waitcnt vmcnt(N).
// The register-staged kernel above keeps ONE step (64 rows) in flight: at one resident 256 x 256 workgroup per CU its
// 0.85 us of MFMA work per step is shorter than a loaded L2/HBM round trip, so its loop runs at the latency, not at the
// matrix cores (620-700 TFLOP/s on the ViT shapes where the forward kernel reaches 800-900).
// LDS image of a step: [32 rows][BA] then [32 rows][BB], rows unpadded (the DMA destination is wave-linear, 1 KiB per
// instruction).  ds_read_b64_tr_b16 is served in 32-lane groups that read rows {R..R+3, R+8..R+11}, 32 bytes each, at
// one column offset: the 32-byte PAIR p of row r lives at pair slot p ^ g(r), g = (r & 3) | ((r >> 3) & 1) << 2 (halved
// for 128-byte rows, where two rows share a 256-byte bank window), so the eight rows hit eight distinct bank groups.
// As everywhere, the swizzle is applied on the SOURCE side: lane l fetches the chunk that belongs in the slot it fills.
// Bias gradient: the workgroups of the first kk tile multiply their dY fragments with a ones operand (AT / NWB extra
// MFMAs per wavefront and step) instead of summing staged registers.
DEVINL int tn_key(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }
template <int CPR> DEVINL int tn_g(int r) { return CPR >= 16 ? tn_key(r) : (tn_key(r) >> 1); }

// Convolution form (PLAIN = false): the gathered operand's byte offset and its (oh, ow) are CARRIED from step to step with adds and selects
// (r05; the r03 form rebuilt the offset from (img, oh, ow) with three 32-bit multiplies per DMA instruction -- 58 full-rate + 10
// quarter-rate vector instructions per 16 MFMAs in the K loop, more issue slots than the MFMAs themselves, profiles/r04_nt_experiments.md
// section 6): ResNet-50 22.01 -> 21.93 ms per step on one box (profiles/r05_nt_experiments.md), host emulation of the recurrence in
// tests/test_r04_host.py, GPU parity in tests/test_gpu_kernels.py.
template <int BA, int BB, int NWA, int NWB, bool PLAIN>
__global__ __launch_bounds__(64 * NWA * NWB) void igemm_tn_dma_kernel(const TNParams p) {
    typedef bf16_t T;
    constexpr int NWAVES = NWA * NWB;
    constexpr int BR = 32;                                 // reduction rows per step: one MFMA k-step
    constexpr int CPR_A = BA / 8, CPR_B = BB / 8;           // 16-byte chunks per row
    constexpr int RPI_A = 64 / CPR_A, RPI_B = 64 / CPR_B;   // rows filled by one DMA wave-instruction
    constexpr int NIA = (BR / RPI_A) / NWAVES, NIB = (BR / RPI_B) / NWAVES;
    static_assert(NIA >= 1 && NIB >= 1 && NIA * NWAVES * RPI_A == BR && NIB * NWAVES * RPI_B == BR, "tile / wavefront split");
    constexpr int LPT = NIA + NIB;                          // loads per thread per step
    constexpr int PITCH_A = BA * 2, PITCH_B = BB * 2;
    constexpr int A_BYTES = BR * PITCH_A, B_BYTES = BR * PITCH_B;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int NSTAGE = 4;
    constexpr int WA = BA / NWA, WB = BB / NWB;
    constexpr int AT = WA / 16, BT = WB / 16;
    static_assert(AT % NWB == 0, "bias-gradient tiles per wavefront");
    constexpr int ABT = AT / NWB;
    constexpr uint32_t OOB = 0xfffffff0u;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wa = wave % NWA, wb = wave / NWA;
    const int l15 = lane & 15, lg = lane >> 4;

    const int ntile = gridDim.x;
    const int lid = xcd_remap(blockIdx.y * ntile + blockIdx.x, ntile * gridDim.y);
    const int split = lid / ntile;
    const int tile = lid - split * ntile;
    const int tile_a = tile % p.tiles_a;
    const int tile_b = tile / p.tiles_a;
    const int m_begin = split * p.m_per_split;
    const int m_end = min(p.M, m_begin + p.m_per_split);
    if (m_begin >= m_end) return;

    const __amdgpu_buffer_rsrc_t dy_rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dy), 0, p.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t src_rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src), 0, p.src_bytes, 0x00020000);

    // ---- per-thread DMA state.  Instruction i of wavefront w fills bytes [(i*NWAVES + w)*1024 + lane*16, +16) of a
    // region: row (i*NWAVES + w)*RPI + lane/CPR, slot lane%CPR, i.e. the logical chunk slot ^ (g(row) << 1)
    uint32_t a_off[NIA];
    int a_m[NIA];
    bool a_ok[NIA];
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
        const int row = (i * NWAVES + wave) * RPI_A + lane / CPR_A;
        const int chunk = (lane % CPR_A) ^ (tn_g<CPR_A>(row) << 1);
        const int n = tile_a * BA + chunk * 8;
        a_ok[i] = n < p.Cout;
        a_m[i] = m_begin + row;
        a_off[i] = (uint32_t)((m_begin + row) * p.Cout + n) * 2u;
    }
    const uint32_t a_step = (uint32_t)(BR * p.Cout) * 2u;
    uint32_t b_off[NIB];                 // byte offset, advanced per step
    int b_m[NIB], g_oh[NIB], g_ow[NIB], b_fr[NIB], b_fs[NIB];
    bool b_ok[NIB];
    const int ohw = p.OH * p.OW;
    // a step moves every DMA row by BR output pixels = (d_img, d_oh, d_ow) in mixed radix; a carry out of the column digit
    // takes OW columns back and adds a row, a carry out of the row digit takes OH rows back and adds an image (wave-uniform scalars)
    const int st_dow = p.stride * p.d_ow, st_ow = p.stride * p.OW, st_doh = p.stride * p.d_oh, st_oh = p.stride * p.OH;     // in input pixels
    const uint32_t inc_base = (uint32_t)(((p.d_img * p.H + st_doh) * p.W + st_dow) * p.C) * 2u;
    const uint32_t inc_cy1 = (uint32_t)((p.stride * p.W - st_ow) * p.C) * 2u;
    const uint32_t inc_cy2 = (uint32_t)(((p.H - st_oh) * p.W) * p.C) * 2u;
#pragma unroll
    for (int j = 0; j < NIB; ++j) {
        const int row = (j * NWAVES + wave) * RPI_B + lane / CPR_B;
        const int chunk = (lane % CPR_B) ^ (tn_g<CPR_B>(row) << 1);
        const int kk = tile_b * BB + chunk * 8;
        b_ok[j] = kk < p.Kd;
        const int m = m_begin + row;
        b_m[j] = m;
        if (PLAIN) {
            b_off[j] = (uint32_t)(m * p.C + kk) * 2u;
            g_oh[j] = g_ow[j] = b_fr[j] = b_fs[j] = 0;
        } else {
            const int tap = kk / p.C;
            const int c0 = kk - tap * p.C;
            b_fr[j] = tap / p.S - p.pad;
            b_fs[j] = tap - (tap / p.S) * p.S - p.pad;
            const int img = (int)fdiv((uint32_t)m, p.fd_ohw);
            const int rem = m - img * ohw;
            g_oh[j] = (int)fdiv((uint32_t)rem, p.fd_ow);
            g_ow[j] = rem - g_oh[j] * p.OW;
            const int ih0 = g_oh[j] * p.stride + b_fr[j], iw0 = g_ow[j] * p.stride + b_fs[j];
            // (wraps when ih / iw are negative: such an offset is never used, `ok` is false there)
            b_off[j] = (uint32_t)(((img * p.H + ih0) * p.W + iw0) * p.C + c0) * 2u;
        }
    }
    const uint32_t b_step = (uint32_t)(BR * p.C) * 2u;

    typedef __attribute__((address_space(3))) void lds_void;
    auto issue_step = [&](int stage) __attribute__((always_inline)) {
        char* base = smem + stage * STAGE + wave * 1024;
#pragma unroll
        for (int i = 0; i < NIA; ++i) {
            const bool ok = a_ok[i] & (a_m[i] < m_end);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(dy_rs, (lds_void*)(base + i * NWAVES * 1024), 16, (int)(ok ? a_off[i] : OOB), 0, 0, 0);
            a_off[i] += a_step;
            a_m[i] += BR;
        }
#pragma unroll
        for (int j = 0; j < NIB; ++j) {
            if (PLAIN) {
                const bool ok = b_ok[j] & (b_m[j] < m_end);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(src_rs, (lds_void*)(base + A_BYTES + j * NWAVES * 1024), 16, (int)(ok ? b_off[j] : OOB), 0, 0, 0);
                b_off[j] += b_step;
            } else {
                // (ih, iw) for the bounds test only: one 24-bit multiply-add each (full rate; oh, ow and the stride are far below 2^24)
                const int ih = __mul24(g_oh[j], p.stride) + b_fr[j];
                const int iw = __mul24(g_ow[j], p.stride) + b_fs[j];
                const bool ok = b_ok[j] & (b_m[j] < m_end) & ((unsigned)ih < (unsigned)p.H) & ((unsigned)iw < (unsigned)p.W);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(src_rs, (lds_void*)(base + A_BYTES + j * NWAVES * 1024), 16, (int)(ok ? b_off[j] : OOB), 0, 0, 0);
                int ow = g_ow[j] + p.d_ow;
                const bool c1 = ow >= p.OW;
                ow -= c1 ? p.OW : 0;
                int oh = g_oh[j] + p.d_oh + (c1 ? 1 : 0);
                const bool c2 = oh >= p.OH;
                oh -= c2 ? p.OH : 0;
                g_ow[j] = ow;
                g_oh[j] = oh;
                b_off[j] += inc_base + (c1 ? inc_cy1 : 0u) + (c2 ? inc_cy2 : 0u);
            }
            b_m[j] += BR;
        }
    };

    f32x4 acc[AT][BT];
#pragma unroll
    for (int ai = 0; ai < AT; ++ai)
#pragma unroll
        for (int bi = 0; bi < BT; ++bi) acc[ai][bi] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool do_bias = (p.dbias != nullptr) && (tile_b == 0);
    f32x4 bacc[ABT];
#pragma unroll
    for (int t = 0; t < ABT; ++t) bacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment offsets within a step: rows row0 .. row0+3 (and +4), 16 columns = one 32-byte pair
    const int row0 = lg * 8 + (l15 >> 2);
    int fa_off[AT], fb_off[BT];
#pragma unroll
    for (int ai = 0; ai < AT; ++ai)
        fa_off[ai] = row0 * PITCH_A + (((wa * AT + ai) ^ tn_g<CPR_A>(row0)) << 5) + (l15 & 3) * 8;
#pragma unroll
    for (int bi = 0; bi < BT; ++bi)
        fb_off[bi] = A_BYTES + row0 * PITCH_B + (((wb * BT + bi) ^ tn_g<CPR_B>(row0)) << 5) + (l15 & 3) * 8;

    const u32x4 ones = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
    // The transposed fragment reads are spelled as asm: behind the BUILTIN the compiler waits vmcnt(0) (an LDS access it can
    // see while LDS-DMA writes are pending), which would drain the whole ring every step.  All reads of a step are issued at
    // once; a counted lgkmcnt wait tied to the B fragment of column bi releases that column's MFMAs (LDS returns in order,
    // and the A fragments were issued first).
    auto compute = [&](int stage) __attribute__((always_inline)) {
        const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(smem + stage * STAGE);
        u32x2 alo[AT], ahi[AT], blo[BT], bhi[BT];
#pragma unroll
        for (int ai = 0; ai < AT; ++ai) {
            const uint32_t q = base + (uint32_t)fa_off[ai];
            asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%3"
                         : "=&v"(alo[ai]), "=&v"(ahi[ai]) : "v"(q), "n"(4 * PITCH_A) : "memory");
        }
#pragma unroll
        for (int bi = 0; bi < BT; ++bi) {
            const uint32_t q = base + (uint32_t)fb_off[bi];
            asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%3"
                         : "=&v"(blo[bi]), "=&v"(bhi[bi]) : "v"(q), "n"(4 * PITCH_B) : "memory");
        }
        u32x4 af[AT];
#pragma unroll
        for (int bi = 0; bi < BT; ++bi) {
            asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(blo[bi]), "+v"(bhi[bi]) : "n"(2 * (BT - 1 - bi)) : "memory");
            if (bi == 0) {
#pragma unroll
                for (int ai = 0; ai < AT; ++ai) af[ai] = u32x4{alo[ai][0], alo[ai][1], ahi[ai][0], ahi[ai][1]};
            }
            const u32x4 bf = {blo[bi][0], blo[bi][1], bhi[bi][0], bhi[bi][1]};
#pragma unroll
            for (int ai = 0; ai < AT; ++ai)
                acc[ai][bi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[ai]),
                                                                      __builtin_bit_cast(bf16x8, bf), acc[ai][bi], 0, 0, 0);
        }
        if (do_bias) {
#pragma unroll
            for (int t = 0; t < ABT; ++t)
                bacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[wb * ABT + t]),
                                                                  __builtin_bit_cast(bf16x8, ones), bacc[t], 0, 0, 0);
        }
    };

    const int nt = (m_end - m_begin + BR - 1) / BR;
    int issued = 0;
    for (; issued < NSTAGE - 1 && issued < nt; ++issued) issue_step(issued);
    int st_c = 0;
    int st_i = issued % NSTAGE;
    for (int t = 0; t < nt; ++t) {
        const int ahead = issued - t - 1;          // wave-uniform
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPT) : "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();              // everyone's part of step t landed; step t-1 is fully consumed
        if (issued < nt) {
            issue_step(st_i);
            ++issued;
            st_i = (st_i + 1 == NSTAGE) ? 0 : st_i + 1;
        }
        compute(st_c);
        st_c = (st_c + 1 == NSTAGE) ? 0 : st_c + 1;
    }

    if (do_bias && l15 == 0) {      // every column of bacc holds the row sums: D row -> n = .. + lg*4 + r
#pragma unroll
        for (int t = 0; t < ABT; ++t) {
            const int n0 = tile_a * BA + wa * WA + (wb * ABT + t) * 16 + lg * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n0 + r < p.Cout) saicv::det_add(p.det, p.dbias + n0 + r, (size_t)p.Cout * p.Kd + n0 + r, split, bacc[t][r]);
        }
    }
    // epilogue: D row -> n = .. + lg*4 + r ; D col -> kk = .. + l15
#pragma unroll
    for (int ai = 0; ai < AT; ++ai) {
        const int n0 = tile_a * BA + wa * WA + ai * 16 + lg * 4;
#pragma unroll
        for (int bi = 0; bi < BT; ++bi) {
            const int kk = tile_b * BB + wb * WB + bi * 16 + l15;
            if (kk < p.Kd) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n0 + r < p.Cout)
                        saicv::det_add(p.det, p.dw + (size_t)(n0 + r) * (size_t)p.Kd + kk, (size_t)(n0 + r) * (size_t)p.Kd + kk, split, acc[ai][bi][r]);
            }
        }
    }
}

FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;                       // l = ceil(log2 d)
    f.mul = (uint32_t)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    f.s1 = l < 1 ? l : 1;
    f.s2 = l > 0 ? l - 1 : 0;
    return f;
}

template <typename K>
void allow_lds(K k, size_t smem) {
    // one-time opt-in for > 64 KiB dynamic LDS (idempotent; cheap enough to repeat)
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
}

// Ticket counters of persistent launches: a ring of 64 sets (64 dwords each) per device, zero whenever no kernel is using
// them (the last workgroup of a launch to leave resets its set).  Successive launches take successive sets, so two launches
// overlapping on different streams do not share one; a captured launch keeps the set it was captured with.
unsigned int* nt_ticket_set() {
    static unsigned int* base[16] = {nullptr};
    static unsigned int seq[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    if (!base[dev]) {
        void* q = nullptr;
        if (hipMalloc(&q, 64 * 64 * sizeof(unsigned int)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (hipMemset(q, 0, 64 * 64 * sizeof(unsigned int)) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(q); return nullptr; }
        base[dev] = static_cast<unsigned int*>(q);
    }
    return base[dev] + (size_t)(seq[dev]++ % 64u) * 64;
}

// ---- launches.  Streaming kernel: persistent launches only (bf16 in and out); everything else: one tile per workgroup.
template <int BM_T, int BN_T, int WM_, int WN_, int MODE>
int launch_nt_stream(NTParams& p, hipStream_t st) {
    constexpr size_t smem = nt_stages(BM_T, BN_T) * (size_t)(BM_T * 64 + BN_T * 64) + 16;      // DMA ring + ticket mailbox
    static_assert(nt_can_persist(2, false, BM_T, BN_T, WM_ * WN_), "geometry without a persistent instantiation");
    p.tickets = nt_ticket_set();
    SAICV_REQUIRE(p.tickets != nullptr, "igemm_nt: no device memory for the ticket counters");
    p.grid_x = 256 * nt_blocks_per_cu(BM_T, BN_T);
    const bool plain = p.R == 1 && p.S == 1 && p.pad == 0 && (MODE == 0 || p.stride == 1);
    // the plain-epilogue kernel: no fused operand, rows of whole 16-byte chunks at a 16-byte aligned pitch
    const bool fusedk = p.act_mode != 0 || p.addend != nullptr || p.row_scale != nullptr || p.bs_y != nullptr ||
                        ((p.ldo * 2) & 15) != 0 || (p.Nn & 7) != 0;
    dim3 grid(p.grid_x, 1), block(64 * WM_ * WN_);
#define NT_STREAM_LAUNCH(PL, FK)                                                                   \
    {                                                                                              \
        auto k = igemm_nt_kernel<bf16_t, BM_T, BN_T, WM_, WN_, MODE, false, PL, FK>;               \
        static bool once = (allow_lds(k, 160 * 1024), true);                                       \
        (void)once;                                                                                \
        hipLaunchKernelGGL(k, grid, block, smem, st, p);                                           \
    }
    if (plain && fusedk) NT_STREAM_LAUNCH(true, true)
    else if (plain) NT_STREAM_LAUNCH(true, false)
    else if (fusedk) NT_STREAM_LAUNCH(false, true)
    else NT_STREAM_LAUNCH(false, false)
#undef NT_STREAM_LAUNCH
    return saicv::check_launch("igemm_nt (persistent)");
}

template <typename T, int BM_T, int BN_T, int WM_, int WN_, int MODE, bool OUT_F32, bool PLAIN, int KC = 4>
void launch_nt1_inst(const NTParams& p, size_t smem, hipStream_t st) {
    auto k = igemm_nt1_kernel<T, BM_T, BN_T, WM_, WN_, MODE, OUT_F32, PLAIN, KC>;
    static bool once = (allow_lds(k, 160 * 1024), true);
    (void)once;
    // SAICV_NT1_ROUNDS=r (tuning aid, default 0 = one tile per workgroup): at most r resident rounds of workgroups, each walking the
    // tiles tix, tix + grid, ... -- the launch-invariant part of the set-up (kernel arguments, divisions: 1.1-2.5 us of a 9-13 us
    // ResNet-50 tile, profiles/r05_nt_timeline.md) is then paid once per workgroup instead of once per tile
    int gx = p.nblk;
    static const int rounds = getenv("SAICV_NT1_ROUNDS") ? atoi(getenv("SAICV_NT1_ROUNDS")) : 0;
    if (rounds > 0) {
        const int per_cu = KC == 8 ? 1 : (int)(160 * 1024 / (smem + 1024)) < 1 ? 1 : (int)(160 * 1024 / (smem + 1024));
        const int cap = 256 * (per_cu > 4 ? 4 : per_cu) * rounds;
        if (gx > cap) gx = cap;
    }
    dim3 grid(gx, (MODE == 1 && p.stride > 1) ? p.stride * p.stride : 1), block(64 * WM_ * WN_);
    hipLaunchKernelGGL(k, grid, block, smem, st, p);
}

template <typename T, int BM_T, int BN_T, int WM_, int WN_, int MODE>
int launch_nt1(NTParams& p, bool out_f32, hipStream_t st) {
    constexpr size_t smem_full = nt_stages(BM_T, BN_T) * (size_t)(BM_T * 64 + BN_T * 64);      // DMA ring
    const size_t epi = BM_T * (size_t)(BN_T * ((out_f32 || sizeof(T) == 4) ? 4 : 2) + 16) +
                       2 * WM_ * BN_T * sizeof(float);      // staged output tile + per-wavefront BN column sums
    size_t smem = smem_full;
    if (smem < epi) smem = epi;
#ifdef SAICV_NT_TIMELINE
    if (getenv("SAICV_NT_SOLO") && smem < 84 * 1024) smem = 84 * 1024;      // debug build: one workgroup per CU (what a lone K loop sustains)
#endif
    p.tickets = nullptr;
    p.grid_x = p.nblk;
    // pointwise taps without padding: the source pixel of a row never leaves the image
    const bool plain = p.R == 1 && p.S == 1 && p.pad == 0 && (MODE == 0 || p.stride == 1);
    // 128-byte K slices: chosen by igemm_nt (p.kc8)
    if constexpr (sizeof(T) == 2 && BM_T == 256) {
        if (p.kc8 && plain && !out_f32) {
            constexpr size_t ring = nt_stages_kc8(BM_T, BN_T) * (size_t)(BM_T + BN_T) * 128;
            launch_nt1_inst<T, BM_T, BN_T, WM_, WN_, MODE, false, true, 8>(p, ring < epi ? epi : ring, st);
            return saicv::check_launch("igemm_nt");
        }
    }
    if (out_f32) {
        if (plain) launch_nt1_inst<T, BM_T, BN_T, WM_, WN_, MODE, true, true>(p, smem, st);
        else launch_nt1_inst<T, BM_T, BN_T, WM_, WN_, MODE, true, false>(p, smem, st);
    } else {
        if (plain) launch_nt1_inst<T, BM_T, BN_T, WM_, WN_, MODE, false, true>(p, smem, st);
        else launch_nt1_inst<T, BM_T, BN_T, WM_, WN_, MODE, false, false>(p, smem, st);
    }
    return saicv::check_launch("igemm_nt");
}

// Tile choice shared by the kernel launch and conv_stat_rows(): relative per-flop speed of each geometry x
// wave-quantisation efficiency on 256 CUs x useful fraction of the tile.  The speeds are fitted to a sweep of
// every geometry over the 23 ResNet-50 shapes and the ViT / SAM linear shapes (scripts/gpu_tiles.sh, r01e): two
// co-resident 256 x 128 workgroups hide each other's epilogue (launch, prologue latency, store drain: ~11 us per
// 64 K outputs per CU whatever the geometry), which beats one 256 x 256 workgroup until the K loop is long enough
// (~100 K tiles) for its lower LDS traffic per flop to matter.
struct NTTile { int bm, bn, wm, blocks_per_cu; float speed; float fixed_us, step_us; };   // tile time = fixed + K tiles x step (fit)
const NTTile kTiles[5] = {{256, 256, 2, 1, 0.80f, 11.8f, 0.905f}, {256, 128, 4, 2, 1.00f, 9.95f, 0.97f},
                           {128, 128, 2, 2, 0.60f, 6.1f, 0.60f}, {128, 64, 2, 3, 0.50f, 4.55f, 0.685f},
                           {256, 128, 2, 2, 0.00f, 9.95f, 0.97f}};      // [4]: 256 x 128 on FOUR wavefronts of 128 x 64 (SAICV_NT_TILE=4)

int pick_tile(int M, int Nn, int nkt, bool f32_out_big) {
    if (const char* force = getenv("SAICV_NT_TILE")) {      // tuning aid: force a geometry
        const int t = atoi(force);
        if (t >= 0 && t < 5 && !(f32_out_big && kTiles[t].bm * kTiles[t].bn * 4 > 150 * 1024)) return t;
    }
    int best = 2;
    float best_score = -1.f;
    for (int t = 0; t < 4; ++t) {
        if (f32_out_big && kTiles[t].bm * kTiles[t].bn * 4 > 150 * 1024) continue;   // fp32 epilogue tile must fit LDS
        const NTTile& g = kTiles[t];
        const long tm = (M + g.bm - 1) / g.bm, tn = (Nn + g.bn - 1) / g.bn;
        const float blocks = (float)(tm * tn);
        const float slots = 256.f * g.blocks_per_cu;
        const float rounds = blocks / slots;
        const float quant = rounds / (float)(long)(rounds + 0.999999f);
        const float useful = ((float)M / (tm * g.bm)) * ((float)Nn / (tn * g.bn));
        float speed = g.speed;
        if (t == 0) speed = nkt >= 100 ? 1.10f : nkt >= 48 ? 0.90f : 0.80f;
        const float score = speed * quant * useful;
        if (score > best_score) { best_score = score; best = t; }
    }
    return best;
}

// What a launch will do, as a pure function of the problem (conv_stat_rows() must size the statistics buffer before the
// launch): geometry, and whether the launch is PERSISTENT -- the streaming kernel, one partial-statistics row per (tile row,
// wavefront row) -- or one tile per workgroup (one row per tile row).  Persistent: bf16 in and out, a geometry with a
// persistent instantiation, more tiles than resident workgroups, a K loop longer than the DMA ring, not a stride > 1 data
// gradient (its parity classes have unequal, possibly empty K loops).  SAICV_NT_PERSIST=0 turns it off.
struct NTPlan { int tile; bool persist; };
NTPlan nt_plan(int dtype, int mode, int stride, int M_tile, int Nn, int nkt, bool f32o) {
    NTPlan pl;
    pl.tile = pick_tile(M_tile, Nn, nkt, f32o);
    const NTTile& g = kTiles[pl.tile];
    // OFF by default (r03 measurement, profiles/r03_streaming_kernel.md): the streaming kernel is 3-14 % ahead of the one-tile
    // kernel on isolated ViT-B GEMMs with wide outputs, level elsewhere -- and BEHIND inside the models on the same box
    // (ResNet-50 23.8 vs 22.6 ms per step with every eligible layer persistent, ViT-B 44.9 vs 44.3 ms with the rule below):
    // a ViT-B launch is ~7 tiles per workgroup, a ResNet-50 launch 1-3 rounds -- the ramp of a staggered start costs what
    // the interleaved store bursts gain.  SAICV_NT_PERSIST=1 turns the rule below on, with SAICV_NT_TILE every eligible launch.
    const char* pe = getenv("SAICV_NT_PERSIST");                       // (read per call: a tuning sweep flips it in-process)
    const int persist_env = pe ? atoi(pe) : 0;
    const long nblk = (long)((Nn + g.bn - 1) / g.bn) * ((M_tile + g.bm - 1) / g.bm);
    const int nwaves = g.wm * (pl.tile == 0 ? 4 : 2);
    const bool can = persist_env && dtype == SAICV_DTYPE_BF16 && !f32o && !(mode == 1 && stride > 1);
    pl.persist = can && getenv("SAICV_NT_TILE") != nullptr && nt_can_persist(2, false, g.bm, g.bn, nwaves) &&
                 nblk > 256L * nt_blocks_per_cu(g.bm, g.bn) && nkt > nt_stages(g.bm, g.bn);
    if (pl.tile == 4 && !pl.persist) pl.tile = 1;      // the four-wavefront 256 x 128 geometry exists in the streaming kernel only
    if (!getenv("SAICV_NT_TILE") && can && !pl.persist) {
        // Where the streaming kernel (256 x 256 tiles, staggered start) measured ahead of the tile picker's one-tile choice
        // (profiles/r03_nt_sweep_*.jsonl: M = 50 432 rows, +9..14 % on N = 2304 / 3072 outputs with K = 768 and on long
        // reductions; level or behind on N = 768 and on every ResNet-50 shape, whose launches are one to three rounds):
        // at least four rounds of tiles per CU, nearly full last round, and wide outputs with a short K loop or a long K loop.
        const long t0 = (long)((Nn + 255) / 256) * ((M_tile + 255) / 256);
        const float rounds = (float)t0 / 256.f;
        const float quant = rounds / (float)(long)(rounds + 0.999999f);
        if (t0 >= 4 * 256 && quant >= 0.85f && nkt > nt_stages(256, 256) && ((Nn >= 2048 && nkt <= 32) || nkt >= 64)) {
            pl.tile = 0;
            pl.persist = true;
        }
    }
    return pl;
}

template <typename T, int BA, int BB, int NWA = 2, int NWB = 2>
int launch_tn(const TNParams& p, int splits, hipStream_t st) {
    constexpr int EPC = ElemTraits<T>::EPC;
    constexpr int BR = 8 * EPC;
    constexpr size_t smem = 2 * (size_t)BR * ((BA + 16) + (BB + 16)) * sizeof(T);
    dim3 grid(p.tiles_a * p.tiles_b, splits), block(64 * NWA * NWB);
    auto k = igemm_tn_kernel<T, BA, BB, NWA, NWB>;
    static bool once = (allow_lds(k, smem), true);
    (void)once;
    hipLaunchKernelGGL(k, grid, block, smem, st, p);
    return saicv::check_launch("igemm_tn");
}

template <int BA, int BB, int NWA = 2, int NWB = 2>
int launch_tn_dma(const TNParams& p, int splits, bool plain, hipStream_t st) {
    constexpr size_t smem = 4 * (size_t)32 * (BA + BB) * 2;
    dim3 grid(p.tiles_a * p.tiles_b, splits), block(64 * NWA * NWB);
    if (plain) {
        auto k = igemm_tn_dma_kernel<BA, BB, NWA, NWB, true>;
        static bool once = (allow_lds(k, smem), true);
        (void)once;
        hipLaunchKernelGGL(k, grid, block, smem, st, p);
    } else {
        auto k = igemm_tn_dma_kernel<BA, BB, NWA, NWB, false>;
        static bool once = (allow_lds(k, smem), true);
        (void)once;
        hipLaunchKernelGGL(k, grid, block, smem, st, p);
    }
    return saicv::check_launch("igemm_tn");
}

}  // namespace

namespace saicv {

// rows of BN partial statistics written by the forward kernel: one per row of workgroups (per row of wavefronts when the
// launch is persistent)
int conv_stat_rows(int M, int Nn, int Kd, int dtype, int form) {
    if (form) {                                            // the streaming kernels' row per workgroup (csrc/pwstream.hip)
        const int pw = form == 1 ? pw_stream_blocks(dtype, M, Nn, Kd, false) : pw3_stream_blocks(dtype, M, Nn, Kd);
        if (pw > 0) return pw;
    }
    const int bk = dtype == SAICV_DTYPE_BF16 ? 32 : 16;
    const NTPlan pl = nt_plan(dtype, 0, 1, M, Nn, (Kd + bk - 1) / bk, dtype == SAICV_DTYPE_F32);
    const NTTile& g = kTiles[pl.tile];
    return ((M + g.bm - 1) / g.bm) * (pl.persist ? g.wm : 1);
}

// partial rows the data gradient writes with EpiExtra::bs_*: (rows of tiles of the largest parity class) x classes
int conv_bwd_stat_rows(int M, int OH, int OW, int Nn, int Kd, int stride, int dtype, int form) {
    if (form && stride == 1) {
        const int pw = form == 1 ? pw_stream_blocks(dtype, M, Nn, Kd, true) : pw3_stream_blocks(dtype, M, Nn, Kd);
        if (pw > 0) return pw;
    }
    const int bk = dtype == SAICV_DTYPE_BF16 ? 32 : 16;
    int M_tile = M;
    if (stride > 1) M_tile = (M / (OH * OW)) * ((OH + stride - 1) / stride) * ((OW + stride - 1) / stride);
    const NTPlan pl = nt_plan(dtype, 1, stride, M_tile, Nn, (Kd + bk - 1) / bk, dtype == SAICV_DTYPE_F32);
    const NTTile& g = kTiles[pl.tile];
    return ((M_tile + g.bm - 1) / g.bm) * stride * stride * (pl.persist ? g.wm : 1);
}

// r05, measured and NOT kept: cutting a badly quantised linear GEMM (ViT-B's 768-wide outputs: 1 182 tiles of 256 x 128 on 512 slots
// = 2.31 rounds of work in 3 rounds of time) into whole rounds of 256 x 128 tiles + one round of 128 x 128 tiles.  Correct (torch fp32
// parity on the row-addressed fused operands), but the ViT-B step went 39.44 -> 40.11 ms on the same box: the second launch pays its
// own ramp, and at the socket power limit the idle CUs of the third round are not lost time (profiles/r05_nt_experiments.md).
int igemm_nt(int dtype, int mode, const void* src, const void* wgt, void* out, const float* bias,
             float* stat_sum, float* stat_sq, int H, int W, int C, int OH, int OW, int R, int S,
             int stride, int pad, int M, int Nn, int Kd, int ldo, int out_f32, hipStream_t st,
             const EpiExtra* ex) {
    const int epc = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    SAICV_REQUIRE(C % epc == 0, "igemm_nt: C=%d must be a multiple of %d", C, epc);
    SAICV_REQUIRE(Kd % epc == 0, "igemm_nt: Kd=%d must be a multiple of %d", Kd, epc);
    SAICV_REQUIRE(M > 0 && Nn > 0 && Kd > 0, "igemm_nt: empty problem");
    SAICV_REQUIRE(!(bias && stat_sum), "igemm_nt: bias and BN statistics are exclusive");
    NTParams p;
    p.src = src; p.wgt = wgt; p.out = out; p.bias = bias; p.stat_sum = stat_sum; p.stat_sq = stat_sq;
    p.addend = ex ? ex->addend : nullptr;
    p.row_scale = ex ? ex->row_scale : nullptr;
    p.rows_per_scale = ex ? ex->rows_per_scale : 1;
    p.out2 = ex ? ex->out2 : nullptr;
    p.act_mode = ex ? ex->act_mode : 0;
    static const int lean_gelu_env = getenv("SAICV_GELU_EPI") ? atoi(getenv("SAICV_GELU_EPI")) : 1;
    p.lean_gelu = lean_gelu_env;
    p.addend_gate = ex ? ex->addend_gate : nullptr;
    p.bs_y = ex ? ex->bs_y : nullptr;
    p.bs_mask = ex ? ex->bs_mask : nullptr;
    p.bs_mean = ex ? ex->bs_mean : nullptr;
    p.bs_invstd = ex ? ex->bs_invstd : nullptr;
    p.bs_g = ex ? ex->bs_g : nullptr;
    p.bs_gx = ex ? ex->bs_gx : nullptr;
    p.bs_rows = 0;
    p.stat_atomic_rows = ex ? ex->stat_atomic_rows : 0;
    p.tickets = nullptr;
#ifdef SAICV_NT_TIMELINE
    p.timeline = g_nt_timeline;
#endif
    SAICV_REQUIRE(p.stat_atomic_rows >= 0 && p.stat_atomic_rows <= 64, "igemm_nt: stat_atomic_rows=%d outside [0, 64]", p.stat_atomic_rows);
    if (p.addend_gate || p.bs_y) {
        const int osz1 = (out_f32 || dtype == SAICV_DTYPE_F32) ? 4 : 2;
        SAICV_REQUIRE(mode == 1 && p.act_mode == 0 && (!out_f32 || dtype == SAICV_DTYPE_F32), "igemm_nt: gated shortcut / BatchNorm-backward sums belong to the data gradient");
        SAICV_REQUIRE((ldo * osz1) % 16 == 0 && Nn % (16 / osz1) == 0 && ldo == Nn,
                      "igemm_nt: gated shortcut / BatchNorm-backward sums need dense rows of whole 16-byte chunks (N=%d)", Nn);
        SAICV_REQUIRE(!p.addend_gate || p.addend, "igemm_nt: a gate without an addend");
        SAICV_REQUIRE(!p.bs_y || (p.bs_mean && p.bs_invstd && p.bs_g && p.bs_gx),
                      "igemm_nt: BatchNorm-backward sums need mean, invstd and both partial buffers");
    }
    if (p.act_mode) {
        const int osz0 = (out_f32 || dtype == SAICV_DTYPE_F32) ? 4 : 2;
        SAICV_REQUIRE((ldo * osz0) % 16 == 0 && Nn % (16 / osz0) == 0,
                      "igemm_nt: the fused GELU epilogues need 16-byte aligned rows (N=%d)", Nn);
        SAICV_REQUIRE(p.act_mode >= 1 && p.act_mode <= 4, "igemm_nt: act_mode %d", p.act_mode);
        SAICV_REQUIRE((p.act_mode & 1) ? p.out2 != nullptr : p.addend != nullptr, "igemm_nt: fused GELU operand missing");
        SAICV_REQUIRE(stat_sum == nullptr && p.row_scale == nullptr, "igemm_nt: fused GELU excludes BN statistics / row scale");
    }
    if (p.addend || p.row_scale) {
        const int osz = (out_f32 || dtype == SAICV_DTYPE_F32) ? 4 : 2;
        SAICV_REQUIRE((ldo * osz) % 16 == 0, "igemm_nt: fused residual needs a 16-byte aligned leading dimension");
        SAICV_REQUIRE(p.rows_per_scale >= 1, "igemm_nt: rows_per_scale must be >= 1");
    }
    // Small-K x small-N pointwise products over many rows (ResNet stage 1-2 1 x 1 convolutions and their data gradients): the
    // weight-resident streaming kernel of pwstream.hip -- no tiles, no workgroup barriers, one partial row per workgroup.
    if (dtype == SAICV_DTYPE_BF16 && !out_f32 && R == 1 && S == 1 && pad == 0 && stride == 1 && ldo == Nn && !bias && !p.act_mode &&
        !p.row_scale && !p.out2 && !(stat_sum && (p.addend || p.bs_y))) {
        static const long min_mb = getenv("SAICV_NT_STREAM_MIN_MB") ? atol(getenv("SAICV_NT_STREAM_MIN_MB")) : 0;
        const int so = (size_t)M * Nn * 2 >= (size_t)min_mb * 1024 * 1024 ? 1 : 0;
        const int rc = pw_stream(M, Nn, Kd, src, wgt, out, stat_sum, stat_sq, p.stat_atomic_rows, ex, so, st);
        if (rc != 0) return rc < 0 ? rc : 0;
    }
    // ... and the 3 x 3 / stride 1 / padding 1 convolution 64 -> 64 of the same stage and its data gradient, as the same stream over nine taps
    if (dtype == SAICV_DTYPE_BF16 && !out_f32 && R == 3 && S == 3 && pad == 1 && stride == 1 && C == 64 && Nn == 64 && Kd == 576 && ldo == Nn &&
        H == OH && W == OW && !bias && !p.act_mode && !p.row_scale && !p.out2 && !(stat_sum && (p.addend || p.bs_y))) {
        static const long min_mb3 = getenv("SAICV_NT_STREAM_MIN_MB") ? atol(getenv("SAICV_NT_STREAM_MIN_MB")) : 0;
        const int so = (size_t)M * Nn * 2 >= (size_t)min_mb3 * 1024 * 1024 ? 1 : 0;
        const int rc = pw3_stream(mode, M, H, W, src, wgt, out, stat_sum, stat_sq, p.stat_atomic_rows, ex, so, st);
        if (rc != 0) return rc < 0 ? rc : 0;
    }
    p.H = H; p.W = W; p.C = C; p.OH = OH; p.OW = OW; p.R = R; p.S = S; p.stride = stride; p.pad = pad;
    p.M = M; p.Nn = Nn; p.Kd = Kd; p.ldo = ldo;
    const size_t esz = dtype == SAICV_DTYPE_BF16 ? 2 : 4;
    const size_t src_bytes = (size_t)(M / (OH * OW)) * H * W * C * esz;
    const size_t wgt_bytes = (size_t)Nn * Kd * esz;
    SAICV_REQUIRE(src_bytes < 0xfffffff0ull && wgt_bytes < 0xfffffff0ull,
                  "igemm_nt: operand larger than 4 GiB (buffer addressing)");
    p.src_bytes = (uint32_t)src_bytes;
    p.wgt_bytes = (uint32_t)wgt_bytes;
    p.fd_ohw = make_fastdiv((uint32_t)(OH * OW));
    p.fd_ow = make_fastdiv((uint32_t)OW);
    const bool f32o = out_f32 != 0 || dtype == SAICV_DTYPE_F32;
    int M_tile = M;
    if (mode == 1 && stride > 1) {
        const int nimg = M / (OH * OW);
        M_tile = nimg * ((OH + stride - 1) / stride) * ((OW + stride - 1) / stride);   // largest parity class
    }
    const int nkt_host = (Kd + 4 * epc - 1) / (4 * epc);      // K tiles (stride > 1 data-gradient classes run fewer)
    NTPlan pl = nt_plan(dtype, mode, stride, M_tile, Nn, nkt_host, f32o);
    {
        // SAICV_NT_PERSIST=2: the streaming kernel only for launches with a plain epilogue (bias at most) -- the fused modes run
        // in its rolled per-wavefront loops.  Statistics launches keep the plan conv_stat_rows() sized their buffers with.
        const char* pe = getenv("SAICV_NT_PERSIST");
        const bool fused = p.act_mode || p.addend || p.row_scale || p.out2 || p.addend_gate || p.bs_y;
        if (pl.persist && pe && atoi(pe) == 2 && fused && !stat_sum) {
            pl.persist = false;
            pl.tile = pick_tile(M_tile, Nn, nkt_host, f32o);
            if (pl.tile == 4) pl.tile = 1;
        }
    }
    int t = pl.tile;
    // 128-byte K slices (igemm_nt1_kernel, KC = 8) on the 256 x 256 tile: one workgroup per CU, so only where the main loop is
    // long against the exposed prologue / epilogue and the launch fills the CUs several times.  SAICV_NT_KC8: 0 never,
    // 1 every eligible launch of a 256-row tile, 2 (default) wide pointwise GEMMs without a fused activation, 3 the same with
    // them -- measured on the ViT-B shapes (profiles/r03_lds_fill_and_kc8.md): +10...13 % for N >= 2048, K = 768; -8 % for
    // N = K = 768; the GELU-fused launches (two output tensors / one more input in the epilogue) lose 10 % with one workgroup
    // per CU: ViT-B step 42.57 ms off, 42.95 with them, 42.31 without.
    const char* ke = getenv("SAICV_NT_KC8");             // read per call
    const int kc8_mode = ke ? atoi(ke) : 2;
    {
        // streaming output stores from SAICV_NT_STREAM_MIN_MB MiB of output on (0: always; a huge value: never)
        static const long min_mb = getenv("SAICV_NT_STREAM_MIN_MB") ? atol(getenv("SAICV_NT_STREAM_MIN_MB")) : 0;
        const size_t out_bytes = (size_t)M * Nn * ((out_f32 || dtype == SAICV_DTYPE_F32) ? 4 : 2);
        p.stream_out = out_bytes >= (size_t)min_mb * 1024 * 1024 ? 1 : 0;
    }
    p.kc8 = 0;
    {
        const bool pointwise = R == 1 && S == 1 && pad == 0 && (mode == 0 || stride == 1);
        const bool ok = dtype == SAICV_DTYPE_BF16 && !f32o && pointwise && !pl.persist && kTiles[t].bm == 256 && Kd >= 256;
        if (ok && kc8_mode == 1) p.kc8 = 1;
        const bool rule_ok = (kc8_mode == 2 && p.act_mode == 0) || kc8_mode == 3;
        if (ok && rule_ok && Nn >= 2048 && Nn % 256 == 0 && Kd >= 512 && (long)((M_tile + 255) / 256) * (Nn / 256) >= 4 * 256) {
            p.kc8 = 1;
            t = 0;
        }
    }
    // (r03's staged-range main loop for 3 x 3 / stride 1 convolutions -- one contiguous input range per channel slice instead of nine
    // gathered taps, LDS-fill traffic / 2.3 -- measured level with the gather on every ResNet-50 layer (profiles/r03_lds_fill_and_kc8.md
    // section 7) and was removed in r05.)
    const NTTile& g = kTiles[t];
    p.tiles_n = (Nn + g.bn - 1) / g.bn;
    p.nblk = p.tiles_n * ((M_tile + g.bm - 1) / g.bm);
    p.bs_rows = (M_tile + g.bm - 1) / g.bm;
    p.stagger_phases = 0;
    p.stagger_sleeps = 0;
    if (pl.persist) {
        // start phases: SAICV_NT_STAGGER = number of phase groups (default 8; 0 or 1 = none), spread over one tile period
        // (estimated from the fitted step time of the geometry; SAICV_NT_STAGGER_US overrides the period)
        const char* sp = getenv("SAICV_NT_STAGGER");
        const char* su = getenv("SAICV_NT_STAGGER_US");
        const int phases = sp ? atoi(sp) : 8;
        if (phases > 1) {
            const float period_us = su ? (float)atof(su) : (float)nkt_host * g.step_us * (g.blocks_per_cu > 1 ? 1.f : 1.f);
            p.stagger_phases = phases;
            p.stagger_sleeps = (int)(period_us / (float)phases / 0.5f + 0.5f);
        }
        if (mode == 0) {
            switch (t) {
                case 0: return launch_nt_stream<256, 256, 2, 4, 0>(p, st);
                case 2: return launch_nt_stream<128, 128, 2, 2, 0>(p, st);
                case 3: return launch_nt_stream<128, 64, 2, 2, 0>(p, st);
                default: return launch_nt_stream<256, 128, 2, 2, 0>(p, st);
            }
        } else {
            switch (t) {
                case 0: return launch_nt_stream<256, 256, 2, 4, 1>(p, st);
                case 2: return launch_nt_stream<128, 128, 2, 2, 1>(p, st);
                case 3: return launch_nt_stream<128, 64, 2, 2, 1>(p, st);
                default: return launch_nt_stream<256, 128, 2, 2, 1>(p, st);
            }
        }
    }
// -DSAICV_NT_T0_WN=2: the 256 x 256 tile on FOUR wavefronts of 128 x 128 (with -DSAICV_NT_MFMA32: the probe's cheapest form per flop)
#ifndef SAICV_NT_T0_WN
#define SAICV_NT_T0_WN 4
#endif
#define NT_T0_WN SAICV_NT_T0_WN
#define NT_DISPATCH(TT, MODE_)                                                              \
    switch (t) {                                                                            \
        case 0: return launch_nt1<TT, 256, 256, 2, NT_T0_WN, MODE_>(p, f32o, st);           \
        case 1: return launch_nt1<TT, 256, 128, 4, 2, MODE_>(p, f32o, st);                  \
        case 2: return launch_nt1<TT, 128, 128, 2, 2, MODE_>(p, f32o, st);                  \
        default: return launch_nt1<TT, 128, 64, 2, 2, MODE_>(p, f32o, st);                  \
    }
    if (dtype == SAICV_DTYPE_BF16) {
        if (mode == 0) { NT_DISPATCH(bf16_t, 0) } else { NT_DISPATCH(bf16_t, 1) }
    } else {
        if (mode == 0) { NT_DISPATCH(float, 0) } else { NT_DISPATCH(float, 1) }
    }
#undef NT_DISPATCH
}

static int igemm_tn_launch(int dtype, TNParams& p, int splits, bool big, int ba, int bb, int BR, hipStream_t st);

int igemm_tn(int dtype, const void* dy, const void* src, float* dw, int H, int W, int C, int OH,
             int OW, int R, int S, int stride, int pad, int M, int Cout, int Kd, hipStream_t st, float* dbias) {
    const int epc = dtype == SAICV_DTYPE_BF16 ? 8 : 4;
    SAICV_REQUIRE(C % epc == 0 && Cout % epc == 0, "igemm_tn: C=%d, Cout=%d must be multiples of %d", C, Cout, epc);
    SAICV_REQUIRE(M > 0 && Cout > 0 && Kd > 0, "igemm_tn: empty problem");
    TNParams p;
    p.dy = dy; p.src = src; p.dw = dw; p.dbias = dbias; p.H = H; p.W = W; p.C = C; p.OH = OH; p.OW = OW;
    p.R = R; p.S = S; p.stride = stride; p.pad = pad; p.M = M; p.Cout = Cout; p.Kd = Kd;
    const size_t esz = dtype == SAICV_DTYPE_BF16 ? 2 : 4;
    const size_t dy_bytes = (size_t)M * Cout * esz;
    const size_t src_bytes = (size_t)(M / (OH * OW)) * H * W * C * esz;
    SAICV_REQUIRE(dy_bytes < 0xfffffff0ull && src_bytes < 0xfffffff0ull,
                  "igemm_tn: operand larger than 4 GiB (buffer addressing)");
    p.dy_bytes = (uint32_t)dy_bytes;
    p.src_bytes = (uint32_t)src_bytes;
    p.fd_ohw = make_fastdiv((uint32_t)(OH * OW));
    p.fd_ow = make_fastdiv((uint32_t)OW);
    const int BR = 8 * epc;
    // 256 x 256 tiles (8 wavefronts of 128 x 64) halve both the HBM/L2 traffic and the LDS fragment reads per
    // MFMA of the 128 x 128 geometry; used when both output dimensions fill them
    // ... when both output dimensions fill them and every workgroup keeps >= 48 reduction steps (below that the
    // 256 KiB of fp32 atomics per workgroup outweigh the gain: the small-M ResNet stages stay on 128 x 128)
    static const int force_big = getenv("SAICV_TN_BIG") ? atoi(getenv("SAICV_TN_BIG")) : -1;
    const int big_tiles = ((Cout + 255) / 256) * ((Kd + 255) / 256);
    const int big_steps = ((M + BR - 1) / BR) / (256 / big_tiles > 0 ? 256 / big_tiles : 1);
    const bool big = force_big >= 0 ? (force_big != 0 && Cout >= 128 && Kd >= 128)
                                    : (Cout % 256 == 0 && Kd >= 256 && (Kd % 256 == 0 || Kd >= 1024) && big_steps >= 48);
    const int ba = big ? 256 : Cout <= 64 ? 64 : 128;
    const int bb = big ? 256 : Kd <= 64 ? 64 : 128;
    p.tiles_a = (Cout + ba - 1) / ba;
    p.tiles_b = (Kd + bb - 1) / bb;
    const int tiles = p.tiles_a * p.tiles_b;
    // split the pixel reduction so that ~4 workgroups per CU are in flight
    const int total_rt = (M + BR - 1) / BR;
    // 128-wide tiles: 2 workgroups per CU are resident (73 KiB LDS each); 256-wide: one (139 KiB)
    // rounded DOWN: one full round of resident workgroups beats a second, nearly empty one
    // The split count is sized for ALL resident slots when this GPU runs nothing else, and for 85 % of them once the process has
    // created an RCCL communicator over more than one rank (g_saicv_comm_world, comm.hip): the one-round design has no slack -- under
    // data-parallel training RCCL's all-reduce kernels hold CU slots (and LDS) during backward, and every workgroup that cannot
    // start with the others costs this kernel a whole second round.  Measured on one GPU (r05, same box, twice each): 100 against
    // 85 is ViT-B 40.02 -> 39.52 ms (weight gradients 10.0 -> 9.4 ms), ResNet-50 21.00 -> 20.95 ms; 70 is ResNet-50 21.60 ms.
    // SAICV_TN_SLOTS_PCT overrides (read per call; e.g. 85 for a torch.distributed process group without the native communicator).
    const char* slots_env = getenv("SAICV_TN_SLOTS_PCT");
    const int slots_pct = slots_env ? atoi(slots_env) : (g_saicv_comm_world > 1 ? 85 : 100);
    int splits = ((big ? 256 : 512) * slots_pct / 100) / tiles;
    if (splits > total_rt) splits = total_rt;
    if (splits < 1) splits = 1;
    int rt_per = (total_rt + splits - 1) / splits;
    splits = (total_rt + rt_per - 1) / rt_per;
    p.m_per_split = rt_per * BR;
    // deterministic mode (det.h): the splits park their tiles (and bias sums) side by side, one launch folds them in split order
    // (not zeroed: every split owns at least one row tile -- (splits - 1) * rt_per < total_rt above -- and writes its whole tile and bias sums)
    DetParts det;
    if (det.begin(st, splits, (size_t)Cout * Kd + (dbias ? Cout : 0), "igemm_tn", /*zero=*/false)) return -1;
    p.det = det.sink();
    const int rc = igemm_tn_launch(dtype, p, splits, big, ba, bb, BR, st);
    if (rc) return rc;
    if (det.fold(dw, 0, (size_t)Cout * Kd, /*wide=*/true)) return -1;
    if (dbias && det.fold(dbias, (size_t)Cout * Kd, (size_t)Cout)) return -1;
    return 0;
}

static int igemm_tn_launch(int dtype, TNParams& p, int splits, bool big, int ba, int bb, int BR, hipStream_t st) {
    const int OH = p.OH, OW = p.OW, R = p.R, S = p.S, stride = p.stride, pad = p.pad, H = p.H, W = p.W;
    const int ohw = OH * OW;
    p.d_img = BR / ohw;
    const int rem = BR - p.d_img * ohw;
    p.d_oh = rem / OW;
    p.d_ow = rem - p.d_oh * OW;
    // bf16: the LDS-DMA ring kernel (32-row steps; SAICV_TN_DMA=0 selects the register-staged kernel for A/B runs)
    const char* de = getenv("SAICV_TN_DMA");             // read per call
    const int use_dma = de ? atoi(de) : 1;
    if (use_dma && dtype == SAICV_DTYPE_BF16) {
        const int br = 32;
        p.d_img = br / ohw;
        const int rem32 = br - p.d_img * ohw;
        p.d_oh = rem32 / OW;
        p.d_ow = rem32 - p.d_oh * OW;
        const bool plain = R == 1 && S == 1 && stride == 1 && pad == 0 && H == OH && W == OW;
        if (big) return launch_tn_dma<256, 256, 2, 4>(p, splits, plain, st);
        if (ba == 64 && bb == 64) return launch_tn_dma<64, 64>(p, splits, plain, st);
        if (ba == 64) return launch_tn_dma<64, 128>(p, splits, plain, st);
        if (bb == 64) return launch_tn_dma<128, 64>(p, splits, plain, st);
        return launch_tn_dma<128, 128>(p, splits, plain, st);
    }
    if (big) {
        if (dtype == SAICV_DTYPE_BF16) return launch_tn<bf16_t, 256, 256, 2, 4>(p, splits, st);
        return launch_tn<float, 256, 256, 2, 4>(p, splits, st);
    }
    if (dtype == SAICV_DTYPE_BF16) {
        if (ba == 64 && bb == 64) return launch_tn<bf16_t, 64, 64>(p, splits, st);
        if (ba == 64) return launch_tn<bf16_t, 64, 128>(p, splits, st);
        if (bb == 64) return launch_tn<bf16_t, 128, 64>(p, splits, st);
        return launch_tn<bf16_t, 128, 128>(p, splits, st);
    } else {
        if (ba == 64 && bb == 64) return launch_tn<float, 64, 64>(p, splits, st);
        if (ba == 64) return launch_tn<float, 64, 128>(p, splits, st);
        if (bb == 64) return launch_tn<float, 128, 64>(p, splits, st);
        return launch_tn<float, 128, 128>(p, splits, st);
    }
}

}  // namespace saicv

#ifdef SAICV_NT_TIMELINE
// debug build only (scripts/nt_timeline.py): 16 u64 per workgroup of the NEXT igemm_nt launches (one tile per workgroup kernels)
extern "C" void saicv_debug_nt_timeline(unsigned long long* buf) { g_nt_timeline = buf; }
#endif
