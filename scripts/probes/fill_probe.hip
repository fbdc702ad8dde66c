// How fast can a CU fill LDS from L2 / HBM?  The GEMM kernels of csrc/igemm.hip all sit at 8-9 TB/s of LDS-fill traffic
// (tiles x (BM + BN) x K x 2 bytes / kernel time), whatever their MFMA schedule: this probe measures the fill path alone.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/fill_probe.hip -o /tmp/fill_probe && /tmp/fill_probe
// A workgroup of W wavefronts runs the main loop of the GEMM kernels without the matrix work: a ring of 4 LDS slots,
// `buffer_load_dwordx4 ... lds` (1 KiB per wave instruction) NI instructions per wavefront and step, counted vmcnt wait,
// one s_barrier per step.  Access patterns (what the 64 lanes of one instruction fetch):
//   nt64  : 16 rows x 64 bytes, rows `pitch` bytes apart       (igemm_nt: 64-byte K slices of 16 tile rows)
//   nt128 : 8 rows x 128 bytes                                  (128-byte K slices)
//   row512: 2 rows x 512 bytes                                  (igemm_tn: 256-column rows)
//   lin   : 1024 contiguous bytes
// Footprints: `hot` = every workgroup of an XCD re-reads one 2 MiB window (L2 hits), `share4` = groups of four workgroups of
// an XCD walk one stream together (what neighbouring GEMM tiles do with an operand), `cold` = every workgroup streams its
// own range (HBM).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct Args {
    const void* src;
    uint32_t bytes;       // size of the source buffer (multiple of 1 MiB)
    int steps;            // ring steps per workgroup
    int pattern;          // 0 nt64, 1 nt128, 2 row512, 3 lin
    int pitch;            // row pitch in bytes for the row patterns
    int footprint;        // 0 hot, 1 share4, 2 cold
    int ni;               // DMA instructions per wavefront and step
};

template <int W>
__global__ __launch_bounds__(64 * W) void fill_kernel(const Args a, unsigned long long* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.src), 0, a.bytes, 0x00020000);
    const int NI = a.ni;
    const uint32_t stage_bytes = (uint32_t)NI * W * 1024u;
    // lane offset inside one instruction's footprint
    uint32_t lane_off;
    uint32_t inst_span;                     // bytes of address space one instruction covers
    if (a.pattern == 0) { lane_off = (uint32_t)(lane >> 2) * a.pitch + (lane & 3) * 16; inst_span = 16u * a.pitch; }
    else if (a.pattern == 1) { lane_off = (uint32_t)(lane >> 3) * a.pitch + (lane & 7) * 16; inst_span = 8u * a.pitch; }
    else if (a.pattern == 2) { lane_off = (uint32_t)(lane >> 5) * a.pitch + (lane & 31) * 16; inst_span = 2u * a.pitch; }
    else { lane_off = lane * 16; inst_span = 1024; }
    // base of this workgroup's walk
    const uint32_t xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    uint32_t base, window;
    if (a.footprint == 0) { window = 2u << 20; base = xcd * window; }
    else if (a.footprint == 1) {
        const uint32_t gpx = (gridDim.x / 8 + 3) / 4;          // groups of four per XCD
        window = (a.bytes / (8 * gpx)) & ~0xfffffu;
        base = (xcd * gpx + idx / 4) * window;
    } else { window = (a.bytes / gridDim.x) & ~0xfffffu; base = blockIdx.x * window; }
    uint32_t pos = 0;                       // walk position inside the window (all workgroups of an XCD in step for 0 / 1)
    auto issue = [&](int slot) {
        char* dst = smem + slot * stage_bytes + wave * 1024;
        for (int i = 0; i < NI; ++i) {
            // row patterns: consecutive instructions walk along the K axis (next 64 / 128 / 512 byte segment of the same rows)
            uint32_t seg = a.pattern == 0 ? 64u : a.pattern == 1 ? 128u : a.pattern == 2 ? 512u : 1024u;
            uint32_t off = base + (pos % window);
            uint32_t o = off + (uint32_t)(i * W + wave) * (a.pattern == 3 ? 1024u : inst_span) + lane_off;
            (void)seg;
            if (o + 16 > a.bytes) o = (o % (a.bytes - 4096)) & ~15u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(dst + i * W * 1024), 16, (int)o, 0, 0, 0);
        }
        // next step: row patterns advance along K by one segment; after pitch bytes move to the next block of rows
        if (a.pattern == 3) pos += stage_bytes;
        else {
            const uint32_t seg = a.pattern == 0 ? 64u : a.pattern == 1 ? 128u : 512u;
            pos += seg;
            if ((pos % a.pitch) == 0) pos += (uint32_t)NI * W * inst_span - a.pitch;
        }
    };
    int issued = 0;
    for (; issued < 3 && issued < a.steps; ++issued) issue(issued);
    int slot = issued & 3;
    for (int t = 0; t < a.steps; ++t) {
        const int ahead = issued - t - 1;
        if (ahead >= 2) {
            // counted wait: the two newer steps stay in flight (NI is a runtime value: pick the immediate)
            switch (NI) {
                case 1: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
                case 4: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                case 6: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
            }
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (issued < a.steps) { issue(slot); ++issued; slot = (slot + 1) & 3; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0 && sink) sink[blockIdx.x] = (unsigned long long)smem[0];
}

template <int W>
double run(const Args& a, int grid, size_t lds, unsigned long long* sink) {
    hipFuncSetAttribute((const void*)fill_kernel<W>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((fill_kernel<W>), dim3(grid), dim3(64 * W), lds, 0, a, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((fill_kernel<W>), dim3(grid), dim3(64 * W), lds, 0, a, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 3.0 * 1e-3;
}

int main() {
    const size_t bytes = (size_t)2 << 30;
    void* src;
    if (hipMalloc(&src, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(src, 1, bytes);
    unsigned long long* sink;
    hipMalloc(&sink, 4096 * 8);
    const char* pn[4] = {"nt64", "nt128", "row512", "lin"};
    const char* fn[3] = {"hot", "share4", "cold"};
    printf("{\"probe\": \"lds fill\", \"unit\": \"TB/s chip, B/clk/CU at 2.4 GHz over 256 CUs\"}\n");
    {
        for (int wgs = 1; wgs <= 2; ++wgs)
            for (int W = 4; W <= 8; W += 4)
                for (int pat = 0; pat < 4; ++pat)
                    for (int fp = 0; fp < 3; ++fp) {
                        Args a;
                        a.src = src; a.bytes = (uint32_t)(bytes - 1 > 0xfffffff0u ? 0xfffffff0u : bytes); a.pattern = pat;
                        a.pitch = pat == 2 ? 6144 : 1536;       // fc1: K = 768 (NT rows), Cout = 3072 (TN rows)
                        a.footprint = fp;
                        // per workgroup ring: 4 slots; LDS per workgroup decides residency (1 or 2 per CU)
                        a.ni = (W == 8) ? (wgs == 1 ? 4 : 2) : (wgs == 1 ? 8 : 4);
                        if (a.ni > 8) a.ni = 8;
                        const size_t stage = (size_t)a.ni * W * 1024;
                        const size_t lds = 4 * stage;                     // 128 KiB (1 / CU) or 64 KiB (2 / CU)
                        a.steps = 400;
                        const int grid = 256 * wgs;
                        double t = (W == 4) ? run<4>(a, grid, lds, sink) : run<8>(a, grid, lds, sink);
                        const double total = (double)grid * a.steps * stage;
                        printf("{\"wg_per_cu\": %d, \"waves\": %d, \"pattern\": \"%s\", \"footprint\": \"%s\", \"stage_kib\": %zu, "
                               "\"us\": %.1f, \"TBps\": %.2f, \"B_per_clk_cu\": %.1f}\n",
                               wgs, W, pn[pat], fn[fp], stage / 1024, t * 1e6, total / t / 1e12,
                               total / t / 256.0 / 2.4e9);
                        fflush(stdout);
                    }
    }
    return 0;
}
