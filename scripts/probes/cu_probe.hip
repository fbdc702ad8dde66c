// Which CU does a workgroup run on, and which workgroups share a CU?  (placement is for SPEED decisions only)
// hipcc --offload-arch=gfx950 -O3 scripts/probes/cu_probe.hip -o /tmp/cu_probe && /tmp/cu_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <vector>
__global__ __launch_bounds__(512) void probe(unsigned* out, unsigned* tickets, int spin) {
    extern __shared__ char smem[];
    unsigned hw = __builtin_amdgcn_s_getreg(4 | (31 << 11));       // HW_REG_HW_ID, 32 bits
    unsigned xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11));      // HW_REG_XCC_ID, 4 bits
    unsigned cu = ((xcc & 15) << 8) | ((hw >> 8) & 0xff);          // xcc | se_id sh_id cu_id
    if (threadIdx.x == 0) {
        unsigned t = atomicAdd(&tickets[cu], 1u);
        unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        out[blockIdx.x * 4 + 0] = hw;
        out[blockIdx.x * 4 + 1] = xcc;
        out[blockIdx.x * 4 + 2] = t;
        out[blockIdx.x * 4 + 3] = (unsigned)t0;
        smem[0] = 1;
    }
    // keep the CU busy so that every workgroup of the grid is resident at once
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(100);
}
int main() {
    const int lds[2] = {72 * 1024, 140 * 1024};
    for (int cfg = 0; cfg < 2; ++cfg) {
        const int grid = cfg == 0 ? 512 : 256;
        unsigned *out, *tk;
        hipMalloc(&out, grid * 16); hipMalloc(&tk, 4096 * 4); hipMemset(tk, 0, 4096 * 4);
        hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(probe, dim3(grid), dim3(512), lds[cfg], 0, out, tk, 200);
            hipDeviceSynchronize();
            std::vector<unsigned> h(grid * 4);
            hipMemcpy(h.data(), out, grid * 16, hipMemcpyDeviceToHost);
            std::map<unsigned, std::vector<int>> by;
            for (int b = 0; b < grid; ++b) by[((h[b * 4 + 1] & 15) << 8) | ((h[b * 4] >> 8) & 0xff)].push_back(b);
            int hist[8] = {0};
            for (auto& kv : by) hist[kv.second.size() < 7 ? kv.second.size() : 7]++;
            printf("cfg lds=%d grid=%d rep=%d: distinct CU ids %zu; CUs with 1/2/3/4 blocks: %d %d %d %d\n", lds[cfg], grid, rep,
                   by.size(), hist[1], hist[2], hist[3], hist[4]);
            int shown = 0;
            for (auto& kv : by) {
                if (shown++ >= 6) break;
                printf("  cu %03x:", kv.first);
                for (int b : kv.second) printf(" blk %d (xcc %u hw %08x ticket %u t %u)", b, h[b * 4 + 1], h[b * 4], h[b * 4 + 2], h[b * 4 + 3]);
                printf("\n");
            }
            int parity_ok = 0, pairs = 0;
            for (auto& kv : by) if (kv.second.size() == 2) { pairs++; parity_ok += ((h[kv.second[0] * 4 + 2] ^ h[kv.second[1] * 4 + 2]) & 1); }
            printf("  pairs %d, with different ticket parity %d\n", pairs, parity_ok);
        }
        hipFree(out); hipFree(tk);
    }
    return 0;
}
