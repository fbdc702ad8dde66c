// What a cross-stream event hand-off costs on a loaded queue, by event flags and by producer stream (null vs created).
// hipcc --offload-arch=gfx950 -O2 scripts/probes/event_probe.hip -o gpurun_out/event_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>

__global__ void touch(float* p, size_t n, float a) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = p[i] * a + 1.f;
}
__global__ void tiny(float* p) { if (threadIdx.x == 0) p[blockIdx.x] += 1.f; }

int main() {
    const size_t n = (size_t)256 << 20;           // 1 GiB of floats: every kernel dirties far more than the caches hold
    float* buf; float* small;
    hipMalloc(&buf, n * sizeof(float)); hipMalloc(&small, 4096);
    hipMemset(buf, 0, n * sizeof(float)); hipMemset(small, 0, 4096);
    hipStream_t side, made;
    hipStreamCreateWithFlags(&side, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&made, hipStreamNonBlocking);
    struct Case { const char* name; unsigned flags; bool use_null; bool sync; } cases[] = {
        {"no events, null stream", 0, true, false},
        {"default flags, null stream", hipEventDefault, true, true},
        {"DisableTiming, null stream", hipEventDisableTiming, true, true},
        {"DisableTiming|DisableSystemFence, null stream", hipEventDisableTiming | hipEventDisableSystemFence, true, true},
        {"DisableTiming|ReleaseToDevice, null stream", hipEventDisableTiming | hipEventReleaseToDevice, true, true},
        {"DisableTiming|ReleaseToSystem, null stream", hipEventDisableTiming | hipEventReleaseToSystem, true, true},
        {"no events, created stream", 0, false, false},
        {"DisableTiming, created stream", hipEventDisableTiming, false, true},
        {"DisableTiming|DisableSystemFence, created stream", hipEventDisableTiming | hipEventDisableSystemFence, false, true},
    };
    for (auto& c : cases) {
        hipStream_t st = c.use_null ? nullptr : made;
        hipEvent_t in, out;
        hipEventCreateWithFlags(&in, c.flags); hipEventCreateWithFlags(&out, c.flags);
        for (int rep = 0; rep < 2; ++rep) {
            hipDeviceSynchronize();
            auto t0 = std::chrono::steady_clock::now();
            for (int k = 0; k < 200; ++k) {
                hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, st, buf, n / 8, 1.0001f);      // ~128 MiB read+write
                if (c.sync && k % 20 == 19) {          // hand-off: producer -> side (tiny kernel) -> producer
                    hipEventRecord(in, st);
                    hipStreamWaitEvent(side, in, 0);
                    hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, side, small);
                    hipEventRecord(out, side);
                    hipStreamWaitEvent(st, out, 0);
                }
            }
            auto t1 = std::chrono::steady_clock::now();
            hipDeviceSynchronize();
            auto t2 = std::chrono::steady_clock::now();
            if (rep == 1)
                printf("%-52s host enqueue %7.2f ms, total %7.2f ms\n", c.name,
                       std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t0).count());
        }
        hipEventDestroy(in); hipEventDestroy(out);
    }
    return 0;
}
