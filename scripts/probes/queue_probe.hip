// Does a wait that sits blocked in a second HIP stream slow the dispatch of many short kernels on the first?
// hipcc --offload-arch=gfx950 -O2 scripts/probes/queue_probe.hip -o scripts/probes/queue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>

__global__ void shortk(float* p, int n, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = i < n ? p[i] : 0.f;
    for (int k = 0; k < iters; ++k) v = v * 1.0001f + 0.5f;
    if (i < n) p[i] = v;
}
__global__ void tiny(float* p) { if (threadIdx.x == 0) p[blockIdx.x] += 1.f; }

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const int N = 1 << 22;                 // 4 M floats, 16384 workgroups of 256: ~10-20 us per kernel
    const int K = 1500;                    // kernels per "step"
    float* buf; float* small;
    hipMalloc(&buf, N * sizeof(float)); hipMalloc(&small, 4096);
    hipMemset(buf, 0, N * sizeof(float)); hipMemset(small, 0, 4096);
    int lo = 0, hi = 0;
    hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipStream_t comp, side_n, side_h;
    hipStreamCreateWithFlags(&comp, hipStreamNonBlocking);
    hipStreamCreateWithPriority(&side_n, hipStreamNonBlocking, lo);
    hipStreamCreateWithPriority(&side_h, hipStreamNonBlocking, hi);
    uint32_t* flag = nullptr;
    if (hipExtMallocWithFlags((void**)&flag, 64, hipMallocSignalMemory) != hipSuccess) flag = nullptr;
    if (flag) hipMemset(flag, 0, 64);
    hipEvent_t ev[8];
    for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    uint32_t seq = 0;
    // mode 0: no hand-off; 1: event wait on normal-priority side; 2: event wait on high-priority side;
    // 3: wait-value on normal side; 4: like 1 but the side stream also runs a tiny kernel after each wait;
    // 5: host-side hipEventSynchronize in the enqueueing thread before launching on side (blocks the host)
    const char* names[] = {"no hand-off", "event wait, normal-priority side stream", "event wait, high-priority side stream",
                           "wait-value, normal-priority side stream", "event wait + tiny kernel on side", "host-side event sync"};
    for (int mode = 0; mode < 6; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipDeviceSynchronize();
            const double t0 = now_ms();
            for (int k = 0; k < K; ++k) {
                hipLaunchKernelGGL(shortk, dim3(N / 256), dim3(256), 0, comp, buf, N, 8);
                if (mode && (k % 500 == 499)) {                 // three hand-offs per step, like three gradient buckets
                    hipStream_t side = mode == 2 ? side_h : side_n;
                    if (mode == 3 && flag) {
                        ++seq;
                        hipStreamWriteValue32(comp, flag, seq, 0);
                        hipStreamWaitValue32(side, flag, seq, hipStreamWaitValueGte, 0xffffffffu);
                    } else if (mode == 5) {
                        hipEventRecord(ev[k / 500], comp);
                        hipEventSynchronize(ev[k / 500]);
                        hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, side, small);
                    } else {
                        hipEventRecord(ev[k / 500], comp);
                        hipStreamWaitEvent(side, ev[k / 500], 0);
                        if (mode == 4) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, side, small);
                    }
                }
            }
            const double t1 = now_ms();
            hipDeviceSynchronize();
            const double t2 = now_ms();
            if (rep == 2) printf("%-48s host enqueue %7.2f ms, total %7.2f ms\n", names[mode], t1 - t0, t2 - t0);
        }
    }
    return 0;
}
