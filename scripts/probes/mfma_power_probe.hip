// What does a bf16 matrix-core loop sustain at the 1 400 W socket limit, by instruction shape and by LDS traffic per flop?
// (r05: the GEMM kernels run AT the power limit on random data -- scripts/power_probe.py -- so throughput = power / energy per flop.)
// Variants (template V):
//   0  v_mfma_f32_16x16x32_bf16, operands rotate through registers only (8 A x 8 B fragments of random bits), 16 accumulators
//   1  v_mfma_f32_32x32x16_bf16, the same (4 A x 4 B fragments, 4 accumulators of 16 registers)
//   2  16x16x32, 64 x 64 wave tile fed from LDS: 8 ds_read_b128 per 16 MFMAs (igemm_nt1 256 x 128 geometry)
//   3  16x16x32, 128 x 64 wave tile fed from LDS: 12 ds_read_b128 per 32 MFMAs (256 x 256 / 8-wave geometry)
//   4  32x32x16, 64 x 64 wave tile fed from LDS: 8 ds_read_b128 per 8 MFMAs (same bytes per flop as 2)
//   5  32x32x16, 128 x 128 wave tile fed from LDS: 16 ds_read_b128 per 32 MFMAs (half the LDS bytes per flop of 2)
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/probes/mfma_power_probe scripts/probes/mfma_power_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__device__ inline uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
// a random bf16 pair with exponents near 1.0 (no inf / nan, realistic mantissa toggling)
__device__ inline uint32_t rnd_pair(uint32_t s) {
    const uint32_t h = hash32(s);
    return (h & 0x807f807fU) | 0x3f003f00U | ((h >> 8) & 0x00800080U);
}
__device__ inline u32x4 rnd_chunk(uint32_t s) { return u32x4{rnd_pair(s), rnd_pair(s + 1), rnd_pair(s + 2), rnd_pair(s + 3)}; }

template <int V>
__global__ __launch_bounds__(256) void probe(float* out, int iters, int zero) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // LDS image: 64 KiB of random chunks
    for (int i = tid; i < 4096; i += 256) reinterpret_cast<u32x4*>(smem)[i] = zero ? u32x4{0, 0, 0, 0} : rnd_chunk(i * 4 + blockIdx.x * 16384);
    __syncthreads();
    float sink = 0.f;
    if constexpr (V == 0) {
        u32x4 a[8], b[8];
        for (int i = 0; i < 8; ++i) { a[i] = zero ? u32x4{0,0,0,0} : rnd_chunk(tid * 64 + i * 4); b[i] = zero ? u32x4{0,0,0,0} : rnd_chunk(tid * 64 + 32 + i * 4 + 7777); }
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[(i + j) & 7]), __builtin_bit_cast(bf16x8, b[(i * 3 + j) & 7]), acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) sink += acc[i][0] + acc[i][3];
    } else if constexpr (V == 1) {
        u32x4 a[4], b[4];
        for (int i = 0; i < 4; ++i) { a[i] = zero ? u32x4{0,0,0,0} : rnd_chunk(tid * 64 + i * 4); b[i] = zero ? u32x4{0,0,0,0} : rnd_chunk(tid * 64 + 32 + i * 4 + 7777); }
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[(i + j) & 3]), __builtin_bit_cast(bf16x8, b[(i + 3 * j + (j >> 2)) & 3]), acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) sink += acc[i][0] + acc[i][15];
    } else if constexpr (V == 2 || V == 3) {
        constexpr int MT = V == 2 ? 4 : 8, NT = 4;
        f32x4 acc[NT][MT];
        for (int n = 0; n < NT; ++n) for (int m = 0; m < MT; ++m) acc[n][m] = f32x4{0, 0, 0, 0};
        const char* base = smem + lane * 16 + wave * 4096;
        for (int it = 0; it < iters; ++it) {
            const char* q = base + ((it & 3) << 14);
            u32x4 af[MT], wf[NT];
#pragma unroll
            for (int m = 0; m < MT; ++m) af[m] = *reinterpret_cast<const u32x4*>(q + ((m * 1024) & 0x3fff));
#pragma unroll
            for (int n = 0; n < NT; ++n) wf[n] = *reinterpret_cast<const u32x4*>(q + (((MT + n) * 1024 + 512) & 0x3fff));
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[n]), __builtin_bit_cast(bf16x8, af[m]), acc[n][m], 0, 0, 0);
        }
        for (int n = 0; n < NT; ++n) for (int m = 0; m < MT; ++m) sink += acc[n][m][0];
    } else {
        constexpr int MT = V == 4 ? 2 : 4, NT = V == 4 ? 2 : 4;      // 32-row blocks
        f32x16 acc[NT][MT];
        for (int n = 0; n < NT; ++n) for (int m = 0; m < MT; ++m) for (int r = 0; r < 16; ++r) acc[n][m][r] = 0.f;
        const char* base = smem + lane * 16 + wave * 4096;
        for (int it = 0; it < iters; ++it) {
            const char* q = base + ((it & 3) << 14);
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {        // two K = 16 halves of a 32-deep step
                u32x4 af[MT], wf[NT];
#pragma unroll
                for (int m = 0; m < MT; ++m) af[m] = *reinterpret_cast<const u32x4*>(q + (((kh * MT + m) * 1024) & 0x3fff));
#pragma unroll
                for (int n = 0; n < NT; ++n) wf[n] = *reinterpret_cast<const u32x4*>(q + (((2 * MT + kh * NT + n) * 1024 + 512) & 0x3fff));
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        acc[n][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[n]), __builtin_bit_cast(bf16x8, af[m]), acc[n][m], 0, 0, 0);
            }
        }
        for (int n = 0; n < NT; ++n) for (int m = 0; m < MT; ++m) sink += acc[n][m][0];
    }
    if (sink == 12345.678f) out[blockIdx.x] = sink;
}

template <int V> void run(const char* name, double flop_per_iter_wave, int wg_per_cu, float* out, int zero) {
    const int iters = 2000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    dim3 grid(256 * wg_per_cu), block(256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(probe<V>, grid, block, 65536, 0, out, iters, zero);
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    int n = 0;
    hipEventRecord(e0);
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 2.0) {
        for (int j = 0; j < 20; ++j) hipLaunchKernelGGL(probe<V>, grid, block, 65536, 0, out, iters, zero);
        n += 20;
        hipDeviceSynchronize();
    }
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = flop_per_iter_wave * iters * 4.0 * grid.x * n;
    const double now = std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
    printf("{\"variant\": %d, \"name\": \"%s\", \"wg_per_cu\": %d, \"zero\": %d, \"tflops\": %.1f, \"t_end\": %.2f}\n", V, name, wg_per_cu, zero, flops / (ms * 1e-3) / 1e12, now);
    fflush(stdout);
}

int main(int argc, char** argv) {
    float* out;
    hipMalloc(&out, 1 << 20);
    const int wg = argc > 1 ? atoi(argv[1]) : 2;
    for (int zero = 0; zero < 2; ++zero) {
        run<0>("16x16x32 registers only", 64.0 * 16384, wg, out, zero);
        run<1>("32x32x16 registers only", 32.0 * 32768, wg, out, zero);
        run<2>("16x16x32 64x64 wave tile from LDS (8 rd / 16 mfma)", 16.0 * 16384, wg, out, zero);
        run<3>("16x16x32 128x64 wave tile from LDS (12 rd / 32 mfma)", 32.0 * 16384, wg, out, zero);
        run<4>("32x32x16 64x64 wave tile from LDS (8 rd / 8 mfma)", 8.0 * 32768, wg, out, zero);
        run<5>("32x32x16 128x128 wave tile from LDS (16 rd / 32 mfma)", 32.0 * 32768, wg, out, zero);
    }
    return 0;
}
