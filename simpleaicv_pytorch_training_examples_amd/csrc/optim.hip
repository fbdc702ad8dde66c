// Flat-buffer fused optimizers and gradient utilities (HBM-bound, fp32), gfx950.
// Parameters, gradients and optimizer state live in flat fp32 arenas owned by the DDP
// engine; every parameter starts on a 1024-element boundary, so one 256-thread workgroup
// (4 floats per thread) never straddles two parameter groups.  Hyper-parameters sit in a
// small device table that the host refreshes each step (keeps the launch graph-replayable).
// Semantics follow torch.optim.SGD / torch.optim.AdamW as configured by the reference
// tools/utils.py:292-679 (build_optimizer) and GradScaler (tools/utils.py:199-200).
#include "common.h"
#include "saicv_internal.h"
#include "det.h"

// cache policy of the fused optimizer kernels (library variant for A/B runs: -DSAICV_OPT_NT = streaming loads and stores: every
// value is read once and written once per step)
#ifdef SAICV_OPT_NT
#define OPT_LD(ptr) __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(ptr))
#define OPT_ST(ptr, v) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(ptr))
#else
#define OPT_LD(ptr) (*reinterpret_cast<const f32x4*>(ptr))
#define OPT_ST(ptr, v) (*reinterpret_cast<f32x4*>(ptr) = (v))
#endif

namespace {

constexpr int kHyper = 8;   // floats per group: lr, wd, momentum|beta1, beta2, eps, 1-beta1, 1-beta2, flags

// Every 1024-element block belongs to ONE parameter.  `has_grad` (nullable, one byte per block) is 0 for the
// blocks of parameters that received no gradient this step: torch.optim skips those entirely (grad is None
// after zero_grad(): no weight decay, no momentum / moment decay, no step count) and so do these kernels.

// SGD with momentum (dampening 0), optional nesterov (flags bit 0)
__global__ __launch_bounds__(256) void sgd_flat_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ mom,
                                                       const int32_t* __restrict__ block_group,
                                                       const float* __restrict__ hyper,
                                                       const float* __restrict__ inv_scale,
                                                       const float* __restrict__ found_inf,
                                                       const uint8_t* __restrict__ has_grad, size_t n) {
    if (found_inf && found_inf[0] != 0.f) return;
    const int grp = block_group[blockIdx.x];
    if (grp < 0) return;
    if (has_grad && !has_grad[blockIdx.x]) return;
    const float* h = hyper + grp * kHyper;
    const float lr = h[0], wd = h[1], mu = h[2];
    const bool nesterov = h[7] != 0.f;
    const float is = inv_scale ? inv_scale[0] : 1.f;
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    f32x4 pv = OPT_LD(p + i);
    const f32x4 gv = OPT_LD(g + i);
    f32x4 mv = OPT_LD(mom + i);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float d = gv[k] * is;
        d = fmaf(wd, pv[k], d);
        if (mu != 0.f) {
            mv[k] = fmaf(mu, mv[k], d);
            d = nesterov ? fmaf(mu, mv[k], d) : mv[k];
        }
        pv[k] = fmaf(-lr, d, pv[k]);
    }
    OPT_ST(p + i, pv);
    OPT_ST(mom + i, mv);
}

__global__ __launch_bounds__(256) void adamw_flat_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v,
                                                         const int32_t* __restrict__ block_group,
                                                         const float* __restrict__ hyper,
                                                         const float* __restrict__ inv_scale,
                                                         const float* __restrict__ found_inf,
                                                         const uint8_t* __restrict__ has_grad,
                                                         float* __restrict__ step_blk, size_t n) {
    if (found_inf && found_inf[0] != 0.f) return;        // a skipped step does not advance state['step'] either
    const int grp = block_group[blockIdx.x];
    if (grp < 0) return;
    if (has_grad && !has_grad[blockIdx.x]) return;
    const float* h = hyper + grp * kHyper;
    // 1 - beta comes from the host rounded from double, as torch passes it (`lerp_(grad, 1 - beta1)`,
    // `addcmul_(grad, grad, value=1 - beta2)`): 1.f - 0.999f is 0.00100005, 4.7e-5 off
    const float lr = h[0], wd = h[1], b1 = h[2], b2 = h[3], eps = h[4], omb1 = h[5], omb2 = h[6];
    // per-parameter step count on the device (torch keeps state['step'] per parameter): bias corrections follow
    // the steps this parameter really took, and nothing about them has to be uploaded by the host per iteration
    const float t = step_blk[blockIdx.x] + 1.f;
    __syncthreads();                                     // every wavefront has read the old count
    if (threadIdx.x == 0) step_blk[blockIdx.x] = t;
    // 1 - beta^t = -expm1(t log1p(-(1 - beta))): accurate for beta2 = 0.999 at small t, where 1.f - powf(0.999f, t) is not
    const float bc1 = -expm1f(t * log1pf(-omb1)), bc2 = -expm1f(t * log1pf(-omb2));
    const float is = inv_scale ? inv_scale[0] : 1.f;
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    f32x4 pv = OPT_LD(p + i);
    const f32x4 gv = OPT_LD(g + i);
    f32x4 mv = OPT_LD(m + i);
    f32x4 vv = OPT_LD(v + i);
    const float step = lr / bc1;
    const float rs2 = rsqrtf(bc2);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float d = gv[k] * is;
        pv[k] *= (1.f - lr * wd);
        mv[k] = fmaf(omb1, d - mv[k], mv[k]);              // exp_avg.lerp_(grad, 1 - beta1)
        vv[k] = fmaf(omb2 * d, d, b2 * vv[k]);             // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
        const float denom = sqrtf(vv[k]) * rs2 + eps;
        pv[k] = fmaf(-step, mv[k] / denom, pv[k]);
    }
    OPT_ST(p + i, pv);
    OPT_ST(m + i, mv);
    OPT_ST(v + i, vv);
}

// found_inf[0] = 1 if any gradient is inf/nan;  sumsq[0] += sum(g^2)  (for clip_grad_norm_)
__global__ __launch_bounds__(256) void grad_stats_kernel(const float* __restrict__ g, size_t n,
                                                         float* __restrict__ found_inf,
                                                         float* __restrict__ sumsq, const saicv::DetSink det) {
    const size_t gstride = (size_t)gridDim.x * blockDim.x * 4;
    float ss = 0.f;
    bool bad = false;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += gstride) {
        const f32x4 gv = *reinterpret_cast<const f32x4*>(g + i);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            ss = fmaf(gv[k], gv[k], ss);
            bad = bad || !(fabsf(gv[k]) <= 3.4028234e38f);
        }
    }
    ss = wave_sum(ss);
    const unsigned long long anybad = __ballot(bad);
    if ((threadIdx.x & 63) == 0) {
        if (sumsq) saicv::det_add(det, sumsq, 0, blockIdx.x * 4 + (threadIdx.x >> 6), ss);      // one partial per wavefront
        if (anybad && found_inf) found_inf[0] = 1.f;
    }
}

// g *= min(1, max_norm / (sqrt(sumsq)*inv_scale + 1e-6)) * (fold ? inv_scale : 1)
__global__ __launch_bounds__(256) void grad_clip_scale_kernel(float* __restrict__ g, size_t n,
                                                              const float* __restrict__ sumsq,
                                                              const float* __restrict__ inv_scale,
                                                              float max_norm) {
    const float is = inv_scale ? inv_scale[0] : 1.f;
    const float total = sqrtf(sumsq[0]) * is;
    float coef = max_norm / (total + 1e-6f);
    coef = fminf(coef, 1.f) * is;
    const size_t gstride = (size_t)gridDim.x * blockDim.x * 4;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += gstride) {
        f32x4 gv = *reinterpret_cast<f32x4*>(g + i);
#pragma unroll
        for (int k = 0; k < 4; ++k) gv[k] *= coef;
        *reinterpret_cast<f32x4*>(g + i) = gv;
    }
}

// torch.nn.utils.clip_grad_value_ over the whole arena, with the GradScaler unscale folded in: g = clamp(g * inv_scale, -v, v)
// (reference tools/scripts.py:211-218 unscales, then clamps).  NaN stays NaN (the step is skipped by the found-inf flag anyway).
__global__ __launch_bounds__(256) void grad_clip_value_kernel(float* __restrict__ g, size_t n, const float* __restrict__ inv_scale,
                                                              float value) {
    const float is = inv_scale ? inv_scale[0] : 1.f;
    const size_t gstride = (size_t)gridDim.x * blockDim.x * 4;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += gstride) {
        f32x4 gv = *reinterpret_cast<f32x4*>(g + i);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float u = gv[k] * is;
            gv[k] = u != u ? u : fminf(fmaxf(u, -value), value);
        }
        *reinterpret_cast<f32x4*>(g + i) = gv;
    }
}

// GradScaler.update(): state = {scale, growth_tracker}
__global__ void scaler_update_kernel(float* __restrict__ state, const float* __restrict__ found_inf,
                                     float growth, float backoff, int interval) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (found_inf[0] != 0.f) {
        state[0] *= backoff;
        state[1] = 0.f;
    } else {
        const float t = state[1] + 1.f;
        if ((int)t >= interval) {
            state[0] *= growth;
            state[1] = 0.f;
        } else {
            state[1] = t;
        }
    }
    state[2] = 1.f / state[0];
}

}  // namespace

namespace saicv {

int sgd_flat(float* p, const float* g, float* mom, const int32_t* block_group, const float* hyper,
             const float* inv_scale, const float* found_inf, const uint8_t* has_grad, size_t n, hipStream_t st) {
    SAICV_REQUIRE(n % 1024 == 0, "sgd_flat: arena length %zu must be a multiple of 1024", n);
    hipLaunchKernelGGL(sgd_flat_kernel, dim3((unsigned)(n / 1024)), dim3(256), 0, st, p, g, mom, block_group, hyper, inv_scale, found_inf, has_grad, n);
    return check_launch("sgd_flat");
}

int adamw_flat(float* p, const float* g, float* m, float* v, const int32_t* block_group,
               const float* hyper, const float* inv_scale, const float* found_inf, const uint8_t* has_grad,
               float* step_blk, size_t n, hipStream_t st) {
    SAICV_REQUIRE(step_blk != nullptr, "adamw_flat: the per-block step counters are required");
    SAICV_REQUIRE(n % 1024 == 0, "adamw_flat: arena length %zu must be a multiple of 1024", n);
    hipLaunchKernelGGL(adamw_flat_kernel, dim3((unsigned)(n / 1024)), dim3(256), 0, st, p, g, m, v, block_group, hyper, inv_scale, found_inf, has_grad, step_blk, n);
    return check_launch("adamw_flat");
}

int grad_stats(const float* g, size_t n, float* found_inf, float* sumsq, hipStream_t st) {
    SAICV_REQUIRE(n % 4 == 0, "grad_stats: length %zu must be a multiple of 4", n);
    size_t b = (n / 4 + 255) / 256;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    DetParts det;
    if (det.begin(st, sumsq ? (int)b * 4 : 0, 1, "grad_stats")) return -1;
    hipLaunchKernelGGL(grad_stats_kernel, dim3((unsigned)b), dim3(256), 0, st, g, n, found_inf, sumsq, det.sink());
    if (check_launch("grad_stats")) return -2;
    return det.fold(sumsq, 0, 1);
}

int grad_clip_scale(float* g, size_t n, const float* sumsq, const float* inv_scale, double max_norm,
                    hipStream_t st) {
    SAICV_REQUIRE(n % 4 == 0, "grad_clip_scale: length %zu must be a multiple of 4", n);
    size_t b = (n / 4 + 255) / 256;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    hipLaunchKernelGGL(grad_clip_scale_kernel, dim3((unsigned)b), dim3(256), 0, st, g, n, sumsq, inv_scale, (float)max_norm);
    return check_launch("grad_clip_scale");
}

int grad_clip_value(float* g, size_t n, const float* inv_scale, double value, hipStream_t st) {
    SAICV_REQUIRE(n % 4 == 0 && value > 0, "grad_clip_value: length %zu must be a multiple of 4, the bound positive", n);
    size_t b = (n / 4 + 255) / 256;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    hipLaunchKernelGGL(grad_clip_value_kernel, dim3((unsigned)b), dim3(256), 0, st, g, n, inv_scale, (float)value);
    return check_launch("grad_clip_value");
}

int scaler_update(float* state, const float* found_inf, double growth, double backoff, int interval,
                  hipStream_t st) {
    hipLaunchKernelGGL(scaler_update_kernel, dim3(1), dim3(64), 0, st, state, found_inf, (float)growth, (float)backoff, interval);
    return check_launch("scaler_update");
}

}  // namespace saicv
