// Deterministic accumulation: the per-stream partials workspace and the ordered fold (det.h).
#include <mutex>

#include "common.h"
#include "det.h"

namespace saicv {

int g_deterministic = 0;

namespace {

// One workspace per stream that ever ran a deterministic reduction eagerly (the step's kernels run on one stream; the optional
// weight-gradient side stream of ops._SideStream is a second -- ops.set_deterministic switches it off).  A reduction's partials live
// from its kernel to its fold, both on the same stream, so successive reductions of a stream reuse the same memory in stream order.
// A stream that is being CAPTURED into a hipGraph (torch captures on a stream of its own) cannot allocate: every capturing stream
// takes the one capture workspace, which is kept as large as the largest eager workspace -- the eager warm-up steps that precede a
// capture have sized it.  Graph nodes of one capture are ordered by the capture's dependencies like launches of one stream.
struct StreamWs { hipStream_t st; float* p; size_t floats; };
// The weight-gradient kernels need at most 512 resident tiles of 128 x 128 (or 256 of 256 x 256) fp32 partials = 32 / 64 MiB; every
// other user stays far below.  Allocated up front so that no allocation falls inside a hipGraph capture.
constexpr size_t kInitialFloats = (size_t)24 << 20;      // 96 MiB
constexpr int MAX_STREAMS = 8;
StreamWs g_ws[MAX_STREAMS] = {};
int g_nws = 0;
StreamWs g_capture = {nullptr, nullptr, 0};
size_t g_max_floats = 0;
std::mutex g_mu;

bool is_capturing(hipStream_t st) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); return false; }
    return cap != hipStreamCaptureStatusNone;
}

bool grow(StreamWs* w, size_t floats, size_t start, const char* who) {
    if (w->floats >= floats) return true;
    size_t want = w->floats ? w->floats : start;
    while (want < floats) want *= 2;
    float* fresh = nullptr;
    const hipError_t e = hipMalloc(&fresh, want * sizeof(float));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        saicv::set_error("%s: deterministic workspace of %zu MiB: %s", who, want * sizeof(float) >> 20, hipGetErrorString(e));
        return false;
    }
    // the old block may still be read by a fold in flight: it is left to the process (growth is rare and doubles)
    w->p = fresh;
    w->floats = want;
    if (want > g_max_floats) g_max_floats = want;
    return true;
}

float* ws_for(hipStream_t st, size_t floats, const char* who) {
    std::lock_guard<std::mutex> lock(g_mu);
    if (is_capturing(st)) {
        if (g_capture.floats < floats) {
            set_error("%s: the deterministic capture workspace holds %zu MiB, %zu MiB needed (a workspace cannot grow inside a hipGraph "
                      "capture: call saicv_deterministic_prepare / run the step eagerly once before capturing it)", who,
                      g_capture.floats * sizeof(float) >> 20, floats * sizeof(float) >> 20);
            return nullptr;
        }
        return g_capture.p;
    }
    StreamWs* w = nullptr;
    for (int i = 0; i < g_nws; ++i)
        if (g_ws[i].st == st) { w = &g_ws[i]; break; }
    if (!w) {
        if (g_nws == MAX_STREAMS) {
            set_error("%s: deterministic reductions on more than %d streams", who, MAX_STREAMS);
            return nullptr;
        }
        w = &g_ws[g_nws++];
        *w = StreamWs{st, nullptr, 0};
    }
    if (!grow(w, floats, kInitialFloats, who)) return nullptr;
    if (!grow(&g_capture, g_max_floats, kInitialFloats, who)) return nullptr;      // (outside any capture here)
    return w->p;
}

// dst[i] += part[0][off + i] + part[1][off + i] + ...   (p ascending; one thread per element, four elements when aligned)
__global__ __launch_bounds__(256) void det_fold_kernel(const float* __restrict__ part, size_t n, int nparts, size_t off, size_t count,
                                                       float* __restrict__ dst) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const float* p = part + off + i;
    float acc = p[0];
    for (int k = 1; k < nparts; ++k) acc += p[(size_t)k * n];
    dst[i] += acc;
}
__global__ __launch_bounds__(256) void det_fold4_kernel(const float* __restrict__ part, size_t n, int nparts, size_t off, size_t count4,
                                                        float* __restrict__ dst) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count4) return;
    const float* p = part + off + i * 4;
    f32x4 acc = *reinterpret_cast<const f32x4*>(p);
    for (int k = 1; k < nparts; ++k) acc += *reinterpret_cast<const f32x4*>(p + (size_t)k * n);
    f32x4* d = reinterpret_cast<f32x4*>(dst + i * 4);
    *d = *d + acc;
}
// The chain above with the LOADS spread over the workgroup (r06): the statistics folds (BatchNorm sums and their backward sums: 2 C
// elements, up to 512 parts) are one workgroup's worth of elements and `nparts` dependent L2 round trips per thread -- 20 ... 45 us
// each, 1 ms of a deterministic ResNet-50 step.  Eight loader lanes per element quad bring 32 parts at a time into LDS (double
// buffered), lane 0 adds them in ascending part order: bit for bit the chain's result (scripts/probes/det_fold_probe.py compares).
constexpr int COOP_J = 8, COOP_E = 256 / COOP_J, COOP_CH = 32;
__global__ __launch_bounds__(256) void det_fold4_coop_kernel(const float* __restrict__ part, size_t n, int nparts, size_t off, size_t count4,
                                                             float* __restrict__ dst) {
    __shared__ f32x4 buf[2][COOP_CH][COOP_E];
    const int e = threadIdx.x % COOP_E, j = threadIdx.x / COOP_E;
    const size_t i = (size_t)blockIdx.x * COOP_E + e;
    const bool live = i < count4;
    const float* p = part + off + (live ? i : 0) * 4;
    const int chunks = (nparts + COOP_CH - 1) / COOP_CH;
    auto load = [&](int c) {
        f32x4 v[COOP_CH / COOP_J];
#pragma unroll
        for (int t = 0; t < COOP_CH / COOP_J; ++t) {
            const int k = c * COOP_CH + t * COOP_J + j;
            v[t] = (live && k < nparts) ? *reinterpret_cast<const f32x4*>(p + (size_t)k * n) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int t = 0; t < COOP_CH / COOP_J; ++t) buf[c & 1][t * COOP_J + j][e] = v[t];
    };
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    load(0);
    __syncthreads();
    for (int c = 0; c < chunks; ++c) {
        if (c + 1 < chunks) load(c + 1);                    // the other buffer: its last readers passed the barrier below
        if (j == 0) {
            const int cnt = min(COOP_CH, nparts - c * COOP_CH);
            int kk = 0;
            if (c == 0) { acc = buf[0][0][e]; kk = 1; }     // the chain starts FROM part 0 (not 0 + part 0: the sign of a zero)
            for (; kk < cnt; ++kk) acc += buf[c & 1][kk][e];
        }
        __syncthreads();
    }
    if (j == 0 && live) {
        f32x4* d = reinterpret_cast<f32x4*>(dst + i * 4);
        *d = *d + acc;
    }
}
// The same for MANY parts (a weight gradient's pixel splits: 32 ... 512 of them over a few thousand elements).  One thread per
// element walking every part is a chain of `nparts` dependent additions on 36 workgroups: 44 us per fold, 2.5 ms of a ResNet-50 step
// in deterministic mode (r06 profile).  Here J = 8 threads share an element quad: thread j adds parts j, j + 8, j + 16, ... in that
// order (four loads in flight), the eight sub-sums are added in the order j = 0 .. 7 through LDS.  A fixed function of (nparts, J):
// as reproducible as the single chain, another (equally valid) association of the same sum.
constexpr int FOLD_J = 8, FOLD_E = 256 / FOLD_J;
__global__ __launch_bounds__(256) void det_fold4_wide_kernel(const float* __restrict__ part, size_t n, int nparts, size_t off, size_t count4,
                                                             float* __restrict__ dst) {
    __shared__ f32x4 red[FOLD_J][FOLD_E];
    const int e = threadIdx.x % FOLD_E, j = threadIdx.x / FOLD_E;
    const size_t i = (size_t)blockIdx.x * FOLD_E + e;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (i < count4) {
        const float* p = part + off + i * 4;
#pragma unroll 4
        for (int k = j; k < nparts; k += FOLD_J) acc += *reinterpret_cast<const f32x4*>(p + (size_t)k * n);
    }
    red[j][e] = acc;
    __syncthreads();
    if (j == 0 && i < count4) {
        f32x4 a = red[0][e];
#pragma unroll
        for (int jj = 1; jj < FOLD_J; ++jj) a += red[jj][e];
        f32x4* d = reinterpret_cast<f32x4*>(dst + i * 4);
        *d = *d + a;
    }
}

}  // namespace

int DetParts::begin(hipStream_t stream, int parts, size_t n, const char* who, bool zero) {
    s = DetSink{nullptr, n};
    nparts = parts;
    st = stream;
    if (!g_deterministic || parts <= 1 || n == 0) return 0;      // one contributor per element: its atomics are already ordered
    float* p = ws_for(stream, (size_t)parts * n, who);
    if (!p) return -1;
    // A kernel that writes every element of every part (`zero == false`) needs no clearing pass.  (r06: under ROCm's graph packet capture
    // a captured step without it replayed wrongly -- the fold read stale partials -- and the pass in front only hid that when the eager
    // warm-up steps had it too; captured steps now run with packet capture off, package __init__.py.  SAICV_ORDERED_ZERO=1 brings the pass
    // back for the A/B, scripts/probes/bench_repro_probe.py.)
    static const bool force = getenv("SAICV_ORDERED_ZERO") && atoi(getenv("SAICV_ORDERED_ZERO")) != 0;
    const bool clear = zero || force;
    if (clear && hipMemsetAsync(p, 0, (size_t)parts * n * sizeof(float), stream) != hipSuccess) {
        set_error("%s: clearing the deterministic workspace failed: %s", who, hipGetErrorString(hipGetLastError()));
        return -1;
    }
    s.part = p;
    return 0;
}

int DetParts::fold(float* dst, size_t offset, size_t count, bool wide_ok) const {
    if (!s.part || count == 0) return 0;
    const bool vec = (s.n % 4 == 0) && (offset % 4 == 0) && (count % 4 == 0) && ((uintptr_t)dst % 16 == 0);
    // SAICV_ORDERED_FOLD=chain: every fold through the one-thread chain (the r06 A/B switch: "wide" changes the association of the weight-
    // gradient sums, "coop" must not change a bit)
    static const bool plain = getenv("SAICV_ORDERED_FOLD") && getenv("SAICV_ORDERED_FOLD")[0] == 'c';
    if (vec && !plain && wide_ok && nparts >= 4 * FOLD_J)
        hipLaunchKernelGGL(det_fold4_wide_kernel, dim3((unsigned)((count / 4 + FOLD_E - 1) / FOLD_E)), dim3(256), 0, st, s.part, s.n, nparts, offset, count / 4, dst);
    else if (vec && !plain && nparts >= COOP_CH)
        hipLaunchKernelGGL(det_fold4_coop_kernel, dim3((unsigned)((count / 4 + COOP_E - 1) / COOP_E)), dim3(256), 0, st, s.part, s.n, nparts, offset, count / 4, dst);
    else if (vec)
        hipLaunchKernelGGL(det_fold4_kernel, dim3((unsigned)((count / 4 + 255) / 256)), dim3(256), 0, st, s.part, s.n, nparts, offset, count / 4, dst);
    else
        hipLaunchKernelGGL(det_fold_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, s.part, s.n, nparts, offset, count, dst);
    return check_launch("det_fold");
}

}  // namespace saicv

extern "C" {

// Reference tools/utils.py:95-107 (set_seed: cudnn.deterministic = True).  on != 0: every reduction of this library is ordered
// (bit-reproducible run to run on the same shapes); the first call allocates the partials workspace of the calling thread's
// current use (96 MiB per stream, lazily for further streams).  Returns the previous setting.
int saicv_set_deterministic(int on) {
    const int prev = saicv::g_deterministic;
    saicv::g_deterministic = on ? 1 : 0;
    return prev;
}

int saicv_get_deterministic(void) { return saicv::g_deterministic; }

// Allocates the deterministic workspace of `stream` now (outside any capture).  0 on success.
int saicv_deterministic_prepare(void* stream) {
    return saicv::ws_for((hipStream_t)stream, saicv::kInitialFloats, "deterministic_prepare") ? 0 : -1;
}

}  // extern "C"
