// On-device batch preparation: what the reference does per sample on CPU workers and per batch in
// its collaters, as HBM-bound streaming kernels on the batch that is already in device memory.
//
//   saicv_mixup_cutmix     reference SimpleAICV/classification/mixupcutmixclassificationcollator.py:140-284 -- batch / pair /
//                          elem Mixup and CutMix of sample i with sample B-1-i from a HOST-drawn plan (lambda, box: the numpy
//                          draws stay on the host, in the reference's order, so a seeded run mixes the same pairs), with the
//                          per-channel affine normalisation of the dataset transform folded in for uint8 inputs.
//   saicv_soft_labels      the smoothed, mixed one-hot labels of the same collater (:262-284).
//   saicv_sam_sample_point reference tools/interactive_segmentation_scripts.py:202-228 -- one click per sample, uniformly over the
//                          error region of the current mask prediction, WITHOUT the [B, 1, H, W, 2] noise tensor the reference
//                          materialises (168 MB at batch 20, four times per step): a counter-based generator gives every
//                          (pixel, label) slot its own 32-bit draw and the arg-max travels as a packed 64-bit key.
//
//   saicv_u8_normalize     reference SimpleAICV/classification/common.py:228-248 (TorchMeanStdNormalize = torchvision ToTensor +
//                          Normalize) on the uint8 NHWC batch the loader shipped: float(v) / 255, - mean[c], / std[c], every step
//                          rounded like the tensor expressions (a quarter of the fp32 batch's PCIe / HBM traffic on the way in).
//   saicv_random_erase     reference common.py:561-640 (RandomErasing): boxes and colours drawn by the HOST with the reference's
//                          numpy calls, pixels filled on the device; mode 'pixel' takes its N(0, 1) values from a counter-based
//                          generator instead of shipping h * w * c host draws.
//
// Arithmetic of the first two is written to be BIT-IDENTICAL to the reference's fp32 tensor expressions (separately rounded
// products and sum, no fused multiply-add): tests compare with the fixture the reference collater produced.
#include "common.h"
#include "saicv_internal.h"
#include "../../include/saicv_hip.h"

// No fused multiply-add anywhere in this file (plain operators under contract(off); HIP's __fmul_rn / __fadd_rn are inline
// functions of a header compiled with contraction ALLOWED, and keep that flag when inlined): a contracted product is not
// rounded, while the reference's tensor expressions round every step.
#pragma clang fp contract(off)

namespace {

// one thread per (pixel, all channels); C <= 4
template <typename TIN>
__global__ __launch_bounds__(256) void mixup_cutmix_kernel(const TIN* __restrict__ src, const saicv_mix_plan* __restrict__ plan,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           float* __restrict__ dst, int B, int H, int W, int C) {
    const size_t npix = (size_t)B * H * W;
    const size_t gstride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gstride) {
        size_t t = i;
        const int x = (int)(t % W); t /= W;
        const int y = (int)(t % H);
        const int b = (int)(t / H);
        const saicv_mix_plan pl = plan[b];
        const size_t own = i * C;
        const size_t other = ((((size_t)(B - 1 - b)) * H + y) * W + x) * C;
        const bool in_box = y >= pl.yl && y < pl.yh && x >= pl.xl && x < pl.xh;
        for (int c = 0; c < C; ++c) {
            float a = (float)src[own + c], o = (float)src[other + c];
            if (scale != nullptr) {                          // dataset normalisation: v * scale[c] + shift[c], rounded as torch does
                a = a * scale[c] + shift[c];
                o = o * scale[c] + shift[c];
            }
            float v = a;
            if (pl.mode == 1) v = a * pl.lam + o * pl.one_minus_lam;
            else if (pl.mode == 2 && in_box) v = o;
            dst[own + c] = v;
        }
    }
}

__global__ __launch_bounds__(256) void soft_labels_kernel(const int64_t* __restrict__ labels, const saicv_mix_plan* __restrict__ plan,
                                                          float off, float on, float* __restrict__ out, int B, int NC) {
    const size_t total = (size_t)B * NC;
    const size_t gstride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gstride) {
        const int b = (int)(i / NC), c = (int)(i - (size_t)b * NC);
        const float ya = labels[b] == c ? on : off;
        const float yb = labels[B - 1 - b] == c ? on : off;
        // y * lam + y.flip(0) * (1 - lam) with the LABEL lambda of the plan (the box-corrected one for CutMix)
        out[i] = ya * plan[b].label_lam + yb * plan[b].label_one_minus_lam;
    }
}

// ---- SAM click sampling
DEVINL uint32_t mix32(uint32_t a, uint32_t b, uint32_t c) {        // counter-based draw: three rounds of a 32-bit mixer over (seed, sample, slot)
    uint32_t h = a ^ 0x9e3779b9u;
    h = (h ^ b) * 0x85ebca6bu; h ^= h >> 13;
    h = (h ^ c) * 0xc2b2ae35u; h ^= h >> 16;
    h *= 0x27d4eb2fu; h ^= h >> 15;
    h *= 0x165667b1u; h ^= h >> 13;
    return h;
}

DEVINL unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long o = __shfl_xor(v, d, 64);
        v = o > v ? o : v;
    }
    return v;
}

// keys[b][0]: best false-positive slot (label 0), [1]: best false-negative slot (label 1), [2]: best background slot (label 0,
// used when the prediction is exact).  key = (draw + 1) << 32 | pixel index; 0 = no candidate.
template <typename TP>
__global__ __launch_bounds__(256) void sample_point_scan_kernel(const float* __restrict__ gt, const TP* __restrict__ pred, long pred_stride,
                                                                const int64_t* __restrict__ pred_index, int pred_channels,
                                                                float gt_threshold, float pred_threshold, uint32_t seed0,
                                                                const uint32_t* __restrict__ seed_dev,
                                                                unsigned long long* __restrict__ keys, int B, int HW) {
    const uint32_t seed = seed_dev ? seed0 + *seed_dev : seed0;       // a captured step draws its seed from device memory (the host bumps it between replays)
    const int b = blockIdx.y;
    const float* g = gt + (size_t)b * HW;
    const TP* p = nullptr;
    if (pred != nullptr) {
        const long ch = pred_index ? (long)pred_index[b] : 0;
        p = pred + (size_t)b * pred_stride * pred_channels + (size_t)ch * pred_stride;
    }
    unsigned long long kfp = 0ull, kfn = 0ull, kbg = 0ull;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
        const bool gm = g[i] > gt_threshold;
        const bool pm = p != nullptr && (float)p[i] > pred_threshold;
        const uint32_t d0 = mix32(seed, (uint32_t)b, 2u * (uint32_t)i), d1 = mix32(seed, (uint32_t)b, 2u * (uint32_t)i + 1u);
        const unsigned long long k0 = (((unsigned long long)d0 + 1ull) << 32) | (unsigned)i;
        const unsigned long long k1 = (((unsigned long long)d1 + 1ull) << 32) | (unsigned)i;
        if (!gm && pm && k0 > kfp) kfp = k0;
        if (gm && !pm && k1 > kfn) kfn = k1;
        if (!gm && !pm && k0 > kbg) kbg = k0;
    }
    kfp = wave_max_u64(kfp); kfn = wave_max_u64(kfn); kbg = wave_max_u64(kbg);
    if ((threadIdx.x & 63) == 0) {
        if (kfp) atomicMax(&keys[(size_t)b * 4 + 0], kfp);
        if (kfn) atomicMax(&keys[(size_t)b * 4 + 1], kfn);
        if (kbg) atomicMax(&keys[(size_t)b * 4 + 2], kbg);
    }
}

__global__ void sample_point_pick_kernel(const unsigned long long* __restrict__ keys, float* __restrict__ points, int B, int W) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const unsigned long long kfp = keys[(size_t)b * 4 + 0], kfn = keys[(size_t)b * 4 + 1], kbg = keys[(size_t)b * 4 + 2];
    unsigned long long k;
    float label;
    if (kfp | kfn) {                               // an error region exists: the largest draw over both label slots wins
        const bool take_fn = kfp == 0ull ? true : (kfn == 0ull ? false : (kfn >> 32) > (kfp >> 32));
        k = take_fn ? kfn : kfp;
        label = take_fn ? 1.f : 0.f;
    } else {                                       // exact prediction: a background pixel, label 0; nothing at all: pixel 0
        k = kbg;
        label = 0.f;
    }
    const unsigned idx = (unsigned)k;
    points[b * 3 + 0] = (float)(idx % (unsigned)W);
    points[b * 3 + 1] = (float)(idx / (unsigned)W);
    points[b * 3 + 2] = label;
}

// uint8 NHWC -> fp32 NHWC: ((float)v / 255 - mean[c]) / std[c], three separately rounded steps (torchvision's
// img.to(float32).div(255) followed by tensor.sub_(mean).div_(std))
__global__ __launch_bounds__(256) void u8_normalize_kernel(const uint8_t* __restrict__ src, const float* __restrict__ mean,
                                                           const float* __restrict__ stdv, float* __restrict__ dst, size_t n, int C) {
    const size_t gstride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gstride) {
        const int c = (int)(i % (size_t)C);
        float v = (float)src[i] / 255.f;
        v = v - mean[c];
        dst[i] = v / stdv[c];
    }
}

DEVINL unsigned erase_hash(unsigned a, unsigned b, unsigned c) {
    unsigned h = a * 0x9E3779B9u + b;
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    h += c * 0x27d4eb2fu;
    h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12; h *= 0x297a2d39u; h ^= h >> 15;
    return h;
}

// one workgroup row (blockIdx.y) per box; mode 0: the box takes color[c]; mode 1: every element its own N(0, 1) draw
// (Box-Muller on two 24-bit uniforms of a counter-based hash of (seed, box, element))
__global__ __launch_bounds__(256) void random_erase_kernel(float* __restrict__ x, const saicv_erase_box* __restrict__ boxes, int H,
                                                           int W, int C, unsigned seed) {
    const saicv_erase_box bx = boxes[blockIdx.y];
    const int n = bx.h * bx.w * C;
    float* base = x + (((size_t)bx.b * H + bx.top) * W + bx.left) * C;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int c = i % C, px = i / C, xx = px % bx.w, yy = px / bx.w;
        float v;
        if (bx.mode == 0) {
            v = bx.color[c];
        } else {
            const unsigned k1 = erase_hash(seed, (unsigned)blockIdx.y, (unsigned)i * 2u), k2 = erase_hash(seed ^ 0x68bc21ebu, (unsigned)blockIdx.y, (unsigned)i * 2u + 1u);
            const float u1 = ((float)(k1 >> 8) + 1.f) * (1.f / 16777216.f);        // (0, 1]
            const float u2 = (float)(k2 >> 8) * (1.f / 16777216.f);                // [0, 1)
            v = sqrtf(-2.f * __logf(u1)) * __cosf(6.28318530717958647692f * u2);
        }
        base[((size_t)yy * W + xx) * C + c] = v;
    }
}

int sgrid(size_t n) {
    size_t g = (n + 255) / 256;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}


// ---- SAM prompt tokens (f3 c): reference segment_anything/prompt_encoder.py:150-190 (embed_points / embed_boxes) and :28-49
// (random-Fourier position encoding) as ONE launch.  Token t of sample b is a click (x, y, label), the padding click the
// reference appends when no box is given, or a box corner; its 2F features are [sin(phase), cos(phase)],
// phase_f = 2 pi ((2 u - 1) G[0][f] + (2 v - 1) G[1][f]), (u, v) = (x + 0.5, y + 0.5) / image_size, plus the row of the learned table
// its kind selects: 0 / 1 = negative / positive click, 2 / 3 = box corners, 4 = "not a point" (label -1: the encoding is
// REPLACED by the table row), 5 = other labels (encoding only).  kinds[b][t] is kept for the backward.
__global__ __launch_bounds__(256) void prompt_tokens_kernel(const float* __restrict__ points, int Np, int pad, const float* __restrict__ boxes,
                                                            const float* __restrict__ gauss, int F, const float* __restrict__ table,
                                                            float inv_size, float* __restrict__ tokens, int* __restrict__ kinds, int B, int T) {
    const int bt = blockIdx.x;
    const int b = bt / T, t = bt - b * T;
    const int npts = points ? Np + pad : 0;
    float x, y;
    int kind;
    if (t < npts) {
        if (t < Np) {
            const float* p = points + ((size_t)b * Np + t) * 3;
            x = p[0] + 0.5f; y = p[1] + 0.5f;
            const float lab = p[2];
            kind = lab == -1.f ? 4 : lab == 0.f ? 0 : lab == 1.f ? 1 : 5;
        } else {                                   // the padding click: coordinates (0, 0) unshifted, label -1
            x = 0.f; y = 0.f; kind = 4;
        }
    } else {
        const int c = t - npts;                    // corner 0 = (x1, y1), corner 1 = (x2, y2)
        const float* q = boxes + (size_t)b * 4 + c * 2;
        x = q[0] + 0.5f; y = q[1] + 0.5f;
        kind = 2 + c;
    }
    if (threadIdx.x == 0) kinds[bt] = kind;
    const float u = 2.f * (x * inv_size) - 1.f, v = 2.f * (y * inv_size) - 1.f;
    float* out = tokens + (size_t)bt * 2 * F;
    for (int f = threadIdx.x; f < F; f += blockDim.x) {
        const float ph = 6.283185307179586f * (u * gauss[f] + v * gauss[F + f]);
        float sn = sinf(ph), cs = cosf(ph);
        if (kind == 4) { sn = 0.f; cs = 0.f; }
        if (kind < 5) { sn += table[(size_t)kind * 2 * F + f]; cs += table[(size_t)kind * 2 * F + F + f]; }
        out[f] = sn;
        out[F + f] = cs;
    }
}

// d table[kind][c] += sum over the tokens of that kind of d tokens[b][t][c]   (the only trainable inputs of the sparse path)
__global__ __launch_bounds__(256) void prompt_tokens_bwd_kernel(const float* __restrict__ dtokens, const int* __restrict__ kinds,
                                                                float* __restrict__ dtable, int BT, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < BT; ++i) {
        const int k = kinds[i];
        const float g = dtokens[(size_t)i * C + c];
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[j] += (k == j) ? g : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) dtable[(size_t)j * C + c] += acc[j];
}

// the same encoding on the centres of an S x S grid (get_dense_pe: reference prompt_encoder.py:28-38): out[c][i][j], c < 2F
__global__ __launch_bounds__(256) void grid_pe_kernel(const float* __restrict__ gauss, int F, int S, float* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= F * S * S) return;
    const int f = idx / (S * S), r = idx - f * S * S, i = r / S, j = r - i * S;
    const float u = 2.f * (((float)j + 0.5f) / (float)S) - 1.f, v = 2.f * (((float)i + 0.5f) / (float)S) - 1.f;
    const float ph = 6.283185307179586f * (u * gauss[f] + v * gauss[F + f]);
    out[(size_t)f * S * S + r] = sinf(ph);
    out[(size_t)(F + f) * S * S + r] = cosf(ph);
}


// ---- DETR sine position embedding (f3): reference detection/models/backbones/detr_resnet.py:28-64.  One block per sample: the
// running counts of un-padded pixels down each column / along each row (the reference's two cumsums) live in LDS, are
// normalised by the column / row totals to [0, 2 pi] and expanded to F sine / cosine features each: out[b][c][y][x], channels
// [0, F) from the row coordinate, [F, 2F) from the column coordinate, feature k = sin (k even) / cos (k odd) of
// coordinate / temperature^(2 (k / 2) / F).  No trainable input, no backward.
__global__ __launch_bounds__(256) void detr_sine_pe_kernel(const unsigned char* __restrict__ mask, float* __restrict__ out, int H, int W,
                                                           int F, float temperature, float eps) {
    extern __shared__ float pe_smem[];
    float* ye = pe_smem;                 // [H][W]
    float* xe = pe_smem + H * W;
    const int b = blockIdx.x;
    const unsigned char* m = mask + (size_t)b * H * W;
    for (int x = threadIdx.x; x < W; x += blockDim.x) {
        float run = 0.f;
        for (int y = 0; y < H; ++y) { run += m[y * W + x] ? 0.f : 1.f; ye[y * W + x] = run; }
    }
    for (int y = threadIdx.x; y < H; y += blockDim.x) {
        float run = 0.f;
        for (int x = 0; x < W; ++x) { run += m[y * W + x] ? 0.f : 1.f; xe[y * W + x] = run; }
    }
    __syncthreads();
    const float two_pi = 6.283185307179586f;
    float* o = out + (size_t)b * 2 * F * H * W;
    // blockIdx.y splits the 2F x H x W outputs of a sample (the counts above are recomputed per block: H + W short loops)
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < 2 * F * H * W; i += gridDim.y * blockDim.x) {
        const int c = i / (H * W), r = i - c * (H * W), y = r / W, x = r - y * W;
        const bool isy = c < F;
        const int k = isy ? c : c - F;
        const float tot = isy ? ye[(H - 1) * W + x] : xe[y * W + (W - 1)];
        const float coord = (isy ? ye[r] : xe[r]) / (tot + eps) * two_pi;
        const float v = coord / powf(temperature, (float)(2 * (k / 2)) / (float)F);
        o[i] = (k & 1) ? cosf(v) : sinf(v);
    }
}

}  // namespace

extern "C" {

int saicv_mixup_cutmix(int src_is_u8, const void* src, const saicv_mix_plan* plan, const float* scale, const float* shift,
                       float* dst, int B, int H, int W, int C, void* stream) {
    SAICV_REQUIRE(src && plan && dst && B > 0 && H > 0 && W > 0 && C > 0 && C <= 4, "saicv_mixup_cutmix: bad arguments (C=%d)", C);
    SAICV_REQUIRE((scale == nullptr) == (shift == nullptr), "saicv_mixup_cutmix: scale and shift come together");
    SAICV_REQUIRE((const void*)src != (const void*)dst, "saicv_mixup_cutmix: not in place (sample B-1-i is read while sample i is written)");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t npix = (size_t)B * H * W;
    if (src_is_u8)
        hipLaunchKernelGGL(mixup_cutmix_kernel<uint8_t>, dim3(sgrid(npix)), dim3(256), 0, st, (const uint8_t*)src, plan, scale, shift, dst, B, H, W, C);
    else
        hipLaunchKernelGGL(mixup_cutmix_kernel<float>, dim3(sgrid(npix)), dim3(256), 0, st, (const float*)src, plan, scale, shift, dst, B, H, W, C);
    return saicv::check_launch("mixup_cutmix");
}

int saicv_soft_labels(const long long* labels, const saicv_mix_plan* plan, float off_value, float on_value, float* out, int B,
                      int num_classes, void* stream) {
    SAICV_REQUIRE(labels && plan && out && B > 0 && num_classes > 0, "saicv_soft_labels: bad arguments");
    hipLaunchKernelGGL(soft_labels_kernel, dim3(sgrid((size_t)B * num_classes)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       (const int64_t*)labels, plan, off_value, on_value, out, B, num_classes);
    return saicv::check_launch("soft_labels");
}

int saicv_sam_sample_point(int pred_dtype, const float* gt, const void* pred, long pred_plane_stride, const long long* pred_index,
                           int pred_channels, float gt_threshold, float pred_threshold, unsigned int seed,
                           unsigned long long* keys_ws, float* points, int B, int H, int W, void* stream) {
    return saicv_sam_sample_point_dseed(pred_dtype, gt, pred, pred_plane_stride, pred_index, pred_channels, gt_threshold, pred_threshold, seed,
                                        nullptr, keys_ws, points, B, H, W, stream);
}

int saicv_sam_sample_point_dseed(int pred_dtype, const float* gt, const void* pred, long pred_plane_stride, const long long* pred_index,
                                 int pred_channels, float gt_threshold, float pred_threshold, unsigned int seed, const unsigned int* seed_device,
                                 unsigned long long* keys_ws, float* points, int B, int H, int W, void* stream) {
    SAICV_REQUIRE(gt && keys_ws && points && B > 0 && H > 0 && W > 0, "saicv_sam_sample_point: bad arguments");
    SAICV_REQUIRE((size_t)H * W < (1ull << 31), "saicv_sam_sample_point: mask too large");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(keys_ws, 0, (size_t)B * 4 * sizeof(unsigned long long), st) != hipSuccess) {
        saicv::set_error("saicv_sam_sample_point: memset failed");
        return -2;
    }
    const int HW = H * W;
    int gx = (HW + 256 * 16 - 1) / (256 * 16);
    if (gx > 256) gx = 256;
    if (gx < 1) gx = 1;
    dim3 grid(gx, B);
    if (pred_dtype == SAICV_DTYPE_BF16)
        hipLaunchKernelGGL(sample_point_scan_kernel<bf16_t>, grid, dim3(256), 0, st, gt, (const bf16_t*)pred, pred_plane_stride,
                           (const int64_t*)pred_index, pred_channels, gt_threshold, pred_threshold, seed, seed_device, keys_ws, B, HW);
    else
        hipLaunchKernelGGL(sample_point_scan_kernel<float>, grid, dim3(256), 0, st, gt, (const float*)pred, pred_plane_stride,
                           (const int64_t*)pred_index, pred_channels, gt_threshold, pred_threshold, seed, seed_device, keys_ws, B, HW);
    hipLaunchKernelGGL(sample_point_pick_kernel, dim3((B + 63) / 64), dim3(64), 0, st, keys_ws, points, B, W);
    return saicv::check_launch("sam_sample_point");
}

int saicv_sam_prompt_tokens(const float* points, int Np, int pad, const float* boxes, const float* gauss, int F, const float* table,
                            float image_size, float* tokens, int* kinds, int B, void* stream) {
    SAICV_REQUIRE((points || boxes) && gauss && table && tokens && kinds && B > 0 && F > 0 && image_size > 0.f,
                  "saicv_sam_prompt_tokens: bad arguments");
    SAICV_REQUIRE(points || (Np == 0 && pad == 0), "saicv_sam_prompt_tokens: point count without points");
    const int T = (points ? Np + pad : 0) + (boxes ? 2 : 0);
    SAICV_REQUIRE(T > 0, "saicv_sam_prompt_tokens: no tokens");
    hipLaunchKernelGGL(prompt_tokens_kernel, dim3(B * T), dim3(F < 256 ? ((F + 63) / 64) * 64 : 256), 0, static_cast<hipStream_t>(stream),
                       points, Np, pad, boxes, gauss, F, table, 1.f / image_size, tokens, kinds, B, T);
    return saicv::check_launch("sam_prompt_tokens");
}

int saicv_sam_prompt_tokens_bwd(const float* dtokens, const int* kinds, float* dtable, int BT, int C, void* stream) {
    SAICV_REQUIRE(dtokens && kinds && dtable && BT > 0 && C > 0, "saicv_sam_prompt_tokens_bwd: bad arguments");
    hipLaunchKernelGGL(prompt_tokens_bwd_kernel, dim3((C + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), dtokens, kinds,
                       dtable, BT, C);
    return saicv::check_launch("sam_prompt_tokens_bwd");
}

int saicv_sam_grid_pe(const float* gauss, int F, int S, float* out, void* stream) {
    SAICV_REQUIRE(gauss && out && F > 0 && S > 0, "saicv_sam_grid_pe: bad arguments");
    hipLaunchKernelGGL(grid_pe_kernel, dim3((F * S * S + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), gauss, F, S, out);
    return saicv::check_launch("sam_grid_pe");
}

int saicv_u8_normalize(const unsigned char* src, const float* mean, const float* stdv, float* dst, size_t n, int C, void* stream) {
    SAICV_REQUIRE(src && mean && stdv && dst && n > 0 && C >= 1 && C <= 4, "saicv_u8_normalize: bad arguments");
    hipLaunchKernelGGL(u8_normalize_kernel, dim3(sgrid(n)), dim3(256), 0, static_cast<hipStream_t>(stream), (const uint8_t*)src, mean, stdv,
                       dst, n, C);
    return saicv::check_launch("u8_normalize");
}

int saicv_random_erase(float* x, const saicv_erase_box* boxes, int nboxes, int B, int H, int W, int C, unsigned int seed, void* stream) {
    SAICV_REQUIRE(x && boxes && nboxes > 0 && B > 0 && H > 0 && W > 0 && C >= 1 && C <= 4, "saicv_random_erase: bad arguments");
    // (the caller orders overlapping boxes of one image into separate calls: within a call boxes are disjoint or of different images)
    hipLaunchKernelGGL(random_erase_kernel, dim3(16, nboxes), dim3(256), 0, static_cast<hipStream_t>(stream), x, boxes, H, W, C, seed);
    return saicv::check_launch("random_erase");
}

int saicv_detr_sine_pe(const unsigned char* mask, float* out, int B, int H, int W, int F, float temperature, float eps, void* stream) {
    SAICV_REQUIRE(mask && out && B > 0 && H > 0 && W > 0 && F > 0, "saicv_detr_sine_pe: bad arguments");
    SAICV_REQUIRE((size_t)H * W * 8 <= 64 * 1024, "saicv_detr_sine_pe: %d x %d feature map (two fp32 planes must fit 64 KiB of LDS)", H, W);
    hipLaunchKernelGGL(detr_sine_pe_kernel, dim3(B, 32), dim3(256), (size_t)H * W * 8, static_cast<hipStream_t>(stream), mask, out, H, W, F,
                       temperature, eps);
    return saicv::check_launch("detr_sine_pe");
}

}  // extern "C"
