// Dense-detector training loss: what reference SimpleAICV/detection/losses.py RetinaLoss (:123-433)
// spends its time on, as three kernels over fp32 head outputs that stay where the heads wrote them (per pyramid level):
//   retina_assign   get_batch_anchors_annotations (:330-396): every anchor against every ground-truth box of its image, IoU >= 0.5
//                   -> class + 1, IoU < 0.4 -> background 0, between -> ignored -1; box target of the best box ([tx,ty,tw,th] for
//                   SmoothL1, :398-416).  Ground truth of the image sits in LDS; the IoU arithmetic is written operation by
//                   operation (no FMA contraction) so that the 0.4 / 0.5 decisions are the reference's.
//   focal_level     compute_batch_focal_loss (:222-262) on one level's [B][A_l][C] probabilities: loss sum AND its gradient in
//                   one pass (the torch formulation makes ~15 passes over B x A x C, 49 M elements for 8 images at 640 x 640).
//   smoothl1_level  compute_batch_smoothl1_loss (:305-328) on the positive anchors of one level, loss sum and gradient.
// Sums are accumulated with one fp32 atomic per workgroup into a zeroed buffer; the host divides by the positive count.
#include "common.h"
#include "saicv_internal.h"
#include "det.h"

namespace {

constexpr int DL_THREADS = 256;
constexpr int DL_MAX_GT = 1024;

// correctly rounded fp32 square root (the double result rounds to the fp32 one): scores and membership tests must equal numpy's / torch's
DEVINL float sqrt_exact(float v) { return (float)sqrt((double)v); }

DEVINL float block_sum(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x == 0) {
        for (int i = 0; i < DL_THREADS / 64; ++i) t += red[i];
    }
    __syncthreads();
    return t;     // valid in thread 0
}

__global__ __launch_bounds__(DL_THREADS) void retina_assign_kernel(const float* __restrict__ anchors, const float* __restrict__ annots,
                                                                   float* __restrict__ targets, float* __restrict__ pos_count,
                                                                   int A, int G, int smoothl1) {
    __shared__ float gt[DL_MAX_GT * 5];
    __shared__ float red[DL_THREADS / 64];
    __shared__ int ngt;
    const int b = blockIdx.y;
    // valid boxes of this image, in order (the reference drops class < 0 rows first, :345-346)
    if (threadIdx.x == 0) {
        int n = 0;
        for (int g = 0; g < G; ++g) {
            const float* r = annots + ((size_t)b * G + g) * 5;
            if (r[4] >= 0.f) {
                for (int j = 0; j < 5; ++j) gt[n * 5 + j] = r[j];
                ++n;
            }
        }
        ngt = n;
    }
    __syncthreads();
    const int n = ngt;
    const int a = blockIdx.x * DL_THREADS + threadIdx.x;
    float is_pos = 0.f;
    if (a < A) {
        const float ax1 = anchors[a * 4 + 0], ay1 = anchors[a * 4 + 1], ax2 = anchors[a * 4 + 2], ay2 = anchors[a * 4 + 3];
        float out[5] = {-1.f, -1.f, -1.f, -1.f, -1.f};
        if (n > 0) {
            const float aw = fmaxf(__fsub_rn(ax2, ax1), 0.f), ah = fmaxf(__fsub_rn(ay2, ay1), 0.f);
            const float a_area = __fmul_rn(aw, ah);
            float best = -1.f;
            int bi = 0;
            for (int g = 0; g < n; ++g) {
                const float gx1 = gt[g * 5 + 0], gy1 = gt[g * 5 + 1], gx2 = gt[g * 5 + 2], gy2 = gt[g * 5 + 3];
                const float ow = fmaxf(__fsub_rn(fminf(ax2, gx2), fmaxf(ax1, gx1)), 0.f);
                const float oh = fmaxf(__fsub_rn(fminf(ay2, gy2), fmaxf(ay1, gy1)), 0.f);
                const float overlap = __fmul_rn(ow, oh);
                const float gw = fmaxf(__fsub_rn(gx2, gx1), 0.f), gh = fmaxf(__fsub_rn(gy2, gy1), 0.f);
                const float uni = fmaxf(__fsub_rn(__fadd_rn(a_area, __fmul_rn(gw, gh)), overlap), 1e-4f);
                const float iou = __fdiv_rn(overlap, uni);
                if (iou > best) { best = iou; bi = g; }
            }
            float cls = -1.f;
            if (best < 0.4f) cls = 0.f;
            if (best >= 0.5f) cls = gt[bi * 5 + 4] + 1.f;
            const float gx1 = gt[bi * 5 + 0], gy1 = gt[bi * 5 + 1], gx2 = gt[bi * 5 + 2], gy2 = gt[bi * 5 + 3];
            if (smoothl1) {
                const float w = ax2 - ax1, h = ay2 - ay1;
                const float cx = ax1 + 0.5f * w, cy = ay1 + 0.5f * h;
                const float gw = fmaxf(gx2 - gx1, 1e-4f), gh = fmaxf(gy2 - gy1, 1e-4f);
                const float gcx = gx1 + 0.5f * gw, gcy = gy1 + 0.5f * gh;
                out[0] = (gcx - cx) / w;
                out[1] = (gcy - cy) / h;
                out[2] = logf(gw / w);
                out[3] = logf(gh / h);
            } else {
                out[0] = gx1; out[1] = gy1; out[2] = gx2; out[3] = gy2;
            }
            out[4] = cls;
            is_pos = cls > 0.f ? 1.f : 0.f;
        }
        float* t = targets + ((size_t)b * A + a) * 5;
#pragma unroll
        for (int j = 0; j < 5; ++j) t[j] = out[j];
    }
    const float cnt = block_sum(is_pos, red);
    if (threadIdx.x == 0 && cnt != 0.f) atomicAdd(pos_count, cnt);
}

// one level: probs [B][Al][C], targets [B][At][5] (this level's anchors start at `off`)
template <bool GAMMA2>
__global__ __launch_bounds__(DL_THREADS) void focal_level_kernel(const float* __restrict__ probs, const float* __restrict__ targets,
                                                                 float* __restrict__ dprobs, float* __restrict__ loss_sum, size_t total,
                                                                 int Al, int At, int off, int C, float alpha, float gamma, const saicv::DetSink det) {
    __shared__ float red[DL_THREADS / 64];
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * DL_THREADS + threadIdx.x; i < total; i += (size_t)gridDim.x * DL_THREADS) {
        const size_t row = i / C;
        const int c = (int)(i - row * C);
        const size_t b = row / Al;
        const int a = (int)(row - b * Al);
        const float cls = targets[(b * At + off + a) * 5 + 4];
        float grad = 0.f;
        if (cls >= 0.f) {
            const float p0 = probs[i];
            const float p = fminf(fmaxf(p0, 1e-4f), 1.f - 1e-4f);
            const bool live = p0 >= 1e-4f && p0 <= 1.f - 1e-4f;         // torch.clamp passes the gradient inside the range only
            const bool hot = cls > 0.f && (int)cls - 1 == c;
            const float q = hot ? p : 1.f - p;                           // probability of the true label
            const float w = hot ? alpha : 1.f - alpha;
            const float lq = logf(q);
            const float omq = 1.f - q;
            float mod, dmod;                                             // (1 - q)^gamma and its derivative by q
            if (GAMMA2) { mod = omq * omq; dmod = -2.f * omq; }
            else { mod = powf(omq, gamma); dmod = omq > 0.f ? -gamma * powf(omq, gamma - 1.f) : 0.f; }
            acc += -w * mod * lq;
            const float dq = -w * (dmod * lq + mod / q);                 // d loss / d q
            grad = live ? (hot ? dq : -dq) : 0.f;
        }
        if (dprobs) dprobs[i] = grad;
    }
    const float t = block_sum(acc, red);
    if (threadIdx.x == 0 && t != 0.f) saicv::det_add(det, loss_sum, 0, blockIdx.x, t);
}

// one level: reg [B][Al][4] against targets[..][0:4] on the rows with class > 0
__global__ __launch_bounds__(DL_THREADS) void smoothl1_level_kernel(const float* __restrict__ reg, const float* __restrict__ targets,
                                                                    float* __restrict__ dreg, float* __restrict__ loss_sum, size_t rows,
                                                                    int Al, int At, int off, float beta, const saicv::DetSink det) {
    __shared__ float red[DL_THREADS / 64];
    float acc = 0.f;
    for (size_t row = (size_t)blockIdx.x * DL_THREADS + threadIdx.x; row < rows; row += (size_t)gridDim.x * DL_THREADS) {
        const size_t b = row / Al;
        const int a = (int)(row - b * Al);
        const float* t = targets + (b * At + off + a) * 5;
        const f32x4 r = *reinterpret_cast<const f32x4*>(reg + row * 4);
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        if (t[4] > 0.f) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = r[j] - t[j];
                const float x = fabsf(d);
                if (x >= beta) { acc += x - 0.5f * beta; g[j] = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); }
                else { acc += 0.5f * x * x / beta; g[j] = d / beta; }
            }
        }
        if (dreg) *reinterpret_cast<f32x4*>(dreg + row * 4) = g;
    }
    const float s = block_sum(acc, red);
    if (threadIdx.x == 0 && s != 0.f) saicv::det_add(det, loss_sum, 0, blockIdx.x, s);
}


// FCOSLoss.get_batch_position_annotations (losses.py:623-842): every feature-map point against every ground-truth box of its
// image.  A box is a candidate for a point when the point lies strictly inside it, within `radius` strides of its centre (centre
// sampling) and the largest of the four side distances falls inside the level's regression range (m0, m1); the candidate with the
// smallest box area wins.  points [P][5] = (x, y, stride, m0, m1).  targets [B][P][5] = (l, t, r, b, class + 1 | 0),
// centerness [B][P] = sqrt(min(l, r) / max(l, r) * min(t, b) / max(t, b)) | 0.  Same operation order as the reference (no FMA
// contraction in the tests that decide membership).
__global__ __launch_bounds__(DL_THREADS) void fcos_assign_kernel(const float* __restrict__ points, const float* __restrict__ annots,
                                                                 float* __restrict__ targets, float* __restrict__ centerness,
                                                                 float* __restrict__ pos_count, int P, int G, float radius, int center_sample) {
    __shared__ float gt[DL_MAX_GT * 5];
    __shared__ float red[DL_THREADS / 64];
    __shared__ int ngt;
    const int b = blockIdx.y;
    if (threadIdx.x == 0) {
        int n = 0;
        for (int g = 0; g < G; ++g) {
            const float* r = annots + ((size_t)b * G + g) * 5;
            if (r[4] >= 0.f) {
                for (int j = 0; j < 5; ++j) gt[n * 5 + j] = r[j];
                ++n;
            }
        }
        ngt = n;
    }
    __syncthreads();
    const int n = ngt;
    const int p = blockIdx.x * DL_THREADS + threadIdx.x;
    float is_pos = 0.f;
    if (p < P) {
        const float x = points[p * 5 + 0], y = points[p * 5 + 1], stride = points[p * 5 + 2], m0 = points[p * 5 + 3], m1 = points[p * 5 + 4];
        const float judge = __fmul_rn(stride, radius);
        float out[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        float ctr = 0.f, best_area = 0.f;
        bool found = false;
        for (int g = 0; g < n; ++g) {
            const float x1 = gt[g * 5 + 0], y1 = gt[g * 5 + 1], x2 = gt[g * 5 + 2], y2 = gt[g * 5 + 3];
            const float l = __fsub_rn(x, x1), t = __fsub_rn(y, y1), r = __fsub_rn(x2, x), bt = __fsub_rn(y2, y);
            const float mn = fminf(fminf(l, t), fminf(r, bt));
            if (!(mn > 0.f)) continue;
            if (center_sample) {
                const float cx = __fdiv_rn(__fadd_rn(x2, x1), 2.f), cy = __fdiv_rn(__fadd_rn(y2, y1), 2.f);
                const float dx = __fsub_rn(x, cx), dy = __fsub_rn(y, cy);
                const float dist = sqrt_exact(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
                if (!(dist < judge)) continue;
            }
            const float mx = fmaxf(fmaxf(l, t), fmaxf(r, bt));
            if (!(mx > m0) || !(mx < m1)) continue;
            const float area = __fmul_rn(__fsub_rn(x2, x1), __fsub_rn(y2, y1));
            if (!found || area < best_area) {
                found = true;
                best_area = area;
                out[0] = l; out[1] = t; out[2] = r; out[3] = bt;
                out[4] = gt[g * 5 + 4] + 1.f;
            }
        }
        if (found) {
            const float l = out[0], t = out[1], r = out[2], bt = out[3];
            ctr = sqrt_exact(__fmul_rn(__fdiv_rn(fminf(l, r), fmaxf(l, r)), __fdiv_rn(fminf(t, bt), fmaxf(t, bt))));
            is_pos = 1.f;
        }
        float* o = targets + ((size_t)b * P + p) * 5;
#pragma unroll
        for (int j = 0; j < 5; ++j) o[j] = out[j];
        centerness[(size_t)b * P + p] = ctr;
    }
    const float cnt = block_sum(is_pos, red);
    if (threadIdx.x == 0 && cnt != 0.f) atomicAdd(pos_count, cnt);
}

// Evaluation-time candidate scores of one pyramid level (reference SimpleAICV/detection/decode.py RetinaDecoder :219-230,
// FCOSDecoder :331-343): per anchor / point the first-maximum class of its C probabilities and that probability -- for FCOS
// sqrt(probability * centre-ness) -- so that only [B][At] scores and classes (and later the few hundred surviving candidates)
// leave the device instead of the [B][At][C] tensor the reference copies to the host.
__global__ __launch_bounds__(DL_THREADS) void best_class_kernel(const float* __restrict__ probs, const float* __restrict__ centerness,
                                                                float* __restrict__ scores, int* __restrict__ classes, size_t rows, int Al,
                                                                int At, int off, int C) {
    for (size_t row = (size_t)blockIdx.x * DL_THREADS + threadIdx.x; row < rows; row += (size_t)gridDim.x * DL_THREADS) {
        const float* p = probs + row * C;
        float best = p[0];
        int bc = 0;
        for (int c = 1; c < C; ++c) {
            const float v = p[c];
            if (v > best) { best = v; bc = c; }
        }
        if (centerness) best = sqrt_exact(__fmul_rn(best, centerness[row]));
        const size_t b = row / Al;
        const size_t o = b * At + off + (row - b * Al);
        scores[o] = best;
        classes[o] = bc;
    }
}

inline int dl_grid(size_t items) {
    size_t g = (items + DL_THREADS - 1) / DL_THREADS;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}


// ---------------------------------------------------------------- DETR: the Hungarian assignment on the device (r05)
// Reference SimpleAICV/detection/losses.py:1009-1090 runs scipy.optimize.linear_sum_assignment per image on the host, between forward
// and loss: a device -> host copy, a synchronisation and a host -> device copy in the middle of every step -- and the reason the step
// could not be ONE captured graph.  This kernel is scipy's algorithm itself (scipy/optimize/rectangular_lsap/rectangular_lsap.cpp,
// v1.15: Crouse's shortest augmenting path, the `remaining` list filled in reverse and swap-removed, ties towards a new sink) restated
// in double precision, one workgroup per image, the scan over the remaining columns spread over the 64 lanes of one wavefront with the
// sequential scan's tie rule reproduced exactly (among equal minima: the LAST unassigned column in scan order if there is one, else the
// FIRST).  cost [B][Q][T] fp32 (queries x padded ground truth), valid [B][T]: the assignment is over the valid columns only;
// a tall matrix (fewer targets than queries, the usual case) is transposed as scipy does.  nan -> 1e5 and one-signed infinities -> a
// finite value beyond any achievable total, as the reference's wrapper (losses.py:1063-1090) does.
// Output, per image: pairs (src = query, tgt = ground-truth row of the padded tensor) in slots [0, min(n, Q)), weight 1; 0 elsewhere.
__global__ __launch_bounds__(64) void detr_assign_kernel(const float* __restrict__ cost, const unsigned char* __restrict__ valid, int Q, int T,
                                                         long long* __restrict__ src, long long* __restrict__ tgt, float* __restrict__ wgt) {
    extern __shared__ __attribute__((aligned(16))) char lsa_smem[];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int NMAX = Q > T ? Q : T;
    double* u = reinterpret_cast<double*>(lsa_smem);           // [NMAX] row duals
    double* v = u + NMAX;                                       // [NMAX] column duals
    double* spc = v + NMAX;                                     // [NMAX] shortest path costs
    int* path = reinterpret_cast<int*>(spc + NMAX);             // [NMAX]
    int* col4row = path + NMAX;
    int* row4col = col4row + NMAX;
    int* remaining = row4col + NMAX;
    int* cols = remaining + NMAX;                               // [T] valid ground-truth rows
    unsigned char* SR = reinterpret_cast<unsigned char*>(cols + NMAX);
    unsigned char* SC = SR + NMAX;
    __shared__ int sh_n;
    const float* cb = cost + (size_t)b * Q * T;
    if (lane == 0) {
        int n = 0;
        for (int t = 0; t < T; ++t)
            if (valid[(size_t)b * T + t]) cols[n++] = t;
        sh_n = n;
    }
    for (int t = lane; t < T; t += 64) { src[(size_t)b * T + t] = 0; tgt[(size_t)b * T + t] = 0; wgt[(size_t)b * T + t] = 0.f; }
    __syncthreads();
    const int n = sh_n;
    if (n == 0) return;
    // ---- the reference wrapper's clean-up of nan / inf (per image block, over the valid columns)
    double lo = INFINITY, hi = -INFINITY;
    int has_neg = 0, has_pos = 0;
    for (int e = lane; e < Q * n; e += 64) {
        double c = (double)cb[(size_t)(e / n) * T + cols[e % n]];
        if (c != c) c = 1e5;
        if (c == INFINITY) has_pos = 1;
        else if (c == -INFINITY) has_neg = 1;
        else { lo = fmin(lo, c); hi = fmax(hi, c); }
    }
    for (int o = 32; o > 0; o >>= 1) {
        lo = fmin(lo, __shfl_xor(lo, o));
        hi = fmax(hi, __shfl_xor(hi, o));
        has_pos |= __shfl_xor(has_pos, o);
        has_neg |= __shfl_xor(has_neg, o);
    }
    double inf_fix = 0.0;
    if (has_pos || has_neg) {                                   // (both at once raise in the reference; here the positive rule wins)
        const double m = (double)(Q < n ? Q : n);
        const double positive = m * (hi - lo + fabs(hi) + fabs(lo) + 1.0);
        inf_fix = has_pos ? (hi + (m - 1.0) * (hi - lo)) + positive : (lo + (m - 1.0) * (lo - hi)) - positive;
    }
    const bool transpose = n < Q;                               // scipy: a tall matrix is transposed (rows = the smaller side)
    const int nr = transpose ? n : Q, nc = transpose ? Q : n;
    auto C = [&](int i, int j) -> double {                      // cost of (row i, column j) of the matrix being solved
        const int q = transpose ? j : i, k = transpose ? i : j;
        double c = (double)cb[(size_t)q * T + cols[k]];
        if (c != c) c = 1e5;
        if (c == INFINITY || c == -INFINITY) c = inf_fix;
        return c;
    };
    for (int i = lane; i < NMAX; i += 64) { u[i] = 0.0; v[i] = 0.0; path[i] = -1; col4row[i] = -1; row4col[i] = -1; }
    __syncthreads();
    for (int cur = 0; cur < nr; ++cur) {
        // ---- augmenting_path(cur)
        for (int j = lane; j < nc; j += 64) { remaining[j] = nc - j - 1; SC[j] = 0; spc[j] = INFINITY; }
        for (int i = lane; i < nr; i += 64) SR[i] = 0;
        __syncthreads();
        double minVal = 0.0;
        int num_remaining = nc, sink = -1, i = cur;
        while (sink == -1) {
            if (lane == 0) SR[i] = 1;
            // scan of the remaining columns, 64 at a time; per lane: the winner among its own positions under the sequential rule
            double best = INFINITY;
            int best_it = -1, best_free = 0;                    // best_free: the winner is an unassigned column
            const double ui = u[i];
            for (int it = lane; it < num_remaining; it += 64) {
                const int j = remaining[it];
                const double r = minVal + C(i, j) - ui - v[j];
                double sj = spc[j];
                if (r < sj) { path[j] = i; spc[j] = r; sj = r; }
                const int free_ = row4col[j] == -1;
                // sequential rule: take j if strictly lower, or equal and unassigned (a later unassigned column replaces an earlier one)
                if (sj < best || (sj == best && free_)) { best = sj; best_it = it; best_free = free_; }
            }
            // combine the lanes: lower value wins; among equal values an unassigned winner beats an assigned one; two unassigned:
            // the LATER position; two assigned: the EARLIER position -- what one sequential scan over all positions yields
            for (int o = 32; o > 0; o >>= 1) {
                const double ob = __shfl_xor(best, o);
                const int oi = __shfl_xor(best_it, o), of = __shfl_xor(best_free, o);
                bool take = false;
                if (oi >= 0) {
                    if (best_it < 0 || ob < best) take = true;
                    else if (ob == best) {
                        if (of && !best_free) take = true;
                        else if (of == best_free) take = of ? (oi > best_it) : (oi < best_it);
                    }
                }
                if (take) { best = ob; best_it = oi; best_free = of; }
            }
            minVal = best;
            if (!(minVal < INFINITY)) break;                    // infeasible (cannot happen after the clean-up): leave unassigned
            const int j = remaining[best_it];
            __syncthreads();                                    // every lane has read remaining[] / spc[] of this scan
            if (row4col[j] == -1) sink = j;
            else i = row4col[j];
            if (lane == 0) {
                SC[j] = 1;
                remaining[best_it] = remaining[num_remaining - 1];
            }
            --num_remaining;
            __syncthreads();
        }
        if (sink < 0) break;
        // ---- dual update, augmentation (serial parts: short)
        if (lane == 0) u[cur] += minVal;
        __syncthreads();
        for (int r_ = lane; r_ < nr; r_ += 64)
            if (SR[r_] && r_ != cur) u[r_] += minVal - spc[col4row[r_]];
        for (int j = lane; j < nc; j += 64)
            if (SC[j]) v[j] -= minVal - spc[j];
        __syncthreads();
        if (lane == 0) {
            int j = sink;
            while (true) {
                const int ii = path[j];
                row4col[j] = ii;
                const int tmp = col4row[ii];
                col4row[ii] = j;
                j = tmp;
                if (ii == cur) break;
            }
        }
        __syncthreads();
    }
    // ---- pairs.  transposed: row k = valid target k -> query col4row[k]; else row q = query -> target cols[col4row[q]]
    for (int r_ = lane; r_ < nr; r_ += 64) {
        const int c_ = col4row[r_];
        if (c_ < 0) continue;
        const size_t o = (size_t)b * T + r_;
        src[o] = transpose ? c_ : r_;
        tgt[o] = transpose ? cols[r_] : cols[c_];
        wgt[o] = 1.f;
    }
}


// ------------------------------------------------------------------------------------------------ DETR box losses (late r06)
// compute_batch_l1_iou_loss (reference SimpleAICV/detection/losses.py:938-954) over the static-shape pair buffers of DETRLoss.forward_static:
// for every decoder layer l, image b and pair k with w[b][k] > 0 the prediction reg[l][b][src[b][k]] (clamped to [lo, hi], cx cy w h)
// against the ground-truth row gt[b][tgt[b][k]][0:4]:  l1[l] = sum w * |p - t|_1 / n,  iou[l] = sum w * (1 - GIoU(p, t)) / n,  n = number
// of ground-truth rows with class >= 0 (0 / 0 = nan for a batch without boxes, as in the reference: the loop skips the step).  The torch
// formulation was ~55 launches forward and ~130 backward on 4 800 pairs.  One workgroup per layer, partial sums folded in a fixed order.
// The GIoU arithmetic follows losses._giou operation by operation (clamps included); its backward is the reverse-mode derivative of exactly
// those operations with ATen's subgradient choices (clamp passes the gradient where the input is inside the closed range, min / max split
// it on a tie).
struct GiouTape { float x1, y1, x2, y2, a10, iw, ih, cw, ch, i0, inter, u0, uni, ew0, eh0, ew, eh, e0, enc; };

#pragma clang fp contract(off)
DEVINL float giou_fwd(const float p[4], const float t[4], float tx[4], GiouTape& g) {
    g.x1 = p[0] - 0.5f * p[2]; g.y1 = p[1] - 0.5f * p[3]; g.x2 = p[0] + 0.5f * p[2]; g.y2 = p[1] + 0.5f * p[3];
    tx[0] = t[0] - 0.5f * t[2]; tx[1] = t[1] - 0.5f * t[3]; tx[2] = t[0] + 0.5f * t[2]; tx[3] = t[1] + 0.5f * t[3];
    g.a10 = (g.x2 - g.x1) * (g.y2 - g.y1);
    const float a1 = fmaxf(g.a10, 0.f);
    const float a2 = fmaxf((tx[2] - tx[0]) * (tx[3] - tx[1]), 0.f);
    g.iw = fminf(g.x2, tx[2]) - fmaxf(g.x1, tx[0]);
    g.ih = fminf(g.y2, tx[3]) - fmaxf(g.y1, tx[1]);
    g.cw = fmaxf(g.iw, 0.f); g.ch = fmaxf(g.ih, 0.f);
    g.i0 = g.cw * g.ch;
    g.inter = fmaxf(g.i0, 0.f);
    g.u0 = a1 + a2 - g.inter;
    g.uni = fmaxf(g.u0, 1e-4f);
    g.ew0 = fmaxf(g.x2, tx[2]) - fminf(g.x1, tx[0]);
    g.eh0 = fmaxf(g.y2, tx[3]) - fminf(g.y1, tx[1]);
    g.ew = fmaxf(g.ew0, 0.f); g.eh = fmaxf(g.eh0, 0.f);
    g.e0 = g.ew * g.eh;
    g.enc = fmaxf(g.e0, 1e-4f);
    return g.inter / g.uni - (g.enc - g.uni) / g.enc;
}

// d giou / d p (cx, cy, w, h)
DEVINL void giou_bwd(const float tx[4], const GiouTape& g, float dp[4]) {
    // giou = inter / uni - (enc - uni) / enc
    float d_inter = 1.f / g.uni;
    const float d_uni = -g.inter / (g.uni * g.uni) + 1.f / g.enc;
    const float d_enc = -1.f / g.enc + (g.enc - g.uni) / (g.enc * g.enc);
    const float d_u0 = g.u0 >= 1e-4f ? d_uni : 0.f;
    const float d_a1 = d_u0;
    d_inter -= d_u0;
    const float d_i0 = g.i0 >= 0.f ? d_inter : 0.f;
    const float d_iw = g.iw >= 0.f ? d_i0 * g.ch : 0.f;
    const float d_ih = g.ih >= 0.f ? d_i0 * g.cw : 0.f;
    const float d_e0 = g.e0 >= 1e-4f ? d_enc : 0.f;
    const float d_ew0 = g.ew0 >= 0.f ? d_e0 * g.eh : 0.f;
    const float d_eh0 = g.eh0 >= 0.f ? d_e0 * g.ew : 0.f;
    const float d_a10 = g.a10 >= 0.f ? d_a1 : 0.f;
    // share of the first argument in min(a, b) / max(a, b): 1 if it is selected, 1/2 on a tie
    auto sel_min = [](float a, float b) { return a < b ? 1.f : (a == b ? 0.5f : 0.f); };
    auto sel_max = [](float a, float b) { return a > b ? 1.f : (a == b ? 0.5f : 0.f); };
    const float dx2 = d_iw * sel_min(g.x2, tx[2]) + d_ew0 * sel_max(g.x2, tx[2]) + d_a10 * (g.y2 - g.y1);
    const float dx1 = -d_iw * sel_max(g.x1, tx[0]) - d_ew0 * sel_min(g.x1, tx[0]) - d_a10 * (g.y2 - g.y1);
    const float dy2 = d_ih * sel_min(g.y2, tx[3]) + d_eh0 * sel_max(g.y2, tx[3]) + d_a10 * (g.x2 - g.x1);
    const float dy1 = -d_ih * sel_max(g.y1, tx[1]) - d_eh0 * sel_min(g.y1, tx[1]) - d_a10 * (g.x2 - g.x1);
    dp[0] = dx1 + dx2; dp[1] = dy1 + dy2; dp[2] = 0.5f * (dx2 - dx1); dp[3] = 0.5f * (dy2 - dy1);
}
#pragma clang fp contract(on)

DEVINL float clamp_keep_nan(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }      // (a nan stays a nan, as torch.clamp)

// fixed-order sum of one value per thread over the workgroup (DL_THREADS threads) -> every thread gets the total
DEVINL float block_total(float v, float* red) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = DL_THREADS / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    const float t = red[0];
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(DL_THREADS) void detr_box_loss_fwd_kernel(const float* __restrict__ reg, const float* __restrict__ gt,
                                                                       const long long* __restrict__ src, const long long* __restrict__ tgt,
                                                                       const float* __restrict__ w, int L, int B, int Q, int T, float lo, float hi,
                                                                       float* __restrict__ out) {
    __shared__ float red[DL_THREADS];
    const int l = blockIdx.x;
    float s_l1 = 0.f, s_iou = 0.f, n = 0.f;
    for (int i = threadIdx.x; i < B * T; i += DL_THREADS) {
        const int b = i / T;
        if (gt[(size_t)i * 5 + 4] >= 0.f) n += 1.f;
        const float wv = w[i];
        if (wv > 0.f) {
            const float* pr = reg + (((size_t)l * B + b) * Q + src[i]) * 4;
            const float* tr = gt + ((size_t)b * T + tgt[i]) * 5;
            float p[4], t[4], tx[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { p[j] = clamp_keep_nan(pr[j], lo, hi); t[j] = tr[j]; }
            GiouTape g;
            const float gi = giou_fwd(p, t, tx, g);
            s_l1 += wv * (fabsf(p[0] - t[0]) + fabsf(p[1] - t[1]) + fabsf(p[2] - t[2]) + fabsf(p[3] - t[3]));
            s_iou += wv * (1.f - gi);
        }
    }
    const float t_l1 = block_total(s_l1, red), t_iou = block_total(s_iou, red), t_n = block_total(n, red);
    if (threadIdx.x == 0) {
        out[l] = t_l1 / t_n;
        out[L + l] = t_iou / t_n;
        if (l == 0) out[2 * L] = t_n;
    }
}

// one thread per (layer, pair): the pair's gradient row; dreg was zeroed in front (a query is matched at most once per image, so the rows
// are disjoint and every other query row keeps its zeros)
__global__ __launch_bounds__(DL_THREADS) void detr_box_loss_bwd_kernel(const float* __restrict__ reg, const float* __restrict__ gt,
                                                                       const long long* __restrict__ src, const long long* __restrict__ tgt,
                                                                       const float* __restrict__ w, const float* __restrict__ d_l1,
                                                                       const float* __restrict__ d_iou, const float* __restrict__ out, int L, int B,
                                                                       int Q, int T, float lo, float hi, float* __restrict__ dreg) {
    const int i = blockIdx.x * DL_THREADS + threadIdx.x, l = blockIdx.y;
    if (i >= B * T) return;
    const float wv = w[i];
    if (!(wv > 0.f)) return;
    const int b = i / T;
    const float inv_n = 1.f / out[2 * L];
    const float g_l1 = (d_l1 ? d_l1[l] : 0.f) * inv_n, g_iou = (d_iou ? d_iou[l] : 0.f) * inv_n;
    const size_t row = (((size_t)l * B + b) * Q + src[i]) * 4;
    const float* tr = gt + ((size_t)b * T + tgt[i]) * 5;
    float raw[4], p[4], t[4], tx[4], dg[4], o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { raw[j] = reg[row + j]; p[j] = clamp_keep_nan(raw[j], lo, hi); t[j] = tr[j]; }
    GiouTape g;
    (void)giou_fwd(p, t, tx, g);
    giou_bwd(tx, g, dg);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float sg = p[j] > t[j] ? 1.f : (p[j] < t[j] ? -1.f : 0.f);
        const float dp = wv * (g_l1 * sg - g_iou * dg[j]);
        o[j] = (raw[j] >= lo && raw[j] <= hi) ? dp : 0.f;                                   // clamp's gradient mask
    }
    *reinterpret_cast<float4*>(dreg + row) = make_float4(o[0], o[1], o[2], o[3]);
}

}  // namespace

extern "C" {

// anchors [A][4] fp32 (x_min, y_min, x_max, y_max; one image's table, shared by the batch), annots [B][G][5] fp32 (box, class;
// class < 0 = padding row) -> targets [B][A][5] (box target, class target -1 / 0 / class + 1), pos_count[0] += positives.
// Reference losses.py:330-416.  G <= 1024.
int saicv_retina_assign(const float* anchors, const float* annots, float* targets, float* pos_count, int B, int A, int G,
                        int smoothl1, void* stream) {
    SAICV_REQUIRE(B > 0 && A > 0 && G >= 0 && G <= DL_MAX_GT, "retina_assign: B=%d A=%d G=%d (G <= %d)", B, A, G, DL_MAX_GT);
    SAICV_REQUIRE(B <= 65535, "retina_assign: batch %d", B);
    hipLaunchKernelGGL(retina_assign_kernel, dim3((A + DL_THREADS - 1) / DL_THREADS, B), dim3(DL_THREADS), 0, (hipStream_t)stream, anchors,
                       annots, targets, pos_count, A, G, smoothl1);
    return saicv::check_launch("retina_assign");
}

// focal loss of one pyramid level: probs [B][Al][C] fp32 against targets [B][At][5] (level offset `off`): loss_sum[0] += the
// un-normalised sum (losses.py:244-258 before the division), dprobs (optional) = its gradient by probs
int saicv_focal_loss_level(const float* probs, const float* targets, float* dprobs, float* loss_sum, int B, int Al, int At, int off,
                           int C, double alpha, double gamma, void* stream) {
    SAICV_REQUIRE(B > 0 && Al > 0 && C > 0 && off >= 0 && off + Al <= At, "focal_loss_level: B=%d Al=%d At=%d off=%d C=%d", B, Al, At, off, C);
    const size_t total = (size_t)B * Al * C;
    // (the positive counts of the assignment kernels stay atomic: sums of small integers are exact in fp32 in any order)
    saicv::DetParts det;
    if (det.begin((hipStream_t)stream, dl_grid(total), 1, "focal_loss_level")) return -1;
    if (gamma == 2.0)
        hipLaunchKernelGGL((focal_level_kernel<true>), dim3(dl_grid(total)), dim3(DL_THREADS), 0, (hipStream_t)stream, probs, targets, dprobs,
                           loss_sum, total, Al, At, off, C, (float)alpha, (float)gamma, det.sink());
    else
        hipLaunchKernelGGL((focal_level_kernel<false>), dim3(dl_grid(total)), dim3(DL_THREADS), 0, (hipStream_t)stream, probs, targets, dprobs,
                           loss_sum, total, Al, At, off, C, (float)alpha, (float)gamma, det.sink());
    if (saicv::check_launch("focal_loss_level")) return -2;
    return det.fold(loss_sum, 0, 1);
}

// SmoothL1 box loss of one level's positive anchors: reg [B][Al][4] (16-byte aligned) against targets[..][0:4]; loss_sum[0] +=
// the un-normalised sum (losses.py:318-326 before the division), dreg (optional) = its gradient by reg
int saicv_smoothl1_level(const float* reg, const float* targets, float* dreg, float* loss_sum, int B, int Al, int At, int off,
                         double beta, void* stream) {
    SAICV_REQUIRE(B > 0 && Al > 0 && off >= 0 && off + Al <= At && beta > 0., "smoothl1_level: B=%d Al=%d At=%d off=%d", B, Al, At, off);
    SAICV_REQUIRE(((uintptr_t)reg & 15) == 0 && ((uintptr_t)dreg & 15) == 0, "smoothl1_level: box tensors must be 16-byte aligned");
    const size_t rows = (size_t)B * Al;
    saicv::DetParts det;
    if (det.begin((hipStream_t)stream, dl_grid(rows), 1, "smoothl1_level")) return -1;
    hipLaunchKernelGGL(smoothl1_level_kernel, dim3(dl_grid(rows)), dim3(DL_THREADS), 0, (hipStream_t)stream, reg, targets, dreg, loss_sum,
                       rows, Al, At, off, (float)beta, det.sink());
    if (saicv::check_launch("smoothl1_level")) return -2;
    return det.fold(loss_sum, 0, 1);
}

// FCOS point assignment (reference losses.py:623-842): points [P][5] fp32 = (x, y, stride, range low, range high) of one image's
// pyramid, annots [B][G][5] -> targets [B][P][5] = (l, t, r, b, class + 1 or 0), centerness [B][P], pos_count[0] += positives.
int saicv_fcos_assign(const float* points, const float* annots, float* targets, float* centerness, float* pos_count, int B, int P,
                      int G, double radius, int center_sample, void* stream) {
    SAICV_REQUIRE(B > 0 && P > 0 && G >= 0 && G <= DL_MAX_GT, "fcos_assign: B=%d P=%d G=%d (G <= %d)", B, P, G, DL_MAX_GT);
    SAICV_REQUIRE(B <= 65535, "fcos_assign: batch %d", B);
    hipLaunchKernelGGL(fcos_assign_kernel, dim3((P + DL_THREADS - 1) / DL_THREADS, B), dim3(DL_THREADS), 0, (hipStream_t)stream, points, annots,
                       targets, centerness, pos_count, P, G, (float)radius, center_sample);
    return saicv::check_launch("fcos_assign");
}

// per-anchor best class and score of one level: probs [B][Al][C] fp32, centerness NULL or [B][Al] fp32 (FCOS: score =
// sqrt(probability * centre-ness)) -> scores / classes [B][At] at this level's offset.  decode.py:219-230, :331-343.
int saicv_det_best_class(const float* probs, const float* centerness, float* scores, int* classes, int B, int Al, int At, int off, int C,
                         void* stream) {
    SAICV_REQUIRE(B > 0 && Al > 0 && C > 0 && off >= 0 && off + Al <= At, "det_best_class: B=%d Al=%d At=%d off=%d C=%d", B, Al, At, off, C);
    const size_t rows = (size_t)B * Al;
    hipLaunchKernelGGL(best_class_kernel, dim3(dl_grid(rows)), dim3(DL_THREADS), 0, (hipStream_t)stream, probs, centerness, scores, classes,
                       rows, Al, At, off, C);
    return saicv::check_launch("det_best_class");
}

// cost fp32 [B][Q][T], valid u8 / bool [B][T] -> src, tgt int64 [B][T], w fp32 [B][T]  (see detr_assign_kernel)
int saicv_detr_assign(const float* cost, const unsigned char* valid, int B, int Q, int T, long long* src, long long* tgt, float* w, void* stream) {
    SAICV_REQUIRE(cost && valid && src && tgt && w && B > 0 && Q > 0 && T > 0, "saicv_detr_assign: bad arguments");
    const int nmax = Q > T ? Q : T;
    SAICV_REQUIRE(nmax <= 2048, "saicv_detr_assign: %d queries / %d ground-truth rows (at most 2048)", Q, T);
    const size_t smem = (size_t)nmax * (3 * sizeof(double) + 5 * sizeof(int) + 2);
    if (smem > 64 * 1024) {       // 46 bytes per row: beyond 1 424 rows the dynamic LDS passes the 64 KiB a kernel gets without asking
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(detr_assign_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        SAICV_REQUIRE(e == hipSuccess, "saicv_detr_assign: %zu bytes of LDS for %d rows: %s", smem, nmax, hipGetErrorString(e));
    }
    hipLaunchKernelGGL(detr_assign_kernel, dim3(B), dim3(64), smem, static_cast<hipStream_t>(stream), cost, valid, Q, T, src, tgt, w);
    return saicv::check_launch("detr_assign");
}


int saicv_detr_box_loss_fwd(const float* reg, const float* gt, const long long* src, const long long* tgt, const float* w, int L, int B, int Q,
                            int T, double lo, double hi, float* out, void* stream) {
    SAICV_REQUIRE(L >= 1 && B >= 1 && Q >= 1 && T >= 1, "detr_box_loss_fwd: empty problem (L=%d B=%d Q=%d T=%d)", L, B, Q, T);
    hipLaunchKernelGGL(detr_box_loss_fwd_kernel, dim3(L), dim3(DL_THREADS), 0, (hipStream_t)stream, reg, gt, src, tgt, w, L, B, Q, T, (float)lo,
                       (float)hi, out);
    return saicv::check_launch("detr_box_loss_fwd");
}

int saicv_detr_box_loss_bwd(const float* reg, const float* gt, const long long* src, const long long* tgt, const float* w, const float* d_l1,
                            const float* d_iou, const float* out, int L, int B, int Q, int T, double lo, double hi, float* dreg, void* stream) {
    SAICV_REQUIRE(L >= 1 && B >= 1 && Q >= 1 && T >= 1, "detr_box_loss_bwd: empty problem (L=%d B=%d Q=%d T=%d)", L, B, Q, T);
    if (hipMemsetAsync(dreg, 0, (size_t)L * B * Q * 4 * sizeof(float), (hipStream_t)stream) != hipSuccess) {
        saicv::set_error("detr_box_loss_bwd: clearing the gradient failed: %s", hipGetErrorString(hipGetLastError()));
        return -1;
    }
    hipLaunchKernelGGL(detr_box_loss_bwd_kernel, dim3((B * T + DL_THREADS - 1) / DL_THREADS, L), dim3(DL_THREADS), 0, (hipStream_t)stream, reg, gt,
                       src, tgt, w, d_l1, d_iou, out, L, B, Q, T, (float)lo, (float)hi, dreg);
    return saicv::check_launch("detr_box_loss_bwd");
}

}  // extern "C"
