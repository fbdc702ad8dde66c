// Deterministic (bit-reproducible run to run) accumulation for every kernel that otherwise adds workgroup partials into a
// gradient / statistics tensor with fp32 atomics.
//
// The reference asks its backend for deterministic kernels (tools/utils.py:95-107: cudnn.deterministic = True).  Here the fast
// default lets the workgroups of a reduction add their partials with `unsafeAtomicAdd` in whatever order they finish; fp32 addition
// is not associative, so a gradient differs in its last bits from run to run and SGD amplifies that over iterations.  With
// saicv_set_deterministic(1) (tools.utils.set_seed() does it, SAICV_DETERMINISTIC=1 does it at import) every such kernel instead
// WRITES partial p of output element i to part[p * n + i] in a library-owned workspace, and a second launch folds the partials in
// index order: dst[i] += ((part[0][i] + part[1][i]) + part[2][i]) + ...  -- the ordered two-pass reduction.  Which workgroup
// produces partial p is a function of the launch geometry only, and the geometry is a function of the problem shape only.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace saicv {

extern int g_deterministic;

// device side: where a kernel puts one partial.  part == nullptr: the fast path (atomics straight into dst).
struct DetSink {
    float* part;     // [nparts][n]
    size_t n;
};

// dst_elem: the element the fast path adds into; (part_id, part_idx): where the ordered path parks the same value
__device__ __forceinline__ void det_add(const DetSink& s, float* dst_elem, size_t part_idx, int part_id, float v) {
    if (s.part) s.part[(size_t)part_id * s.n + part_idx] = v;
    else unsafeAtomicAdd(dst_elem, v);
}

// host side of one reduction.  begin(): in deterministic mode with more than one contributing part, takes nparts * n floats of the
// stream's workspace and zeroes them (a kernel need not write the elements it does not own; `zero = false` for a kernel that
// writes every element of every part); otherwise leaves sink().part null.
// fold(dst, offset, count): dst[i] += sum over p of part[p][offset + i], p ascending, for i in [0, count).  `wide` (r06, for the
// weight gradients: many parts over up to millions of elements, nothing else in the step reads the result): eight threads share an
// element quad, thread j adds parts j, j + 8, ... in that order and the eight sub-sums are added in the order j = 0 .. 7 -- another
// fixed association of the same sum, 2 ms of a ResNet-50 step faster than one thread walking every part (csrc/det.hip).
struct DetParts {
    DetSink s{nullptr, 0};
    int nparts = 0;
    hipStream_t st = nullptr;
    int begin(hipStream_t stream, int parts, size_t n, const char* who, bool zero = true);
    int fold(float* dst, size_t offset, size_t count, bool wide = false) const;
    bool on() const { return s.part != nullptr; }
    const DetSink& sink() const { return s; }
};

}  // namespace saicv
