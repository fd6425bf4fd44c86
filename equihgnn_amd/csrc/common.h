// Shared helpers for the gfx950 kernels of libequihgnn_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "equihgnn_hip.h"

#define EQH_WAVE 64

#define EQH_CHECK_LAUNCH()                                  \
    do {                                                    \
        if (hipGetLastError() != hipSuccess) return EQH_ERR_LAUNCH; \
    } while (0)

static inline bool eqh_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static inline int eqh_grid_for(int64_t work_items, int per_block, int cap) {
    int64_t g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return static_cast<int>(g);
}

__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void f4_fma(float4& a, const float4& v, float w) {
    a.x = fmaf(v.x, w, a.x); a.y = fmaf(v.y, w, a.y); a.z = fmaf(v.z, w, a.z); a.w = fmaf(v.w, w, a.w);
}
__device__ __forceinline__ void f4_add(float4& a, const float4& v) {
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
}

// Zero-fill by kernel.  hipMemsetAsync is avoided everywhere in this library: memset nodes of a
// stream-captured hipGraph were observed (ROCm 7.2, gfx950) not to clear again on replay.
static __global__ void eqh_k_zero(float* __restrict__ p, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = 0.f;
}
static inline int eqh_zero_async(float* p, int64_t n, hipStream_t stream) {
    if (n <= 0) return EQH_OK;
    hipLaunchKernelGGL(eqh_k_zero, dim3(eqh_grid_for(n, 256, 1024)), dim3(256), 0, stream, p, n);
    return hipGetLastError() == hipSuccess ? EQH_OK : EQH_ERR_LAUNCH;
}

// out[e] = sum_b slab[b][e] for e < elems, summed in a FIXED order (bitwise reproducible).  A block
// covers 64 consecutive elements with 16 slab-groups (b = g, g+16, ...); each group keeps 4 independent
// partial sums and issues eight loads before the first add (the loop is latency-, not bandwidth-bound:
// the typical call reduces 100-300 slabs of a few hundred floats), and the groups are combined through
// LDS in group order.
// Output: up to three segments (out0: elements [0, len0), out1: the next len1, out2: the rest; out1 ==
// nullptr means one contiguous run of `elems`), overwritten or, with accumulate != 0, added to (gradient
// accumulators of parameters that are used several times per step).
struct EqhReduceDesc {
    const float* slab;
    float* out0;
    float* out1;
    float* out2;
    int64_t elems, len0, len1;
    int64_t out_ld;   // row_len > 0: element e goes to out0[(e / row_len) * out_ld + e % row_len] (2-D block)
    int row_len;
    int n_slabs;
    int first_block;  // eqh_k_reduce_many: first block of this descriptor in the batched grid
};

// one 64-element chunk (starting at e0) of one reduction, by a 1024-thread block
static __device__ __forceinline__ void eqh_reduce_chunk(const EqhReduceDesc& d, int64_t e0, int accumulate,
                                                        float* s_part) {
    const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int64_t e = e0 + col;
    const int64_t elems = d.elems;
    const int n_slabs = d.n_slabs;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (e < elems) {
        const float* __restrict__ p = d.slab + e;
        int b = grp;
        for (; b + 112 < n_slabs; b += 128) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = p[(int64_t)(b + 16 * i) * elems];
            a0 += v[0]; a1 += v[1]; a2 += v[2]; a3 += v[3];
            a0 += v[4]; a1 += v[5]; a2 += v[6]; a3 += v[7];
        }
        {   // ragged end: same accumulator assignment (i mod 4), loads still issued together
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (b + 16 * i < n_slabs) ? p[(int64_t)(b + 16 * i) * elems] : 0.f;
            a0 += v[0]; a1 += v[1]; a2 += v[2]; a3 += v[3];
            a0 += v[4]; a1 += v[5]; a2 += v[6]; a3 += v[7];
        }
    }
    s_part[threadIdx.x] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (grp == 0 && e < elems) {
        float t = s_part[col];
#pragma unroll
        for (int g = 1; g < 16; ++g) t += s_part[64 * g + col];
        float* dst = d.out0 + (d.row_len > 0 ? (e / d.row_len) * d.out_ld + (e % d.row_len) : e);
        if (d.out1 != nullptr && e >= d.len0)
            dst = (e < d.len0 + d.len1) ? d.out1 + (e - d.len0) : d.out2 + (e - d.len0 - d.len1);
        *dst = accumulate ? *dst + t : t;
    }
    __syncthreads();
}

static __global__ void __launch_bounds__(1024)
eqh_k_reduce_slabs(EqhReduceDesc d, int accumulate) {
    __shared__ float s_part[1024];
    for (int64_t e0 = (int64_t)blockIdx.x * 64; e0 < d.elems; e0 += (int64_t)gridDim.x * 64)
        eqh_reduce_chunk(d, e0, accumulate, s_part);
}

// Deferred reductions (api.hip).  Between eqh_defer_begin(stream) and eqh_defer_flush(stream) every
// ACCUMULATING slab reduction issued on that stream is recorded instead of launched, and the flush
// runs them all in one launch: the accumulators (parameter gradients) are only read by the optimiser,
// so ~25 five-microsecond launches per training step become one.  Returns true if recorded.
bool eqh_defer_try(hipStream_t stream, const EqhReduceDesc& d);

static inline int eqh_reduce_slabs3_async(const float* slab, int n_slabs, int64_t elems, float* out0, float* out1,
                                          float* out2, int64_t len0, int64_t len1, int accumulate,
                                          hipStream_t stream) {
    EqhReduceDesc d{slab, out0, out1, out2, elems, len0, len1, 0, 0, n_slabs, 0};
    if (accumulate && eqh_defer_try(stream, d)) return EQH_OK;
    hipLaunchKernelGGL(eqh_k_reduce_slabs, dim3(eqh_grid_for(elems, 64, 2048)), dim3(1024), 0, stream, d, accumulate);
    return hipGetLastError() == hipSuccess ? EQH_OK : EQH_ERR_LAUNCH;
}
// 2-D destination block: rows of row_len elements, out_ld apart
static inline int eqh_reduce_slabs2d_async(const float* slab, int n_slabs, int64_t rows, int row_len, float* out,
                                           int64_t out_ld, int accumulate, hipStream_t stream) {
    EqhReduceDesc d{slab, out, nullptr, nullptr, rows * row_len, rows * row_len, 0, out_ld, row_len, n_slabs, 0};
    if (accumulate && eqh_defer_try(stream, d)) return EQH_OK;
    hipLaunchKernelGGL(eqh_k_reduce_slabs, dim3(eqh_grid_for(d.elems, 64, 2048)), dim3(1024), 0, stream, d, accumulate);
    return hipGetLastError() == hipSuccess ? EQH_OK : EQH_ERR_LAUNCH;
}
static inline int eqh_reduce_slabs_async(const float* slab, int n_slabs, int64_t elems, float* out,
                                         hipStream_t stream, int accumulate = 0) {
    return eqh_reduce_slabs3_async(slab, n_slabs, elems, out, nullptr, nullptr, elems, 0, accumulate, stream);
}
