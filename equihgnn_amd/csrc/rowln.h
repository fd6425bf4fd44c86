// Row-wise LayerNorm arithmetic shared by the aggregation kernels (incidence.hip) and the panel GEMM kernels
// (panel.hip): one 64-lane wavefront per row, a row of C <= 1024 channels is NV float4 per lane (lane l holds channels
// 4 (l + 64 i) .. + 3).  ONE definition, so that a LayerNorm fused into a GEMM prologue / epilogue gives bit-identical
// results to the stand-alone row kernels (mlp.py:91-99: Linear -> ReLU -> LayerNorm).
#pragma once
#include "common.h"

namespace {

// Wavefront all-reduce on the DPP cross-lane network (no LDS round trips): quad butterflies, then
// rotations inside each 16-lane row, then the four row totals are combined through scalar reads.
// Every lane of the wavefront must be active.  Fixed summation order => reproducible.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_move<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_move<0x124>(v);  // row_ror:4
    v += dpp_move<0x128>(v);  // row_ror:8
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    const int bits = __float_as_int(row16_sum(v));  // readlane moves 32-bit integers
    return (__int_as_float(__builtin_amdgcn_readlane(bits, 0)) + __int_as_float(__builtin_amdgcn_readlane(bits, 16))) +
           (__int_as_float(__builtin_amdgcn_readlane(bits, 32)) + __int_as_float(__builtin_amdgcn_readlane(bits, 48)));
}
__device__ __forceinline__ void wave_sum2(float& a, float& b) {
    a = wave_sum(a);
    b = wave_sum(b);
}

template <int NV>
struct Row {
    float4 v[NV];
};

// h = relu(u + w) (RELU) or u + w; returns xhat in `x`, rstd in *rstd, relu mask in `pos` (bit per comp)
template <int NV, bool RELU = true>
__device__ __forceinline__ void norm_pair(const Row<NV>& u, const Row<NV>& w, int C, int lane, float inv_c,
                                          float eps, Row<NV>& x, unsigned& pos, float* rstd) {
    float s = 0.f;
    pos = 0u;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float4 h = make_float4(u.v[i].x + w.v[i].x, u.v[i].y + w.v[i].y, u.v[i].z + w.v[i].z, u.v[i].w + w.v[i].w);
        if (RELU) {
            pos |= ((h.x > 0.f) ? 1u : 0u) << (4 * i) | ((h.y > 0.f) ? 2u : 0u) << (4 * i) |
                   ((h.z > 0.f) ? 4u : 0u) << (4 * i) | ((h.w > 0.f) ? 8u : 0u) << (4 * i);
            h.x = fmaxf(h.x, 0.f); h.y = fmaxf(h.y, 0.f); h.z = fmaxf(h.z, 0.f); h.w = fmaxf(h.w, 0.f);
        } else {
            pos |= 15u << (4 * i);
        }
        x.v[i] = h;
        s += (h.x + h.y) + (h.z + h.w);
    }
    const float mu = wave_sum(s) * inv_c;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        float4 d = x.v[i];
        if (c < C) { d.x -= mu; d.y -= mu; d.z -= mu; d.w -= mu; }
        x.v[i] = d;
        ss += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
    }
    const float r = 1.0f / sqrtf(wave_sum(ss) * inv_c + eps);
    *rstd = r;
#pragma unroll
    for (int i = 0; i < NV; ++i) { x.v[i].x *= r; x.v[i].y *= r; x.v[i].z *= r; x.v[i].w *= r; }
}


}  // namespace
