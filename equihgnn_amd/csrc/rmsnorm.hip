// Degree-0 branch of the Equiformer's Norm (equiformer_layer.py:194-225): per row
//   out = t / max(rms, eps) * g,   rms = ||t||_2 * C^-1/2,   g = transforms.0 [C, 1]
// As torch ops this is five launches forward and about eight backward on [N, C] rows, three times per step; here
// one launch each way, a wavefront per row (float4 per lane, C <= 1024), the scale gradient through per-workgroup
// slabs and the fixed-order reducer.
#include "common.h"

namespace {

constexpr int RN_THREADS = 256;
constexpr int RN_WAVES = RN_THREADS / 64;

template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    v += dpp_move<0x124>(v);
    v += dpp_move<0x128>(v);
    const int bits = __float_as_int(v);
    return (__int_as_float(__builtin_amdgcn_readlane(bits, 0)) + __int_as_float(__builtin_amdgcn_readlane(bits, 16))) +
           (__int_as_float(__builtin_amdgcn_readlane(bits, 32)) + __int_as_float(__builtin_amdgcn_readlane(bits, 48)));
}

template <int NV>
struct Row {
    float4 v[NV];
};

template <int NV>
__device__ __forceinline__ void load_row(const float* __restrict__ p, int64_t r, int C, int lane, Row<NV>& x) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        x.v[i] = c < C ? *reinterpret_cast<const float4*>(p + r * C + c) : f4_zero();
    }
}

template <int NV>
__device__ __forceinline__ float sum_sq(const Row<NV>& x) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (x.v[i].x * x.v[i].x + x.v[i].y * x.v[i].y) + (x.v[i].z * x.v[i].z + x.v[i].w * x.v[i].w);
    return wave_sum(s);
}

template <int NV>
__global__ void __launch_bounds__(RN_THREADS)
k_rms_fwd(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ out, int n_rows, int C,
          float scale, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    Row<NV> gam;
    load_row<NV>(g, 0, C, lane, gam);
    for (int r = blockIdx.x * RN_WAVES + wave; r < n_rows; r += gridDim.x * RN_WAVES) {
        Row<NV> t;
        load_row<NV>(x, r, C, lane, t);
        const float d = fmaxf(sqrtf(sum_sq<NV>(t)) * scale, eps);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (c < C) {
                float4 o;
                o.x = t.v[i].x / d * gam.v[i].x; o.y = t.v[i].y / d * gam.v[i].y;
                o.z = t.v[i].z / d * gam.v[i].z; o.w = t.v[i].w / d * gam.v[i].w;
                *reinterpret_cast<float4*>(out + (int64_t)r * C + c) = o;
            }
        }
    }
}

template <int NV>
__global__ void __launch_bounds__(RN_THREADS)
k_rms_bwd(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ dy,
          float* __restrict__ dx, float* __restrict__ slab, int n_rows, int C, float scale, float eps) {
    __shared__ float4 s_red[RN_THREADS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    Row<NV> gam, a_dg;
    load_row<NV>(g, 0, C, lane, gam);
#pragma unroll
    for (int i = 0; i < NV; ++i) a_dg.v[i] = f4_zero();
    for (int r = blockIdx.x * RN_WAVES + wave; r < n_rows; r += gridDim.x * RN_WAVES) {
        Row<NV> t, d;
        load_row<NV>(x, r, C, lane, t);
        load_row<NV>(dy, r, C, lane, d);
        const float rms = sqrtf(sum_sq<NV>(t)) * scale;
        const bool open = rms >= eps;                    // clamp(min=eps) passes the gradient where rms >= eps
        const float den = open ? rms : eps;
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            // dg += dy * t / den;  u = dy * g
            a_dg.v[i].x = fmaf(d.v[i].x, t.v[i].x / den, a_dg.v[i].x); a_dg.v[i].y = fmaf(d.v[i].y, t.v[i].y / den, a_dg.v[i].y);
            a_dg.v[i].z = fmaf(d.v[i].z, t.v[i].z / den, a_dg.v[i].z); a_dg.v[i].w = fmaf(d.v[i].w, t.v[i].w / den, a_dg.v[i].w);
            d.v[i].x *= gam.v[i].x; d.v[i].y *= gam.v[i].y; d.v[i].z *= gam.v[i].z; d.v[i].w *= gam.v[i].w;
            dot += (d.v[i].x * t.v[i].x + d.v[i].y * t.v[i].y) + (d.v[i].z * t.v[i].z + d.v[i].w * t.v[i].w);
        }
        dot = wave_sum(dot);
        // dt = u / den - t * (u . t) * scale^2 / den^3   (second term only while the clamp is open)
        const float k = open ? dot * scale * scale / (den * den * den) : 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (c < C) {
                float4 o;
                o.x = d.v[i].x / den - t.v[i].x * k; o.y = d.v[i].y / den - t.v[i].y * k;
                o.z = d.v[i].z / den - t.v[i].z * k; o.w = d.v[i].w / den - t.v[i].w * k;
                *reinterpret_cast<float4*>(dx + (int64_t)r * C + c) = o;
            }
        }
    }
    float* __restrict__ sl = slab + (int64_t)blockIdx.x * C;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        s_red[threadIdx.x] = a_dg.v[i];
        __syncthreads();
        if (wave == 0) {
            float4 t = s_red[lane];
            for (int w = 1; w < RN_WAVES; ++w) f4_add(t, s_red[w * 64 + lane]);
            const int c = (lane + 64 * i) * 4;
            if (c < C) *reinterpret_cast<float4*>(sl + c) = t;
        }
        __syncthreads();
    }
}

inline int rn_blocks(int64_t rows) { return eqh_grid_for(rows, RN_WAVES * 4, 256); }

template <typename F>
int rn_dispatch(int C, F&& f) {
    if (C <= 256) return f(std::integral_constant<int, 1>{});
    if (C <= 512) return f(std::integral_constant<int, 2>{});
    return f(std::integral_constant<int, 4>{});
}

int rn_check(int64_t n_rows, int32_t C) {
    if (n_rows < 0 || C <= 0 || n_rows > INT32_MAX) return EQH_ERR_ARG;
    if ((C & 3) || C > 1024) return EQH_ERR_RANGE;
    return EQH_OK;
}

}  // namespace

extern "C" int eqf_rms_norm_fwd(const float* x, const float* g, int64_t n_rows, int32_t C, float scale, float eps,
                                float* out, void* stream_) {
    int rc = rn_check(n_rows, C);
    if (rc) return rc;
    if (n_rows == 0) return EQH_OK;
    if (!x || !g || !out) return EQH_ERR_ARG;
    if (!eqh_aligned16(x) || !eqh_aligned16(g) || !eqh_aligned16(out)) return EQH_ERR_ALIGN;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    return rn_dispatch(C, [&](auto nv) {
        constexpr int NV = decltype(nv)::value;
        hipLaunchKernelGGL((k_rms_fwd<NV>), dim3(eqh_grid_for(n_rows, RN_WAVES, 4096)), dim3(RN_THREADS), 0, stream, x, g,
                           out, (int)n_rows, (int)C, scale, eps);
        EQH_CHECK_LAUNCH();
        return EQH_OK;
    });
}

extern "C" size_t eqf_rms_norm_bwd_workspace_bytes(int64_t n_rows, int32_t C) {
    if (n_rows < 0 || C <= 0) return 0;
    return (size_t)rn_blocks(n_rows) * (size_t)C * sizeof(float);
}

extern "C" int eqf_rms_norm_bwd(const float* x, const float* g, const float* dy, int64_t n_rows, int32_t C, float scale,
                                float eps, float* dx, float* dg, int32_t accumulate, void* workspace,
                                size_t workspace_bytes, void* stream_) {
    int rc = rn_check(n_rows, C);
    if (rc) return rc;
    if (!dg) return EQH_ERR_ARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n_rows == 0) return accumulate ? EQH_OK : eqh_zero_async(dg, C, stream);
    if (!x || !g || !dy || !dx || !workspace) return EQH_ERR_ARG;
    if (!eqh_aligned16(x) || !eqh_aligned16(g) || !eqh_aligned16(dy) || !eqh_aligned16(dx) || !eqh_aligned16(workspace))
        return EQH_ERR_ALIGN;
    if (workspace_bytes < eqf_rms_norm_bwd_workspace_bytes(n_rows, C)) return EQH_ERR_ARG;
    const int blocks = rn_blocks(n_rows);
    float* slab = static_cast<float*>(workspace);
    return rn_dispatch(C, [&](auto nv) {
        constexpr int NV = decltype(nv)::value;
        hipLaunchKernelGGL((k_rms_bwd<NV>), dim3(blocks), dim3(RN_THREADS), 0, stream, x, g, dy, dx, slab, (int)n_rows,
                           (int)C, scale, eps);
        EQH_CHECK_LAUNCH();
        return eqh_reduce_slabs_async(slab, blocks, C, dg, stream, accumulate);
    });
}
