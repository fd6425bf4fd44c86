// Elementwise passes of FAFormer's SwiGLU MLP on the per-edge, per-frame tensors (fa_former_layer.py:241-289 and
// the 8-frame average of :61-120).  At the Molecule3D batch these tensors are [E * 8, 256] = 2 GB each, every
// torch op on them is a 0.5-2 ms memory pass, and the reference's dropouts (p = 0.1, active in training) add a
// mask pass forward and backward.  Here
//   swiglu_drop:   out = dropout(SiLU(a) * b),  pre = [a | b] halves of one row          (one pass each way)
//   drop_mean:     out[r] = mean over F consecutive rows of dropout(x)                   (one pass each way)
// The dropout decision of element i is a hash of (seed, i) -- recomputed in the backward pass, so no mask is ever
// stored -- with the seed in device memory (a fresh draw per launch, also under hipGraph replay).
#include "common.h"

namespace {

__device__ __forceinline__ uint32_t mix64(uint64_t x) {   // splitmix64 finaliser, top 32 bits
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return (uint32_t)(x >> 32);
}
// keep-scale of element i: 0 (dropped) or 1 / (1 - p); threshold = p * 2^32
__device__ __forceinline__ float keep_scale(uint64_t seed, uint64_t i, uint32_t threshold, float inv_keep) {
    return mix64(seed + i * 0x9E3779B97F4A7C15ull) >= threshold ? inv_keep : 0.f;
}
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

__global__ void __launch_bounds__(256)
k_swiglu_drop_fwd(const float* __restrict__ pre, int64_t R, int H, const int64_t* __restrict__ seed_ptr,
                  uint32_t threshold, float inv_keep, float* __restrict__ out) {
    const uint64_t seed = threshold ? (uint64_t)*seed_ptr : 0;
    const int h4 = H >> 2;
    const int64_t total = R * h4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / h4;
        const int c = (int)(i - r * h4) * 4;
        const float4 a = *reinterpret_cast<const float4*>(pre + r * 2 * H + c);
        const float4 b = *reinterpret_cast<const float4*>(pre + r * 2 * H + H + c);
        float4 o = make_float4(a.x * sigmoid_fast(a.x) * b.x, a.y * sigmoid_fast(a.y) * b.y,
                               a.z * sigmoid_fast(a.z) * b.z, a.w * sigmoid_fast(a.w) * b.w);
        if (threshold) {
            const uint64_t e = (uint64_t)(r * H + c);
            o.x *= keep_scale(seed, e, threshold, inv_keep); o.y *= keep_scale(seed, e + 1, threshold, inv_keep);
            o.z *= keep_scale(seed, e + 2, threshold, inv_keep); o.w *= keep_scale(seed, e + 3, threshold, inv_keep);
        }
        *reinterpret_cast<float4*>(out + r * H + c) = o;
    }
}

__global__ void __launch_bounds__(256)
k_swiglu_drop_bwd(const float* __restrict__ pre, const float* __restrict__ dout, int64_t R, int H,
                  const int64_t* __restrict__ seed_ptr, uint32_t threshold, float inv_keep, float* __restrict__ dpre) {
    const uint64_t seed = threshold ? (uint64_t)*seed_ptr : 0;
    const int h4 = H >> 2;
    const int64_t total = R * h4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / h4;
        const int c = (int)(i - r * h4) * 4;
        const float4 a = *reinterpret_cast<const float4*>(pre + r * 2 * H + c);
        const float4 b = *reinterpret_cast<const float4*>(pre + r * 2 * H + H + c);
        float4 g = *reinterpret_cast<const float4*>(dout + r * H + c);
        if (threshold) {
            const uint64_t e = (uint64_t)(r * H + c);
            g.x *= keep_scale(seed, e, threshold, inv_keep); g.y *= keep_scale(seed, e + 1, threshold, inv_keep);
            g.z *= keep_scale(seed, e + 2, threshold, inv_keep); g.w *= keep_scale(seed, e + 3, threshold, inv_keep);
        }
        const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w}, gv[4] = {g.x, g.y, g.z, g.w};
        float da[4], db[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float sig = sigmoid_fast(av[k]);
            const float s = av[k] * sig;
            da[k] = gv[k] * bv[k] * fmaf(s, 1.0f - sig, sig);
            db[k] = gv[k] * s;
        }
        *reinterpret_cast<float4*>(dpre + r * 2 * H + c) = make_float4(da[0], da[1], da[2], da[3]);
        *reinterpret_cast<float4*>(dpre + r * 2 * H + H + c) = make_float4(db[0], db[1], db[2], db[3]);
    }
}

// out[r, :] = (1 / F) * sum_f dropout(x[r * F + f, :])
__global__ void __launch_bounds__(256)
k_drop_mean_fwd(const float* __restrict__ x, int64_t R, int F, int C, const int64_t* __restrict__ seed_ptr,
                uint32_t threshold, float inv_keep, float* __restrict__ out) {
    const uint64_t seed = threshold ? (uint64_t)*seed_ptr : 0;
    const int c4 = C >> 2;
    const int64_t total = R * c4;
    const float inv_f = 1.0f / (float)F;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / c4;
        const int c = (int)(i - r * c4) * 4;
        float4 acc = f4_zero();
        for (int f = 0; f < F; ++f) {
            const int64_t row = r * F + f;
            float4 v = *reinterpret_cast<const float4*>(x + row * C + c);
            if (threshold) {
                const uint64_t e = (uint64_t)(row * C + c);
                v.x *= keep_scale(seed, e, threshold, inv_keep); v.y *= keep_scale(seed, e + 1, threshold, inv_keep);
                v.z *= keep_scale(seed, e + 2, threshold, inv_keep); v.w *= keep_scale(seed, e + 3, threshold, inv_keep);
            }
            f4_add(acc, v);
        }
        acc.x *= inv_f; acc.y *= inv_f; acc.z *= inv_f; acc.w *= inv_f;
        *reinterpret_cast<float4*>(out + r * C + c) = acc;
    }
}

__global__ void __launch_bounds__(256)
k_drop_mean_bwd(const float* __restrict__ dout, int64_t R, int F, int C, const int64_t* __restrict__ seed_ptr,
                uint32_t threshold, float inv_keep, float* __restrict__ dx) {
    const uint64_t seed = threshold ? (uint64_t)*seed_ptr : 0;
    const int c4 = C >> 2;
    const int64_t total = R * F * c4;
    const float inv_f = 1.0f / (float)F;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / c4;
        const int c = (int)(i - row * c4) * 4;
        float4 g = *reinterpret_cast<const float4*>(dout + (row / F) * C + c);
        g.x *= inv_f; g.y *= inv_f; g.z *= inv_f; g.w *= inv_f;
        if (threshold) {
            const uint64_t e = (uint64_t)(row * C + c);
            g.x *= keep_scale(seed, e, threshold, inv_keep); g.y *= keep_scale(seed, e + 1, threshold, inv_keep);
            g.z *= keep_scale(seed, e + 2, threshold, inv_keep); g.w *= keep_scale(seed, e + 3, threshold, inv_keep);
        }
        *reinterpret_cast<float4*>(dx + row * C + c) = g;
    }
}

int ew_check(int64_t R, int32_t C, float p) {
    if (R < 0 || C <= 0 || !(p >= 0.f) || !(p < 1.f)) return EQH_ERR_ARG;
    if (C & 3) return EQH_ERR_ALIGN;
    return EQH_OK;
}
inline uint32_t ew_threshold(float p) { return p > 0.f ? (uint32_t)((double)p * 4294967296.0) : 0u; }

}  // namespace

extern "C" int faf_swiglu_dropout_fwd(const float* pre, int64_t R, int32_t H, float p, const int64_t* seed, float* out,
                                      void* stream_) {
    int rc = ew_check(R, H, p);
    if (rc) return rc;
    if (R == 0) return EQH_OK;
    if (!pre || !out || (p > 0.f && !seed)) return EQH_ERR_ARG;
    if (!eqh_aligned16(pre) || !eqh_aligned16(out)) return EQH_ERR_ALIGN;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(k_swiglu_drop_fwd, dim3(eqh_grid_for(R * (H / 4), 256, 8192)), dim3(256), 0, stream, pre, R, (int)H,
                       seed, ew_threshold(p), 1.0f / (1.0f - p), out);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}

extern "C" int faf_swiglu_dropout_bwd(const float* pre, const float* dout, int64_t R, int32_t H, float p,
                                      const int64_t* seed, float* dpre, void* stream_) {
    int rc = ew_check(R, H, p);
    if (rc) return rc;
    if (R == 0) return EQH_OK;
    if (!pre || !dout || !dpre || (p > 0.f && !seed)) return EQH_ERR_ARG;
    if (!eqh_aligned16(pre) || !eqh_aligned16(dout) || !eqh_aligned16(dpre)) return EQH_ERR_ALIGN;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(k_swiglu_drop_bwd, dim3(eqh_grid_for(R * (H / 4), 256, 8192)), dim3(256), 0, stream, pre, dout, R,
                       (int)H, seed, ew_threshold(p), 1.0f / (1.0f - p), dpre);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}

extern "C" int faf_dropout_mean_fwd(const float* x, int64_t R, int32_t F, int32_t C, float p, const int64_t* seed,
                                    float* out, void* stream_) {
    int rc = ew_check(R, C, p);
    if (rc || F <= 0) return rc ? rc : EQH_ERR_ARG;
    if (R == 0) return EQH_OK;
    if (!x || !out || (p > 0.f && !seed)) return EQH_ERR_ARG;
    if (!eqh_aligned16(x) || !eqh_aligned16(out)) return EQH_ERR_ALIGN;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(k_drop_mean_fwd, dim3(eqh_grid_for(R * (C / 4), 256, 8192)), dim3(256), 0, stream, x, R, (int)F,
                       (int)C, seed, ew_threshold(p), 1.0f / (1.0f - p), out);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}

extern "C" int faf_dropout_mean_bwd(const float* dout, int64_t R, int32_t F, int32_t C, float p, const int64_t* seed,
                                    float* dx, void* stream_) {
    int rc = ew_check(R, C, p);
    if (rc || F <= 0) return rc ? rc : EQH_ERR_ARG;
    if (R == 0) return EQH_OK;
    if (!dout || !dx || (p > 0.f && !seed)) return EQH_ERR_ARG;
    if (!eqh_aligned16(dout) || !eqh_aligned16(dx)) return EQH_ERR_ALIGN;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(k_drop_mean_bwd, dim3(eqh_grid_for(R * F * (C / 4), 256, 8192)), dim3(256), 0, stream, dout, R, (int)F,
                       (int)C, seed, ew_threshold(p), 1.0f / (1.0f - p), dx);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}
