// Elementwise passes of FAFormer's SwiGLU MLP on the per-edge, per-frame tensors (fa_former_layer.py:241-289 and
// the 8-frame average of :61-120).  At the Molecule3D batch these tensors are [E * 8, 256] = 2 GB each, every
// torch op on them is a 0.5-2 ms memory pass, and the reference's dropouts (p = 0.1, active in training) add a
// mask pass forward and backward.  Here
//   swiglu_drop:   out = dropout(SiLU(a) * b),  pre = [a | b] halves of one row          (one pass each way)
//   drop_mean:     out[r] = mean over F consecutive rows of dropout(x)                   (one pass each way)
// The dropout decision of element i is a hash of (seed, i) -- recomputed in the backward pass, so no mask is ever
// stored -- with the seed in device memory (a fresh draw per launch, also under hipGraph replay).
#include <cstdlib>

#include "common.h"
#include "drop_hash.h"

namespace {

// (dropout decisions: drop_hash.h -- shared with the fused fc2 epilogue of gemm_x6.hip)
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

__global__ void __launch_bounds__(256)
k_swiglu_drop_fwd(const float* __restrict__ pre, int64_t R, int H, const int64_t* __restrict__ seed_ptr,
                  uint32_t threshold, float inv_keep, float* __restrict__ out) {
    const DropKey seed = drop_key(threshold ? (uint64_t)*seed_ptr : 0);
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
            keep_scale4(seed, e, threshold, inv_keep, o);
        }
        *reinterpret_cast<float4*>(out + r * H + c) = o;
    }
}

__global__ void __launch_bounds__(256)
k_swiglu_drop_bwd(const float* __restrict__ pre, const float* __restrict__ dout, int64_t R, int H,
                  const int64_t* __restrict__ seed_ptr, uint32_t threshold, float inv_keep, float* __restrict__ dpre) {
    const DropKey seed = drop_key(threshold ? (uint64_t)*seed_ptr : 0);
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
            keep_scale4(seed, e, threshold, inv_keep, g);
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

// out[r, :] = (1 / F) * sum_f dropout(x[r * F + f, :]);  FT = F at compile time (0: any F) so that the F row loads of an
// output element are all in flight together (FAFormer's 8 sign frames)
template <int FT>
__global__ void __launch_bounds__(256)
k_drop_mean_fwd(const float* __restrict__ x, int64_t R, int F_rt, int C, const int64_t* __restrict__ seed_ptr,
                uint32_t threshold, float inv_keep, float* __restrict__ out) {
    const int F = FT ? FT : F_rt;
    const DropKey seed = drop_key(threshold ? (uint64_t)*seed_ptr : 0);
    const int c4 = C >> 2;
    const int64_t total = R * c4;
    const float inv_f = 1.0f / (float)F;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / c4;
        const int c = (int)(i - r * c4) * 4;
        float4 acc = f4_zero();
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const int64_t row = r * F + f;
            float4 v = *reinterpret_cast<const float4*>(x + row * C + c);
            if (threshold) {
                const uint64_t e = (uint64_t)(row * C + c);
                keep_scale4(seed, e, threshold, inv_keep, v);
            }
            f4_add(acc, v);
        }
        acc.x *= inv_f; acc.y *= inv_f; acc.z *= inv_f; acc.w *= inv_f;
        *reinterpret_cast<float4*>(out + r * C + c) = acc;
    }
}

// COLSUM: the column sums of dx ride along (the bias gradient of the Linear that produced the forward's input: one
// more pass over the 2 GB [E * 8, 256] tensor of FAFormer's frame MLP otherwise, 0.6 ms).  The grid stride and the block
// size are multiples of C / 4, so a thread always works on the same four columns: per-thread partial sums -> the
// block's slab row [C] (threads sharing columns summed in thread order) -> fixed-order slab reduction.
template <bool COLSUM>
__global__ void __launch_bounds__(256)
k_drop_mean_bwd(const float* __restrict__ dout, int64_t R, int F, int C, const int64_t* __restrict__ seed_ptr,
                uint32_t threshold, float inv_keep, float* __restrict__ dx, float* __restrict__ slab) {
    const DropKey seed = drop_key(threshold ? (uint64_t)*seed_ptr : 0);
    const int c4 = C >> 2;
    const int64_t total = R * F * c4;
    const float inv_f = 1.0f / (float)F;
    float4 cs = f4_zero();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / c4;
        const int c = (int)(i - row * c4) * 4;
        float4 g = *reinterpret_cast<const float4*>(dout + (row / F) * C + c);
        g.x *= inv_f; g.y *= inv_f; g.z *= inv_f; g.w *= inv_f;
        if (threshold) {
            const uint64_t e = (uint64_t)(row * C + c);
            keep_scale4(seed, e, threshold, inv_keep, g);
        }
        *reinterpret_cast<float4*>(dx + row * C + c) = g;
        if (COLSUM) f4_add(cs, g);
    }
    if (COLSUM) {
        __shared__ float4 s_cs[256];
        s_cs[threadIdx.x] = cs;
        __syncthreads();
        if ((int)threadIdx.x < c4) {
            float4 t = s_cs[threadIdx.x];
            for (int k = threadIdx.x + c4; k < 256; k += c4) f4_add(t, s_cs[k]);
            *reinterpret_cast<float4*>(slab + (int64_t)blockIdx.x * C + threadIdx.x * 4) = t;
        }
    }
}

// The same for F = 8 frames of C = 256 channels (FAFormer's frame MLPs): a wavefront owns one row of dout and writes its
// eight frame rows -- one load per eight 1 KiB stores, no 64-bit divisions (the general kernel above spends two per float4
// and has one load per store in flight: 3.5 TB/s on the 2 GB gradient, against ~6 TB/s for plain stores).  Bitwise the
// same dx; the column sums are taken over a different partition of the rows.
template <bool COLSUM>
__global__ void __launch_bounds__(256)
k_drop_mean_bwd_f8c256(const float* __restrict__ dout, int64_t R, const int64_t* __restrict__ seed_ptr, uint32_t threshold,
                       float inv_keep, float* __restrict__ dx, float* __restrict__ slab) {
    const DropKey seed = drop_key(threshold ? (uint64_t)*seed_ptr : 0);
    const int lane = threadIdx.x & 63;
    const int64_t nw = (int64_t)gridDim.x * 4;
    float4 cs = f4_zero();
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < R; r += nw) {
        float4 g = *reinterpret_cast<const float4*>(dout + r * 256 + lane * 4);
        g.x *= 0.125f; g.y *= 0.125f; g.z *= 0.125f; g.w *= 0.125f;
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const int64_t row = r * 8 + f;
            float4 o = g;
            if (threshold) {
                const uint64_t e = (uint64_t)(row * 256 + lane * 4);
                keep_scale4(seed, e, threshold, inv_keep, o);
            }
            *reinterpret_cast<float4*>(dx + row * 256 + lane * 4) = o;
            if (COLSUM) f4_add(cs, o);
        }
    }
    if (COLSUM) {
        __shared__ float4 s_cs[256];
        s_cs[threadIdx.x] = cs;
        __syncthreads();
        if (threadIdx.x < 64) {
            float4 t = s_cs[threadIdx.x];
            for (int k = threadIdx.x + 64; k < 256; k += 64) f4_add(t, s_cs[k]);
            *reinterpret_cast<float4*>(slab + (int64_t)blockIdx.x * 256 + threadIdx.x * 4) = t;
        }
    }
}

// out = res + dropout_p(x) (res may be NULL) and its backward dx = dout * keep: the residual connections behind FAFormer's
// MLPs (fa_former_layer.py:289 with :508, :606) -- torch runs them as a dropout kernel (which also writes a mask tensor)
// and an add; elements are float4 groups, the keep decision is the hash of the element index
__global__ void __launch_bounds__(256)
k_dropout_add(const float* __restrict__ x, const float* __restrict__ res, int64_t n4, const int64_t* __restrict__ seed_ptr,
              uint32_t threshold, float inv_keep, float* __restrict__ out) {
    const DropKey seed = drop_key(threshold ? (uint64_t)*seed_ptr : 0);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 v = reinterpret_cast<const float4*>(x)[i];
        if (threshold) {
            const uint64_t e = (uint64_t)i * 4;
            keep_scale4(seed, e, threshold, inv_keep, v);
        }
        if (res) f4_add(v, reinterpret_cast<const float4*>(res)[i]);
        reinterpret_cast<float4*>(out)[i] = v;
    }
}

int ew_check(int64_t R, int32_t C, float p) {
    if (R < 0 || C <= 0 || !(p >= 0.f) || !(p < 1.f)) return EQH_ERR_ARG;
    if (C & 3) return EQH_ERR_ALIGN;
    return EQH_OK;
}
inline uint32_t ew_threshold(float p) { return drop_threshold(p); }

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
                       seed, ew_threshold(p), drop_inv_keep(p), out);
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
                       (int)H, seed, ew_threshold(p), drop_inv_keep(p), dpre);
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
    if (F == 8)
        hipLaunchKernelGGL(k_drop_mean_fwd<8>, dim3(eqh_grid_for(R * (C / 4), 256, 8192)), dim3(256), 0, stream, x, R, (int)F,
                           (int)C, seed, ew_threshold(p), drop_inv_keep(p), out);
    else
        hipLaunchKernelGGL(k_drop_mean_fwd<0>, dim3(eqh_grid_for(R * (C / 4), 256, 8192)), dim3(256), 0, stream, x, R, (int)F,
                           (int)C, seed, ew_threshold(p), drop_inv_keep(p), out);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}

extern "C" int faf_dropout_add(const float* x, const float* res, int64_t n, float p, const int64_t* seed, float* out,
                               void* stream_) {
    if (n < 0 || (n & 3) || !(p >= 0.f) || !(p < 1.f)) return EQH_ERR_ARG;
    if (n == 0) return EQH_OK;
    if (!x || !out || (p > 0.f && !seed)) return EQH_ERR_ARG;
    if (!eqh_aligned16(x) || !eqh_aligned16(res) || !eqh_aligned16(out)) return EQH_ERR_ALIGN;
    hipLaunchKernelGGL(k_dropout_add, dim3(eqh_grid_for(n / 4, 256, 8192)), dim3(256), 0, static_cast<hipStream_t>(stream_), x,
                       res, n / 4, seed, ew_threshold(p), drop_inv_keep(p), out);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}

// (EQH_NO_DM_F8=1: the general kernel also for F = 8, C = 256 -- same-box timing of the two)
static inline bool dm_f8(int32_t F, int32_t C) {
    static const bool off = getenv("EQH_NO_DM_F8") != nullptr;
    return !off && F == 8 && C == 256;
}

extern "C" int faf_dropout_mean_bwd(const float* dout, int64_t R, int32_t F, int32_t C, float p, const int64_t* seed,
                                    float* dx, void* stream_) {
    int rc = ew_check(R, C, p);
    if (rc || F <= 0) return rc ? rc : EQH_ERR_ARG;
    if (R == 0) return EQH_OK;
    if (!dout || !dx || (p > 0.f && !seed)) return EQH_ERR_ARG;
    if (!eqh_aligned16(dout) || !eqh_aligned16(dx)) return EQH_ERR_ALIGN;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (dm_f8(F, C))
        hipLaunchKernelGGL(k_drop_mean_bwd_f8c256<false>, dim3(eqh_grid_for(R, 4, 8192)), dim3(256), 0, stream, dout, R, seed,
                           ew_threshold(p), drop_inv_keep(p), dx, nullptr);
    else
        hipLaunchKernelGGL(k_drop_mean_bwd<false>, dim3(eqh_grid_for(R * F * (C / 4), 256, 8192)), dim3(256), 0, stream, dout, R,
                           (int)F, (int)C, seed, ew_threshold(p), drop_inv_keep(p), dx, nullptr);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}

// the column-sum rider needs a thread to stay on its columns: C / 4 divides the block size (C = 4 .. 1024, a power of two)
static inline bool dm_colsum_ok(int32_t C) { return C >= 4 && C <= 1024 && (C & (C - 1)) == 0; }
static inline int dm_colsum_blocks(int64_t R, int32_t F, int32_t C) { return eqh_grid_for(R * F * (C / 4), 256, 4096); }

extern "C" size_t faf_dropout_mean_bwd_colsum_workspace_bytes(int64_t R, int32_t F, int32_t C) {
    if (R <= 0 || F <= 0 || !dm_colsum_ok(C)) return 0;
    return (size_t)dm_colsum_blocks(R, F, C) * (size_t)C * sizeof(float);
}

extern "C" int faf_dropout_mean_bwd_colsum(const float* dout, int64_t R, int32_t F, int32_t C, float p, const int64_t* seed,
                                           float* dx, float* colsum, int32_t accumulate, void* workspace,
                                           size_t workspace_bytes, void* stream_) {
    int rc = ew_check(R, C, p);
    if (rc || F <= 0) return rc ? rc : EQH_ERR_ARG;
    if (!dm_colsum_ok(C)) return EQH_ERR_ARG;
    if (!colsum) return EQH_ERR_ARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (R == 0) return accumulate ? EQH_OK : eqh_zero_async(colsum, C, stream);
    if (!dout || !dx || !workspace || (p > 0.f && !seed)) return EQH_ERR_ARG;
    if (!eqh_aligned16(dout) || !eqh_aligned16(dx) || !eqh_aligned16(workspace)) return EQH_ERR_ALIGN;
    if (workspace_bytes < faf_dropout_mean_bwd_colsum_workspace_bytes(R, F, C)) return EQH_ERR_ARG;
    const int blocks = dm_colsum_blocks(R, F, C);
    float* slab = static_cast<float*>(workspace);
    if (dm_f8(F, C))
        hipLaunchKernelGGL(k_drop_mean_bwd_f8c256<true>, dim3(blocks), dim3(256), 0, stream, dout, R, seed, ew_threshold(p),
                           drop_inv_keep(p), dx, slab);
    else
        hipLaunchKernelGGL(k_drop_mean_bwd<true>, dim3(blocks), dim3(256), 0, stream, dout, R, (int)F, (int)C, seed,
                           ew_threshold(p), drop_inv_keep(p), dx, slab);
    EQH_CHECK_LAUNCH();
    return eqh_reduce_slabs_async(slab, blocks, C, colsum, stream, accumulate ? 1 : 0);
}

// ---------------------------------------------------------------------------------------------------------------
// First Linear of the frame-averaged MLP (fa_former_layer.py:61-120 with :241-289): for every row e (an edge or a
// node) and each of the 8 sign frames f,   pre[e, f, :] = W3 (y_e * s_f) + base_e,   s_f = (+-1, +-1, +-1),
// W3 = fc1.weight[:, :3], base_e = fc1.weight[:, 3:] extra_e + bias (computed once per row, not per frame).
// As torch ops: build u = y * s [E, 8, 3], a K = 3 GEMM, a broadcast add -- and backward a [3 x 2M] . [2M x 256]
// GEMM, a [2M x 256] . [256 x 3] GEMM and a sum over frames, each a pass over the 2 GB tensor.  Here the forward
// forms t_d = y_d * W3[:, d] once per row and writes the 8 sign combinations; the backward reads d pre ONCE (a
// wavefront per row) and yields d base = sum_f, r_d = sum_f s_fd d pre_f, dy_d = r_d . W3[:, d] and the row's
// contribution y_d * r_d to d W3 (per-lane accumulators -> workgroup slabs -> fixed-order reducer).
namespace {

constexpr int FP_THREADS = 256;
constexpr int FP_WAVES = FP_THREADS / 64;

template <int CTRL>
__device__ __forceinline__ float fp_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float fp_wave_sum(float v) {
    v += fp_dpp<0xB1>(v);
    v += fp_dpp<0x4E>(v);
    v += fp_dpp<0x124>(v);
    v += fp_dpp<0x128>(v);
    const int b = __float_as_int(v);
    return (__int_as_float(__builtin_amdgcn_readlane(b, 0)) + __int_as_float(__builtin_amdgcn_readlane(b, 16))) +
           (__int_as_float(__builtin_amdgcn_readlane(b, 32)) + __int_as_float(__builtin_amdgcn_readlane(b, 48)));
}

__device__ __forceinline__ float4 f4_combo(const float4& t0, const float4& t1, const float4& t2, const float4& b, int f) {
    const float s0 = (f & 4) ? 1.f : -1.f, s1 = (f & 2) ? 1.f : -1.f, s2 = (f & 1) ? 1.f : -1.f;
    return make_float4(fmaf(s0, t0.x, fmaf(s1, t1.x, fmaf(s2, t2.x, b.x))), fmaf(s0, t0.y, fmaf(s1, t1.y, fmaf(s2, t2.y, b.y))),
                       fmaf(s0, t0.z, fmaf(s1, t1.z, fmaf(s2, t2.z, b.z))), fmaf(s0, t0.w, fmaf(s1, t1.w, fmaf(s2, t2.w, b.w))));
}

// H = 256: one float4 of h per lane, a wavefront per row
__global__ void __launch_bounds__(FP_THREADS)
k_frame_pre_fwd(const float* __restrict__ y, const float* __restrict__ w3, const float* __restrict__ base, int64_t E,
                float* __restrict__ out) {
    constexpr int H = 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane * 4;
    float4 w0, w1, w2;   // W3[h .. h+3][d]
    w0 = make_float4(w3[(h + 0) * 3 + 0], w3[(h + 1) * 3 + 0], w3[(h + 2) * 3 + 0], w3[(h + 3) * 3 + 0]);
    w1 = make_float4(w3[(h + 0) * 3 + 1], w3[(h + 1) * 3 + 1], w3[(h + 2) * 3 + 1], w3[(h + 3) * 3 + 1]);
    w2 = make_float4(w3[(h + 0) * 3 + 2], w3[(h + 1) * 3 + 2], w3[(h + 2) * 3 + 2], w3[(h + 3) * 3 + 2]);
    for (int64_t e = (int64_t)blockIdx.x * FP_WAVES + wave; e < E; e += (int64_t)gridDim.x * FP_WAVES) {
        const float y0 = y[e * 3], y1 = y[e * 3 + 1], y2 = y[e * 3 + 2];
        const float4 b = *reinterpret_cast<const float4*>(base + e * H + h);
        const float4 t0 = make_float4(y0 * w0.x, y0 * w0.y, y0 * w0.z, y0 * w0.w);
        const float4 t1 = make_float4(y1 * w1.x, y1 * w1.y, y1 * w1.z, y1 * w1.w);
        const float4 t2 = make_float4(y2 * w2.x, y2 * w2.y, y2 * w2.z, y2 * w2.w);
#pragma unroll
        for (int f = 0; f < 8; ++f) *reinterpret_cast<float4*>(out + (e * 8 + f) * H + h) = f4_combo(t0, t1, t2, b, f);
    }
}

__global__ void __launch_bounds__(FP_THREADS)
k_frame_pre_bwd(const float* __restrict__ y, const float* __restrict__ w3, const float* __restrict__ dpre, int64_t E,
                float* __restrict__ dy, float* __restrict__ dbase, float* __restrict__ slab) {
    constexpr int H = 256;
    __shared__ float4 s_red[FP_WAVES][3][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane * 4;
    float4 w0, w1, w2;
    w0 = make_float4(w3[(h + 0) * 3 + 0], w3[(h + 1) * 3 + 0], w3[(h + 2) * 3 + 0], w3[(h + 3) * 3 + 0]);
    w1 = make_float4(w3[(h + 0) * 3 + 1], w3[(h + 1) * 3 + 1], w3[(h + 2) * 3 + 1], w3[(h + 3) * 3 + 1]);
    w2 = make_float4(w3[(h + 0) * 3 + 2], w3[(h + 1) * 3 + 2], w3[(h + 2) * 3 + 2], w3[(h + 3) * 3 + 2]);
    float4 a0 = f4_zero(), a1 = f4_zero(), a2 = f4_zero();   // d W3[h .. h+3][d]
    for (int64_t e = (int64_t)blockIdx.x * FP_WAVES + wave; e < E; e += (int64_t)gridDim.x * FP_WAVES) {
        float4 g[8];
#pragma unroll
        for (int f = 0; f < 8; ++f) g[f] = *reinterpret_cast<const float4*>(dpre + (e * 8 + f) * H + h);
        float4 sb = f4_zero(), r0 = f4_zero(), r1 = f4_zero(), r2 = f4_zero();
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            f4_add(sb, g[f]);
            f4_fma(r0, g[f], (f & 4) ? 1.f : -1.f);
            f4_fma(r1, g[f], (f & 2) ? 1.f : -1.f);
            f4_fma(r2, g[f], (f & 1) ? 1.f : -1.f);
        }
        *reinterpret_cast<float4*>(dbase + e * H + h) = sb;
        const float d0 = fp_wave_sum((r0.x * w0.x + r0.y * w0.y) + (r0.z * w0.z + r0.w * w0.w));
        const float d1 = fp_wave_sum((r1.x * w1.x + r1.y * w1.y) + (r1.z * w1.z + r1.w * w1.w));
        const float d2 = fp_wave_sum((r2.x * w2.x + r2.y * w2.y) + (r2.z * w2.z + r2.w * w2.w));
        if (lane == 0) { dy[e * 3] = d0; dy[e * 3 + 1] = d1; dy[e * 3 + 2] = d2; }
        const float y0 = y[e * 3], y1 = y[e * 3 + 1], y2 = y[e * 3 + 2];
        f4_fma(a0, r0, y0);
        f4_fma(a1, r1, y1);
        f4_fma(a2, r2, y2);
    }
    s_red[wave][0][lane] = a0; s_red[wave][1][lane] = a1; s_red[wave][2][lane] = a2;
    __syncthreads();
    if (wave == 0) {
        float* __restrict__ sl = slab + (int64_t)blockIdx.x * (H * 3);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            float4 t = s_red[0][d][lane];
            for (int w = 1; w < FP_WAVES; ++w) f4_add(t, s_red[w][d][lane]);
            sl[(h + 0) * 3 + d] = t.x; sl[(h + 1) * 3 + d] = t.y; sl[(h + 2) * 3 + d] = t.z; sl[(h + 3) * 3 + d] = t.w;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The whole hidden layer of the frame-averaged MLP in one kernel each way:
//   hn[e, f, :] = LayerNorm( dropout( SiLU(a_f) * b_f ) ),   [a_f | b_f] = W3 (y_e * s_f) + base_e      (H = 256 -> 128)
// i.e. k_frame_pre + k_swiglu_drop + the row LayerNorm without the two intermediate tensors: at the Molecule3D batch the
// edge MLP's [E * 8, 256] pre-activations (2 GB) and [E * 8, 128] gated values (1 GB) were written and read back by
// three kernels forward (1.6 ms) and three backward (2.3 ms); here the forward reads y and base (one row per EDGE, not
// per frame) and writes hn, the backward reads d hn and writes d base / dy -- the eight frames of a row live in registers.
// A wavefront per row e; lane l owns hidden units j = 2l, 2l+1, i.e. pre-activation channels a: 2l, 2l+1 and b: 128 + 2l,
// 128 + 2l + 1, so the SwiGLU product needs no cross-lane traffic and the stores are 8-byte pieces of one 512-byte row.
// Dropout decisions: the same hash of (seed, element of the [E * 8, 128] tensor) as k_swiglu_drop.
__device__ __forceinline__ float fh_combo(float t0, float t1, float t2, float b, int f) {
    const float s0 = (f & 4) ? 1.f : -1.f, s1 = (f & 2) ? 1.f : -1.f, s2 = (f & 1) ? 1.f : -1.f;
    return fmaf(s0, t0, fmaf(s1, t1, fmaf(s2, t2, b)));
}

struct FhLane {          // per-lane constants: W3 rows of the four channels, gamma / beta of the two hidden units
    float wa0[3], wa1[3], wb0[3], wb1[3];
    float g0, g1, be0, be1;
};
__device__ __forceinline__ FhLane fh_load(const float* __restrict__ w3, int w_ld, const float* __restrict__ gamma,
                                          const float* __restrict__ beta, int lane) {
    FhLane L;
    const int ca = 2 * lane, cb = 128 + 2 * lane;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        L.wa0[d] = w3[ca * w_ld + d]; L.wa1[d] = w3[(ca + 1) * w_ld + d];
        L.wb0[d] = w3[cb * w_ld + d]; L.wb1[d] = w3[(cb + 1) * w_ld + d];
    }
    L.g0 = gamma[2 * lane]; L.g1 = gamma[2 * lane + 1];
    L.be0 = beta ? beta[2 * lane] : 0.f; L.be1 = beta ? beta[2 * lane + 1] : 0.f;
    return L;
}

// `base`: one row per point (base_ld = 256), or -- with wx -- the bias vector of fc1, the point's row being
// bias + extra[e] * wx (the K = 1 Linear of the squared distance, fc1.weight[:, 3]; extra NULL: the bias alone)
__global__ void __launch_bounds__(FP_THREADS)
k_frame_hidden_fwd(const float* __restrict__ y, const float* __restrict__ w3, const float* __restrict__ base, int64_t base_ld,
                   const float* __restrict__ extra, const float* __restrict__ wx,
                   const float* __restrict__ gamma, const float* __restrict__ beta, int64_t E,
                   const int64_t* __restrict__ seed_ptr, uint32_t threshold, float inv_keep, float eps,
                   float* __restrict__ out, int w_ld) {
    const DropKey seed = drop_key(threshold ? (uint64_t)*seed_ptr : 0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const FhLane L = fh_load(w3, w_ld, gamma, beta, lane);
    float2 wxa = make_float2(0.f, 0.f), wxb = make_float2(0.f, 0.f);
    if (wx) {      // (w_ld = 4: w3 and wx are columns 0-2 and 3 of one [256, 4] matrix, wx = w3 + 3)
        const int ws = w_ld == 4 ? 4 : 1;
        wxa = make_float2(wx[(2 * lane) * ws], wx[(2 * lane + 1) * ws]);
        wxb = make_float2(wx[(128 + 2 * lane) * ws], wx[(128 + 2 * lane + 1) * ws]);
    }
    for (int64_t e = (int64_t)blockIdx.x * FP_WAVES + wave; e < E; e += (int64_t)gridDim.x * FP_WAVES) {
        const float y0 = y[e * 3], y1 = y[e * 3 + 1], y2 = y[e * 3 + 2];
        float2 ba = *reinterpret_cast<const float2*>(base + e * base_ld + 2 * lane);
        float2 bb = *reinterpret_cast<const float2*>(base + e * base_ld + 128 + 2 * lane);
        if (wx && extra) {
            const float ex = extra[e];
            ba.x = fmaf(ex, wxa.x, ba.x); ba.y = fmaf(ex, wxa.y, ba.y);
            bb.x = fmaf(ex, wxb.x, bb.x); bb.y = fmaf(ex, wxb.y, bb.y);
        }
        const float ta0[3] = {y0 * L.wa0[0], y1 * L.wa0[1], y2 * L.wa0[2]}, ta1[3] = {y0 * L.wa1[0], y1 * L.wa1[1], y2 * L.wa1[2]};
        const float tb0[3] = {y0 * L.wb0[0], y1 * L.wb0[1], y2 * L.wb0[2]}, tb1[3] = {y0 * L.wb1[0], y1 * L.wb1[1], y2 * L.wb1[2]};
        float v0[8], v1[8], mu[8];
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const float a0 = fh_combo(ta0[0], ta0[1], ta0[2], ba.x, f), a1 = fh_combo(ta1[0], ta1[1], ta1[2], ba.y, f);
            const float b0 = fh_combo(tb0[0], tb0[1], tb0[2], bb.x, f), b1 = fh_combo(tb1[0], tb1[1], tb1[2], bb.y, f);
            float h0 = a0 * sigmoid_fast(a0) * b0, h1 = a1 * sigmoid_fast(a1) * b1;
            if (threshold) {
                const uint64_t idx = (uint64_t)((e * 8 + f) * 128 + 2 * lane);
                float kk0, kk1;
                keep_scale2(seed, idx, threshold, inv_keep, kk0, kk1);      // (idx is even: one hash for the pair)
                h0 *= kk0; h1 *= kk1;
            }
            v0[f] = h0; v1[f] = h1;
        }
#pragma unroll
        for (int f = 0; f < 8; ++f) mu[f] = fp_wave_sum(v0[f] + v1[f]) * (1.0f / 128.0f);
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const float d0 = v0[f] - mu[f], d1 = v1[f] - mu[f];
            const float r = 1.0f / sqrtf(fp_wave_sum(d0 * d0 + d1 * d1) * (1.0f / 128.0f) + eps);
            float2 o;
            o.x = fmaf(L.g0, d0 * r, L.be0);
            o.y = fmaf(L.g1, d1 * r, L.be1);
            *reinterpret_cast<float2*>(out + (e * 8 + f) * 128 + 2 * lane) = o;
        }
    }
}

// slab per workgroup: [d W3 (256 x 3) | d gamma (128) | d beta (128)]; in the vector-base form a second slab array behind
// the first holds [d bias (256) | d wx (256)] per workgroup
constexpr int FH_SLAB = 256 * 3 + 128 + 128;
constexpr int FH_SLAB_VEC = FH_SLAB + 256 + 256;

template <bool VEC>
__global__ void __launch_bounds__(FP_THREADS)
k_frame_hidden_bwd(const float* __restrict__ y, const float* __restrict__ w3, const float* __restrict__ base, int64_t base_ld,
                   const float* __restrict__ extra, const float* __restrict__ wx,
                   const float* __restrict__ gamma, const float* __restrict__ dhn, int64_t E,
                   const int64_t* __restrict__ seed_ptr, uint32_t threshold, float inv_keep, float eps,
                   float* __restrict__ dy, float* __restrict__ dbase, float* __restrict__ dextra, float* __restrict__ slab, int w_ld) {
    __shared__ float s_red[FP_WAVES][VEC ? 24 : 16][64];
    const DropKey seed = drop_key(threshold ? (uint64_t)*seed_ptr : 0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const FhLane L = fh_load(w3, w_ld, gamma, nullptr, lane);
    float aw[4][3];              // d W3 of channels (a0, a1, b0, b1) x d
    float ag0 = 0.f, ag1 = 0.f, ab0 = 0.f, ab1 = 0.f;   // d gamma, d beta of the two hidden units
    float adb[4] = {0.f, 0.f, 0.f, 0.f}, adx[4] = {0.f, 0.f, 0.f, 0.f};   // VEC: d bias, d wx of the four channels
    float wxv[4] = {0.f, 0.f, 0.f, 0.f};
    if (VEC) {
        const int ws = w_ld == 4 ? 4 : 1;
        wxv[0] = wx[(2 * lane) * ws]; wxv[1] = wx[(2 * lane + 1) * ws];
        wxv[2] = wx[(128 + 2 * lane) * ws]; wxv[3] = wx[(128 + 2 * lane + 1) * ws];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) aw[c][0] = aw[c][1] = aw[c][2] = 0.f;
    for (int64_t e = (int64_t)blockIdx.x * FP_WAVES + wave; e < E; e += (int64_t)gridDim.x * FP_WAVES) {
        const float y0 = y[e * 3], y1 = y[e * 3 + 1], y2 = y[e * 3 + 2];
        float2 ba = *reinterpret_cast<const float2*>(base + e * base_ld + 2 * lane);
        float2 bb = *reinterpret_cast<const float2*>(base + e * base_ld + 128 + 2 * lane);
        float ex = 0.f;
        if (VEC && extra) {
            ex = extra[e];
            ba.x = fmaf(ex, wxv[0], ba.x); ba.y = fmaf(ex, wxv[1], ba.y);
            bb.x = fmaf(ex, wxv[2], bb.x); bb.y = fmaf(ex, wxv[3], bb.y);
        }
        float2 g[8];
#pragma unroll
        for (int f = 0; f < 8; ++f) g[f] = *reinterpret_cast<const float2*>(dhn + (e * 8 + f) * 128 + 2 * lane);
        const float ta0[3] = {y0 * L.wa0[0], y1 * L.wa0[1], y2 * L.wa0[2]}, ta1[3] = {y0 * L.wa1[0], y1 * L.wa1[1], y2 * L.wa1[2]};
        const float tb0[3] = {y0 * L.wb0[0], y1 * L.wb0[1], y2 * L.wb0[2]}, tb1[3] = {y0 * L.wb1[0], y1 * L.wb1[1], y2 * L.wb1[2]};
        float sb[4] = {0.f, 0.f, 0.f, 0.f};           // sum over frames of d pre (a0, a1, b0, b1)
        float r[4][3];                                // sum over frames of s_fd * d pre
#pragma unroll
        for (int c = 0; c < 4; ++c) r[c][0] = r[c][1] = r[c][2] = 0.f;
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const float a0 = fh_combo(ta0[0], ta0[1], ta0[2], ba.x, f), a1 = fh_combo(ta1[0], ta1[1], ta1[2], ba.y, f);
            const float b0 = fh_combo(tb0[0], tb0[1], tb0[2], bb.x, f), b1 = fh_combo(tb1[0], tb1[1], tb1[2], bb.y, f);
            const float sg0 = sigmoid_fast(a0), sg1 = sigmoid_fast(a1);
            const float s0 = a0 * sg0, s1 = a1 * sg1;
            float k0 = 1.f, k1 = 1.f;
            if (threshold) {
                const uint64_t idx = (uint64_t)((e * 8 + f) * 128 + 2 * lane);
                keep_scale2(seed, idx, threshold, inv_keep, k0, k1);
            }
            const float h0 = s0 * b0 * k0, h1 = s1 * b1 * k1;
            const float mu = fp_wave_sum(h0 + h1) * (1.0f / 128.0f);
            const float d0 = h0 - mu, d1 = h1 - mu;
            const float rstd = 1.0f / sqrtf(fp_wave_sum(d0 * d0 + d1 * d1) * (1.0f / 128.0f) + eps);
            const float x0 = d0 * rstd, x1 = d1 * rstd;
            ab0 += g[f].x; ab1 += g[f].y;
            ag0 = fmaf(g[f].x, x0, ag0); ag1 = fmaf(g[f].y, x1, ag1);
            const float q0 = g[f].x * L.g0, q1 = g[f].y * L.g1;
            const float m1 = fp_wave_sum(q0 + q1) * (1.0f / 128.0f);
            const float m2 = fp_wave_sum(q0 * x0 + q1 * x1) * (1.0f / 128.0f);
            const float dh0 = rstd * (q0 - m1 - x0 * m2) * k0, dh1 = rstd * (q1 - m1 - x1 * m2) * k1;
            const float dp[4] = {dh0 * b0 * fmaf(s0, 1.0f - sg0, sg0), dh1 * b1 * fmaf(s1, 1.0f - sg1, sg1), dh0 * s0, dh1 * s1};
            const float sg[3] = {(f & 4) ? 1.f : -1.f, (f & 2) ? 1.f : -1.f, (f & 1) ? 1.f : -1.f};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                sb[c] += dp[c];
#pragma unroll
                for (int d = 0; d < 3; ++d) r[c][d] = fmaf(sg[d], dp[c], r[c][d]);
            }
        }
        if (!VEC) {
            *reinterpret_cast<float2*>(dbase + e * 256 + 2 * lane) = make_float2(sb[0], sb[1]);
            *reinterpret_cast<float2*>(dbase + e * 256 + 128 + 2 * lane) = make_float2(sb[2], sb[3]);
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                adb[c] += sb[c];
                adx[c] = fmaf(sb[c], ex, adx[c]);
            }
            if (extra) {
                const float de = fp_wave_sum((sb[0] * wxv[0] + sb[1] * wxv[1]) + (sb[2] * wxv[2] + sb[3] * wxv[3]));
                if (lane == 0) dextra[e] = de;
            }
        }
        const float dd0 = fp_wave_sum((r[0][0] * L.wa0[0] + r[1][0] * L.wa1[0]) + (r[2][0] * L.wb0[0] + r[3][0] * L.wb1[0]));
        const float dd1 = fp_wave_sum((r[0][1] * L.wa0[1] + r[1][1] * L.wa1[1]) + (r[2][1] * L.wb0[1] + r[3][1] * L.wb1[1]));
        const float dd2 = fp_wave_sum((r[0][2] * L.wa0[2] + r[1][2] * L.wa1[2]) + (r[2][2] * L.wb0[2] + r[3][2] * L.wb1[2]));
        if (lane == 0) { dy[e * 3] = dd0; dy[e * 3 + 1] = dd1; dy[e * 3 + 2] = dd2; }
        const float yv[3] = {y0, y1, y2};
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int d = 0; d < 3; ++d) aw[c][d] = fmaf(r[c][d], yv[d], aw[c][d]);
    }
    // workgroup slab: the four wavefronts' accumulators meet in LDS, summed in wavefront order
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int d = 0; d < 3; ++d) s_red[wave][c * 3 + d][lane] = aw[c][d];
    s_red[wave][12][lane] = ag0; s_red[wave][13][lane] = ag1; s_red[wave][14][lane] = ab0; s_red[wave][15][lane] = ab1;
    if (VEC) {
#pragma unroll
        for (int c = 0; c < 4; ++c) { s_red[wave][16 + c][lane] = adb[c]; s_red[wave][20 + c][lane] = adx[c]; }
    }
    __syncthreads();
    if (wave == 0) {
        constexpr int NQ = VEC ? 24 : 16;
        float* __restrict__ sl = slab + (int64_t)blockIdx.x * FH_SLAB;
        float* __restrict__ sl2 = slab + (int64_t)gridDim.x * FH_SLAB + (int64_t)blockIdx.x * 512;   // VEC: [d bias | d wx]
        float t[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            t[q] = s_red[0][q][lane];
            for (int w = 1; w < FP_WAVES; ++w) t[q] += s_red[w][q][lane];
        }
        const int ch[4] = {2 * lane, 2 * lane + 1, 128 + 2 * lane, 128 + 2 * lane + 1};
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int d = 0; d < 3; ++d) sl[ch[c] * 3 + d] = t[c * 3 + d];
        sl[768 + 2 * lane] = t[12]; sl[768 + 2 * lane + 1] = t[13];
        sl[896 + 2 * lane] = t[14]; sl[896 + 2 * lane + 1] = t[15];
        if (VEC) {
#pragma unroll
            for (int c = 0; c < 4; ++c) { sl2[ch[c]] = t[16 + c]; sl2[256 + ch[c]] = t[20 + c]; }
        }
    }
}

inline int fp_blocks(int64_t E) { return eqh_grid_for(E, FP_WAVES * 8, 1024); }

}  // namespace

extern "C" int faf_frame_pre_fwd(const float* y, const float* w3, const float* base, int64_t E, int32_t H, float* out,
                                 void* stream_) {
    if (E < 0 || H != 256) return EQH_ERR_ARG;
    if (E == 0) return EQH_OK;
    if (!y || !w3 || !base || !out) return EQH_ERR_ARG;
    if (!eqh_aligned16(base) || !eqh_aligned16(out)) return EQH_ERR_ALIGN;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(k_frame_pre_fwd, dim3(eqh_grid_for(E, FP_WAVES, 8192)), dim3(FP_THREADS), 0, stream, y, w3, base, E,
                       out);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}

extern "C" size_t faf_frame_pre_bwd_workspace_bytes(int64_t E, int32_t H) {
    if (E <= 0 || H != 256) return 0;
    return (size_t)fp_blocks(E) * (size_t)H * 3 * sizeof(float);
}

extern "C" int faf_frame_pre_bwd(const float* y, const float* w3, const float* dpre, int64_t E, int32_t H, float* dy,
                                 float* dbase, float* dw3, int32_t accumulate, void* workspace, size_t workspace_bytes,
                                 void* stream_) {
    if (E < 0 || H != 256 || !dw3) return EQH_ERR_ARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (E == 0) return accumulate ? EQH_OK : eqh_zero_async(dw3, (int64_t)H * 3, stream);
    if (!y || !w3 || !dpre || !dy || !dbase || !workspace) return EQH_ERR_ARG;
    if (!eqh_aligned16(dpre) || !eqh_aligned16(dbase) || !eqh_aligned16(workspace)) return EQH_ERR_ALIGN;
    if (workspace_bytes < faf_frame_pre_bwd_workspace_bytes(E, H)) return EQH_ERR_ARG;
    const int blocks = fp_blocks(E);
    float* slab = static_cast<float*>(workspace);
    hipLaunchKernelGGL(k_frame_pre_bwd, dim3(blocks), dim3(FP_THREADS), 0, stream, y, w3, dpre, E, dy, dbase, slab);
    EQH_CHECK_LAUNCH();
    return eqh_reduce_slabs_async(slab, blocks, (int64_t)H * 3, dw3, stream, accumulate);
}

/* frame_pre + SwiGLU + dropout + LayerNorm in one launch each way (see k_frame_hidden_fwd): hn [E, 8, 128] from y [E, 3],
   w3 [256, 3] (w_ld = 3) or, with w_ld = 4, columns 0-2 of fc1.weight [256, 4] read in place (wx = w3 + 3 is then its
   column 3), base ([E, 256] with base_ld = 256, or one row with base_ld = 0), gamma / beta [128].  bwd: dy [E, 3],
   dbase [E, 256] (per row, also for a broadcast base), dw3 [256, 3] (packed in either case), dgamma, dbeta [128]
   (overwritten or accumulated). */
extern "C" int faf_frame_hidden_fwd(const float* y, const float* w3, const float* base, int64_t base_ld, const float* extra,
                                    const float* wx, const float* gamma, const float* beta, int64_t E, float p,
                                    const int64_t* seed, float eps, float* out, int32_t w_ld, void* stream_) {
    if (E < 0 || !(p >= 0.f) || !(p < 1.f) || (base_ld != 0 && base_ld != 256) || (wx && base_ld != 0) || (extra && !wx))
        return EQH_ERR_ARG;
    if ((w_ld != 3 && w_ld != 4) || (w_ld == 4 && wx && wx != w3 + 3)) return EQH_ERR_ARG;
    if (E == 0) return EQH_OK;
    if (!y || !w3 || !base || !gamma || !beta || !out || (p > 0.f && !seed)) return EQH_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(base) & 7) || (reinterpret_cast<uintptr_t>(out) & 7)) return EQH_ERR_ALIGN;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(k_frame_hidden_fwd, dim3(eqh_grid_for(E, FP_WAVES, 8192)), dim3(FP_THREADS), 0, stream, y, w3, base,
                       base_ld, extra, wx, gamma, beta, E, seed, ew_threshold(p), drop_inv_keep(p), eps, out, (int)w_ld);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}

extern "C" size_t faf_frame_hidden_bwd_workspace_bytes(int64_t E) {
    if (E <= 0) return 0;
    return (size_t)fp_blocks(E) * (size_t)FH_SLAB_VEC * sizeof(float);
}

/* Row form (wx NULL): dbase [E, 256] is written (also for a broadcast base).  Vector form (wx given, base = the bias vector,
   base_ld 0): dbase receives d bias [256] and dwx d wx [256] instead, dextra [E] the gradient of extra (extra may be NULL). */
extern "C" int faf_frame_hidden_bwd(const float* y, const float* w3, const float* base, int64_t base_ld, const float* extra,
                                    const float* wx, const float* gamma, const float* dhn, int64_t E, float p,
                                    const int64_t* seed, float eps, float* dy, float* dbase, float* dwx, float* dextra,
                                    float* dw3, float* dgamma, float* dbeta, int32_t accumulate, int32_t w_ld, void* workspace,
                                    size_t workspace_bytes, void* stream_) {
    if (E < 0 || !(p >= 0.f) || !(p < 1.f) || (base_ld != 0 && base_ld != 256) || !dw3 || !dgamma || !dbeta) return EQH_ERR_ARG;
    if ((w_ld != 3 && w_ld != 4) || (w_ld == 4 && wx && wx != w3 + 3)) return EQH_ERR_ARG;
    if ((wx && (base_ld != 0 || !dwx || !dbase)) || (extra && (!wx || !dextra))) return EQH_ERR_ARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (E == 0) {
        if (accumulate) return EQH_OK;
        if (eqh_zero_async(dw3, 768, stream) || eqh_zero_async(dgamma, 128, stream)) return EQH_ERR_LAUNCH;
        if (wx && (eqh_zero_async(dbase, 256, stream) || eqh_zero_async(dwx, 256, stream))) return EQH_ERR_LAUNCH;
        return eqh_zero_async(dbeta, 128, stream);
    }
    if (!y || !w3 || !base || !gamma || !dhn || !dy || !dbase || !workspace || (p > 0.f && !seed)) return EQH_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(base) & 7) || (reinterpret_cast<uintptr_t>(dhn) & 7) ||
        (reinterpret_cast<uintptr_t>(dbase) & 7) || !eqh_aligned16(workspace))
        return EQH_ERR_ALIGN;
    if (workspace_bytes < faf_frame_hidden_bwd_workspace_bytes(E)) return EQH_ERR_ARG;
    const int blocks = fp_blocks(E);
    float* slab = static_cast<float*>(workspace);
    if (!wx) {
        hipLaunchKernelGGL(k_frame_hidden_bwd<false>, dim3(blocks), dim3(FP_THREADS), 0, stream, y, w3, base, base_ld, extra, wx,
                           gamma, dhn, E, seed, ew_threshold(p), drop_inv_keep(p), eps, dy, dbase, dextra, slab, (int)w_ld);
        EQH_CHECK_LAUNCH();
        return eqh_reduce_slabs3_async(slab, blocks, FH_SLAB, dw3, dgamma, dbeta, 768, 128, accumulate, stream);
    }
    hipLaunchKernelGGL(k_frame_hidden_bwd<true>, dim3(blocks), dim3(FP_THREADS), 0, stream, y, w3, base, base_ld, extra, wx, gamma,
                       dhn, E, seed, ew_threshold(p), drop_inv_keep(p), eps, dy, dbase, dextra, slab, (int)w_ld);
    EQH_CHECK_LAUNCH();
    int rc = eqh_reduce_slabs3_async(slab, blocks, FH_SLAB, dw3, dgamma, dbeta, 768, 128, accumulate, stream);
    if (rc) return rc;
    return eqh_reduce_slabs3_async(slab + (size_t)blocks * FH_SLAB, blocks, 512, dbase, dwx, nullptr, 256, 256, accumulate, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// Attention-weighted neighbour sum:  out[n, c] = sum_m attn[n, c / D, m] * x[n, m, c]      (c < H * D, m < K)
//
// Replaces (reference) the two einsums of fa_former_layer.py:497-506, "nhm,nmhd->nhd" over the gathered values and
// over the edge values -- which torch runs as 2 x 30 k batched [1 x 16] x [16 x 128] products (0.46 ms each, 4.6 ms
// per step with their backward) for what is one pass over x.  HBM-bound: 4 C K N bytes read, 4 C N written.
// One sub-group of C / 4 lanes per node (float4 per lane; a wavefront holds 256 / C nodes), the K attention weights
// of the lane's head in registers.  Backward: dx[n, m, c] = attn * dout[n, c];  dattn[n, h, m] = sum over the head's
// channels of dout * x, a butterfly over the D / 4 lanes of the head (fixed order).  No atomics.
namespace {

constexpr int AS_MAXK = 16;

template <bool BWD>
__global__ void __launch_bounds__(256)
k_attn_sum(const float* __restrict__ attn, const float* __restrict__ x, const float* __restrict__ dout, int64_t N,
           int K, int H, int D, float* __restrict__ out, float* __restrict__ dx, float* __restrict__ dattn) {
    const int C = H * D, LPR = C / 4;                         // lanes per node
    const int lane = threadIdx.x & 63;
    const int per_wave = 64 / LPR;
    const int sub = lane / LPR, sl = lane % LPR;
    const int c = 4 * sl, h = c / D;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t n = wave * per_wave + sub;
    const bool live = n < N;
    const int64_t nc = live ? n : 0;
    float a[AS_MAXK];
#pragma unroll
    for (int m = 0; m < AS_MAXK; ++m) a[m] = (m < K) ? attn[(nc * H + h) * K + m] : 0.f;
    const float* xr = x + nc * K * C + c;
    if (!BWD) {
        float4 acc = f4_zero();
#pragma unroll 4
        for (int m = 0; m < K; ++m) f4_fma(acc, *reinterpret_cast<const float4*>(xr + (int64_t)m * C), a[m]);
        if (live) *reinterpret_cast<float4*>(out + n * C + c) = acc;
    } else {
        const float4 g = *reinterpret_cast<const float4*>(dout + nc * C + c);
        float* dxr = dx + nc * K * C + c;
        const int hl = D / 4;                                  // lanes per head (power of two)
        for (int m = 0; m < K; ++m) {
            const float4 v = *reinterpret_cast<const float4*>(xr + (int64_t)m * C);
            if (live) *reinterpret_cast<float4*>(dxr + (int64_t)m * C) = make_float4(a[m] * g.x, a[m] * g.y, a[m] * g.z, a[m] * g.w);
            float p = (g.x * v.x + g.y * v.y) + (g.z * v.z + g.w * v.w);
            for (int o = 1; o < hl; o <<= 1) p += __shfl_xor(p, o, 64);
            if (live && (sl % hl) == 0) dattn[(n * H + h) * K + m] = p;
        }
    }
}

// The same sum over GATHERED rows: out[n, c] = sum_m attn[n, c / D, m] * x[nbr[n, m], c] for node values x [*, C] (row
// stride x_ld: x may be a column block of a wider product).  The gathered [N, K, C] tensor (258 MB at the Molecule3D
// batch) is neither written by a gather kernel nor read back here: the 16 MB table of node rows stays in L2.
// BWD (receiver side): dattn[n, h, m] = sum over the head's channels of dout[n, c] * x[nbr[n, m], c].
template <bool BWD>
__global__ void __launch_bounds__(256)
k_attn_gsum(const float* __restrict__ attn, const float* __restrict__ x, int64_t x_ld, const int32_t* __restrict__ nbr,
            const float* __restrict__ dout, int64_t N, int K, int H, int D, float* __restrict__ out,
            float* __restrict__ dattn) {
    const int C = H * D, LPR = C / 4;
    const int lane = threadIdx.x & 63;
    const int per_wave = 64 / LPR;
    const int sub = lane / LPR, sl = lane % LPR;
    const int c = 4 * sl, h = c / D;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t n = wave * per_wave + sub;
    const bool live = n < N;
    const int64_t nc = live ? n : 0;
    int64_t row[AS_MAXK];
#pragma unroll
    for (int m = 0; m < AS_MAXK; ++m) row[m] = (m < K) ? (int64_t)nbr[nc * K + m] * x_ld + c : 0;
    if (!BWD) {
        float a[AS_MAXK];
#pragma unroll
        for (int m = 0; m < AS_MAXK; ++m) a[m] = (m < K) ? attn[(nc * H + h) * K + m] : 0.f;
        float4 acc = f4_zero();
#pragma unroll 4
        for (int m = 0; m < K; ++m) f4_fma(acc, *reinterpret_cast<const float4*>(x + row[m]), a[m]);
        if (live) *reinterpret_cast<float4*>(out + n * C + c) = acc;
    } else {
        const float4 g = *reinterpret_cast<const float4*>(dout + nc * C + c);
        const int hl = D / 4;
#pragma unroll 4
        for (int m = 0; m < K; ++m) {
            const float4 v = *reinterpret_cast<const float4*>(x + row[m]);
            float p = (g.x * v.x + g.y * v.y) + (g.z * v.z + g.w * v.w);
            for (int o = 1; o < hl; o <<= 1) p += __shfl_xor(p, o, 64);
            if (live && (sl % hl) == 0) dattn[(n * H + h) * K + m] = p;
        }
    }
}

// BWD (sender side): dx[j, c] = sum over the edges e = (n, m) that read node j of attn[n, h, m] * dout[n, c] -- the rows of
// the transposed neighbour CSR (rowptr, perm: entry ids n * K + m, ascending: fixed summation order), four entries in
// flight per step (an entry is a chain of three loads).
__global__ void __launch_bounds__(256)
k_attn_gsum_send(const float* __restrict__ attn, const float* __restrict__ dout, const int32_t* __restrict__ rowptr,
                 const int32_t* __restrict__ perm, int64_t NS, int K, int H, int D, float* __restrict__ dx) {
    const int C = H * D, LPR = C / 4;
    const int lane = threadIdx.x & 63;
    const int per_wave = 64 / LPR;
    const int sub = lane / LPR, sl = lane % LPR;
    const int c = 4 * sl, h = c / D;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t j = wave * per_wave + sub;
    if (j >= NS) return;
    const int beg = rowptr[j], end = rowptr[j + 1];
    float4 acc = f4_zero();
    int p = beg;
    for (; p + 4 <= end; p += 4) {
        int e[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) e[u] = perm[p + u];
        float a[4];
        float4 g[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t n = e[u] / K;
            const int m = e[u] - (int)n * K;
            a[u] = attn[(n * H + h) * K + m];
            g[u] = *reinterpret_cast<const float4*>(dout + n * C + c);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) f4_fma(acc, g[u], a[u]);
    }
    for (; p < end; ++p) {
        const int e = perm[p];
        const int64_t n = e / K;
        const int m = e - (int)n * K;
        f4_fma(acc, *reinterpret_cast<const float4*>(dout + n * C + c), attn[(n * H + h) * K + m]);
    }
    *reinterpret_cast<float4*>(dx + j * C + c) = acc;
}

int attn_sum_check(int64_t N, int K, int H, int D) {
    if (N < 0 || K < 1 || K > AS_MAXK || H < 1 || D < 4) return EQH_ERR_ARG;
    const int C = H * D, lpr = C / 4, hl = D / 4;
    if ((D & 3) || lpr > 64 || (lpr & (lpr - 1)) || (hl & (hl - 1))) return EQH_ERR_ARG;
    return EQH_OK;
}

}  // namespace

extern "C" int faf_attn_sum_fwd(const float* attn, const float* x, int64_t N, int32_t K, int32_t H, int32_t D, float* out,
                                void* stream) {
    const int rc = attn_sum_check(N, K, H, D);
    if (rc) return rc;
    if (N == 0) return EQH_OK;
    if (!attn || !x || !out) return EQH_ERR_ARG;
    const int per_wave = 64 / (H * D / 4);
    const int64_t waves = (N + per_wave - 1) / per_wave;
    hipLaunchKernelGGL(k_attn_sum<false>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, attn, x,
                       (const float*)nullptr, N, (int)K, (int)H, (int)D, out, (float*)nullptr, (float*)nullptr);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}

extern "C" int faf_attn_sum_bwd(const float* attn, const float* x, const float* dout, int64_t N, int32_t K, int32_t H,
                                int32_t D, float* dx, float* dattn, void* stream) {
    const int rc = attn_sum_check(N, K, H, D);
    if (rc) return rc;
    if (N == 0) return EQH_OK;
    if (!attn || !x || !dout || !dx || !dattn) return EQH_ERR_ARG;
    const int per_wave = 64 / (H * D / 4);
    const int64_t waves = (N + per_wave - 1) / per_wave;
    hipLaunchKernelGGL(k_attn_sum<true>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, attn, x, dout,
                       N, (int)K, (int)H, (int)D, (float*)nullptr, dx, dattn);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}

/* the gathered form (see k_attn_gsum): x [NS, H*D] node rows with row stride x_ld floats, nbr [N, K] int32 (< NS);
   bwd: dattn [N, H, K] and dx [NS, H*D] through the transposed neighbour CSR (rowptr [NS + 1], perm: entry ids) */
extern "C" int faf_attn_gather_sum_fwd(const float* attn, const float* x, int64_t x_ld, const int32_t* nbr, int64_t N,
                                       int32_t K, int32_t H, int32_t D, float* out, void* stream) {
    const int rc = attn_sum_check(N, K, H, D);
    if (rc) return rc;
    if (N == 0) return EQH_OK;
    if (!attn || !x || !nbr || !out || x_ld < (int64_t)H * D || (x_ld & 3) || !eqh_aligned16(x)) return EQH_ERR_ARG;
    const int per_wave = 64 / (H * D / 4);
    const int64_t waves = (N + per_wave - 1) / per_wave;
    hipLaunchKernelGGL(k_attn_gsum<false>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, attn, x, x_ld,
                       nbr, (const float*)nullptr, N, (int)K, (int)H, (int)D, out, (float*)nullptr);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}

extern "C" int faf_attn_gather_sum_bwd(const float* attn, const float* x, int64_t x_ld, const int32_t* nbr,
                                       const float* dout, const int32_t* rowptr, const int32_t* perm, int64_t N, int64_t NS,
                                       int32_t K, int32_t H, int32_t D, float* dattn, float* dx, void* stream) {
    const int rc = attn_sum_check(N, K, H, D);
    if (rc) return rc;
    if (NS < 0) return EQH_ERR_ARG;
    if (N > 0) {
        if (!attn || !x || !nbr || !dout || !dattn || x_ld < (int64_t)H * D || (x_ld & 3) || !eqh_aligned16(x))
            return EQH_ERR_ARG;
        const int per_wave = 64 / (H * D / 4);
        const int64_t waves = (N + per_wave - 1) / per_wave;
        hipLaunchKernelGGL(k_attn_gsum<true>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, attn, x, x_ld,
                           nbr, dout, N, (int)K, (int)H, (int)D, (float*)nullptr, dattn);
        EQH_CHECK_LAUNCH();
    }
    if (NS > 0 && dx) {
        if (!rowptr || !perm || !attn || !dout) return EQH_ERR_ARG;
        const int per_wave = 64 / (H * D / 4);
        const int64_t waves = (NS + per_wave - 1) / per_wave;
        hipLaunchKernelGGL(k_attn_gsum_send, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, attn, dout,
                           rowptr, perm, NS, (int)K, (int)H, (int)D, dx);
        EQH_CHECK_LAUNCH();
    }
    return EQH_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Row-wise dot products and the sigmoid gate built on one: Linear(C, J) with J <= 4 outputs on ~250 k edge rows
// (fa_former_layer.py:340-400 att_mlp = Linear(d, 1) + Sigmoid; :483-489 the per-head edge logits).  As GEMMs with one or
// two output columns the library needs ~2 ms each; as torch ops (multiply, then reduce) every one is two passes over a
// 250 MB tensor forward and five backward.  Here: a wavefront per row, the row in registers, J butterfly sums.
//   rowdot:  y[r, j] = x[r, :] . U[j, :] + b[j]          bwd: dx[r, :] = (dx_add[r, :] +) sum_j dy[r, j] U[j, :],  dU, (db = colsum dy)
//   gate:    out[r, :] = (res[r, :] +) xd[r, :] * sigmoid(xd[r, :] . w + b),  xd = dropout_p(x)   (dropout hash as above)
// ---------------------------------------------------------------------------------------------------------------
namespace {

constexpr int RD_THREADS = 256;
constexpr int RD_WAVES = RD_THREADS / 64;

template <int NV>
struct RdRow {
    float4 v[NV];
};
template <int NV>
__device__ __forceinline__ void rd_load(const float* __restrict__ p, int C, int lane, RdRow<NV>& r) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        r.v[i] = (c < C) ? *reinterpret_cast<const float4*>(p + c) : f4_zero();
    }
}
template <int NV>
__device__ __forceinline__ float rd_dot(const RdRow<NV>& a, const RdRow<NV>& b) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        s += (a.v[i].x * b.v[i].x + a.v[i].y * b.v[i].y) + (a.v[i].z * b.v[i].z + a.v[i].w * b.v[i].w);
    return fp_wave_sum(s);
}

template <int NV, int J>
__global__ void __launch_bounds__(RD_THREADS)
k_rowdot_fwd(const float* __restrict__ x, const float* __restrict__ U, const float* __restrict__ bias, int64_t R, int C,
             float* __restrict__ y) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    RdRow<NV> u[J];
#pragma unroll
    for (int j = 0; j < J; ++j) rd_load<NV>(U + (int64_t)j * C, C, lane, u[j]);
    for (int64_t r = (int64_t)blockIdx.x * RD_WAVES + wave; r < R; r += (int64_t)gridDim.x * RD_WAVES) {
        RdRow<NV> xr;
        rd_load<NV>(x + r * C, C, lane, xr);
        float o[J];
#pragma unroll
        for (int j = 0; j < J; ++j) o[j] = rd_dot<NV>(xr, u[j]) + (bias ? bias[j] : 0.f);
        if (lane == 0) {
#pragma unroll
            for (int j = 0; j < J; ++j) y[r * J + j] = o[j];
        }
    }
}

template <int NV, int J>
__global__ void __launch_bounds__(RD_THREADS)
k_rowdot_bwd(const float* __restrict__ x, const float* __restrict__ U, const float* __restrict__ dy,
             const float* __restrict__ dx_add, int64_t R, int C, float* __restrict__ dx, float* __restrict__ slab) {
    __shared__ float4 s_red[RD_THREADS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    RdRow<NV> u[J], au[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        rd_load<NV>(U + (int64_t)j * C, C, lane, u[j]);
#pragma unroll
        for (int i = 0; i < NV; ++i) au[j].v[i] = f4_zero();
    }
    for (int64_t r = (int64_t)blockIdx.x * RD_WAVES + wave; r < R; r += (int64_t)gridDim.x * RD_WAVES) {
        RdRow<NV> xr, acc;
        rd_load<NV>(x + r * C, C, lane, xr);
        if (dx_add) rd_load<NV>(dx_add + r * C, C, lane, acc);
        else {
#pragma unroll
            for (int i = 0; i < NV; ++i) acc.v[i] = f4_zero();
        }
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const float g = dy[r * J + j];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                f4_fma(acc.v[i], u[j].v[i], g);
                f4_fma(au[j].v[i], xr.v[i], g);
            }
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (c < C) *reinterpret_cast<float4*>(dx + r * C + c) = acc.v[i];
        }
    }
    float* __restrict__ sl = slab + (int64_t)blockIdx.x * J * C;
#pragma unroll
    for (int j = 0; j < J; ++j) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            s_red[threadIdx.x] = au[j].v[i];
            __syncthreads();
            if (wave == 0) {
                float4 t = s_red[lane];
                for (int w = 1; w < RD_WAVES; ++w) f4_add(t, s_red[w * 64 + lane]);
                const int c = (lane + 64 * i) * 4;
                if (c < C) *reinterpret_cast<float4*>(sl + (int64_t)j * C + c) = t;
            }
            __syncthreads();
        }
    }
}

template <int NV>
__device__ __forceinline__ void rd_dropout(RdRow<NV>& xr, const DropKey& seed, int64_t r, int C, int lane, uint32_t threshold,
                                           float inv_keep) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        const uint64_t e = (uint64_t)(r * C + c);
        keep_scale4(seed, e, threshold, inv_keep, xr.v[i]);
    }
}

template <int NV>
__global__ void __launch_bounds__(RD_THREADS)
k_gate_fwd(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ res,
           int64_t R, int C, const int64_t* __restrict__ seed_ptr, uint32_t threshold, float inv_keep, float* __restrict__ out) {
    const DropKey seed = drop_key(threshold ? (uint64_t)*seed_ptr : 0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    RdRow<NV> wr;
    rd_load<NV>(w, C, lane, wr);
    const float b0 = b[0];
    for (int64_t r = (int64_t)blockIdx.x * RD_WAVES + wave; r < R; r += (int64_t)gridDim.x * RD_WAVES) {
        RdRow<NV> xr, rr;
        rd_load<NV>(x + r * C, C, lane, xr);
        if (res) rd_load<NV>(res + r * C, C, lane, rr);
        if (threshold) rd_dropout<NV>(xr, seed, r, C, lane, threshold, inv_keep);
        const float g = sigmoid_fast(rd_dot<NV>(xr, wr) + b0);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 4;
            float4 o = make_float4(xr.v[i].x * g, xr.v[i].y * g, xr.v[i].z * g, xr.v[i].w * g);
            if (res) f4_add(o, rr.v[i]);
            if (c < C) *reinterpret_cast<float4*>(out + r * C + c) = o;
        }
    }
}

// slab per workgroup: [dw (C) | db, 0, 0, 0]
template <int NV>
__global__ void __launch_bounds__(RD_THREADS)
k_gate_bwd(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ dout,
           int64_t R, int C, const int64_t* __restrict__ seed_ptr, uint32_t threshold, float inv_keep, float* __restrict__ dx,
           float* __restrict__ slab, float* __restrict__ slab_dx) {
    __shared__ float4 s_red[RD_THREADS];
    __shared__ float s_b[RD_WAVES];
    const DropKey seed = drop_key(threshold ? (uint64_t)*seed_ptr : 0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    RdRow<NV> wr, aw, ax;          // ax: column sums of dx (slab_dx given) = the bias gradient of the Linear that produced x
    rd_load<NV>(w, C, lane, wr);
#pragma unroll
    for (int i = 0; i < NV; ++i) aw.v[i] = ax.v[i] = f4_zero();
    float ab = 0.f;
    const float b0 = b[0];
    for (int64_t r = (int64_t)blockIdx.x * RD_WAVES + wave; r < R; r += (int64_t)gridDim.x * RD_WAVES) {
        RdRow<NV> xr, dr;
        rd_load<NV>(x + r * C, C, lane, xr);
        rd_load<NV>(dout + r * C, C, lane, dr);
        if (threshold) rd_dropout<NV>(xr, seed, r, C, lane, threshold, inv_keep);
        const float g = sigmoid_fast(rd_dot<NV>(xr, wr) + b0);
        const float coef = rd_dot<NV>(dr, xr) * g * (1.0f - g);      // d loss / d (xd . w + b)
        ab += coef;
        RdRow<NV> dxd;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            dxd.v[i] = make_float4(fmaf(dr.v[i].x, g, coef * wr.v[i].x), fmaf(dr.v[i].y, g, coef * wr.v[i].y),
                                   fmaf(dr.v[i].z, g, coef * wr.v[i].z), fmaf(dr.v[i].w, g, coef * wr.v[i].w));
            f4_fma(aw.v[i], xr.v[i], coef);
        }
        if (threshold) rd_dropout<NV>(dxd, seed, r, C, lane, threshold, inv_keep);    // d x = keep * d xd
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (c < C) *reinterpret_cast<float4*>(dx + r * C + c) = dxd.v[i];
            f4_add(ax.v[i], dxd.v[i]);
        }
    }
    if (slab_dx) {
        float* __restrict__ sx = slab_dx + (int64_t)blockIdx.x * C;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            s_red[threadIdx.x] = ax.v[i];
            __syncthreads();
            if (wave == 0) {
                float4 t = s_red[lane];
                for (int q = 1; q < RD_WAVES; ++q) f4_add(t, s_red[q * 64 + lane]);
                const int c = (lane + 64 * i) * 4;
                if (c < C) *reinterpret_cast<float4*>(sx + c) = t;
            }
            __syncthreads();
        }
    }
    float* __restrict__ sl = slab + (int64_t)blockIdx.x * (C + 4);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        s_red[threadIdx.x] = aw.v[i];
        __syncthreads();
        if (wave == 0) {
            float4 t = s_red[lane];
            for (int q = 1; q < RD_WAVES; ++q) f4_add(t, s_red[q * 64 + lane]);
            const int c = (lane + 64 * i) * 4;
            if (c < C) *reinterpret_cast<float4*>(sl + c) = t;
        }
        __syncthreads();
    }
    if (lane == 0) s_b[wave] = ab;      // (ab is wavefront-uniform: coef comes out of a butterfly sum)
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = s_b[0];
        for (int q = 1; q < RD_WAVES; ++q) t += s_b[q];
        *reinterpret_cast<float4*>(sl + C) = make_float4(t, 0.f, 0.f, 0.f);
    }
}

inline int rd_blocks(int64_t R) { return eqh_grid_for(R, RD_WAVES * 16, 1024); }
inline int rd_check(int64_t R, int32_t C) {
    if (R < 0 || C <= 0) return EQH_ERR_ARG;
    if ((C & 3) || C > 1024) return EQH_ERR_ALIGN;
    return EQH_OK;
}
template <typename F>
int rd_dispatch(int C, F&& f) {
    if (C <= 256) return f(std::integral_constant<int, 1>{});
    if (C <= 512) return f(std::integral_constant<int, 2>{});
    return f(std::integral_constant<int, 4>{});
}

}  // namespace

extern "C" int faf_rowdot_fwd(const float* x, const float* U, const float* bias, int64_t R, int32_t C, int32_t J, float* y,
                              void* stream_) {
    int rc = rd_check(R, C);
    if (rc) return rc;
    if (J < 1 || J > 4) return EQH_ERR_ARG;
    if (R == 0) return EQH_OK;
    if (!x || !U || !y) return EQH_ERR_ARG;
    if (!eqh_aligned16(x) || !eqh_aligned16(U)) return EQH_ERR_ALIGN;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const dim3 grid(eqh_grid_for(R, RD_WAVES, 8192));
    return rd_dispatch(C, [&](auto nv) {
        constexpr int NV = decltype(nv)::value;
        switch (J) {
            case 1: hipLaunchKernelGGL((k_rowdot_fwd<NV, 1>), grid, dim3(RD_THREADS), 0, stream, x, U, bias, R, (int)C, y); break;
            case 2: hipLaunchKernelGGL((k_rowdot_fwd<NV, 2>), grid, dim3(RD_THREADS), 0, stream, x, U, bias, R, (int)C, y); break;
            case 3: hipLaunchKernelGGL((k_rowdot_fwd<NV, 3>), grid, dim3(RD_THREADS), 0, stream, x, U, bias, R, (int)C, y); break;
            default: hipLaunchKernelGGL((k_rowdot_fwd<NV, 4>), grid, dim3(RD_THREADS), 0, stream, x, U, bias, R, (int)C, y); break;
        }
        EQH_CHECK_LAUNCH();
        return EQH_OK;
    });
}

extern "C" size_t faf_rowdot_bwd_workspace_bytes(int64_t R, int32_t C, int32_t J) {
    if (R <= 0 || C <= 0 || J < 1) return 0;
    return (size_t)rd_blocks(R) * (size_t)J * (size_t)C * sizeof(float);
}

/* dx [R, C] = (dx_add, may be NULL) + dy U;  dU [J, C] overwritten or accumulated */
extern "C" int faf_rowdot_bwd(const float* x, const float* U, const float* dy, const float* dx_add, int64_t R, int32_t C,
                              int32_t J, float* dx, float* dU, int32_t accumulate, void* workspace, size_t workspace_bytes,
                              void* stream_) {
    int rc = rd_check(R, C);
    if (rc) return rc;
    if (J < 1 || J > 4 || !dU) return EQH_ERR_ARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (R == 0) return accumulate ? EQH_OK : eqh_zero_async(dU, (int64_t)J * C, stream);
    if (!x || !U || !dy || !dx || !workspace) return EQH_ERR_ARG;
    if (!eqh_aligned16(x) || !eqh_aligned16(U) || !eqh_aligned16(dx) || !eqh_aligned16(dx_add) || !eqh_aligned16(workspace))
        return EQH_ERR_ALIGN;
    if (workspace_bytes < faf_rowdot_bwd_workspace_bytes(R, C, J)) return EQH_ERR_ARG;
    const int blocks = rd_blocks(R);
    float* slab = static_cast<float*>(workspace);
    return rd_dispatch(C, [&](auto nv) {
        constexpr int NV = decltype(nv)::value;
        switch (J) {
            case 1: hipLaunchKernelGGL((k_rowdot_bwd<NV, 1>), dim3(blocks), dim3(RD_THREADS), 0, stream, x, U, dy, dx_add, R, (int)C, dx, slab); break;
            case 2: hipLaunchKernelGGL((k_rowdot_bwd<NV, 2>), dim3(blocks), dim3(RD_THREADS), 0, stream, x, U, dy, dx_add, R, (int)C, dx, slab); break;
            case 3: hipLaunchKernelGGL((k_rowdot_bwd<NV, 3>), dim3(blocks), dim3(RD_THREADS), 0, stream, x, U, dy, dx_add, R, (int)C, dx, slab); break;
            default: hipLaunchKernelGGL((k_rowdot_bwd<NV, 4>), dim3(blocks), dim3(RD_THREADS), 0, stream, x, U, dy, dx_add, R, (int)C, dx, slab); break;
        }
        EQH_CHECK_LAUNCH();
        return eqh_reduce_slabs_async(slab, blocks, (int64_t)J * C, dU, stream, accumulate);
    });
}

extern "C" int faf_gate_fwd(const float* x, const float* w, const float* b, const float* res, int64_t R, int32_t C, float p,
                            const int64_t* seed, float* out, void* stream_) {
    int rc = rd_check(R, C);
    if (rc) return rc;
    if (!(p >= 0.f) || !(p < 1.f)) return EQH_ERR_ARG;
    if (R == 0) return EQH_OK;
    if (!x || !w || !b || !out || (p > 0.f && !seed)) return EQH_ERR_ARG;
    if (!eqh_aligned16(x) || !eqh_aligned16(w) || !eqh_aligned16(res) || !eqh_aligned16(out)) return EQH_ERR_ALIGN;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    return rd_dispatch(C, [&](auto nv) {
        constexpr int NV = decltype(nv)::value;
        hipLaunchKernelGGL((k_gate_fwd<NV>), dim3(eqh_grid_for(R, RD_WAVES, 8192)), dim3(RD_THREADS), 0, stream, x, w, b, res, R,
                           (int)C, seed, ew_threshold(p), drop_inv_keep(p), out);
        EQH_CHECK_LAUNCH();
        return EQH_OK;
    });
}

extern "C" size_t faf_gate_bwd_workspace_bytes(int64_t R, int32_t C) {
    if (R <= 0 || C <= 0) return 0;
    // dw / db slabs + a discard row for their padding floats, then the slabs of the dx column sums
    return ((size_t)rd_blocks(R) * (size_t)(2 * C + 4) + 4) * sizeof(float);
}

/* dx [R, C]; dw [C] and db [1] overwritten or accumulated (the gradient of `res` is dout itself).  dx_colsum [C] (may be
   NULL): the column sums of dx, overwritten or (accumulate_colsum != 0) added to -- the bias gradient of the Linear that
   produced x, which then needs no pass of its own over dx. */
extern "C" int faf_gate_bwd(const float* x, const float* w, const float* b, const float* dout, int64_t R, int32_t C, float p,
                            const int64_t* seed, float* dx, float* dw, float* db, int32_t accumulate, float* dx_colsum,
                            int32_t accumulate_colsum, void* workspace, size_t workspace_bytes, void* stream_) {
    int rc = rd_check(R, C);
    if (rc) return rc;
    if (!(p >= 0.f) || !(p < 1.f) || !dw || !db) return EQH_ERR_ARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (R == 0) {
        if (dx_colsum && !accumulate_colsum && eqh_zero_async(dx_colsum, C, stream)) return EQH_ERR_LAUNCH;
        if (accumulate) return EQH_OK;
        if (eqh_zero_async(dw, C, stream)) return EQH_ERR_LAUNCH;
        return eqh_zero_async(db, 1, stream);
    }
    if (!x || !w || !b || !dout || !dx || !workspace || (p > 0.f && !seed)) return EQH_ERR_ARG;
    if (!eqh_aligned16(x) || !eqh_aligned16(w) || !eqh_aligned16(dout) || !eqh_aligned16(dx) || !eqh_aligned16(workspace))
        return EQH_ERR_ALIGN;
    if (workspace_bytes < faf_gate_bwd_workspace_bytes(R, C)) return EQH_ERR_ARG;
    const int blocks = rd_blocks(R);
    float* slab = static_cast<float*>(workspace);
    float* slab_dx = dx_colsum ? slab + (size_t)blocks * (C + 4) + 4 : nullptr;
    return rd_dispatch(C, [&](auto nv) {
        constexpr int NV = decltype(nv)::value;
        hipLaunchKernelGGL((k_gate_bwd<NV>), dim3(blocks), dim3(RD_THREADS), 0, stream, x, w, b, dout, R, (int)C, seed,
                           ew_threshold(p), drop_inv_keep(p), dx, slab, slab_dx);
        EQH_CHECK_LAUNCH();
        if (slab_dx)
            if (int e = eqh_reduce_slabs_async(slab_dx, blocks, C, dx_colsum, stream, accumulate_colsum ? 1 : 0)) return e;
        // segments: dw (C), db (1), the slab row's three padding floats (to the discard row)
        return eqh_reduce_slabs3_async(slab, blocks, (int64_t)C + 4, dw, db, slab + (size_t)blocks * (C + 4), C, 1, accumulate,
                                       stream);
    });
}

// ---------------------------------------------------------------------------------------------------------------
// Hidden layer of EdgeModule's edge MLP (fa_former_layer.py:340-400 with :241-289) on the kNN edges:
//   hn[i, k, :] = LayerNorm( dropout( SiLU(a) * b ) ),   [a | b] = A[i] + B[nbr[i, k]] + Cf[i, k]          (256 -> 128)
// with the first Linear of the MLP split by input block (layers of faformer.py): A = W_tok_i tok + bias and B = W_tok_j tok
// at node level, Cf = W_feat feats per edge.  As separate launches: a row gather, two broadcast adds, the SwiGLU pass and
// the row LayerNorm (five [16N, 256] tensors forward, as many backward).  Here a wavefront walks the K edges of a node:
// A[i] once, B rows gathered directly, lane = two hidden units as in k_frame_hidden; the backward recomputes the forward,
// writes d pre (the gradient of Cf, and of B through the transposed kNN CSR) and sums d A[i] over the node's edges in
// registers.
// ---------------------------------------------------------------------------------------------------------------
namespace {

struct EhRow { float2 a, b; };   // this lane's two a-channels and two b-channels of a 256-wide row

__device__ __forceinline__ EhRow eh_load(const float* __restrict__ row, int lane) {
    EhRow r;
    r.a = *reinterpret_cast<const float2*>(row + 2 * lane);
    r.b = *reinterpret_cast<const float2*>(row + 128 + 2 * lane);
    return r;
}

__global__ void __launch_bounds__(FP_THREADS)
k_edge_hidden_fwd(const float* __restrict__ A, const float* __restrict__ B, const float* __restrict__ Cf,
                  const int* __restrict__ nbr, const float* __restrict__ gamma, const float* __restrict__ beta, int64_t N, int K,
                  const int64_t* __restrict__ seed_ptr, uint32_t threshold, float inv_keep, float eps, float* __restrict__ out) {
    const DropKey seed = drop_key(threshold ? (uint64_t)*seed_ptr : 0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float g0 = gamma[2 * lane], g1 = gamma[2 * lane + 1], be0 = beta[2 * lane], be1 = beta[2 * lane + 1];
    for (int64_t i = (int64_t)blockIdx.x * FP_WAVES + wave; i < N; i += (int64_t)gridDim.x * FP_WAVES) {
        const EhRow ra = eh_load(A + i * 256, lane);
        const int my_j = (lane < K) ? nbr[i * K + lane] : 0;
        EhRow nb = eh_load(B + (int64_t)__builtin_amdgcn_readlane(my_j, 0) * 256, lane);
        EhRow nc = eh_load(Cf + (i * K) * 256, lane);
        for (int k = 0; k < K; ++k) {
            const EhRow rb = nb, rc = nc;
            const int kn = (k + 1 < K) ? k + 1 : k;
            nb = eh_load(B + (int64_t)__builtin_amdgcn_readlane(my_j, kn) * 256, lane);
            nc = eh_load(Cf + (i * K + kn) * 256, lane);
            const float a0 = (ra.a.x + rb.a.x) + rc.a.x, a1 = (ra.a.y + rb.a.y) + rc.a.y;
            const float b0 = (ra.b.x + rb.b.x) + rc.b.x, b1 = (ra.b.y + rb.b.y) + rc.b.y;
            float h0 = a0 * sigmoid_fast(a0) * b0, h1 = a1 * sigmoid_fast(a1) * b1;
            const int64_t r = i * K + k;
            if (threshold) {
                const uint64_t idx = (uint64_t)(r * 128 + 2 * lane);
                float kk0, kk1;
                keep_scale2(seed, idx, threshold, inv_keep, kk0, kk1);      // (idx is even: one hash for the pair)
                h0 *= kk0; h1 *= kk1;
            }
            const float mu = fp_wave_sum(h0 + h1) * (1.0f / 128.0f);
            const float d0 = h0 - mu, d1 = h1 - mu;
            const float rstd = 1.0f / sqrtf(fp_wave_sum(d0 * d0 + d1 * d1) * (1.0f / 128.0f) + eps);
            *reinterpret_cast<float2*>(out + r * 128 + 2 * lane) = make_float2(fmaf(g0, d0 * rstd, be0), fmaf(g1, d1 * rstd, be1));
        }
    }
}

// slab per workgroup: [d gamma (128) | d beta (128)]
__global__ void __launch_bounds__(FP_THREADS)
k_edge_hidden_bwd(const float* __restrict__ A, const float* __restrict__ B, const float* __restrict__ Cf,
                  const int* __restrict__ nbr, const float* __restrict__ gamma, const float* __restrict__ dhn, int64_t N, int K,
                  const int64_t* __restrict__ seed_ptr, uint32_t threshold, float inv_keep, float eps,
                  float* __restrict__ dpre, float* __restrict__ dA, float* __restrict__ slab) {
    __shared__ float s_red[FP_WAVES][4][64];
    const DropKey seed = drop_key(threshold ? (uint64_t)*seed_ptr : 0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float g0 = gamma[2 * lane], g1 = gamma[2 * lane + 1];
    float ag0 = 0.f, ag1 = 0.f, ab0 = 0.f, ab1 = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * FP_WAVES + wave; i < N; i += (int64_t)gridDim.x * FP_WAVES) {
        const EhRow ra = eh_load(A + i * 256, lane);
        const int my_j = (lane < K) ? nbr[i * K + lane] : 0;
        EhRow nb = eh_load(B + (int64_t)__builtin_amdgcn_readlane(my_j, 0) * 256, lane);
        EhRow nc = eh_load(Cf + (i * K) * 256, lane);
        float2 ng = *reinterpret_cast<const float2*>(dhn + (i * K) * 128 + 2 * lane);
        float sa0 = 0.f, sa1 = 0.f, sb0 = 0.f, sb1 = 0.f;      // d A[i] = sum over the node's edges of d pre
        for (int k = 0; k < K; ++k) {
            const EhRow rb = nb, rc = nc;
            const float2 g = ng;
            const int kn = (k + 1 < K) ? k + 1 : k;
            nb = eh_load(B + (int64_t)__builtin_amdgcn_readlane(my_j, kn) * 256, lane);
            nc = eh_load(Cf + (i * K + kn) * 256, lane);
            ng = *reinterpret_cast<const float2*>(dhn + (i * K + kn) * 128 + 2 * lane);
            const float a0 = (ra.a.x + rb.a.x) + rc.a.x, a1 = (ra.a.y + rb.a.y) + rc.a.y;
            const float b0 = (ra.b.x + rb.b.x) + rc.b.x, b1 = (ra.b.y + rb.b.y) + rc.b.y;
            const float sg0 = sigmoid_fast(a0), sg1 = sigmoid_fast(a1);
            const float s0 = a0 * sg0, s1 = a1 * sg1;
            const int64_t r = i * K + k;
            float k0 = 1.f, k1 = 1.f;
            if (threshold) {
                const uint64_t idx = (uint64_t)(r * 128 + 2 * lane);
                keep_scale2(seed, idx, threshold, inv_keep, k0, k1);
            }
            const float h0 = s0 * b0 * k0, h1 = s1 * b1 * k1;
            const float mu = fp_wave_sum(h0 + h1) * (1.0f / 128.0f);
            const float d0 = h0 - mu, d1 = h1 - mu;
            const float rstd = 1.0f / sqrtf(fp_wave_sum(d0 * d0 + d1 * d1) * (1.0f / 128.0f) + eps);
            const float x0 = d0 * rstd, x1 = d1 * rstd;
            ab0 += g.x; ab1 += g.y;
            ag0 = fmaf(g.x, x0, ag0); ag1 = fmaf(g.y, x1, ag1);
            const float q0 = g.x * g0, q1 = g.y * g1;
            const float m1 = fp_wave_sum(q0 + q1) * (1.0f / 128.0f);
            const float m2 = fp_wave_sum(q0 * x0 + q1 * x1) * (1.0f / 128.0f);
            const float dh0 = rstd * (q0 - m1 - x0 * m2) * k0, dh1 = rstd * (q1 - m1 - x1 * m2) * k1;
            const float da0 = dh0 * b0 * fmaf(s0, 1.0f - sg0, sg0), da1 = dh1 * b1 * fmaf(s1, 1.0f - sg1, sg1);
            const float db0 = dh0 * s0, db1 = dh1 * s1;
            *reinterpret_cast<float2*>(dpre + r * 256 + 2 * lane) = make_float2(da0, da1);
            *reinterpret_cast<float2*>(dpre + r * 256 + 128 + 2 * lane) = make_float2(db0, db1);
            sa0 += da0; sa1 += da1; sb0 += db0; sb1 += db1;
        }
        *reinterpret_cast<float2*>(dA + i * 256 + 2 * lane) = make_float2(sa0, sa1);
        *reinterpret_cast<float2*>(dA + i * 256 + 128 + 2 * lane) = make_float2(sb0, sb1);
    }
    s_red[wave][0][lane] = ag0; s_red[wave][1][lane] = ag1; s_red[wave][2][lane] = ab0; s_red[wave][3][lane] = ab1;
    __syncthreads();
    if (wave == 0) {
        float t[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            t[q] = s_red[0][q][lane];
            for (int w = 1; w < FP_WAVES; ++w) t[q] += s_red[w][q][lane];
        }
        float* __restrict__ sl = slab + (int64_t)blockIdx.x * 256;
        sl[2 * lane] = t[0]; sl[2 * lane + 1] = t[1];
        sl[128 + 2 * lane] = t[2]; sl[128 + 2 * lane + 1] = t[3];
    }
}

inline int eh_blocks(int64_t N) { return eqh_grid_for(N, FP_WAVES, 1024); }

}  // namespace

extern "C" int faf_edge_hidden_fwd(const float* A, const float* B, const float* Cf, const int32_t* nbr, const float* gamma,
                                   const float* beta, int64_t N, int32_t K, float p, const int64_t* seed, float eps, float* out,
                                   void* stream_) {
    if (N < 0 || K < 1 || K > 64 || !(p >= 0.f) || !(p < 1.f)) return EQH_ERR_ARG;
    if (N == 0) return EQH_OK;
    if (!A || !B || !Cf || !nbr || !gamma || !beta || !out || (p > 0.f && !seed)) return EQH_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(A) & 7) || (reinterpret_cast<uintptr_t>(B) & 7) || (reinterpret_cast<uintptr_t>(Cf) & 7) ||
        (reinterpret_cast<uintptr_t>(out) & 7))
        return EQH_ERR_ALIGN;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(k_edge_hidden_fwd, dim3(eqh_grid_for(N, FP_WAVES, 8192)), dim3(FP_THREADS), 0, stream, A, B, Cf, nbr, gamma,
                       beta, N, (int)K, seed, ew_threshold(p), drop_inv_keep(p), eps, out);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}

extern "C" size_t faf_edge_hidden_bwd_workspace_bytes(int64_t N) {
    if (N <= 0) return 0;
    return (size_t)eh_blocks(N) * 256 * sizeof(float);
}

/* dpre [N * K, 256] (the gradient of Cf; reduce it over the transposed neighbour CSR for d B), dA [N, 256],
   dgamma / dbeta [128] (overwritten or accumulated) */
extern "C" int faf_edge_hidden_bwd(const float* A, const float* B, const float* Cf, const int32_t* nbr, const float* gamma,
                                   const float* dhn, int64_t N, int32_t K, float p, const int64_t* seed, float eps, float* dpre,
                                   float* dA, float* dgamma, float* dbeta, int32_t accumulate, void* workspace,
                                   size_t workspace_bytes, void* stream_) {
    if (N < 0 || K < 1 || K > 64 || !(p >= 0.f) || !(p < 1.f) || !dgamma || !dbeta) return EQH_ERR_ARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (N == 0) {
        if (accumulate) return EQH_OK;
        if (eqh_zero_async(dgamma, 128, stream)) return EQH_ERR_LAUNCH;
        return eqh_zero_async(dbeta, 128, stream);
    }
    if (!A || !B || !Cf || !nbr || !gamma || !dhn || !dpre || !dA || !workspace || (p > 0.f && !seed)) return EQH_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(A) & 7) || (reinterpret_cast<uintptr_t>(B) & 7) || (reinterpret_cast<uintptr_t>(Cf) & 7) ||
        (reinterpret_cast<uintptr_t>(dhn) & 7) || (reinterpret_cast<uintptr_t>(dpre) & 7) || (reinterpret_cast<uintptr_t>(dA) & 7) ||
        !eqh_aligned16(workspace))
        return EQH_ERR_ALIGN;
    if (workspace_bytes < faf_edge_hidden_bwd_workspace_bytes(N)) return EQH_ERR_ARG;
    const int blocks = eh_blocks(N);
    float* slab = static_cast<float*>(workspace);
    hipLaunchKernelGGL(k_edge_hidden_bwd, dim3(blocks), dim3(FP_THREADS), 0, stream, A, B, Cf, nbr, gamma, dhn, N, (int)K, seed,
                       ew_threshold(p), drop_inv_keep(p), eps, dpre, dA, slab);
    EQH_CHECK_LAUNCH();
    return eqh_reduce_slabs3_async(slab, blocks, 256, dgamma, dbeta, nullptr, 128, 128, accumulate, stream);
}
