// Tail of the Equiformer's MLP attention (equiformer_layer.py:871-955, heads = 1): per node, over its own slot
// and its K neighbour slots,
//   logit_s = scale * w . LeakyReLU(x_s[0:4]);   attn = softmax over the valid slots (slot 0 = self is always valid,
//   :877-878; masked slots are filled with -max before the softmax, :912-915);
//   out = sum_s attn_s * (SiLU(x_s[v_off : v_off + V]) @ Wv)  =  (sum_s attn_s SiLU(x_s[v_off:])) @ Wv.
// As torch ops on [N, 17, .] tensors this is ~25 launches forward and ~40 backward at the in-graph floor; here one
// launch each way, a wavefront per node: lane s holds slot s of the softmax, lane j holds value channel j, the
// V x V matrix sits in registers (a column per lane forward, a row per lane backward) and is applied through
// v_readlane.  x_0 is the node's own row (`me`), x_1..x_K its K edge rows: the [N, 1+K, D] concatenation of the
// reference is never built.  d Wv / d w leave through per-workgroup slabs and the fixed-order reducer.
#include <float.h>

#include "common.h"

namespace {

constexpr int AP_THREADS = 256;
constexpr int AP_WAVES = AP_THREADS / 64;
constexpr int AP_K = 16;     // neighbour slots
constexpr int AP_V = 48;     // value channels
constexpr int AP_F = 4;      // logit features

template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16(float v) {
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    v += dpp_move<0x124>(v);
    v += dpp_move<0x128>(v);
    return v;
}
__device__ __forceinline__ float bcast(float v, int j) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j));
}
__device__ __forceinline__ float wave_sum(float v) {
    v = row16(v);
    return (bcast(v, 0) + bcast(v, 16)) + (bcast(v, 32) + bcast(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

__device__ __forceinline__ const float* slot_row(const float* __restrict__ me, const float* __restrict__ edge, int64_t n,
                                                 int s, int D) {
    return s == 0 ? me + n * D : edge + (n * AP_K + (s - 1)) * D;
}

// softmax weights of node n in lanes 0..K (other lanes 0); x4 = the lane's logit features
__device__ __forceinline__ float slot_softmax(const float* __restrict__ me, const float* __restrict__ edge,
                                              const float* __restrict__ mask, const float4 w, int64_t n, int lane, int D,
                                              float scale, float slope, float4* x4_out) {
    float l = -INFINITY;
    float4 x4 = f4_zero();
    if (lane <= AP_K) {
        x4 = *reinterpret_cast<const float4*>(slot_row(me, edge, n, lane, D));
        const bool valid = lane == 0 || mask[n * AP_K + lane - 1] != 0.f;
        const float a = (x4.x > 0.f ? x4.x : slope * x4.x) * w.x + (x4.y > 0.f ? x4.y : slope * x4.y) * w.y +
                        (x4.z > 0.f ? x4.z : slope * x4.z) * w.z + (x4.w > 0.f ? x4.w : slope * x4.w) * w.w;
        l = valid ? a * scale : -FLT_MAX;
    }
    *x4_out = x4;
    const float m = wave_max(l);
    const float p = lane <= AP_K ? __expf(l - m) : 0.f;
    return p / wave_sum(p);
}

__global__ void __launch_bounds__(AP_THREADS)
k_attn_pool_fwd(const float* __restrict__ me, const float* __restrict__ edge, const float* __restrict__ mask,
                const float* __restrict__ w_logit, const float* __restrict__ wv, int64_t N, int D, int v_off,
                float scale, float slope, float* __restrict__ out, float* __restrict__ attn_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane < AP_V ? lane : AP_V - 1;
    float wcol[AP_V];                                   // Wv[:, lane]
#pragma unroll
    for (int i = 0; i < AP_V; ++i) wcol[i] = wv[i * AP_V + j];
    const float4 w = *reinterpret_cast<const float4*>(w_logit);
    for (int64_t n = (int64_t)blockIdx.x * AP_WAVES + wave; n < N; n += (int64_t)gridDim.x * AP_WAVES) {
        float4 x4;
        const float attn = slot_softmax(me, edge, mask, w, n, lane, D, scale, slope, &x4);
        if (lane <= AP_K) attn_out[n * (AP_K + 1) + lane] = attn;
        float u = 0.f;                                   // sum_s attn_s SiLU(x_s[v_off + lane])
#pragma unroll
        for (int s = 0; s <= AP_K; ++s) {
            const float x = slot_row(me, edge, n, s, D)[v_off + j];
            u = fmaf(bcast(attn, s), x * sigmoid_fast(x), u);
        }
        float o0 = 0.f, o1 = 0.f;
#pragma unroll
        for (int i = 0; i < AP_V; i += 2) {
            o0 = fmaf(bcast(u, i), wcol[i], o0);
            o1 = fmaf(bcast(u, i + 1), wcol[i + 1], o1);
        }
        if (lane < AP_V) out[n * AP_V + lane] = o0 + o1;
    }
}

__global__ void __launch_bounds__(AP_THREADS)
k_attn_pool_bwd(const float* __restrict__ me, const float* __restrict__ edge, const float* __restrict__ mask,
                const float* __restrict__ w_logit, const float* __restrict__ wv, const float* __restrict__ attn_in,
                const float* __restrict__ dout, int64_t N, int D, int v_off, float scale, float slope,
                float* __restrict__ dme, float* __restrict__ dedge, float* __restrict__ slab) {
    __shared__ float s_acc[AP_WAVES][AP_V * AP_V + AP_F];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane < AP_V ? lane : AP_V - 1;
    float wrow[AP_V], dW[AP_V];                         // Wv[lane, :] and its gradient
#pragma unroll
    for (int i = 0; i < AP_V; ++i) { wrow[i] = wv[j * AP_V + i]; dW[i] = 0.f; }
    const float4 w = *reinterpret_cast<const float4*>(w_logit);
    float4 dw = f4_zero();
    const int zero4 = (v_off - AP_F) / 4;               // float4 columns between the logit features and the values
    for (int64_t n = (int64_t)blockIdx.x * AP_WAVES + wave; n < N; n += (int64_t)gridDim.x * AP_WAVES) {
        const float attn = lane <= AP_K ? attn_in[n * (AP_K + 1) + lane] : 0.f;
        const float g = dout[n * AP_V + j];
        // du[lane] = sum_c Wv[lane, c] dout[c];  u recomputed for d Wv
        float du = 0.f;
#pragma unroll
        for (int c = 0; c < AP_V; ++c) du = fmaf(wrow[c], bcast(g, c), du);
        if (lane >= AP_V) du = 0.f;
        float u = 0.f, dattn = 0.f;
#pragma unroll
        for (int s = 0; s <= AP_K; ++s) {
            const float x = slot_row(me, edge, n, s, D)[v_off + j];
            const float sig = sigmoid_fast(x);
            const float sx = x * sig;
            const float a = bcast(attn, s);
            u = fmaf(a, sx, u);
            const float red = wave_sum(du * sx);         // d attn_s (lanes >= V contribute 0)
            if (lane == s) dattn = red;
            float* drow = (s == 0 ? dme + n * D : dedge + (n * AP_K + (s - 1)) * D);
            if (lane < AP_V) drow[v_off + lane] = a * du * fmaf(sx, 1.0f - sig, sig);
            if (lane < zero4) *reinterpret_cast<float4*>(drow + AP_F + 4 * lane) = f4_zero();
            for (int c = v_off + AP_V + lane; c < D; c += 64) drow[c] = 0.f;
        }
#pragma unroll
        for (int c = 0; c < AP_V; ++c) dW[c] = fmaf(u, bcast(g, c), dW[c]);
        // softmax, LeakyReLU and the 4 -> 1 Linear
        const float dot = wave_sum(attn * dattn);
        const float dl = attn * (dattn - dot) * scale;   // masked slots: attn = 0
        if (lane <= AP_K) {
            const float4 x4 = *reinterpret_cast<const float4*>(slot_row(me, edge, n, lane, D));
            float4 d4;
            d4.x = dl * w.x * (x4.x > 0.f ? 1.f : slope); d4.y = dl * w.y * (x4.y > 0.f ? 1.f : slope);
            d4.z = dl * w.z * (x4.z > 0.f ? 1.f : slope); d4.w = dl * w.w * (x4.w > 0.f ? 1.f : slope);
            float* drow = (lane == 0 ? dme + n * D : dedge + (n * AP_K + (lane - 1)) * D);
            *reinterpret_cast<float4*>(drow) = d4;
            dw.x = fmaf(dl, x4.x > 0.f ? x4.x : slope * x4.x, dw.x); dw.y = fmaf(dl, x4.y > 0.f ? x4.y : slope * x4.y, dw.y);
            dw.z = fmaf(dl, x4.z > 0.f ? x4.z : slope * x4.z, dw.z); dw.w = fmaf(dl, x4.w > 0.f ? x4.w : slope * x4.w, dw.w);
        }
    }
    // per-wavefront partials -> LDS -> one slab per workgroup (wavefront order)
    float* mine = s_acc[wave];
    if (lane < AP_V) {
#pragma unroll
        for (int c = 0; c < AP_V; ++c) mine[lane * AP_V + c] = dW[c];
    }
    const float sx = wave_sum(dw.x), sy = wave_sum(dw.y), sz = wave_sum(dw.z), sw = wave_sum(dw.w);
    if (lane == 0) { mine[AP_V * AP_V] = sx; mine[AP_V * AP_V + 1] = sy; mine[AP_V * AP_V + 2] = sz; mine[AP_V * AP_V + 3] = sw; }
    __syncthreads();
    float* __restrict__ sl = slab + (int64_t)blockIdx.x * (AP_V * AP_V + AP_F);
    for (int i = threadIdx.x; i < AP_V * AP_V + AP_F; i += AP_THREADS) {
        float t = s_acc[0][i];
        for (int wv_ = 1; wv_ < AP_WAVES; ++wv_) t += s_acc[wv_][i];
        sl[i] = t;
    }
}

inline int ap_blocks(int64_t N) { return eqh_grid_for(N, AP_WAVES * 2, 512); }

int ap_check(int64_t N, int32_t K, int32_t D, int32_t v_off, int32_t V) {
    if (N < 0 || K != AP_K || V != AP_V) return EQH_ERR_ARG;
    if (D <= 0 || (D & 3) || v_off < AP_F || (v_off & 3) || v_off + V > D) return EQH_ERR_ARG;
    return EQH_OK;
}

}  // namespace

extern "C" int eqf_attn_pool_fwd(const float* me, const float* edge, const float* mask, const float* w_logit,
                                 const float* wv, int64_t N, int32_t K, int32_t D, int32_t v_off, int32_t V, float scale,
                                 float slope, float* out, float* attn, void* stream_) {
    int rc = ap_check(N, K, D, v_off, V);
    if (rc) return rc;
    if (N == 0) return EQH_OK;
    if (!me || !edge || !mask || !w_logit || !wv || !out || !attn) return EQH_ERR_ARG;
    if (!eqh_aligned16(me) || !eqh_aligned16(edge) || !eqh_aligned16(w_logit)) return EQH_ERR_ALIGN;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(k_attn_pool_fwd, dim3(eqh_grid_for(N, AP_WAVES, 2048)), dim3(AP_THREADS), 0, stream, me, edge, mask,
                       w_logit, wv, N, (int)D, (int)v_off, scale, slope, out, attn);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}

extern "C" size_t eqf_attn_pool_bwd_workspace_bytes(int64_t N) {
    if (N <= 0) return 0;
    return (size_t)ap_blocks(N) * (AP_V * AP_V + AP_F) * sizeof(float);
}

extern "C" int eqf_attn_pool_bwd(const float* me, const float* edge, const float* mask, const float* w_logit,
                                 const float* wv, const float* attn, const float* dout, int64_t N, int32_t K, int32_t D,
                                 int32_t v_off, int32_t V, float scale, float slope, float* dme, float* dedge,
                                 float* dw_logit, float* dwv, int32_t accumulate, void* workspace, size_t workspace_bytes,
                                 void* stream_) {
    int rc = ap_check(N, K, D, v_off, V);
    if (rc) return rc;
    if (!dw_logit || !dwv) return EQH_ERR_ARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (N == 0) {
        if (accumulate) return EQH_OK;
        if (eqh_zero_async(dw_logit, AP_F, stream)) return EQH_ERR_LAUNCH;
        return eqh_zero_async(dwv, AP_V * AP_V, stream);
    }
    if (!me || !edge || !mask || !w_logit || !wv || !attn || !dout || !dme || !dedge || !workspace) return EQH_ERR_ARG;
    if (!eqh_aligned16(me) || !eqh_aligned16(edge) || !eqh_aligned16(w_logit) || !eqh_aligned16(dme) ||
        !eqh_aligned16(dedge) || !eqh_aligned16(workspace))
        return EQH_ERR_ALIGN;
    if (workspace_bytes < eqf_attn_pool_bwd_workspace_bytes(N)) return EQH_ERR_ARG;
    const int blocks = ap_blocks(N);
    float* slab = static_cast<float*>(workspace);
    hipLaunchKernelGGL(k_attn_pool_bwd, dim3(blocks), dim3(AP_THREADS), 0, stream, me, edge, mask, w_logit, wv, attn, dout,
                       N, (int)D, (int)v_off, scale, slope, dme, dedge, slab);
    EQH_CHECK_LAUNCH();
    return eqh_reduce_slabs3_async(slab, blocks, AP_V * AP_V + AP_F, dwv, dw_logit, nullptr, AP_V * AP_V, AP_F, accumulate,
                                   stream);
}
