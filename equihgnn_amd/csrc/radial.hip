// Trunk of the Equiformer radial network (equiformer_layer.py:451-479, `rp[0..5]`), per neighbour edge:
//
//   h1 = LN(SiLU(d * w0 + b0)) * g1 + be1          Linear(1, 64) -> SiLU -> LayerNorm(gamma, beta = 0 buffer)
//   h2 = LN(SiLU(W1 h1 + b1)) * g2 + be2           Linear(64, 64) -> SiLU -> LayerNorm
//
// (the last Linear(64, lo*li) of the network is never applied per edge: it is folded into the per-node
// contraction, csrc/rowgemm.hip).  As library calls this is 2 GEMMs, 2 SiLUs and 2 LayerNorms forward and ~10
// kernels backward on [E, 64] rows, four times per step: ~1.5 ms of a 9.4 ms step at the BASELINE batch, for
// 0.3 GFLOP.  Here ONE launch each way: a wavefront walks its edges with lane = channel; the 64 x 64 weight
// lives in registers (lane i holds row i, and in the backward also column i), the matrix-vector products read
// the other lanes' values through v_readlane (an SGPR operand: no LDS traffic), the LayerNorm sums run on the
// DPP network.  The backward recomputes the forward from d (nothing is saved but the input) and keeps the
// weight-gradient row of its lane in registers; wavefronts are combined through LDS in wavefront order and
// workgroups by the fixed-order slab reducer: bitwise reproducible.
#include "common.h"

namespace {

constexpr int RT_M = 64;           // width of the trunk = one wavefront
constexpr int RT_THREADS = 512;
constexpr int RT_WAVES = RT_THREADS / 64;
constexpr int RT_VEC = 5;          // db1 | dg2 | dg1 | db0 | dw0
constexpr int RT_SLAB = RT_M * RT_M + RT_VEC * RT_M;

template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// all-lanes sum of a wavefront: quad butterflies, row rotations, then the four row totals through scalar reads
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    v += dpp_move<0x124>(v);
    v += dpp_move<0x128>(v);
    const int bits = __float_as_int(v);
    return (__int_as_float(__builtin_amdgcn_readlane(bits, 0)) + __int_as_float(__builtin_amdgcn_readlane(bits, 16))) +
           (__int_as_float(__builtin_amdgcn_readlane(bits, 32)) + __int_as_float(__builtin_amdgcn_readlane(bits, 48)));
}
__device__ __forceinline__ float bcast(float v, int j) {   // j: compile-time constant after unrolling
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j));
}
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

struct RtParams {
    const float *w0, *b0, *g1, *be1, *w1, *b1, *g2, *be2;
};

struct RtLane {   // this lane's channel of every parameter vector
    float w0, b0, g1, be1, b1, g2, be2;
};

__device__ __forceinline__ RtLane rt_lane(const RtParams& p, int lane) {
    return RtLane{p.w0[lane], p.b0[lane], p.g1[lane], p.be1[lane], p.b1[lane], p.g2[lane], p.be2[lane]};
}

// SiLU + LayerNorm of one edge, lane = channel: returns xhat, writes the activation derivative and rstd
__device__ __forceinline__ float silu_ln(float z, float eps, float* dsilu, float* rstd) {
    const float sig = sigmoid_fast(z);
    const float a = z * sig;
    *dsilu = fmaf(a, 1.0f - sig, sig);
    const float mu = wave_sum(a) * (1.0f / RT_M);
    const float c = a - mu;
    const float r = 1.0f / sqrtf(wave_sum(c * c) * (1.0f / RT_M) + eps);
    *rstd = r;
    return c * r;
}

// y[lane] = bias + sum_j w[j] * x[j] with x[j] taken from lane j
__device__ __forceinline__ float matvec(const float (&w)[RT_M], float x, float bias) {
    float s0 = bias, s1 = 0.f;
#pragma unroll
    for (int j = 0; j < RT_M; j += 2) {
        s0 = fmaf(w[j], bcast(x, j), s0);
        s1 = fmaf(w[j + 1], bcast(x, j + 1), s1);
    }
    return s0 + s1;
}

__device__ __forceinline__ void wave_edges(int64_t E, int64_t* beg, int64_t* end) {
    const int64_t n_waves = (int64_t)gridDim.x * RT_WAVES;
    const int64_t wave = (int64_t)blockIdx.x * RT_WAVES + (threadIdx.x >> 6);
    const int64_t per = (E + n_waves - 1) / n_waves;
    *beg = wave * per < E ? wave * per : E;
    *end = *beg + per < E ? *beg + per : E;
}

__global__ void __launch_bounds__(RT_THREADS)
k_radial_fwd(const float* __restrict__ dist, RtParams p, int64_t E, float eps, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const RtLane v = rt_lane(p, lane);
    float wrow[RT_M];
#pragma unroll
    for (int j = 0; j < RT_M; j += 4) {
        const float4 t = *reinterpret_cast<const float4*>(p.w1 + lane * RT_M + j);
        wrow[j] = t.x; wrow[j + 1] = t.y; wrow[j + 2] = t.z; wrow[j + 3] = t.w;
    }
    int64_t beg, end;
    wave_edges(E, &beg, &end);
    float d_next = beg < end ? dist[beg] : 0.f;
    for (int64_t e = beg; e < end; ++e) {
        const float d = d_next;
        d_next = e + 1 < end ? dist[e + 1] : 0.f;
        float ds, r;
        const float xh1 = silu_ln(fmaf(d, v.w0, v.b0), eps, &ds, &r);
        const float h1 = fmaf(xh1, v.g1, v.be1);
        const float xh2 = silu_ln(matvec(wrow, h1, v.b1), eps, &ds, &r);
        out[e * RT_M + lane] = fmaf(xh2, v.g2, v.be2);
    }
}

__global__ void __launch_bounds__(RT_THREADS)
k_radial_bwd(const float* __restrict__ dist, RtParams p, const float* __restrict__ dh, int64_t E, float eps,
             float* __restrict__ slab) {
    __shared__ float s_acc[RT_M * (RT_M + 1) + RT_VEC * RT_M];   // dW1 transposed with a padded stride, then the vectors
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const RtLane v = rt_lane(p, lane);
    float wrow[RT_M], wcol[RT_M], dW[RT_M];
#pragma unroll
    for (int j = 0; j < RT_M; j += 4) {
        const float4 t = *reinterpret_cast<const float4*>(p.w1 + lane * RT_M + j);
        wrow[j] = t.x; wrow[j + 1] = t.y; wrow[j + 2] = t.z; wrow[j + 3] = t.w;
    }
#pragma unroll
    for (int k = 0; k < RT_M; ++k) {
        wcol[k] = p.w1[k * RT_M + lane];
        dW[k] = 0.f;
    }
    float a_db1 = 0.f, a_dg2 = 0.f, a_dg1 = 0.f, a_db0 = 0.f, a_dw0 = 0.f;
    int64_t beg, end;
    wave_edges(E, &beg, &end);
    float d_next = 0.f, g_next = 0.f;
    if (beg < end) { d_next = dist[beg]; g_next = dh[beg * RT_M + lane]; }
    for (int64_t e = beg; e < end; ++e) {
        const float d = d_next, g = g_next;
        if (e + 1 < end) { d_next = dist[e + 1]; g_next = dh[(e + 1) * RT_M + lane]; }
        // forward, recomputed
        float ds1, r1, ds2, r2;
        const float xh1 = silu_ln(fmaf(d, v.w0, v.b0), eps, &ds1, &r1);
        const float h1 = fmaf(xh1, v.g1, v.be1);
        const float xh2 = silu_ln(matvec(wrow, h1, v.b1), eps, &ds2, &r2);
        // LayerNorm 2 and SiLU 2
        a_dg2 = fmaf(g, xh2, a_dg2);
        float dx = g * v.g2;
        float m1 = wave_sum(dx) * (1.0f / RT_M), m2 = wave_sum(dx * xh2) * (1.0f / RT_M);
        const float dz2 = r2 * (dx - m1 - xh2 * m2) * ds2;
        a_db1 += dz2;
        // dW1[lane][j] += dz2[lane] * h1[j];  dh1[lane] = sum_k W1[k][lane] dz2[k]
#pragma unroll
        for (int j = 0; j < RT_M; ++j) dW[j] = fmaf(dz2, bcast(h1, j), dW[j]);
        const float dh1 = matvec(wcol, dz2, 0.f);
        // LayerNorm 1 and SiLU 1
        a_dg1 = fmaf(dh1, xh1, a_dg1);
        dx = dh1 * v.g1;
        m1 = wave_sum(dx) * (1.0f / RT_M);
        m2 = wave_sum(dx * xh1) * (1.0f / RT_M);
        const float dz1 = r1 * (dx - m1 - xh1 * m2) * ds1;
        a_db0 += dz1;
        a_dw0 = fmaf(dz1, d, a_dw0);
    }
    // the workgroup's wavefronts, in wavefront order
    constexpr int LDT = RT_M + 1;
    float* s_vec = s_acc + RT_M * LDT;
    for (int w = 0; w < RT_WAVES; ++w) {
        if (wave == w) {
#pragma unroll
            for (int j = 0; j < RT_M; ++j) s_acc[j * LDT + lane] = (w == 0 ? 0.f : s_acc[j * LDT + lane]) + dW[j];
            const float vec[RT_VEC] = {a_db1, a_dg2, a_dg1, a_db0, a_dw0};
#pragma unroll
            for (int q = 0; q < RT_VEC; ++q) s_vec[q * RT_M + lane] = (w == 0 ? 0.f : s_vec[q * RT_M + lane]) + vec[q];
        }
        __syncthreads();
    }
    // three slab regions, each contiguous per workgroup as the reducer wants them:
    // [blocks][M*M] dW1 | [blocks][3M] db1 dg2 dg1 | [blocks][2M] db0 dw0
    const int64_t nb = gridDim.x, b = blockIdx.x;
    float* __restrict__ sw = slab + b * (RT_M * RT_M);
    float* __restrict__ sv1 = slab + nb * (RT_M * RT_M) + b * (3 * RT_M);
    float* __restrict__ sv2 = slab + nb * (RT_M * RT_M + 3 * RT_M) + b * (2 * RT_M);
    for (int idx = threadIdx.x; idx < RT_M * RT_M; idx += RT_THREADS) sw[idx] = s_acc[(idx & 63) * LDT + (idx >> 6)];
    for (int idx = threadIdx.x; idx < 3 * RT_M; idx += RT_THREADS) sv1[idx] = s_vec[idx];
    for (int idx = threadIdx.x; idx < 2 * RT_M; idx += RT_THREADS) sv2[idx] = s_vec[3 * RT_M + idx];
}

inline int rt_blocks(int64_t E) { return eqh_grid_for(E, RT_WAVES * 8, 256); }

int rt_check(const float* const* params, int32_t M) {
    if (M != RT_M || !params) return EQH_ERR_ARG;
    for (int i = 0; i < 8; ++i)
        if (!params[i]) return EQH_ERR_ARG;
    if (!eqh_aligned16(params[4])) return EQH_ERR_ALIGN;
    return EQH_OK;
}

}  // namespace

extern "C" int eqf_radial_trunk_fwd(const float* dist, const float* const* params, int64_t E, int32_t M, float eps,
                                    float* out, void* stream_) {
    int rc = rt_check(params, M);
    if (rc || E < 0) return rc ? rc : EQH_ERR_ARG;
    if (E == 0) return EQH_OK;
    if (!dist || !out) return EQH_ERR_ARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const RtParams p{params[0], params[1], params[2], params[3], params[4], params[5], params[6], params[7]};
    hipLaunchKernelGGL(k_radial_fwd, dim3(rt_blocks(E)), dim3(RT_THREADS), 0, stream, dist, p, E, eps, out);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}

extern "C" size_t eqf_radial_trunk_bwd_workspace_bytes(int64_t E) {
    if (E <= 0) return 0;
    return (size_t)rt_blocks(E) * RT_SLAB * sizeof(float);
}

extern "C" int eqf_radial_trunk_bwd(const float* dist, const float* const* params, const float* dh, int64_t E,
                                    int32_t M, float eps, float* const* dparams, int32_t accumulate, void* workspace,
                                    size_t workspace_bytes, void* stream_) {
    int rc = rt_check(params, M);
    if (rc || E < 0 || !dparams) return rc ? rc : EQH_ERR_ARG;
    // dparams: {dw0, db0, dg1, dw1, db1, dg2} (the LayerNorm betas are buffers in the reference: no gradient)
    for (int i = 0; i < 6; ++i)
        if (!dparams[i]) return EQH_ERR_ARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (E == 0) {
        if (accumulate) return EQH_OK;
        const int64_t n[6] = {RT_M, RT_M, RT_M, RT_M * RT_M, RT_M, RT_M};
        for (int i = 0; i < 6; ++i)
            if (eqh_zero_async(dparams[i], n[i], stream)) return EQH_ERR_LAUNCH;
        return EQH_OK;
    }
    if (!dist || !dh || !workspace) return EQH_ERR_ARG;
    if (workspace_bytes < eqf_radial_trunk_bwd_workspace_bytes(E)) return EQH_ERR_ARG;
    const int blocks = rt_blocks(E);
    float* slab = static_cast<float*>(workspace);
    const RtParams p{params[0], params[1], params[2], params[3], params[4], params[5], params[6], params[7]};
    hipLaunchKernelGGL(k_radial_bwd, dim3(blocks), dim3(RT_THREADS), 0, stream, dist, p, dh, E, eps, slab);
    EQH_CHECK_LAUNCH();
    // dparams: {dw0, db0, dg1, dw1, db1, dg2}
    const int64_t nb = blocks;
    rc = eqh_reduce_slabs_async(slab, blocks, RT_M * RT_M, dparams[3], stream, accumulate);
    if (rc) return rc;
    rc = eqh_reduce_slabs3_async(slab + nb * (RT_M * RT_M), blocks, 3 * RT_M, dparams[4], dparams[5], dparams[2], RT_M,
                                 RT_M, accumulate, stream);
    if (rc) return rc;
    rc = eqh_reduce_slabs3_async(slab + nb * (RT_M * RT_M + 3 * RT_M), blocks, 2 * RT_M, dparams[1], dparams[0], nullptr,
                                 RT_M, RT_M, accumulate, stream);
    if (rc) return rc;
    return EQH_OK;
}
