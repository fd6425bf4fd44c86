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

// ---- round 5: the same trunk with the 64 x 64 layer on the matrix pipe.  The kernels above walk ONE edge per wavefront step and
// form W1 h1 from 64 v_readlane + v_fma pairs (an SGPR written by a VALU instruction and read by the next one: ~2.7 k cycles
// per edge measured, 25 us per call for 39 k edges).  Here a wavefront takes 16 EDGES per step: lane (r, q) = (edge r, channel
// group q) evaluates layer 0 for its 16 channels 16 t + 4 q + i -- which is the A operand of v_mfma_f32_16x16x4f32 with the k
// index of step (t, i) = 16 t + 4 q + i -- and W1 (LDS, row stride 68) is the B operand by float4; the products land with lane
// (r, q) holding channels 16 nt + r of edges 4 q + g, where the second SiLU + LayerNorm runs (row sums by four DPP steps).
constexpr int RM_THREADS = 256;
constexpr int RM_WAVES = RM_THREADS / 64;
constexpr int RM_LD = RT_M + 4;

typedef float rt_f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float row16_sum(float v) {    // sum over the 16 lanes of a DPP row, in every lane of the row
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    v += dpp_move<0x124>(v);
    v += dpp_move<0x128>(v);
    return v;
}
__device__ __forceinline__ float q4_sum(float v) {       // sum over the four lanes r, r + 16, r + 32, r + 48
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

struct __attribute__((aligned(16))) RmShared {
    float w1[RT_M * RM_LD];
    float vec[7][RT_M];      // w0 b0 g1 be1 b1 g2 be2
};

__device__ __forceinline__ void rm_stage(RmShared& S, const RtParams& p) {
    for (int idx = threadIdx.x; idx < RT_M * RT_M / 4; idx += RM_THREADS) {
        const int row = idx >> 4, c4 = (idx & 15) * 4;
        *reinterpret_cast<float4*>(S.w1 + row * RM_LD + c4) = *reinterpret_cast<const float4*>(p.w1 + row * RT_M + c4);
    }
    const float* src[7] = {p.w0, p.b0, p.g1, p.be1, p.b1, p.g2, p.be2};
    for (int idx = threadIdx.x; idx < 7 * RT_M; idx += RM_THREADS) S.vec[idx >> 6][idx & 63] = src[idx >> 6][idx & 63];
}

// layer 0 of edge r for the lane's 16 channels (A-operand order): h1[t] = channels 16 t + 4 q .. + 3
__device__ __forceinline__ void rm_layer0(const float (&vec)[7][RT_M], float d, int q, float eps, float4 (&h1)[4]) {
    float4 a[4];
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float4 w0 = *reinterpret_cast<const float4*>(&vec[0][16 * t + 4 * q]);
        const float4 b0 = *reinterpret_cast<const float4*>(&vec[1][16 * t + 4 * q]);
        const float z0 = fmaf(d, w0.x, b0.x), z1 = fmaf(d, w0.y, b0.y), z2 = fmaf(d, w0.z, b0.z), z3 = fmaf(d, w0.w, b0.w);
        a[t] = make_float4(z0 * sigmoid_fast(z0), z1 * sigmoid_fast(z1), z2 * sigmoid_fast(z2), z3 * sigmoid_fast(z3));
        s += (a[t].x + a[t].y) + (a[t].z + a[t].w);
    }
    const float mu = q4_sum(s) * (1.0f / RT_M);
    float ss = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        a[t].x -= mu; a[t].y -= mu; a[t].z -= mu; a[t].w -= mu;
        ss += (a[t].x * a[t].x + a[t].y * a[t].y) + (a[t].z * a[t].z + a[t].w * a[t].w);
    }
    const float rstd = 1.0f / sqrtf(q4_sum(ss) * (1.0f / RT_M) + eps);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float4 g1 = *reinterpret_cast<const float4*>(&vec[2][16 * t + 4 * q]);
        const float4 be = *reinterpret_cast<const float4*>(&vec[3][16 * t + 4 * q]);
        h1[t] = make_float4(fmaf(a[t].x * rstd, g1.x, be.x), fmaf(a[t].y * rstd, g1.y, be.y), fmaf(a[t].z * rstd, g1.z, be.z),
                            fmaf(a[t].w * rstd, g1.w, be.w));
    }
}

// acc[nt][g] = sum_k h1[edge 4 q + g ... as the A operand][k] W1[16 nt + r][k]  (no bias)
__device__ __forceinline__ void rm_matmul(const float* w1, const float4 (&h1)[4], int r, int q, rt_f32x4 (&acc)[4]) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[nt] = rt_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const float4 w = *reinterpret_cast<const float4*>(w1 + (16 * nt + r) * RM_LD + 16 * t + 4 * q);
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(h1[t].x, w.x, acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(h1[t].y, w.y, acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(h1[t].z, w.z, acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(h1[t].w, w.w, acc[nt], 0, 0, 0);
        }
}

__global__ void __launch_bounds__(RM_THREADS)
k_radial_fwd_mfma(const float* __restrict__ dist, RtParams p, int64_t E, float eps, float* __restrict__ out) {
    __shared__ RmShared S;
    rm_stage(S, p);
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const int64_t tiles = (E + 15) / 16, stride = (int64_t)gridDim.x * RM_WAVES;
    for (int64_t tile = (int64_t)blockIdx.x * RM_WAVES + (threadIdx.x >> 6); tile < tiles; tile += stride) {
        const int64_t e0 = tile * 16;
        const float d = e0 + r < E ? dist[e0 + r] : 0.f;
        float4 h1[4];
        rm_layer0(S.vec, d, q, eps, h1);
        rt_f32x4 acc[4];
        rm_matmul(S.w1, h1, r, q, acc);
        // lane (r, q): channels n = 16 nt + r of edges 4 q + g
        float b1[4], g2[4], be2[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) { b1[nt] = S.vec[4][16 * nt + r]; g2[nt] = S.vec[5][16 * nt + r]; be2[nt] = S.vec[6][16 * nt + r]; }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float a[4], s = 0.f;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const float z = acc[nt][g] + b1[nt];
                a[nt] = z * sigmoid_fast(z);
                s += a[nt];
            }
            const float mu = row16_sum(s) * (1.0f / RT_M);
            float ss = 0.f;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) { a[nt] -= mu; ss = fmaf(a[nt], a[nt], ss); }
            const float rstd = 1.0f / sqrtf(row16_sum(ss) * (1.0f / RT_M) + eps);
            const int64_t e = e0 + 4 * q + g;
            if (e < E) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) out[e * RT_M + 16 * nt + r] = fmaf(a[nt] * rstd, g2[nt], be2[nt]);
            }
        }
    }
}

// Backward on the matrix pipe, 16 edges per wavefront step (the forward is recomputed from dist, as above).  Layouts of a
// [16 edges x 64 channels] tile:  AL = lane (r, q) holds edge r, channels 16 t + 4 q + i  (an A operand);
//                                 DL = lane (r, q) holds edges 4 q + g, channels 16 nt + r  (what an MFMA returns).
//   z2 = h1 W1^T          : A = h1 in AL, B = W1 by float4 from LDS                                   -> DL
//   dW1 += dz2^T h1       : A = dz2 in DL (the step index IS g), B = h1 in DL (layer 0 evaluated a second time in DL: 16
//                           SiLUs, cheaper than a transposition through LDS), accumulated in 64 registers over the steps
//   dh1 = dz2 W1          : A = dz2 in AL (through a per-wavefront LDS tile), B = W1^T by float4 from LDS -> DL
// Every LayerNorm reduction is a row sum over 16 lanes (DL) or over the four q lanes (AL).
struct __attribute__((aligned(16))) RmSharedBwd {
    float w1[RT_M * RM_LD];
    float w1t[RT_M * RM_LD];
    float vec[7][RT_M];
    float tile[RM_WAVES][16 * RM_LD];                 // dz2 of the wavefront's 16 edges, [edge][channel]
};

__device__ __forceinline__ float silu_d(float z, float* ds) {
    const float sig = sigmoid_fast(z);
    const float a = z * sig;
    *ds = fmaf(a, 1.0f - sig, sig);
    return a;
}

__global__ void __launch_bounds__(RM_THREADS)
k_radial_bwd_mfma(const float* __restrict__ dist, RtParams p, const float* __restrict__ dh, int64_t E, float eps,
                  float* __restrict__ slab) {
    __shared__ __attribute__((aligned(16))) RmSharedBwd S;
    for (int idx = threadIdx.x; idx < RT_M * RT_M / 4; idx += RM_THREADS) {
        const int row = idx >> 4, c4 = (idx & 15) * 4;
        const float4 v = *reinterpret_cast<const float4*>(p.w1 + row * RT_M + c4);
        *reinterpret_cast<float4*>(S.w1 + row * RM_LD + c4) = v;
        S.w1t[(c4 + 0) * RM_LD + row] = v.x; S.w1t[(c4 + 1) * RM_LD + row] = v.y;
        S.w1t[(c4 + 2) * RM_LD + row] = v.z; S.w1t[(c4 + 3) * RM_LD + row] = v.w;
    }
    {
        const float* src[7] = {p.w0, p.b0, p.g1, p.be1, p.b1, p.g2, p.be2};
        for (int idx = threadIdx.x; idx < 7 * RT_M; idx += RM_THREADS) S.vec[idx >> 6][idx & 63] = src[idx >> 6][idx & 63];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
    float* wt = S.tile[wave];
    // this lane's channels 16 x + r of every parameter vector (DL)
    float w0c[4], b0c[4], g1c[4], be1c[4], b1c[4], g2c[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        w0c[x] = S.vec[0][16 * x + r]; b0c[x] = S.vec[1][16 * x + r]; g1c[x] = S.vec[2][16 * x + r];
        be1c[x] = S.vec[3][16 * x + r]; b1c[x] = S.vec[4][16 * x + r]; g2c[x] = S.vec[5][16 * x + r];
    }
    rt_f32x4 accW[4][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) accW[nt][kt] = rt_f32x4{0.f, 0.f, 0.f, 0.f};
    float a_db1[4] = {0, 0, 0, 0}, a_dg2[4] = {0, 0, 0, 0}, a_dg1[4] = {0, 0, 0, 0}, a_db0[4] = {0, 0, 0, 0}, a_dw0[4] = {0, 0, 0, 0};
    const int64_t tiles = (E + 15) / 16, stride = (int64_t)gridDim.x * RM_WAVES;
    for (int64_t tile = (int64_t)blockIdx.x * RM_WAVES + wave; tile < tiles; tile += stride) {
        const int64_t e0 = tile * 16;
        const float d = e0 + r < E ? dist[e0 + r] : 0.f;
        float dg_[4];                                   // dist of the DL edges
        bool live[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            live[g] = e0 + 4 * q + g < E;
            dg_[g] = live[g] ? dist[e0 + 4 * q + g] : 0.f;
        }
        // ---- forward, recomputed: layer 0 in AL (operand of z2) and in DL (operand of dW1, layer-0 backward)
        float4 h1a[4];
        rm_layer0(S.vec, d, q, eps, h1a);
        float xh1[4][4], ds1[4][4], h1d[4][4], r1[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float a[4], s = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) { a[kt] = silu_d(fmaf(dg_[g], w0c[kt], b0c[kt]), &ds1[kt][g]); s += a[kt]; }
            const float mu = row16_sum(s) * (1.0f / RT_M);
            float ss = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) { a[kt] -= mu; ss = fmaf(a[kt], a[kt], ss); }
            r1[g] = 1.0f / sqrtf(row16_sum(ss) * (1.0f / RT_M) + eps);
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) { xh1[kt][g] = a[kt] * r1[g]; h1d[kt][g] = fmaf(xh1[kt][g], g1c[kt], be1c[kt]); }
        }
        rt_f32x4 z2[4];
        rm_matmul(S.w1, h1a, r, q, z2);
        // ---- LayerNorm 2 / SiLU 2 backward (DL)
        float dz2[4][4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float a[4], ds2[4], s = 0.f;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) { a[nt] = silu_d(z2[nt][g] + b1c[nt], &ds2[nt]); s += a[nt]; }
            const float mu = row16_sum(s) * (1.0f / RT_M);
            float ss = 0.f;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) { a[nt] -= mu; ss = fmaf(a[nt], a[nt], ss); }
            const float r2 = 1.0f / sqrtf(row16_sum(ss) * (1.0f / RT_M) + eps);
            float dx[4], xh2[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const float gr = live[g] ? dh[(e0 + 4 * q + g) * RT_M + 16 * nt + r] : 0.f;
                xh2[nt] = a[nt] * r2;
                a_dg2[nt] = fmaf(gr, xh2[nt], a_dg2[nt]);
                dx[nt] = gr * g2c[nt];
                s1 += dx[nt];
                s2 = fmaf(dx[nt], xh2[nt], s2);
            }
            const float m1 = row16_sum(s1) * (1.0f / RT_M), m2 = row16_sum(s2) * (1.0f / RT_M);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                dz2[nt][g] = r2 * (dx[nt] - m1 - xh2[nt] * m2) * ds2[nt];
                a_db1[nt] += dz2[nt][g];
            }
        }
        // ---- dW1[n][k] += sum_e dz2[e][n] h1[e][k]: both operands as they stand (DL), step s = g
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int sgn = 0; sgn < 4; ++sgn)
                    accW[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dz2[nt][sgn], h1d[kt][sgn], accW[nt][kt], 0, 0, 0);
        // ---- dh1 = dz2 W1: dz2 into AL through the wavefront's LDS tile
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) wt[(4 * q + g) * RM_LD + 16 * nt + r] = dz2[nt][g];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float4 dz2a[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) dz2a[t] = *reinterpret_cast<const float4*>(wt + r * RM_LD + 16 * t + 4 * q);
        __builtin_amdgcn_wave_barrier();
        rt_f32x4 dh1[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) dh1[kt] = rt_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const float4 w = *reinterpret_cast<const float4*>(S.w1t + (16 * kt + r) * RM_LD + 16 * t + 4 * q);
                dh1[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dz2a[t].x, w.x, dh1[kt], 0, 0, 0);
                dh1[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dz2a[t].y, w.y, dh1[kt], 0, 0, 0);
                dh1[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dz2a[t].z, w.z, dh1[kt], 0, 0, 0);
                dh1[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dz2a[t].w, w.w, dh1[kt], 0, 0, 0);
            }
        // ---- LayerNorm 1 / SiLU 1 backward (DL)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float dx[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const float gh = live[g] ? dh1[kt][g] : 0.f;
                a_dg1[kt] = fmaf(gh, xh1[kt][g], a_dg1[kt]);
                dx[kt] = gh * g1c[kt];
                s1 += dx[kt];
                s2 = fmaf(dx[kt], xh1[kt][g], s2);
            }
            const float m1 = row16_sum(s1) * (1.0f / RT_M), m2 = row16_sum(s2) * (1.0f / RT_M);
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const float dz1 = r1[g] * (dx[kt] - m1 - xh1[kt][g] * m2) * ds1[kt][g];
                a_db0[kt] += dz1;
                a_dw0[kt] = fmaf(dz1, dg_[g], a_dw0[kt]);
            }
        }
    }
    // ---- the workgroup's wavefronts in wavefront order; w1 / w1t are dead: their LDS holds the sums
    __syncthreads();
    constexpr int LDT = RT_M + 1;
    float* s_acc = S.w1;                               // [k][n] with stride LDT (RT_M * LDT <= 2 * RT_M * RM_LD floats)
    float* s_vec = S.vec[0];                           // 5 x 64 of the 7 x 64
    for (int w = 0; w < RM_WAVES; ++w) {
        if (wave == w) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int n = 16 * nt + 4 * q + g, k = 16 * kt + r;
                        s_acc[k * LDT + n] = (w == 0 ? 0.f : s_acc[k * LDT + n]) + accW[nt][kt][g];
                    }
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const float v5[RT_VEC] = {q4_sum(a_db1[x]), q4_sum(a_dg2[x]), q4_sum(a_dg1[x]), q4_sum(a_db0[x]), q4_sum(a_dw0[x])};
                if (q == 0) {
#pragma unroll
                    for (int i = 0; i < RT_VEC; ++i)
                        s_vec[i * RT_M + 16 * x + r] = (w == 0 ? 0.f : s_vec[i * RT_M + 16 * x + r]) + v5[i];
                }
            }
        }
        __syncthreads();
    }
    const int64_t nb = gridDim.x, b = blockIdx.x;
    float* __restrict__ sw = slab + b * (RT_M * RT_M);
    float* __restrict__ sv1 = slab + nb * (RT_M * RT_M) + b * (3 * RT_M);
    float* __restrict__ sv2 = slab + nb * (RT_M * RT_M + 3 * RT_M) + b * (2 * RT_M);
    for (int idx = threadIdx.x; idx < RT_M * RT_M; idx += RM_THREADS) sw[idx] = s_acc[(idx & 63) * LDT + (idx >> 6)];
    for (int idx = threadIdx.x; idx < 3 * RT_M; idx += RM_THREADS) sv1[idx] = s_vec[idx];
    for (int idx = threadIdx.x; idx < 2 * RT_M; idx += RM_THREADS) sv2[idx] = s_vec[3 * RT_M + idx];
}

inline int rm_bwd_blocks(int64_t E) { return eqh_grid_for((E + 15) / 16, RM_WAVES * 2, 256); }

inline bool rt_mfma_on() {
    static const bool on = [] { const char* e = getenv("EQH_RADIAL_READLANE"); return !(e && e[0] == '1'); }();
    return on;
}
inline int rm_blocks(int64_t E) { return eqh_grid_for((E + 15) / 16, RM_WAVES * 2, 512); }

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
    if (rt_mfma_on())
        hipLaunchKernelGGL(k_radial_fwd_mfma, dim3(rm_blocks(E)), dim3(RM_THREADS), 0, stream, dist, p, E, eps, out);
    else
        hipLaunchKernelGGL(k_radial_fwd, dim3(rt_blocks(E)), dim3(RT_THREADS), 0, stream, dist, p, E, eps, out);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}

extern "C" size_t eqf_radial_trunk_bwd_workspace_bytes(int64_t E) {
    if (E <= 0) return 0;
    const int b = rt_blocks(E) > rm_bwd_blocks(E) ? rt_blocks(E) : rm_bwd_blocks(E);      // either kernel's slabs
    return (size_t)b * RT_SLAB * sizeof(float);
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
    const int blocks = rt_mfma_on() ? rm_bwd_blocks(E) : rt_blocks(E);
    float* slab = static_cast<float*>(workspace);
    const RtParams p{params[0], params[1], params[2], params[3], params[4], params[5], params[6], params[7]};
    if (rt_mfma_on())
        hipLaunchKernelGGL(k_radial_bwd_mfma, dim3(blocks), dim3(RM_THREADS), 0, stream, dist, p, dh, E, eps, slab);
    else
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
