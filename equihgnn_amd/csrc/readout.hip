// Readout head of every wrapper on the path, forward + loss + complete backward in ONE launch:
//
//   x_b   = sum_{n in molecule b} X[n, :]                     global_add_pool, equihnn_egnn.py:167 / mhnn.py:216
//   h1    = LN1(relu(W1 x + b1)),  h2 = LN2(relu(W2 h1 + b2))   MLP(C -> H -> H -> 1), mlp.py:91-99 (norm after ReLU)
//   y_b   = w3 . h2 + b3                                        equihnn_egnn.py:168-169 (.view(-1))
//   loss  = mean_{b < n_real} (y_b - t_b)^2                     F.mse_loss, main.py:49-63
//
// At the BASELINE batch this head is 257 rows: as separate launches (pool, 3 GEMMs, 2 LayerNorms, loss, and
// their ~14 backward kernels) it cost ~150 us of a 1.73 ms step for ~75 MFLOP of work; every one of those
// launches sat at the in-graph floor.  Here one workgroup owns 16 molecules: it pools their atoms into LDS,
// runs the three layers with fp32 MFMA (16 x 16 x 4, the 16 molecules are the M dimension), forms
// dy = 2 (y - t) / n_real, walks back through the layers from the activations still in LDS, writes dX for its
// atoms and one partial slab of every parameter gradient; the slabs are summed by the library's fixed-order
// reducer (deferred into the step's batched reduction when that is active).  No atomics; bitwise reproducible.
//
// Round 6, measured with tools/readout_stamps.py (-DRH_STAMPS: s_memtime stamps of every phase): 26 us of the launch's 34 are
// the chain of phases itself -- pooling 5.4 us (rowptr -> rows -> LDS: two dependent round trips to memory), the layer products
// 3.2 + 1.7, the four LayerNorm phases ~0.9 each, the backward products 3.7 + 2.1, the weight-gradient slabs 4.0 -- and none of
// them shortens with fewer molecules per workgroup: the variant with FOUR molecules per workgroup (65 workgroups for a batch of
// 256, two wavefronts pooling one molecule) took 36.2 us against 34.1 (same box; the step 1.179 vs 1.172 ms) and left 65
// gradient slabs instead of 17 for the batched reduction (12.8 vs 9.4 us).  Dropped; in the graphed step the 17-workgroup launch
// is where the NEXT batch's index build runs on the idle CUs (trainer.GraphedTrainStep, signal point "readout").
//
// MFMA operand convention (lane l: r = l & 15, q = l >> 4): A[m = r][k = q], B[k = q][n = r],
// D[m = 4 q + g][n = r] in accumulator component g.  Four MFMAs share one float4 of K: the j-th of them
// takes component j on both sides, i.e. k = 16 s + 4 q + j (a permutation of K, which a sum does not see).
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef RH_STAMPS   // diagnostic build only (tools/readout_stamps.py): s_memtime stamps of the phases, wavefront 0 of every workgroup
__device__ unsigned long long* rh_stamp_buf = nullptr;
#define RH_STAMP(slot)                                                                                       \
    do {                                                                                                     \
        if (rh_stamp_buf && threadIdx.x == 0) rh_stamp_buf[(size_t)blockIdx.x * 32 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define RH_STAMP(slot) do { } while (0)
#endif

constexpr int RH_THREADS = 512;
constexpr int RH_WAVES = RH_THREADS / 64;
constexpr int RH_ROWS = 16;  // molecules per workgroup = the M of one MFMA tile

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

struct RhWeights {
    const float *w1, *b1, *g1, *be1, *w2, *b2, *g2, *be2, *w3, *b3;
};
struct RhSlabs {   // per-workgroup partial gradients: [n_wg][H*C], [n_wg][H*H], [n_wg][3H] x 2, [n_wg][H+4]
    float *w1, *w2, *v1, *v2, *v3;
};

// sum over the 32 lanes that share a row
__device__ __forceinline__ float half_sum(float v) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 1);
    return v;
}

// out[m][n] = sum_k A[m][k] W[n][k]: A in LDS (16 rows, stride lda), W [N][K] row-major in global memory.
// The two halves are separate so that a tile's weights can be requested a phase ahead of their use (the kernel
// is a chain of short dependent phases on 17 workgroups: every exposed memory latency is step time).
template <int K>
__device__ __forceinline__ void wt_tile_load(const float* __restrict__ W, int t, int lane, float4 (&b)[K / 16]) {
    const float* wrow = W + (int64_t)(16 * t + (lane & 15)) * K + 4 * (lane >> 4);
#pragma unroll
    for (int s = 0; s < K / 16; ++s) b[s] = *reinterpret_cast<const float4*>(wrow + 16 * s);
}
template <int K>
__device__ __forceinline__ void wt_tile_mma(const float* sA, int lda, const float4 (&b)[K / 16], float* sOut, int ldo,
                                            int t, int lane) {
    const int r = lane & 15, q = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < K / 16; ++s) {
        const float4 a = *reinterpret_cast<const float4*>(sA + r * lda + 16 * s + 4 * q);
        acc = mfma16(a.x, b[s].x, acc);
        acc = mfma16(a.y, b[s].y, acc);
        acc = mfma16(a.z, b[s].z, acc);
        acc = mfma16(a.w, b[s].w, acc);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) sOut[(4 * q + g) * ldo + 16 * t + r] = acc[g];
}

// out[m][i] = sum_o A[m][o] W[o][i]: A in LDS (stride lda), W [O][I] row-major in global memory
template <int O, int I>
__device__ __forceinline__ void w_cols_load(const float* __restrict__ W, int t, int lane, float (&b)[O / 4]) {
    const float* wcol = W + (int64_t)(4 * (lane >> 4)) * I + 16 * t + (lane & 15);
#pragma unroll
    for (int s = 0; s < O / 16; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) b[4 * s + j] = wcol[(int64_t)(16 * s + j) * I];
}
template <int O>
__device__ __forceinline__ void w_cols_mma(const float* sA, int lda, const float (&b)[O / 4], float* sOut, int ldo,
                                           int t, int lane) {
    const int r = lane & 15, q = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < O / 16; ++s) {
        const float4 a = *reinterpret_cast<const float4*>(sA + r * lda + 16 * s + 4 * q);
        acc = mfma16(a.x, b[4 * s + 0], acc);
        acc = mfma16(a.y, b[4 * s + 1], acc);
        acc = mfma16(a.z, b[4 * s + 2], acc);
        acc = mfma16(a.w, b[4 * s + 3], acc);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) sOut[(4 * q + g) * ldo + 16 * t + r] = acc[g];
}

// out[o][i] = sum_{m < 16} D[m][o] In[m][i] (both in LDS) -> this workgroup's slab [O][I] in global memory
template <int O, int I>
__device__ __forceinline__ void outer_rows(const float* sD, int ldd, const float* sIn, int ldi,
                                           float* __restrict__ out, int wave, int lane) {
    const int r = lane & 15, q = lane >> 4;
    for (int t = wave; t < (O / 16) * (I / 16); t += RH_WAVES) {
        const int to = t / (I / 16), ti = t - to * (I / 16);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc = mfma16(sD[(4 * q + j) * ldd + 16 * to + r], sIn[(4 * q + j) * ldi + 16 * ti + r], acc);
#pragma unroll
        for (int g = 0; g < 4; ++g) out[(int64_t)(16 * to + 4 * q + g) * I + 16 * ti + r] = acc[g];
    }
}

// in place: sZ <- a = relu(z + bias); sXH <- (a - mean) * rstd; sH <- sXH * gamma + beta; 32 lanes per row
template <int H>
__device__ __forceinline__ void relu_ln_rows(float* sZ, const float* __restrict__ bias,
                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                             float* sXH, float* sH, float* sRstd, int ld, float eps) {
    constexpr int CPT = H / 32;
    const int row = threadIdx.x >> 5, l = threadIdx.x & 31;
    float a[CPT];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int c = l + 32 * j;
        a[j] = fmaxf(sZ[row * ld + c] + bias[c], 0.f);
        s += a[j];
    }
    const float mu = half_sum(s) * (1.0f / H);
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < CPT; ++j) v = fmaf(a[j] - mu, a[j] - mu, v);
    const float rstd = 1.0f / sqrtf(half_sum(v) * (1.0f / H) + eps);
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int c = l + 32 * j;
        const float xh = (a[j] - mu) * rstd;
        sZ[row * ld + c] = a[j];
        sXH[row * ld + c] = xh;
        sH[row * ld + c] = fmaf(xh, gamma[c], beta[c]);
    }
    if (l == 0) sRstd[row] = rstd;
}

// dz = [a > 0] * rstd * (dxh - mean(dxh) - xh * mean(dxh * xh)), dxh = dh * gamma; dh from sDH, or (sDH null)
// dh[m][c] = dy[m] * w3[c] for the last hidden layer
template <int H>
__device__ __forceinline__ void relu_ln_bwd_rows(const float* sDH, const float* __restrict__ w3, const float* sDy,
                                                 const float* sA, const float* sXH, const float* __restrict__ gamma,
                                                 const float* sRstd, float* sDZ, int ld) {
    constexpr int CPT = H / 32;
    const int row = threadIdx.x >> 5, l = threadIdx.x & 31;
    float dxh[CPT], xh[CPT];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int c = l + 32 * j;
        const float dh = sDH ? sDH[row * ld + c] : sDy[row] * w3[c];
        xh[j] = sXH[row * ld + c];
        dxh[j] = dh * gamma[c];
        s1 += dxh[j];
        s2 = fmaf(dxh[j], xh[j], s2);
    }
    s1 = half_sum(s1) * (1.0f / H);
    s2 = half_sum(s2) * (1.0f / H);
    const float rstd = sRstd[row];
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int c = l + 32 * j;
        const float da = rstd * (dxh[j] - s1 - xh[j] * s2);
        sDZ[row * ld + c] = sA[row * ld + c] > 0.f ? da : 0.f;
    }
}

// this workgroup's [dbias | dgamma | dbeta] (3 H floats): column sums over the 16 rows, in row order
template <int H>
__device__ __forceinline__ void ln_param_sums(const float* sDH, const float* __restrict__ w3, const float* sDy,
                                              const float* sXH, const float* sDZ, int ld, float* __restrict__ out) {
    static_assert(3 * H <= RH_THREADS, "one thread per (quantity, column)");
    const int t = threadIdx.x;
    if (t >= 3 * H) return;
    const int which = t / H, c = t - which * H;
    float s = 0.f;
    if (which == 0) {
        for (int m = 0; m < RH_ROWS; ++m) s += sDZ[m * ld + c];
    } else {
        const float wc = sDH ? 0.f : w3[c];
        for (int m = 0; m < RH_ROWS; ++m) {
            const float dh = sDH ? sDH[m * ld + c] : sDy[m] * wc;
            s += which == 1 ? dh * sXH[m * ld + c] : dh;
        }
    }
    out[t] = s;
}

template <int C, int H>
struct RhLds {
    static constexpr int LX = C + 4, LH = H + 4;
    static constexpr int X = 0, DX = X + RH_ROWS * LX, A1 = DX + RH_ROWS * LX, XH1 = A1 + RH_ROWS * LH,
                         H1 = XH1 + RH_ROWS * LH, A2 = H1 + RH_ROWS * LH, XH2 = A2 + RH_ROWS * LH,
                         H2 = XH2 + RH_ROWS * LH, DZ2 = H2 + RH_ROWS * LH, DH1 = DZ2 + RH_ROWS * LH,
                         DZ1 = DH1 + RH_ROWS * LH, STAT = DZ1 + RH_ROWS * LH, VEC = STAT + 4 * RH_ROWS,
                         TOTAL = VEC + 7 * H + 4;   // VEC: b1 g1 be1 b2 g2 be2 w3 b3
};

template <int C, int H>
__global__ void __launch_bounds__(RH_THREADS)
k_readout_mse(const float* __restrict__ X, const int* __restrict__ rowptr, int n_graphs, int n_real, RhWeights w,
              float eps, const float* __restrict__ target, float* __restrict__ y, float* __restrict__ loss,
              float* __restrict__ dX, RhSlabs slab, float* __restrict__ loss_part, int* __restrict__ state) {
    using L = RhLds<C, H>;
    constexpr int LX = L::LX, LH = L::LH;
    extern __shared__ __attribute__((aligned(16))) float s_mem[];
    float* sX = s_mem + L::X;
    float* sDX = s_mem + L::DX;
    float* sA1 = s_mem + L::A1;
    float* sXH1 = s_mem + L::XH1;
    float* sH1 = s_mem + L::H1;
    float* sA2 = s_mem + L::A2;
    float* sXH2 = s_mem + L::XH2;
    float* sH2 = s_mem + L::H2;
    float* sDZ2 = s_mem + L::DZ2;
    float* sDH1 = s_mem + L::DH1;
    float* sDZ1 = s_mem + L::DZ1;
    float* sR1 = s_mem + L::STAT;
    float* sR2 = sR1 + RH_ROWS;
    float* sDy = sR2 + RH_ROWS;
    float* sSq = sDy + RH_ROWS;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b0 = blockIdx.x * RH_ROWS;
    const bool train = target != nullptr;
    RH_STAMP(0);

    // ---- requests issued first: the small parameter vectors (to LDS) and this wavefront's tile of W1
    float* sV = s_mem + L::VEC;
    const float *vb1 = sV, *vg1 = sV + H, *vbe1 = sV + 2 * H, *vb2 = sV + 3 * H, *vg2 = sV + 4 * H,
                *vbe2 = sV + 5 * H, *vw3 = sV + 6 * H, *vb3 = sV + 7 * H;
    constexpr int VPT = (7 * H + RH_THREADS - 1) / RH_THREADS;
    float vreg[VPT];
    {
        const float* src[7] = {w.b1, w.g1, w.be1, w.b2, w.g2, w.be2, w.w3};
#pragma unroll
        for (int k = 0; k < VPT; ++k) {
            const int i = threadIdx.x + k * RH_THREADS;
            vreg[k] = i < 7 * H ? src[i / H][i % H] : 0.f;
        }
    }
    const float vb3_reg = w.b3[0];
    static_assert(H / 16 <= RH_WAVES && C / 16 <= 2 * RH_WAVES, "one H tile, two C tiles per wavefront");
    const bool has_h_tile = wave < H / 16;
    float4 bw1[C / 16];
    if (has_h_tile) wt_tile_load<C>(w.w1, wave, lane, bw1);

    // ---- global_add_pool: wavefront w owns molecules 2w, 2w+1 of the tile; a row of X is C/4 float4 lanes, so
    // 64 / (C/4) atoms are read side by side, eight deep for each of the two molecules at once
    constexpr int LPR = C / 4, AP = 64 / LPR;
    static_assert(LPR <= 64 && 64 % LPR == 0, "C in {16..256}, power-of-two float4 lanes");
    const int sub = lane / LPR, cl = lane - sub * LPR;
    {
        const int m0 = b0 + 2 * wave, m1 = m0 + 1;
        int beg0 = 0, end0 = 0, beg1 = 0, end1 = 0;
        if (m0 < n_real) { beg0 = rowptr[m0]; end0 = rowptr[m0 + 1]; }
        if (m1 < n_real) { beg1 = rowptr[m1]; end1 = rowptr[m1 + 1]; }
        const int last0 = end0 > 0 ? end0 - 1 : 0, last1 = end1 > 0 ? end1 - 1 : 0;
        float4 acc0 = f4_zero(), acc1 = f4_zero();
        constexpr int DEEP = 8;   // rows in flight per molecule: a QM9 molecule (<= 29 atoms) takes <= 4 rounds
        for (int n0 = beg0 + sub, n1 = beg1 + sub; n0 < end0 || n1 < end1; n0 += DEEP * AP, n1 += DEEP * AP) {
            float4 v0[DEEP], v1[DEEP];
#pragma unroll
            for (int u = 0; u < DEEP; ++u) {
                const int a0 = n0 + u * AP, a1 = n1 + u * AP;
                v0[u] = *reinterpret_cast<const float4*>(X + (int64_t)(a0 < end0 ? a0 : last0) * C + 4 * cl);
                v1[u] = *reinterpret_cast<const float4*>(X + (int64_t)(a1 < end1 ? a1 : last1) * C + 4 * cl);
            }
#pragma unroll
            for (int u = 0; u < DEEP; ++u) {
                if (n0 + u * AP < end0) f4_add(acc0, v0[u]);
                if (n1 + u * AP < end1) f4_add(acc1, v1[u]);
            }
        }
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1) {
            acc0.x += __shfl_xor(acc0.x, off); acc0.y += __shfl_xor(acc0.y, off);
            acc0.z += __shfl_xor(acc0.z, off); acc0.w += __shfl_xor(acc0.w, off);
            acc1.x += __shfl_xor(acc1.x, off); acc1.y += __shfl_xor(acc1.y, off);
            acc1.z += __shfl_xor(acc1.z, off); acc1.w += __shfl_xor(acc1.w, off);
        }
        if (sub == 0) {
            *reinterpret_cast<float4*>(sX + (2 * wave) * LX + 4 * cl) = acc0;
            *reinterpret_cast<float4*>(sX + (2 * wave + 1) * LX + 4 * cl) = acc1;
        }
    }
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
        const int i = threadIdx.x + k * RH_THREADS;
        if (i < 7 * H) sV[i] = vreg[k];
    }
    if (threadIdx.x == 0) sV[7 * H] = vb3_reg;
    RH_STAMP(1);
    __syncthreads();
    RH_STAMP(2);

    // ---- forward
    float4 bw2[H / 16];
    if (has_h_tile) {
        wt_tile_mma<C>(sX, LX, bw1, sA1, LH, wave, lane);
        wt_tile_load<H>(w.w2, wave, lane, bw2);          // in flight during LayerNorm 1
    }
    __syncthreads();
    RH_STAMP(3);
    relu_ln_rows<H>(sA1, vb1, vg1, vbe1, sXH1, sH1, sR1, LH, eps);
    __syncthreads();
    RH_STAMP(4);
    float cw2[H / 4];
    if (has_h_tile) {
        wt_tile_mma<H>(sH1, LH, bw2, sA2, LH, wave, lane);
        if (train) w_cols_load<H, H>(w.w2, wave, lane, cw2);   // for d h1, three phases ahead
    }
    __syncthreads();
    RH_STAMP(5);
    relu_ln_rows<H>(sA2, vb2, vg2, vbe2, sXH2, sH2, sR2, LH, eps);
    __syncthreads();
    RH_STAMP(6);
    {
        constexpr int CPT = H / 32;
        const int row = threadIdx.x >> 5, l = threadIdx.x & 31;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < CPT; ++j) s = fmaf(sH2[row * LH + l + 32 * j], vw3[l + 32 * j], s);
        s = half_sum(s) + vb3[0];
        if (l == 0) {
            const int b = b0 + row;
            if (b < n_graphs) y[b] = s;
            float dy = 0.f, sq = 0.f;
            if (train && b < n_real) {
                const float d = s - target[b];
                dy = 2.0f * d / (float)n_real;
                sq = d * d;
            }
            sDy[row] = dy;
            sSq[row] = sq;
        }
    }
    __syncthreads();
    RH_STAMP(7);
    if (!train) return;

    // ---- backward
    const int64_t wg = blockIdx.x;
    relu_ln_bwd_rows<H>(nullptr, vw3, sDy, sA2, sXH2, vg2, sR2, sDZ2, LH);
    {   // dw3 | db3 | pad, and the squared-error partial (row order)
        float* v3 = slab.v3 + wg * (H + 4);
        const int t = threadIdx.x;
        if (t < H) {
            float s = 0.f;
            for (int m = 0; m < RH_ROWS; ++m) s = fmaf(sDy[m], sH2[m * LH + t], s);
            v3[t] = s;
        } else if (t == H) {
            float s = 0.f, e = 0.f;
            for (int m = 0; m < RH_ROWS; ++m) { s += sDy[m]; e += sSq[m]; }
            v3[H] = s;
            v3[H + 1] = 0.f; v3[H + 2] = 0.f; v3[H + 3] = 0.f;
            // agent-scope atomic store: visible to the other XCDs without a cache write-back; it has long
            // completed when this workgroup takes its ticket at the end of the kernel
            __hip_atomic_store(loss_part + wg, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    RH_STAMP(8);
    float cw1a[H / 4], cw1b[H / 4];   // W1 columns of this wavefront's (up to) two C tiles, for dx
    const bool has_c0 = wave < C / 16, has_c1 = wave + RH_WAVES < C / 16;
    if (has_h_tile) w_cols_mma<H>(sDZ2, LH, cw2, sDH1, LH, wave, lane);
    if (has_c0) w_cols_load<H, C>(w.w1, wave, lane, cw1a);
    if (has_c1) w_cols_load<H, C>(w.w1, wave + RH_WAVES, lane, cw1b);
    ln_param_sums<H>(nullptr, vw3, sDy, sXH2, sDZ2, LH, slab.v2 + wg * 3 * H);
    outer_rows<H, H>(sDZ2, LH, sH1, LH, slab.w2 + wg * H * H, wave, lane);
    RH_STAMP(9);
    __syncthreads();
    RH_STAMP(10);
    relu_ln_bwd_rows<H>(sDH1, nullptr, nullptr, sA1, sXH1, vg1, sR1, sDZ1, LH);
    __syncthreads();
    RH_STAMP(11);
    if (has_c0) w_cols_mma<H>(sDZ1, LH, cw1a, sDX, LX, wave, lane);
    if (has_c1) w_cols_mma<H>(sDZ1, LH, cw1b, sDX, LX, wave + RH_WAVES, lane);
    __syncthreads();
    RH_STAMP(12);

    // ---- dX[n, :] = dx[molecule(n), :] (zero rows for the padding molecules b >= n_real: their dy is 0)
    for (int mi = 2 * wave; mi < 2 * wave + 2; ++mi) {
        const int b = b0 + mi;
        if (b >= n_graphs) continue;
        const float4 g = *reinterpret_cast<const float4*>(sDX + mi * LX + 4 * cl);
        const int beg = rowptr[b], end = rowptr[b + 1];
        for (int n = beg + sub; n < end; n += AP) *reinterpret_cast<float4*>(dX + (int64_t)n * C + 4 * cl) = g;
    }

    RH_STAMP(13);
    // ---- the two large gradient slabs last: nothing in this kernel waits for them
    ln_param_sums<H>(sDH1, nullptr, nullptr, sXH1, sDZ1, LH, slab.v1 + wg * 3 * H);
    outer_rows<H, C>(sDZ1, LH, sX, LX, slab.w1 + wg * (int64_t)H * C, wave, lane);
    RH_STAMP(14);

    // ---- loss = sum of the workgroups' partials / n_real, by the last workgroup to arrive, in workgroup order
    if (threadIdx.x == H) {   // the thread that stored this workgroup's partial: its store is ordered before its ticket
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        const int ticket = __hip_atomic_fetch_add(state, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ticket == (int)gridDim.x - 1) {
            float e = 0.f;
            for (unsigned i = 0; i < gridDim.x; ++i)
                e += __hip_atomic_load(loss_part + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            loss[0] = e / (float)n_real;
            __hip_atomic_store(state, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        }
    }
}

struct RhPlan {
    int n_wg;
    size_t off_w1, off_w2, off_v1, off_v2, off_v3, off_loss, off_discard, total;   // in floats
};

RhPlan rh_plan(int n_graphs, int C, int H) {
    RhPlan p;
    p.n_wg = (n_graphs + RH_ROWS - 1) / RH_ROWS;
    size_t o = 0;
    auto take = [&](size_t n) { size_t at = o; o += (n + 63) & ~(size_t)63; return at; };
    p.off_w1 = take((size_t)p.n_wg * H * C);
    p.off_w2 = take((size_t)p.n_wg * H * H);
    p.off_v1 = take((size_t)p.n_wg * 3 * H);
    p.off_v2 = take((size_t)p.n_wg * 3 * H);
    p.off_v3 = take((size_t)p.n_wg * (H + 4));
    p.off_loss = take((size_t)p.n_wg);
    p.off_discard = take(4);
    p.total = o;
    return p;
}

template <int C, int H>
int rh_launch(const float* x, const int32_t* rowptr, int n_graphs, int n_real, const RhWeights& w, float eps,
              const float* target, float* y, float* loss, float* dx, const RhSlabs& slab, float* loss_part, int* state,
              int n_wg, hipStream_t stream) {
    constexpr size_t lds = (size_t)RhLds<C, H>::TOTAL * sizeof(float);
    static_assert(lds <= 160 * 1024, "LDS budget");
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_readout_mse<C, H>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return EQH_ERR_LAUNCH;
        attr_set = true;
    }
    hipLaunchKernelGGL((k_readout_mse<C, H>), dim3(n_wg), dim3(RH_THREADS), lds, stream, x, rowptr, n_graphs, n_real, w,
                       eps, target, y, loss, dx, slab, loss_part, state);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}

}  // namespace

#ifdef RH_STAMPS
extern "C" int hg_readout_debug_stamps(void* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(rh_stamp_buf), &buf, sizeof(buf)) == hipSuccess ? EQH_OK : EQH_ERR_LAUNCH;
}
#endif

extern "C" int hg_readout_mse_supported(int32_t C, int32_t H) {
    return (C == 64 || C == 128 || C == 256) && (H == 64 || H == 128);
}

extern "C" size_t hg_readout_mse_workspace_bytes(int32_t n_graphs, int32_t C, int32_t H) {
    if (n_graphs <= 0 || !hg_readout_mse_supported(C, H)) return 0;
    return rh_plan(n_graphs, C, H).total * sizeof(float);
}

extern "C" int hg_readout_mse_f32(const float* x, const int32_t* rowptr, int32_t n_graphs, int32_t n_real, int32_t C,
                                  int32_t H, const float* const* weights, float eps, const float* target, float* y,
                                  float* loss, float* dx, float* const* dweights, int32_t accumulate, void* workspace,
                                  size_t workspace_bytes, int32_t* state, void* stream_) {
    if (n_graphs < 0 || n_real < 0 || n_real > n_graphs || !hg_readout_mse_supported(C, H)) return EQH_ERR_ARG;
    if (n_graphs == 0) return EQH_OK;
    if (!x || !rowptr || !weights || !y) return EQH_ERR_ARG;
    for (int i = 0; i < 10; ++i)
        if (!weights[i]) return EQH_ERR_ARG;
    if (!eqh_aligned16(x) || !eqh_aligned16(weights[0]) || !eqh_aligned16(weights[4])) return EQH_ERR_ALIGN;
    const bool train = target != nullptr;
    if (train) {
        if (n_real == 0 || !loss || !dx || !dweights || !workspace || !state) return EQH_ERR_ARG;
        for (int i = 0; i < 10; ++i)
            if (!dweights[i]) return EQH_ERR_ARG;
        if (!eqh_aligned16(dx) || !eqh_aligned16(workspace)) return EQH_ERR_ALIGN;
        if (workspace_bytes < hg_readout_mse_workspace_bytes(n_graphs, C, H)) return EQH_ERR_ARG;
    }
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const RhPlan p = rh_plan(n_graphs, C, H);
    float* ws = static_cast<float*>(workspace);
    RhWeights w{weights[0], weights[1], weights[2], weights[3], weights[4],
                weights[5], weights[6], weights[7], weights[8], weights[9]};
    RhSlabs slab{nullptr, nullptr, nullptr, nullptr, nullptr};
    float* loss_part = nullptr;
    if (train) {
        slab = RhSlabs{ws + p.off_w1, ws + p.off_w2, ws + p.off_v1, ws + p.off_v2, ws + p.off_v3};
        loss_part = ws + p.off_loss;
    }
    int rc = EQH_ERR_ARG;
#define RH_CASE(CC, HH)                                                                                           \
    if (C == CC && H == HH)                                                                                       \
        rc = rh_launch<CC, HH>(x, rowptr, n_graphs, n_real, w, eps, target, y, loss, dx, slab, loss_part, state, \
                               p.n_wg, stream);
    RH_CASE(64, 64) RH_CASE(64, 128) RH_CASE(128, 64) RH_CASE(128, 128) RH_CASE(256, 64) RH_CASE(256, 128)
#undef RH_CASE
    if (rc || !train) return rc;
    // parameter gradients: fixed-order sums over the workgroup slabs (deferred into the step's batched
    // reduction when that is active)
    rc = eqh_reduce_slabs_async(slab.w1, p.n_wg, (int64_t)H * C, dweights[0], stream, accumulate);
    if (rc) return rc;
    rc = eqh_reduce_slabs_async(slab.w2, p.n_wg, (int64_t)H * H, dweights[4], stream, accumulate);
    if (rc) return rc;
    rc = eqh_reduce_slabs3_async(slab.v1, p.n_wg, 3 * (int64_t)H, dweights[1], dweights[2], dweights[3], H, H,
                                 accumulate, stream);
    if (rc) return rc;
    rc = eqh_reduce_slabs3_async(slab.v2, p.n_wg, 3 * (int64_t)H, dweights[5], dweights[6], dweights[7], H, H,
                                 accumulate, stream);
    if (rc) return rc;
    return eqh_reduce_slabs3_async(slab.v3, p.n_wg, (int64_t)H + 4, dweights[8], dweights[9], ws + p.off_discard, H, 1,
                                   accumulate, stream);
}
