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
// MFMA operand convention (lane l: r = l & 15, q = l >> 4): A[m = r][k = q], B[k = q][n = r],
// D[m = 4 q + g][n = r] in accumulator component g.  Four MFMAs share one float4 of K: the j-th of them
// takes component j on both sides, i.e. k = 16 s + 4 q + j (a permutation of K, which a sum does not see).
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

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

// out[m][n] = sum_k A[m][k] W[n][k]: A in LDS (16 rows, stride lda), W [N][K] row-major in global memory
template <int K, int N>
__device__ __forceinline__ void rows_times_wt(const float* sA, int lda, const float* __restrict__ W, float* sOut,
                                              int ldo, int wave, int lane) {
    const int r = lane & 15, q = lane >> 4;
    for (int t = wave; t < N / 16; t += RH_WAVES) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* wrow = W + (int64_t)(16 * t + r) * K + 4 * q;
        float4 b[K / 16];
#pragma unroll
        for (int s = 0; s < K / 16; ++s) b[s] = *reinterpret_cast<const float4*>(wrow + 16 * s);
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
}

// out[m][i] = sum_o A[m][o] W[o][i]: A in LDS (stride lda), W [O][I] row-major in global memory
template <int O, int I>
__device__ __forceinline__ void rows_times_w(const float* sA, int lda, const float* __restrict__ W, float* sOut,
                                             int ldo, int wave, int lane) {
    const int r = lane & 15, q = lane >> 4;
    for (int t = wave; t < I / 16; t += RH_WAVES) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* wcol = W + (int64_t)(4 * q) * I + 16 * t + r;
#pragma unroll 4
        for (int s = 0; s < O / 16; ++s) {
            const float4 a = *reinterpret_cast<const float4*>(sA + r * lda + 16 * s + 4 * q);
            const float* w = wcol + (int64_t)(16 * s) * I;
            const float b0 = w[0], b1 = w[I], b2 = w[2 * I], b3 = w[3 * I];
            acc = mfma16(a.x, b0, acc);
            acc = mfma16(a.y, b1, acc);
            acc = mfma16(a.z, b2, acc);
            acc = mfma16(a.w, b3, acc);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) sOut[(4 * q + g) * ldo + 16 * t + r] = acc[g];
    }
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
                         DZ1 = DH1 + RH_ROWS * LH, STAT = DZ1 + RH_ROWS * LH, TOTAL = STAT + 4 * RH_ROWS;
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

    // ---- global_add_pool: wavefront w owns molecules 2w, 2w+1 of the tile; a row of X is C/4 float4 lanes, so
    // 64 / (C/4) atoms are read side by side, four deep
    constexpr int LPR = C / 4, AP = 64 / LPR;
    static_assert(LPR <= 64 && 64 % LPR == 0, "C in {16..256}, power-of-two float4 lanes");
    const int sub = lane / LPR, cl = lane - sub * LPR;
    for (int mi = 2 * wave; mi < 2 * wave + 2; ++mi) {
        const int b = b0 + mi;
        float4 acc = f4_zero();
        if (b < n_real) {
            const int beg = rowptr[b], end = rowptr[b + 1];
            for (int n = beg + sub; n < end; n += 4 * AP) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int nn = n + u * AP;
                    const int nc = nn < end ? nn : end - 1;
                    v[u] = *reinterpret_cast<const float4*>(X + (int64_t)nc * C + 4 * cl);
                    if (nn >= end) v[u] = f4_zero();
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) f4_add(acc, v[u]);
            }
        }
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1) {
            acc.x += __shfl_xor(acc.x, off);
            acc.y += __shfl_xor(acc.y, off);
            acc.z += __shfl_xor(acc.z, off);
            acc.w += __shfl_xor(acc.w, off);
        }
        if (sub == 0) *reinterpret_cast<float4*>(sX + mi * LX + 4 * cl) = acc;
    }
    __syncthreads();

    // ---- forward
    rows_times_wt<C, H>(sX, LX, w.w1, sA1, LH, wave, lane);
    __syncthreads();
    relu_ln_rows<H>(sA1, w.b1, w.g1, w.be1, sXH1, sH1, sR1, LH, eps);
    __syncthreads();
    rows_times_wt<H, H>(sH1, LH, w.w2, sA2, LH, wave, lane);
    __syncthreads();
    relu_ln_rows<H>(sA2, w.b2, w.g2, w.be2, sXH2, sH2, sR2, LH, eps);
    __syncthreads();
    {
        constexpr int CPT = H / 32;
        const int row = threadIdx.x >> 5, l = threadIdx.x & 31;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < CPT; ++j) s = fmaf(sH2[row * LH + l + 32 * j], w.w3[l + 32 * j], s);
        s = half_sum(s) + w.b3[0];
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
    if (!train) return;

    // ---- backward
    const int64_t wg = blockIdx.x;
    relu_ln_bwd_rows<H>(nullptr, w.w3, sDy, sA2, sXH2, w.g2, sR2, sDZ2, LH);
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
            loss_part[wg] = e;
        }
    }
    __syncthreads();
    ln_param_sums<H>(nullptr, w.w3, sDy, sXH2, sDZ2, LH, slab.v2 + wg * 3 * H);
    outer_rows<H, H>(sDZ2, LH, sH1, LH, slab.w2 + wg * H * H, wave, lane);
    rows_times_w<H, H>(sDZ2, LH, w.w2, sDH1, LH, wave, lane);
    __syncthreads();
    relu_ln_bwd_rows<H>(sDH1, nullptr, nullptr, sA1, sXH1, w.g1, sR1, sDZ1, LH);
    __syncthreads();
    ln_param_sums<H>(sDH1, nullptr, nullptr, sXH1, sDZ1, LH, slab.v1 + wg * 3 * H);
    outer_rows<H, C>(sDZ1, LH, sX, LX, slab.w1 + wg * (int64_t)H * C, wave, lane);
    rows_times_w<H, C>(sDZ1, LH, w.w1, sDX, LX, wave, lane);
    __syncthreads();

    // ---- dX[n, :] = dx[molecule(n), :] (zero rows for the padding molecules b >= n_real: their dy is 0)
    for (int mi = 2 * wave; mi < 2 * wave + 2; ++mi) {
        const int b = b0 + mi;
        if (b >= n_graphs) continue;
        const float4 g = *reinterpret_cast<const float4*>(sDX + mi * LX + 4 * cl);
        const int beg = rowptr[b], end = rowptr[b + 1];
        for (int n = beg + sub; n < end; n += AP) *reinterpret_cast<float4*>(dX + (int64_t)n * C + 4 * cl) = g;
    }

    // ---- loss = sum of the workgroups' partials / n_real, by the last workgroup to arrive, in workgroup order
    __shared__ int s_last;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) s_last = (atomicAdd(state, 1) == (int)gridDim.x - 1);
    __syncthreads();
    if (s_last && threadIdx.x == 0) {
        __threadfence();
        float e = 0.f;
        for (unsigned i = 0; i < gridDim.x; ++i) e += __builtin_nontemporal_load(loss_part + i);
        loss[0] = e / (float)n_real;
        *state = 0;   // ready for the next launch (graph replay)
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
    static_assert(lds + 64 <= 160 * 1024, "LDS budget (dynamic + the static ticket flag)");
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
