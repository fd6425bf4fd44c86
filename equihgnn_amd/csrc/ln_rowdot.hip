// LayerNorm over dense rows with a few row-wise dot products of its OUTPUT riding along:
//   xe = LayerNorm(x),   le[r, j] = xe[r, :] . U[j, :] + cb[j]      (j < J <= 2)
// FAFormer's edge attention (fa_former_layer.py:436-441, 483-489): the LayerNorm in front of the edge Linear and the
// per-head edge logits, which are linear in the normalised edge features (folded weights U, faf_edge_logit_weights).
// As separate kernels (hg_layer_norm_* + faf_rowdot_*) the [E, C] normalised tensor was read once more forward
// (51 us at the Molecule3D batch) and, backward, the row-dot pass read xe and its two upstream gradients and wrote
// their sum for the LayerNorm backward to read again (127 us).  Here the forward forms the dot products from the row it
// has in registers, and the backward adds dle[r, j] U[j, :] to the upstream gradient on the fly and accumulates
// dU[j, :] += dle[r, j] xe[r, :] beside d gamma / d beta.  Row arithmetic of rowln.h (bit-identical LayerNorm).
#include <type_traits>
#include <utility>

#include "common.h"
#include "rowln.h"

namespace {

constexpr int LR_THREADS = 256;
constexpr int LR_WAVES = LR_THREADS / 64;

template <int NV, int J>
__global__ void __launch_bounds__(LR_THREADS)
k_ln_rowdot_fwd(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                const float* __restrict__ U, const float* __restrict__ cb, int n_rows, int C, float eps,
                float* __restrict__ out, float* __restrict__ le) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inv_c = 1.0f / (float)C;
    Row<NV> gam, bet, u[J], zero;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        const bool ok = c < C;
        gam.v[i] = ok ? *reinterpret_cast<const float4*>(gamma + c) : f4_zero();
        bet.v[i] = ok ? *reinterpret_cast<const float4*>(beta + c) : f4_zero();
        zero.v[i] = f4_zero();
#pragma unroll
        for (int j = 0; j < J; ++j) u[j].v[i] = ok ? *reinterpret_cast<const float4*>(U + (int64_t)j * C + c) : f4_zero();
    }
    for (int r = blockIdx.x * LR_WAVES + wave; r < n_rows; r += gridDim.x * LR_WAVES) {
        Row<NV> hr, xh;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 4;
            hr.v[i] = (c < C) ? *reinterpret_cast<const float4*>(x + (int64_t)r * C + c) : f4_zero();
        }
        unsigned pos;
        float rstd;
        norm_pair<NV, false>(hr, zero, C, lane, inv_c, eps, xh, pos, &rstd);
        float s[J];
#pragma unroll
        for (int j = 0; j < J; ++j) s[j] = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (c < C) {
                float4 o;
                o.x = fmaf(gam.v[i].x, xh.v[i].x, bet.v[i].x); o.y = fmaf(gam.v[i].y, xh.v[i].y, bet.v[i].y);
                o.z = fmaf(gam.v[i].z, xh.v[i].z, bet.v[i].z); o.w = fmaf(gam.v[i].w, xh.v[i].w, bet.v[i].w);
                *reinterpret_cast<float4*>(out + (int64_t)r * C + c) = o;
#pragma unroll
                for (int j = 0; j < J; ++j)
                    s[j] += (o.x * u[j].v[i].x + o.y * u[j].v[i].y) + (o.z * u[j].v[i].z + o.w * u[j].v[i].w);
            }
        }
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const float t = wave_sum(s[j]);
            if (lane == 0) le[(int64_t)r * J + j] = t + (cb ? cb[j] : 0.f);
        }
    }
}

// slabA per workgroup: [d gamma | d beta] (2 C floats); slabB per workgroup: [dU_0 | .. | dU_{J-1}] (J C floats)
template <int NV, int J>
__global__ void __launch_bounds__(LR_THREADS)
k_ln_rowdot_bwd(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                const float* __restrict__ U, const float* __restrict__ dy, int64_t dy_ld, const float* __restrict__ dle,
                const float* __restrict__ add, int n_rows, int C, float eps, float* __restrict__ dx,
                float* __restrict__ slab_a, float* __restrict__ slab_b) {
    __shared__ float4 s_red[LR_THREADS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inv_c = 1.0f / (float)C;
    Row<NV> gam, bet, u[J], zero, a_dg, a_dbeta, a_du[J];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        const bool ok = c < C;
        gam.v[i] = ok ? *reinterpret_cast<const float4*>(gamma + c) : f4_zero();
        bet.v[i] = ok ? *reinterpret_cast<const float4*>(beta + c) : f4_zero();
        zero.v[i] = a_dg.v[i] = a_dbeta.v[i] = f4_zero();
#pragma unroll
        for (int j = 0; j < J; ++j) {
            u[j].v[i] = ok ? *reinterpret_cast<const float4*>(U + (int64_t)j * C + c) : f4_zero();
            a_du[j].v[i] = f4_zero();
        }
    }
    // the operands of the next row are in flight while this one is normalised (as k_rowln_bwd)
    const int stride = gridDim.x * LR_WAVES;
    int r = blockIdx.x * LR_WAVES + wave;
    Row<NV> nh, nd, na;
    float nl[J];
#pragma unroll
    for (int i = 0; i < NV; ++i) nd.v[i] = na.v[i] = f4_zero();
    auto fetch = [&](int row) {
        const int rr = row < n_rows ? row : n_rows - 1;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 4;
            nh.v[i] = (c < C) ? *reinterpret_cast<const float4*>(x + (int64_t)rr * C + c) : f4_zero();
            if (dy) nd.v[i] = (c < C) ? *reinterpret_cast<const float4*>(dy + (int64_t)rr * dy_ld + c) : f4_zero();
            if (add) na.v[i] = (c < C) ? *reinterpret_cast<const float4*>(add + (int64_t)rr * C + c) : f4_zero();
        }
#pragma unroll
        for (int j = 0; j < J; ++j) nl[j] = dle ? dle[(int64_t)rr * J + j] : 0.f;
    };
    if (r < n_rows) fetch(r);
    for (; r < n_rows; r += stride) {
        const Row<NV> ch = nh, cd = nd, ca = na;
        float cl[J];
#pragma unroll
        for (int j = 0; j < J; ++j) cl[j] = nl[j];
        fetch(r + stride);
        Row<NV> xh, g;
        unsigned pos;
        float rstd;
        norm_pair<NV, false>(ch, zero, C, lane, inv_c, eps, xh, pos, &rstd);
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            float4 d = cd.v[i];
            const float4 xe = make_float4(fmaf(gam.v[i].x, xh.v[i].x, bet.v[i].x), fmaf(gam.v[i].y, xh.v[i].y, bet.v[i].y),
                                          fmaf(gam.v[i].z, xh.v[i].z, bet.v[i].z), fmaf(gam.v[i].w, xh.v[i].w, bet.v[i].w));
#pragma unroll
            for (int j = 0; j < J; ++j) {
                f4_fma(a_du[j].v[i], xe, cl[j]);                 // dU_j += dle_j * xe
                f4_fma(d, u[j].v[i], cl[j]);                     // the row dots' share of d xe
            }
            f4_add(a_dbeta.v[i], d);
            a_dg.v[i].x = fmaf(d.x, xh.v[i].x, a_dg.v[i].x); a_dg.v[i].y = fmaf(d.y, xh.v[i].y, a_dg.v[i].y);
            a_dg.v[i].z = fmaf(d.z, xh.v[i].z, a_dg.v[i].z); a_dg.v[i].w = fmaf(d.w, xh.v[i].w, a_dg.v[i].w);
            d.x *= gam.v[i].x; d.y *= gam.v[i].y; d.z *= gam.v[i].z; d.w *= gam.v[i].w;
            g.v[i] = d;
            m1 += (d.x + d.y) + (d.z + d.w);
            m2 += (d.x * xh.v[i].x + d.y * xh.v[i].y) + (d.z * xh.v[i].z + d.w * xh.v[i].w);
        }
        wave_sum2(m1, m2);
        m1 *= inv_c;
        m2 *= inv_c;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 4;
            float4 o;
            o.x = rstd * (g.v[i].x - m1 - xh.v[i].x * m2); o.y = rstd * (g.v[i].y - m1 - xh.v[i].y * m2);
            o.z = rstd * (g.v[i].z - m1 - xh.v[i].z * m2); o.w = rstd * (g.v[i].w - m1 - xh.v[i].w * m2);
            f4_add(o, ca.v[i]);
            if (c < C) *reinterpret_cast<float4*>(dx + (int64_t)r * C + c) = o;
        }
    }
    // the workgroup's four wavefronts in a fixed order, one slab row per quantity
    float* __restrict__ sa = slab_a + (int64_t)blockIdx.x * 2 * C;
    float* __restrict__ sb = slab_b + (int64_t)blockIdx.x * J * C;
#pragma unroll
    for (int which = 0; which < 2 + J; ++which) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            s_red[threadIdx.x] = which == 0 ? a_dg.v[i] : (which == 1 ? a_dbeta.v[i] : a_du[which - 2 < J ? which - 2 : 0].v[i]);
            __syncthreads();
            if (wave == 0) {
                float4 t = s_red[lane];
                for (int w2 = 1; w2 < LR_WAVES; ++w2) f4_add(t, s_red[w2 * 64 + lane]);
                const int c = (lane + 64 * i) * 4;
                if (c < C) {
                    if (which < 2) *reinterpret_cast<float4*>(sa + which * C + c) = t;
                    else *reinterpret_cast<float4*>(sb + (which - 2) * C + c) = t;
                }
            }
            __syncthreads();
        }
    }
}

inline int lr_blocks(int64_t rows) { return eqh_grid_for(rows, LR_WAVES * 4, rows > 65536 ? 2048 : 256); }

template <typename F>
int lr_dispatch(int C, int J, F&& f) {
    auto with_j = [&](auto nv) {
        if (J == 1) return f(nv, std::integral_constant<int, 1>{});
        return f(nv, std::integral_constant<int, 2>{});
    };
    if (C <= 256) return with_j(std::integral_constant<int, 1>{});
    if (C <= 512) return with_j(std::integral_constant<int, 2>{});
    return with_j(std::integral_constant<int, 4>{});
}

inline int lr_check(int64_t R, int32_t C, int32_t J) {
    if (R < 0 || C <= 0 || J < 1 || J > 2 || R >= ((int64_t)1 << 31)) return EQH_ERR_ARG;
    if ((C & 3) || C > 1024) return EQH_ERR_ALIGN;
    return EQH_OK;
}

}  // namespace

extern "C" int faf_ln_rowdot_fwd(const float* x, const float* gamma, const float* beta, const float* U, const float* cb,
                                 int64_t R, int32_t C, int32_t J, float eps, float* out, float* le, void* stream_) {
    int rc = lr_check(R, C, J);
    if (rc) return rc;
    if (R == 0) return EQH_OK;
    if (!x || !gamma || !beta || !U || !out || !le) return EQH_ERR_ARG;
    if (!eqh_aligned16(x) || !eqh_aligned16(gamma) || !eqh_aligned16(beta) || !eqh_aligned16(U) || !eqh_aligned16(out))
        return EQH_ERR_ALIGN;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    return lr_dispatch(C, J, [&](auto nv, auto jj) {
        constexpr int NV = decltype(nv)::value, JJ = decltype(jj)::value;
        hipLaunchKernelGGL((k_ln_rowdot_fwd<NV, JJ>), dim3(eqh_grid_for(R, LR_WAVES, 4096)), dim3(LR_THREADS), 0, stream, x, gamma,
                           beta, U, cb, (int)R, (int)C, eps, out, le);
        EQH_CHECK_LAUNCH();
        return EQH_OK;
    });
}

extern "C" size_t faf_ln_rowdot_bwd_workspace_bytes(int64_t R, int32_t C, int32_t J) {
    if (R < 0 || C <= 0 || J < 1) return 0;
    return (size_t)lr_blocks(R) * (size_t)(2 + J) * (size_t)C * sizeof(float);
}

extern "C" int faf_ln_rowdot_bwd(const float* x, const float* gamma, const float* beta, const float* U, const float* dy,
                                 int64_t dy_ld, const float* dle, const float* add, int64_t R, int32_t C, int32_t J, float eps,
                                 float* dx, float* dgamma, float* dbeta, int32_t accumulate, float* dU, void* workspace,
                                 size_t workspace_bytes, void* stream_) {
    int rc = lr_check(R, C, J);
    if (rc) return rc;
    if (!dgamma || !dbeta || !dU) return EQH_ERR_ARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (R == 0) {
        if (eqh_zero_async(dU, (int64_t)J * C, stream)) return EQH_ERR_LAUNCH;
        if (accumulate) return EQH_OK;
        if (eqh_zero_async(dgamma, C, stream)) return EQH_ERR_LAUNCH;
        return eqh_zero_async(dbeta, C, stream);
    }
    if (!x || !gamma || !beta || !U || !dx || !workspace) return EQH_ERR_ARG;
    if (!eqh_aligned16(x) || !eqh_aligned16(gamma) || !eqh_aligned16(beta) || !eqh_aligned16(U) || !eqh_aligned16(dy) ||
        !eqh_aligned16(add) || !eqh_aligned16(dx) || !eqh_aligned16(workspace))
        return EQH_ERR_ALIGN;
    if (dy && (dy_ld < C || (dy_ld & 3))) return EQH_ERR_ARG;
    if (workspace_bytes < faf_ln_rowdot_bwd_workspace_bytes(R, C, J)) return EQH_ERR_ARG;
    const int blocks = lr_blocks(R);
    float* slab_a = static_cast<float*>(workspace);
    float* slab_b = slab_a + (size_t)blocks * 2 * C;
    return lr_dispatch(C, J, [&](auto nv, auto jj) {
        constexpr int NV = decltype(nv)::value, JJ = decltype(jj)::value;
        hipLaunchKernelGGL((k_ln_rowdot_bwd<NV, JJ>), dim3(blocks), dim3(LR_THREADS), 0, stream, x, gamma, beta, U, dy, dy_ld, dle,
                           add, (int)R, (int)C, eps, dx, slab_a, slab_b);
        EQH_CHECK_LAUNCH();
        // dU is an intermediate's gradient (the folded weights): reduced at once; d gamma / d beta may join the deferred batch
        if (int e = eqh_reduce_slabs_async(slab_b, blocks, (int64_t)JJ * C, dU, stream, 0)) return e;
        return eqh_reduce_slabs3_async(slab_a, blocks, 2 * (int64_t)C, dgamma, dbeta, nullptr, C, C, accumulate, stream);
    });
}
