// Training-mode BatchNorm1d over node rows whose batch statistics count the REAL rows only (mhnn.py:182,206 under the
// padded static-shape batches of the graphed trainer): y = (x - mean) * rstd * gamma + beta with
//   mean[c] = sum_i m_i x[i, c] / n,   var[c] = sum_i m_i (x[i, c] - mean[c])^2 / n,   n = sum_i m_i   (m: row mask, NULL = 1)
// and the running-buffer updates of nn.BatchNorm1d (running_var takes the unbiased variance).  As torch ops this was
// ~10 launches forward and ~15 backward per layer, each at the in-graph launch floor.  Here one launch each way: a
// workgroup owns FOUR columns and walks all rows three times (mean, variance, output; the [R, C] matrix is a few MB and
// stays in L2), so the column statistics never leave the workgroup, nothing is atomic and the results are bitwise
// reproducible.  Backward: dgamma = sum dy xhat, dbeta = sum dy, dx = gamma rstd (dy - m dbeta / n - m xhat dgamma / n).
// Degenerate masks: nn.BatchNorm1d refuses a training batch of fewer than two rows ("Expected more than 1 value per
// channel").  A kernel inside a replayed graph cannot raise, so: n is clamped to >= 1 in both directions (a mask that
// selects no row gives mean 0, var 0 and finite outputs instead of NaN), and the running buffers and the batch counter are
// left untouched when n < 2 (no unbiased variance exists; nothing is written that would poison later evaluations).
#include "common.h"

namespace {

constexpr int BN_THREADS = 256;

__device__ __forceinline__ float4 bn_block_sum(float4 v, float4* s_red) {
    // butterfly inside the wavefront, then the four wavefronts through LDS; every thread gets the total
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        v.x += __shfl_xor(v.x, o, 64); v.y += __shfl_xor(v.y, o, 64);
        v.z += __shfl_xor(v.z, o, 64); v.w += __shfl_xor(v.w, o, 64);
    }
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[wave] = v;
    __syncthreads();
    float4 t = s_red[0];
#pragma unroll
    for (int w = 1; w < BN_THREADS / 64; ++w) f4_add(t, s_red[w]);
    return t;
}

__global__ void __launch_bounds__(BN_THREADS)
k_bn_rows_fwd(const float* __restrict__ x, const float* __restrict__ mask, const float* __restrict__ gamma,
              const float* __restrict__ beta, float* __restrict__ run_mean, float* __restrict__ run_var,
              int64_t* __restrict__ n_tracked, float momentum, float eps, int64_t R, int C, float* __restrict__ y,
              float* __restrict__ save_mean, float* __restrict__ save_rstd) {
    __shared__ float4 s_red[BN_THREADS / 64];
    const int c = blockIdx.x * 4;
    float4 s = f4_zero();
    float cnt = 0.f;
    for (int64_t i = threadIdx.x; i < R; i += BN_THREADS) {
        const float m = mask ? mask[i] : 1.0f;
        const float4 v = *reinterpret_cast<const float4*>(x + i * C + c);
        f4_fma(s, v, m);
        cnt += m;
    }
    const float4 tot = bn_block_sum(s, s_red);
    const float n_rows = bn_block_sum(make_float4(cnt, 0.f, 0.f, 0.f), s_red).x;
    const float n = fmaxf(n_rows, 1.0f);
    const float inv_n = 1.0f / n;
    const float4 mean = make_float4(tot.x * inv_n, tot.y * inv_n, tot.z * inv_n, tot.w * inv_n);
    float4 ss = f4_zero();
    for (int64_t i = threadIdx.x; i < R; i += BN_THREADS) {
        const float m = mask ? mask[i] : 1.0f;
        const float4 v = *reinterpret_cast<const float4*>(x + i * C + c);
        const float dx = v.x - mean.x, dy = v.y - mean.y, dz = v.z - mean.z, dw = v.w - mean.w;
        ss.x = fmaf(m * dx, dx, ss.x); ss.y = fmaf(m * dy, dy, ss.y); ss.z = fmaf(m * dz, dz, ss.z); ss.w = fmaf(m * dw, dw, ss.w);
    }
    const float4 sq = bn_block_sum(ss, s_red);
    const float4 var = make_float4(sq.x * inv_n, sq.y * inv_n, sq.z * inv_n, sq.w * inv_n);
    const float4 rstd = make_float4(1.0f / sqrtf(var.x + eps), 1.0f / sqrtf(var.y + eps), 1.0f / sqrtf(var.z + eps),
                                    1.0f / sqrtf(var.w + eps));
    const float4 g = *reinterpret_cast<const float4*>(gamma + c), b = *reinterpret_cast<const float4*>(beta + c);
    const float4 sc = make_float4(rstd.x * g.x, rstd.y * g.y, rstd.z * g.z, rstd.w * g.w);
    for (int64_t i = threadIdx.x; i < R; i += BN_THREADS) {
        const float4 v = *reinterpret_cast<const float4*>(x + i * C + c);
        *reinterpret_cast<float4*>(y + i * C + c) = make_float4(fmaf(v.x - mean.x, sc.x, b.x), fmaf(v.y - mean.y, sc.y, b.y),
                                                                 fmaf(v.z - mean.z, sc.z, b.z), fmaf(v.w - mean.w, sc.w, b.w));
    }
    if (threadIdx.x == 0) {
        *reinterpret_cast<float4*>(save_mean + c) = mean;
        *reinterpret_cast<float4*>(save_rstd + c) = rstd;
        if (run_mean && n_rows >= 2.0f) {
            const float unb = n / (n - 1.0f);                     // nn.BatchNorm1d stores the unbiased variance
            float4 rm = *reinterpret_cast<const float4*>(run_mean + c), rv = *reinterpret_cast<const float4*>(run_var + c);
            rm.x += momentum * (mean.x - rm.x); rm.y += momentum * (mean.y - rm.y);
            rm.z += momentum * (mean.z - rm.z); rm.w += momentum * (mean.w - rm.w);
            rv.x += momentum * (var.x * unb - rv.x); rv.y += momentum * (var.y * unb - rv.y);
            rv.z += momentum * (var.z * unb - rv.z); rv.w += momentum * (var.w * unb - rv.w);
            *reinterpret_cast<float4*>(run_mean + c) = rm;
            *reinterpret_cast<float4*>(run_var + c) = rv;
            if (blockIdx.x == 0 && n_tracked) *n_tracked += 1;
        }
    }
}

__global__ void __launch_bounds__(BN_THREADS)
k_bn_rows_bwd(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mask,
              const float* __restrict__ gamma, const float* __restrict__ save_mean, const float* __restrict__ save_rstd,
              int64_t R, int C, float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float4 s_red[BN_THREADS / 64];
    const int c = blockIdx.x * 4;
    const float4 mean = *reinterpret_cast<const float4*>(save_mean + c), rstd = *reinterpret_cast<const float4*>(save_rstd + c);
    float4 s1 = f4_zero(), s2 = f4_zero();
    float cnt = 0.f;
    for (int64_t i = threadIdx.x; i < R; i += BN_THREADS) {
        const float4 v = *reinterpret_cast<const float4*>(x + i * C + c), g = *reinterpret_cast<const float4*>(dy + i * C + c);
        f4_add(s1, g);
        s2.x = fmaf(g.x, (v.x - mean.x) * rstd.x, s2.x); s2.y = fmaf(g.y, (v.y - mean.y) * rstd.y, s2.y);
        s2.z = fmaf(g.z, (v.z - mean.z) * rstd.z, s2.z); s2.w = fmaf(g.w, (v.w - mean.w) * rstd.w, s2.w);
        cnt += mask ? mask[i] : 1.0f;
    }
    const float4 db = bn_block_sum(s1, s_red), dg = bn_block_sum(s2, s_red);
    const float inv_n = 1.0f / fmaxf(bn_block_sum(make_float4(cnt, 0.f, 0.f, 0.f), s_red).x, 1.0f);
    const float4 g4 = *reinterpret_cast<const float4*>(gamma + c);
    const float4 sc = make_float4(rstd.x * g4.x, rstd.y * g4.y, rstd.z * g4.z, rstd.w * g4.w);
    for (int64_t i = threadIdx.x; i < R; i += BN_THREADS) {
        const float m = (mask ? mask[i] : 1.0f) * inv_n;
        const float4 v = *reinterpret_cast<const float4*>(x + i * C + c), g = *reinterpret_cast<const float4*>(dy + i * C + c);
        float4 o;
        o.x = sc.x * (g.x - m * db.x - m * ((v.x - mean.x) * rstd.x) * dg.x);
        o.y = sc.y * (g.y - m * db.y - m * ((v.y - mean.y) * rstd.y) * dg.y);
        o.z = sc.z * (g.z - m * db.z - m * ((v.z - mean.z) * rstd.z) * dg.z);
        o.w = sc.w * (g.w - m * db.w - m * ((v.w - mean.w) * rstd.w) * dg.w);
        *reinterpret_cast<float4*>(dx + i * C + c) = o;
    }
    if (threadIdx.x == 0) {
        *reinterpret_cast<float4*>(dgamma + c) = dg;
        *reinterpret_cast<float4*>(dbeta + c) = db;
    }
}

int bn_check(int64_t R, int32_t C) {
    if (R < 0 || C <= 0) return EQH_ERR_ARG;
    if (C & 3) return EQH_ERR_ALIGN;
    return EQH_OK;
}

}  // namespace

extern "C" int hg_batch_norm_rows_fwd(const float* x, const float* row_mask, const float* gamma, const float* beta,
                                      float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                                      float eps, int64_t R, int32_t C, float* y, float* save_mean, float* save_rstd,
                                      void* stream_) {
    int rc = bn_check(R, C);
    if (rc) return rc;
    if (R == 0) return EQH_OK;
    if (!x || !gamma || !beta || !y || !save_mean || !save_rstd || (running_mean && !running_var)) return EQH_ERR_ARG;
    if (!eqh_aligned16(x) || !eqh_aligned16(y) || !eqh_aligned16(gamma) || !eqh_aligned16(beta) || !eqh_aligned16(save_mean) ||
        !eqh_aligned16(save_rstd) || !eqh_aligned16(running_mean) || !eqh_aligned16(running_var))
        return EQH_ERR_ALIGN;
    hipLaunchKernelGGL(k_bn_rows_fwd, dim3(C / 4), dim3(BN_THREADS), 0, static_cast<hipStream_t>(stream_), x, row_mask, gamma, beta,
                       running_mean, running_var, num_batches_tracked, momentum, eps, R, (int)C, y, save_mean, save_rstd);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}

extern "C" int hg_batch_norm_rows_bwd(const float* x, const float* dy, const float* row_mask, const float* gamma,
                                      const float* save_mean, const float* save_rstd, int64_t R, int32_t C, float* dx,
                                      float* dgamma, float* dbeta, void* stream_) {
    int rc = bn_check(R, C);
    if (rc) return rc;
    if (!dgamma || !dbeta) return EQH_ERR_ARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (R == 0) {
        if (eqh_zero_async(dgamma, C, stream)) return EQH_ERR_LAUNCH;
        return eqh_zero_async(dbeta, C, stream);
    }
    if (!x || !dy || !gamma || !save_mean || !save_rstd || !dx) return EQH_ERR_ARG;
    if (!eqh_aligned16(x) || !eqh_aligned16(dy) || !eqh_aligned16(dx) || !eqh_aligned16(gamma) || !eqh_aligned16(save_mean) ||
        !eqh_aligned16(save_rstd) || !eqh_aligned16(dgamma) || !eqh_aligned16(dbeta))
        return EQH_ERR_ALIGN;
    hipLaunchKernelGGL(k_bn_rows_bwd, dim3(C / 4), dim3(BN_THREADS), 0, stream, x, dy, row_mask, gamma, save_mean, save_rstd, R,
                       (int)C, dx, dgamma, dbeta);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}
