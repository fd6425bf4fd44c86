// Training-mode BatchNorm1d over node rows whose batch statistics count the REAL rows only (mhnn.py:182,206 under the
// padded static-shape batches of the graphed trainer): y = (x - mean) * rstd * gamma + beta with
//   mean[c] = sum_i m_i x[i, c] / n,   var[c] = sum_i m_i (x[i, c] - mean[c])^2 / n,   n = sum_i m_i   (m: row mask, NULL = 1)
// and the running-buffer updates of nn.BatchNorm1d (running_var takes the unbiased variance).  As torch ops this was
// ~10 launches forward and ~15 backward per layer, each at the in-graph launch floor.
//
// Two launches each way (round 5; the first version was ONE launch whose workgroups owned four columns each and walked all
// rows three times -- 64 workgroups reading 16 bytes of every 1-KB row: 43 us for a 4.8 MB matrix):
//   1. k_bn_partial: a workgroup sums BN_ROWS rows x 64 columns (whole 256-byte row pieces per 16 lanes) into float64
//      partials  [chunk][2][C]:  sum m x and sum m x^2  (forward),  sum dy and sum dy xhat  (backward).  The products of two
//      floats are exact in float64 and at most 2^20 of them are summed, so E[x^2] - mean^2 carries no cancellation error that
//      a float could see;
//   2. k_bn_apply: every workgroup first sums the partials of ITS columns in chunk order (a few KB from L2; fixed order:
//      bitwise reproducible, nothing atomic), then writes its rows.  Workgroup row 0 also writes the saved statistics, the
//      running buffers and the parameter gradients.
// Backward: dgamma = sum dy xhat, dbeta = sum dy, dx = gamma rstd (dy - m dbeta / n - m xhat dgamma / n).
// Degenerate masks: nn.BatchNorm1d refuses a training batch of fewer than two rows ("Expected more than 1 value per
// channel").  A kernel inside a replayed graph cannot raise, so: n is clamped to >= 1 in both directions (a mask that
// selects no row gives mean 0, var 0 and finite outputs instead of NaN), and the running buffers and the batch counter are
// left untouched when n < 2 (no unbiased variance exists; nothing is written that would poison later evaluations).
#include "common.h"

namespace {

constexpr int BN_THREADS = 256;
constexpr int BN_ROWS = 128;      // rows per partial (stats kernel): 16 row slots x 8 rows
constexpr int BN_COLS = 64;       // columns per stats workgroup: 16 lanes x float4
constexpr int BN_APPLY_ROWS = 32; // rows per apply workgroup (all columns of a 256-column group)

struct BnPartials {
    double* sums;   // [chunks][2][C]
    double* cnt;    // [chunks]
    int chunks;
};

// partial column sums of one (row chunk, 64-column group): thread = (row slot s = tid / 16, lane l = tid % 16 -> columns 4 l ..)
template <bool BWD>
__global__ void __launch_bounds__(BN_THREADS)
k_bn_partial(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mask,
             const float* __restrict__ save_mean, const float* __restrict__ save_rstd, int64_t R, int C, BnPartials P,
             const float* __restrict__ gamma, const float* __restrict__ beta, int relu) {
    __shared__ double s_a[16][BN_COLS], s_b[16][BN_COLS], s_n[16];
    const int l = threadIdx.x & 15, s = threadIdx.x >> 4;
    const int c = blockIdx.y * BN_COLS + 4 * l;
    const bool act = c < C;
    const int64_t r0 = (int64_t)blockIdx.x * BN_ROWS;
    double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0}, n = 0;
    float4 mu = f4_zero(), rs = f4_zero(), ga = f4_zero(), be = f4_zero();
    if (BWD && act) {
        mu = *reinterpret_cast<const float4*>(save_mean + c);
        rs = *reinterpret_cast<const float4*>(save_rstd + c);
        if (relu) { ga = *reinterpret_cast<const float4*>(gamma + c); be = *reinterpret_cast<const float4*>(beta + c); }
    }
#pragma unroll 4
    for (int k = 0; k < BN_ROWS / 16; ++k) {
        const int64_t i = r0 + s + 16 * k;
        if (i >= R) break;
        const float m = mask ? mask[i] : 1.0f;
        n += m;
        if (!act) continue;
        const float4 v = *reinterpret_cast<const float4*>(x + i * C + c);
        if (!BWD) {
            const double m_ = m;
            a[0] += m_ * v.x; a[1] += m_ * v.y; a[2] += m_ * v.z; a[3] += m_ * v.w;
            b[0] += m_ * ((double)v.x * v.x); b[1] += m_ * ((double)v.y * v.y);
            b[2] += m_ * ((double)v.z * v.z); b[3] += m_ * ((double)v.w * v.w);
        } else {
            float4 g = *reinterpret_cast<const float4*>(dy + i * C + c);
            const float xh0 = (v.x - mu.x) * rs.x, xh1 = (v.y - mu.y) * rs.y, xh2 = (v.z - mu.z) * rs.z, xh3 = (v.w - mu.w) * rs.w;
            if (relu) {         // the ReLU behind the normalisation (mhnn.py:208-214): its gate recomputed from x
                // (the forward's own expression, fmaf(x - mean, rstd * gamma, beta): the same bits decide both ways)
                if (!(fmaf(v.x - mu.x, rs.x * ga.x, be.x) > 0.f)) g.x = 0.f;
                if (!(fmaf(v.y - mu.y, rs.y * ga.y, be.y) > 0.f)) g.y = 0.f;
                if (!(fmaf(v.z - mu.z, rs.z * ga.z, be.z) > 0.f)) g.z = 0.f;
                if (!(fmaf(v.w - mu.w, rs.w * ga.w, be.w) > 0.f)) g.w = 0.f;
            }
            a[0] += g.x; a[1] += g.y; a[2] += g.z; a[3] += g.w;
            b[0] += (double)g.x * xh0; b[1] += (double)g.y * xh1; b[2] += (double)g.z * xh2; b[3] += (double)g.w * xh3;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { s_a[s][4 * l + j] = a[j]; s_b[s][4 * l + j] = b[j]; }
    if (l == 0) s_n[s] = n;
    __syncthreads();
    if (threadIdx.x < BN_COLS) {            // the 16 row slots in order
        const int cc = blockIdx.y * BN_COLS + threadIdx.x;
        if (cc < C) {
            double ta = 0, tb = 0;
#pragma unroll
            for (int q = 0; q < 16; ++q) { ta += s_a[q][threadIdx.x]; tb += s_b[q][threadIdx.x]; }
            P.sums[((int64_t)blockIdx.x * 2 + 0) * C + cc] = ta;
            P.sums[((int64_t)blockIdx.x * 2 + 1) * C + cc] = tb;
        }
    }
    if (threadIdx.x == 0 && blockIdx.y == 0) {
        double tn = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) tn += s_n[q];
        P.cnt[blockIdx.x] = tn;
    }
}

// the statistics of this workgroup's 256-column group from the partials (thread = column), then its rows
__global__ void __launch_bounds__(BN_THREADS)
k_bn_apply_fwd(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
               float* __restrict__ run_mean, float* __restrict__ run_var, int64_t* __restrict__ n_tracked, float momentum,
               float eps, int64_t R, int C, BnPartials P, float* __restrict__ y, float* __restrict__ save_mean,
               float* __restrict__ save_rstd, int relu) {
    __shared__ float s_mean[256], s_scale[256], s_shift[256];
    const int cg = blockIdx.y * 256;
    {
        const int c = cg + threadIdx.x;
        if (c < C) {
            double sa = 0, sb = 0, nr = 0;
            for (int k = 0; k < P.chunks; ++k) {
                sa += P.sums[((int64_t)k * 2 + 0) * C + c];
                sb += P.sums[((int64_t)k * 2 + 1) * C + c];
                nr += P.cnt[k];
            }
            const double n = nr < 1.0 ? 1.0 : nr;
            const double mean = sa / n;
            double var = sb / n - mean * mean;
            if (var < 0.0) var = 0.0;
            const float meanf = (float)mean, varf = (float)var;
            const float rstd = 1.0f / sqrtf(varf + eps);
            const float sc = rstd * gamma[c];
            s_mean[threadIdx.x] = meanf;
            s_scale[threadIdx.x] = sc;
            s_shift[threadIdx.x] = beta[c];
            if (blockIdx.x == 0) {
                save_mean[c] = meanf;
                save_rstd[c] = rstd;
                if (run_mean && nr >= 2.0) {
                    const float unb = (float)(n / (n - 1.0));                 // nn.BatchNorm1d stores the unbiased variance
                    run_mean[c] += momentum * (meanf - run_mean[c]);
                    run_var[c] += momentum * (varf * unb - run_var[c]);
                    if (c == 0 && n_tracked) *n_tracked += 1;
                }
            }
        }
    }
    __syncthreads();
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = cg + 4 * l;
    if (c >= C) return;
    const float4 mu = *reinterpret_cast<const float4*>(s_mean + 4 * l), sc = *reinterpret_cast<const float4*>(s_scale + 4 * l),
                 sh = *reinterpret_cast<const float4*>(s_shift + 4 * l);
    const int64_t r0 = (int64_t)blockIdx.x * BN_APPLY_ROWS;
#pragma unroll 4
    for (int k = w; k < BN_APPLY_ROWS; k += BN_THREADS / 64) {
        const int64_t i = r0 + k;
        if (i >= R) break;
        const float4 v = *reinterpret_cast<const float4*>(x + i * C + c);
        float4 o = make_float4(fmaf(v.x - mu.x, sc.x, sh.x), fmaf(v.y - mu.y, sc.y, sh.y), fmaf(v.z - mu.z, sc.z, sh.z),
                               fmaf(v.w - mu.w, sc.w, sh.w));
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        *reinterpret_cast<float4*>(y + i * C + c) = o;
    }
}

__global__ void __launch_bounds__(BN_THREADS)
k_bn_apply_bwd(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mask,
               const float* __restrict__ gamma, const float* __restrict__ save_mean, const float* __restrict__ save_rstd,
               int64_t R, int C, BnPartials P, float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta,
               const float* __restrict__ beta, int relu) {
    __shared__ float s_db[256], s_dg[256], s_mean[256], s_rstd[256], s_scale[256], s_beta[256];
    __shared__ float s_inv_n;
    const int cg = blockIdx.y * 256;
    {
        const int c = cg + threadIdx.x;
        if (c < C) {
            double sa = 0, sb = 0;
            for (int k = 0; k < P.chunks; ++k) {
                sa += P.sums[((int64_t)k * 2 + 0) * C + c];
                sb += P.sums[((int64_t)k * 2 + 1) * C + c];
            }
            const float db = (float)sa, dg = (float)sb, rstd = save_rstd[c];
            s_db[threadIdx.x] = db;
            s_dg[threadIdx.x] = dg;
            s_mean[threadIdx.x] = save_mean[c];
            s_rstd[threadIdx.x] = rstd;
            s_scale[threadIdx.x] = rstd * gamma[c];
            s_beta[threadIdx.x] = relu ? beta[c] : 0.f;
            if (blockIdx.x == 0) { dgamma[c] = dg; dbeta[c] = db; }
        }
        if (threadIdx.x == 0) {
            double nr = 0;
            for (int k = 0; k < P.chunks; ++k) nr += P.cnt[k];
            s_inv_n = (float)(1.0 / (nr < 1.0 ? 1.0 : nr));
        }
    }
    __syncthreads();
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = cg + 4 * l;
    if (c >= C) return;
    const float4 mu = *reinterpret_cast<const float4*>(s_mean + 4 * l), rs = *reinterpret_cast<const float4*>(s_rstd + 4 * l),
                 sc = *reinterpret_cast<const float4*>(s_scale + 4 * l), db = *reinterpret_cast<const float4*>(s_db + 4 * l),
                 dg = *reinterpret_cast<const float4*>(s_dg + 4 * l), be = *reinterpret_cast<const float4*>(s_beta + 4 * l);
    const float inv_n = s_inv_n;
    const int64_t r0 = (int64_t)blockIdx.x * BN_APPLY_ROWS;
#pragma unroll 4
    for (int k = w; k < BN_APPLY_ROWS; k += BN_THREADS / 64) {
        const int64_t i = r0 + k;
        if (i >= R) break;
        const float m = (mask ? mask[i] : 1.0f) * inv_n;
        const float4 v = *reinterpret_cast<const float4*>(x + i * C + c);
        float4 g = *reinterpret_cast<const float4*>(dy + i * C + c);
        if (relu) {
            if (!(fmaf(v.x - mu.x, sc.x, be.x) > 0.f)) g.x = 0.f;
            if (!(fmaf(v.y - mu.y, sc.y, be.y) > 0.f)) g.y = 0.f;
            if (!(fmaf(v.z - mu.z, sc.z, be.z) > 0.f)) g.z = 0.f;
            if (!(fmaf(v.w - mu.w, sc.w, be.w) > 0.f)) g.w = 0.f;
        }
        float4 o;
        o.x = sc.x * (g.x - m * db.x - m * ((v.x - mu.x) * rs.x) * dg.x);
        o.y = sc.y * (g.y - m * db.y - m * ((v.y - mu.y) * rs.y) * dg.y);
        o.z = sc.z * (g.z - m * db.z - m * ((v.z - mu.z) * rs.z) * dg.z);
        o.w = sc.w * (g.w - m * db.w - m * ((v.w - mu.w) * rs.w) * dg.w);
        *reinterpret_cast<float4*>(dx + i * C + c) = o;
    }
}

int bn_check(int64_t R, int32_t C) {
    if (R < 0 || C <= 0) return EQH_ERR_ARG;
    if (C & 3) return EQH_ERR_ALIGN;
    if (R >= ((int64_t)1 << 31)) return EQH_ERR_RANGE;
    return EQH_OK;
}

inline int bn_chunks(int64_t R) { return (int)((R + BN_ROWS - 1) / BN_ROWS); }

inline BnPartials bn_partials(void* ws, int64_t R, int32_t C) {
    BnPartials p;
    p.chunks = bn_chunks(R);
    p.sums = static_cast<double*>(ws);
    p.cnt = p.sums + (size_t)p.chunks * 2 * C;
    return p;
}

}  // namespace

extern "C" size_t hg_batch_norm_rows_workspace_bytes(int64_t R, int32_t C) {
    if (R <= 0 || C <= 0) return 0;
    return ((size_t)bn_chunks(R) * 2 * (size_t)C + (size_t)bn_chunks(R)) * sizeof(double);
}

extern "C" int hg_batch_norm_rows_fwd(const float* x, const float* row_mask, const float* gamma, const float* beta,
                                      float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                                      float eps, int64_t R, int32_t C, float* y, float* save_mean, float* save_rstd,
                                      int32_t relu, void* workspace, size_t workspace_bytes, void* stream_) {
    int rc = bn_check(R, C);
    if (rc) return rc;
    if (R == 0) return EQH_OK;
    if (!x || !gamma || !beta || !y || !save_mean || !save_rstd || (running_mean && !running_var)) return EQH_ERR_ARG;
    if (!eqh_aligned16(x) || !eqh_aligned16(y)) return EQH_ERR_ALIGN;
    if (!workspace || workspace_bytes < hg_batch_norm_rows_workspace_bytes(R, C) || ((uintptr_t)workspace & 7)) return EQH_ERR_ARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const BnPartials P = bn_partials(workspace, R, C);
    hipLaunchKernelGGL(k_bn_partial<false>, dim3(P.chunks, (C + BN_COLS - 1) / BN_COLS), dim3(BN_THREADS), 0, stream, x,
                       (const float*)nullptr, row_mask, (const float*)nullptr, (const float*)nullptr, R, (int)C, P,
                       (const float*)nullptr, (const float*)nullptr, 0);
    EQH_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_bn_apply_fwd, dim3((unsigned)((R + BN_APPLY_ROWS - 1) / BN_APPLY_ROWS), (C + 255) / 256), dim3(BN_THREADS),
                       0, stream, x, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, R, (int)C, P, y,
                       save_mean, save_rstd, (int)relu);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}

extern "C" int hg_batch_norm_rows_bwd(const float* x, const float* dy, const float* row_mask, const float* gamma,
                                      const float* save_mean, const float* save_rstd, int64_t R, int32_t C, float* dx,
                                      float* dgamma, float* dbeta, const float* beta, int32_t relu, void* workspace,
                                      size_t workspace_bytes, void* stream_) {
    int rc = bn_check(R, C);
    if (rc) return rc;
    if (!dgamma || !dbeta) return EQH_ERR_ARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (R == 0) {
        if (eqh_zero_async(dgamma, C, stream)) return EQH_ERR_LAUNCH;
        return eqh_zero_async(dbeta, C, stream);
    }
    if (!x || !dy || !gamma || !save_mean || !save_rstd || !dx || (relu && !beta)) return EQH_ERR_ARG;
    if (!eqh_aligned16(x) || !eqh_aligned16(dy) || !eqh_aligned16(dx) || !eqh_aligned16(save_mean) || !eqh_aligned16(save_rstd))
        return EQH_ERR_ALIGN;
    if (!workspace || workspace_bytes < hg_batch_norm_rows_workspace_bytes(R, C) || ((uintptr_t)workspace & 7)) return EQH_ERR_ARG;
    const BnPartials P = bn_partials(workspace, R, C);
    hipLaunchKernelGGL(k_bn_partial<true>, dim3(P.chunks, (C + BN_COLS - 1) / BN_COLS), dim3(BN_THREADS), 0, stream, x, dy, row_mask,
                       save_mean, save_rstd, R, (int)C, P, gamma, beta, (int)relu);
    EQH_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_bn_apply_bwd, dim3((unsigned)((R + BN_APPLY_ROWS - 1) / BN_APPLY_ROWS), (C + 255) / 256), dim3(BN_THREADS),
                       0, stream, x, dy, row_mask, gamma, save_mean, save_rstd, R, (int)C, P, dx, dgamma, dbeta, beta, (int)relu);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}
