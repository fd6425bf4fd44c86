// Batched SMALL matrix products with arbitrary operand strides, rank-1 addends and matrix-vector riders -- the
// weight-level algebra of layers.MHNNSConv._prepare_merged (conv.py:172-181: two Linears with only a linear map between
// them are one Linear whose weight is the product of the two):
//
//     C[m, n]  (+)= alpha * sum_k A(m, k) B(k, n)  (+ u[m] v[n])                      A(m, k) = a[m * a_rs + k * a_cs]
//     y[m]     (+)= sum_k A(m, k) x[k]  (+ z[m])                                      B(k, n) = b[k * b_rs + n * b_cs]
//     w[m]      += u[m]
//
// Forward, per layer: Wc = W2[:, C:2C] W1b, bc = W2[:, C:2C] b1 + b2 and Wd = W3a W2c -- two [256 x 256] x [256 x 256]
// products, a matrix-vector product and a vector add that were five launches at the in-graph launch floor (25 us); backward:
// dA += dWc B^T + dbc (x) bb, dB += A^T dWc, dbb += A^T dbc, dbo += dbc for both, seven launches (39 us).  Here each
// direction is ONE launch.  67-134 MFLOP: plain FMAs (fp64 accumulators) from LDS tiles (32 x 32 outputs per 256-thread workgroup,
// K walked 32 at a time), fixed summation order, no atomics -- bitwise reproducible.
#include "common.h"

namespace {

constexpr int SM_MAXP = 8;
constexpr int SM_T = 32;

struct SmProb {
    const float* a; int64_t a_rs, a_cs;
    const float* b; int64_t b_rs, b_cs;
    float* c; int64_t ldc;
    const float* u; const float* v;        // rank-1 addend (both or neither)
    const float* x; const float* z;        // y = A x (+ z)
    float* y;
    float* w;                              // w += u
    int m, n, k;
    float alpha;
    int acc_c, acc_y;
    int first_block, tiles_n;
};

struct SmBatch {
    SmProb p[SM_MAXP];
    int n;
};

__global__ void __launch_bounds__(256)
k_small_mm(const SmBatch batch) {
    __shared__ float s_a[SM_T][SM_T + 1], s_b[SM_T][SM_T + 1], s_x[SM_T];
    int pi = 0;
#pragma unroll
    for (int i = 1; i < SM_MAXP; ++i)
        if (i < batch.n && (int)blockIdx.x >= batch.p[i].first_block) pi = i;
    const SmProb P = batch.p[pi];
    const int tile = (int)blockIdx.x - P.first_block;
    const int m0 = (tile / P.tiles_n) * SM_T, n0 = (tile % P.tiles_n) * SM_T;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;         // outputs (2 ty + {0, 1}, 2 tx + {0, 1})
    const bool rider = P.y != nullptr && n0 == 0;                    // the column-0 tiles also carry y = A x
    // fp64 accumulation: the merged weight is the correctly rounded product, whatever the summation order (a merged Linear
    // then differs from the two it replaces by the rounding of ONE weight, not by an order-dependent 1e-7)
    double c00 = 0., c01 = 0., c10 = 0., c11 = 0., yv = 0.;
    for (int k0 = 0; k0 < P.k; k0 += SM_T) {
        for (int e = threadIdx.x; e < SM_T * SM_T; e += 256) {
            const int r = e >> 5, q = e & 31;                        // A tile: (row r, k q);  B tile: (k r, col q)
            const int gm = m0 + r, gk = k0 + q;
            s_a[r][q] = (gm < P.m && gk < P.k) ? P.a[(int64_t)gm * P.a_rs + (int64_t)gk * P.a_cs] : 0.f;
            const int gk2 = k0 + r, gn = n0 + q;
            s_b[r][q] = (gk2 < P.k && gn < P.n) ? P.b[(int64_t)gk2 * P.b_rs + (int64_t)gn * P.b_cs] : 0.f;
        }
        if (rider && threadIdx.x < SM_T) s_x[threadIdx.x] = (k0 + (int)threadIdx.x < P.k) ? P.x[k0 + threadIdx.x] : 0.f;
        __syncthreads();
#pragma unroll 8
        for (int kk = 0; kk < SM_T; ++kk) {
            const double a0 = s_a[2 * ty][kk], a1 = s_a[2 * ty + 1][kk];
            const double b0 = s_b[kk][2 * tx], b1 = s_b[kk][2 * tx + 1];
            c00 = fma(a0, b0, c00); c01 = fma(a0, b1, c01);
            c10 = fma(a1, b0, c10); c11 = fma(a1, b1, c11);
        }
        if (rider && threadIdx.x < SM_T) {
#pragma unroll 8
            for (int kk = 0; kk < SM_T; ++kk) yv = fma((double)s_a[threadIdx.x][kk], (double)s_x[kk], yv);
        }
        __syncthreads();
    }
    const double cc[2][2] = {{c00, c01}, {c10, c11}};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int gm = m0 + 2 * ty + i, gn = n0 + 2 * tx + j;
            if (gm < P.m && gn < P.n) {
                double o = (double)P.alpha * cc[i][j];
                if (P.u) o = fma((double)P.u[gm], (double)P.v[gn], o);
                float* dst = P.c + (int64_t)gm * P.ldc + gn;
                *dst = (float)(P.acc_c ? (double)*dst + o : o);
            }
        }
    if (rider && threadIdx.x < SM_T && m0 + (int)threadIdx.x < P.m) {
        const int gm = m0 + threadIdx.x;
        const double o = yv + (P.z ? (double)P.z[gm] : 0.);
        P.y[gm] = (float)(P.acc_y ? (double)P.y[gm] + o : o);
    }
    if (P.w && n0 == 0 && threadIdx.x < SM_T && m0 + (int)threadIdx.x < P.m) P.w[m0 + threadIdx.x] += P.u[m0 + threadIdx.x];
}

}  // namespace

extern "C" int hg_small_mm_batch(int32_t n_problems, const HgSmallMM* pr, void* stream_) {
    if (n_problems <= 0 || n_problems > SM_MAXP || !pr) return EQH_ERR_ARG;
    SmBatch b;
    b.n = n_problems;
    int first = 0;
    for (int i = 0; i < n_problems; ++i) {
        const HgSmallMM& q = pr[i];
        if (q.m <= 0 || q.n <= 0 || q.k <= 0 || !q.a || !q.b || !q.c) return EQH_ERR_ARG;
        if ((q.u == nullptr) != (q.v == nullptr) || (q.y && !q.x) || (q.w && !q.u)) return EQH_ERR_ARG;
        if (q.m > 65536 || q.n > 65536) return EQH_ERR_RANGE;         // (a SMALL product: weights)
        SmProb& p = b.p[i];
        p.a = q.a; p.a_rs = q.a_rs; p.a_cs = q.a_cs;
        p.b = q.b; p.b_rs = q.b_rs; p.b_cs = q.b_cs;
        p.c = q.c; p.ldc = q.ldc; p.u = q.u; p.v = q.v; p.x = q.x; p.z = q.z; p.y = q.y; p.w = q.w;
        p.m = q.m; p.n = q.n; p.k = q.k; p.alpha = q.alpha; p.acc_c = q.accumulate_c; p.acc_y = q.accumulate_y;
        p.tiles_n = (q.n + SM_T - 1) / SM_T;
        p.first_block = first;
        first += ((q.m + SM_T - 1) / SM_T) * p.tiles_n;
    }
    for (int i = n_problems; i < SM_MAXP; ++i) b.p[i] = b.p[0], b.p[i].first_block = 0x7fffffff;
    hipLaunchKernelGGL(k_small_mm, dim3(first), dim3(256), 0, static_cast<hipStream_t>(stream_), b);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}
