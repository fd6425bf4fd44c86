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
// direction is ONE launch.  The products run on the fp64 matrix cores (v_mfma_f64_16x16x4_f64, operands converted from
// the fp32 LDS tiles): a merged weight is then the CORRECTLY ROUNDED product whatever the summation order -- a merged
// Linear differs from the two it replaces by the rounding of one weight, not by an order-dependent 1e-7 that decides on
// which side of a ReLU kink a pre-activation falls (observed: a 6 % gradient difference on one golden case between two
// fp32 summation orders).  Fixed order, no atomics: bitwise reproducible.
#include "common.h"

namespace {

constexpr int SM_MAXP = 8;
constexpr int SM_T = 32;       // output tile edge (four wavefronts, 16 x 16 each)
constexpr int SM_KC = 256;     // K chunk held in LDS
constexpr int SM_LD = SM_KC + 2;   // row stride (floats): bank = 2 row + k, conflict-free operand reads

typedef double f64x4 __attribute__((ext_vector_type(4)));

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

// dst[r][c] = src[(r0 + r) * rs + (c0 + c) * cs] for r < 32, c < SM_KC (zero outside R x C): the tile of one operand
// with its K index along the LDS row.  float4 along whichever index is contiguous in memory, else scalar.
__device__ __forceinline__ void sm_load(float (*dst)[SM_LD], const float* __restrict__ src, int64_t rs, int64_t cs, int r0,
                                        int c0, int R, int C) {
    const bool al = ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
    // whole tiles (the merged [C x C] weights): all of a thread's eight float4 loads in flight before the first LDS store --
    // one at a time the loop below is a chain of eight L2 round trips per operand (13 -> 8 us per launch at the BASELINE batch)
    constexpr int PER = SM_T * (SM_KC / 4) / 256;
    if (cs == 1 && al && (rs & 3) == 0 && (c0 & 3) == 0 && r0 + SM_T <= R && c0 + SM_KC <= C) {
        float4 v[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int e = threadIdx.x + 256 * u, r = e / (SM_KC / 4), c = (e % (SM_KC / 4)) * 4;
            v[u] = *reinterpret_cast<const float4*>(src + (int64_t)(r0 + r) * rs + (c0 + c));
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int e = threadIdx.x + 256 * u, r = e / (SM_KC / 4), c = (e % (SM_KC / 4)) * 4;
            dst[r][c] = v[u].x; dst[r][c + 1] = v[u].y; dst[r][c + 2] = v[u].z; dst[r][c + 3] = v[u].w;
        }
        return;
    }
    if (rs == 1 && al && (cs & 3) == 0 && (r0 & 3) == 0 && r0 + SM_T <= R && c0 + SM_KC <= C) {
        float4 v[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int e = threadIdx.x + 256 * u, r = (e % (SM_T / 4)) * 4, c = e / (SM_T / 4);
            v[u] = *reinterpret_cast<const float4*>(src + (int64_t)(c0 + c) * cs + (r0 + r));
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int e = threadIdx.x + 256 * u, r = (e % (SM_T / 4)) * 4, c = e / (SM_T / 4);
            dst[r][c] = v[u].x; dst[r + 1][c] = v[u].y; dst[r + 2][c] = v[u].z; dst[r + 3][c] = v[u].w;
        }
        return;
    }
    if (cs == 1 && al && (rs & 3) == 0 && (c0 & 3) == 0) {              // contiguous along k
        for (int e = threadIdx.x; e < SM_T * (SM_KC / 4); e += 256) {
            const int r = e / (SM_KC / 4), c = (e % (SM_KC / 4)) * 4;
            float4 v = f4_zero();
            if (r0 + r < R && c0 + c + 3 < C) v = *reinterpret_cast<const float4*>(src + (int64_t)(r0 + r) * rs + (c0 + c));
            else if (r0 + r < R) {
                const float* p = src + (int64_t)(r0 + r) * rs + (c0 + c);
                v.x = c0 + c < C ? p[0] : 0.f; v.y = c0 + c + 1 < C ? p[1] : 0.f; v.z = c0 + c + 2 < C ? p[2] : 0.f;
            }
            dst[r][c] = v.x; dst[r][c + 1] = v.y; dst[r][c + 2] = v.z; dst[r][c + 3] = v.w;
        }
    } else if (rs == 1 && al && (cs & 3) == 0 && (r0 & 3) == 0) {       // contiguous along the tile's row index
        for (int e = threadIdx.x; e < (SM_T / 4) * SM_KC; e += 256) {
            const int r = (e % (SM_T / 4)) * 4, c = e / (SM_T / 4);
            float4 v = f4_zero();
            if (c0 + c < C && r0 + r + 3 < R) v = *reinterpret_cast<const float4*>(src + (int64_t)(c0 + c) * cs + (r0 + r));
            else if (c0 + c < C) {
                const float* p = src + (int64_t)(c0 + c) * cs + (r0 + r);
                v.x = r0 + r < R ? p[0] : 0.f; v.y = r0 + r + 1 < R ? p[1] : 0.f; v.z = r0 + r + 2 < R ? p[2] : 0.f;
            }
            dst[r][c] = v.x; dst[r + 1][c] = v.y; dst[r + 2][c] = v.z; dst[r + 3][c] = v.w;
        }
    } else {
        for (int e = threadIdx.x; e < SM_T * SM_KC; e += 256) {
            const int r = e / SM_KC, c = e % SM_KC;
            dst[r][c] = (r0 + r < R && c0 + c < C) ? src[(int64_t)(r0 + r) * rs + (int64_t)(c0 + c) * cs] : 0.f;
        }
    }
}

__global__ void __launch_bounds__(256)
k_small_mm(const SmBatch batch) {
    __shared__ float s_a[SM_T][SM_LD], s_b[SM_T][SM_LD], s_x[SM_KC];     // A tile [m][k], B tile TRANSPOSED [n][k]
    int pi = 0;
#pragma unroll
    for (int i = 1; i < SM_MAXP; ++i)
        if (i < batch.n && (int)blockIdx.x >= batch.p[i].first_block) pi = i;
    const SmProb P = batch.p[pi];
    const int tile = (int)blockIdx.x - P.first_block;
    const int m0 = (tile / P.tiles_n) * SM_T, n0 = (tile % P.tiles_n) * SM_T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int fi = lane & 15, fq = lane >> 4;
    const bool rider = P.y != nullptr && n0 == 0;                    // the column-0 tiles also carry y = A x
    f64x4 acc = {0., 0., 0., 0.};
    double yv = 0.;
    for (int k0 = 0; k0 < P.k; k0 += SM_KC) {
        sm_load(s_a, P.a, P.a_rs, P.a_cs, m0, k0, P.m, P.k);
        sm_load(s_b, P.b, P.b_cs, P.b_rs, n0, k0, P.n, P.k);           // (row index of the tile = n: strides swapped)
        if (rider)
            for (int e = threadIdx.x; e < SM_KC; e += 256) s_x[e] = (k0 + e < P.k) ? P.x[k0 + e] : 0.f;
        __syncthreads();
        const int kc = (P.k - k0 < SM_KC) ? ((P.k - k0 + 3) & ~3) : SM_KC;
        const float* pa = &s_a[wi * 16 + fi][fq];
        const float* pb = &s_b[wj * 16 + fi][fq];
#pragma unroll 8
        for (int kk = 0; kk < kc; kk += 4)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)pa[kk], (double)pb[kk], acc, 0, 0, 0);
        if (rider) {       // rows 8 wave .. + 7, eight k slices of 32 each, then a butterfly over the slices
            const int r = lane >> 3, sl = lane & 7;
            const float* ra = &s_a[wave * 8 + r][sl * (SM_KC / 8)];
            const float* rx = &s_x[sl * (SM_KC / 8)];
            double t = 0.;
#pragma unroll 8
            for (int kk = 0; kk < SM_KC / 8; ++kk) t = fma((double)ra[kk], (double)rx[kk], t);
            t += __shfl_xor(t, 1, 64);
            t += __shfl_xor(t, 2, 64);
            t += __shfl_xor(t, 4, 64);
            yv += t;
        }
        __syncthreads();
    }
    // D of v_mfma_f64_16x16x4_f64: column = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int gm = m0 + wi * 16 + fq + 4 * g, gn = n0 + wj * 16 + fi;
        if (gm < P.m && gn < P.n) {
            double o = (double)P.alpha * acc[g];
            if (P.u) o = fma((double)P.u[gm], (double)P.v[gn], o);
            float* dst = P.c + (int64_t)gm * P.ldc + gn;
            *dst = (float)(P.acc_c ? (double)*dst + o : o);
        }
    }
    if (rider && (lane & 7) == 0) {
        const int gm = m0 + wave * 8 + (lane >> 3);
        if (gm < P.m) {
            const double o = yv + (P.z ? (double)P.z[gm] : 0.);
            P.y[gm] = (float)(P.acc_y ? (double)P.y[gm] + o : o);
        }
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
