// FAFormer's small geometric steps, each as one launch (plus one partial-sum pass where the step holds a reduction over
// the whole cloud).  Before this file they were ~390 elementwise / reduction launches of 5 us each per training step of
// the Molecule3D batch (2.3 of 21.5 ms); none of them moves more than a few MB.
//
//   k_moments            per-workgroup partial sums of w_n (1, a_n, a_n a_n^T) over the rows of a [N, 3] array, in
//                        float64 and in a FIXED order (bitwise reproducible): the masked centroid / covariance of the
//                        cloud (fa_former_layer.py:86-113 create_frame on the whole cloud, :552-566 the attention's
//                        geometric context) and the reductions of their backward
//   k_centre_mix_*       x_i' = c g_i + x_i (1 - g_i), g = sigmoid(gate logit), c = masked centroid (:566-571)
//   k_cloud_frame_*      y = (x - c m) V, V = eigenvectors of the masked covariance about c (no gradient through V, :98-99)
//   k_edge_frame_*       the same per atom over its K <= 16 neighbour offsets x_i - x_j, with |x_i - x_j|^2 (:357-372)
//   k_attn_logits_*      logits a_q[i] + a_k[j] + l_e, radius mask, softmax over the K slots, dropout (:483-496)
#include "common.h"
#include "drop_hash.h"
#include "eigh3.h"

namespace {

constexpr int GM_THREADS = 256;
constexpr int GM_MAXP = 128;     // workgroups of a moments pass = rows of its partial table (<= GM_THREADS, sum_partials)
constexpr int GM_NQ = 10;        // weight, weight * a (3), weight * a a^T upper triangle (6)

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// partial[b][q], b = blockIdx.x.  w_n = (m ? (m[n] > 0) : 1) * (s ? (s_sigmoid ? sigmoid(s[n]) : s[n]) : 1)
__global__ void __launch_bounds__(GM_THREADS)
k_moments(const float* __restrict__ a, const float* __restrict__ m, const float* __restrict__ s, int s_sigmoid, int64_t N,
          double* __restrict__ partial) {
    __shared__ double s_w[GM_THREADS / 64][GM_NQ];
    double acc[GM_NQ];
#pragma unroll
    for (int q = 0; q < GM_NQ; ++q) acc[q] = 0.0;
    const int64_t stride = (int64_t)gridDim.x * GM_THREADS;
    for (int64_t n = (int64_t)blockIdx.x * GM_THREADS + threadIdx.x; n < N; n += stride) {
        if (m && !(m[n] > 0.f)) continue;
        double w = 1.0;
        if (s) w = s_sigmoid ? (double)sigmoid_f(s[n]) : (double)s[n];
        const double x = a[n * 3 + 0], y = a[n * 3 + 1], z = a[n * 3 + 2];
        acc[0] += w;
        acc[1] += w * x; acc[2] += w * y; acc[3] += w * z;
        acc[4] += w * x * x; acc[5] += w * x * y; acc[6] += w * x * z;
        acc[7] += w * y * y; acc[8] += w * y * z; acc[9] += w * z * z;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < GM_NQ; ++q) {
        double v = acc[q];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d, 64);
        if (lane == 0) s_w[wave][q] = v;
    }
    __syncthreads();
    if (threadIdx.x < GM_NQ) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < GM_THREADS / 64; ++w) v += s_w[w][threadIdx.x];
        partial[(int64_t)blockIdx.x * GM_NQ + threadIdx.x] = v;
    }
}

// the GM_NQ totals of a partial table (nP <= GM_MAXP <= GM_THREADS rows): thread b holds row b, ten loads in flight per
// thread instead of a chain of nP dependent ones (that chain cost 10 us per launch), then the same fixed-order tree as
// k_moments
__device__ __forceinline__ void sum_partials(const double* __restrict__ partial, int nP, double* s_q) {
    __shared__ double s_w[GM_THREADS / 64][GM_NQ];
    double v[GM_NQ];
#pragma unroll
    for (int q = 0; q < GM_NQ; ++q) v[q] = (int)threadIdx.x < nP ? partial[threadIdx.x * GM_NQ + q] : 0.0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < GM_NQ; ++q) {
        double t = v[q];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) t += __shfl_down(t, d, 64);
        if (lane == 0) s_w[wave][q] = t;
    }
    __syncthreads();
    if (threadIdx.x < GM_NQ) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < GM_THREADS / 64; ++w) t += s_w[w][threadIdx.x];
        s_q[threadIdx.x] = t;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------
// centre mix.  aux[0..2] = centroid (fp32, as the float64 mean rounded once), aux[3] = number of valid rows
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(GM_THREADS)
k_centre_mix_fwd(const float* __restrict__ geo, const float* __restrict__ logit, const double* __restrict__ partial, int nP,
                 int64_t N, float* __restrict__ out, float* __restrict__ aux) {
    __shared__ double s_q[GM_NQ];
    sum_partials(partial, nP, s_q);
    const double cnt = s_q[0];
    const float c0 = (float)(s_q[1] / cnt), c1 = (float)(s_q[2] / cnt), c2 = (float)(s_q[3] / cnt);
    if (blockIdx.x == 0 && threadIdx.x == 0) { aux[0] = c0; aux[1] = c1; aux[2] = c2; aux[3] = (float)cnt; }
    const int64_t n = (int64_t)blockIdx.x * GM_THREADS + threadIdx.x;
    if (n >= N) return;
    const float g = sigmoid_f(logit[n]), h = 1.0f - g;
    out[n * 3 + 0] = c0 * g + geo[n * 3 + 0] * h;
    out[n * 3 + 1] = c1 * g + geo[n * 3 + 1] * h;
    out[n * 3 + 2] = c2 * g + geo[n * 3 + 2] * h;
}

// partial: moments of a = dout with weight sigmoid(logit) (no mask): d centre = sum_n g_n dout_n
__global__ void __launch_bounds__(GM_THREADS)
k_centre_mix_bwd(const float* __restrict__ dout, const float* __restrict__ geo, const float* __restrict__ logit,
                 const float* __restrict__ m, const float* __restrict__ aux, const double* __restrict__ partial, int nP,
                 int64_t N, float* __restrict__ dgeo, float* __restrict__ dlogit) {
    __shared__ double s_q[GM_NQ];
    sum_partials(partial, nP, s_q);
    const int64_t n = (int64_t)blockIdx.x * GM_THREADS + threadIdx.x;
    if (n >= N) return;
    const float cnt = aux[3];
    const float share = (m && !(m[n] > 0.f)) ? 0.f : 1.0f / cnt;
    const float g = sigmoid_f(logit[n]), h = 1.0f - g;
    float dg = 0.f;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float G = dout[n * 3 + d];
        dgeo[n * 3 + d] = G * h + (float)s_q[1 + d] * share;
        dg += G * (aux[d] - geo[n * 3 + d]);
    }
    dlogit[n] = dg * g * h;
}

// ---------------------------------------------------------------------------------------------------------------
// cloud frame.  aux[0..8] = V (row-major, eigenvectors in columns), aux[9..11] = centre, aux[12] = max(valid rows, 1)
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(GM_THREADS)
k_cloud_frame_fwd(const float* __restrict__ x, const float* __restrict__ m, const double* __restrict__ partial, int nP,
                  int64_t N, float* __restrict__ y, float* __restrict__ aux) {
    __shared__ double s_q[GM_NQ];
    __shared__ float s_a[13];
    sum_partials(partial, nP, s_q);
    if (threadIdx.x == 0) {
        const double cnt = s_q[0], cntc = cnt > 1.0 ? cnt : 1.0;
        const float c[3] = {(float)(s_q[1] / cntc), (float)(s_q[2] / cntc), (float)(s_q[3] / cntc)};
        // sum_n m_n (x_n - c)(x_n - c)^T about the ROUNDED centre, from the raw moments (float64: no cancellation to speak of)
        const double S1[3] = {s_q[1], s_q[2], s_q[3]};
        const double S2[6] = {s_q[4], s_q[5], s_q[6], s_q[7], s_q[8], s_q[9]};
        const int ia[6] = {0, 0, 0, 1, 1, 2}, ib[6] = {0, 1, 2, 1, 2, 2};
        float up[6];
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const double ca = c[ia[t]], cb = c[ib[t]];
            up[t] = (float)(S2[t] - ca * S1[ib[t]] - cb * S1[ia[t]] + cnt * ca * cb);
        }
        float v[9];
        eigh3_upper(up, v, nullptr);
#pragma unroll
        for (int i = 0; i < 9; ++i) s_a[i] = v[i];
        s_a[9] = c[0]; s_a[10] = c[1]; s_a[11] = c[2]; s_a[12] = (float)cntc;
        if (blockIdx.x == 0)
            for (int i = 0; i < 13; ++i) aux[i] = s_a[i];
    }
    __syncthreads();
    const int64_t n = (int64_t)blockIdx.x * GM_THREADS + threadIdx.x;
    if (n >= N) return;
    const float mk = m ? m[n] : 1.0f;
    const float x0 = x[n * 3 + 0] - s_a[9] * mk, x1 = x[n * 3 + 1] - s_a[10] * mk, x2 = x[n * 3 + 2] - s_a[11] * mk;
#pragma unroll
    for (int e = 0; e < 3; ++e) y[n * 3 + e] = x0 * s_a[e] + x1 * s_a[3 + e] + x2 * s_a[6 + e];
}

// partial: moments of a = dy with the row mask: sum_n m_n dy_n
__global__ void __launch_bounds__(GM_THREADS)
k_cloud_frame_bwd(const float* __restrict__ dy, const float* __restrict__ m, const float* __restrict__ aux,
                  const double* __restrict__ partial, int nP, int64_t N, float* __restrict__ dx) {
    __shared__ double s_q[GM_NQ];
    sum_partials(partial, nP, s_q);
    const int64_t n = (int64_t)blockIdx.x * GM_THREADS + threadIdx.x;
    if (n >= N) return;
    const float share = (m && !(m[n] > 0.f)) ? 0.f : 1.0f / aux[12];
    const float S[3] = {(float)s_q[1], (float)s_q[2], (float)s_q[3]};
    const float g0 = dy[n * 3 + 0], g1 = dy[n * 3 + 1], g2 = dy[n * 3 + 2];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float v0 = aux[3 * d + 0], v1 = aux[3 * d + 1], v2 = aux[3 * d + 2];
        // d xc = dy V^T; the centre takes -sum m d xc and hands it back to every valid row
        dx[n * 3 + d] = (g0 * v0 + g1 * v1 + g2 * v2) - (S[0] * v0 + S[1] * v1 + S[2] * v2) * share;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// edge frame: one thread per atom
// ---------------------------------------------------------------------------------------------------------------
constexpr int EF_MAXK = 16;

__global__ void __launch_bounds__(128)
k_edge_frame_fwd(const float* __restrict__ geo, const float4* __restrict__ gj, const uint8_t* __restrict__ mask, int64_t N,
                 int K, float* __restrict__ y, float* __restrict__ d2, float* __restrict__ V) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float xi = geo[n * 3 + 0], yi = geo[n * 3 + 1], zi = geo[n * 3 + 2];
    float r[EF_MAXK][3];
    float mk[EF_MAXK];
    float cnt = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int k = 0; k < EF_MAXK; ++k) {
        if (k < K) {
            const float4 p = gj[n * K + k];
            r[k][0] = xi - p.x; r[k][1] = yi - p.y; r[k][2] = zi - p.z;
            mk[k] = mask[n * K + k] ? 1.f : 0.f;
            d2[n * K + k] = r[k][0] * r[k][0] + r[k][1] * r[k][1] + r[k][2] * r[k][2];
            if (mk[k] > 0.f) { c0 += r[k][0]; c1 += r[k][1]; c2 += r[k][2]; cnt += 1.f; }
        }
    }
    const float cntc = cnt > 1.f ? cnt : 1.f;
    c0 /= cntc; c1 /= cntc; c2 /= cntc;
    float up[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < EF_MAXK; ++k) {
        if (k < K) {
            r[k][0] -= c0 * mk[k]; r[k][1] -= c1 * mk[k]; r[k][2] -= c2 * mk[k];     // xc: masked slots keep the raw offset (:94)
            if (mk[k] > 0.f) {
                up[0] += r[k][0] * r[k][0]; up[1] += r[k][0] * r[k][1]; up[2] += r[k][0] * r[k][2];
                up[3] += r[k][1] * r[k][1]; up[4] += r[k][1] * r[k][2]; up[5] += r[k][2] * r[k][2];
            }
        }
    }
    float v[9];
    eigh3_upper(up, v, nullptr);
#pragma unroll
    for (int i = 0; i < 9; ++i) V[n * 9 + i] = v[i];
#pragma unroll
    for (int k = 0; k < EF_MAXK; ++k) {
        if (k < K) {
#pragma unroll
            for (int e = 0; e < 3; ++e)
                y[(n * K + k) * 3 + e] = r[k][0] * v[e] + r[k][1] * v[3 + e] + r[k][2] * v[6 + e];
        }
    }
}

// d rel_k = dy_k V^T - m_k / cnt * sum_l m_l dy_l V^T + 2 rel_k dd2_k;  d geo_i = sum_k d rel_k, d gj_k = -d rel_k
__global__ void __launch_bounds__(128)
k_edge_frame_bwd(const float* __restrict__ geo, const float4* __restrict__ gj, const uint8_t* __restrict__ mask,
                 const float* __restrict__ V, const float* __restrict__ dy, const float* __restrict__ dd2, int64_t N, int K,
                 float* __restrict__ dgeo, float4* __restrict__ dgj) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float xi = geo[n * 3 + 0], yi = geo[n * 3 + 1], zi = geo[n * 3 + 2];
    float v[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) v[i] = V[n * 9 + i];
    float g[EF_MAXK][3];
    float cnt = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < EF_MAXK; ++k) {
        if (k < K) {
            const float a = dy ? dy[(n * K + k) * 3 + 0] : 0.f, b = dy ? dy[(n * K + k) * 3 + 1] : 0.f,
                        c = dy ? dy[(n * K + k) * 3 + 2] : 0.f;
#pragma unroll
            for (int d = 0; d < 3; ++d) g[k][d] = a * v[3 * d] + b * v[3 * d + 1] + c * v[3 * d + 2];
            if (mask[n * K + k]) { s0 += g[k][0]; s1 += g[k][1]; s2 += g[k][2]; cnt += 1.f; }
        }
    }
    const float inv = 1.0f / (cnt > 1.f ? cnt : 1.f);
    float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int k = 0; k < EF_MAXK; ++k) {
        if (k < K) {
            const float4 p = gj[n * K + k];
            const float mkk = mask[n * K + k] ? inv : 0.f;
            const float q = dd2 ? 2.0f * dd2[n * K + k] : 0.f;
            const float r0 = g[k][0] - s0 * mkk + q * (xi - p.x), r1 = g[k][1] - s1 * mkk + q * (yi - p.y),
                        r2 = g[k][2] - s2 * mkk + q * (zi - p.z);
            t0 += r0; t1 += r1; t2 += r2;
            dgj[n * K + k] = make_float4(-r0, -r1, -r2, 0.f);
        }
    }
    dgeo[n * 3 + 0] = t0; dgeo[n * 3 + 1] = t1; dgeo[n * 3 + 2] = t2;
}

// ---------------------------------------------------------------------------------------------------------------
// attention logits.  qa [N, 4]: (a_q of the H heads, a_k of the H heads); qan [N, K, 4]: qa gathered by neighbour;
// le [N, K, H]; one 16-lane group per (atom, head), lane = neighbour slot
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T group16_sum(T v) {
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ float group16_max(float v) {
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
    return v;
}

__global__ void __launch_bounds__(GM_THREADS)
k_attn_logits_fwd(const float* __restrict__ qa, const float* __restrict__ qan, const float* __restrict__ le,
                  const uint8_t* __restrict__ mask, int64_t N, int K, int H, float p, const int64_t* __restrict__ seed,
                  float* __restrict__ prob, float* __restrict__ attn) {
    const int64_t t = (int64_t)blockIdx.x * GM_THREADS + threadIdx.x;
    const int k = (int)(t & 15);
    const int64_t nh = t >> 4;
    if (nh >= N * H) return;                         // (whole 16-lane groups leave together)
    const int64_t n = nh / H;
    const int hd = (int)(nh - n * H);
    const bool valid = k < K;
    float l = -INFINITY;
    if (valid) {
        const int64_t e = n * K + k;
        l = qa[n * 4 + hd] + qan[e * 4 + H + hd] + le[e * H + hd];
        if (!mask[e]) l = -1e9f;                     // :490 masked_fill(~mask, -1e9): a row without neighbours is uniform
    }
    const float mx = group16_max(l);
    const float ex = valid ? expf(l - mx) : 0.f;
    const float pr = ex / group16_sum(ex);
    if (!valid) return;
    const int64_t o = nh * K + k;
    float a = pr;
    if (p > 0.f) {
        const DropKey key = drop_key((uint64_t)seed[0]);
        a = pr * keep_scale(key, (uint64_t)o, drop_threshold(p), 1.0f / (1.0f - p));
        prob[o] = pr;
    }
    attn[o] = a;
}

__global__ void __launch_bounds__(GM_THREADS)
k_attn_logits_bwd(const float* __restrict__ prob, const float* __restrict__ dattn, const uint8_t* __restrict__ mask, int64_t N,
                  int K, int H, float p, const int64_t* __restrict__ seed, float* __restrict__ dqa, float* __restrict__ dqan,
                  float* __restrict__ dle) {
    const int64_t t = (int64_t)blockIdx.x * GM_THREADS + threadIdx.x;
    const int k = (int)(t & 15);
    const int64_t nh = t >> 4;
    if (nh >= N * H) return;
    const int64_t n = nh / H;
    const int hd = (int)(nh - n * H);
    const bool valid = k < K;
    const int64_t o = nh * K + k;
    float pr = 0.f, dp = 0.f;
    if (valid) {
        pr = prob[o];
        dp = dattn[o];
        if (p > 0.f) {
            const DropKey key = drop_key((uint64_t)seed[0]);
            dp *= keep_scale(key, (uint64_t)o, drop_threshold(p), 1.0f / (1.0f - p));
        }
    }
    const float inner = group16_sum(pr * dp);
    float dl = pr * (dp - inner);
    const int64_t e = n * K + k;
    if (valid && !mask[e]) dl = 0.f;
    const float tot = group16_sum(valid ? dl : 0.f);
    if (valid) {
        dle[e * H + hd] = dl;
        dqan[e * 4 + H + hd] = dl;
        dqan[e * 4 + hd] = 0.f;
        if (hd == 0)
            for (int c = 2 * H; c < 4; ++c) dqan[e * 4 + c] = 0.f;
    }
    if (k == 0) {
        dqa[n * 4 + hd] = tot;
        dqa[n * 4 + H + hd] = 0.f;
        if (hd == 0)
            for (int c = 2 * H; c < 4; ++c) dqa[n * 4 + c] = 0.f;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The edge logits' folded weights (fa_former_layer.py:483-489: Linear(deh, 1) of the edge QUERY, itself a Linear of the
// edge features -- folded at weight level, faformer.MLPAttnEdgeAggregation):
//   u[h, j] = sum_d w_e[d] W[h * deh + d, j],   c[h] = sum_d w_e[d] b[h * deh + d]          (W: the query rows of the Linear)
// and their backward.  Twenty elementwise / reduction launches per layer and step as torch expressions.
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_elw_fwd(const float* __restrict__ W, int64_t ldw, const float* __restrict__ b, const float* __restrict__ we, int H, int deh,
          int de, float* __restrict__ u, float* __restrict__ c) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < H * de) {
        const int hh = t / de, j = t - hh * de;
        // (514 threads in all: each sums deh rows.  Sixteen loads in flight per thread -- one at a time the 128-deep chain of
        // L2 round trips took 50 us per call; the products are summed in the same order)
        const float* __restrict__ wp = W + (int64_t)(hh * deh) * ldw + j;
        float acc = 0.f;
        int d = 0;
        for (; d + 16 <= deh; d += 16) {
            float v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = wp[(int64_t)(d + q) * ldw];
#pragma unroll
            for (int q = 0; q < 16; ++q) acc = fmaf(we[d + q], v[q], acc);
        }
        for (; d < deh; ++d) acc = fmaf(we[d], wp[(int64_t)d * ldw], acc);
        u[t] = acc;
    } else if (t < H * de + H) {
        const int hh = t - H * de;
        float acc = 0.f;
        for (int d = 0; d < deh; ++d) acc = fmaf(we[d], b[hh * deh + d], acc);
        c[hh] = acc;
    }
}

// one workgroup per d: dW[h * deh + d, :] (+)= w_e[d] du[h, :], db[h * deh + d] (+)= w_e[d] dc[h],
// dwe[d] (+)= sum_h (W[h * deh + d, :] . du[h, :] + b[h * deh + d] dc[h])
__global__ void __launch_bounds__(256)
k_elw_bwd(const float* __restrict__ W, int64_t ldw, const float* __restrict__ b, const float* __restrict__ we,
          const float* __restrict__ du, const float* __restrict__ dc, int H, int deh, int de, float* __restrict__ dW,
          int64_t lddw, float* __restrict__ db, float* __restrict__ dwe, int acc_w, int acc_b, int acc_e) {
    __shared__ float s_w[4];
    const int d = blockIdx.x;
    const float wd = we[d];
    float part = 0.f;
    for (int hh = 0; hh < H; ++hh) {
        const int64_t r = hh * deh + d;
        for (int j = threadIdx.x; j < de; j += 256) {
            const float g = du ? du[hh * de + j] : 0.f;
            float* o = dW + r * lddw + j;
            *o = acc_w ? *o + wd * g : wd * g;
            part = fmaf(W[r * ldw + j], g, part);
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) part += __shfl_down(part, o, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
        for (int hh = 0; hh < H; ++hh) {
            const float g = dc ? dc[hh] : 0.f;
            const int r = hh * deh + d;
            tot = fmaf(b[r], g, tot);
            db[r] = acc_b ? db[r] + wd * g : wd * g;
        }
        dwe[d] = acc_e ? dwe[d] + tot : tot;
    }
}

static inline int moments_grid(int64_t N) { return eqh_grid_for(N, GM_THREADS, GM_MAXP); }

}  // namespace

extern "C" size_t faf_moments_workspace_bytes(int64_t N) {
    return N < 0 ? 0 : (size_t)moments_grid(N) * GM_NQ * sizeof(double);
}

static int launch_moments(const float* a, const float* m, const float* s, int s_sigmoid, int64_t N, double* partial,
                          hipStream_t stream) {
    hipLaunchKernelGGL(k_moments, dim3(moments_grid(N)), dim3(GM_THREADS), 0, stream, a, m, s, s_sigmoid, N, partial);
    return hipGetLastError() == hipSuccess ? EQH_OK : EQH_ERR_LAUNCH;
}

extern "C" int faf_centre_mix_fwd(const float* geo, const float* logit, const float* row_mask, int64_t N, float* out,
                                  float* aux, void* workspace, size_t workspace_bytes, void* stream_) {
    if (N < 0) return EQH_ERR_ARG;
    if (N == 0) return EQH_OK;
    if (!geo || !logit || !out || !aux || !workspace || workspace_bytes < faf_moments_workspace_bytes(N)) return EQH_ERR_ARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    double* partial = static_cast<double*>(workspace);
    if (int rc = launch_moments(geo, row_mask, nullptr, 0, N, partial, stream)) return rc;
    hipLaunchKernelGGL(k_centre_mix_fwd, dim3(eqh_grid_for(N, GM_THREADS, 1 << 30)), dim3(GM_THREADS), 0, stream, geo, logit,
                       partial, moments_grid(N), N, out, aux);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}

extern "C" int faf_centre_mix_bwd(const float* dout, const float* geo, const float* logit, const float* row_mask,
                                  const float* aux, int64_t N, float* dgeo, float* dlogit, void* workspace,
                                  size_t workspace_bytes, void* stream_) {
    if (N < 0) return EQH_ERR_ARG;
    if (N == 0) return EQH_OK;
    if (!dout || !geo || !logit || !aux || !dgeo || !dlogit || !workspace || workspace_bytes < faf_moments_workspace_bytes(N))
        return EQH_ERR_ARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    double* partial = static_cast<double*>(workspace);
    if (int rc = launch_moments(dout, nullptr, logit, 1, N, partial, stream)) return rc;
    hipLaunchKernelGGL(k_centre_mix_bwd, dim3(eqh_grid_for(N, GM_THREADS, 1 << 30)), dim3(GM_THREADS), 0, stream, dout, geo,
                       logit, row_mask, aux, partial, moments_grid(N), N, dgeo, dlogit);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}

extern "C" int faf_cloud_frame_fwd(const float* x, const float* row_mask, int64_t N, float* y, float* aux, void* workspace,
                                   size_t workspace_bytes, void* stream_) {
    if (N < 0) return EQH_ERR_ARG;
    if (N == 0) return EQH_OK;
    if (!x || !y || !aux || !workspace || workspace_bytes < faf_moments_workspace_bytes(N)) return EQH_ERR_ARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    double* partial = static_cast<double*>(workspace);
    if (int rc = launch_moments(x, row_mask, nullptr, 0, N, partial, stream)) return rc;
    hipLaunchKernelGGL(k_cloud_frame_fwd, dim3(eqh_grid_for(N, GM_THREADS, 1 << 30)), dim3(GM_THREADS), 0, stream, x, row_mask,
                       partial, moments_grid(N), N, y, aux);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}

extern "C" int faf_cloud_frame_bwd(const float* dy, const float* row_mask, const float* aux, int64_t N, float* dx,
                                   void* workspace, size_t workspace_bytes, void* stream_) {
    if (N < 0) return EQH_ERR_ARG;
    if (N == 0) return EQH_OK;
    if (!dy || !aux || !dx || !workspace || workspace_bytes < faf_moments_workspace_bytes(N)) return EQH_ERR_ARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    double* partial = static_cast<double*>(workspace);
    if (int rc = launch_moments(dy, row_mask, nullptr, 0, N, partial, stream)) return rc;
    hipLaunchKernelGGL(k_cloud_frame_bwd, dim3(eqh_grid_for(N, GM_THREADS, 1 << 30)), dim3(GM_THREADS), 0, stream, dy, row_mask,
                       aux, partial, moments_grid(N), N, dx);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}

extern "C" int faf_edge_frame_fwd(const float* geo, const float* gj, const uint8_t* mask, int64_t N, int32_t K, float* y,
                                  float* d2, float* V, void* stream_) {
    if (N < 0 || K < 1 || K > EF_MAXK) return EQH_ERR_ARG;
    if (N == 0) return EQH_OK;
    if (!geo || !gj || !mask || !y || !d2 || !V || !eqh_aligned16(gj)) return EQH_ERR_ARG;
    hipLaunchKernelGGL(k_edge_frame_fwd, dim3(eqh_grid_for(N, 128, 1 << 30)), dim3(128), 0, static_cast<hipStream_t>(stream_),
                       geo, reinterpret_cast<const float4*>(gj), mask, N, K, y, d2, V);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}

extern "C" int faf_edge_frame_bwd(const float* geo, const float* gj, const uint8_t* mask, const float* V, const float* dy,
                                  const float* dd2, int64_t N, int32_t K, float* dgeo, float* dgj, void* stream_) {
    if (N < 0 || K < 1 || K > EF_MAXK) return EQH_ERR_ARG;
    if (N == 0) return EQH_OK;
    if (!geo || !gj || !mask || !V || !dgeo || !dgj || !eqh_aligned16(gj) || !eqh_aligned16(dgj)) return EQH_ERR_ARG;
    hipLaunchKernelGGL(k_edge_frame_bwd, dim3(eqh_grid_for(N, 128, 1 << 30)), dim3(128), 0, static_cast<hipStream_t>(stream_),
                       geo, reinterpret_cast<const float4*>(gj), mask, V, dy, dd2, N, K, dgeo, reinterpret_cast<float4*>(dgj));
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}

extern "C" int faf_attn_logits_fwd(const float* qa, const float* qan, const float* le, const uint8_t* mask, int64_t N,
                                   int32_t K, int32_t H, float p, const int64_t* seed, float* prob, float* attn,
                                   void* stream_) {
    if (N < 0 || K < 1 || K > 16 || H < 1 || H > 2 || p < 0.f || p >= 1.f) return EQH_ERR_ARG;
    if (N == 0) return EQH_OK;
    if (!qa || !qan || !le || !mask || !attn || (p > 0.f && (!seed || !prob))) return EQH_ERR_ARG;
    hipLaunchKernelGGL(k_attn_logits_fwd, dim3(eqh_grid_for(N * H * 16, GM_THREADS, 1 << 30)), dim3(GM_THREADS), 0,
                       static_cast<hipStream_t>(stream_), qa, qan, le, mask, N, K, H, p, seed, prob, attn);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}

extern "C" int faf_attn_logits_bwd(const float* prob, const float* dattn, const uint8_t* mask, int64_t N, int32_t K,
                                   int32_t H, float p, const int64_t* seed, float* dqa, float* dqan, float* dle,
                                   void* stream_) {
    if (N < 0 || K < 1 || K > 16 || H < 1 || H > 2 || p < 0.f || p >= 1.f) return EQH_ERR_ARG;
    if (N == 0) return EQH_OK;
    if (!prob || !dattn || !mask || !dqa || !dqan || !dle || (p > 0.f && !seed)) return EQH_ERR_ARG;
    hipLaunchKernelGGL(k_attn_logits_bwd, dim3(eqh_grid_for(N * H * 16, GM_THREADS, 1 << 30)), dim3(GM_THREADS), 0,
                       static_cast<hipStream_t>(stream_), prob, dattn, mask, N, K, H, p, seed, dqa, dqan, dle);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}

extern "C" int faf_edge_logit_weights_fwd(const float* W, int64_t ldw, const float* b, const float* we, int32_t H, int32_t deh,
                                          int32_t de, float* u, float* c, void* stream_) {
    if (H < 1 || deh < 1 || de < 1 || ldw < de || !W || !b || !we || !u || !c) return EQH_ERR_ARG;
    hipLaunchKernelGGL(k_elw_fwd, dim3(eqh_grid_for((int64_t)H * de + H, 256, 1 << 30)), dim3(256), 0,
                       static_cast<hipStream_t>(stream_), W, ldw, b, we, (int)H, (int)deh, (int)de, u, c);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}

extern "C" int faf_edge_logit_weights_bwd(const float* W, int64_t ldw, const float* b, const float* we, const float* du,
                                          const float* dc, int32_t H, int32_t deh, int32_t de, float* dW, int64_t lddw,
                                          float* db, float* dwe, int32_t acc_w, int32_t acc_b, int32_t acc_e, void* stream_) {
    if (H < 1 || deh < 1 || de < 1 || ldw < de || lddw < de || !W || !b || !we || !dW || !db || !dwe) return EQH_ERR_ARG;
    hipLaunchKernelGGL(k_elw_bwd, dim3(deh), dim3(256), 0, static_cast<hipStream_t>(stream_), W, ldw, b, we, du, dc, (int)H,
                       (int)deh, (int)de, dW, lddw, db, dwe, (int)acc_w, (int)acc_b, (int)acc_e);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}
