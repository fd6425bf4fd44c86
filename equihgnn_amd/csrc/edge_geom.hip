// Per-edge geometry of the Equiformer front-end in ONE launch: rel_pos = x_i - x_j, the column of the degree-1
// Wigner matrix D that the live tensor-product pairs read, the radius mask and the masked-mean weights.
//
// Replaces (reference) equiformer_layer.py:1250-1252,1317-1346 (rel_pos of the selected neighbours, the radius
// mask) and equiformer/basis.py:194-215 -> :169-191 -> irr_repr.py:105-118,23-32 (rotation of r_ij onto the
// y axis in float64, ZYZ Euler angles, D[1] = Z(a) J Z(b) J Z(c)), plus the mask/count arithmetic of
// equiformer/utils.py:71-82 (masked_mean) -- ~10 elementwise launches in the first version of this repo.
//
// Which part of D is needed: only the pairs (0 -> 1) and (1 -> 0) reach the type-0 output, and
// both read D[:, m = 0], the image of the y axis:  D e_1 = Z(a) J Z(b) J Z(c) e_1 = (sin a sin b, cos b, cos a sin b)
// (Z(c) fixes e_1, J swaps e_0 and e_1 and flips e_2).  With v = normalize(R y) the Euler angles are
// b = acos(v_y), a = atan2(v_x, v_z) (irr_repr.py:110-114), so the column is v again -- computed here through
// the same fp32 acos / atan2 / sin / cos round trip as the reference, so that the rounding is the same.
// For a generic direction v = r_hat.  It is NOT r_hat where the reference's rotation is degenerate
// (basis.py:187-190): R = 2 (x + y)(x + y)^T / max(|x + y|^2, 1e-6) - I with x = r_hat (float64), which stops
// being a rotation when |x + y|^2 < 1e-6 (r_hat within 1e-3 rad of -y): then R y = (s / 1e-6)(x + y) - y,
// s = |x + y|^2, which slides from x (s = 1e-6) to -y (s = 0).  Coincident atoms give x = 0 and R y = y.
// All of it is reproduced: tests/golden/equiformer_D.npz holds the reference's D for such rows.
//
// Layout: 16 lanes per node (lane = neighbour slot, K <= 16), four nodes per wavefront; the per-node count
// of in-radius neighbours is a popcount of the wavefront ballot.  No gradient (positions are data and the
// reference builds D under no_grad).
#include <math.h>

#include "common.h"

namespace {

__global__ void __launch_bounds__(256)
k_edge_geom(const float* __restrict__ pos, const int* __restrict__ nbr, const float* __restrict__ dist, int64_t N, int K,
            float radius, float* __restrict__ rhat, float* __restrict__ maskf, float* __restrict__ mean_w,
            float* __restrict__ mean_w_rhat, float* __restrict__ dmat) {
    const int lane = threadIdx.x & 63;
    const int slot = lane & 15, sub = lane >> 4;
    const int64_t node0 = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 4;
    const int64_t i = node0 + sub;
    const bool live = i < N && slot < K;
    float c0 = 0.f, c1 = 1.f, c2 = 0.f, d = INFINITY;
    if (live) {
        const int64_t e = i * K + slot;
        const int j = nbr[e];
        d = dist[e];
        const float rx = __fsub_rn(pos[3 * i], pos[3 * j]);          // x_i - x_j, equiformer_layer.py:1250
        const float ry = __fsub_rn(pos[3 * i + 1], pos[3 * j + 1]);
        const float rz = __fsub_rn(pos[3 * i + 2], pos[3 * j + 2]);
        // basis.py:183-190 in float64: x = l2norm(r) (F.normalize, eps 1e-12), xy = x + y, R y = 2 xy xy_1 / max(s, 1e-6) - y
        const double x0 = rx, x1 = ry, x2 = rz;
        double nrm = sqrt(x0 * x0 + x1 * x1 + x2 * x2);
        nrm = nrm > 1e-12 ? nrm : 1e-12;
        const double a0 = x0 / nrm, a1 = x1 / nrm + 1.0, a2 = x2 / nrm;
        double s = a0 * a0 + a1 * a1 + a2 * a2;
        s = s > 1e-6 ? s : 1e-6;
        const double t = 2.0 * a1 / s;
        const float u0 = (float)(t * a0), u1 = (float)(t * a1 - 1.0), u2 = (float)(t * a2);   // R.type(float32) @ y
        // irr_repr.py:110-114 in float32: v = l2norm(R y).clamp(-1, 1); b = acos(v_y); a = atan2(v_x, v_z)
        float n2 = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(u0, u0), __fmul_rn(u1, u1)), __fmul_rn(u2, u2)));
        n2 = n2 > 1e-12f ? n2 : 1e-12f;
        const float v0 = fminf(fmaxf(u0 / n2, -1.f), 1.f), v1 = fminf(fmaxf(u1 / n2, -1.f), 1.f),
                    v2 = fminf(fmaxf(u2 / n2, -1.f), 1.f);
        const float b = acosf(v1), a = atan2f(v0, v2);
        const float sb = sinf(b);
        c0 = __fmul_rn(sinf(a), sb);          // D[:, m=0] = (sin a sin b, cos b, cos a sin b)
        c1 = cosf(b);
        c2 = __fmul_rn(cosf(a), sb);
        if (dmat) {
            // the whole D[1] = Z(a) J Z(b) J Z(c) (irr_repr.py:23-32), needed by the degree-1 outputs (depth > 1): the
            // third angle comes from rot(a, b, 0)^T R (irr_repr.py:116-117), with R the float32 cast of the float64
            // rotation above: R = t' xy xy^T - I, t' = 2 / max(s, 1e-6)
            const double tt = 2.0 / s;
            const double xy[3] = {a0, a1, a2};
            float R[3][3];
            for (int p = 0; p < 3; ++p)
                for (int q = 0; q < 3; ++q) R[p][q] = (float)(tt * xy[p] * xy[q] - (p == q ? 1.0 : 0.0));
            const float ca = cosf(a), sa = sinf(a), cb = cosf(b);
            // first column of rot_z(a) rot_y(b) = (ca cb, sa cb, -sb); r2[0][j] = that column . R[:, j]
            const float r00 = ca * cb * R[0][0] + sa * cb * R[1][0] - sb * R[2][0];
            const float r02 = ca * cb * R[0][2] + sa * cb * R[1][2] - sb * R[2][2];
            const float g = atan2f(r02, r00);
            const float cg = cosf(g), sg = sinf(g);
            // Z(t) = [[c,0,s],[0,1,0],[-s,0,c]], J = [[0,1,0],[1,0,0],[0,0,-1]]:  J Z(t) = [[0,1,0],[c,0,s],[s,0,-c]]
            const float JB[3][3] = {{0.f, 1.f, 0.f}, {cb, 0.f, sb}, {sb, 0.f, -cb}};
            const float JC[3][3] = {{0.f, 1.f, 0.f}, {cg, 0.f, sg}, {sg, 0.f, -cg}};
            const float ZA[3][3] = {{ca, 0.f, sa}, {0.f, 1.f, 0.f}, {-sa, 0.f, ca}};
            float M[3][3], Dm[3][3];
            for (int p = 0; p < 3; ++p)
                for (int q = 0; q < 3; ++q) M[p][q] = JB[p][0] * JC[0][q] + JB[p][1] * JC[1][q] + JB[p][2] * JC[2][q];
            for (int p = 0; p < 3; ++p)
                for (int q = 0; q < 3; ++q) Dm[p][q] = ZA[p][0] * M[0][q] + ZA[p][1] * M[1][q] + ZA[p][2] * M[2][q];
            float* dst = dmat + 9 * e;
            for (int p = 0; p < 3; ++p)
                for (int q = 0; q < 3; ++q) dst[3 * p + q] = Dm[p][q];
        }
    }
    const bool in = live && d <= radius;                                  // neighbor_mask, :1339
    const unsigned long long bal = __ballot(in);
    const int cnt = __popcll((bal >> (16 * sub)) & 0xffffull);
    if (live) {
        const int64_t e = i * K + slot;
        const float w = in ? 1.f / (float)cnt : 0.f;                      // masked mean weight; 0 for an empty set
        rhat[3 * e] = c0; rhat[3 * e + 1] = c1; rhat[3 * e + 2] = c2;
        maskf[e] = in ? 1.f : 0.f;
        mean_w[e] = w;
        mean_w_rhat[3 * e] = __fmul_rn(w, c0); mean_w_rhat[3 * e + 1] = __fmul_rn(w, c1); mean_w_rhat[3 * e + 2] = __fmul_rn(w, c2);
    }
}

// out[n, m, c] = sum_k w3[n, k, m] t[n, k, c]  (m < 3; component-major rows, so that every later per-channel product sees
// [3 N, C] rows): the masked mean over the K neighbour slots of t[e, c] r_hat[e, m]
// (the pooled (0 -> 1) pair, equiformer_layer.py:432-436 with utils.py:71-82) without the [E, C, 3] product and without the
// batched [C x K] . [K x 3] library products, whose backward handed the next kernel a transposed [E, C] gradient (a 52 us
// strided copy at the BASELINE batch).  One wavefront per node, lane = four channels; the backward is the same loop turned
// around: dt[n, k, c] = sum_m w3[n, k, m] dout[n, m, c].
template <bool BWD>
__global__ void __launch_bounds__(256)
k_pool3(const float* __restrict__ src, const float* __restrict__ w3, int64_t N, int K, int C, float* __restrict__ dst) {
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const float* __restrict__ wn = w3 + n * K * 3;
    for (int c = 4 * lane; c < C; c += 256) {
        if (!BWD) {
            float4 a0 = f4_zero(), a1 = f4_zero(), a2 = f4_zero();
            for (int k = 0; k < K; ++k) {
                const float4 t = *reinterpret_cast<const float4*>(src + (n * K + k) * C + c);
                const float w0 = wn[3 * k], w1 = wn[3 * k + 1], w2 = wn[3 * k + 2];
                a0.x = fmaf(w0, t.x, a0.x); a0.y = fmaf(w0, t.y, a0.y); a0.z = fmaf(w0, t.z, a0.z); a0.w = fmaf(w0, t.w, a0.w);
                a1.x = fmaf(w1, t.x, a1.x); a1.y = fmaf(w1, t.y, a1.y); a1.z = fmaf(w1, t.z, a1.z); a1.w = fmaf(w1, t.w, a1.w);
                a2.x = fmaf(w2, t.x, a2.x); a2.y = fmaf(w2, t.y, a2.y); a2.z = fmaf(w2, t.z, a2.z); a2.w = fmaf(w2, t.w, a2.w);
            }
            float* __restrict__ o = dst + n * 3 * C + c;
            *reinterpret_cast<float4*>(o) = a0;
            *reinterpret_cast<float4*>(o + C) = a1;
            *reinterpret_cast<float4*>(o + 2 * C) = a2;
        } else {
            const float* __restrict__ d = src + n * 3 * C + c;
            const float4 d0 = *reinterpret_cast<const float4*>(d), d1 = *reinterpret_cast<const float4*>(d + C),
                         d2 = *reinterpret_cast<const float4*>(d + 2 * C);
            for (int k = 0; k < K; ++k) {
                const float w0 = wn[3 * k], w1 = wn[3 * k + 1], w2 = wn[3 * k + 2];
                float4 r;
                r.x = fmaf(w2, d2.x, fmaf(w1, d1.x, w0 * d0.x));
                r.y = fmaf(w2, d2.y, fmaf(w1, d1.y, w0 * d0.y));
                r.z = fmaf(w2, d2.z, fmaf(w1, d1.z, w0 * d0.z));
                r.w = fmaf(w2, d2.w, fmaf(w1, d1.w, w0 * d0.w));
                *reinterpret_cast<float4*>(dst + (n * K + k) * C + c) = r;
            }
        }
    }
}

}  // namespace

extern "C" int eqf_pool3(const float* src, const float* w3, int64_t N, int32_t K, int32_t C, int32_t backward, float* dst,
                         void* stream) {
    if (N < 0 || K < 1 || C < 4 || (C & 3)) return EQH_ERR_ARG;
    if (N == 0) return EQH_OK;
    if (!src || !w3 || !dst) return EQH_ERR_ARG;
    if (!eqh_aligned16(src) || !eqh_aligned16(dst)) return EQH_ERR_ALIGN;
    const int64_t blocks = (N + 3) / 4;
    if (blocks > 0x7fffffff) return EQH_ERR_RANGE;
    if (backward)
        hipLaunchKernelGGL(k_pool3<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, w3, N, (int)K, (int)C, dst);
    else
        hipLaunchKernelGGL(k_pool3<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, w3, N, (int)K, (int)C, dst);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}

extern "C" int eqf_edge_geometry(const float* pos, const int32_t* nbr, const float* dist, int64_t N, int32_t K,
                                 float radius, float* rhat, float* maskf, float* mean_w, float* mean_w_rhat,
                                 float* dmat, void* stream) {
    if (N < 0 || K < 1 || K > 16) return EQH_ERR_ARG;
    if (N == 0) return EQH_OK;
    if (!pos || !nbr || !dist || !rhat || !maskf || !mean_w || !mean_w_rhat) return EQH_ERR_ARG;
    const int64_t waves = (N + 3) / 4;
    const int64_t blocks = (waves + 3) / 4;
    if (blocks > 0x7fffffff) return EQH_ERR_ARG;
    hipLaunchKernelGGL(k_edge_geom, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pos, nbr, dist, N, (int)K, radius,
                       rhat, maskf, mean_w, mean_w_rhat, dmat);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}
