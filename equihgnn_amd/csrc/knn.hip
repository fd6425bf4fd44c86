// k nearest neighbours over the whole batch point cloud (brute force, never materialises N x N).
//
// Replaces the dense rel_coors[1,N,N,3] / rel_dist[1,N,N] build + torch.topk of
// egnn_layer.py:253-288 (mode 0: squared distance, self included) and
// equiformer_layer.py:1216-1346 (mode 1: true distance, self excluded).  As in the reference the
// search runs over the concatenated cloud of the whole batch: neighbours may belong to other
// molecules.
//
// One 64-lane wavefront owns FOUR queries (each candidate load is shared by them).  Candidates are
// streamed 64 at a time (one per lane);
// the running k-best list lives in lanes 0..k-1 of two registers (distance, index), sorted by
// (distance, index).  A candidate beats the list only if it is strictly closer than the current
// k-th entry; insertion is O(1) wave operations (ballot -> popcount -> shuffle-up).  A first pass over the
// candidates bounds the k-th distance from above (the k-th smallest of the 64 per-lane minima), so the second pass
// inserts about k + a few candidates instead of ~k*ln(N/k) + 64.
// Distance arithmetic: (dx*dx + dy*dy) + dz*dz with separately rounded products (no FMA).
// (Measured and not kept: holding all positions of a <= 12 k-atom cloud in LDS -- 64-84 us against 47 us at
// 4.7 k atoms.  The molecules of a batch overlap in space, so the k-th distance shrinks slowly and a query
// performs ~k ln(N/k) + 64 list insertions of ~25 wavefront instructions each: the kernel is bound by that
// serial chain, not by the candidate loads.)
#include <limits.h>
#include <math.h>

#include "common.h"

namespace {

// QPW = queries per wavefront.  With QPW = 4 every candidate load is shared by four queries (6x
// faster at 30 k atoms, where the one-query form is bound by L2 reads); small clouds prefer QPW = 1
// (more wavefronts in flight: 80 vs 127 us at 4.6 k atoms).
template <int MODE, int QPW>
__global__ void __launch_bounds__(256)
k_knn(const float* __restrict__ pos, int N, int k, int* __restrict__ nbr, float* __restrict__ dist,
      int* __restrict__ counts) {
    // Candidates reach the wavefronts through LDS: the workgroup's four wavefronts (four query groups) walk the same
    // candidate stream, so each block of 256 candidates is fetched from memory once per workgroup, as one 12-byte
    // strided load per thread, instead of once per wavefront (those strided loads -- 36 cache-line accesses per
    // 64 candidates -- kept the CU's L1 busy for the whole kernel once the insertion chain was shortened).
    __shared__ float s_pos[2][3][256];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int waves_per_block = blockDim.x >> 6;   // 4
    const int wave_stride = gridDim.x * waves_per_block;
    const int n_groups = (N + QPW - 1) / QPW;
    auto stage = [&](int buf, int c0) {
        const int j = c0 + (int)threadIdx.x;
        float x = 0.f, y = 0.f, z = 0.f;
        if (j < N) { x = pos[3 * j]; y = pos[3 * j + 1]; z = pos[3 * j + 2]; }
        s_pos[buf][0][threadIdx.x] = x; s_pos[buf][1][threadIdx.x] = y; s_pos[buf][2][threadIdx.x] = z;
    };
    // the LAST queries first: a padded batch parks its padding atoms at the end of the cloud and far away from it,
    // where every real atom is a near-tie candidate (many list insertions); started last, those few long-running
    // wavefronts were the tail of the launch
    for (int base = blockIdx.x * waves_per_block; base < n_groups; base += wave_stride) {   // workgroup-uniform
        const int g_ = base + wave;
        const bool active = g_ < n_groups;
        const int grp = active ? n_groups - 1 - g_ : 0;
        const int i0 = grp * QPW;
        float qx[QPW], qy[QPW], qz[QPW], ld[QPW], tau[QPW], lmin[QPW], T[QPW];
        int li[QPW];
#pragma unroll
        for (int t = 0; t < QPW; ++t) {
            const int i = (i0 + t < N) ? i0 + t : N - 1;  // tail queries repeat the last one (not stored)
            qx[t] = pos[3 * i]; qy[t] = pos[3 * i + 1]; qz[t] = pos[3 * i + 2];
            ld[t] = INFINITY;   // list: distance (lanes >= k stay +inf and never take part)
            li[t] = -1;         //       index
            tau[t] = INFINITY;
            lmin[t] = INFINITY;
            T[t] = INFINITY;
        }
        // Pass A: every lane keeps the smallest distance among ITS candidates (j = lane mod 64).  The 64 lane minima belong
        // to 64 distinct points, so their k-th smallest, T, is an upper bound of the final k-th distance: pass B only has to
        // offer candidates with d <= T to the list (about k + a few instead of ~k ln(N/k) + 64 serial insertions of ~25
        // wavefront instructions each, which bound the one-pass form).  The final list is the same: it is the k smallest
        // (distance, index) pairs either way.
        for (int pass = 0; pass < 2; ++pass) {
            __syncthreads();                       // the previous stream's last buffer is free
            stage(0, 0);
            for (int c0 = 0, buf = 0; c0 < N; c0 += 256, buf ^= 1) {
                __syncthreads();                   // buffer `buf` is staged; the other one is free
                if (c0 + 256 < N) stage(buf ^ 1, c0 + 256);
                if (!active) continue;
#pragma unroll
                for (int sub = 0; sub < 4; ++sub) {
                    const int j = c0 + 64 * sub + lane;
                    if (c0 + 64 * sub >= N) break;
                    const bool in_range = j < N;
                    const float px = s_pos[buf][0][64 * sub + lane], py = s_pos[buf][1][64 * sub + lane];
                    const float pz = s_pos[buf][2][64 * sub + lane];
#pragma unroll
                    for (int t = 0; t < QPW; ++t) {
                        bool valid = in_range;
                        if (MODE == 1) valid = valid && (j != i0 + t);
                        const float dx = __fsub_rn(qx[t], px);
                        const float dy = __fsub_rn(qy[t], py);
                        const float dz = __fsub_rn(qz[t], pz);
                        float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                        if (MODE == 1) d = __fsqrt_rn(d);
                        if (pass == 0) {
                            if (valid && d < lmin[t]) lmin[t] = d;
                            continue;
                        }
                        unsigned long long mask = __ballot(valid && d <= T[t] && d < tau[t]);
                        while (mask) {
                            const int bsel = __ffsll((long long)mask) - 1;  // lowest lane = lowest index first
                            mask &= mask - 1;
                            const float xd = __shfl(d, bsel, 64);
                            if (!(xd < tau[t])) continue;
                            const int p = __popcll(__ballot(lane < k && ld[t] <= xd));
                            const float ud = __shfl_up(ld[t], 1, 64);
                            const int ui = __shfl_up(li[t], 1, 64);
                            if (lane < k) {
                                if (lane > p) { ld[t] = ud; li[t] = ui; }
                                else if (lane == p) { ld[t] = xd; li[t] = c0 + 64 * sub + bsel; }
                            }
                            tau[t] = __shfl(ld[t], k - 1, 64);
                        }
                    }
                }
            }
            if (pass == 0) {
#pragma unroll
                for (int t = 0; t < QPW; ++t) {
                    // rank of this lane's minimum among the 64 (ties by lane); the lane of rank k - 1 holds T
                    // (+inf when fewer than k lanes saw a point: no filtering then)
                    int rank = 0;
                    for (int l = 0; l < 64; ++l) {
                        const float o = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lmin[t]), l));
                        rank += (o < lmin[t] || (o == lmin[t] && l < lane)) ? 1 : 0;
                    }
                    const unsigned long long sel = __ballot(rank == k - 1);
                    T[t] = __shfl(lmin[t], __ffsll((long long)sel) - 1, 64);
                }
            }
        }
        if (active) {
#pragma unroll
            for (int t = 0; t < QPW; ++t) {
                if (i0 + t < N && lane < k) {
                    nbr[(int64_t)(i0 + t) * k + lane] = li[t];
                    dist[(int64_t)(i0 + t) * k + lane] = ld[t];
                    // the histogram of the lists' entries = the row lengths of the TRANSPOSED neighbour graph, whose CSR
                    // the callers build next (integer atomics: order-free)
                    if (counts && li[t] >= 0) atomicAdd(&counts[li[t]], 1);
                }
            }
        }
    }
}

}  // namespace

static int knn_launch(const float* pos, int64_t N, int32_t k, int32_t mode, int32_t* nbr, float* dist, int32_t* counts,
                      void* stream_) {
    if (N < 0 || k < 1 || k > 64 || (mode != 0 && mode != 1)) return EQH_ERR_ARG;
    if (N == 0) return EQH_OK;
    if (!pos || !nbr || !dist) return EQH_ERR_ARG;
    if (N >= ((int64_t)1 << 31) / 3) return EQH_ERR_RANGE;
    if ((mode == 0 && N < k) || (mode == 1 && N - 1 < k)) return EQH_ERR_ARG;  // torch.topk raises
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const bool big = N > 8192;
    const int grid = eqh_grid_for(big ? (N + 3) / 4 : N, 4, 256 * 8);
    if (mode == 0 && !big)
        hipLaunchKernelGGL((k_knn<0, 1>), dim3(grid), dim3(256), 0, stream, pos, (int)N, (int)k, nbr, dist, counts);
    else if (mode == 0)
        hipLaunchKernelGGL((k_knn<0, 4>), dim3(grid), dim3(256), 0, stream, pos, (int)N, (int)k, nbr, dist, counts);
    else if (!big)
        hipLaunchKernelGGL((k_knn<1, 1>), dim3(grid), dim3(256), 0, stream, pos, (int)N, (int)k, nbr, dist, counts);
    else
        hipLaunchKernelGGL((k_knn<1, 4>), dim3(grid), dim3(256), 0, stream, pos, (int)N, (int)k, nbr, dist, counts);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}

extern "C" int geo_knn(const float* pos, int64_t N, int32_t k, int32_t mode, int32_t* nbr,
                       float* dist, void* stream_) {
    return knn_launch(pos, N, k, mode, nbr, dist, nullptr, stream_);
}

/* geo_knn that also counts how often every point is listed: counts[j] += 1 per list entry j (counts [N] zeroed by the
   caller, e.g. by hg_index_aux): the row lengths hg_csr_build_i32_counted starts from. */
extern "C" int geo_knn_counted(const float* pos, int64_t N, int32_t k, int32_t mode, int32_t* nbr, float* dist,
                               int32_t* counts, void* stream_) {
    if (N > 0 && !counts) return EQH_ERR_ARG;
    return knn_launch(pos, N, k, mode, nbr, dist, counts, stream_);
}
