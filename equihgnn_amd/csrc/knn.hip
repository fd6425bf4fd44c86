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
// k-th entry; insertion is O(1) wave operations (ballot -> popcount -> shuffle-up), and happens
// ~k*ln(N/k) times per query, so the loop is dominated by the distance evaluations.
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
k_knn(const float* __restrict__ pos, int N, int k, int* __restrict__ nbr, float* __restrict__ dist) {
    const int lane = threadIdx.x & 63;
    const int waves_per_block = blockDim.x >> 6;
    const int wave0 = blockIdx.x * waves_per_block + (threadIdx.x >> 6);
    const int wave_stride = gridDim.x * waves_per_block;
    const int n_groups = (N + QPW - 1) / QPW;
    // the LAST queries first: a padded batch parks its padding atoms at the end of the cloud and far away from it,
    // where every real atom is a near-tie candidate (many list insertions); started last, those few long-running
    // wavefronts were the tail of the launch
    for (int g_ = wave0; g_ < n_groups; g_ += wave_stride) {
        const int grp = n_groups - 1 - g_;
        const int i0 = grp * QPW;
        float qx[QPW], qy[QPW], qz[QPW], ld[QPW], tau[QPW];
        int li[QPW];
#pragma unroll
        for (int t = 0; t < QPW; ++t) {
            const int i = (i0 + t < N) ? i0 + t : N - 1;  // tail queries repeat the last one (not stored)
            qx[t] = pos[3 * i]; qy[t] = pos[3 * i + 1]; qz[t] = pos[3 * i + 2];
            ld[t] = INFINITY;   // list: distance (lanes >= k stay +inf and never take part)
            li[t] = -1;         //       index
            tau[t] = INFINITY;
        }
        for (int c0 = 0; c0 < N; c0 += 64) {
            const int j = c0 + lane;
            const bool in_range = j < N;
            float px = 0.f, py = 0.f, pz = 0.f;
            if (in_range) { px = pos[3 * j]; py = pos[3 * j + 1]; pz = pos[3 * j + 2]; }
#pragma unroll
            for (int t = 0; t < QPW; ++t) {
                bool valid = in_range;
                if (MODE == 1) valid = valid && (j != i0 + t);
                const float dx = __fsub_rn(qx[t], px);
                const float dy = __fsub_rn(qy[t], py);
                const float dz = __fsub_rn(qz[t], pz);
                float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                if (MODE == 1) d = __fsqrt_rn(d);
                unsigned long long mask = __ballot(valid && d < tau[t]);
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
                        else if (lane == p) { ld[t] = xd; li[t] = c0 + bsel; }
                    }
                    tau[t] = __shfl(ld[t], k - 1, 64);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < QPW; ++t) {
            if (i0 + t < N && lane < k) {
                nbr[(int64_t)(i0 + t) * k + lane] = li[t];
                dist[(int64_t)(i0 + t) * k + lane] = ld[t];
            }
        }
    }
}

}  // namespace

extern "C" int geo_knn(const float* pos, int64_t N, int32_t k, int32_t mode, int32_t* nbr,
                       float* dist, void* stream_) {
    if (N < 0 || k < 1 || k > 64 || (mode != 0 && mode != 1)) return EQH_ERR_ARG;
    if (N == 0) return EQH_OK;
    if (!pos || !nbr || !dist) return EQH_ERR_ARG;
    if (N >= ((int64_t)1 << 31) / 3) return EQH_ERR_RANGE;
    if ((mode == 0 && N < k) || (mode == 1 && N - 1 < k)) return EQH_ERR_ARG;  // torch.topk raises
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const bool big = N > 8192;
    const int grid = eqh_grid_for(big ? (N + 3) / 4 : N, 4, 256 * 8);
    if (mode == 0 && !big)
        hipLaunchKernelGGL((k_knn<0, 1>), dim3(grid), dim3(256), 0, stream, pos, (int)N, (int)k, nbr, dist);
    else if (mode == 0)
        hipLaunchKernelGGL((k_knn<0, 4>), dim3(grid), dim3(256), 0, stream, pos, (int)N, (int)k, nbr, dist);
    else if (!big)
        hipLaunchKernelGGL((k_knn<1, 1>), dim3(grid), dim3(256), 0, stream, pos, (int)N, (int)k, nbr, dist);
    else
        hipLaunchKernelGGL((k_knn<1, 4>), dim3(grid), dim3(256), 0, stream, pos, (int)N, (int)k, nbr, dist);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}
