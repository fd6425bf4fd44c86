// k nearest neighbours over the whole batch point cloud (brute force, never materialises N x N).
//
// Replaces the dense rel_coors[1,N,N,3] / rel_dist[1,N,N] build + torch.topk of
// egnn_layer.py:253-288 (mode 0: squared distance, self included) and
// equiformer_layer.py:1216-1346 (mode 1: true distance, self excluded).  As in the reference the
// search runs over the concatenated cloud of the whole batch: neighbours may belong to other
// molecules.
//
// One 64-lane wavefront owns one query.  Candidates are streamed 64 at a time (one per lane);
// the running k-best list lives in lanes 0..k-1 of two registers (distance, index), sorted by
// (distance, index).  A candidate beats the list only if it is strictly closer than the current
// k-th entry; insertion is O(1) wave operations (ballot -> popcount -> shuffle-up), and happens
// ~k*ln(N/k) times per query, so the loop is dominated by the distance evaluations.
// Distance arithmetic: (dx*dx + dy*dy) + dz*dz with separately rounded products (no FMA).
#include <limits.h>
#include <math.h>

#include "common.h"

namespace {

template <int MODE>
__global__ void __launch_bounds__(256)
k_knn(const float* __restrict__ pos, int N, int k, int* __restrict__ nbr, float* __restrict__ dist) {
    const int lane = threadIdx.x & 63;
    const int waves_per_block = blockDim.x >> 6;
    const int wave0 = blockIdx.x * waves_per_block + (threadIdx.x >> 6);
    const int wave_stride = gridDim.x * waves_per_block;
    for (int i = wave0; i < N; i += wave_stride) {
        const float qx = pos[3 * i], qy = pos[3 * i + 1], qz = pos[3 * i + 2];
        float ld = INFINITY;  // list: distance (lanes >= k stay +inf and never take part)
        int li = -1;          //       index
        float tau = INFINITY;
        for (int c0 = 0; c0 < N; c0 += 64) {
            const int j = c0 + lane;
            bool valid = j < N;
            if (MODE == 1) valid = valid && (j != i);
            float d = INFINITY;
            if (valid) {
                const float dx = __fsub_rn(qx, pos[3 * j]);
                const float dy = __fsub_rn(qy, pos[3 * j + 1]);
                const float dz = __fsub_rn(qz, pos[3 * j + 2]);
                d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                if (MODE == 1) d = __fsqrt_rn(d);
            }
            unsigned long long mask = __ballot(valid && d < tau);
            while (mask) {
                const int b = __ffsll((long long)mask) - 1;  // lowest lane = lowest index first
                mask &= mask - 1;
                const float xd = __shfl(d, b, 64);
                if (!(xd < tau)) continue;
                const int p = __popcll(__ballot(lane < k && ld <= xd));
                const float ud = __shfl_up(ld, 1, 64);
                const int ui = __shfl_up(li, 1, 64);
                if (lane < k) {
                    if (lane > p) { ld = ud; li = ui; }
                    else if (lane == p) { ld = xd; li = c0 + b; }
                }
                tau = __shfl(ld, k - 1, 64);
            }
        }
        if (lane < k) {
            nbr[(int64_t)i * k + lane] = li;
            dist[(int64_t)i * k + lane] = ld;
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
    const int grid = eqh_grid_for(N, 4, 256 * 8);
    if (mode == 0)
        hipLaunchKernelGGL((k_knn<0>), dim3(grid), dim3(256), 0, stream, pos, (int)N, (int)k, nbr, dist);
    else
        hipLaunchKernelGGL((k_knn<1>), dim3(grid), dim3(256), 0, stream, pos, (int)N, (int)k, nbr, dist);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}
