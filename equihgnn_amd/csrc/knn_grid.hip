// k nearest neighbours over the batch point cloud through a uniform cell grid: the same result, bit for bit,
// as the brute-force geo_knn (csrc/knn.hip) -- neighbours ordered by (distance, index), the reference's
// distance arithmetic -- at O(N) instead of O(N^2) distance evaluations.
//
// egnn_layer.py:253-288 (mode 0: squared distance, self included) and equiformer_layer.py:1216-1346 /
// fa_former_layer.py:651-668 (mode 1: true distance, self excluded) search the concatenated cloud of the whole
// batch, ~30 k atoms at the PCQM batch size: 9e8 pairs per step for 16 neighbours that all lie within ~2 A.
//
// Two launches:
//   k_grid_build (one workgroup): bounding box of the first n_box points (the real atoms: a padded batch parks
//     its padding atoms far away, they must not stretch the box), cubic cells, 1.5 * cbrt(n) of them (at most
//     32) along the longest axis; points outside the box are clamped into the boundary cells (clamping is 1-Lipschitz, so
//     "within r cells" still bounds "within r*h"); counting sort by cell in LDS -> cell_start[], sorted[] =
//     (x, y, z, original index).
//   k_knn_grid (one wavefront per query, queries in cell order): scan the 3 x 3 x 3 cells around the query,
//     then shells r = 2..8, each as row segments of the sorted array; stop as soon as the k-th distance is
//     below the r*h already covered (with a 1e-4 margin for the rounding of the cell coordinates), or the cube
//     covers the grid.  Queries outside the box (padding atoms) and the rare query still open after r = 8
//     restart as an exhaustive scan.  The running list lives in lanes 0..k-1, ordered by (distance, index); a
//     candidate enters if it is lexicographically smaller than the k-th entry, so the arrival order is free.
//
// Measured (MI355X, k = 16, mode 0; brute force = csrc/knn.hip): 4.6 k atoms 75 us against 50 us, 15 k atoms
// 243 / 244 us, 31 k atoms 470 / 520 us (build 45 us of it: one workgroup).  The batch's molecules are all
// centred at the origin, so even at 32 cells per axis the middle cells hold ~50 atoms and a query still offers
// >1000 candidates to a list whose insertion is ~25 wavefront instructions; the host wrapper therefore uses the
// grid only from ~24 k atoms.  Next steps if it is to matter: a two-level (per-molecule) grid, a multi-workgroup
// build, several queries per wavefront sharing the candidate loads as the brute-force kernel does.
#include <limits.h>
#include <math.h>

#include "common.h"

namespace {

constexpr int GB_THREADS = 1024;
constexpr int G_MAX = 32;                   // cells per axis
constexpr int MAX_CELLS = G_MAX * G_MAX * G_MAX;
constexpr int R_CAP = 8;                    // last shell before the exhaustive restart
constexpr float FINE = 1.5f;                // cells along the longest axis = FINE * cbrt(points), at most G_MAX

struct Grid {
    float lo[3];
    float inv_h, h;
    int g[3];
    int n_cells;
};

__device__ __forceinline__ int cell_axis(float p, float lo, float inv_h, int g) {
    const float t = __fmul_rn(__fsub_rn(p, lo), inv_h);
    int c = (int)floorf(t);
    if (!(t >= 0.f)) c = 0;       // also NaN
    return c > g - 1 ? g - 1 : c;
}

__global__ void __launch_bounds__(GB_THREADS)
k_grid_build(const float* __restrict__ pos, int N, const int* __restrict__ n_box_ptr, Grid* __restrict__ grid_out,
             int* __restrict__ cell_start, float4* __restrict__ sorted) {
    extern __shared__ int s_cnt[];                       // MAX_CELLS counters; first used as reduction scratch
    __shared__ Grid s_grid;
    __shared__ int s_scan[GB_THREADS];
    const int tid = threadIdx.x;
    int n_box = n_box_ptr ? *n_box_ptr : N;
    if (n_box < 1 || n_box > N) n_box = N;
    // ---- bounding box of the first n_box points
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = tid; i < n_box; i += GB_THREADS) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = pos[3 * i + a];
            mn[a] = fminf(mn[a], v);
            mx[a] = fmaxf(mx[a], v);
        }
    }
    float* s_f = reinterpret_cast<float*>(s_cnt);
#pragma unroll
    for (int a = 0; a < 3; ++a) { s_f[a * GB_THREADS + tid] = mn[a]; s_f[(3 + a) * GB_THREADS + tid] = mx[a]; }
    __syncthreads();
    for (int off = GB_THREADS / 2; off > 0; off >>= 1) {
        if (tid < off) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                s_f[a * GB_THREADS + tid] = fminf(s_f[a * GB_THREADS + tid], s_f[a * GB_THREADS + tid + off]);
                s_f[(3 + a) * GB_THREADS + tid] = fmaxf(s_f[(3 + a) * GB_THREADS + tid], s_f[(3 + a) * GB_THREADS + tid + off]);
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        Grid g;
        float ext[3];
        float vol = 1.f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            g.lo[a] = s_f[a * GB_THREADS];
            ext[a] = fmaxf(s_f[(3 + a) * GB_THREADS] - g.lo[a], 1e-3f);
            vol *= ext[a];
        }
        // The molecules of a batch are all centred at the origin, so the cloud is sharply peaked (tens of atoms
        // per cubic Angstrom in the middle, where most queries live): cells sized for the MEAN density would put
        // hundreds of candidates into the 27 central cells.  Take the finest grid the counters allow along the
        // longest axis instead; the sparse rim pays with more (mostly empty) shells.
        const float ext_max = fmaxf(ext[0], fmaxf(ext[1], ext[2]));
        int g_target = (int)(FINE * cbrtf((float)n_box) + 0.5f);
        g_target = g_target < 4 ? 4 : (g_target > G_MAX ? G_MAX : g_target);
        const float h = ext_max / (float)g_target;
        (void)vol;
        g.h = h;
        g.inv_h = 1.0f / h;
        g.n_cells = 1;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            int c = (int)ceilf(ext[a] * g.inv_h) + 1;    // + 1: the maximum itself falls into a cell of its own
            c = c < 1 ? 1 : (c > G_MAX ? G_MAX : c);
            g.g[a] = c;
            g.n_cells *= c;
        }
        s_grid = g;
        *grid_out = g;
    }
    __syncthreads();
    const Grid g = s_grid;
    // ---- counting sort by cell
    for (int c = tid; c < g.n_cells; c += GB_THREADS) s_cnt[c] = 0;
    __syncthreads();
    for (int i = tid; i < N; i += GB_THREADS) {
        const int cx = cell_axis(pos[3 * i], g.lo[0], g.inv_h, g.g[0]);
        const int cy = cell_axis(pos[3 * i + 1], g.lo[1], g.inv_h, g.g[1]);
        const int cz = cell_axis(pos[3 * i + 2], g.lo[2], g.inv_h, g.g[2]);
        atomicAdd(&s_cnt[(cz * g.g[1] + cy) * g.g[0] + cx], 1);
    }
    __syncthreads();
    // exclusive scan: each thread owns a contiguous slice of the counters
    const int per = (g.n_cells + GB_THREADS - 1) / GB_THREADS;
    const int c0 = tid * per, c1 = (c0 + per < g.n_cells) ? c0 + per : g.n_cells;
    int local = 0;
    for (int c = c0; c < c1; ++c) local += s_cnt[c];
    s_scan[tid] = local;
    __syncthreads();
    for (int off = 1; off < GB_THREADS; off <<= 1) {
        const int v = tid >= off ? s_scan[tid - off] : 0;
        __syncthreads();
        s_scan[tid] += v;
        __syncthreads();
    }
    int run = s_scan[tid] - local;
    for (int c = c0; c < c1; ++c) {
        const int n = s_cnt[c];
        cell_start[c] = run;
        s_cnt[c] = run;                                  // becomes the fill cursor
        run += n;
    }
    if (tid == GB_THREADS - 1) cell_start[g.n_cells] = N;
    __syncthreads();
    for (int i = tid; i < N; i += GB_THREADS) {
        const float x = pos[3 * i], y = pos[3 * i + 1], z = pos[3 * i + 2];
        const int cx = cell_axis(x, g.lo[0], g.inv_h, g.g[0]);
        const int cy = cell_axis(y, g.lo[1], g.inv_h, g.g[1]);
        const int cz = cell_axis(z, g.lo[2], g.inv_h, g.g[2]);
        const int at = atomicAdd(&s_cnt[(cz * g.g[1] + cy) * g.g[0] + cx], 1);
        sorted[at] = make_float4(x, y, z, __int_as_float(i));
    }
}

// the running k-best list of one query: lanes 0..k-1 hold (distance, index) in lexicographic order
struct Best {
    float d;      // lanes >= k: +inf, never take part
    int i;
    float tau;    // the k-th entry, broadcast
    int tau_i;
};

__device__ __forceinline__ void best_reset(Best& b) {
    b.d = INFINITY; b.i = INT_MAX; b.tau = INFINITY; b.tau_i = INT_MAX;
}

// offer the candidates sorted[s .. e) to the list of query (qx, qy, qz, qi)
// smallest distance among this lane's candidates of sorted[s .. e) (first pass: see k_knn_grid)
template <int MODE>
__device__ __forceinline__ void min_segment(const float4* __restrict__ sorted, int s, int e, float qx, float qy,
                                            float qz, int qi, int lane, float& lmin) {
    for (int c0 = s; c0 < e; c0 += 64) {
        const int j = c0 + lane;
        bool valid = j < e;
        const float4 p = sorted[valid ? j : s];
        if (MODE == 1) valid = valid && (__float_as_int(p.w) != qi);
        const float dx = __fsub_rn(qx, p.x), dy = __fsub_rn(qy, p.y), dz = __fsub_rn(qz, p.z);
        float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        if (MODE == 1) d = __fsqrt_rn(d);
        if (valid && d < lmin) lmin = d;
    }
}

template <int MODE>
__device__ __forceinline__ void scan_segment(const float4* __restrict__ sorted, int s, int e, float qx, float qy,
                                             float qz, int qi, int k, int lane, Best& b, const float T = INFINITY) {
    for (int c0 = s; c0 < e; c0 += 64) {
        const int j = c0 + lane;
        bool valid = j < e;
        const float4 p = sorted[valid ? j : s];
        const int pi = __float_as_int(p.w);
        if (MODE == 1) valid = valid && (pi != qi);
        const float dx = __fsub_rn(qx, p.x), dy = __fsub_rn(qy, p.y), dz = __fsub_rn(qz, p.z);
        float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        if (MODE == 1) d = __fsqrt_rn(d);
        unsigned long long mask = __ballot(valid && d <= T && (d < b.tau || (d == b.tau && pi < b.tau_i)));
        while (mask) {
            const int sel = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const float xd = __shfl(d, sel, 64);
            const int xi = __shfl(pi, sel, 64);
            if (!(xd < b.tau || (xd == b.tau && xi < b.tau_i))) continue;
            const int at = __popcll(__ballot(lane < k && (b.d < xd || (b.d == xd && b.i < xi))));
            const float ud = __shfl_up(b.d, 1, 64);
            const int ui = __shfl_up(b.i, 1, 64);
            if (lane < k) {
                if (lane > at) { b.d = ud; b.i = ui; }
                else if (lane == at) { b.d = xd; b.i = xi; }
            }
            b.tau = __shfl(b.d, k - 1, 64);
            b.tau_i = __shfl(b.i, k - 1, 64);
        }
    }
}

template <int MODE>
__global__ void __launch_bounds__(256)
k_knn_grid(const float4* __restrict__ sorted, const int* __restrict__ cell_start, const Grid* __restrict__ grid_in,
           int N, int k, int* __restrict__ nbr, float* __restrict__ dist) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= N) return;
    const Grid g = *grid_in;
    const float4 qp = sorted[q];
    const float qx = qp.x, qy = qp.y, qz = qp.z;
    const int qi = __float_as_int(qp.w);
    int cc[3];
    bool inside = true;
    {
        const float pv[3] = {qx, qy, qz};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            cc[a] = cell_axis(pv[a], g.lo[a], g.inv_h, g.g[a]);
            const float t = __fmul_rn(__fsub_rn(pv[a], g.lo[a]), g.inv_h);
            inside = inside && t >= 0.f && t < (float)g.g[a];
        }
    }
    const int gx = g.g[0], gy = g.g[1], gz = g.g[2];
    Best b;
    best_reset(b);
    bool done = false;
    // slots of shell r: two per (dz, dy) row of the cube; a row on the cube's boundary (or every row at r = 1, where the
    // whole 3 x 3 x 3 cube is new) is one full x-run, an inner row the two end cells
    auto slot_segment = [&](int r, int slot, int& seg_s, int& seg_e) {
        const int w = 2 * r + 1;
        seg_s = seg_e = 0;
        if (slot < 2 * w * w) {
            const int row = slot >> 1, second = slot & 1;
            const int dz = row / w - r, dy = row % w - r;
            const int z = cc[2] + dz, y = cc[1] + dy;
            if (z >= 0 && z < gz && y >= 0 && y < gy) {
                const bool full = r == 1 || dz == -r || dz == r || dy == -r || dy == r;
                int x0, x1;
                if (full) { x0 = cc[0] - r; x1 = second ? x0 - 1 : cc[0] + r; }
                else { x0 = second ? cc[0] + r : cc[0] - r; x1 = x0; }
                if (x0 < 0 && x1 >= 0 && full) x0 = 0;
                if (x1 > gx - 1 && x0 <= gx - 1 && full) x1 = gx - 1;
                if (x0 >= 0 && x1 <= gx - 1 && x0 <= x1) {
                    const int base = (z * gy + y) * gx;
                    seg_s = cell_start[base + x0];
                    seg_e = cell_start[base + x1 + 1];
                }
            }
        }
    };
    // First pass over the 3 x 3 x 3 cube (as in k_knn): every lane keeps the smallest distance among ITS candidates; the
    // k-th smallest of the 64 lane minima, T, bounds the final k-th distance from above, so the list below is only offered
    // candidates with d <= T -- about k + a few serial insertions per query instead of ~k ln(n / k) + k of the several
    // hundred atoms the central cells of a batch hold.  Fewer than k lanes with a candidate: T = +inf, no filtering.
    float T = INFINITY;
    if (inside) {
        float lmin = INFINITY;
        int seg_s, seg_e;
        slot_segment(1, lane, seg_s, seg_e);               // 18 slots
        unsigned long long live = __ballot(seg_e > seg_s);
        while (live) {
            const int sl = __ffsll((long long)live) - 1;
            live &= live - 1;
            min_segment<MODE>(sorted, __shfl(seg_s, sl, 64), __shfl(seg_e, sl, 64), qx, qy, qz, qi, lane, lmin);
        }
        int rank = 0;
        for (int l = 0; l < 64; ++l) {
            const float o = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lmin), l));
            rank += (o < lmin || (o == lmin && l < lane)) ? 1 : 0;
        }
        const unsigned long long sel = __ballot(rank == k - 1);
        T = __shfl(lmin, __ffsll((long long)sel) - 1, 64);
    }
    if (inside) {
        for (int r = 1; r <= R_CAP && !done; ++r) {
            const int w = 2 * r + 1;
            const int n_slots = 2 * w * w;
            for (int s0 = 0; s0 < n_slots; s0 += 64) {
                int seg_s, seg_e;
                slot_segment(r, s0 + lane, seg_s, seg_e);
                unsigned long long live = __ballot(seg_e > seg_s);
                while (live) {
                    const int sl = __ffsll((long long)live) - 1;
                    live &= live - 1;
                    scan_segment<MODE>(sorted, __shfl(seg_s, sl, 64), __shfl(seg_e, sl, 64), qx, qy, qz, qi, k, lane, b, T);
                }
            }
            const float covered = (float)r * g.h * 0.9999f;
            const float reach = MODE == 1 ? covered : covered * covered;
            const bool whole = cc[0] - r <= 0 && cc[0] + r >= gx - 1 && cc[1] - r <= 0 && cc[1] + r >= gy - 1 &&
                               cc[2] - r <= 0 && cc[2] + r >= gz - 1;
            done = whole || b.tau < reach;
        }
    }
    if (!done) {   // outside the box, or still open after the last shell: exhaustive, from scratch
        best_reset(b);
        scan_segment<MODE>(sorted, 0, N, qx, qy, qz, qi, k, lane, b);
    }
    if (lane < k) {
        nbr[(int64_t)qi * k + lane] = b.i;
        dist[(int64_t)qi * k + lane] = b.d;
    }
}

struct GridPlan {
    size_t off_sorted, off_cells, off_grid, total;   // bytes
};
GridPlan grid_plan(int64_t N) {
    GridPlan p;
    p.off_sorted = 0;
    p.off_cells = ((size_t)N * sizeof(float4) + 255) & ~(size_t)255;
    p.off_grid = p.off_cells + ((((size_t)MAX_CELLS + 1) * sizeof(int) + 255) & ~(size_t)255);
    p.total = p.off_grid + 256;
    return p;
}

}  // namespace

extern "C" int64_t geo_knn_grid_max_points(void) { return (int64_t)GB_THREADS * 64; }

extern "C" size_t geo_knn_grid_workspace_bytes(int64_t N) {
    if (N <= 0 || N > geo_knn_grid_max_points()) return 0;
    return grid_plan(N).total;
}

extern "C" int geo_knn_grid(const float* pos, int64_t N, int32_t k, int32_t mode, const int32_t* n_box, int32_t* nbr,
                            float* dist, void* workspace, size_t workspace_bytes, void* stream_) {
    if (N < 0 || k < 1 || k > 64 || (mode != 0 && mode != 1)) return EQH_ERR_ARG;
    if (N == 0) return EQH_OK;
    if (N > geo_knn_grid_max_points()) return EQH_ERR_RANGE;
    if (!pos || !nbr || !dist || !workspace) return EQH_ERR_ARG;
    if ((mode == 0 && N < k) || (mode == 1 && N - 1 < k)) return EQH_ERR_ARG;  // torch.topk raises
    if (!eqh_aligned16(workspace)) return EQH_ERR_ALIGN;
    const GridPlan p = grid_plan(N);
    if (workspace_bytes < p.total) return EQH_ERR_ARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    char* ws = static_cast<char*>(workspace);
    float4* sorted = reinterpret_cast<float4*>(ws + p.off_sorted);
    int* cells = reinterpret_cast<int*>(ws + p.off_cells);
    Grid* grid = reinterpret_cast<Grid*>(ws + p.off_grid);
    constexpr size_t lds = (size_t)MAX_CELLS * sizeof(int);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_grid_build), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return EQH_ERR_LAUNCH;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_grid_build, dim3(1), dim3(GB_THREADS), lds, stream, pos, (int)N, n_box, grid, cells, sorted);
    EQH_CHECK_LAUNCH();
    const int blocks = (int)((N + 3) / 4);
    if (mode == 0)
        hipLaunchKernelGGL((k_knn_grid<0>), dim3(blocks), dim3(256), 0, stream, sorted, cells, grid, (int)N, (int)k, nbr,
                           dist);
    else
        hipLaunchKernelGGL((k_knn_grid<1>), dim3(blocks), dim3(256), 0, stream, sorted, cells, grid, (int)N, (int)k, nbr,
                           dist);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}
