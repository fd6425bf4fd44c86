// Embedding-sum forward / backward — ogb AtomEncoder (equihnn_egnn.py:121,157; mhnn.py:164,
// 201: nine tables summed in feature order) and the bond-type nn.Embedding(6, C)
// (mhnn.py:165,202).  The F tables are passed as ONE concatenated [table_rows, C] matrix plus
// per-feature row offsets.
//
// fwd: HBM/L2-bound row gather: F table rows (1 KiB each at C=256) summed per node in registers.
// bwd: d table[g,:] = sum over nodes whose feature value selects row g of d out[n,:].  A table
//      row can be selected by thousands of nodes (e.g. the 2-row "is aromatic" table), so the sum
//      is split over chunks of 128 nodes: a workgroup (table row, chunk) first COMPACTS the ids of
//      its matching nodes (ballot + prefix: node order), then its four wavefronts add those rows
//      with four loads in flight each -- the first version tested node after node and loaded a row
//      only after the comparison (one dependent memory latency per node: 60 us) -- and the
//      per-chunk partial sums are added in chunk order by the common slab reducer (optionally
//      into the destination: the tables are parameters).  No atomics; bitwise reproducible.
#include "common.h"

namespace {

constexpr int THREADS = 256;
constexpr int MAX_F = 16;
constexpr int BWD_CHUNK = 128;  // nodes per partial sum (two wavefronts' worth of comparisons)

struct Offsets { int off[MAX_F + 1]; };

template <int LPR>
__global__ void __launch_bounds__(THREADS)
k_embed_fwd(const int64_t* __restrict__ x, const float* __restrict__ table, Offsets offs, int F,
            int64_t N, int C, float* __restrict__ out) {
    constexpr int ROWS_PER_BLOCK = THREADS / LPR;
    const int sub = threadIdx.x / LPR, sl = threadIdx.x % LPR;
    const int64_t stride = (int64_t)gridDim.x * ROWS_PER_BLOCK;
    for (int64_t n = (int64_t)blockIdx.x * ROWS_PER_BLOCK + sub; n < N; n += stride) {
        int64_t rows[MAX_F];
#pragma unroll
        for (int f = 0; f < MAX_F; ++f)
            if (f < F) rows[f] = offs.off[f] + x[n * F + f];
        for (int c = sl * 4; c < C; c += LPR * 4) {
            float4 acc = *reinterpret_cast<const float4*>(table + rows[0] * C + c);
#pragma unroll
            for (int f = 1; f < MAX_F; ++f)
                if (f < F) f4_add(acc, *reinterpret_cast<const float4*>(table + rows[f] * C + c));
            *reinterpret_cast<float4*>(out + n * C + c) = acc;
        }
    }
}

// grid = (table_rows, n_chunks); each block sums the matching d out rows of its node chunk.
template <int LPR>
__global__ void __launch_bounds__(THREADS)
k_embed_bwd_partial(const int64_t* __restrict__ x, const float* __restrict__ dout, Offsets offs,
                    int F, int64_t N, int C, int table_rows, float* __restrict__ part) {
    constexpr int SUBS = THREADS / LPR;
    __shared__ float4 s_acc[THREADS];
    __shared__ int s_list[BWD_CHUNK];
    __shared__ int s_cnt[2];
    const int g = blockIdx.x, chunk = blockIdx.y;
    int f = 0;
    for (int t = 1; t < F; ++t)
        if (g >= offs.off[t]) f = t;
    const int64_t value = g - offs.off[f];
    const int64_t n0 = (int64_t)chunk * BWD_CHUNK;
    // compaction: threads 0..127 (wavefronts 0, 1) test one node each; matching ids in node order
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bool hit = false;
    if (threadIdx.x < BWD_CHUNK) {
        const int64_t n = n0 + threadIdx.x;
        hit = n < N && x[n * F + f] == value;
    }
    const unsigned long long mask = __ballot(hit);
    if (wave < 2 && lane == 0) s_cnt[wave] = __popcll(mask);
    __syncthreads();
    if (hit) s_list[(wave == 1 ? s_cnt[0] : 0) + __popcll(mask & ((1ull << lane) - 1ull))] = threadIdx.x;
    __syncthreads();
    const int m = s_cnt[0] + s_cnt[1];
    const int sub = threadIdx.x / LPR, sl = threadIdx.x % LPR;
    for (int c = sl * 4; c < C; c += LPR * 4) {
        float4 a0 = f4_zero(), a1 = f4_zero(), a2 = f4_zero(), a3 = f4_zero();
        int i = sub;
        for (; i + 3 * SUBS < m; i += 4 * SUBS) {   // four independent row loads in flight
            const float4 v0 = *reinterpret_cast<const float4*>(dout + (n0 + s_list[i]) * C + c);
            const float4 v1 = *reinterpret_cast<const float4*>(dout + (n0 + s_list[i + SUBS]) * C + c);
            const float4 v2 = *reinterpret_cast<const float4*>(dout + (n0 + s_list[i + 2 * SUBS]) * C + c);
            const float4 v3 = *reinterpret_cast<const float4*>(dout + (n0 + s_list[i + 3 * SUBS]) * C + c);
            f4_add(a0, v0); f4_add(a1, v1); f4_add(a2, v2); f4_add(a3, v3);
        }
        for (; i < m; i += SUBS) f4_add(a0, *reinterpret_cast<const float4*>(dout + (n0 + s_list[i]) * C + c));
        f4_add(a0, a1);
        f4_add(a2, a3);
        f4_add(a0, a2);
        s_acc[threadIdx.x] = a0;
        __syncthreads();
        if (sub == 0) {
            for (int s = 1; s < SUBS; ++s) f4_add(a0, s_acc[s * LPR + sl]);
            *reinterpret_cast<float4*>(part + ((int64_t)chunk * table_rows + g) * C + c) = a0;
        }
        __syncthreads();
    }
}

template <typename L>
int dispatch_lpr(int C, L&& launch) {
    const int lanes = C / 4;
    if (lanes > 32) return launch(std::integral_constant<int, 64>{});
    if (lanes > 16) return launch(std::integral_constant<int, 32>{});
    if (lanes > 8) return launch(std::integral_constant<int, 16>{});
    return launch(std::integral_constant<int, 8>{});
}

int fill_offsets(Offsets& o, const int32_t* off_host, int F, int64_t table_rows) {
    if (!off_host || F < 1 || F > MAX_F) return EQH_ERR_ARG;
    for (int f = 0; f < F; ++f) {
        o.off[f] = off_host[f];
        if (off_host[f] < 0 || off_host[f] >= table_rows || (f > 0 && off_host[f] <= off_host[f - 1]))
            return EQH_ERR_ARG;
    }
    for (int f = F; f <= MAX_F; ++f) o.off[f] = (int)table_rows;
    return EQH_OK;
}

}  // namespace

extern "C" int hg_embed_sum_fwd(const int64_t* x, const float* table, const int32_t* off_host,
                                int32_t F, int64_t N, int32_t C, int64_t table_rows, float* out,
                                void* stream_) {
    if (N < 0 || C <= 0 || table_rows <= 0) return EQH_ERR_ARG;
    Offsets o;
    int rc = fill_offsets(o, off_host, F, table_rows);
    if (rc) return rc;
    if (N == 0) return EQH_OK;
    if (!x || !table || !out) return EQH_ERR_ARG;
    if ((C & 3) || !eqh_aligned16(table) || !eqh_aligned16(out)) return EQH_ERR_ALIGN;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    return dispatch_lpr(C, [&](auto lpr) {
        constexpr int LPR = decltype(lpr)::value;
        const int grid = eqh_grid_for(N, THREADS / LPR, 256 * 16);
        hipLaunchKernelGGL((k_embed_fwd<LPR>), dim3(grid), dim3(THREADS), 0, stream, x, table, o, F, N,
                           C, out);
        EQH_CHECK_LAUNCH();
        return EQH_OK;
    });
}

extern "C" size_t hg_embed_sum_bwd_workspace_bytes(int64_t N, int32_t C, int64_t table_rows) {
    if (N < 0 || C <= 0 || table_rows <= 0) return 0;
    const int64_t n_chunks = (N + BWD_CHUNK - 1) / BWD_CHUNK;
    return (size_t)(n_chunks > 0 ? n_chunks : 1) * (size_t)table_rows * (size_t)C * sizeof(float);
}

extern "C" int hg_embed_sum_bwd(const int64_t* x, const float* dout, const int32_t* off_host,
                                int32_t F, int64_t N, int32_t C, int64_t table_rows, float* dtable,
                                int32_t accumulate, void* workspace, size_t workspace_bytes, void* stream_) {
    if (N < 0 || C <= 0 || table_rows <= 0 || !dtable) return EQH_ERR_ARG;
    Offsets o;
    int rc = fill_offsets(o, off_host, F, table_rows);
    if (rc) return rc;
    if ((C & 3) || !eqh_aligned16(dout) || !eqh_aligned16(dtable) || !eqh_aligned16(workspace))
        return EQH_ERR_ALIGN;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int64_t row_elems = table_rows * C;
    if (N == 0) return accumulate ? EQH_OK : eqh_zero_async(dtable, row_elems, stream);
    if (!x || !dout || !workspace) return EQH_ERR_ARG;
    if (workspace_bytes < hg_embed_sum_bwd_workspace_bytes(N, C, table_rows)) return EQH_ERR_ARG;
    const int n_chunks = (int)((N + BWD_CHUNK - 1) / BWD_CHUNK);
    if (n_chunks > 65535) return EQH_ERR_RANGE;
    float* part = static_cast<float*>(workspace);
    rc = dispatch_lpr(C, [&](auto lpr) {
        constexpr int LPR = decltype(lpr)::value;
        hipLaunchKernelGGL((k_embed_bwd_partial<LPR>), dim3((unsigned)table_rows, (unsigned)n_chunks),
                           dim3(THREADS), 0, stream, x, dout, o, F, N, C, (int)table_rows, part);
        EQH_CHECK_LAUNCH();
        return EQH_OK;
    });
    if (rc) return rc;
    return eqh_reduce_slabs_async(part, n_chunks, row_elems, dtable, stream, accumulate);
}
