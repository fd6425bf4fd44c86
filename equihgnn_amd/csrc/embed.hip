// Embedding-sum forward / backward — ogb AtomEncoder (equihnn_egnn.py:121,157; mhnn.py:164,
// 201: nine tables summed in feature order) and the bond-type nn.Embedding(6, C)
// (mhnn.py:165,202).  The F tables are passed as ONE concatenated [table_rows, C] matrix plus
// per-feature row offsets.
//
// fwd: HBM/L2-bound row gather: F table rows (1 KiB each at C=256) summed per node in registers.
// bwd: d table[g,:] = sum over nodes whose feature value selects row g of d out[n,:].  A table
//      row can be selected by thousands of nodes (e.g. the 2-row "is aromatic" table), so the sum
//      is split over node chunks: pass 1 writes per-(chunk,row) partial sums, pass 2 adds the
//      partials in chunk order.  No atomics; bitwise reproducible.
#include "common.h"

namespace {

constexpr int THREADS = 256;
constexpr int MAX_F = 16;
constexpr int BWD_CHUNK = 512;  // nodes per partial sum

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
    const int g = blockIdx.x, chunk = blockIdx.y;
    int f = 0;
    for (int t = 1; t < F; ++t)
        if (g >= offs.off[t]) f = t;
    const int64_t value = g - offs.off[f];
    const int sub = threadIdx.x / LPR, sl = threadIdx.x % LPR;
    const int64_t n0 = (int64_t)chunk * BWD_CHUNK;
    const int64_t n1 = (n0 + BWD_CHUNK < N) ? n0 + BWD_CHUNK : N;
    for (int c = sl * 4; c < C; c += LPR * 4) {
        float4 acc = f4_zero();
        for (int64_t n = n0 + sub; n < n1; n += SUBS)
            if (x[n * F + f] == value) f4_add(acc, *reinterpret_cast<const float4*>(dout + n * C + c));
        s_acc[threadIdx.x] = acc;
        __syncthreads();
        if (sub == 0) {
            for (int s = 1; s < SUBS; ++s) f4_add(acc, s_acc[s * LPR + sl]);
            *reinterpret_cast<float4*>(part + ((int64_t)chunk * table_rows + g) * C + c) = acc;
        }
        __syncthreads();
    }
}

__global__ void k_embed_bwd_final(const float* __restrict__ part, int n_chunks, int64_t row_elems,
                                  float* __restrict__ dtable) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i * 4 < row_elems; i += stride) {
        float4 acc = f4_zero();
        for (int k = 0; k < n_chunks; ++k)
            f4_add(acc, *reinterpret_cast<const float4*>(part + (int64_t)k * row_elems + i * 4));
        *reinterpret_cast<float4*>(dtable + i * 4) = acc;
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
                                void* workspace, size_t workspace_bytes, void* stream_) {
    if (N < 0 || C <= 0 || table_rows <= 0 || !dtable) return EQH_ERR_ARG;
    Offsets o;
    int rc = fill_offsets(o, off_host, F, table_rows);
    if (rc) return rc;
    if ((C & 3) || !eqh_aligned16(dout) || !eqh_aligned16(dtable) || !eqh_aligned16(workspace))
        return EQH_ERR_ALIGN;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int64_t row_elems = table_rows * C;
    if (N == 0) {
        return eqh_zero_async(dtable, row_elems, stream);
    }
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
    hipLaunchKernelGGL(k_embed_bwd_final, dim3(eqh_grid_for(row_elems / 4, 256, 1024)), dim3(256), 0,
                       stream, part, n_chunks, row_elems, dtable);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}
