// Segmented row reduction / row gather over a CSR — the node<->hyperedge aggregation kernel.
//
// Replaces torch_scatter.scatter(src, index, dim=-2, reduce="sum"|"mean") (conv.py:91-93,97,
// 173,177), the advanced-index gathers X[..., idx, :] (conv.py:90,96,172,175,176), their
// backward passes, and global_add_pool (equihnn_egnn.py:167, mhnn.py:216).
//
// HBM-bound.  Layout: a row of C fp32 channels is read as float4 per lane — at C=256 one
// 64-lane wavefront covers exactly one 1 KiB row per load instruction (fully coalesced); for
// narrower rows a wavefront is split into 64/LPR sub-groups that each own one output row.
// The per-row sum lives in registers; nothing is atomic; the output row is written once.
// Up to 4 source rows are in flight per sub-group to cover HBM/L2 latency; occupancy
// (<= 40 VGPRs) supplies the rest.  Algorithmic bytes per call:
//   4*C*nnz (gathered rows) + 4*nnz (col idx) + 4*(R+1) (rowptr) + 4*C*R (output).
#include "common.h"

namespace {

constexpr int THREADS = 256;

__device__ __forceinline__ float src_weight(const int* __restrict__ wptr, int j) {
    if (!wptr) return 1.0f;
    if (j < 0) return 0.0f;
    int d = wptr[j + 1] - wptr[j];
    return 1.0f / (float)(d > 1 ? d : 1);
}

// w[q] = 1 / max(deg(idx[q]), 1) for every entry q of a CSR (deg from ANOTHER CSR's rowptr): computed once per batch, the
// per-entry form of the mean weights -- the weighted reduce then loads idx[q] and w[q] side by side instead of
// chasing idx[q] -> wptr[j], wptr[j + 1] (two dependent loads per gathered row: 7.9 us against 5.3 us in the step)
__global__ void k_entry_weights(const int* __restrict__ idx, const int* __restrict__ wptr, int64_t nnz,
                                float* __restrict__ w) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nnz; q += stride) w[q] = src_weight(wptr, idx[q]);
}

// A NEGATIVE source index is a null entry (the padded incidences of a static-shape batch, batch.pad_batch): it
// contributes a zero row.  In the gather form (rowptr == NULL) that makes the backward of a per-incidence scatter
// hand null incidences a zero gradient instead of row 0's.
__device__ __forceinline__ float4 load_row(const float* __restrict__ src, int j, int C, int c) {
    const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)(j < 0 ? 0 : j) * C + c);
    return j < 0 ? f4_zero() : v;
}

// LPR = lanes per row (power of two, <= 64).  Column blocks of LPR*4 floats.
template <int LPR, bool WEIGHTED>
__global__ void __launch_bounds__(THREADS)
k_segment_reduce(const float* __restrict__ src, const int* __restrict__ idx,
                 const int* __restrict__ rowptr, const int* __restrict__ wptr, const float* __restrict__ ew,
                 float* __restrict__ out, int64_t n_rows, int C, int mean) {
    constexpr int ROWS_PER_BLOCK = THREADS / LPR;
    const int sub = threadIdx.x / LPR;
    const int sl = threadIdx.x % LPR;
    const int64_t row_stride = (int64_t)gridDim.x * ROWS_PER_BLOCK;
    for (int64_t r = (int64_t)blockIdx.x * ROWS_PER_BLOCK + sub; r < n_rows; r += row_stride) {
        int beg, end;
        if (rowptr) { beg = rowptr[r]; end = rowptr[r + 1]; }
        else { beg = (int)r; end = (int)r + 1; }
        const int deg = end - beg;
        const float denom = (mean && deg > 1) ? (float)deg : 1.0f;  // sum / clamp(count, 1)
        for (int c = sl * 4; c < C; c += LPR * 4) {
            float4 acc = f4_zero();
            int q = beg;
            for (; q + 4 <= end; q += 4) {
                int j0, j1, j2, j3;
                if (idx) { j0 = idx[q]; j1 = idx[q + 1]; j2 = idx[q + 2]; j3 = idx[q + 3]; }
                else { j0 = q; j1 = q + 1; j2 = q + 2; j3 = q + 3; }
                const float4 v0 = load_row(src, j0, C, c), v1 = load_row(src, j1, C, c);
                const float4 v2 = load_row(src, j2, C, c), v3 = load_row(src, j3, C, c);
                if (WEIGHTED) {
                    if (ew) {   // per-entry weights (null entries carry weight 0 and a zero row)
                        f4_fma(acc, v0, ew[q]); f4_fma(acc, v1, ew[q + 1]); f4_fma(acc, v2, ew[q + 2]); f4_fma(acc, v3, ew[q + 3]);
                    } else {
                        f4_fma(acc, v0, src_weight(wptr, j0)); f4_fma(acc, v1, src_weight(wptr, j1));
                        f4_fma(acc, v2, src_weight(wptr, j2)); f4_fma(acc, v3, src_weight(wptr, j3));
                    }
                } else {
                    f4_add(acc, v0); f4_add(acc, v1); f4_add(acc, v2); f4_add(acc, v3);
                }
            }
            for (; q < end; ++q) {
                const int j = idx ? idx[q] : q;
                const float4 v = load_row(src, j, C, c);
                if (WEIGHTED) f4_fma(acc, v, ew ? ew[q] : src_weight(wptr, j));
                else f4_add(acc, v);
            }
            acc.x /= denom; acc.y /= denom; acc.z /= denom; acc.w /= denom;
            *reinterpret_cast<float4*>(out + r * C + c) = acc;
        }
    }
}

template <int LPR>
int launch(const float* src, const int* idx, const int* rowptr, const int* wptr, const float* ew, float* out,
           int64_t n_rows, int C, int mean, hipStream_t stream) {
    constexpr int ROWS_PER_BLOCK = THREADS / LPR;
    // enough workgroups to fill 256 CUs x 8 blocks, grid-stride beyond that
    const int grid = eqh_grid_for(n_rows, ROWS_PER_BLOCK, 256 * 16);
    if (wptr || ew)
        hipLaunchKernelGGL((k_segment_reduce<LPR, true>), dim3(grid), dim3(THREADS), 0, stream, src,
                           idx, rowptr, wptr, ew, out, n_rows, C, mean);
    else
        hipLaunchKernelGGL((k_segment_reduce<LPR, false>), dim3(grid), dim3(THREADS), 0, stream, src,
                           idx, rowptr, wptr, ew, out, n_rows, C, mean);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}

}  // namespace

static int segment_reduce_any(const float* src, const int32_t* idx, const int32_t* rowptr, const int32_t* src_wptr,
                              const float* entry_w, float* out, int64_t n_out_rows, int32_t C, int32_t mean,
                              void* stream_) {
    if (n_out_rows < 0 || C <= 0) return EQH_ERR_ARG;
    if (n_out_rows == 0) return EQH_OK;
    if (!src || !out) return EQH_ERR_ARG;
    if (!rowptr && !idx) return EQH_ERR_ARG;  // a gather needs an index
    if ((C & 3) || !eqh_aligned16(src) || !eqh_aligned16(out)) return EQH_ERR_ALIGN;
    if (n_out_rows >= ((int64_t)1 << 31) - 1) return EQH_ERR_RANGE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int lanes = C / 4;
    if (lanes > 32) return launch<64>(src, idx, rowptr, src_wptr, entry_w, out, n_out_rows, C, mean, stream);
    if (lanes > 16) return launch<32>(src, idx, rowptr, src_wptr, entry_w, out, n_out_rows, C, mean, stream);
    if (lanes > 8) return launch<16>(src, idx, rowptr, src_wptr, entry_w, out, n_out_rows, C, mean, stream);
    if (lanes > 4) return launch<8>(src, idx, rowptr, src_wptr, entry_w, out, n_out_rows, C, mean, stream);
    return launch<4>(src, idx, rowptr, src_wptr, entry_w, out, n_out_rows, C, mean, stream);
}

extern "C" int hg_segment_reduce_f32(const float* src, const int32_t* idx, const int32_t* rowptr,
                                     const int32_t* src_wptr, float* out, int64_t n_out_rows,
                                     int32_t C, int32_t mean, void* stream_) {
    return segment_reduce_any(src, idx, rowptr, src_wptr, nullptr, out, n_out_rows, C, mean, stream_);
}

/* the same with per-ENTRY weights w[q] (hg_entry_weights) instead of per-source weights looked up through a rowptr */
extern "C" int hg_segment_reduce_w_f32(const float* src, const int32_t* idx, const int32_t* rowptr, const float* entry_w,
                                       float* out, int64_t n_out_rows, int32_t C, void* stream_) {
    if (!entry_w || !rowptr || !idx) return EQH_ERR_ARG;
    return segment_reduce_any(src, idx, rowptr, nullptr, entry_w, out, n_out_rows, C, 0, stream_);
}

extern "C" int hg_entry_weights(const int32_t* idx, const int32_t* wptr, int64_t nnz, float* w, void* stream_) {
    if (nnz < 0) return EQH_ERR_ARG;
    if (nnz == 0) return EQH_OK;
    if (!idx || !wptr || !w) return EQH_ERR_ARG;
    hipLaunchKernelGGL(k_entry_weights, dim3(eqh_grid_for(nnz, 256, 1024)), dim3(256), 0, static_cast<hipStream_t>(stream_),
                       idx, wptr, nnz, w);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}
