// Incidence COO -> CSR (counting sort on the device, stable inside a row).
//
// Replaces the "unsorted int64 index + atomicAdd" contract of torch_scatter.scatter
// (reference call sites conv.py:91-93,97,173,177): done once per batch, after which every
// node<->hyperedge aggregation is an atomic-free segmented reduction over rowptr/col.
//
// Pipeline (all on `stream`, no host sync):
//   k_clear -> k_hist (int atomics) -> 3-kernel exclusive scan -> k_fill (int atomics into a
//   scratch permutation) -> k_sort_rows (one wavefront per row: rank sort by shuffles) and
//   k_sort_long (rows > 2048 entries: one workgroup per row, bitonic sort in LDS).  Sorting each row by entry
//   id makes the summation order, and therefore every result, bitwise reproducible.
#include "common.h"

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;  // 2048 counters per block
constexpr int SHORT_ROW = 64;  // sizing of the long-row list: rows longer than this go on it
constexpr int LONG_THREADS = 256;
constexpr int LONG_LDS_CAP = 16384;  // ints (64 KiB)

// counters are cleared by a kernel, not hipMemsetAsync: memset nodes of a captured hipGraph were
// observed (ROCm 7.2) not to re-clear the 4-byte long-row counter on replay, which overflows
// long_rows after a few replays.
__global__ void k_clear(int* __restrict__ cnt, int64_t n_items, int* __restrict__ long_count) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_items; i += stride) cnt[i] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) *long_count = 0;
}

template <typename KeyT>
__global__ void k_hist(const KeyT* __restrict__ key, int64_t nnz, int64_t n_rows,
                       int* __restrict__ cnt) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < nnz; p += stride) {
        int64_t k = key[p];
        if (k >= 0 && k < n_rows) atomicAdd(&cnt[k], 1);
    }
}

// block-wide exclusive scan of per-thread totals; returns the exclusive prefix of this thread
// and (in *block_total) the sum over the block.
__device__ int block_exclusive_scan(int v, int* block_total) {
    __shared__ int wave_sum[SCAN_THREADS / EQH_WAVE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    if (lane == 63) wave_sum[wave] = inc;
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / EQH_WAVE; ++w) {
        int s = wave_sum[w];
        if (w < wave) base += s;
        total += s;
    }
    __syncthreads();
    *block_total = total;
    return base + inc - v;
}

__global__ void k_scan_block_sums(const int* __restrict__ cnt, int64_t n_items,
                                  int* __restrict__ block_sums) {
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i)
        if (base + i < n_items) s += cnt[base + i];
    int total;
    block_exclusive_scan(s, &total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// one block, any number of block sums: sequential chunks with a running carry
__global__ void k_scan_carry(int* __restrict__ block_sums, int n_blocks) {
    int carry = 0;
    for (int c0 = 0; c0 < n_blocks; c0 += SCAN_THREADS) {
        int i = c0 + threadIdx.x;
        int v = i < n_blocks ? block_sums[i] : 0;
        int total;
        int ex = block_exclusive_scan(v, &total);
        if (i < n_blocks) block_sums[i] = carry + ex;
        carry += total;
    }
}

// rowptr[i] = exclusive prefix; cnt[i] is overwritten with the same value (fill cursor)
__global__ void k_scan_apply(int* __restrict__ cnt, int64_t n_items,
                             const int* __restrict__ block_sums, int* __restrict__ rowptr) {
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS];
    int s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        v[i] = (base + i < n_items) ? cnt[base + i] : 0;
        s += v[i];
    }
    int total;
    int run = block_exclusive_scan(s, &total) + block_sums[blockIdx.x];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        if (base + i < n_items) {
            rowptr[base + i] = run;
            cnt[base + i] = run;
        }
        run += v[i];
    }
}

// whole scan in ONE launch for n_items <= 64 Ki: a single 1024-thread block walks the counters in chunks of 4096 (four
// consecutive counters per thread: coalesced 16-byte accesses) with a running carry; 36 -> ~10 us at the 31 k rows of a
// PCQM batch of 1024 molecules.  (Measured and not kept: one contiguous segment per thread -- 62 dependent strided
// loads per thread, 65 us.)
__global__ void __launch_bounds__(1024)
k_scan_small(int* __restrict__ cnt, int n_items, int* __restrict__ rowptr) {
    __shared__ int s_wave[2][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int carry = 0;
    int buf = 0;
    const bool vec = (((uintptr_t)cnt | (uintptr_t)rowptr) & 15) == 0;     // 16-byte accesses need aligned arrays
    for (int c0 = 0; c0 < n_items; c0 += 4096, buf ^= 1) {
        const int i = c0 + 4 * threadIdx.x;
        int v0 = 0, v1 = 0, v2 = 0, v3 = 0;
        if (vec && i + 3 < n_items) {
            const int4 v = *reinterpret_cast<const int4*>(cnt + i);
            v0 = v.x; v1 = v.y; v2 = v.z; v3 = v.w;
        } else {
            if (i < n_items) v0 = cnt[i];
            if (i + 1 < n_items) v1 = cnt[i + 1];
            if (i + 2 < n_items) v2 = cnt[i + 2];
            if (i + 3 < n_items) v3 = cnt[i + 3];
        }
        const int sum = (v0 + v1) + (v2 + v3);
        int inc = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(inc, d, 64);
            if (lane >= d) inc += t;
        }
        if (lane == 63) s_wave[buf][wave] = inc;
        __syncthreads();                      // (double-buffered wave sums: one barrier per chunk)
        int base = carry, total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const int t = s_wave[buf][w];
            if (w < wave) base += t;
            total += t;
        }
        const int e0 = base + inc - sum, e1 = e0 + v0, e2 = e1 + v1, e3 = e2 + v2;
        if (vec && i + 3 < n_items) {
            *reinterpret_cast<int4*>(rowptr + i) = make_int4(e0, e1, e2, e3);
            *reinterpret_cast<int4*>(cnt + i) = make_int4(e0, e1, e2, e3);
        } else {
            if (i < n_items) { rowptr[i] = e0; cnt[i] = e0; }
            if (i + 1 < n_items) { rowptr[i + 1] = e1; cnt[i + 1] = e1; }
            if (i + 2 < n_items) { rowptr[i + 2] = e2; cnt[i + 2] = e2; }
            if (i + 3 < n_items) { rowptr[i + 3] = e3; cnt[i + 3] = e3; }
        }
        carry += total;
    }
}

template <typename KeyT>
__global__ void k_fill(const KeyT* __restrict__ key, int64_t nnz, int64_t n_rows,
                       int* __restrict__ cursor, int* __restrict__ tmp_perm) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < nnz; p += stride) {
        int64_t k = key[p];
        if (k >= 0 && k < n_rows) {
            int pos = atomicAdd(&cursor[k], 1);
            tmp_perm[pos] = (int)p;
        }
    }
}

__device__ __forceinline__ void emit(int pos, int entry, const int64_t* other, int col_div,
                                     int* perm, int* col) {
    perm[pos] = entry;
    if (col) col[pos] = other ? (int)other[entry] : entry / col_div;
}

// One wavefront per row (grid-stride).  Every lane ranks the entry it holds against the whole row with
// wavefront shuffles (rows of up to 64 entries); longer rows go to the LDS bitonic kernel (a 255-entry row
// -- the padding molecule of a batch padded by 255 atoms -- ranked from memory in O(deg^2/64) kept one
// wavefront busy for 50 us).
constexpr int LONG_ROW = 64;   // longer rows: one workgroup, LDS bitonic (k_sort_long)
__global__ void __launch_bounds__(256)
k_sort_rows(const int* __restrict__ rowptr, int64_t n_rows, const int* __restrict__ tmp_perm,
            const int64_t* __restrict__ other, int col_div, int* __restrict__ perm,
            int* __restrict__ col, int* __restrict__ long_count, int* __restrict__ long_rows) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    for (int64_t r = wave0; r < n_rows; r += (int64_t)gridDim.x * 4) {
        const int beg = rowptr[r], deg = rowptr[r + 1] - beg;
        if (deg == 0) continue;
        if (deg > LONG_ROW) {
            if (lane == 0) long_rows[atomicAdd(long_count, 1)] = (int)r;
            continue;
        }
        const int x = lane < deg ? tmp_perm[beg + lane] : 0x7fffffff;
        int rank = 0;
        for (int j = 0; j < deg; ++j) rank += (__shfl(x, j, 64) < x) ? 1 : 0;
        if (lane < deg) emit(beg + rank, x, other, col_div, perm, col);
    }
}

// one workgroup per long row; "normalised" bitonic network (always ascending compares), which
// tolerates virtual +inf padding above the row length.
__global__ void __launch_bounds__(LONG_THREADS)
k_sort_long(const int* __restrict__ rowptr, const int* __restrict__ tmp_perm,
            const int64_t* __restrict__ other, int col_div, int* __restrict__ perm,
            int* __restrict__ col, const int* __restrict__ long_count,
            const int* __restrict__ long_rows) {
    extern __shared__ int s_row[];
    const int n_long = *long_count;
    for (int li = blockIdx.x; li < n_long; li += gridDim.x) {
        const int r = long_rows[li];
        const int beg = rowptr[r], deg = rowptr[r + 1] - beg;
        if (deg > LONG_LDS_CAP) {  // keeps fill order: reproducible to fp32 rounding only
            for (int a = threadIdx.x; a < deg; a += LONG_THREADS)
                emit(beg + a, tmp_perm[beg + a], other, col_div, perm, col);
            continue;
        }
        for (int a = threadIdx.x; a < deg; a += LONG_THREADS) s_row[a] = tmp_perm[beg + a];
        __syncthreads();
        int p2 = 1;
        while (p2 < deg) p2 <<= 1;
        for (int k = 2; k <= p2; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = threadIdx.x; i < p2; i += LONG_THREADS) {
                    const int l = (j == (k >> 1)) ? (i ^ (k - 1)) : (i ^ j);
                    if (l > i && l < deg) {
                        const int a = s_row[i], b = s_row[l];
                        if (a > b) { s_row[i] = b; s_row[l] = a; }
                    }
                }
                __syncthreads();
            }
        }
        for (int a = threadIdx.x; a < deg; a += LONG_THREADS)
            emit(beg + a, s_row[a], other, col_div, perm, col);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// Batched build: up to CSR_MAX_BATCH independent problems in three launches.  One 1024-thread workgroup
// per problem does clear + histogram + scan + fill with the row counters in LDS (a training batch has
// 4-30 thousand rows); the row sorts then run over all problems at once.  A model step builds three
// CSRs from the batch structure (by hyperedge, by node, by molecule): 18 launches become 3.
// ------------------------------------------------------------------------------------------------
constexpr int CSR_MAX_BATCH = 4;
constexpr int FRONT_THREADS = 1024;
constexpr int64_t FRONT_MAX_ITEMS = 36 * 1024;  // row counters that fit in LDS (144 KB)
constexpr int64_t FRONT_MAX_NNZ = 32 * 1024;    // beyond this one workgroup is slower than the chip-wide path

struct Problem {
    const int64_t* key;
    const int64_t* other;
    int64_t nnz, n_rows;
    int col_div;
    int* rowptr;
    int* perm;
    int* col;
    int* tmp_perm;
    int* long_count;
    int* long_rows;
};
struct Batch {
    Problem p[CSR_MAX_BATCH];
};

__global__ void __launch_bounds__(FRONT_THREADS) k_csr_front(Batch b) {
    extern __shared__ int s_cnt[];
    __shared__ int s_wave[FRONT_THREADS / 64];
    __shared__ int s_carry;
    const Problem& q = b.p[blockIdx.x];
    const int n_items = (int)q.n_rows + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < n_items; i += FRONT_THREADS) s_cnt[i] = 0;
    if (tid == 0) { *q.long_count = 0; s_carry = 0; }
    __syncthreads();
    // keys are read eight at a time per thread (one wait per eight loads: a single workgroup is latency-bound)
    for (int64_t p0 = tid; p0 < q.nnz; p0 += 8 * FRONT_THREADS) {
        int64_t k[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t p = p0 + (int64_t)u * FRONT_THREADS;
            k[u] = p < q.nnz ? q.key[p] : -1;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (k[u] >= 0 && k[u] < q.n_rows) atomicAdd(&s_cnt[k[u]], 1);
    }
    __syncthreads();
    // exclusive scan in chunks of 1024 with a running carry: rowptr (global) and fill cursors (LDS)
    for (int c0 = 0; c0 < n_items; c0 += FRONT_THREADS) {
        const int i = c0 + tid;
        const int v = i < n_items ? s_cnt[i] : 0;
        int inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(inc, d, 64);
            if (lane >= d) inc += t;
        }
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        int base = s_carry;
        for (int w = 0; w < wave; ++w) base += s_wave[w];
        const int ex = base + inc - v;
        if (i < n_items) { q.rowptr[i] = ex; s_cnt[i] = ex; }
        __syncthreads();
        if (tid == FRONT_THREADS - 1) s_carry = ex + v;
        __syncthreads();
    }
    for (int64_t p0 = tid; p0 < q.nnz; p0 += 8 * FRONT_THREADS) {
        int64_t k[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t p = p0 + (int64_t)u * FRONT_THREADS;
            k[u] = p < q.nnz ? q.key[p] : -1;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (k[u] >= 0 && k[u] < q.n_rows)
                q.tmp_perm[atomicAdd(&s_cnt[k[u]], 1)] = (int)(p0 + (int64_t)u * FRONT_THREADS);
    }
}

// the row sorts of all problems: blockIdx.y selects the problem
__device__ __forceinline__ void sort_rows_of(const Problem& q, int64_t wave0, int64_t wave_stride, int lane) {
    for (int64_t r = wave0; r < q.n_rows; r += wave_stride) {
        const int beg = q.rowptr[r], deg = q.rowptr[r + 1] - beg;
        if (deg == 0) continue;
        if (deg > LONG_ROW) {
            if (lane == 0) q.long_rows[atomicAdd(q.long_count, 1)] = (int)r;
            continue;
        }
        const int x = lane < deg ? q.tmp_perm[beg + lane] : 0x7fffffff;
        int rank = 0;
        for (int j = 0; j < deg; ++j) rank += (__shfl(x, j, 64) < x) ? 1 : 0;
        if (lane < deg) emit(beg + rank, x, q.other, q.col_div, q.perm, q.col);
    }
}

__global__ void __launch_bounds__(256) k_sort_rows_batch(Batch b) {
    const Problem& q = b.p[blockIdx.y];
    sort_rows_of(q, (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), (int64_t)gridDim.x * 4, threadIdx.x & 63);
}

__global__ void __launch_bounds__(LONG_THREADS) k_sort_long_batch(Batch b) {
    extern __shared__ int s_row[];
    const Problem& q = b.p[blockIdx.y];
    const int n_long = *q.long_count;
    for (int li = blockIdx.x; li < n_long; li += gridDim.x) {
        const int r = q.long_rows[li];
        const int beg = q.rowptr[r], deg = q.rowptr[r + 1] - beg;
        if (deg > LONG_LDS_CAP) {  // keeps fill order: reproducible to fp32 rounding only
            for (int a = threadIdx.x; a < deg; a += LONG_THREADS)
                emit(beg + a, q.tmp_perm[beg + a], q.other, q.col_div, q.perm, q.col);
            continue;
        }
        for (int a = threadIdx.x; a < deg; a += LONG_THREADS) s_row[a] = q.tmp_perm[beg + a];
        __syncthreads();
        int p2 = 1;
        while (p2 < deg) p2 <<= 1;
        for (int k = 2; k <= p2; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = threadIdx.x; i < p2; i += LONG_THREADS) {
                    const int l = (j == (k >> 1)) ? (i ^ (k - 1)) : (i ^ j);
                    if (l > i && l < deg) {
                        const int x = s_row[i], y = s_row[l];
                        if (x > y) { s_row[i] = y; s_row[l] = x; }
                    }
                }
                __syncthreads();
            }
        }
        for (int a = threadIdx.x; a < deg; a += LONG_THREADS)
            emit(beg + a, s_row[a], q.other, q.col_div, q.perm, q.col);
        __syncthreads();
    }
}

// per-batch index vectors every layer reads, in ONE launch (the torch expressions they replace were 13
// elementwise launches per step): int32 copies of the incidence coordinates and of `batch` (null
// incidences stay -1: hg_segment_reduce_f32 reads a negative index as a zero row, and every other consumer
// walks CSR rows, which never list them) and the 0/1 "row has an incidence" masks of both CSRs
__global__ void k_index_aux(const int64_t* __restrict__ vertex, const int64_t* __restrict__ edges, int64_t nnz,
                            const int64_t* __restrict__ batch, int64_t n_nodes, int64_t n_edges,
                            const int* __restrict__ rowptr_v, const int* __restrict__ rowptr_e,
                            int* __restrict__ v32, int* __restrict__ e32, int* __restrict__ batch32,
                            float* __restrict__ has_v, float* __restrict__ has_e,
                            const int* __restrict__ col_v, const int* __restrict__ col_e, float* __restrict__ ew_v,
                            float* __restrict__ ew_e, int* __restrict__ zero_buf, int64_t zero_n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t i = t0; i < zero_n; i += stride) zero_buf[i] = 0;     // (the counters a later geo_knn_counted adds into)
    // per-entry mean weights of the two CSRs with respect to each other's rows (the backward of a gathered mean reads
    // w[q] next to col[q] instead of chasing col[q] -> rowptr): entry q of the node CSR lists hyperedge col_v[q]
    if (ew_v) {
        const int64_t nv = rowptr_v[n_nodes];
        for (int64_t q = t0; q < nv; q += stride) {
            const int e = col_v[q];
            const int d = rowptr_e[e + 1] - rowptr_e[e];
            ew_v[q] = 1.0f / (float)(d > 1 ? d : 1);
        }
    }
    if (ew_e) {
        const int64_t ne = rowptr_e[n_edges];
        for (int64_t q = t0; q < ne; q += stride) {
            const int v = col_e[q];
            const int d = rowptr_v[v + 1] - rowptr_v[v];
            ew_e[q] = 1.0f / (float)(d > 1 ? d : 1);
        }
    }
    for (int64_t p = t0; p < nnz; p += stride) {
        const int64_t v = vertex[p], e = edges[p];
        v32[p] = (v >= 0 && v < n_nodes && e >= 0 && e < n_edges) ? (int)v : -1;
        e32[p] = (v >= 0 && v < n_nodes && e >= 0 && e < n_edges) ? (int)e : -1;
    }
    for (int64_t i = t0; i < n_nodes; i += stride) {
        if (batch32) batch32[i] = (int)batch[i];
        has_v[i] = rowptr_v[i + 1] > rowptr_v[i] ? 1.f : 0.f;
    }
    for (int64_t i = t0; i < n_edges; i += stride) has_e[i] = rowptr_e[i + 1] > rowptr_e[i] ? 1.f : 0.f;
}

struct Workspace {
    int* tmp_perm;
    int* cnt;
    int* block_sums;
    int* long_count;
    int* long_rows;
    size_t bytes;
};

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

Workspace carve(void* base, int64_t nnz, int64_t n_rows) {
    Workspace w;
    const int64_t n_items = n_rows + 1;
    const int64_t n_blocks = (n_items + SCAN_TILE - 1) / SCAN_TILE;
    size_t off = 0;
    char* b = static_cast<char*>(base);
    w.tmp_perm = reinterpret_cast<int*>(b + off); off += align_up((size_t)(nnz > 0 ? nnz : 1) * 4);
    w.cnt = reinterpret_cast<int*>(b + off); off += align_up((size_t)n_items * 4);
    w.block_sums = reinterpret_cast<int*>(b + off); off += align_up((size_t)n_blocks * 4);
    w.long_count = reinterpret_cast<int*>(b + off); off += 256;
    w.long_rows = reinterpret_cast<int*>(b + off); off += align_up((size_t)(nnz / (SHORT_ROW + 1) + 1) * 4);
    w.bytes = off;
    return w;
}

}  // namespace

extern "C" size_t hg_csr_build_workspace_bytes(int64_t nnz, int64_t n_rows) {
    if (nnz < 0 || n_rows < 0) return 0;
    return carve(nullptr, nnz, n_rows).bytes;
}

// ``counted``: [n_rows + 2] ints holding the histogram of the keys (entries n_rows and n_rows + 1 zero) -- the clear and
// histogram launches are then skipped (the producer of the keys counted them, see geo_knn_counted); the array is consumed
template <typename KeyT>
static int csr_build_impl(const KeyT* key, const int64_t* other, int64_t nnz, int64_t n_rows,
                            int32_t col_div, int32_t* rowptr, int32_t* perm, int32_t* col,
                            void* workspace, size_t workspace_bytes, void* stream_, int32_t* counted = nullptr) {
    if (nnz < 0 || n_rows < 0 || !rowptr || !workspace) return EQH_ERR_ARG;
    if (nnz > 0 && (!key || !perm)) return EQH_ERR_ARG;
    if (!other && col && col_div < 1) return EQH_ERR_ARG;
    if (nnz >= (int64_t)1 << 31 || n_rows >= ((int64_t)1 << 31) - 1) return EQH_ERR_RANGE;
    Workspace w = carve(workspace, nnz, n_rows);
    if (workspace_bytes < w.bytes) return EQH_ERR_ARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);

    const int64_t n_items = n_rows + 1;
    const int n_blocks = (int)((n_items + SCAN_TILE - 1) / SCAN_TILE);
    const int g_nnz = eqh_grid_for(nnz, 256, 2048);
    if (counted) {
        w.cnt = counted;
        w.long_count = counted + n_items;
    } else {
        hipLaunchKernelGGL(k_clear, dim3(eqh_grid_for(n_items, 256, 1024)), dim3(256), 0, stream, w.cnt, n_items,
                           w.long_count);
        EQH_CHECK_LAUNCH();
        if (nnz > 0) {
            hipLaunchKernelGGL(k_hist<KeyT>, dim3(g_nnz), dim3(256), 0, stream, key, nnz, n_rows, w.cnt);
            EQH_CHECK_LAUNCH();
        }
    }
    if (n_items <= 65536) {
        hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(1024), 0, stream, w.cnt, (int)n_items, rowptr);
    } else {
        hipLaunchKernelGGL(k_scan_block_sums, dim3(n_blocks), dim3(SCAN_THREADS), 0, stream, w.cnt,
                           n_items, w.block_sums);
        hipLaunchKernelGGL(k_scan_carry, dim3(1), dim3(SCAN_THREADS), 0, stream, w.block_sums, n_blocks);
        hipLaunchKernelGGL(k_scan_apply, dim3(n_blocks), dim3(SCAN_THREADS), 0, stream, w.cnt, n_items,
                           w.block_sums, rowptr);
    }
    EQH_CHECK_LAUNCH();
    if (nnz > 0) {
        hipLaunchKernelGGL(k_fill<KeyT>, dim3(g_nnz), dim3(256), 0, stream, key, nnz, n_rows, w.cnt,
                           w.tmp_perm);
        hipLaunchKernelGGL(k_sort_rows, dim3(eqh_grid_for(n_rows, 4, 4096)), dim3(256), 0, stream,
                           rowptr, n_rows, w.tmp_perm, other, col_div, perm, col, w.long_count,
                           w.long_rows);
        // rows longer than LONG_ROW (none in molecular batches): LDS bitonic, one workgroup per row
        hipLaunchKernelGGL(k_sort_long, dim3(nnz > 4 * LONG_ROW ? 256 : 1), dim3(LONG_THREADS),
                           LONG_LDS_CAP * sizeof(int), stream, rowptr, w.tmp_perm, other, col_div, perm, col,
                           w.long_count, w.long_rows);
        EQH_CHECK_LAUNCH();
    }
    return EQH_OK;
}

extern "C" int hg_csr_build(const int64_t* key, const int64_t* other, int64_t nnz, int64_t n_rows,
                            int32_t col_div, int32_t* rowptr, int32_t* perm, int32_t* col,
                            void* workspace, size_t workspace_bytes, void* stream_) {
    return csr_build_impl<int64_t>(key, other, nnz, n_rows, col_div, rowptr, perm, col, workspace, workspace_bytes,
                                   stream_);
}

/* same with int32 keys (the neighbour lists geo_knn produces: no widening copy in front of the build) */
extern "C" int hg_csr_build_i32(const int32_t* key, const int64_t* other, int64_t nnz, int64_t n_rows,
                                int32_t col_div, int32_t* rowptr, int32_t* perm, int32_t* col,
                                void* workspace, size_t workspace_bytes, void* stream_) {
    return csr_build_impl<int32_t>(key, other, nnz, n_rows, col_div, rowptr, perm, col, workspace, workspace_bytes,
                                   stream_);
}

/* hg_csr_build_i32 for keys whose histogram exists already: counts [n_rows + 2] = occurrences of every key, then two
 * zeros (zeroed before the producer of the keys counted into it: hg_index_aux's zero_buf, geo_knn_counted).  Two launches
 * fewer; the array is consumed (it becomes the fill cursors). */
extern "C" int hg_csr_build_i32_counted(const int32_t* key, const int64_t* other, int64_t nnz, int64_t n_rows,
                                        int32_t col_div, int32_t* rowptr, int32_t* perm, int32_t* col, int32_t* counts,
                                        void* workspace, size_t workspace_bytes, void* stream_) {
    if (!counts) return EQH_ERR_ARG;
    return csr_build_impl<int32_t>(key, other, nnz, n_rows, col_div, rowptr, perm, col, workspace, workspace_bytes,
                                   stream_, counts);
}

/* Several CSRs in three launches (see k_csr_front).  Problems whose row counters do not fit LDS, or with
 * more than 32 Ki entries (where one workgroup loses to the chip-wide path: measured 70 vs 27 us at
 * 73 728 entries), go through hg_csr_build one by one. */
extern "C" size_t hg_csr_build_batch_workspace_bytes(int32_t n, const int64_t* nnz, const int64_t* n_rows) {
    if (n < 0 || (n > 0 && (!nnz || !n_rows))) return 0;
    size_t total = 0;
    for (int i = 0; i < n; ++i) {
        if (nnz[i] < 0 || n_rows[i] < 0) return 0;
        total += carve(nullptr, nnz[i], n_rows[i]).bytes;
    }
    return total;
}

extern "C" int hg_csr_build_batch(int32_t n, const int64_t* const* key, const int64_t* const* other,
                                  const int64_t* nnz, const int64_t* n_rows, const int32_t* col_div,
                                  int32_t* const* rowptr, int32_t* const* perm, int32_t* const* col,
                                  void* workspace, size_t workspace_bytes, void* stream_) {
    if (n < 0 || (n > 0 && (!key || !other || !nnz || !n_rows || !col_div || !rowptr || !perm || !col || !workspace)))
        return EQH_ERR_ARG;
    if (workspace_bytes < hg_csr_build_batch_workspace_bytes(n, nnz, n_rows)) return EQH_ERR_ARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    char* ws = static_cast<char*>(workspace);
    int i = 0;
    while (i < n) {
        Batch b;
        int m = 0;
        int64_t max_rows = 0, max_items = 0, max_nnz = 0;
        for (; i < n && m < CSR_MAX_BATCH; ++i) {
            if (nnz[i] < 0 || n_rows[i] < 0 || !rowptr[i]) return EQH_ERR_ARG;
            if (nnz[i] > 0 && (!key[i] || !perm[i])) return EQH_ERR_ARG;
            if (!other[i] && col[i] && col_div[i] < 1) return EQH_ERR_ARG;
            if (nnz[i] >= (int64_t)1 << 31 || n_rows[i] >= ((int64_t)1 << 31) - 1) return EQH_ERR_RANGE;
            const Workspace w = carve(ws, nnz[i], n_rows[i]);
            if (n_rows[i] + 1 > FRONT_MAX_ITEMS || nnz[i] > FRONT_MAX_NNZ) {  // the general, chip-wide path
                const int rc = hg_csr_build(key[i], other[i], nnz[i], n_rows[i], col_div[i], rowptr[i], perm[i], col[i],
                                            ws, w.bytes, stream_);
                if (rc) return rc;
                ws += w.bytes;
                continue;
            }
            Problem& q = b.p[m++];
            q.key = key[i]; q.other = other[i]; q.nnz = nnz[i]; q.n_rows = n_rows[i]; q.col_div = col_div[i];
            q.rowptr = rowptr[i]; q.perm = perm[i]; q.col = col[i];
            q.tmp_perm = w.tmp_perm; q.long_count = w.long_count; q.long_rows = w.long_rows;
            ws += w.bytes;
            if (n_rows[i] > max_rows) max_rows = n_rows[i];
            if (n_rows[i] + 1 > max_items) max_items = n_rows[i] + 1;
            if (nnz[i] > max_nnz) max_nnz = nnz[i];
        }
        if (m == 0) continue;
        static bool attr_set = false;
        if (!attr_set) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_csr_front), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(FRONT_MAX_ITEMS * sizeof(int))) != hipSuccess)
                return EQH_ERR_LAUNCH;
            attr_set = true;
        }
        hipLaunchKernelGGL(k_csr_front, dim3(m), dim3(FRONT_THREADS), (size_t)max_items * sizeof(int), stream, b);
        EQH_CHECK_LAUNCH();
        if (max_nnz > 0) {
            hipLaunchKernelGGL(k_sort_rows_batch, dim3(eqh_grid_for(max_rows, 4, 4096), m), dim3(256), 0, stream, b);
            hipLaunchKernelGGL(k_sort_long_batch, dim3(max_nnz > 4 * LONG_ROW ? 64 : 1, m), dim3(LONG_THREADS),
                               LONG_LDS_CAP * sizeof(int), stream, b);
            EQH_CHECK_LAUNCH();
        }
    }
    return EQH_OK;
}

extern "C" int hg_index_aux(const int64_t* vertex, const int64_t* edges, int64_t nnz, const int64_t* batch,
                            int64_t n_nodes, int64_t n_edges, const int32_t* rowptr_v, const int32_t* rowptr_e,
                            int32_t* v32, int32_t* e32, int32_t* batch32, float* has_v, float* has_e,
                            const int32_t* col_v, const int32_t* col_e, float* ew_v, float* ew_e, int32_t* zero_buf,
                            int64_t zero_n, void* stream_) {
    if (nnz < 0 || n_nodes < 0 || n_edges < 0 || zero_n < 0 || (zero_n > 0 && !zero_buf)) return EQH_ERR_ARG;
    if (nnz > 0 && (!vertex || !edges || !v32 || !e32)) return EQH_ERR_ARG;
    if (n_nodes > 0 && (!rowptr_v || !has_v || (batch32 && !batch))) return EQH_ERR_ARG;
    if (n_edges > 0 && (!rowptr_e || !has_e)) return EQH_ERR_ARG;
    int64_t most = nnz > n_nodes ? (nnz > n_edges ? nnz : n_edges) : (n_nodes > n_edges ? n_nodes : n_edges);
    if (zero_n > most) most = zero_n;
    if (most == 0) return EQH_OK;
    hipLaunchKernelGGL(k_index_aux, dim3(eqh_grid_for(most, 256, 1024)), dim3(256), 0, static_cast<hipStream_t>(stream_),
                       vertex, edges, nnz, batch, n_nodes, n_edges, rowptr_v, rowptr_e, v32, e32, batch32, has_v, has_e,
                       (ew_v && col_v) ? col_v : nullptr, (ew_e && col_e) ? col_e : nullptr, col_v ? ew_v : nullptr,
                       col_e ? ew_e : nullptr, zero_buf, zero_n);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}
