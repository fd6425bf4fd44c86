// Fused per-incidence message + aggregation:
//   S[r] = reduce_{p in row r}  LayerNorm( ReLU( PA[ia[p]] + QB[ib[p]] ) )            (sum | mean)
//
// This is the hidden layer of the per-incidence MLPs of conv.py:90-93,96-97,175-177
//   W(cat(X[v], E[e]))  ->  ReLU -> LayerNorm (mlp.py:93-97)  ->  scatter(reduce)
// after the two algebraic moves of layers.py (first Linear split into node/hyperedge-level GEMMs
// PA = X·Waᵀ, QB = E·Wbᵀ + b; last Linear applied after the aggregation).  It replaces two row
// gathers, an add, a ReLU, a LayerNorm and a segmented reduction (six launches and four [nnz,C]
// round trips through memory) by one launch that reads each gathered row once and writes one
// row per output.  HBM-bound: algorithmic bytes 4C·(2·nnz + R) + 12·nnz + 4(R+1).
//
// One 64-lane wavefront per output row; a row of C <= 1024 channels is NV float4 per lane; the
// LayerNorm statistics are wavefront butterfly reductions (no LDS, no atomics).  Nothing is saved
// for the backward: it recomputes h, mean and rstd per incidence.  The backward runs twice — once
// over the CSR keyed by `ia` (giving dPA) and once over the CSR keyed by `ib` (giving dQB) — so
// that each gradient row is owned by one wavefront; d gamma is accumulated per workgroup into a
// slab reduced in block order (bitwise reproducible).
#include "common.h"
#include "rowln.h"

namespace {

constexpr int THREADS = 256;
constexpr int WAVES = THREADS / 64;

// the two gathered operand rows of one incidence (loads only: issued ahead of their use)
template <int NV>
__device__ __forceinline__ void gather_pair(const float* __restrict__ pa, const float* __restrict__ qb, int a, int b,
                                            int C, int lane, Row<NV>& u, Row<NV>& w) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        u.v[i] = (c < C) ? *reinterpret_cast<const float4*>(pa + (int64_t)a * C + c) : f4_zero();
        w.v[i] = (c < C) ? *reinterpret_cast<const float4*>(qb + (int64_t)b * C + c) : f4_zero();
    }
}

template <int NV>
__device__ __forceinline__ void load_norm(const float* __restrict__ pa, const float* __restrict__ qb,
                                          int a, int b, int C, int lane, float inv_c, float eps,
                                          Row<NV>& x, unsigned& pos, float* rstd) {
    Row<NV> u, w;
    gather_pair<NV>(pa, qb, a, b, C, lane, u, w);
    norm_pair<NV>(u, w, C, lane, inv_c, eps, x, pos, rstd);
}

template <int NV>
__global__ void __launch_bounds__(THREADS)
k_inc_fwd(const float* __restrict__ pa, const float* __restrict__ qb, const int* __restrict__ ia,
          const int* __restrict__ ib, const int* __restrict__ rowptr, const int* __restrict__ perm,
          const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ out,
          int n_rows, int C, int mean, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inv_c = 1.0f / (float)C;
    for (int r = blockIdx.x * WAVES + wave; r < n_rows; r += gridDim.x * WAVES) {
        const int beg = rowptr[r], end = rowptr[r + 1];
        Row<NV> acc;
#pragma unroll
        for (int i = 0; i < NV; ++i) acc.v[i] = f4_zero();
        for (int q0 = beg; q0 < end; q0 += 64) {
            // the row's incidence ids and operand rows: one coalesced load each, then broadcast
            const int cnt = (end - q0 < 64) ? (end - q0) : 64;
            int my_a = 0, my_b = 0;
            if (lane < cnt) {
                const int p = perm[q0 + lane];
                my_a = ia[p];
                my_b = ib[p];
            }
            for (int j = 0; j < cnt; ++j) {
                Row<NV> x;
                unsigned pos;
                float rstd;
                load_norm<NV>(pa, qb, __shfl(my_a, j, 64), __shfl(my_b, j, 64), C, lane, inv_c, eps, x, pos, &rstd);
#pragma unroll
                for (int i = 0; i < NV; ++i) f4_add(acc.v[i], x.v[i]);
            }
        }
        const int deg = end - beg;
        const float den = (mean && deg > 1) ? (float)deg : 1.0f;
        const float bscale = mean ? (deg > 0 ? 1.0f : 0.0f) : (float)deg;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (c < C) {
                const float4 g = *reinterpret_cast<const float4*>(gamma + c);
                const float4 b = *reinterpret_cast<const float4*>(beta + c);
                float4 o;
                o.x = fmaf(g.x, acc.v[i].x / den, b.x * bscale);
                o.y = fmaf(g.y, acc.v[i].y / den, b.y * bscale);
                o.z = fmaf(g.z, acc.v[i].z / den, b.z * bscale);
                o.w = fmaf(g.w, acc.v[i].w / den, b.w * bscale);
                *reinterpret_cast<float4*>(out + (int64_t)r * C + c) = o;
            }
        }
    }
}

// The same sum when the OUTPUT row is one of the two operands' indices (always the case in conv.py: messages are
// reduced over the hyperedges or over the nodes of the incidences): row r's own operand row is r itself and the other
// operand's row is the CSR's `col` entry, so the chain rowptr -> perm -> (ia, ib) -> rows shortens to rowptr -> col ->
// rows, and the rows of incidence j+1 are gathered while incidence j is normalised.
template <int NV>
__global__ void __launch_bounds__(THREADS)
k_inc_fwd_col(const float* __restrict__ pa, const float* __restrict__ qb, const int* __restrict__ rowptr,
              const int* __restrict__ col, int row_is_a, const float* __restrict__ gamma, const float* __restrict__ beta,
              float* __restrict__ out, int n_rows, int C, int mean, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inv_c = 1.0f / (float)C;
    const float* __restrict__ own = row_is_a ? pa : qb;      // the operand indexed by the output row
    const float* __restrict__ oth = row_is_a ? qb : pa;      // the operand indexed by the CSR's col
    Row<NV> gam, bet;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        gam.v[i] = (c < C) ? *reinterpret_cast<const float4*>(gamma + c) : f4_zero();
        bet.v[i] = (c < C) ? *reinterpret_cast<const float4*>(beta + c) : f4_zero();
    }
    auto fetch = [&](const float* __restrict__ base, int o, Row<NV>& u) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 4;
            u.v[i] = (c < C) ? *reinterpret_cast<const float4*>(base + (int64_t)o * C + c) : f4_zero();
        }
    };
    for (int r = blockIdx.x * WAVES + wave; r < n_rows; r += gridDim.x * WAVES) {
        const int beg = rowptr[r], end = rowptr[r + 1];
        Row<NV> acc, mine;
        fetch(own, r, mine);                                 // once per row (it does not wait for the index chain)
#pragma unroll
        for (int i = 0; i < NV; ++i) acc.v[i] = f4_zero();
        auto add_norm = [&](const Row<NV>& w) {
#if defined(INC_ABLATE) && INC_ABLATE >= 1      // diagnostic build (tools/inc_ablation.py): the rows are gathered, the LayerNorm is not
#pragma unroll
            for (int i = 0; i < NV; ++i) { f4_add(acc.v[i], mine.v[i]); f4_add(acc.v[i], w.v[i]); }
            return;
#endif
            Row<NV> x;
            unsigned pos;
            float rstd;
            norm_pair<NV>(mine, w, C, lane, inv_c, eps, x, pos, &rstd);
#pragma unroll
            for (int i = 0; i < NV; ++i) f4_add(acc.v[i], x.v[i]);
        };
        for (int q0 = beg; q0 < end; q0 += 64) {
            const int cnt = (end - q0 < 64) ? (end - q0) : 64;
            const int my_o = (lane < cnt) ? col[q0 + lane] : 0;
#if defined(INC_ABLATE) && INC_ABLATE >= 2      // ... and not even gathered: index chain (rowptr -> col) + own row + store only
            acc.v[0].x += (float)my_o;
            continue;
#endif
            for (int j = 0; j < cnt; j += 4) {               // four entries' rows in flight together
                Row<NV> w0, w1, w2, w3;
                const int last = cnt - 1;
                fetch(oth, __builtin_amdgcn_readlane(my_o, j), w0);
                fetch(oth, __builtin_amdgcn_readlane(my_o, (j + 1 < cnt) ? j + 1 : last), w1);
                fetch(oth, __builtin_amdgcn_readlane(my_o, (j + 2 < cnt) ? j + 2 : last), w2);
                fetch(oth, __builtin_amdgcn_readlane(my_o, (j + 3 < cnt) ? j + 3 : last), w3);
                add_norm(w0);
                if (j + 1 < cnt) add_norm(w1);
                if (j + 2 < cnt) add_norm(w2);
                if (j + 3 < cnt) add_norm(w3);
            }
        }
        const int deg = end - beg;
        const float den = (mean && deg > 1) ? (float)deg : 1.0f;
        const float bscale = mean ? (deg > 0 ? 1.0f : 0.0f) : (float)deg;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (c < C) {
                const float4 g = gam.v[i], b = bet.v[i];
                float4 o;
                o.x = fmaf(g.x, acc.v[i].x / den, b.x * bscale);
                o.y = fmaf(g.y, acc.v[i].y / den, b.y * bscale);
                o.z = fmaf(g.z, acc.v[i].z / den, b.z * bscale);
                o.w = fmaf(g.w, acc.v[i].w / den, b.w * bscale);
                *reinterpret_cast<float4*>(out + (int64_t)r * C + c) = o;
            }
        }
    }
}

// ONE gathered operand: out[r] = gamma * reduce_{q in row r} xhat(relu(h[col[q]] + bias)) + beta * [..] -- a dense-row hidden
// layer (Linear -> ReLU -> LayerNorm, mlp.py:91-99) whose output is only ever consumed through the gathered reduction
// that follows it (conv.py:172-173: scatter(W1(X)[vertex], edges) after moving W1's last Linear behind the mean).
// The [N, C] normalised tensor is never written: k_rowln_fwd + k_segment_reduce in one launch.
template <int NV>
__global__ void __launch_bounds__(THREADS)
k_gather_ln_fwd(const float* __restrict__ h, const float* __restrict__ bias, const int* __restrict__ rowptr,
                const int* __restrict__ col, const float* __restrict__ gamma, const float* __restrict__ beta,
                float* __restrict__ out, int n_rows, int C, int mean, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inv_c = 1.0f / (float)C;
    Row<NV> bias_row, gam, bet;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        bias_row.v[i] = (c < C) ? *reinterpret_cast<const float4*>(bias + c) : f4_zero();
        gam.v[i] = (c < C) ? *reinterpret_cast<const float4*>(gamma + c) : f4_zero();
        bet.v[i] = (c < C) ? *reinterpret_cast<const float4*>(beta + c) : f4_zero();
    }
    auto fetch = [&](int o, Row<NV>& u) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 4;
            u.v[i] = (c < C) ? *reinterpret_cast<const float4*>(h + (int64_t)o * C + c) : f4_zero();
        }
    };
    for (int r = blockIdx.x * WAVES + wave; r < n_rows; r += gridDim.x * WAVES) {
        const int beg = rowptr[r], end = rowptr[r + 1];
        Row<NV> acc;
#pragma unroll
        for (int i = 0; i < NV; ++i) acc.v[i] = f4_zero();
        auto add_norm = [&](const Row<NV>& u) {
            Row<NV> x;
            unsigned pos;
            float rstd;
            norm_pair<NV>(u, bias_row, C, lane, inv_c, eps, x, pos, &rstd);
#pragma unroll
            for (int i = 0; i < NV; ++i) f4_add(acc.v[i], x.v[i]);
        };
        for (int q0 = beg; q0 < end; q0 += 64) {
            const int cnt = (end - q0 < 64) ? (end - q0) : 64;
            const int my_o = (lane < cnt) ? col[q0 + lane] : 0;
            // the rows of up to four entries are requested together (hyperedges have 2-3 nodes: one round trip per
            // row instead of one per entry); entries past the end re-read the last one and are not added
            for (int j = 0; j < cnt; j += 4) {
                Row<NV> u0, u1, u2, u3;
                const int last = cnt - 1;
                fetch(__builtin_amdgcn_readlane(my_o, j), u0);
                fetch(__builtin_amdgcn_readlane(my_o, (j + 1 < cnt) ? j + 1 : last), u1);
                fetch(__builtin_amdgcn_readlane(my_o, (j + 2 < cnt) ? j + 2 : last), u2);
                fetch(__builtin_amdgcn_readlane(my_o, (j + 3 < cnt) ? j + 3 : last), u3);
                add_norm(u0);
                if (j + 1 < cnt) add_norm(u1);
                if (j + 2 < cnt) add_norm(u2);
                if (j + 3 < cnt) add_norm(u3);
            }
        }
        const int deg = end - beg;
        const float den = (mean && deg > 1) ? (float)deg : 1.0f;
        const float bscale = mean ? (deg > 0 ? 1.0f : 0.0f) : (float)deg;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (c < C) {
                const float4 g = gam.v[i], b = bet.v[i];
                float4 o;
                o.x = fmaf(g.x, acc.v[i].x / den, b.x * bscale);
                o.y = fmaf(g.y, acc.v[i].y / den, b.y * bscale);
                o.z = fmaf(g.z, acc.v[i].z / den, b.z * bscale);
                o.w = fmaf(g.w, acc.v[i].w / den, b.w * bscale);
                *reinterpret_cast<float4*>(out + (int64_t)r * C + c) = o;
            }
        }
    }
}

// One side of the backward.  Rows of (side_rowptr, side_perm) group the incidences by the index of
// the operand whose gradient is produced (SIDE_A: ia, else ib -- so the row of an incidence is simply
// that index); `okey[p]` is the OUTPUT row of incidence p and `orowptr` the forward CSR's rowptr (for
// the mean weight).
// Hypergraph rows are short (2-3 incidences), so one row per wavefront spends its time in the dependent
// load chain rowptr -> perm -> (ia, ib, okey) -> orowptr -> rows.  A wavefront therefore owns a RANGE of
// consecutive rows: one chain fetches the metadata of up to 64 incidences at once (a lane each), the
// operand rows of incidence j+1 are gathered while incidence j is normalised, and the per-row sums are
// flushed when the (wavefront-uniform) row index changes.
template <int NV, bool DGAMMA, bool SIDE_A>
__device__ __forceinline__ void
inc_bwd_body(const int block, float4* s_g, const float* __restrict__ pa, const float* __restrict__ qb,
             const int* __restrict__ ia, const int* __restrict__ ib, const int* __restrict__ side_rowptr,
             const int* __restrict__ side_perm, const int* __restrict__ okey,
             const int* __restrict__ orowptr, const float* __restrict__ ds,
             const float* __restrict__ gamma, float* __restrict__ dside, float* __restrict__ slab_dgamma,
             int n_side_rows, int C, int mean, float eps, int rows_per_wave) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inv_c = 1.0f / (float)C;
    Row<NV> gam, dgam;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        gam.v[i] = (c < C) ? *reinterpret_cast<const float4*>(gamma + c) : f4_zero();
        dgam.v[i] = f4_zero();
    }
    const int64_t s_beg64 = (int64_t)(block * WAVES + wave) * rows_per_wave;
    const int s_beg = (s_beg64 < n_side_rows) ? (int)s_beg64 : n_side_rows;
    const int s_end = (s_beg + rows_per_wave < n_side_rows) ? s_beg + rows_per_wave : n_side_rows;
    if (s_beg < s_end) {
        const int p_beg = side_rowptr[s_beg], p_end = side_rowptr[s_end];
        int cur_row = s_beg;  // row whose sum `acc` is building
        Row<NV> acc;
#pragma unroll
        for (int i = 0; i < NV; ++i) acc.v[i] = f4_zero();
        auto flush_until = [&](int row) {  // rows [cur_row, row) are complete (empty ones store zeros)
            for (; cur_row < row; ++cur_row) {
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const int c = (lane + 64 * i) * 4;
                    if (c < C) *reinterpret_cast<float4*>(dside + (int64_t)cur_row * C + c) = acc.v[i];
                    acc.v[i] = f4_zero();
                }
            }
        };
        for (int q0 = p_beg; q0 < p_end; q0 += 64) {
            const int cnt = (p_end - q0 < 64) ? (p_end - q0) : 64;
            int my_a = 0, my_b = 0, my_r = 0;
            float my_w = 1.0f;
            if (lane < cnt) {
                const int p = side_perm[q0 + lane];
                my_a = ia[p];
                my_b = ib[p];
                my_r = okey[p];
                const int deg = orowptr[my_r + 1] - orowptr[my_r];
                my_w = (mean && deg > 1) ? 1.0f / (float)deg : 1.0f;
            }
            Row<NV> nu, nw, nd;  // operands of the next incidence
            {
                const int a0 = __builtin_amdgcn_readlane(my_a, 0), b0 = __builtin_amdgcn_readlane(my_b, 0);
                const int r0 = __builtin_amdgcn_readlane(my_r, 0);
                gather_pair<NV>(pa, qb, a0, b0, C, lane, nu, nw);
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const int c = (lane + 64 * i) * 4;
                    nd.v[i] = (c < C) ? *reinterpret_cast<const float4*>(ds + (int64_t)r0 * C + c) : f4_zero();
                }
            }
            for (int j = 0; j < cnt; ++j) {
                const Row<NV> cu = nu, cw = nw, draw = nd;
                const int a_j = __builtin_amdgcn_readlane(my_a, j), b_j = __builtin_amdgcn_readlane(my_b, j);
                const float w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_w), j));
                {
                    const int jn = (j + 1 < cnt) ? j + 1 : j;  // the last one re-reads itself
                    const int a1 = __builtin_amdgcn_readlane(my_a, jn), b1 = __builtin_amdgcn_readlane(my_b, jn);
                    const int r1 = __builtin_amdgcn_readlane(my_r, jn);
                    gather_pair<NV>(pa, qb, a1, b1, C, lane, nu, nw);
#pragma unroll
                    for (int i = 0; i < NV; ++i) {
                        const int c = (lane + 64 * i) * 4;
                        nd.v[i] = (c < C) ? *reinterpret_cast<const float4*>(ds + (int64_t)r1 * C + c) : f4_zero();
                    }
                }
                flush_until(SIDE_A ? a_j : b_j);
                Row<NV> x;
                unsigned pos;
                float rstd;
                norm_pair<NV>(cu, cw, C, lane, inv_c, eps, x, pos, &rstd);
                Row<NV> g;
                float m1 = 0.f, m2 = 0.f;
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    float4 d = draw.v[i];
                    d.x *= w; d.y *= w; d.z *= w; d.w *= w;
                    if (DGAMMA) {
                        dgam.v[i].x = fmaf(d.x, x.v[i].x, dgam.v[i].x); dgam.v[i].y = fmaf(d.y, x.v[i].y, dgam.v[i].y);
                        dgam.v[i].z = fmaf(d.z, x.v[i].z, dgam.v[i].z); dgam.v[i].w = fmaf(d.w, x.v[i].w, dgam.v[i].w);
                    }
                    d.x *= gam.v[i].x; d.y *= gam.v[i].y; d.z *= gam.v[i].z; d.w *= gam.v[i].w;
                    g.v[i] = d;
                    m1 += (d.x + d.y) + (d.z + d.w);
                    m2 += (d.x * x.v[i].x + d.y * x.v[i].y) + (d.z * x.v[i].z + d.w * x.v[i].w);
                }
                wave_sum2(m1, m2);
                m1 *= inv_c;
                m2 *= inv_c;
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const unsigned b = pos >> (4 * i);
                    float4 dx;
                    dx.x = (b & 1u) ? rstd * (g.v[i].x - m1 - x.v[i].x * m2) : 0.f;
                    dx.y = (b & 2u) ? rstd * (g.v[i].y - m1 - x.v[i].y * m2) : 0.f;
                    dx.z = (b & 4u) ? rstd * (g.v[i].z - m1 - x.v[i].z * m2) : 0.f;
                    dx.w = (b & 8u) ? rstd * (g.v[i].w - m1 - x.v[i].w * m2) : 0.f;
                    f4_add(acc.v[i], dx);
                }
            }
        }
        flush_until(s_end);
    }
    if (DGAMMA) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            s_g[threadIdx.x] = dgam.v[i];
            __syncthreads();
            if (wave == 0) {
                float4 t = s_g[lane];
                for (int w2 = 1; w2 < WAVES; ++w2) f4_add(t, s_g[w2 * 64 + lane]);
                const int c = (lane + 64 * i) * 4;
                if (c < C) *reinterpret_cast<float4*>(slab_dgamma + (int64_t)block * C + c) = t;
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Row-wise hidden layer of mlp.py:91-99 on dense rows: y = LayerNorm(relu(h + b)).  Same arithmetic
// as load_norm with a = row, b = the bias row; the backward also produces the column sums that are
// the bias gradient of the preceding Linear, d gamma and d beta (per-workgroup slabs, fixed order).
// ------------------------------------------------------------------------------------------------
// RELU: LayerNorm(relu(h + bias)) (mlp.py:91-99); !RELU: plain LayerNorm(h) (bias unused; egnn_layer.py:192)
template <int NV, bool RELU>
__global__ void __launch_bounds__(THREADS)
k_rowln_fwd(const float* __restrict__ h, const float* __restrict__ bias, const float* __restrict__ gamma,
            const float* __restrict__ beta, float* __restrict__ out, int n_rows, int C, float eps,
            const float* __restrict__ pre_add = nullptr, float h_scale = 1.0f) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inv_c = 1.0f / (float)C;
    Row<NV> bias_row;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        bias_row.v[i] = (RELU && c < C) ? *reinterpret_cast<const float4*>(bias + c) : f4_zero();
    }
    for (int r = blockIdx.x * WAVES + wave; r < n_rows; r += gridDim.x * WAVES) {
        Row<NV> x, hr;
        unsigned pos;
        float rstd;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 4;
            hr.v[i] = (c < C) ? *reinterpret_cast<const float4*>(h + (int64_t)r * C + c) : f4_zero();
            if (pre_add) {      // pre-activation = h_scale * h + pre_add[r] (+ bias): a GEMM's beta = 1 addend applied here
                const float4 a = (c < C) ? *reinterpret_cast<const float4*>(pre_add + (int64_t)r * C + c) : f4_zero();
                hr.v[i] = make_float4(fmaf(h_scale, hr.v[i].x, a.x), fmaf(h_scale, hr.v[i].y, a.y),
                                      fmaf(h_scale, hr.v[i].z, a.z), fmaf(h_scale, hr.v[i].w, a.w));
            }
        }
        norm_pair<NV, RELU>(hr, bias_row, C, lane, inv_c, eps, x, pos, &rstd);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (c < C) {
                const float4 g = *reinterpret_cast<const float4*>(gamma + c);
                const float4 b = *reinterpret_cast<const float4*>(beta + c);
                float4 o;
                o.x = fmaf(g.x, x.v[i].x, b.x); o.y = fmaf(g.y, x.v[i].y, b.y);
                o.z = fmaf(g.z, x.v[i].z, b.z); o.w = fmaf(g.w, x.v[i].w, b.w);
                *reinterpret_cast<float4*>(out + (int64_t)r * C + c) = o;
            }
        }
    }
}

// both sides of the backward in ONE launch: workgroups [0, blocks_a) take the side keyed by ia (d pa and the
// d gamma slabs), the rest the side keyed by ib (d qb).  The two sides are independent, so one launch saves a
// kernel boundary and lets the shorter side's tail overlap the other.
template <int NV>
__global__ void __launch_bounds__(THREADS)
k_inc_bwd_both(const float* __restrict__ pa, const float* __restrict__ qb, const int* __restrict__ ia,
               const int* __restrict__ ib, const int* __restrict__ a_rowptr, const int* __restrict__ a_perm,
               const int* __restrict__ b_rowptr, const int* __restrict__ b_perm, const int* __restrict__ okey,
               const int* __restrict__ orowptr, const float* __restrict__ ds, const float* __restrict__ gamma,
               float* __restrict__ dpa, float* __restrict__ dqb, float* __restrict__ slab_dgamma, int n_a_rows,
               int n_b_rows, int blocks_a, int C, int mean, float eps, int rpw_a, int rpw_b) {
    __shared__ float4 s_g[THREADS];
    if ((int)blockIdx.x < blocks_a)
        inc_bwd_body<NV, true, true>((int)blockIdx.x, s_g, pa, qb, ia, ib, a_rowptr, a_perm, okey, orowptr, ds, gamma, dpa,
                                     slab_dgamma, n_a_rows, C, mean, eps, rpw_a);
    else
        inc_bwd_body<NV, false, false>((int)blockIdx.x - blocks_a, s_g, pa, qb, ia, ib, b_rowptr, b_perm, okey, orowptr,
                                       ds, gamma, dqb, nullptr, n_b_rows, C, mean, eps, rpw_b);
}

// slab layout per workgroup: [dbias | dgamma | dbeta], each C floats
template <int NV, bool RELU>
__global__ void __launch_bounds__(THREADS)
k_rowln_bwd(const float* __restrict__ h, const float* __restrict__ bias, const float* __restrict__ gamma,
            const float* __restrict__ dy, const float* __restrict__ add, float* __restrict__ dh,
            float* __restrict__ slab, int n_rows, int C, float eps, int64_t dy_ld, float* __restrict__ acc_out = nullptr,
            int acc_first = 0, const float* __restrict__ pre_add = nullptr, float h_scale = 1.0f) {
    __shared__ float4 s_red[THREADS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inv_c = 1.0f / (float)C;
    Row<NV> gam, a_db, a_dg, a_dbeta;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        gam.v[i] = (c < C) ? *reinterpret_cast<const float4*>(gamma + c) : f4_zero();
        a_db.v[i] = a_dg.v[i] = a_dbeta.v[i] = f4_zero();
    }
    // rows r, r + stride, ...: the operands of the next row are in flight while this one is normalised (a
    // wavefront walks ~5 rows; unpipelined, each waited a full memory latency for h and again for dy)
    const int stride = gridDim.x * WAVES;
    int r = blockIdx.x * WAVES + wave;
    Row<NV> bias_row, nh, nd, na;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        bias_row.v[i] = (RELU && c < C) ? *reinterpret_cast<const float4*>(bias + c) : f4_zero();
        na.v[i] = f4_zero();
    }
    auto fetch = [&](int row) {
        const int rr = row < n_rows ? row : n_rows - 1;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 4;
            nh.v[i] = (c < C) ? *reinterpret_cast<const float4*>(h + (int64_t)rr * C + c) : f4_zero();
            if (RELU && pre_add) {
                const float4 a = (c < C) ? *reinterpret_cast<const float4*>(pre_add + (int64_t)rr * C + c) : f4_zero();
                nh.v[i] = make_float4(fmaf(h_scale, nh.v[i].x, a.x), fmaf(h_scale, nh.v[i].y, a.y),
                                      fmaf(h_scale, nh.v[i].z, a.z), fmaf(h_scale, nh.v[i].w, a.w));
            }
            nd.v[i] = (c < C) ? *reinterpret_cast<const float4*>(dy + (int64_t)rr * dy_ld + c) : f4_zero();
            if (!RELU && add) na.v[i] = (c < C) ? *reinterpret_cast<const float4*>(add + (int64_t)rr * C + c) : f4_zero();
        }
    };
    if (r < n_rows) fetch(r);
    for (; r < n_rows; r += stride) {
        const Row<NV> ch = nh, cd = nd, ca = na;
        fetch(r + stride);
        Row<NV> x, g;
        unsigned pos;
        float rstd;
        norm_pair<NV, RELU>(ch, bias_row, C, lane, inv_c, eps, x, pos, &rstd);
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            float4 d = cd.v[i];
            f4_add(a_dbeta.v[i], d);
            a_dg.v[i].x = fmaf(d.x, x.v[i].x, a_dg.v[i].x); a_dg.v[i].y = fmaf(d.y, x.v[i].y, a_dg.v[i].y);
            a_dg.v[i].z = fmaf(d.z, x.v[i].z, a_dg.v[i].z); a_dg.v[i].w = fmaf(d.w, x.v[i].w, a_dg.v[i].w);
            d.x *= gam.v[i].x; d.y *= gam.v[i].y; d.z *= gam.v[i].z; d.w *= gam.v[i].w;
            g.v[i] = d;
            m1 += (d.x + d.y) + (d.z + d.w);
            m2 += (d.x * x.v[i].x + d.y * x.v[i].y) + (d.z * x.v[i].z + d.w * x.v[i].w);
        }
        wave_sum2(m1, m2);
        m1 *= inv_c;
        m2 *= inv_c;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 4;
            const unsigned b = pos >> (4 * i);
            float4 dx;
            dx.x = (b & 1u) ? rstd * (g.v[i].x - m1 - x.v[i].x * m2) : 0.f;
            dx.y = (b & 2u) ? rstd * (g.v[i].y - m1 - x.v[i].y * m2) : 0.f;
            dx.z = (b & 4u) ? rstd * (g.v[i].z - m1 - x.v[i].z * m2) : 0.f;
            dx.w = (b & 8u) ? rstd * (g.v[i].w - m1 - x.v[i].w * m2) : 0.f;
            f4_add(a_db.v[i], dx);
            if (!RELU) f4_add(dx, ca.v[i]);      // plain LayerNorm: a second gradient of the same input rides along
            if (c < C) *reinterpret_cast<float4*>(dh + (int64_t)r * C + c) = dx;
            if (acc_out && c < C) {              // the same gradient summed over several applications (see hg_bias_relu_ln_bwd_acc)
                float4 t = dx;
                if (!acc_first) f4_add(t, *reinterpret_cast<const float4*>(acc_out + (int64_t)r * C + c));
                *reinterpret_cast<float4*>(acc_out + (int64_t)r * C + c) = t;
            }
        }
    }
    // combine the workgroup's four wavefronts in a fixed order, one slab row per quantity
    float* __restrict__ sl = slab + (int64_t)blockIdx.x * 3 * C;
#pragma unroll
    for (int which = 0; which < 3; ++which) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            s_red[threadIdx.x] = which == 0 ? a_db.v[i] : (which == 1 ? a_dg.v[i] : a_dbeta.v[i]);
            __syncthreads();
            if (wave == 0) {
                float4 t = s_red[lane];
                for (int w2 = 1; w2 < WAVES; ++w2) f4_add(t, s_red[w2 * 64 + lane]);
                const int c = (lane + 64 * i) * 4;
                if (c < C) *reinterpret_cast<float4*>(sl + which * C + c) = t;
            }
            __syncthreads();
        }
    }
}

// Backward of k_gather_ln_fwd.  The LayerNorm backward is linear in its upstream gradient and every incidence of source
// row v shares v's statistics, so  dh[v] = LNbwd_v( sum_{q in row v of the TRANSPOSED CSR} w[q] * dout[col[q]] ):
// the weighted gathered reduction (k_segment_reduce<weighted>) and k_rowln_bwd in one launch, no [N, C] gradient between
// them.  w[q] = 1 / deg(col[q]) for the mean (hg_entry_weights), NULL for the sum.  A wavefront owns a range of
// consecutive rows: one chain fetches the row ends and up to 64 entries of the range (a lane each), the next gathered row
// and the next h row are in flight while the current ones are used.  Slab layout as k_rowln_bwd: [dbias | dgamma | dbeta].
constexpr int GL_WAVES = 8;    // wavefronts per workgroup of k_gather_ln_bwd (one slab per workgroup)
template <int NV>
__global__ void __launch_bounds__(GL_WAVES * 64)
k_gather_ln_bwd(const float* __restrict__ h, const float* __restrict__ bias, const float* __restrict__ gamma,
                const float* __restrict__ dout, const int* __restrict__ t_rowptr, const int* __restrict__ t_col,
                const float* __restrict__ t_w, float* __restrict__ dh, float* __restrict__ slab, int n_rows, int C,
                float eps, int rows_per_wave) {
    __shared__ float4 s_red[GL_WAVES * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inv_c = 1.0f / (float)C;
    Row<NV> gam, a_db, a_dg, a_dbeta, bias_row;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        gam.v[i] = (c < C) ? *reinterpret_cast<const float4*>(gamma + c) : f4_zero();
        bias_row.v[i] = (c < C) ? *reinterpret_cast<const float4*>(bias + c) : f4_zero();
        a_db.v[i] = a_dg.v[i] = a_dbeta.v[i] = f4_zero();
    }
    const int64_t s_beg64 = (int64_t)(blockIdx.x * GL_WAVES + wave) * rows_per_wave;   // rows_per_wave <= 64
    const int s_beg = (s_beg64 < n_rows) ? (int)s_beg64 : n_rows;
    const int s_end = (s_beg + rows_per_wave < n_rows) ? s_beg + rows_per_wave : n_rows;
    if (s_beg < s_end) {
        const int p_beg = t_rowptr[s_beg];
        const int my_rend = (s_beg + lane < s_end) ? t_rowptr[s_beg + lane + 1] : 0;   // lane i: end of row s_beg + i
        const int p_end = t_rowptr[s_end];
        auto fetch_h = [&](int row, Row<NV>& u) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = (lane + 64 * i) * 4;
                u.v[i] = (c < C) ? *reinterpret_cast<const float4*>(h + (int64_t)row * C + c) : f4_zero();
            }
        };
        auto fetch_d = [&](int e, Row<NV>& u) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = (lane + 64 * i) * 4;
                u.v[i] = (c < C) ? *reinterpret_cast<const float4*>(dout + (int64_t)e * C + c) : f4_zero();
            }
        };
        Row<NV> nh;
        fetch_h(s_beg, nh);                        // does not wait for the index chain
        int q0 = p_beg;
        int cnt = (p_end - q0 < 64) ? (p_end - q0) : 64;
        int my_c = (lane < cnt) ? t_col[q0 + lane] : 0;
        float my_w = (t_w && lane < cnt) ? t_w[q0 + lane] : 1.0f;
        int q = p_beg;
        for (int row = s_beg; row < s_end; ++row) {
            const int rend = __builtin_amdgcn_readlane(my_rend, row - s_beg);
            const Row<NV> ch = nh;
            fetch_h(row + 1 < s_end ? row + 1 : row, nh);
            Row<NV> dsum;
#pragma unroll
            for (int i = 0; i < NV; ++i) dsum.v[i] = f4_zero();
            while (q < rend) {
                if (q - q0 >= 64) {     // next chunk of the range's entries (rare: ranges hold a few short rows)
                    q0 += 64;
                    cnt = (p_end - q0 < 64) ? (p_end - q0) : 64;
                    my_c = (lane < cnt) ? t_col[q0 + lane] : 0;
                    my_w = (t_w && lane < cnt) ? t_w[q0 + lane] : 1.0f;
                }
                // up to four of the row's entries in flight together (those past the row / chunk end re-read the
                // last valid one with weight 0)
                const int j = q - q0;
                int lim = rend - q0;
                if (lim > cnt) lim = cnt;
                const int n = (lim - j < 4) ? lim - j : 4;
                Row<NV> d0, d1, d2, d3;
                const int j1 = (n > 1) ? j + 1 : j, j2 = (n > 2) ? j + 2 : j, j3 = (n > 3) ? j + 3 : j;
                fetch_d(__builtin_amdgcn_readlane(my_c, j), d0);
                fetch_d(__builtin_amdgcn_readlane(my_c, j1), d1);
                fetch_d(__builtin_amdgcn_readlane(my_c, j2), d2);
                fetch_d(__builtin_amdgcn_readlane(my_c, j3), d3);
                const float w0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_w), j));
                const float w1 = (n > 1) ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_w), j1)) : 0.f;
                const float w2 = (n > 2) ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_w), j2)) : 0.f;
                const float w3 = (n > 3) ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_w), j3)) : 0.f;
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    f4_fma(dsum.v[i], d0.v[i], w0);
                    f4_fma(dsum.v[i], d1.v[i], w1);
                    f4_fma(dsum.v[i], d2.v[i], w2);
                    f4_fma(dsum.v[i], d3.v[i], w3);
                }
                q += n;
            }
            Row<NV> x, g;
            unsigned pos;
            float rstd;
            norm_pair<NV>(ch, bias_row, C, lane, inv_c, eps, x, pos, &rstd);
            float m1 = 0.f, m2 = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                float4 d = dsum.v[i];
                f4_add(a_dbeta.v[i], d);
                a_dg.v[i].x = fmaf(d.x, x.v[i].x, a_dg.v[i].x); a_dg.v[i].y = fmaf(d.y, x.v[i].y, a_dg.v[i].y);
                a_dg.v[i].z = fmaf(d.z, x.v[i].z, a_dg.v[i].z); a_dg.v[i].w = fmaf(d.w, x.v[i].w, a_dg.v[i].w);
                d.x *= gam.v[i].x; d.y *= gam.v[i].y; d.z *= gam.v[i].z; d.w *= gam.v[i].w;
                g.v[i] = d;
                m1 += (d.x + d.y) + (d.z + d.w);
                m2 += (d.x * x.v[i].x + d.y * x.v[i].y) + (d.z * x.v[i].z + d.w * x.v[i].w);
            }
            wave_sum2(m1, m2);
            m1 *= inv_c;
            m2 *= inv_c;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = (lane + 64 * i) * 4;
                const unsigned b = pos >> (4 * i);
                float4 dx;
                dx.x = (b & 1u) ? rstd * (g.v[i].x - m1 - x.v[i].x * m2) : 0.f;
                dx.y = (b & 2u) ? rstd * (g.v[i].y - m1 - x.v[i].y * m2) : 0.f;
                dx.z = (b & 4u) ? rstd * (g.v[i].z - m1 - x.v[i].z * m2) : 0.f;
                dx.w = (b & 8u) ? rstd * (g.v[i].w - m1 - x.v[i].w * m2) : 0.f;
                f4_add(a_db.v[i], dx);
                if (c < C) *reinterpret_cast<float4*>(dh + (int64_t)row * C + c) = dx;
            }
        }
    }
    float* __restrict__ sl = slab + (int64_t)blockIdx.x * 3 * C;
#pragma unroll
    for (int which = 0; which < 3; ++which) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            s_red[threadIdx.x] = which == 0 ? a_db.v[i] : (which == 1 ? a_dg.v[i] : a_dbeta.v[i]);
            __syncthreads();
            if (wave == 0) {
                float4 t = s_red[lane];
                for (int w2 = 1; w2 < GL_WAVES; ++w2) f4_add(t, s_red[w2 * 64 + lane]);
                const int c = (lane + 64 * i) * 4;
                if (c < C) *reinterpret_cast<float4*>(sl + which * C + c) = t;
            }
            __syncthreads();
        }
    }
}

// a slab of 3C partial sums per workgroup: few workgroups for the usual ~5 k rows, but the 2 M-row frame tensors of
// FAFormer need the whole chip's worth of wavefronts in flight (each walks its rows one memory round trip at a time)
inline int rowln_blocks(int64_t rows) { return eqh_grid_for(rows, WAVES * 4, rows > 65536 ? 2048 : 256); }

// rows per wavefront of the backward: about 2048 wavefronts per launch (two per SIMD)
inline int bwd_rpw(int64_t rows) {
    const int64_t r = (rows + 2047) / 2048;
    return (int)(r < 1 ? 1 : r);
}
inline int bwd_blocks(int64_t rows) {
    const int64_t per_block = (int64_t)bwd_rpw(rows) * WAVES;
    const int64_t b = (rows + per_block - 1) / per_block;
    return (int)(b < 1 ? 1 : b);
}

// k_gather_ln_bwd keeps its range's row ends one per lane
// (about 8192 wavefronts per launch: the chip's worth, so that every index chain is in flight at once)
inline int gl_rpw(int64_t rows) { const int64_t r = (rows + 8191) / 8192; return r > 64 ? 64 : (r < 1 ? 1 : (int)r); }
inline int gl_blocks(int64_t rows) {
    const int64_t per_block = (int64_t)gl_rpw(rows) * GL_WAVES;
    const int64_t b = (rows + per_block - 1) / per_block;
    return (int)(b < 1 ? 1 : b);
}

int check(int64_t rows, int C) {
    if (rows < 0 || C <= 0) return EQH_ERR_ARG;
    if ((C & 3) || C > 1024) return EQH_ERR_ALIGN;
    if (rows >= ((int64_t)1 << 31) - 1) return EQH_ERR_RANGE;
    return EQH_OK;
}

template <typename F>
int dispatch_nv(int C, F&& f) {
    if (C <= 256) return f(std::integral_constant<int, 1>{});
    if (C <= 512) return f(std::integral_constant<int, 2>{});
    return f(std::integral_constant<int, 4>{});
}

}  // namespace

extern "C" int hg_incidence_ln_reduce_fwd(const float* pa, const float* qb, const int32_t* ia,
                                          const int32_t* ib, const int32_t* rowptr, const int32_t* perm,
                                          const float* gamma, const float* beta, int64_t n_rows,
                                          int32_t C, int32_t mean, float eps, float* out, void* stream_) {
    int rc = check(n_rows, C);
    if (rc) return rc;
    if (n_rows == 0) return EQH_OK;
    if (!pa || !qb || !ia || !ib || !rowptr || !perm || !gamma || !beta || !out) return EQH_ERR_ARG;
    if (!eqh_aligned16(pa) || !eqh_aligned16(qb) || !eqh_aligned16(gamma) || !eqh_aligned16(beta) ||
        !eqh_aligned16(out))
        return EQH_ERR_ALIGN;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    return dispatch_nv(C, [&](auto nv) {
        constexpr int NV = decltype(nv)::value;
        hipLaunchKernelGGL((k_inc_fwd<NV>), dim3(eqh_grid_for(n_rows, WAVES, 4096)), dim3(THREADS), 0, stream,
                           pa, qb, ia, ib, rowptr, perm, gamma, beta, out, (int)n_rows, (int)C, (int)mean, eps);
        EQH_CHECK_LAUNCH();
        return EQH_OK;
    });
}

extern "C" int hg_incidence_ln_reduce_fwd_col(const float* pa, const float* qb, const int32_t* rowptr, const int32_t* col,
                                              int32_t row_is_a, const float* gamma, const float* beta, int64_t n_rows,
                                              int32_t C, int32_t mean, float eps, float* out, void* stream_) {
    int rc = check(n_rows, C);
    if (rc) return rc;
    if (n_rows == 0) return EQH_OK;
    if (!pa || !qb || !rowptr || !col || !gamma || !beta || !out) return EQH_ERR_ARG;
    if (!eqh_aligned16(pa) || !eqh_aligned16(qb) || !eqh_aligned16(gamma) || !eqh_aligned16(beta) ||
        !eqh_aligned16(out))
        return EQH_ERR_ALIGN;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    return dispatch_nv(C, [&](auto nv) {
        constexpr int NV = decltype(nv)::value;
        hipLaunchKernelGGL((k_inc_fwd_col<NV>), dim3(eqh_grid_for(n_rows, WAVES, 4096)), dim3(THREADS), 0, stream,
                           pa, qb, rowptr, col, (int)row_is_a, gamma, beta, out, (int)n_rows, (int)C, (int)mean, eps);
        EQH_CHECK_LAUNCH();
        return EQH_OK;
    });
}

extern "C" size_t hg_incidence_ln_reduce_bwd_workspace_bytes(int64_t n_a_rows, int32_t C) {
    if (n_a_rows < 0 || C <= 0) return 0;
    return (size_t)bwd_blocks(n_a_rows) * (size_t)C * sizeof(float);
}

extern "C" int hg_incidence_ln_reduce_bwd(const float* pa, const float* qb, const int32_t* ia,
                                          const int32_t* ib, const int32_t* a_rowptr,
                                          const int32_t* a_perm, int64_t n_a_rows,
                                          const int32_t* b_rowptr, const int32_t* b_perm,
                                          int64_t n_b_rows, const int32_t* okey, const int32_t* orowptr,
                                          const float* ds, const float* gamma, int32_t C, int32_t mean,
                                          float eps, float* dpa, float* dqb, float* dgamma,
                                          int32_t accumulate, void* workspace, size_t workspace_bytes,
                                          void* stream_) {
    int rc = check(n_a_rows, C);
    if (rc) return rc;
    rc = check(n_b_rows, C);
    if (rc) return rc;
    if (!dgamma || !gamma) return EQH_ERR_ARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n_a_rows == 0 && n_b_rows == 0) return accumulate ? EQH_OK : eqh_zero_async(dgamma, C, stream);
    if (!pa || !qb || !ia || !ib || !a_rowptr || !a_perm || !b_rowptr || !b_perm || !okey || !orowptr || !ds ||
        !dpa || !dqb || !workspace)
        return EQH_ERR_ARG;
    if (!eqh_aligned16(ds) || !eqh_aligned16(dpa) || !eqh_aligned16(dqb) || !eqh_aligned16(workspace))
        return EQH_ERR_ALIGN;
    if (workspace_bytes < hg_incidence_ln_reduce_bwd_workspace_bytes(n_a_rows, C)) return EQH_ERR_ARG;
    float* slab = static_cast<float*>(workspace);
    const int blocks_a = bwd_blocks(n_a_rows);
    return dispatch_nv(C, [&](auto nv) {
        constexpr int NV = decltype(nv)::value;
        hipLaunchKernelGGL((k_inc_bwd_both<NV>), dim3(blocks_a + bwd_blocks(n_b_rows)), dim3(THREADS), 0, stream, pa, qb,
                           ia, ib, a_rowptr, a_perm, b_rowptr, b_perm, okey, orowptr, ds, gamma, dpa, dqb, slab,
                           (int)n_a_rows, (int)n_b_rows, blocks_a, (int)C, (int)mean, eps, bwd_rpw(n_a_rows),
                           bwd_rpw(n_b_rows));
        EQH_CHECK_LAUNCH();
        return eqh_reduce_slabs_async(slab, blocks_a, C, dgamma, stream, accumulate);
    });
}

extern "C" int hg_bias_relu_ln_fwd(const float* h, const float* bias, const float* gamma,
                                   const float* beta, int64_t n_rows, int32_t C, float eps, float* out,
                                   void* stream_) {
    int rc = check(n_rows, C);
    if (rc) return rc;
    if (n_rows == 0) return EQH_OK;
    if (!h || !bias || !gamma || !beta || !out) return EQH_ERR_ARG;
    if (!eqh_aligned16(h) || !eqh_aligned16(bias) || !eqh_aligned16(gamma) || !eqh_aligned16(beta) ||
        !eqh_aligned16(out))
        return EQH_ERR_ALIGN;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    return dispatch_nv(C, [&](auto nv) {
        constexpr int NV = decltype(nv)::value;
        hipLaunchKernelGGL((k_rowln_fwd<NV, true>), dim3(eqh_grid_for(n_rows, WAVES, 4096)), dim3(THREADS), 0, stream, h,
                           bias, gamma, beta, out, (int)n_rows, (int)C, eps);
        EQH_CHECK_LAUNCH();
        return EQH_OK;
    });
}

extern "C" size_t hg_bias_relu_ln_bwd_workspace_bytes(int64_t n_rows, int32_t C) {
    if (n_rows < 0 || C <= 0) return 0;
    return (size_t)rowln_blocks(n_rows) * 3 * (size_t)C * sizeof(float);
}

static int bias_relu_ln_bwd_impl(const float* h, const float* bias, const float* gamma, const float* dy,
                                 int64_t n_rows, int32_t C, float eps, float* dh, float* dbias, float* dgamma,
                                 float* dbeta, int32_t accumulate, void* workspace, size_t workspace_bytes,
                                 float* acc_out, int32_t acc_first, const float* pre_add, float h_scale, void* stream_) {
    int rc = check(n_rows, C);
    if (rc) return rc;
    if (!dbias || !dgamma || !dbeta) return EQH_ERR_ARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n_rows == 0) {
        if (accumulate) return EQH_OK;
        if (eqh_zero_async(dbias, C, stream) || eqh_zero_async(dgamma, C, stream)) return EQH_ERR_LAUNCH;
        return eqh_zero_async(dbeta, C, stream);
    }
    if (!h || !bias || !gamma || !dy || !dh || !workspace) return EQH_ERR_ARG;
    if (!eqh_aligned16(h) || !eqh_aligned16(dy) || !eqh_aligned16(dh) || !eqh_aligned16(workspace) ||
        !eqh_aligned16(bias) || !eqh_aligned16(gamma))
        return EQH_ERR_ALIGN;
    if (workspace_bytes < hg_bias_relu_ln_bwd_workspace_bytes(n_rows, C)) return EQH_ERR_ARG;
    const int blocks = rowln_blocks(n_rows);
    float* slab = static_cast<float*>(workspace);
    return dispatch_nv(C, [&](auto nv) {
        constexpr int NV = decltype(nv)::value;
        hipLaunchKernelGGL((k_rowln_bwd<NV, true>), dim3(blocks), dim3(THREADS), 0, stream, h, bias, gamma, dy,
                           (const float*)nullptr, dh, slab,
                           (int)n_rows, (int)C, eps, (int64_t)C, acc_out, (int)acc_first, pre_add, h_scale);
        EQH_CHECK_LAUNCH();
        return eqh_reduce_slabs3_async(slab, blocks, 3 * (int64_t)C, dbias, dgamma, dbeta, C, C, accumulate, stream);
    });
}

extern "C" int hg_bias_relu_ln_bwd(const float* h, const float* bias, const float* gamma, const float* dy,
                                   int64_t n_rows, int32_t C, float eps, float* dh, float* dbias, float* dgamma,
                                   float* dbeta, int32_t accumulate, void* workspace, size_t workspace_bytes,
                                   void* stream_) {
    return bias_relu_ln_bwd_impl(h, bias, gamma, dy, n_rows, C, eps, dh, dbias, dgamma, dbeta, accumulate, workspace,
                                 workspace_bytes, nullptr, 0, nullptr, 1.0f, stream_);
}

/* Generalised form: the pre-activation is h_scale * h + pre_add[r] + bias (pre_add [n_rows, C] may be NULL: then h_scale
   must be 1) -- the beta = 1 addend and the alpha of the GEMM that produced h, applied here instead of by a copy of the
   addend into the GEMM's output -- and dh (the gradient of the PRE-ACTIVATION: the caller's GEMMs carry h_scale) is also
   summed into acc_out [n_rows, C] when given (overwritten if acc_first != 0, else added to): the gradient of pre_add over
   several applications of the layer (the layer-independent term of conv.py:179-180) without add kernels. */
extern "C" int hg_bias_relu_ln_fwd_ex(const float* h, float h_scale, const float* pre_add, const float* bias, const float* gamma,
                                      const float* beta, int64_t n_rows, int32_t C, float eps, float* out, void* stream_) {
    int rc = check(n_rows, C);
    if (rc) return rc;
    if (n_rows == 0) return EQH_OK;
    if (!h || !bias || !gamma || !beta || !out || (!pre_add && h_scale != 1.0f)) return EQH_ERR_ARG;
    if (!eqh_aligned16(h) || !eqh_aligned16(bias) || !eqh_aligned16(gamma) || !eqh_aligned16(beta) || !eqh_aligned16(out) ||
        !eqh_aligned16(pre_add))
        return EQH_ERR_ALIGN;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    return dispatch_nv(C, [&](auto nv) {
        constexpr int NV = decltype(nv)::value;
        hipLaunchKernelGGL((k_rowln_fwd<NV, true>), dim3(eqh_grid_for(n_rows, WAVES, 4096)), dim3(THREADS), 0, stream, h,
                           bias, gamma, beta, out, (int)n_rows, (int)C, eps, pre_add, h_scale);
        EQH_CHECK_LAUNCH();
        return EQH_OK;
    });
}

extern "C" int hg_bias_relu_ln_bwd_ex(const float* h, float h_scale, const float* pre_add, const float* bias, const float* gamma,
                                      const float* dy, int64_t n_rows, int32_t C, float eps, float* dh, float* dbias,
                                      float* dgamma, float* dbeta, int32_t accumulate, void* workspace, size_t workspace_bytes,
                                      float* acc_out, int32_t acc_first, void* stream_) {
    if ((!pre_add && h_scale != 1.0f) || !eqh_aligned16(pre_add) || !eqh_aligned16(acc_out)) return EQH_ERR_ARG;
    return bias_relu_ln_bwd_impl(h, bias, gamma, dy, n_rows, C, eps, dh, dbias, dgamma, dbeta, accumulate, workspace,
                                 workspace_bytes, acc_out, acc_first, pre_add, h_scale, stream_);
}

/* Linear -> ReLU -> LayerNorm hidden layer on dense rows, consumed only through a gathered reduction (see k_gather_ln_fwd):
   out[r] = gamma * reduce_{q in row r} xhat(relu(h[col[q]] + bias)) + beta * (mean ? [deg r > 0] : deg r) */
extern "C" int hg_gather_ln_reduce_fwd(const float* h, const float* bias, const float* gamma, const float* beta,
                                       const int32_t* rowptr, const int32_t* col, int64_t n_rows, int32_t C, int32_t mean,
                                       float eps, float* out, void* stream_) {
    int rc = check(n_rows, C);
    if (rc) return rc;
    if (n_rows == 0) return EQH_OK;
    if (!h || !bias || !gamma || !beta || !rowptr || !col || !out) return EQH_ERR_ARG;
    if (!eqh_aligned16(h) || !eqh_aligned16(bias) || !eqh_aligned16(gamma) || !eqh_aligned16(beta) || !eqh_aligned16(out))
        return EQH_ERR_ALIGN;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    return dispatch_nv(C, [&](auto nv) {
        constexpr int NV = decltype(nv)::value;
        hipLaunchKernelGGL((k_gather_ln_fwd<NV>), dim3(eqh_grid_for(n_rows, WAVES, 4096)), dim3(THREADS), 0, stream, h, bias,
                           rowptr, col, gamma, beta, out, (int)n_rows, (int)C, (int)mean, eps);
        EQH_CHECK_LAUNCH();
        return EQH_OK;
    });
}

extern "C" size_t hg_gather_ln_reduce_bwd_workspace_bytes(int64_t n_src_rows, int32_t C) {
    if (n_src_rows < 0 || C <= 0) return 0;
    return (size_t)gl_blocks(n_src_rows) * 3 * (size_t)C * sizeof(float);
}

/* backward: (t_rowptr, t_col) = the TRANSPOSED CSR (rows = rows of h), t_w = per-entry weights 1 / deg(col) for the mean
   (hg_entry_weights) or NULL for the sum; dout = gradient of the forward output */
extern "C" int hg_gather_ln_reduce_bwd(const float* h, const float* bias, const float* gamma, const float* dout,
                                       const int32_t* t_rowptr, const int32_t* t_col, const float* t_w, int64_t n_src_rows,
                                       int32_t C, float eps, float* dh, float* dbias, float* dgamma, float* dbeta,
                                       int32_t accumulate, void* workspace, size_t workspace_bytes, void* stream_) {
    int rc = check(n_src_rows, C);
    if (rc) return rc;
    if (!dbias || !dgamma || !dbeta) return EQH_ERR_ARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n_src_rows == 0) {
        if (accumulate) return EQH_OK;
        if (eqh_zero_async(dbias, C, stream) || eqh_zero_async(dgamma, C, stream)) return EQH_ERR_LAUNCH;
        return eqh_zero_async(dbeta, C, stream);
    }
    if (!h || !bias || !gamma || !dout || !t_rowptr || !t_col || !dh || !workspace) return EQH_ERR_ARG;
    if (!eqh_aligned16(h) || !eqh_aligned16(dout) || !eqh_aligned16(dh) || !eqh_aligned16(workspace) ||
        !eqh_aligned16(bias) || !eqh_aligned16(gamma))
        return EQH_ERR_ALIGN;
    if (workspace_bytes < hg_gather_ln_reduce_bwd_workspace_bytes(n_src_rows, C)) return EQH_ERR_ARG;
    const int blocks = gl_blocks(n_src_rows);
    float* slab = static_cast<float*>(workspace);
    return dispatch_nv(C, [&](auto nv) {
        constexpr int NV = decltype(nv)::value;
        hipLaunchKernelGGL((k_gather_ln_bwd<NV>), dim3(blocks), dim3(GL_WAVES * 64), 0, stream, h, bias, gamma, dout, t_rowptr,
                           t_col, t_w, dh, slab, (int)n_src_rows, (int)C, eps, gl_rpw(n_src_rows));
        EQH_CHECK_LAUNCH();
        return eqh_reduce_slabs3_async(slab, blocks, 3 * (int64_t)C, dbias, dgamma, dbeta, C, C, accumulate, stream);
    });
}

/* plain LayerNorm over dense rows (the EGNN node_norm, egnn_layer.py:192) on the same row machinery */
extern "C" int hg_layer_norm_fwd(const float* x, const float* gamma, const float* beta, int64_t n_rows, int32_t C,
                                 float eps, float* out, void* stream_) {
    int rc = check(n_rows, C);
    if (rc) return rc;
    if (n_rows == 0) return EQH_OK;
    if (!x || !gamma || !beta || !out) return EQH_ERR_ARG;
    if (!eqh_aligned16(x) || !eqh_aligned16(gamma) || !eqh_aligned16(beta) || !eqh_aligned16(out)) return EQH_ERR_ALIGN;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    return dispatch_nv(C, [&](auto nv) {
        constexpr int NV = decltype(nv)::value;
        hipLaunchKernelGGL((k_rowln_fwd<NV, false>), dim3(eqh_grid_for(n_rows, WAVES, 4096)), dim3(THREADS), 0, stream, x,
                           (const float*)nullptr, gamma, beta, out, (int)n_rows, (int)C, eps);
        EQH_CHECK_LAUNCH();
        return EQH_OK;
    });
}

extern "C" size_t hg_layer_norm_bwd_workspace_bytes(int64_t n_rows, int32_t C) {
    if (n_rows < 0 || C <= 0) return 0;
    return hg_bias_relu_ln_bwd_workspace_bytes(n_rows, C) + (size_t)C * sizeof(float);  // + a discarded "d bias" row
}

extern "C" int hg_layer_norm_bwd(const float* x, const float* gamma, const float* dy, int64_t dy_ld, const float* add,
                                 int64_t n_rows, int32_t C, float eps, float* dx, float* dgamma, float* dbeta, int32_t accumulate,
                                 void* workspace, size_t workspace_bytes, void* stream_) {
    int rc = check(n_rows, C);
    if (rc) return rc;
    if (!dgamma || !dbeta) return EQH_ERR_ARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n_rows == 0) {
        if (accumulate) return EQH_OK;
        if (eqh_zero_async(dgamma, C, stream)) return EQH_ERR_LAUNCH;
        return eqh_zero_async(dbeta, C, stream);
    }
    if (!x || !gamma || !dy || !dx || !workspace) return EQH_ERR_ARG;
    if (!eqh_aligned16(x) || !eqh_aligned16(dy) || !eqh_aligned16(dx) || !eqh_aligned16(workspace) || !eqh_aligned16(gamma) ||
        !eqh_aligned16(add))
        return EQH_ERR_ALIGN;
    if (workspace_bytes < hg_layer_norm_bwd_workspace_bytes(n_rows, C)) return EQH_ERR_ARG;
    if (dy_ld < C || (dy_ld & 3)) return EQH_ERR_ARG;
    const int blocks = rowln_blocks(n_rows);
    float* slab = static_cast<float*>(workspace);
    float* discard = slab + (size_t)blocks * 3 * C;
    return dispatch_nv(C, [&](auto nv) {
        constexpr int NV = decltype(nv)::value;
        hipLaunchKernelGGL((k_rowln_bwd<NV, false>), dim3(blocks), dim3(THREADS), 0, stream, x, (const float*)nullptr, gamma,
                           dy, add, dx, slab, (int)n_rows, (int)C, eps, dy_ld);
        EQH_CHECK_LAUNCH();
        // the slab's first segment (column sums of dx) has no consumer here; it goes to the discard row.  Never
        // deferred-with-accumulate for that segment's sake: accumulate applies to all three alike, harmlessly.
        return eqh_reduce_slabs3_async(slab, blocks, 3 * (int64_t)C, discard, dgamma, dbeta, C, C, accumulate, stream);
    });
}
