// Dense layer on the fp32 matrix cores with fused operand prologues and epilogues:
//
//     out[M,N] = alpha * A'[M,K] · op(B) (+ bias[N]) (+ C[M,N])
//
// B is either an nn.Linear weight [N,K] (forward, Y = X Wᵀ: "k-major", both operands contiguous along K) or
// the same weight read as [K,N] (input gradient, dX = dY W: "n-major").  A' is A itself or A produced on the fly
// while the operand tile is staged into LDS (the row-op PROLOGUES):
//   * segment:  A'[m] = s(m) · Σ_{q in CSR row m} w(idx[q]) · src[idx[q]]   -- the node<->hyperedge aggregation
//               (torch_scatter.scatter mean/sum of gathered rows, conv.py:172-173, and its backward), so an
//               aggregation followed by a Linear is ONE launch and the aggregated matrix is written only if the
//               backward pass needs it (a_out);
//   * relu_ln:  A'[m] = LayerNorm(relu(A[m] + pbias)) · gamma + beta          -- the hidden layer of mlp.py:91-99
//               between two Linears (row statistics in a pre-pass over the block's rows).
//
// Stands where the reference has nn.Linear / F.linear (mlp.py:91-99, conv.py:169-182, egnn_layer.py:180-208) and
// their input-gradient products.  STATUS (round 2, measured on MI355X, profiles/r02_dense_*): correct to fp32
// rounding in every mode, but NOT faster than the tuned library GEMM at the step's shapes -- [4736 x 256] x
// [256 x 256]: 12.9 us against 12.3 us (hipBLASLt default) / 9.4 us (TunableOp selection, in-graph); 82 against
// 109 TFLOP/s at 31 k rows -- so the models keep the library GEMMs and this kernel is exercised by the tests and
// tools/dense_bench.py only.  Why: PMC counters (SQ_VALU_MFMA_BUSY_CYCLES = 52 % of the kernel, SQ_WAIT_INST_ANY =
// 58 % of wave cycles, 810 MB of operand loads per 4 GFLOP) show the 32 x 64 tile L2 -> LDS bound: 96 KB of operands
// per 1.05 MFLOP is 14 TB/s at the fp32 MFMA peak, the measured ceiling of the L2 -> CU path (16.8-18.8 TB/s), while
// larger tiles leave CUs idle at 4.7 k rows (1.2 tiles per CU).  The fp32 MFMA runs at the VALU rate (64
// FLOP/clk/SIMD), so there is no headroom to buy back with a cleverer inner loop; what would pay is keeping the
// weight tile resident in LDS across row tiles, which needs more row tiles per workgroup than a 4.7 k-row batch has.
//
// Structure.  Tile 32 x 64 per 256-thread workgroup (a [4736 x 256] output is 592 tiles: 2.3 per CU, all resident
// at once, against 1.2 per CU for 64 x 64 tiles -- the CU that holds two then sets the time), four wavefronts as
// 2 (rows) x 2 (columns), each 16 x 32 = two v_mfma_f32_16x16x4_f32 accumulators (two independent dependency
// chains: the instruction issues every 32 cycles but has a 40-cycle dependent latency).  K is walked in chunks
// of BK through two LDS buffers: global -> registers (next chunk, issued before the MFMAs of the current one)
// -> LDS after them, one barrier per chunk.  Operand reads are ds_read_b128: lane (r, g) reads k = 4g .. 4g+3 of
// its row and feeds the four values to four consecutive MFMAs -- the MFMA's k index is a label, so "lane group g
// holds k = 4g + s in MFMA s" is as good as the natural order as long as A and B agree.  Row stride BK + 8 floats
// makes those reads conflict-free (bank = dword address mod 64 over the instruction's 16-lane groups).  For the
// n-major B the tile is stored [k][n] with stride BN + 4 and read with ds_read_b32 (32 consecutive banks per
// half-wave).  Results are bitwise reproducible (fixed k order, no atomics, no split-K).
//
// Up to 8 problems share one launch (independent Linears of one layer fill the chip together); the block -> tile
// map keeps the column tiles of one row tile on one XCD (they share the A rows in that XCD's L2).
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 32, BN = 64;
constexpr int THREADS = 256;
constexpr int MAX_PROBS = 8;
constexpr int GATHER = 4;   // gathered source rows kept in registers per staged element (longer CSR rows: loop)

enum : int { F_KMAJOR = 1, F_MEAN = 2, F_SEGMENT = 4, F_RELU_LN = 8 };

struct Prob {
    const float* A;      // plain / relu_ln: [M, K] (lda); segment: the SOURCE rows [*, K] (lda)
    const float* B;      // k-major: [N, K] (ldb); n-major: [K, N] (ldb)
    const float* bias;   // [N] or null
    const float* C;      // [M, N] (ldc) or null: added to the result
    float* out;          // [M, N] (ldo)
    float* a_out;        // prologue result [M, K] (ld_aout) or null
    const int* rowptr;   // segment: CSR over the M output rows
    const int* idx;      //          source row of every entry (negative: null entry)
    const int* wptr;     //          optional CSR rowptr giving per-source mean weights 1 / max(deg, 1)
    const float* pbias;  // relu_ln: bias added before the ReLU, LayerNorm gamma / beta
    const float* gamma;
    const float* beta;
    int M, N, K;
    int lda, ldb, ldc, ldo, ld_aout;
    float alpha, eps;
    int flags;
    int first_block, tiles_n, n_blocks;
};

struct Batch {
    Prob p[MAX_PROBS];
    int n;
};

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

template <int BK>
struct Tile {
    static constexpr int KV = BK / 4;                 // float4 per k-major tile row
    static constexpr int LDA = BK + 8;                // floats
    static constexpr int LDBN = BN + 4;
    static constexpr int A_ITER = BM * KV / THREADS;  // float4 per thread and chunk
    static constexpr int B_ITER = BN * KV / THREADS;
    static constexpr int A_FLOATS = BM * LDA;
    static constexpr int B_FLOATS = (BN * LDA > BK * LDBN) ? BN * LDA : BK * LDBN;
    static constexpr int BUF_FLOATS = A_FLOATS + B_FLOATS;
    static_assert(A_ITER >= 1 && B_ITER >= 1, "tile too small for 256 threads");
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// ---- operand staging: global -> registers ------------------------------------------------------------------
// Every load is UNCONDITIONAL (addresses clamped into the matrix, values masked afterwards): a guarded load
// ("in range ? load : 0") makes hipcc branch around each load and wait for it before the next one, which
// serialises the whole stage (first version of this file: 15 us for a product the MFMAs need 4 us for).
enum : int { MODE_PLAIN = 0, MODE_SEGMENT = 1, MODE_RELU_LN = 2 };

template <int BK, int MODE>
struct StageRegs {
    static constexpr int G = (MODE == MODE_SEGMENT) ? GATHER : 1;
    float4 a[Tile<BK>::A_ITER][G];
    float aw[Tile<BK>::A_ITER][G];
    int beg[Tile<BK>::A_ITER], end[Tile<BK>::A_ITER];   // segment: the CSR row (kept: a reload at write time would
    float4 b[Tile<BK>::B_ITER];                          // be the NEWEST load in flight and drain the prefetch)
};

__device__ __forceinline__ float4 f4_sel(bool ok, const float4& v) {
    return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}

// FULL: the tile lies inside the matrix and K is a whole number of chunks -- no clamps, no masks (k0 is then a chunk
// start that the caller has already clamped to the last chunk).
template <int BK, bool KMAJOR, bool FULL>
__device__ __forceinline__ void load_b(const Prob& p, int n0, int k0, float4 (&b)[Tile<BK>::B_ITER]) {
    using T = Tile<BK>;
#pragma unroll
    for (int it = 0; it < T::B_ITER; ++it) {
        const int e = threadIdx.x + it * THREADS;
        int n, k;
        if (KMAJOR) { n = n0 + e / T::KV; k = k0 + 4 * (e % T::KV); }          // B[n][k]: tile rows = n
        else { k = k0 + e / (BN / 4); n = n0 + 4 * (e % (BN / 4)); }            // B[k][n]: tile rows = k
        if (!FULL) { n = n < p.N ? n : 0; k = k < p.K ? k : 0; }    // masked when written to LDS (store_stage): touching
        b[it] = KMAJOR ? ld4(p.B + (int64_t)n * p.ldb + k) : ld4(p.B + (int64_t)k * p.ldb + n);   // the value here would wait for it
    }
}

// plain rows (also the raw rows of the relu_ln prologue)
template <int BK, int MODE, bool FULL>
__device__ __forceinline__ void load_a_plain(const Prob& p, int m0, int k0, StageRegs<BK, MODE>& st) {
    using T = Tile<BK>;
#pragma unroll
    for (int it = 0; it < T::A_ITER; ++it) {
        const int e = threadIdx.x + it * THREADS;
        int m = m0 + e / T::KV, k = k0 + 4 * (e % T::KV);
        if (!FULL) { m = m < p.M ? m : 0; k = k < p.K ? k : 0; }    // masked in store_stage
        st.a[it][0] = ld4(p.A + (int64_t)m * p.lda + k);
    }
}

// segment prologue: up to GATHER gathered source rows per element stay in registers until the MFMAs of the
// current chunk have been issued; CSR rows longer than that are finished by a loop at write time.  `last`: index
// of the last CSR entry (entries past a row's end are read from a clamped position and given weight 0).
template <int BK>
__device__ __forceinline__ void load_a_segment(const Prob& p, int m0, int k0, int last, StageRegs<BK, MODE_SEGMENT>& st) {
    using T = Tile<BK>;
#pragma unroll
    for (int it = 0; it < T::A_ITER; ++it) {
        const int e = threadIdx.x + it * THREADS;
        const int m = m0 + e / T::KV, k = k0 + 4 * (e % T::KV);
        const int mc = m < p.M ? m : 0, kc = k < p.K ? k : 0;
        const int beg = p.rowptr[mc];
        const int end = (m < p.M && k < p.K) ? p.rowptr[mc + 1] : beg;
        st.beg[it] = beg;
        st.end[it] = end;
        int j[GATHER];
#pragma unroll
        for (int g = 0; g < GATHER; ++g) {
            const int q = beg + g;
            const int qc = q < last ? q : last;
            const int jj = p.idx ? p.idx[qc] : qc;
            j[g] = (q < end) ? jj : -1;
        }
#pragma unroll
        for (int g = 0; g < GATHER; ++g) {
            const int jc = j[g] < 0 ? 0 : j[g];
            float w = 1.f;
            if (p.wptr) { const int d = p.wptr[jc + 1] - p.wptr[jc]; w = 1.0f / (float)(d > 1 ? d : 1); }
            st.a[it][g] = ld4(p.A + (int64_t)jc * p.lda + kc);
            st.aw[it][g] = j[g] < 0 ? 0.f : w;
        }
    }
}

template <int BK>
__device__ __forceinline__ float4 finish_segment(const Prob& p, int k, int beg, int end, const float4 (&v)[GATHER],
                                                 const float (&w)[GATHER]) {
    float4 acc = f4_zero();      // (rows / chunks out of range have end == beg and all weights 0)
#pragma unroll
    for (int g = 0; g < GATHER; ++g) {
        const float4 t = f4_sel(w[g] != 0.f, v[g]);      // (a masked-out slot may hold anything, NaN included)
        f4_fma(acc, t, w[g]);
    }
    for (int q = beg + GATHER; q < end; ++q) {      // rare: rows with more than GATHER entries
        const int j = p.idx ? p.idx[q] : q;
        if (j < 0) continue;
        float wq = 1.f;
        if (p.wptr) { const int d = p.wptr[j + 1] - p.wptr[j]; wq = 1.0f / (float)(d > 1 ? d : 1); }
        f4_fma(acc, ld4(p.A + (int64_t)j * p.lda + k), wq);
    }
    if (p.flags & F_MEAN) {
        const int deg = end - beg;
        const float s = 1.0f / (float)(deg > 1 ? deg : 1);
        acc.x *= s; acc.y *= s; acc.z *= s; acc.w *= s;
    }
    return acc;
}

// ---- registers -> LDS (with the prologue arithmetic) -----------------------------------------------------------
template <int BK, bool KMAJOR, int MODE, bool FULL>
__device__ __forceinline__ void store_stage(const Prob& p, int m0, int n0, int k0, const StageRegs<BK, MODE>& st,
                                            float* __restrict__ As, float* __restrict__ Bs,
                                            const float* __restrict__ row_stats, const float* __restrict__ ln_par,
                                            bool write_a_out, bool real_chunk) {
    using T = Tile<BK>;
#pragma unroll
    for (int it = 0; it < T::A_ITER; ++it) {
        const int e = threadIdx.x + it * THREADS;
        const int r = e / T::KV, c = 4 * (e % T::KV);
        const int m = m0 + r, k = k0 + c;
        float4 v;
        if constexpr (MODE == MODE_SEGMENT) {
            v = finish_segment<BK>(p, k, st.beg[it], st.end[it], st.a[it], st.aw[it]);
        } else if constexpr (MODE == MODE_RELU_LN) {
            const bool ok = FULL || (m < p.M && k < p.K);
            const int kc = (FULL || k < p.K) ? k : 0;
            const int kp = (p.K + 3) & ~3;    // LayerNorm vectors staged in LDS once per workgroup (dense_tile)
            const float4 x = st.a[it][0], pb = ld4(ln_par + kc), ga = ld4(ln_par + kp + kc), be = ld4(ln_par + 2 * kp + kc);
            const float mu = row_stats[2 * r], rs = row_stats[2 * r + 1];
            v.x = (fmaxf(x.x + pb.x, 0.f) - mu) * rs * ga.x + be.x;
            v.y = (fmaxf(x.y + pb.y, 0.f) - mu) * rs * ga.y + be.y;
            v.z = (fmaxf(x.z + pb.z, 0.f) - mu) * rs * ga.z + be.z;
            v.w = (fmaxf(x.w + pb.w, 0.f) - mu) * rs * ga.w + be.w;
            if (!FULL) v = f4_sel(ok, v);
        } else {
            v = FULL ? st.a[it][0] : f4_sel(m < p.M && k < p.K, st.a[it][0]);
        }
        *reinterpret_cast<float4*>(As + r * T::LDA + c) = v;
        if (MODE != MODE_PLAIN && write_a_out && (FULL ? real_chunk : (m < p.M && k < p.K)))
            *reinterpret_cast<float4*>(p.a_out + (int64_t)m * p.ld_aout + k) = v;
    }
#pragma unroll
    for (int it = 0; it < T::B_ITER; ++it) {
        const int e = threadIdx.x + it * THREADS;
        if (KMAJOR) {
            const bool ok = FULL || (n0 + e / T::KV < p.N && k0 + 4 * (e % T::KV) < p.K);
            *reinterpret_cast<float4*>(Bs + (e / T::KV) * T::LDA + 4 * (e % T::KV)) = FULL ? st.b[it] : f4_sel(ok, st.b[it]);
        } else {
            const bool ok = FULL || (k0 + e / (BN / 4) < p.K && n0 + 4 * (e % (BN / 4)) < p.N);
            *reinterpret_cast<float4*>(Bs + (e / (BN / 4)) * T::LDBN + 4 * (e % (BN / 4))) = FULL ? st.b[it] : f4_sel(ok, st.b[it]);
        }
    }
}

// mean and 1/std of relu(A[m] + pbias) over the K columns, for the BM rows of the tile: eight lanes per row
// (32 rows x 8 lanes = 256 threads), float4 loads, butterfly over the eight lanes (fixed order)
__device__ __forceinline__ void relu_ln_stats(const Prob& p, int m0, float* __restrict__ row_stats) {
    const int r = threadIdx.x >> 3, sub = threadIdx.x & 7;
    const int m = m0 + r;
    const float* row = p.A + (int64_t)(m < p.M ? m : 0) * p.lda;
    float s = 0.f, ss = 0.f;
    for (int k = 4 * sub; k < p.K; k += 32) {
        const float4 a = ld4(row + k), pb = ld4(p.pbias + k);
        const float x0 = fmaxf(a.x + pb.x, 0.f), x1 = fmaxf(a.y + pb.y, 0.f), x2 = fmaxf(a.z + pb.z, 0.f),
                    x3 = fmaxf(a.w + pb.w, 0.f);
        s += (x0 + x1) + (x2 + x3);
    }
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    const float mu = s / (float)p.K;
    for (int k = 4 * sub; k < p.K; k += 32) {
        const float4 a = ld4(row + k), pb = ld4(p.pbias + k);
        const float x0 = fmaxf(a.x + pb.x, 0.f) - mu, x1 = fmaxf(a.y + pb.y, 0.f) - mu,
                    x2 = fmaxf(a.z + pb.z, 0.f) - mu, x3 = fmaxf(a.w + pb.w, 0.f) - mu;
        ss += (x0 * x0 + x1 * x1) + (x2 * x2 + x3 * x3);
    }
    ss += __shfl_xor(ss, 1, 64); ss += __shfl_xor(ss, 2, 64); ss += __shfl_xor(ss, 4, 64);
    if (sub == 0) {
        row_stats[2 * r] = mu;
        row_stats[2 * r + 1] = 1.0f / sqrtf(ss / (float)p.K + p.eps);
    }
}

// ---- the MFMAs of one chunk -----------------------------------------------------------------------------------
template <int BK, bool KMAJOR>
__device__ __forceinline__ void compute_chunk(const float* __restrict__ As, const float* __restrict__ Bs, int wm, int wn,
                                              int r, int g, f32x4 (&acc)[2]) {
    using T = Tile<BK>;
    const float* a_row = As + (16 * wm + r) * T::LDA + 4 * g;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
        const float4 a4 = *reinterpret_cast<const float4*>(a_row + 16 * kk);
        const float av[4] = {a4.x, a4.y, a4.z, a4.w};
        if (KMAJOR) {
            const float4 b0 = *reinterpret_cast<const float4*>(Bs + (32 * wn + r) * T::LDA + 16 * kk + 4 * g);
            const float4 b1 = *reinterpret_cast<const float4*>(Bs + (32 * wn + 16 + r) * T::LDA + 16 * kk + 4 * g);
            const float b0v[4] = {b0.x, b0.y, b0.z, b0.w};
            const float b1v[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc[0] = mfma16(av[s], b0v[s], acc[0]);
                acc[1] = mfma16(av[s], b1v[s], acc[1]);
            }
        } else {
            const float* b_col = Bs + (16 * kk + 4 * g) * T::LDBN + 32 * wn + r;
            float b0v[4], b1v[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) { b0v[s] = b_col[s * T::LDBN]; b1v[s] = b_col[s * T::LDBN + 16]; }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc[0] = mfma16(av[s], b0v[s], acc[0]);
                acc[1] = mfma16(av[s], b1v[s], acc[1]);
            }
        }
    }
}

template <int BK, bool KMAJOR, int MODE, int STAGES, bool FULL>
__device__ __forceinline__ void dense_tile(const Prob& p, int tm, int tn, float* __restrict__ lds) {
    using T = Tile<BK>;
    const int m0 = tm * BM, n0 = tn * BN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave & 1, wn = wave >> 1, r = lane & 15, g = lane >> 4;
    float* row_stats = lds + 2 * T::BUF_FLOATS;
    const bool write_a_out = p.a_out != nullptr && tn == 0;
    int last = 0;
    if (MODE == MODE_SEGMENT) { last = p.rowptr[p.M] - 1; last = last < 0 ? 0 : last; }
    float* ln_par = row_stats + 2 * BM;
    if (MODE == MODE_RELU_LN) {
        relu_ln_stats(p, m0, row_stats);
        const int kp = (p.K + 3) & ~3;
        for (int k = threadIdx.x; k < p.K; k += THREADS) {
            ln_par[k] = p.pbias[k];
            ln_par[kp + k] = p.gamma[k];
            ln_par[2 * kp + k] = p.beta[k];
        }
        __syncthreads();
    }
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    // Software pipeline: two LDS buffers and two register stages.  The global loads of chunk c+2 are issued two
    // MFMA phases before their values are needed (they are written to LDS after the MFMAs of chunk c+1), so a
    // workgroup tolerates ~2 x (BK / 16) x 8 x 32 cycles of L2 latency on its own; co-resident workgroups cover the rest.
    static_assert(STAGES % 2 == 0, "the LDS buffer of a chunk is its parity");
    StageRegs<BK, MODE> rg[STAGES];
    const int n_chunks = (p.K + BK - 1) / BK;
    float* lds0 = lds;
    float* lds1 = lds + T::BUF_FLOATS;
    // The loop body is STRAIGHT-LINE code: no stage is skipped at the ends.  Chunks past the last one are loaded
    // from clamped addresses and masked to zero when written to LDS (k >= K), so they add nothing.  (With the loads
    // under `if (chunk < n_chunks)` hipcc loses count of the outstanding loads at every join and waits vmcnt(0) before
    // each LDS write -- the two-deep prefetch degenerates to none: 15 us instead of 7 for [4736 x 256] x [256 x 256].)
    // FULL tiles read chunks past the end from the LAST chunk (in bounds, never used: the MFMAs of such a chunk are
    // skipped); the a_out copy of a prologue is written for real chunks only.
    auto load = [&](StageRegs<BK, MODE>& st, int c) {
        const int k0 = FULL ? (c < n_chunks ? c : n_chunks - 1) * BK : c * BK;
        if constexpr (MODE == MODE_SEGMENT) load_a_segment<BK>(p, m0, k0, last, st);
        else load_a_plain<BK, MODE, FULL>(p, m0, k0, st);
        load_b<BK, KMAJOR, FULL>(p, n0, k0, st.b);
    };
    auto store = [&](const StageRegs<BK, MODE>& st, int c, float* buf) {
        const int k0 = FULL ? (c < n_chunks ? c : n_chunks - 1) * BK : c * BK;
        store_stage<BK, KMAJOR, MODE, FULL>(p, m0, n0, k0, st, buf, buf + T::A_FLOATS, row_stats, ln_par, write_a_out,
                                            c < n_chunks);
    };
    // STAGES register stages in flight: the loads of chunk c + STAGES are issued right after chunk c's registers
    // have been written to LDS, i.e. STAGES - 1 MFMA phases before they are needed.
#pragma unroll
    for (int i = 0; i < STAGES; ++i) load(rg[i], i);
    store(rg[0], 0, lds0);
    __syncthreads();
    load(rg[0], STAGES);
    for (int c = 0; c < n_chunks; c += STAGES) {
#pragma unroll
        for (int i = 0; i < STAGES; ++i) {          // chunk c + i from LDS buffer i & 1 (STAGES is even)
            float* cur = (i & 1) ? lds1 : lds0;
            float* nxt = (i & 1) ? lds0 : lds1;
            if (c + i < n_chunks) compute_chunk<BK, KMAJOR>(cur, cur + T::A_FLOATS, wm, wn, r, g, acc);
            store(rg[(i + 1) % STAGES], c + i + 1, nxt);
            __syncthreads();
            load(rg[(i + 1) % STAGES], c + i + 1 + STAGES);
        }
    }
    // epilogue: acc[j][i] = out[m0 + 16 wm + 4 g + i][n0 + 32 wn + 16 j + r]
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + 32 * wn + 16 * j + r;
        if (n >= p.N) continue;
        const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + 16 * wm + 4 * g + i;
            if (m >= p.M) continue;
            float v = p.alpha * acc[j][i] + bv;
            if (p.C) v += p.C[(int64_t)m * p.ldc + n];
            p.out[(int64_t)m * p.ldo + n] = v;
        }
    }
}

// One kernel per prologue mode (a launch holds problems of ONE mode): the register stages of the segment prologue are
// four gathered rows wide, and a kernel's VGPR allocation -- hence how many workgroups share a CU -- is that of its
// widest path.
template <int BK, int MODE, int STAGES>
__global__ void __launch_bounds__(THREADS)
k_dense(Batch batch, int total_blocks) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // XCD-aware order: consecutive block ids land on different XCDs (id mod 8 labels the XCD), so give each XCD a
    // contiguous range of logical tiles -- the column tiles of a row tile then share that XCD's L2 copy of A
    const int per_xcd = (total_blocks + 7) / 8;
    int logical = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (logical >= total_blocks) return;   // (grid is rounded up to a multiple of 8)
    int pi = 0;
    for (int i = 1; i < batch.n; ++i)
        if (logical >= batch.p[i].first_block) pi = i;
    const Prob p = batch.p[pi];      // a copy: the fields stay in scalar registers instead of being re-read in the loop
    const int t = logical - p.first_block;
    const int tm = t / p.tiles_n, tn = t % p.tiles_n;
    const bool full = (tm + 1) * BM <= p.M && (tn + 1) * BN <= p.N && p.K % BK == 0;
    if (p.flags & F_KMAJOR) {
        if (full) dense_tile<BK, true, MODE, STAGES, true>(p, tm, tn, lds);
        else dense_tile<BK, true, MODE, STAGES, false>(p, tm, tn, lds);
    } else {
        if (full) dense_tile<BK, false, MODE, STAGES, true>(p, tm, tn, lds);
        else dense_tile<BK, false, MODE, STAGES, false>(p, tm, tn, lds);
    }
}


inline bool aligned_ld(int ld) { return ld % 4 == 0; }

int validate(const Prob& p) {
    if (p.M < 0 || p.N <= 0 || p.K <= 0) return EQH_ERR_ARG;
    if (!p.A || !p.B || !p.out) return EQH_ERR_ARG;
    if ((p.K & 3) || (p.N & 3)) return EQH_ERR_ALIGN;
    if (!aligned_ld(p.lda) || !aligned_ld(p.ldb) || !eqh_aligned16(p.A) || !eqh_aligned16(p.B)) return EQH_ERR_ALIGN;
    if ((p.flags & F_SEGMENT) && (p.flags & F_RELU_LN)) return EQH_ERR_ARG;
    if ((p.flags & F_SEGMENT) && !p.rowptr) return EQH_ERR_ARG;
    if ((p.flags & F_RELU_LN) && (!p.pbias || !p.gamma || !p.beta)) return EQH_ERR_ARG;
    if (p.a_out && (!aligned_ld(p.ld_aout) || !eqh_aligned16(p.a_out))) return EQH_ERR_ALIGN;
    return EQH_OK;
}

}  // namespace

extern "C" int hg_dense_batch_f32(int32_t n, const HgDenseProblem* probs, void* stream_) {
    if (n < 1 || n > MAX_PROBS || !probs) return EQH_ERR_ARG;
    Batch b;
    b.n = 0;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        const HgDenseProblem& q = probs[i];
        Prob p;
        p.A = q.a; p.B = q.b; p.bias = q.bias; p.C = q.c; p.out = q.out; p.a_out = q.a_out;
        p.rowptr = q.seg_rowptr; p.idx = q.seg_idx; p.wptr = q.seg_wptr;
        p.pbias = q.ln_bias; p.gamma = q.ln_gamma; p.beta = q.ln_beta;
        p.M = (int)q.m; p.N = q.n; p.K = q.k;
        p.lda = (int)q.lda; p.ldb = (int)q.ldb; p.ldc = (int)q.ldc; p.ldo = (int)q.ldo; p.ld_aout = (int)q.ld_aout;
        p.alpha = q.alpha; p.eps = q.ln_eps;
        p.flags = (q.b_is_nk ? F_KMAJOR : 0) | (q.seg_mean ? F_MEAN : 0) | (q.seg_rowptr ? F_SEGMENT : 0) |
                  (q.ln_gamma ? F_RELU_LN : 0);
        if (q.m >= ((int64_t)1 << 31) - BM) return EQH_ERR_RANGE;
        const int rc = validate(p);
        if (rc != EQH_OK) return rc;
        if (p.M == 0) continue;
        p.tiles_n = (p.N + BN - 1) / BN;
        p.n_blocks = ((p.M + BM - 1) / BM) * p.tiles_n;
        p.first_block = total;
        total += p.n_blocks;
        b.p[b.n++] = p;
    }
    if (b.n == 0) return EQH_OK;
    const int grid = (total + 7) / 8 * 8;
    const int mode_flags = b.p[0].flags & (F_SEGMENT | F_RELU_LN);
    int ln_k = 0;     // LDS copy of the LayerNorm vectors of the widest relu_ln problem
    for (int i = 0; i < b.n; ++i) {
        if ((b.p[i].flags & (F_SEGMENT | F_RELU_LN)) != mode_flags) return EQH_ERR_ARG;   // one prologue mode per launch
        if ((b.p[i].flags & F_RELU_LN) && b.p[i].K > ln_k) ln_k = b.p[i].K;
    }
    if (ln_k > 4096) return EQH_ERR_ARG;
    constexpr int BK = 32;
    size_t lds_bytes = (2 * Tile<BK>::BUF_FLOATS + 2 * BM) * sizeof(float) + 3 * (size_t)((ln_k + 3) & ~3) * sizeof(float);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (mode_flags & F_SEGMENT)
        hipLaunchKernelGGL((k_dense<BK, MODE_SEGMENT, 2>), dim3(grid), dim3(THREADS), lds_bytes, stream, b, total);
    else if (mode_flags & F_RELU_LN)
        hipLaunchKernelGGL((k_dense<BK, MODE_RELU_LN, 2>), dim3(grid), dim3(THREADS), lds_bytes, stream, b, total);
    else
        hipLaunchKernelGGL((k_dense<BK, MODE_PLAIN, 2>), dim3(grid), dim3(THREADS), lds_bytes, stream, b, total);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}

