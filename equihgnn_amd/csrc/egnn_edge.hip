// Fused EGNN edge update (forward + backward) for k = 16 neighbours and m_dim = 16.
//
// Replaces, for the configuration equihnn_egnn.py:123-129 uses, the reference's
//   feats_j = batched_index_select(feats, nbhd)            egnn_layer.py:298 (+ :18-32)
//   edge_input = cat(feats_i, feats_j, rel_dist)           egnn_layer.py:302-305
//   m_ij = edge_mlp(edge_input)                            egnn_layer.py:180-186,310
//   m_i = m_ij.sum(dim=-2)                                 egnn_layer.py:357-358
// and the autograd of all four (whose gather-backward allocates zeros[1,N,N,C] in the reference).
//
// Formulation.  The first edge Linear is split by input block,
//   h_ij = W1·cat(f_i, f_j, d2_ij) + b1 = A[i] + B[j] + wd·d2_ij,   A = f·W1iᵀ + b1,  B = f·W1jᵀ
// with A and B produced at NODE level by one library GEMM (`ab` = [A | B], row stride 2·Hp).  This
// kernel then does, per node i, for its 16 neighbours j and the Hp hidden units (H = 2(2C+1)
// zero-padded to a multiple of 16; silu(0) = 0 so the padding is inert):
//   s_ij = silu(h_ij);  pre2_ij = W2·s_ij + b2  (a 16 x Hp x 16 product per node -> fp32 MFMA
//   16x16x4, edges on the M axis, outputs on the N axis);  m_i = Σ_j silu(pre2_ij).
// h and s (N·16·Hp floats, 0.3 GB at the BASELINE batch) never touch memory: the backward
// recomputes them from `ab`.
//
// One 64-lane wavefront per node.  MFMA lane map (guide §3): lane l -> r = l & 15, q = l >> 4;
// A operand A[row r][k q], B operand B[k q][col r], accumulator reg g <-> (row 4q+g, col r).
// The K index of every product below is permuted so that each lane consumes 4 CONSECUTIVE hidden
// units (one float4 load) per four MFMAs.
//
// Backward: pass 1 (by receiver i) gives dA, dW2, dwd and stores dpre2; pass 2 (by sender j, over
// the transposed neighbour CSR) gives dB.  No atomics; per-block partial slabs are reduced by a
// third kernel in block order, so every result is bitwise reproducible.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KNB = 16;    // neighbours per node
constexpr int MDIM = 16;   // m_dim
constexpr int THREADS = 256;
constexpr int WAVES = THREADS / 64;
constexpr int BWD_CHUNK = 16;  // nodes per workgroup in backward pass 1
constexpr int PLD = 20;        // LDS row stride (floats) of the 16x16 dpre2 tiles: 16 + 4 pad

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// sigmoid via v_exp_f32 / v_rcp_f32 (each ~1 ulp); silu(x) = x * sigmoid(x)
__device__ __forceinline__ float sigmoid_fast(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}
__device__ __forceinline__ float silu_fast(float x) { return x * sigmoid_fast(x); }
typedef float f32x2 __attribute__((ext_vector_type(2)));
// the same silu on two values with the non-transcendental steps packed
__device__ __forceinline__ f32x2 silu_fast2(f32x2 x) {
    const f32x2 t = x * f32x2{-1.442695041f, -1.442695041f};
    const f32x2 d = f32x2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + f32x2{1.0f, 1.0f};
    return x * f32x2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
}
// returns silu(x) and writes d silu / dx = sig + silu*(1 - sig)
__device__ __forceinline__ float silu_grad(float x, float* ds) {
    const float sig = sigmoid_fast(x);
    const float s = x * sig;
    *ds = fmaf(s, 1.0f - sig, sig);
    return s;
}

__host__ __device__ inline int lds_row_stride(int Hp) {  // (stride mod 64) == 24: conflict-free b128
    return Hp + ((24 - (Hp & 63)) + 64) % 64;
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// Measured facts that shape it (rocprofv3 PMC + in-kernel s_memtime stamps at the BASELINE batch,
// N = 4608, Hp = 1088; profiles/r01_edge_fwd_notes.md):
//  * the fp32 MFMA does not run beside the VALU: a SIMD's time per 16-unit step is the SUM of the
//    SiLU's VALU cycles (v_exp / v_rcp at quarter rate) and the 4 x 32 MFMA cycles, ~420 cycles, whatever
//    the memory system does -- the kernel is issue-bound once its loads are hidden;
//  * loading the B rows directly in MFMA operand order (neighbour r = lane & 15 in the LOW lane bits)
//    costs 38 L1 (TCP) accesses per load instruction and kept the TCP busy for the whole kernel (92 us).
//    Here every wavefront loads its B rows row-contiguously (a quarter-wavefront reads 256 consecutive
//    bytes: 15 accesses per instruction), parks them in a private LDS tile and re-reads them in MFMA
//    order (conflict-free with a 4-float row pad); the receiver's A row takes the same route;
//  * the kNN graph ignores molecule boundaries (egnn_layer.py:253-288 ranks all atoms of the batch), so
//    the gather has no locality to exploit: 341 MB per launch through L2 at a 38% hit rate.
// A 1024-thread workgroup (16 wavefronts, 4 per SIMD) owns a CU and a contiguous run of nodes; four
// wavefronts share one node, each walking a quarter of the hidden units in stages of 64.  Per wavefront
// the work is one software pipeline over (group, stage): loads of the next stage, the neighbour list of
// the next group and the first stage of the next group are all in flight behind SiLU/MFMA work.  The
// four partial 16x16 accumulators meet in LDS and are added in wavefront order: bitwise reproducible.
// W2 lives in LDS when it fits beside the tiles (Hp <= 1280), else it is read through L1.
constexpr int FWD_THREADS = 1024;
constexpr int FWD_SPLIT = 4;                             // wavefronts per node
constexpr int FWD_NODES = FWD_THREADS / 64 / FWD_SPLIT;  // nodes per workgroup iteration
constexpr int FWD_UB = 4;                                // 16-unit steps per stage (64 hidden units)
constexpr int FWD_LD = 16 * FWD_UB + 4;                  // floats per staged row: 64 + 4 pad
constexpr int FWD_TILE = (KNB + 1) * FWD_LD;             // floats per wavefront tile: 16 B rows + the A row

template <int NU>
__device__ __forceinline__ f32x4 fwd_stage(const float* __restrict__ t_rd, const float* __restrict__ t_rda,
                                           const float* __restrict__ wrow, const float* __restrict__ crow,
                                           float dd, f32x4 acc) {
    const f32x2 dd2 = {dd, dd};
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const float4 b4 = *reinterpret_cast<const float4*>(t_rd + 16 * u);
        const float4 a4 = *reinterpret_cast<const float4*>(t_rda + 16 * u);
        const float4 w4 = *reinterpret_cast<const float4*>(wrow + 16 * u);
        const float4 c4 = *reinterpret_cast<const float4*>(crow + 16 * u);
        // two hidden units per packed instruction (v_pk_add/fma/mul_f32 run at twice the scalar-f32 rate);
        // the fp32 MFMA shares the VALU's FMA lanes, so every VALU cycle saved here is wall time
        const f32x2 h01 = f32x2{c4.x, c4.y} * dd2 + (f32x2{a4.x, a4.y} + f32x2{b4.x, b4.y});
        const f32x2 h23 = f32x2{c4.z, c4.w} * dd2 + (f32x2{a4.z, a4.w} + f32x2{b4.z, b4.w});
        const f32x2 s01 = silu_fast2(h01);
        const f32x2 s23 = silu_fast2(h23);
        acc = mfma16(s01.x, w4.x, acc);
        acc = mfma16(s01.y, w4.y, acc);
        acc = mfma16(s23.x, w4.z, acc);
        acc = mfma16(s23.y, w4.w, acc);
    }
    return acc;
}

template <bool W2_LDS>
__global__ void __launch_bounds__(FWD_THREADS)
k_edge_fwd(const float* __restrict__ ab, const float* __restrict__ wd, const float* __restrict__ w2,
           const float* __restrict__ b2, const int* __restrict__ nbr, const float* __restrict__ d2,
           float* __restrict__ m, float* __restrict__ pre2, int N, int Hp) {
    extern __shared__ __attribute__((aligned(16))) float s_mem[];
    const int ldw = W2_LDS ? lds_row_stride(Hp) : Hp;
    float* s_w2 = s_mem;                               // [16][ldw] (absent if !W2_LDS)
    float* s_wd = s_w2 + (W2_LDS ? MDIM * ldw : 0);    // [Hp]
    float* s_tile = s_wd + Hp;                         // [16 wavefronts][FWD_TILE]
    if (W2_LDS) {
        for (int idx = threadIdx.x * 4; idx < MDIM * Hp; idx += FWD_THREADS * 4) {
            const int o = idx / Hp, k = idx - o * Hp;
            *reinterpret_cast<float4*>(s_w2 + o * ldw + k) = *reinterpret_cast<const float4*>(w2 + idx);
        }
    }
    for (int idx = threadIdx.x * 4; idx < Hp; idx += FWD_THREADS * 4)
        *reinterpret_cast<float4*>(s_wd + idx) = *reinterpret_cast<const float4*>(wd + idx);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int slot = wave / FWD_SPLIT, part = wave % FWD_SPLIT;
    const int per = (Hp >> 4) / FWD_SPLIT;  // Hp % 64 == 0: every wavefront gets the same step count
    const int s_beg = part * per;
    const int nfull = per / FWD_UB, tail = per % FWD_UB;
    const int nstages = nfull + (tail ? 1 : 0);
    const float bo = b2[r];
    const float* __restrict__ wrow = (W2_LDS ? s_w2 : w2) + r * ldw + 4 * q + 16 * s_beg;
    const float* __restrict__ crow = s_wd + 4 * q + 16 * s_beg;
    float* tile = s_tile + wave * FWD_TILE;
    float* t_wr = tile + q * FWD_LD + 4 * r;        // row 4i + q (i-th load), 16-byte piece r
    const float* t_rd = tile + r * FWD_LD + 4 * q;  // neighbour r, k-quarter q
    const float* t_rda = tile + KNB * FWD_LD + 4 * q;  // the receiver's own A row (broadcast over r)
    float* t_wra = tile + KNB * FWD_LD + 4 * r;        // written by lanes 0..15
    const int groups = (N + FWD_NODES - 1) / FWD_NODES;
    const int gpb = (groups + gridDim.x - 1) / gridDim.x;
    const int g_beg = blockIdx.x * gpb;
    const int g_end = (g_beg + gpb < groups) ? g_beg + gpb : groups;
    // column (float) offsets of this lane's pieces, clamped into the row: the ragged last stage of the
    // last quarter re-reads the row's final piece instead of running past it (loads stay unconditional:
    // a predicated load would be merged into its destination and waited for on the spot)
    const int b_col0 = 16 * s_beg + 4 * r;
    if (g_beg >= g_end) return;  // whole workgroup
    const int bc0 = (b_col0 < Hp - 4) ? b_col0 : Hp - 4;
    // The per-wavefront work is ONE software pipeline over (group, stage): the loads of stage s+1 fly
    // during the SiLU/MFMA work of stage s, the neighbour list of the next group is fetched during the
    // first stage of the current one, and the first stage of the next group is requested before the last
    // stage of the current one is computed, so nothing but the two barriers separates two groups.
    // (Named scalars, not arrays: an array assigned on both sides of a branch lands in scratch.)
    int node_raw = g_beg * FWD_NODES + slot;
    int node = (node_raw < N) ? node_raw : N - 1;  // tail: recompute the last node, not stored
    float dd = d2[node * KNB + r];
    const float* __restrict__ arow = ab + (int64_t)node * 2 * Hp;
    const float* __restrict__ brow0 = ab + (int64_t)nbr[node * KNB + q] * 2 * Hp + Hp;
    const float* __restrict__ brow1 = ab + (int64_t)nbr[node * KNB + 4 + q] * 2 * Hp + Hp;
    const float* __restrict__ brow2 = ab + (int64_t)nbr[node * KNB + 8 + q] * 2 * Hp + Hp;
    const float* __restrict__ brow3 = ab + (int64_t)nbr[node * KNB + 12 + q] * 2 * Hp + Hp;
    float4 nb0 = *reinterpret_cast<const float4*>(brow0 + bc0);
    float4 nb1 = *reinterpret_cast<const float4*>(brow1 + bc0);
    float4 nb2 = *reinterpret_cast<const float4*>(brow2 + bc0);
    float4 nb3 = *reinterpret_cast<const float4*>(brow3 + bc0);
    // the A row goes through LDS too, fetched by 16 lanes: sixteen lanes asking the TCP for the SAME
    // not-yet-resident line (the MFMA-order load) stall its pipeline until the line arrives
    float4 na = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q == 0) na = *reinterpret_cast<const float4*>(arow + bc0);
    for (int group = g_beg; group < g_end; ++group) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        int j0 = 0, j1 = 0, j2 = 0, j3 = 0, node_raw_n = 0, node_n = 0;
        float dd_n = 0.f;
        for (int st = 0; st < nstages; ++st) {
            *reinterpret_cast<float4*>(t_wr) = nb0;
            *reinterpret_cast<float4*>(t_wr + 4 * FWD_LD) = nb1;
            *reinterpret_cast<float4*>(t_wr + 8 * FWD_LD) = nb2;
            *reinterpret_cast<float4*>(t_wr + 12 * FWD_LD) = nb3;
            if (q == 0) *reinterpret_cast<float4*>(t_wra) = na;
            if (st == 0) {  // next group's neighbour list (the last group re-reads its own)
                const int gn = (group + 1 < g_end) ? group + 1 : group;
                node_raw_n = gn * FWD_NODES + slot;
                node_n = (node_raw_n < N) ? node_raw_n : N - 1;
                j0 = nbr[node_n * KNB + q];
                j1 = nbr[node_n * KNB + 4 + q];
                j2 = nbr[node_n * KNB + 8 + q];
                j3 = nbr[node_n * KNB + 12 + q];
                dd_n = d2[node_n * KNB + r];
            }
            const float* __restrict__ p0 = brow0;
            const float* __restrict__ p1 = brow1;
            const float* __restrict__ p2 = brow2;
            const float* __restrict__ p3 = brow3;
            const float* __restrict__ pa = arow;
            int bc;
            if (st + 1 < nstages) {  // next stage of this group
                const int off = 16 * FWD_UB * (st + 1);
                bc = (b_col0 + off < Hp - 4) ? b_col0 + off : Hp - 4;
            } else {  // first stage of the next group
                // opaque to the optimiser: otherwise the index arithmetic is hoisted above the branch and
                // the neighbour list is waited for right after it was requested
                asm volatile("" : "+v"(j0), "+v"(j1), "+v"(j2), "+v"(j3), "+v"(node_n));
                p0 = ab + (int64_t)j0 * 2 * Hp + Hp;
                p1 = ab + (int64_t)j1 * 2 * Hp + Hp;
                p2 = ab + (int64_t)j2 * 2 * Hp + Hp;
                p3 = ab + (int64_t)j3 * 2 * Hp + Hp;
                pa = ab + (int64_t)node_n * 2 * Hp;
                bc = bc0;
            }
            nb0 = *reinterpret_cast<const float4*>(p0 + bc);
            nb1 = *reinterpret_cast<const float4*>(p1 + bc);
            nb2 = *reinterpret_cast<const float4*>(p2 + bc);
            nb3 = *reinterpret_cast<const float4*>(p3 + bc);
            if (q == 0) na = *reinterpret_cast<const float4*>(pa + bc);
            __builtin_amdgcn_sched_barrier(0);
            const float* wr = wrow + 16 * FWD_UB * st;
            const float* cr = crow + 16 * FWD_UB * st;
            if (st < nfull) acc = fwd_stage<FWD_UB>(t_rd, t_rda, wr, cr, dd, acc);
            else if (tail == 1) acc = fwd_stage<1>(t_rd, t_rda, wr, cr, dd, acc);
            else if (tail == 2) acc = fwd_stage<2>(t_rd, t_rda, wr, cr, dd, acc);
            else acc = fwd_stage<3>(t_rd, t_rda, wr, cr, dd, acc);
            __builtin_amdgcn_sched_barrier(0);
            if (st + 1 == nstages) { brow0 = p0; brow1 = p1; brow2 = p2; brow3 = p3; arow = pa; }
        }
        // partial accumulators: wavefronts 1..3 of the node park theirs in their own (now idle) tile
        if (part > 0) *reinterpret_cast<f32x4*>(tile + lane * 4) = acc;
        __syncthreads();
        if (part == 0 && node_raw < N) {
#pragma unroll
            for (int pp = 1; pp < FWD_SPLIT; ++pp)
                acc += *reinterpret_cast<const f32x4*>(tile + pp * FWD_TILE + lane * 4);
            // acc[g] = pre2[j = 4q+g][o = r] (before bias)
            float msum = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float pv = acc[g] + bo;
                pre2[(int64_t)node * (KNB * MDIM) + (4 * q + g) * MDIM + r] = pv;
                msum += silu_fast(pv);
            }
            msum += __shfl_xor(msum, 16, 64);
            msum += __shfl_xor(msum, 32, 64);
            if (q == 0) m[(int64_t)node * MDIM + r] = msum;
        }
        __syncthreads();  // tiles are overwritten by the next group's first stage
        node_raw = node_raw_n;
        node = node_n;
        dd = dd_n;
    }
}

// ------------------------------------------------------------------------------------------------
// backward pass 1: by receiver node.  One workgroup owns BWD_CHUNK consecutive nodes; wave w owns
// the hidden tiles t = w, w+4, ... (16 hidden units each) for ALL nodes of the chunk, so its dW2 /
// dwd tile accumulators stay in registers across the chunk and need no cross-wave reduction.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(THREADS)
k_edge_bwd_recv(const float* __restrict__ ab, const float* __restrict__ wd,
                const float* __restrict__ w2, const int* __restrict__ nbr,
                const float* __restrict__ d2, const float* __restrict__ pre2,
                const float* __restrict__ dm, float* __restrict__ dpre2_out,
                float* __restrict__ dab, float* __restrict__ slab_w2, float* __restrict__ slab_wd,
                int N, int Hp) {
    __shared__ __attribute__((aligned(16))) float s_p[BWD_CHUNK][KNB * PLD];   // dpre2[n][j][o]
    __shared__ __attribute__((aligned(16))) float s_pt[BWD_CHUNK][MDIM * PLD];  // dpre2[n][o][j]
    __shared__ int s_nbr[BWD_CHUNK][KNB];
    __shared__ float s_d2[BWD_CHUNK][KNB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int base = blockIdx.x * BWD_CHUNK;
    const int cnt = (N - base < BWD_CHUNK) ? (N - base) : BWD_CHUNK;

    // phase A: dpre2 = dm * silu'(pre2), one float4 (4 consecutive o of one j) per lane
    for (int n = wave; n < cnt; n += WAVES) {
        const int node = base + n;
        const int jj = lane >> 2, o0 = (lane & 3) * 4;
        const float4 p4 = *reinterpret_cast<const float4*>(pre2 + (int64_t)node * 256 + lane * 4);
        const float4 g4 = *reinterpret_cast<const float4*>(dm + (int64_t)node * MDIM + o0);
        float ds;
        float4 d4;
        silu_grad(p4.x, &ds); d4.x = g4.x * ds;
        silu_grad(p4.y, &ds); d4.y = g4.y * ds;
        silu_grad(p4.z, &ds); d4.z = g4.z * ds;
        silu_grad(p4.w, &ds); d4.w = g4.w * ds;
        *reinterpret_cast<float4*>(&s_p[n][jj * PLD + o0]) = d4;
        s_pt[n][(o0 + 0) * PLD + jj] = d4.x;
        s_pt[n][(o0 + 1) * PLD + jj] = d4.y;
        s_pt[n][(o0 + 2) * PLD + jj] = d4.z;
        s_pt[n][(o0 + 3) * PLD + jj] = d4.w;
        *reinterpret_cast<float4*>(dpre2_out + (int64_t)node * 256 + lane * 4) = d4;
        if (lane < KNB) {
            s_nbr[n][lane] = nbr[node * KNB + lane];
            s_d2[n][lane] = d2[node * KNB + lane];
        }
    }
    __syncthreads();

    // phase B: 64 hidden units per step (four MFMA column tiles); lane (r, q) owns the FOUR
    // consecutive units k4 .. k4+3 with k4 = 64 T + 4 r, so every operand is a float4 load/store
    // (column n = r of tile c is hidden unit k4 + c).
    const int stiles = Hp >> 6;
    for (int T = wave; T < stiles; T += WAVES) {
        const int k4 = 64 * T + 4 * r;
        float4 wv[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) wv[s] = *reinterpret_cast<const float4*>(w2 + (4 * q + s) * Hp + k4);
        const float4 wd4 = *reinterpret_cast<const float4*>(wd + k4);
        f32x4 acc_w[4];  // acc_w[c][g] = dW2[o = 4q+g][k4 + c]
#pragma unroll
        for (int c = 0; c < 4; ++c) acc_w[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        float acc_wd[4] = {0.f, 0.f, 0.f, 0.f};
        for (int n = 0; n < cnt; ++n) {
            const int node = base + n;
            const float4 pa = *reinterpret_cast<const float4*>(&s_p[n][r * PLD + 4 * q]);
            const float pav[4] = {pa.x, pa.y, pa.z, pa.w};
            f32x4 gk[4];  // gk[c][g] = sum_o dpre2[j = 4q+g][o] * W2[o][k4 + c]
#pragma unroll
            for (int c = 0; c < 4; ++c) gk[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                gk[0] = mfma16(pav[s], wv[s].x, gk[0]);
                gk[1] = mfma16(pav[s], wv[s].y, gk[1]);
                gk[2] = mfma16(pav[s], wv[s].z, gk[2]);
                gk[3] = mfma16(pav[s], wv[s].w, gk[3]);
            }
            const float4 ai4 = *reinterpret_cast<const float4*>(ab + (int64_t)node * 2 * Hp + k4);
            const float aiv[4] = {ai4.x, ai4.y, ai4.z, ai4.w};
            const float wdv[4] = {wd4.x, wd4.y, wd4.z, wd4.w};
            float sv[4][4];  // sv[c][g] = silu(h[j = 4q+g][k4 + c])
            float da[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int jn = s_nbr[n][4 * q + g];
                const float dd = s_d2[n][4 * q + g];
                const float4 b4 = *reinterpret_cast<const float4*>(ab + (int64_t)jn * 2 * Hp + Hp + k4);
                const float bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float h = fmaf(wdv[c], dd, aiv[c] + bv[c]);
                    float ds;
                    sv[c][g] = silu_grad(h, &ds);
                    const float dh = gk[c][g] * ds;
                    da[c] += dh;
                    acc_wd[c] = fmaf(dh, dd, acc_wd[c]);
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                da[c] += __shfl_xor(da[c], 16, 64);
                da[c] += __shfl_xor(da[c], 32, 64);
            }
            if (q == 0)
                *reinterpret_cast<float4*>(dab + (int64_t)node * 2 * Hp + k4) = make_float4(da[0], da[1], da[2], da[3]);
            // dW2[o][k] += sum_j dpre2[j][o] * s[j][k]
            const float4 pt = *reinterpret_cast<const float4*>(&s_pt[n][r * PLD + 4 * q]);
            const float ptv[4] = {pt.x, pt.y, pt.z, pt.w};
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                acc_w[0] = mfma16(ptv[g], sv[0][g], acc_w[0]);
                acc_w[1] = mfma16(ptv[g], sv[1][g], acc_w[1]);
                acc_w[2] = mfma16(ptv[g], sv[2][g], acc_w[2]);
                acc_w[3] = mfma16(ptv[g], sv[3][g], acc_w[3]);
            }
        }
        float* __restrict__ sw = slab_w2 + (int64_t)blockIdx.x * MDIM * Hp;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(sw + (4 * q + g) * Hp + k4) =
                make_float4(acc_w[0][g], acc_w[1][g], acc_w[2][g], acc_w[3][g]);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            acc_wd[c] += __shfl_xor(acc_wd[c], 16, 64);
            acc_wd[c] += __shfl_xor(acc_wd[c], 32, 64);
        }
        if (q == 0)
            *reinterpret_cast<float4*>(slab_wd + (int64_t)blockIdx.x * Hp + k4) =
                make_float4(acc_wd[0], acc_wd[1], acc_wd[2], acc_wd[3]);
    }
}

// ------------------------------------------------------------------------------------------------
// backward pass 2: by sender node j, over the transposed neighbour CSR (entries e = i*16 + slot).
// dB[j][k] = sum over in-edges of g_e[k] * silu'(h_e[k]).  One wavefront per sender; in-edges are
// consumed 16 at a time (one MFMA tile); groups after the first accumulate into the row this wave
// owns exclusively.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(THREADS)
k_edge_bwd_send(const float* __restrict__ ab, const float* __restrict__ wd,
                const float* __restrict__ w2, const float* __restrict__ d2,
                const float* __restrict__ dpre2, const int* __restrict__ t_rowptr,
                const int* __restrict__ t_perm, float* __restrict__ dab, int N, int Hp) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    for (int node = blockIdx.x * WAVES + wave; node < N; node += gridDim.x * WAVES) {
        const int beg = t_rowptr[node], end = t_rowptr[node + 1];
        float* __restrict__ drow = dab + (int64_t)node * 2 * Hp + Hp;
        const float* __restrict__ brow = ab + (int64_t)node * 2 * Hp + Hp;
        if (beg == end) {
            for (int k = lane; k < Hp; k += 64) drow[k] = 0.f;
            continue;
        }
        for (int g0 = beg; g0 < end; g0 += 16) {
            const bool first = (g0 == beg);
            // entry owned by lane r (as MFMA row), replicated over q
            const int e_r = (g0 + r < end) ? t_perm[g0 + r] : -1;
            // the four entries whose results land in this lane: rows 4q+g
            int e_g[4];
            float dd_g[4];
            int64_t arow_g[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                e_g[g] = __shfl(e_r, 4 * q + g, 64);
                dd_g[g] = e_g[g] >= 0 ? d2[e_g[g]] : 0.f;
                arow_g[g] = (int64_t)(e_g[g] >= 0 ? (e_g[g] >> 4) : 0) * 2 * Hp;
            }
            float4 pa = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e_r >= 0) pa = *reinterpret_cast<const float4*>(dpre2 + (int64_t)e_r * MDIM + 4 * q);
            const float pav[4] = {pa.x, pa.y, pa.z, pa.w};
            const int stiles = Hp >> 6;
            for (int T = 0; T < stiles; ++T) {
                const int k4 = 64 * T + 4 * r;  // this lane's four consecutive hidden units
                f32x4 gk[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) gk[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float4 w4 = *reinterpret_cast<const float4*>(w2 + (4 * q + s) * Hp + k4);
                    gk[0] = mfma16(pav[s], w4.x, gk[0]);
                    gk[1] = mfma16(pav[s], w4.y, gk[1]);
                    gk[2] = mfma16(pav[s], w4.z, gk[2]);
                    gk[3] = mfma16(pav[s], w4.w, gk[3]);
                }
                const float4 bj4 = *reinterpret_cast<const float4*>(brow + k4);
                const float4 wd4 = *reinterpret_cast<const float4*>(wd + k4);
                const float bjv[4] = {bj4.x, bj4.y, bj4.z, bj4.w};
                const float wdv[4] = {wd4.x, wd4.y, wd4.z, wd4.w};
                float db[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 a4 = *reinterpret_cast<const float4*>(ab + arow_g[g] + k4);
                    const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float h = fmaf(wdv[c], dd_g[g], av[c] + bjv[c]);
                        float ds;
                        silu_grad(h, &ds);
                        db[c] = fmaf(gk[c][g], ds, db[c]);  // gk is exactly 0 for padded entries
                    }
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    db[c] += __shfl_xor(db[c], 16, 64);
                    db[c] += __shfl_xor(db[c], 32, 64);
                }
                if (q == 0) {
                    float4 o = make_float4(db[0], db[1], db[2], db[3]);
                    if (!first) {
                        const float4 prev = *reinterpret_cast<const float4*>(drow + k4);
                        o.x += prev.x; o.y += prev.y; o.z += prev.z; o.w += prev.w;
                    }
                    *reinterpret_cast<float4*>(drow + k4) = o;
                }
            }
        }
    }
}

int check_common(int64_t N, int Hp) {
    if (N < 0 || Hp <= 0) return EQH_ERR_ARG;
    if (Hp & 63) return EQH_ERR_ALIGN;  // the backward walks 64 hidden units per step
    if (N * 2 * (int64_t)Hp >= ((int64_t)1 << 31) * 4) return EQH_ERR_RANGE;
    if (N * KNB >= ((int64_t)1 << 31)) return EQH_ERR_RANGE;
    return EQH_OK;
}

}  // namespace

extern "C" int egnn_edge_fwd(const float* ab, const float* wd, const float* w2, const float* b2,
                             const int32_t* nbr, const float* d2, int64_t N, int32_t Hp, float* m,
                             float* pre2, void* stream_) {
    int rc = check_common(N, Hp);
    if (rc) return rc;
    if (N == 0) return EQH_OK;
    if (!ab || !wd || !w2 || !b2 || !nbr || !d2 || !m || !pre2) return EQH_ERR_ARG;
    if (!eqh_aligned16(ab) || !eqh_aligned16(wd) || !eqh_aligned16(w2)) return EQH_ERR_ALIGN;
    const size_t lds_rest = ((size_t)Hp + (size_t)(FWD_THREADS / 64) * FWD_TILE) * sizeof(float);
    const size_t lds_w2 = (size_t)MDIM * lds_row_stride(Hp) * sizeof(float);
    const bool w2_lds = lds_rest + lds_w2 <= 160 * 1024;
    const size_t lds = lds_rest + (w2_lds ? lds_w2 : 0);
    if (lds > 160 * 1024) return EQH_ERR_RANGE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_edge_fwd<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_edge_fwd<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return EQH_ERR_LAUNCH;
        attr_set = true;
    }
    const int grid = eqh_grid_for(N, FWD_NODES, 256);  // one 16-wavefront workgroup per CU
    if (w2_lds)
        hipLaunchKernelGGL(k_edge_fwd<true>, dim3(grid), dim3(FWD_THREADS), lds, stream, ab, wd, w2, b2, nbr, d2,
                           m, pre2, (int)N, (int)Hp);
    else
        hipLaunchKernelGGL(k_edge_fwd<false>, dim3(grid), dim3(FWD_THREADS), lds, stream, ab, wd, w2, b2, nbr, d2,
                           m, pre2, (int)N, (int)Hp);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}

extern "C" size_t egnn_edge_bwd_workspace_bytes(int64_t N, int32_t Hp) {
    if (N < 0 || Hp <= 0) return 0;
    const int64_t blocks = (N + BWD_CHUNK - 1) / BWD_CHUNK;
    return (size_t)(blocks > 0 ? blocks : 1) * (size_t)(MDIM + 1) * (size_t)Hp * sizeof(float);
}

extern "C" int egnn_edge_bwd(const float* ab, const float* wd, const float* w2, const int32_t* nbr,
                             const float* d2, const float* pre2, const float* dm,
                             const int32_t* t_rowptr, const int32_t* t_perm, int64_t N, int32_t Hp,
                             float* dab, float* dwd, float* dw2, float* dpre2, void* workspace,
                             size_t workspace_bytes, void* stream_) {
    int rc = check_common(N, Hp);
    if (rc) return rc;
    if (!dwd || !dw2) return EQH_ERR_ARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (N == 0) {
        if (eqh_zero_async(dwd, Hp, stream) || eqh_zero_async(dw2, (int64_t)MDIM * Hp, stream))
            return EQH_ERR_LAUNCH;
        return EQH_OK;
    }
    if (!ab || !wd || !w2 || !nbr || !d2 || !pre2 || !dm || !t_rowptr || !t_perm || !dab || !dpre2 ||
        !workspace)
        return EQH_ERR_ARG;
    if (!eqh_aligned16(pre2) || !eqh_aligned16(dm) || !eqh_aligned16(dpre2) || !eqh_aligned16(workspace))
        return EQH_ERR_ALIGN;
    if (workspace_bytes < egnn_edge_bwd_workspace_bytes(N, Hp)) return EQH_ERR_ARG;
    const int blocks = (int)((N + BWD_CHUNK - 1) / BWD_CHUNK);
    float* slab_w2 = static_cast<float*>(workspace);
    float* slab_wd = slab_w2 + (size_t)blocks * MDIM * Hp;
    hipLaunchKernelGGL(k_edge_bwd_recv, dim3(blocks), dim3(THREADS), 0, stream, ab, wd, w2, nbr, d2, pre2,
                       dm, dpre2, dab, slab_w2, slab_wd, (int)N, (int)Hp);
    EQH_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_edge_bwd_send, dim3(eqh_grid_for(N, WAVES, 2048)), dim3(THREADS), 0, stream, ab,
                       wd, w2, d2, dpre2, t_rowptr, t_perm, dab, (int)N, (int)Hp);
    EQH_CHECK_LAUNCH();
    if (eqh_reduce_slabs_async(slab_w2, blocks, (int64_t)MDIM * Hp, dw2, stream)) return EQH_ERR_LAUNCH;
    return eqh_reduce_slabs_async(slab_wd, blocks, (int64_t)Hp, dwd, stream);
}
