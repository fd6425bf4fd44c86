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
#include <cstdlib>

#include "common.h"
#include "bf16x3.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KNB = 16;    // neighbours per node
constexpr int MDIM = 16;   // m_dim
constexpr int THREADS = 256;
constexpr int WAVES = THREADS / 64;

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

// blockIdx -> work item such that the items handled by one XCD are CONSECUTIVE: blocks are dealt round-robin over the eight
// XCDs (b and b + 8 share one: MI355X_MICROARCH.md, speed only -- nothing depends on it), so with the nodes in spatial order an
// XCD's 4 MB L2 sees a compact region of the cloud and its neighbours.  The grid is rounded up to a multiple of 8; items
// past the end return -1.
__device__ __forceinline__ int xcd_item(int b, int grid, int n_items, int remap) {
    if (!remap) return b < n_items ? b : -1;
    const int per = grid >> 3;
    const int item = (b & 7) * per + (b >> 3);
    return item < n_items ? item : -1;
}
static inline int xcd_grid(int n_items, int remap) { return remap ? ((n_items + 7) / 8) * 8 : n_items; }
// (Measured and not kept for the backward kernels: walking an XCD's chunks TILE-MAJOR -- all resident blocks on one 256-byte
// column block of `ab`, 1.2 MB, L2-sized -- instead of all 17 column blocks at once: recv 96 -> 122 us, send 85 -> 114.  With the
// nodes in Morton order the chunk-major mapping above is worth 7 % on all three kernels (scratch experiment);
// in molecule order, where a node's 16 neighbours lie anywhere in the batch, it is worth 1 %.)

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
// forward, round 5: the 16 x Hp -> 16 contraction on the bf16 matrix cores.
// The fp32 MFMA above shares the VALU's FMA lanes -- a SIMD's time per 16 hidden units is the SUM of the SiLU's VALU cycles and
// 4 x 32 MFMA cycles (~420).  Here the activations s = silu(h) are split EXACTLY into three bf16 planes (bf16x3.h: s = s0 + s1 +
// s2, the split of gemm_x6.hip and the panel kernels) and multiplied with W2's three planes by six v_mfma_f32_16x16x32_bf16 per 32
// hidden units, smallest terms first into one fp32 accumulator: the matrix pipe runs BESIDE the VALU (96 of its cycles per 32
// units against ~300 VALU cycles for the SiLU and the split), so the step costs what its VALU work costs.  Accuracy is that of
// the x6 products (dropped terms 2^-24 relative; tests compare with float64 at the same bounds as before).
// Geometry: 8 wavefronts per workgroup and CU (two per SIMD, 256 registers each), four wavefronts per node, each walking a
// contiguous quarter of the hidden units in steps of 32; a wavefront's W2 fragments (its k-steps x 3 planes x 16 bytes per lane)
// are split ONCE at kernel start and stay in registers; B rows / the A row go row-contiguously through a private LDS tile
// into MFMA operand order, as above (lane (j = lane & 15, kg = lane >> 4) multiplies units 8 kg .. 8 kg + 7 of neighbour j).
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int F2_SPLIT = 4;                              // wavefronts per node
constexpr int F2_LD = 32 + 4;                            // floats per staged row (32 hidden units + pad)
constexpr int F2_TILE = (KNB + 1) * F2_LD;               // 16 B rows + the A row
constexpr int F2_MAXSTEPS = 9;                           // k-steps (of 32 units) per wavefront: Hp <= 4 * 9 * 32 = 1152

__device__ __forceinline__ void split8(const float (&v)[8], uint4& p0, uint4& p1, uint4& p2) {
    split_pair(v[0], v[1], p0.x, p1.x, p2.x);
    split_pair(v[2], v[3], p0.y, p1.y, p2.y);
    split_pair(v[4], v[5], p0.z, p1.z, p2.z);
    split_pair(v[6], v[7], p0.w, p1.w, p2.w);
}

template <int NS, int THREADS_, bool W2_LDS>
__global__ void __launch_bounds__(THREADS_)
k_edge_fwd_x3(const float* __restrict__ ab, const float* __restrict__ wd, const float* __restrict__ w2,
              const float* __restrict__ b2, const int* __restrict__ nbr, const float* __restrict__ d2,
              float* __restrict__ m, float* __restrict__ pre2, int N, int Hp, int n_items, int remap) {
    extern __shared__ __attribute__((aligned(16))) float s_mem[];
    constexpr int NWAVES = THREADS_ / 64, NODES = NWAVES / F2_SPLIT;
    const int item = xcd_item((int)blockIdx.x, (int)gridDim.x, n_items, remap);
    if (item < 0) return;
    const int total = Hp >> 5;                            // k-steps of 32 hidden units
    float* s_wd = s_mem;                                  // [Hp]
    float* s_tile = s_wd + Hp;                            // [wavefronts][F2_TILE]
    uint4* s_w2 = reinterpret_cast<uint4*>(s_tile + NWAVES * F2_TILE);   // W2_LDS: [k-step][plane 3][lane 64] x 16 bytes
    for (int idx = threadIdx.x * 4; idx < Hp; idx += THREADS_ * 4)
        *reinterpret_cast<float4*>(s_wd + idx) = *reinterpret_cast<const float4*>(wd + idx);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int slot = wave / F2_SPLIT, part = wave % F2_SPLIT;
    const int per = (total + F2_SPLIT - 1) / F2_SPLIT;    // <= NS
    const int s_beg = part * per;
    const int cnt = (total - s_beg < per) ? (total - s_beg > 0 ? total - s_beg : 0) : per;
    // W2 fragments: lane (o = r, kg = q) holds W2[o][32 step + 8 q .. + 7], split into planes -- this wavefront's own k-steps in
    // registers (two wavefronts per SIMD), or every k-step once per workgroup in LDS (four per SIMD: 128 registers each)
    auto w2_frag = [&](int step, uint4& p0, uint4& p1, uint4& p2) {
        float v[8];
        const int k0 = 32 * step + 8 * q;
        const float4 x0 = *reinterpret_cast<const float4*>(w2 + r * Hp + k0);
        const float4 x1 = *reinterpret_cast<const float4*>(w2 + r * Hp + k0 + 4);
        v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
        split8(v, p0, p1, p2);
    };
    uint4 wf[W2_LDS ? 1 : NS][3];
    if constexpr (W2_LDS) {
        for (int step = wave; step < total; step += NWAVES) {
            uint4 p0, p1, p2;
            w2_frag(step, p0, p1, p2);
            s_w2[(step * 3 + 0) * 64 + lane] = p0;
            s_w2[(step * 3 + 1) * 64 + lane] = p1;
            s_w2[(step * 3 + 2) * 64 + lane] = p2;
        }
    } else {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s < cnt) w2_frag(s_beg + s, wf[s][0], wf[s][1], wf[s][2]);
            else wf[s][0] = wf[s][1] = wf[s][2] = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    __syncthreads();
    const float bo = b2[r];
    float* tile = s_tile + wave * F2_TILE;
    // staging: two 16-byte pieces per lane and step -- rows (lane >> 3) and 8 + (lane >> 3), piece lane & 7 (a row of 32 units
    // = 128 bytes = eight lanes); the A row by lanes 0..7
    const int lrow = lane >> 3, piece = lane & 7;
    float* t_wr0 = tile + lrow * F2_LD + 4 * piece;
    float* t_wr1 = t_wr0 + 8 * F2_LD;
    float* t_wra = tile + KNB * F2_LD + 4 * piece;
    const float* t_rd = tile + r * F2_LD + 8 * q;         // neighbour r, units 8 q .. 8 q + 7 of the step
    const float* t_rda = tile + KNB * F2_LD + 8 * q;      // the receiver's own A row (broadcast over r)
    const int groups = (N + NODES - 1) / NODES;
    const int gpb = (groups + n_items - 1) / n_items;
    const int g_beg = item * gpb;
    const int g_end = (g_beg + gpb < groups) ? g_beg + gpb : groups;
    if (g_beg >= g_end) return;  // whole workgroup
    const int col0 = 32 * s_beg + 4 * piece;              // float offset of this lane's piece in step 0 of its range
    int node_raw = g_beg * NODES + slot;
    int node = (node_raw < N) ? node_raw : N - 1;  // tail: recompute the last node, not stored
    float dd = d2[node * KNB + r];
    const float* __restrict__ arow = ab + (int64_t)node * 2 * Hp;
    const float* __restrict__ brow0 = ab + (int64_t)nbr[node * KNB + lrow] * 2 * Hp + Hp;
    const float* __restrict__ brow1 = ab + (int64_t)nbr[node * KNB + 8 + lrow] * 2 * Hp + Hp;
    float4 nb0 = f4_zero(), nb1 = f4_zero(), na = f4_zero();
    if (cnt > 0) {
        nb0 = *reinterpret_cast<const float4*>(brow0 + col0);
        nb1 = *reinterpret_cast<const float4*>(brow1 + col0);
        if (lane < 8) na = *reinterpret_cast<const float4*>(arow + col0);
    }
    for (int group = g_beg; group < g_end; ++group) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        int j0 = 0, j1 = 0, node_raw_n = 0, node_n = 0;
        float dd_n = 0.f;
        {   // next group's neighbour list (the last group re-reads its own): in flight behind this group's first steps
            const int gn = (group + 1 < g_end) ? group + 1 : group;
            node_raw_n = gn * NODES + slot;
            node_n = (node_raw_n < N) ? node_raw_n : N - 1;
            j0 = nbr[node_n * KNB + lrow];
            j1 = nbr[node_n * KNB + 8 + lrow];
            dd_n = d2[node_n * KNB + r];
        }
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            if (st < cnt) {
                *reinterpret_cast<float4*>(t_wr0) = nb0;
                *reinterpret_cast<float4*>(t_wr1) = nb1;
                if (lane < 8) *reinterpret_cast<float4*>(t_wra) = na;
                const float* __restrict__ p0 = brow0;
                const float* __restrict__ p1 = brow1;
                const float* __restrict__ pa = arow;
                int bc;
                if (st + 1 < cnt) {                       // next step of this node
                    bc = col0 + 32 * (st + 1);
                } else {                                  // first step of the next group's node
                    asm volatile("" : "+v"(j0), "+v"(j1), "+v"(node_n));   // (keeps the index arithmetic below the branch)
                    p0 = ab + (int64_t)j0 * 2 * Hp + Hp;
                    p1 = ab + (int64_t)j1 * 2 * Hp + Hp;
                    pa = ab + (int64_t)node_n * 2 * Hp;
                    bc = col0;
                }
                nb0 = *reinterpret_cast<const float4*>(p0 + bc);
                nb1 = *reinterpret_cast<const float4*>(p1 + bc);
                if (lane < 8) na = *reinterpret_cast<const float4*>(pa + bc);
                if (st + 1 >= cnt) { brow0 = p0; brow1 = p1; arow = pa; }
                __builtin_amdgcn_sched_barrier(0);
                const float* cr = s_wd + 32 * (s_beg + st) + 8 * q;
                const float4 b0 = *reinterpret_cast<const float4*>(t_rd), b1 = *reinterpret_cast<const float4*>(t_rd + 4);
                const float4 a0 = *reinterpret_cast<const float4*>(t_rda), a1 = *reinterpret_cast<const float4*>(t_rda + 4);
                const float4 c0 = *reinterpret_cast<const float4*>(cr), c1 = *reinterpret_cast<const float4*>(cr + 4);
#ifdef EDGE_ABLATE      // diagnostic build: the gather + LDS round trip without the SiLU / split / MFMA work
                acc[0] += (b0.x + b1.w) + (a0.y + a1.z) + (c0.x + c1.x);
                __builtin_amdgcn_sched_barrier(0);
                continue;
#endif
                const f32x2 dd2 = {dd, dd};
                // (the arithmetic of fwd_stage above: h = c * dd + (a + b), silu by v_exp / v_rcp)
                const f32x2 s01 = silu_fast2(f32x2{c0.x, c0.y} * dd2 + (f32x2{a0.x, a0.y} + f32x2{b0.x, b0.y}));
                const f32x2 s23 = silu_fast2(f32x2{c0.z, c0.w} * dd2 + (f32x2{a0.z, a0.w} + f32x2{b0.z, b0.w}));
                const f32x2 s45 = silu_fast2(f32x2{c1.x, c1.y} * dd2 + (f32x2{a1.x, a1.y} + f32x2{b1.x, b1.y}));
                const f32x2 s67 = silu_fast2(f32x2{c1.z, c1.w} * dd2 + (f32x2{a1.z, a1.w} + f32x2{b1.z, b1.w}));
                uint4 p0_, p1_, p2_;
                split_pair(s01.x, s01.y, p0_.x, p1_.x, p2_.x);
                split_pair(s23.x, s23.y, p0_.y, p1_.y, p2_.y);
                split_pair(s45.x, s45.y, p0_.z, p1_.z, p2_.z);
                split_pair(s67.x, s67.y, p0_.w, p1_.w, p2_.w);
                const bf16x8 x0 = __builtin_bit_cast(bf16x8, p0_), x1 = __builtin_bit_cast(bf16x8, p1_), x2 = __builtin_bit_cast(bf16x8, p2_);
                uint4 q0, q1, q2;
                if constexpr (W2_LDS) {
                    const uint4* wp = s_w2 + (s_beg + st) * 3 * 64 + lane;
                    q0 = wp[0]; q1 = wp[64]; q2 = wp[128];
                } else {
                    q0 = wf[st][0]; q1 = wf[st][1]; q2 = wf[st][2];
                }
                const bf16x8 w0 = __builtin_bit_cast(bf16x8, q0), w1 = __builtin_bit_cast(bf16x8, q1), w2p = __builtin_bit_cast(bf16x8, q2);
                // smallest terms first, as gemm_x6.hip: s1 w1, s0 w2, s2 w0, s0 w1, s1 w0, s0 w0
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x1, w1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x0, w2p, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x2, w0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x0, w1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x1, w0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x0, w0, acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // partial accumulators: wavefronts 1..3 of the node park theirs in their own (now idle) tile
        if (part > 0) *reinterpret_cast<f32x4*>(tile + lane * 4) = acc;
        __syncthreads();
        if (part == 0 && node_raw < N) {
#pragma unroll
            for (int pp = 1; pp < F2_SPLIT; ++pp)
                acc += *reinterpret_cast<const f32x4*>(tile + pp * F2_TILE + lane * 4);
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
        __syncthreads();  // tiles are overwritten by the next group's first step
        node_raw = node_raw_n;
        node = node_n;
        dd = dd_n;
    }
}

// ------------------------------------------------------------------------------------------------
// backward.  Three kernels, no atomics:
//   prep : dpre2 = dm * silu'(pre2), stored twice ([n][j][o] and [n][o][j]: the two MFMA operand orders)
//   recv : by receiver i  -> dA (= dab[:, :Hp]), per-chunk partial slabs of dW2 and dwd
//   send : by sender j, over the transposed neighbour CSR -> dB (= dab[:, Hp:])
// Both main kernels recompute h = A_i + B_j + wd d2_ij and silu'(h) (the forward saves no per-edge
// tensor) and use the same decomposition: grid = (64-unit hidden tile) x (chunk of nodes / senders),
// 4 wavefronts per workgroup, every wavefront walking its share of the chunk for ONE hidden tile.  All
// wavefronts of the launch have the same amount of work whatever Hp / 64 is (the earlier version gave a
// wavefront all nodes of a 16-node chunk for hidden tiles w, w+4, ...: 17 tiles over 4 wavefronts and
// 288 workgroups over 256 CUs lost ~45% to quantisation), and the chunk count is chosen so that the
// whole grid is resident at once (4 workgroups per CU).
// ------------------------------------------------------------------------------------------------
constexpr int BW_RESIDENT = 768;  // 256 CUs x 3 workgroups of 4 wavefronts (129..168 VGPRs per lane)

__host__ __device__ inline int bw_chunks(int64_t n_items, int tiles) {
    int64_t c = BW_RESIDENT / tiles;
    if (c < 1) c = 1;
    const int64_t most = (n_items + WAVES - 1) / WAVES;  // at least one item per wavefront
    if (c > most) c = most;
    return (int)(c < 1 ? 1 : c);
}

__global__ void __launch_bounds__(THREADS)
k_edge_bwd_prep(const float* __restrict__ pre2, const float* __restrict__ dm, float* __restrict__ dpre2,
                float* __restrict__ dpre2_t, const int* __restrict__ t_perm, const int* __restrict__ nbr,
                const float* __restrict__ d2, int4* __restrict__ rec, float* __restrict__ slab_b2, int N, int64_t dm_ld) {
    // records for the sender pass: one 16-byte record per position p of the transposed CSR, (entry
    // e = i*16 + slot, sender j = nbr[e], d2[e], -), so that its pipeline has no dependent load chain
    // (t_perm -> nbr / d2) in front of the row gathers
    const int64_t E = (int64_t)N * KNB;
    for (int64_t p = (int64_t)blockIdx.x * THREADS + threadIdx.x; p < E; p += (int64_t)gridDim.x * THREADS) {
        const int e = t_perm[p];
        rec[p] = make_int4(e, nbr[e], __float_as_int(d2[e]), 0);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 acc_b2 = f4_zero();   // sum over this wavefront's nodes of dpre2[node][j = lane >> 2][o0 .. o0 + 3]
    for (int node = blockIdx.x * WAVES + wave; node < N; node += gridDim.x * WAVES) {
        const int jj = lane >> 2, o0 = (lane & 3) * 4;  // one float4 (4 consecutive o of one j) per lane
        const float4 p4 = *reinterpret_cast<const float4*>(pre2 + (int64_t)node * 256 + lane * 4);
        const float4 g4 = *reinterpret_cast<const float4*>(dm + (int64_t)node * dm_ld + o0);
        float ds;
        float4 d4;
        silu_grad(p4.x, &ds); d4.x = g4.x * ds;
        silu_grad(p4.y, &ds); d4.y = g4.y * ds;
        silu_grad(p4.z, &ds); d4.z = g4.z * ds;
        silu_grad(p4.w, &ds); d4.w = g4.w * ds;
        *reinterpret_cast<float4*>(dpre2 + (int64_t)node * 256 + lane * 4) = d4;
        float* __restrict__ t = dpre2_t + (int64_t)node * 256 + jj;
        t[(o0 + 0) * KNB] = d4.x;
        t[(o0 + 1) * KNB] = d4.y;
        t[(o0 + 2) * KNB] = d4.z;
        t[(o0 + 3) * KNB] = d4.w;
        f4_add(acc_b2, d4);
    }
    if (slab_b2) {   // d b2[o] = sum over nodes and slots j of dpre2: lanes sharing (lane & 3), then the wavefronts
#pragma unroll
        for (int off = 4; off < 64; off <<= 1) {
            acc_b2.x += __shfl_xor(acc_b2.x, off); acc_b2.y += __shfl_xor(acc_b2.y, off);
            acc_b2.z += __shfl_xor(acc_b2.z, off); acc_b2.w += __shfl_xor(acc_b2.w, off);
        }
        __shared__ float4 s_b2[WAVES][4];
        if (lane < 4) s_b2[wave][lane] = acc_b2;
        __syncthreads();
        if (threadIdx.x < 4) {
            float4 v = s_b2[0][threadIdx.x];
            for (int w = 1; w < WAVES; ++w) f4_add(v, s_b2[w][threadIdx.x]);
            *reinterpret_cast<float4*>(slab_b2 + (int64_t)blockIdx.x * MDIM + 4 * threadIdx.x) = v;
        }
    }
}

// Per-node operands of the receiver pass, as named scalars (see the forward: no arrays across branches).
struct RecvOps {
    float4 pa;   // dpre2[node][j = r][o = 4q .. 4q+3]
    float4 pt;   // dpre2[node][j = 4q .. 4q+3][o = r]   (from the transposed copy)
    float4 a;    // A_i[k4 .. k4+3]
    float4 dd;   // d2[node][4q .. 4q+3]
    float4 b0, b1, b2, b3;  // B_j[k4 .. k4+3] for the neighbours j = nbr[node][4q + g]
};

__device__ __forceinline__ void recv_load(RecvOps& o, const float* __restrict__ ab, const float* __restrict__ d2,
                                          const float* __restrict__ dpre2, const float* __restrict__ dpre2_t,
                                          int node, int4 nb, int Hp, int k4, int r, int q) {
    o.pa = *reinterpret_cast<const float4*>(dpre2 + (int64_t)node * 256 + r * MDIM + 4 * q);
    o.pt = *reinterpret_cast<const float4*>(dpre2_t + (int64_t)node * 256 + r * KNB + 4 * q);
    o.a = *reinterpret_cast<const float4*>(ab + (int64_t)node * 2 * Hp + k4);
    o.dd = *reinterpret_cast<const float4*>(d2 + (int64_t)node * KNB + 4 * q);
    o.b0 = *reinterpret_cast<const float4*>(ab + (int64_t)nb.x * 2 * Hp + Hp + k4);
    o.b1 = *reinterpret_cast<const float4*>(ab + (int64_t)nb.y * 2 * Hp + Hp + k4);
    o.b2 = *reinterpret_cast<const float4*>(ab + (int64_t)nb.z * 2 * Hp + Hp + k4);
    o.b3 = *reinterpret_cast<const float4*>(ab + (int64_t)nb.w * 2 * Hp + Hp + k4);
}

// One hidden unit column c (of the lane's four) of one node: the four edges g = 0..3 of this lane.
// Returns sum_g dh, accumulates dwd, and feeds dW2[o][k4 + c] += sum_j dpre2[j][o] * silu(h[j][k4 + c]).
__device__ __forceinline__ float recv_column(float a, float wdc, const float4& dd, float b0, float b1, float b2,
                                             float b3, const f32x4& gk, const float4& pt, float& acc_wd,
                                             f32x4& acc_w) {
    float ds0, ds1, ds2, ds3;
    const float s0 = silu_grad(fmaf(wdc, dd.x, a + b0), &ds0);
    const float s1 = silu_grad(fmaf(wdc, dd.y, a + b1), &ds1);
    const float s2 = silu_grad(fmaf(wdc, dd.z, a + b2), &ds2);
    const float s3 = silu_grad(fmaf(wdc, dd.w, a + b3), &ds3);
    const float dh0 = gk[0] * ds0, dh1 = gk[1] * ds1, dh2 = gk[2] * ds2, dh3 = gk[3] * ds3;
    acc_wd = fmaf(dh0, dd.x, acc_wd);
    acc_wd = fmaf(dh1, dd.y, acc_wd);
    acc_wd = fmaf(dh2, dd.z, acc_wd);
    acc_wd = fmaf(dh3, dd.w, acc_wd);
    acc_w = mfma16(pt.x, s0, acc_w);
    acc_w = mfma16(pt.y, s1, acc_w);
    acc_w = mfma16(pt.z, s2, acc_w);
    acc_w = mfma16(pt.w, s3, acc_w);
    return (dh0 + dh1) + (dh2 + dh3);
}

__global__ void __launch_bounds__(THREADS)
k_edge_bwd_recv(const float* __restrict__ ab, const float* __restrict__ wd, const float* __restrict__ w2,
                const int* __restrict__ nbr, const float* __restrict__ d2, const float* __restrict__ dpre2,
                const float* __restrict__ dpre2_t, float* __restrict__ dab, float* __restrict__ slab_w2,
                float* __restrict__ slab_wd, int N, int Hp, int tiles, int chunk_nodes, int n_items, int remap) {
    __shared__ __attribute__((aligned(16))) float s_red[WAVES - 1][64 * 20];  // 16 dW2 + 4 dwd floats per lane
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int item = xcd_item((int)blockIdx.x, (int)gridDim.x, n_items, remap);
    if (item < 0) return;
    const int T = item % tiles, chunk = item / tiles;
    const int base = chunk * chunk_nodes;
    const int cnt = (N - base < chunk_nodes) ? (N - base) : chunk_nodes;
    // lane (r, q) owns the FOUR consecutive hidden units k4 .. k4+3, k4 = 64 T + 4 r (column n = r of MFMA
    // tile c is unit k4 + c), so every operand is one float4 and a quarter-wavefront reads 256 contiguous bytes
    const int k4 = 64 * T + 4 * r;
    const float4 w0 = *reinterpret_cast<const float4*>(w2 + (4 * q + 0) * Hp + k4);
    const float4 w1 = *reinterpret_cast<const float4*>(w2 + (4 * q + 1) * Hp + k4);
    const float4 w2v = *reinterpret_cast<const float4*>(w2 + (4 * q + 2) * Hp + k4);
    const float4 w3 = *reinterpret_cast<const float4*>(w2 + (4 * q + 3) * Hp + k4);
    const float4 wd4 = *reinterpret_cast<const float4*>(wd + k4);
    f32x4 aw0 = {0.f, 0.f, 0.f, 0.f}, aw1 = aw0, aw2 = aw0, aw3 = aw0;  // aw_c[g] = dW2[o = 4q+g][k4 + c]
    float ad0 = 0.f, ad1 = 0.f, ad2 = 0.f, ad3 = 0.f;                   // dwd[k4 + c], this lane's edges
    // software pipeline over this wavefront's nodes n = wave, wave + 4, ...: operands of node n+1 and the
    // neighbour list of node n+2 are in flight while node n is computed
    const int last = cnt - 1;
    int n = wave;
    if (n < cnt) {
        const int4* __restrict__ nbr4 = reinterpret_cast<const int4*>(nbr);
        auto clampn = [&](int x) { return base + (x < last ? x : last); };
        int4 nb_next = nbr4[(int64_t)clampn(n) * 4 + q];
        RecvOps nx;
        recv_load(nx, ab, d2, dpre2, dpre2_t, clampn(n), nb_next, Hp, k4, r, q);
        nb_next = nbr4[(int64_t)clampn(n + WAVES) * 4 + q];
        for (; n < cnt; n += WAVES) {
            const RecvOps cur = nx;
            const int node = base + n;
            recv_load(nx, ab, d2, dpre2, dpre2_t, clampn(n + WAVES), nb_next, Hp, k4, r, q);
            nb_next = nbr4[(int64_t)clampn(n + 2 * WAVES) * 4 + q];
            __builtin_amdgcn_sched_barrier(0);
#ifdef EDGE_ABLATE      // diagnostic build: operand traffic of the receiver pass without its MFMA / SiLU work
            {
                const float t = (cur.pa.x + cur.pt.y) + (cur.a.x + cur.dd.y) + ((cur.b0.x + cur.b1.y) + (cur.b2.z + cur.b3.w));
                if (q == 0) *reinterpret_cast<float4*>(dab + (int64_t)node * 2 * Hp + k4) = make_float4(t, t, t, t);
                ad0 += t;
                continue;
            }
#endif
            // gk_c[g] = sum_o dpre2[j = 4q+g][o] * W2[o][k4 + c]
            f32x4 g0 = {0.f, 0.f, 0.f, 0.f}, g1 = g0, g2 = g0, g3 = g0;
            g0 = mfma16(cur.pa.x, w0.x, g0); g1 = mfma16(cur.pa.x, w0.y, g1);
            g2 = mfma16(cur.pa.x, w0.z, g2); g3 = mfma16(cur.pa.x, w0.w, g3);
            g0 = mfma16(cur.pa.y, w1.x, g0); g1 = mfma16(cur.pa.y, w1.y, g1);
            g2 = mfma16(cur.pa.y, w1.z, g2); g3 = mfma16(cur.pa.y, w1.w, g3);
            g0 = mfma16(cur.pa.z, w2v.x, g0); g1 = mfma16(cur.pa.z, w2v.y, g1);
            g2 = mfma16(cur.pa.z, w2v.z, g2); g3 = mfma16(cur.pa.z, w2v.w, g3);
            g0 = mfma16(cur.pa.w, w3.x, g0); g1 = mfma16(cur.pa.w, w3.y, g1);
            g2 = mfma16(cur.pa.w, w3.z, g2); g3 = mfma16(cur.pa.w, w3.w, g3);
            float da0 = recv_column(cur.a.x, wd4.x, cur.dd, cur.b0.x, cur.b1.x, cur.b2.x, cur.b3.x, g0, cur.pt, ad0, aw0);
            float da1 = recv_column(cur.a.y, wd4.y, cur.dd, cur.b0.y, cur.b1.y, cur.b2.y, cur.b3.y, g1, cur.pt, ad1, aw1);
            float da2 = recv_column(cur.a.z, wd4.z, cur.dd, cur.b0.z, cur.b1.z, cur.b2.z, cur.b3.z, g2, cur.pt, ad2, aw2);
            float da3 = recv_column(cur.a.w, wd4.w, cur.dd, cur.b0.w, cur.b1.w, cur.b2.w, cur.b3.w, g3, cur.pt, ad3, aw3);
            da0 += __shfl_xor(da0, 16, 64); da1 += __shfl_xor(da1, 16, 64);
            da2 += __shfl_xor(da2, 16, 64); da3 += __shfl_xor(da3, 16, 64);
            da0 += __shfl_xor(da0, 32, 64); da1 += __shfl_xor(da1, 32, 64);
            da2 += __shfl_xor(da2, 32, 64); da3 += __shfl_xor(da3, 32, 64);
            if (q == 0)
                *reinterpret_cast<float4*>(dab + (int64_t)node * 2 * Hp + k4) = make_float4(da0, da1, da2, da3);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // the four wavefronts' partial dW2 / dwd tiles meet in LDS and are added in wavefront order
    ad0 += __shfl_xor(ad0, 16, 64); ad1 += __shfl_xor(ad1, 16, 64);
    ad2 += __shfl_xor(ad2, 16, 64); ad3 += __shfl_xor(ad3, 16, 64);
    ad0 += __shfl_xor(ad0, 32, 64); ad1 += __shfl_xor(ad1, 32, 64);
    ad2 += __shfl_xor(ad2, 32, 64); ad3 += __shfl_xor(ad3, 32, 64);
    if (wave > 0) {
        float* p = &s_red[wave - 1][lane * 20];
        *reinterpret_cast<f32x4*>(p) = aw0;
        *reinterpret_cast<f32x4*>(p + 4) = aw1;
        *reinterpret_cast<f32x4*>(p + 8) = aw2;
        *reinterpret_cast<f32x4*>(p + 12) = aw3;
        *reinterpret_cast<float4*>(p + 16) = make_float4(ad0, ad1, ad2, ad3);
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int w = 0; w < WAVES - 1; ++w) {
            const float* p = &s_red[w][lane * 20];
            aw0 += *reinterpret_cast<const f32x4*>(p);
            aw1 += *reinterpret_cast<const f32x4*>(p + 4);
            aw2 += *reinterpret_cast<const f32x4*>(p + 8);
            aw3 += *reinterpret_cast<const f32x4*>(p + 12);
            const float4 d = *reinterpret_cast<const float4*>(p + 16);
            ad0 += d.x; ad1 += d.y; ad2 += d.z; ad3 += d.w;
        }
        float* __restrict__ sw = slab_w2 + (int64_t)chunk * (MDIM + 1) * Hp;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(sw + (4 * q + g) * Hp + k4) = make_float4(aw0[g], aw1[g], aw2[g], aw3[g]);
        if (q == 0)
            *reinterpret_cast<float4*>(slab_wd + (int64_t)chunk * (MDIM + 1) * Hp + k4) = make_float4(ad0, ad1, ad2, ad3);
    }
}

// ------------------------------------------------------------------------------------------------
// sender pass.  dB[j][k] = sum over in-edges e = (i -> j) of g_e[k] * silu'(h_e[k]).  The entries of the
// transposed neighbour CSR (sorted by sender) are consumed 16 at a time as ONE MFMA tile regardless of
// sender boundaries -- the earlier one-sender-per-tile version padded every in-degree to a multiple of 16
// (+45% work on a kNN graph) -- and the 16 x 64 tile of dh is summed per sender by a segmented walk:
// the tile goes through LDS, lane u owns hidden unit u and adds rows while the (wavefront-uniform) sender
// stays the same, storing a finished sender's 256-byte row piece exactly once.
// ------------------------------------------------------------------------------------------------
constexpr int SEND_LD = 68;  // floats per staged row: 64 + 4 pad

struct SendOps {
    float4 pa;                       // dpre2[e_r][4q .. 4q+3] (MFMA row r)
    float4 a0, a1, a2, a3;           // A_i[k4 .. k4+3] of the receivers of MFMA rows 4q + g
    float4 b0, b1, b2, b3;           // B_j[k4 .. k4+3] of their senders
    float dd0, dd1, dd2, dd3;
};

__device__ __forceinline__ void send_load(SendOps& o, const float* __restrict__ ab,
                                          const float* __restrict__ dpre2, int4 rc, int Hp, int k4, int q) {
    o.pa = *reinterpret_cast<const float4*>(dpre2 + (int64_t)rc.x * MDIM + 4 * q);
    const int e0 = __shfl(rc.x, 4 * q + 0, 64), e1 = __shfl(rc.x, 4 * q + 1, 64);
    const int e2 = __shfl(rc.x, 4 * q + 2, 64), e3 = __shfl(rc.x, 4 * q + 3, 64);
    const int j0 = __shfl(rc.y, 4 * q + 0, 64), j1 = __shfl(rc.y, 4 * q + 1, 64);
    const int j2 = __shfl(rc.y, 4 * q + 2, 64), j3 = __shfl(rc.y, 4 * q + 3, 64);
    o.dd0 = __int_as_float(__shfl(rc.z, 4 * q + 0, 64));
    o.dd1 = __int_as_float(__shfl(rc.z, 4 * q + 1, 64));
    o.dd2 = __int_as_float(__shfl(rc.z, 4 * q + 2, 64));
    o.dd3 = __int_as_float(__shfl(rc.z, 4 * q + 3, 64));
    o.a0 = *reinterpret_cast<const float4*>(ab + (int64_t)(e0 >> 4) * 2 * Hp + k4);
    o.a1 = *reinterpret_cast<const float4*>(ab + (int64_t)(e1 >> 4) * 2 * Hp + k4);
    o.a2 = *reinterpret_cast<const float4*>(ab + (int64_t)(e2 >> 4) * 2 * Hp + k4);
    o.a3 = *reinterpret_cast<const float4*>(ab + (int64_t)(e3 >> 4) * 2 * Hp + k4);
    o.b0 = *reinterpret_cast<const float4*>(ab + (int64_t)j0 * 2 * Hp + Hp + k4);
    o.b1 = *reinterpret_cast<const float4*>(ab + (int64_t)j1 * 2 * Hp + Hp + k4);
    o.b2 = *reinterpret_cast<const float4*>(ab + (int64_t)j2 * 2 * Hp + Hp + k4);
    o.b3 = *reinterpret_cast<const float4*>(ab + (int64_t)j3 * 2 * Hp + Hp + k4);
}

__device__ __forceinline__ float4 send_row(const float4& wd4, float dd, const float4& a4, const float4& b4,
                                           float gk0, float gk1, float gk2, float gk3) {
    float ds;
    float4 dh;
    silu_grad(fmaf(wd4.x, dd, a4.x + b4.x), &ds); dh.x = gk0 * ds;
    silu_grad(fmaf(wd4.y, dd, a4.y + b4.y), &ds); dh.y = gk1 * ds;
    silu_grad(fmaf(wd4.z, dd, a4.z + b4.z), &ds); dh.z = gk2 * ds;
    silu_grad(fmaf(wd4.w, dd, a4.w + b4.w), &ds); dh.w = gk3 * ds;
    return dh;
}

__global__ void __launch_bounds__(THREADS)
k_edge_bwd_send(const float* __restrict__ ab, const float* __restrict__ wd, const float* __restrict__ w2,
                const float* __restrict__ dpre2, const int* __restrict__ t_rowptr, const int4* __restrict__ rec,
                float* __restrict__ dab, int N, int Hp, int tiles, int chunk_senders, int n_items, int remap) {
    __shared__ __attribute__((aligned(16))) float s_tile[WAVES][KNB * SEND_LD];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int item = xcd_item((int)blockIdx.x, (int)gridDim.x, n_items, remap);
    if (item < 0) return;
    const int T = item % tiles, chunk = item / tiles;
    const int k4 = 64 * T + 4 * r;
    // this wavefront's senders [s_beg, s_end): an equal share of the chunk
    const int c_beg = chunk * chunk_senders;
    const int c_cnt = (N - c_beg < chunk_senders) ? (N - c_beg) : chunk_senders;
    const int per = (c_cnt + WAVES - 1) / WAVES;
    const int s_beg = c_beg + ((wave * per < c_cnt) ? wave * per : c_cnt);
    const int s_end = c_beg + (((wave + 1) * per < c_cnt) ? (wave + 1) * per : c_cnt);
    if (s_beg >= s_end) return;
    const int p_beg = t_rowptr[s_beg], p_end = t_rowptr[s_end];
    float* drow = dab + Hp + 64 * T + lane;  // + sender * 2 Hp: this lane's hidden unit of dB
    if (p_beg == p_end) {  // no in-edge in the whole range
        for (int sj = s_beg; sj < s_end; ++sj) drow[(int64_t)sj * 2 * Hp] = 0.f;
        return;
    }
    const float4 w0 = *reinterpret_cast<const float4*>(w2 + (4 * q + 0) * Hp + k4);
    const float4 w1 = *reinterpret_cast<const float4*>(w2 + (4 * q + 1) * Hp + k4);
    const float4 w2v = *reinterpret_cast<const float4*>(w2 + (4 * q + 2) * Hp + k4);
    const float4 w3 = *reinterpret_cast<const float4*>(w2 + (4 * q + 3) * Hp + k4);
    const float4 wd4 = *reinterpret_cast<const float4*>(wd + k4);
    float* tile = s_tile[wave];
    // end positions of up to 64 consecutive senders live in one VGPR (lane i: sender rp_base + i),
    // read with v_readlane as the walk advances
    int rp_base = s_beg;
    int rp = t_rowptr[((rp_base + lane < N) ? rp_base + lane : N - 1) + 1];
    int cur = s_beg;                           // sender whose sum is being built
    int cur_end = __builtin_amdgcn_readlane(rp, 0);  // one past its last entry position
    float run = 0.f;
    // pipeline: records two tiles ahead, row gathers one tile ahead (positions clamped into the range; the
    // rows of a ragged last tile are made inert by zeroing their dpre2 operand)
    const int p_last = p_end - 1;
    auto pos = [&](int p) { return (p + r < p_last) ? p + r : p_last; };
    int4 rc_next = rec[pos(p_beg)];
    SendOps nx;
    send_load(nx, ab, dpre2, rc_next, Hp, k4, q);
    rc_next = rec[pos(p_beg + KNB)];
    for (int p0 = p_beg; p0 < p_end; p0 += KNB) {
        const SendOps cur_ops = nx;
        send_load(nx, ab, dpre2, rc_next, Hp, k4, q);
        rc_next = rec[pos(p0 + 2 * KNB)];
        __builtin_amdgcn_sched_barrier(0);
        const bool live_r = p0 + r < p_end;
        const float4 pa = live_r ? cur_ops.pa : make_float4(0.f, 0.f, 0.f, 0.f);
#ifdef EDGE_ABLATE      // diagnostic build: operand traffic of the sender pass without its MFMA / SiLU work and segmented walk
        {
            run += (pa.x + cur_ops.dd0) + ((cur_ops.a0.x + cur_ops.a1.y) + (cur_ops.a2.z + cur_ops.a3.w)) +
                   ((cur_ops.b0.x + cur_ops.b1.y) + (cur_ops.b2.z + cur_ops.b3.w));
            if ((p0 & 255) == 0) drow[(int64_t)cur * 2 * Hp] = run;
            continue;
        }
#endif
        f32x4 g0 = {0.f, 0.f, 0.f, 0.f}, g1 = g0, g2 = g0, g3 = g0;
        g0 = mfma16(pa.x, w0.x, g0); g1 = mfma16(pa.x, w0.y, g1); g2 = mfma16(pa.x, w0.z, g2); g3 = mfma16(pa.x, w0.w, g3);
        g0 = mfma16(pa.y, w1.x, g0); g1 = mfma16(pa.y, w1.y, g1); g2 = mfma16(pa.y, w1.z, g2); g3 = mfma16(pa.y, w1.w, g3);
        g0 = mfma16(pa.z, w2v.x, g0); g1 = mfma16(pa.z, w2v.y, g1); g2 = mfma16(pa.z, w2v.z, g2); g3 = mfma16(pa.z, w2v.w, g3);
        g0 = mfma16(pa.w, w3.x, g0); g1 = mfma16(pa.w, w3.y, g1); g2 = mfma16(pa.w, w3.z, g2); g3 = mfma16(pa.w, w3.w, g3);
        // dh of the four entries whose results land in this lane (MFMA rows 4q + g); gk = 0 on padding rows
        float* trow = tile + 4 * q * SEND_LD + 4 * r;
        *reinterpret_cast<float4*>(trow) = send_row(wd4, cur_ops.dd0, cur_ops.a0, cur_ops.b0, g0[0], g1[0], g2[0], g3[0]);
        *reinterpret_cast<float4*>(trow + SEND_LD) = send_row(wd4, cur_ops.dd1, cur_ops.a1, cur_ops.b1, g0[1], g1[1], g2[1], g3[1]);
        *reinterpret_cast<float4*>(trow + 2 * SEND_LD) = send_row(wd4, cur_ops.dd2, cur_ops.a2, cur_ops.b2, g0[2], g1[2], g2[2], g3[2]);
        *reinterpret_cast<float4*>(trow + 3 * SEND_LD) = send_row(wd4, cur_ops.dd3, cur_ops.a3, cur_ops.b3, g0[3], g1[3], g2[3], g3[3]);
        // segmented walk over the 16 rows (the same wavefront wrote the tile; LDS ops are in order)
        float v[KNB];
#pragma unroll
        for (int row = 0; row < KNB; ++row) v[row] = tile[row * SEND_LD + lane];
        const int rows = (p_end - p0 < KNB) ? (p_end - p0) : KNB;
#pragma unroll
        for (int row = 0; row < KNB; ++row) {
            if (row < rows) {
                while (p0 + row >= cur_end) {  // sender `cur` is complete (possibly without any entry)
                    drow[(int64_t)cur * 2 * Hp] = run;
                    run = 0.f;
                    ++cur;
                    if (cur - rp_base >= 64) {
                        rp_base += 64;
                        rp = t_rowptr[((rp_base + lane < N) ? rp_base + lane : N - 1) + 1];
                    }
                    cur_end = __builtin_amdgcn_readlane(rp, cur - rp_base);
                }
                run += v[row];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    for (; cur < s_end; ++cur) {  // the last sender with entries, then trailing senders without any
        drow[(int64_t)cur * 2 * Hp] = run;
        run = 0.f;
    }
}

// EQH_EDGE_XCD=0 switches the XCD-aware block -> work mapping off (same-box A/B runs)
inline int edge_xcd_remap() {
    static const int v = [] { const char* e = std::getenv("EQH_EDGE_XCD"); return (e && e[0] == '0') ? 0 : 1; }();
    return v;
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
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    // round 5: the bf16 x 3 kernel (k_edge_fwd_x3) wherever a wavefront's quarter of the hidden units is <= 9 k-steps
    // (Hp <= 1152: every MLP_hidden <= 256); EQH_EDGE_F32=1 keeps the fp32-MFMA kernel for same-box A/B runs
    static const bool use_f32 = [] { const char* e = std::getenv("EQH_EDGE_F32"); return e && e[0] == '1'; }();
    const int per = ((Hp >> 5) + F2_SPLIT - 1) / F2_SPLIT;
    if (!use_f32 && per <= F2_MAXSTEPS && (size_t)N * 2 * (size_t)Hp < ((size_t)1 << 31)) {
        // 16 wavefronts (four per SIMD, W2's planes in LDS) by default; EQH_EDGE_W8=1: 8 wavefronts with W2 in registers
        static const bool w8 = [] { const char* e = std::getenv("EQH_EDGE_W8"); return e && e[0] == '1'; }();
        const int threads = w8 ? 512 : 1024;
        const size_t lds2 = ((size_t)Hp + (size_t)(threads / 64) * F2_TILE) * sizeof(float) + (w8 ? 0 : (size_t)(Hp >> 5) * 3 * 64 * 16);
        const int items2 = eqh_grid_for(N, threads / 64 / F2_SPLIT, 256);  // one workgroup per CU
        const int remap = edge_xcd_remap();
        const int grid2 = xcd_grid(items2, remap);
        static bool attr2 = false;
#define F2_ATTR(NS_)                                                                                                          \
        (hipFuncSetAttribute(reinterpret_cast<const void*>(k_edge_fwd_x3<NS_, 1024, true>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                             160 * 1024) == hipSuccess)
        if (!attr2) {
            if (!(F2_ATTR(1) && F2_ATTR(2) && F2_ATTR(3) && F2_ATTR(5) && F2_ATTR(9))) return EQH_ERR_LAUNCH;
            attr2 = true;
        }
#undef F2_ATTR
        if (lds2 > 160 * 1024) return EQH_ERR_RANGE;
#define F2_LAUNCH(NS_)                                                                                                         \
        do {                                                                                                                   \
            if (w8) hipLaunchKernelGGL((k_edge_fwd_x3<NS_, 512, false>), dim3(grid2), dim3(512), lds2, stream, ab, wd, w2, b2, nbr, d2, m, \
                                       pre2, (int)N, (int)Hp, items2, remap);                                                  \
            else hipLaunchKernelGGL((k_edge_fwd_x3<NS_, 1024, true>), dim3(grid2), dim3(1024), lds2, stream, ab, wd, w2, b2, nbr, d2, m, \
                                    pre2, (int)N, (int)Hp, items2, remap);                                                     \
        } while (0)
        if (per <= 1) F2_LAUNCH(1);
        else if (per <= 2) F2_LAUNCH(2);
        else if (per <= 3) F2_LAUNCH(3);
        else if (per <= 5) F2_LAUNCH(5);
        else F2_LAUNCH(9);
#undef F2_LAUNCH
        EQH_CHECK_LAUNCH();
        return EQH_OK;
    }
    const size_t lds_rest = ((size_t)Hp + (size_t)(FWD_THREADS / 64) * FWD_TILE) * sizeof(float);
    const size_t lds_w2 = (size_t)MDIM * lds_row_stride(Hp) * sizeof(float);
    const bool w2_lds = lds_rest + lds_w2 <= 160 * 1024;
    const size_t lds = lds_rest + (w2_lds ? lds_w2 : 0);
    if (lds > 160 * 1024) return EQH_ERR_RANGE;
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

static inline int prep_blocks(int64_t N) { return eqh_grid_for(N, WAVES, 2048); }

extern "C" size_t egnn_edge_bwd_workspace_bytes(int64_t N, int32_t Hp) {
    if (N < 0 || Hp <= 0) return 0;
    const int chunks = bw_chunks(N, Hp >> 6);
    // [chunks][16 + 1][Hp] partial slabs of dW2 / dwd, the transposed copy of dpre2 [N][16][16] and the
    // 16-byte records of the transposed CSR [16 N]
    // ... and the per-workgroup partial sums of d b2 [prep blocks][16]
    return ((size_t)chunks * (size_t)(MDIM + 1) * (size_t)Hp + (size_t)(N > 0 ? N : 1) * 256 +
            (size_t)(N > 0 ? N : 1) * KNB * 4 + (size_t)prep_blocks(N) * MDIM) * sizeof(float);
}

extern "C" int egnn_edge_bwd(const float* ab, const float* wd, const float* w2, const int32_t* nbr,
                             const float* d2, const float* pre2, const float* dm, int64_t dm_ld,
                             const int32_t* t_rowptr, const int32_t* t_perm, int64_t N, int32_t Hp,
                             float* dab, float* dwd, float* dw2, float* dpre2, float* db2,
                             int32_t db2_accumulate, int32_t dw_accumulate, void* workspace, size_t workspace_bytes,
                             void* stream_) {
    int rc = check_common(N, Hp);
    if (rc) return rc;
    if (!dwd || !dw2) return EQH_ERR_ARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (N == 0) {
        if (!dw_accumulate && (eqh_zero_async(dwd, Hp, stream) || eqh_zero_async(dw2, (int64_t)MDIM * Hp, stream)))
            return EQH_ERR_LAUNCH;
        if (db2 && !db2_accumulate && eqh_zero_async(db2, MDIM, stream)) return EQH_ERR_LAUNCH;
        return EQH_OK;
    }
    if (!ab || !wd || !w2 || !nbr || !d2 || !pre2 || !dm || !t_rowptr || !t_perm || !dab || !dpre2 ||
        !workspace)
        return EQH_ERR_ARG;
    if (!eqh_aligned16(ab) || !eqh_aligned16(wd) || !eqh_aligned16(w2) || !eqh_aligned16(nbr) ||
        !eqh_aligned16(d2) || !eqh_aligned16(pre2) || !eqh_aligned16(dm) || !eqh_aligned16(dpre2) ||
        !eqh_aligned16(dab) || !eqh_aligned16(workspace))
        return EQH_ERR_ALIGN;
    if (workspace_bytes < egnn_edge_bwd_workspace_bytes(N, Hp)) return EQH_ERR_ARG;
    if (dm_ld < MDIM || (dm_ld & 3)) return EQH_ERR_ARG;
    const int tiles = Hp >> 6;
    const int chunks = bw_chunks(N, tiles);
    const int chunk_items = (int)((N + chunks - 1) / chunks);
    const int used = (int)((N + chunk_items - 1) / chunk_items);  // chunks that own at least one item
    float* slab_w2 = static_cast<float*>(workspace);  // per chunk: [16][Hp] of dW2, then [Hp] of dwd
    float* slab_wd = slab_w2 + (size_t)MDIM * Hp;
    float* dpre2_t = slab_w2 + (size_t)chunks * (MDIM + 1) * Hp;
    int4* rec = reinterpret_cast<int4*>(dpre2_t + (size_t)N * 256);
    float* slab_b2 = db2 ? reinterpret_cast<float*>(rec + (size_t)N * KNB) : nullptr;
    const int pblocks = prep_blocks(N);
    hipLaunchKernelGGL(k_edge_bwd_prep, dim3(pblocks), dim3(THREADS), 0, stream, pre2, dm,
                       dpre2, dpre2_t, t_perm, nbr, d2, rec, slab_b2, (int)N, dm_ld);
    EQH_CHECK_LAUNCH();
    if (db2) {   // (deferred into the step's batched reduction when it accumulates and a window is open)
        rc = eqh_reduce_slabs_async(slab_b2, pblocks, MDIM, db2, stream, db2_accumulate);
        if (rc) return rc;
    }
    const int remap = edge_xcd_remap();
    hipLaunchKernelGGL(k_edge_bwd_recv, dim3(xcd_grid(used * tiles, remap)), dim3(THREADS), 0, stream, ab, wd, w2, nbr, d2, dpre2,
                       dpre2_t, dab, slab_w2, slab_wd, (int)N, (int)Hp, tiles, chunk_items, used * tiles, remap);
    EQH_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_edge_bwd_send, dim3(xcd_grid(used * tiles, remap)), dim3(THREADS), 0, stream, ab, wd, w2, dpre2, t_rowptr,
                       rec, dab, (int)N, (int)Hp, tiles, chunk_items, used * tiles, remap);
    EQH_CHECK_LAUNCH();
    // (dw_accumulate: dw2 / dwd are accumulators -- the packed weights' own, ops.egnn_pack_weights -- and the reduction joins
    // the step's batched one when a deferral window is open)
    return eqh_reduce_slabs3_async(slab_w2, used, (int64_t)(MDIM + 1) * Hp, dw2, dwd, dwd, (int64_t)MDIM * Hp, Hp,
                                   dw_accumulate ? 1 : 0, stream);
}
