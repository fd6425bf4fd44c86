// Row-panel kernels for the conv-sized dense products of the path, [~5 k rows x C] . [C x C] with C = MLP_hidden <= 256
// (mlp.py:91-99 inside conv.py:169-182: 36 of them per training step of egnn_equihnns at the BASELINE batch), fused with the
// row-wise work either side of them.
//
// Why its own kernel.  At these sizes a tiled GEMM -- the library's fp32-MFMA kernels or gemm_x6.hip -- is bound by its fixed
// costs: 296 tiles of 64 x 64 are two rounds on 256 CUs, every tile re-stages (and, for x6, re-splits) its slice of the weight,
// and the bias / ReLU / LayerNorm that follows is one more launch and one more [rows, C] round trip at ~4.5 us per launch inside
// a replayed graph.  Here a workgroup owns a PANEL of 32 consecutive rows over the WHOLE width:
//   * the weight is split into its three bf16 planes ONCE per step, ahead of time, in MFMA operand order (hg_panel_pack;
//     the conv layer's weights are shared by its L applications and by forward and backward); a wavefront streams the
//     fragments of its own output columns straight from L2 into registers -- no LDS staging and no VALU split for the weight;
//   * the panel's rows are produced by a row PROLOGUE in the wave-per-row layout of the aggregation kernels (rowln.h), split
//     (bf16x3.h) and laid into LDS as the A image; they are 48 KB for all of K = 256, shared by the four wavefronts;
//   * products are the six bf16 MFMAs of gemm_x6.hip (fp32-grade results, tests compare with float64);
//   * the accumulators go through an fp32 LDS staging tile back into the wave-per-row layout, where the row EPILOGUE runs
//     (scale, addend, bias, ReLU, LayerNorm and its backward: the same device functions as the stand-alone row kernels, so
//     the fused and the unfused forms agree bit for bit given the same GEMM result), stores whole 1 KB rows, and -- in the chained
//     forms -- lays the next product's A image without leaving the workgroup.
// A panel needs the whole [C x C] weight (384 KB of planes at C = 256) through its CU's vector memory path: that, equal to the
// MFMA time of 192 MFMAs per wavefront (2.6 us at 2.4 GHz), is what bounds a panel; with ~150 panels on 256 CUs the launch is one
// round.  MFMA 32 x 32 x 16 with the operands swapped (the accumulator holds C^T: a lane owns 4 consecutive columns of one row).
#include <initializer_list>

#include "common.h"
#include "bf16x3.h"
#include "rowln.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int PN_ROWS = 32;        // rows per panel = one MFMA row tile
constexpr int PN_THREADS = 256;    // four wavefronts: wavefront w multiplies the column tiles w, w + 4
constexpr int PN_STG_LD = 260;     // floats per staged row: 256 + 4 keeps the accumulators' 16-byte stores conflict-free
constexpr int PN_PF = 3;           // K steps (of 16) of weight fragments in flight per wavefront

#ifdef PN_STAMPS   // diagnostic build only (tools/panel_stamps.py): per-wavefront s_memtime stamps of the phases
__device__ unsigned long long* pn_stamp_buf = nullptr;
#define PN_STAMP(slot)                                                                                              \
    do {                                                                                                            \
        if (pn_stamp_buf && (threadIdx.x & 63) == 0)                                                                 \
            pn_stamp_buf[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define PN_STAMP(slot) do { } while (0)
#endif

// ---- weights in MFMA operand order ------------------------------------------------------------------------------------------
// image[tile = n / 32][kstep = k / 16][plane 3][lane 64] x 16 bytes: lane (fh = lane >> 5, fr = lane & 31) holds the eight bf16
// B[16 kstep + 8 fh + 0..7][32 tile + fr] of one plane.  Several weights may be stacked along K in one image (kstep0).
struct PackItem {
    const float* w;      // trans: B[k][n] = w[n * ld + k]  (an nn.Linear weight used as x W^T);  else B[k][n] = w[k * ld + n]
    int64_t ld;
    uint4* dst;
    int K, N, trans, kstep0, ksteps_total;
    int n_valid;         // columns n >= n_valid of B are zero (N padded up to a multiple of 32)
};
constexpr int PN_MAXPACK = 32;
struct PackBatch {
    PackItem it[PN_MAXPACK];
    int first[PN_MAXPACK + 1];      // prefix sums of the items' (tile, kstep) units
    int n;
};
static_assert(sizeof(PackBatch) <= 4096, "PackBatch is a by-value kernel argument");

__global__ void __launch_bounds__(256) k_panel_pack(const PackBatch b) {
    const int unit = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6);
    if (unit >= b.first[b.n]) return;
    int i = 0;
    while (i + 1 < b.n && unit >= b.first[i + 1]) ++i;
    const PackItem it = b.it[i];
    const int local = unit - b.first[i], ks = it.K >> 4;
    const int tile = local / ks, kstep = local - tile * ks;
    const int lane = threadIdx.x & 63, fh = lane >> 5, fr = lane & 31;
    const int n = tile * 32 + fr, k0 = kstep * 16 + 8 * fh;
    float v[8];
    if (n >= it.n_valid) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
    } else if (it.trans) {
        const float4 a = *reinterpret_cast<const float4*>(it.w + (int64_t)n * it.ld + k0);
        const float4 c = *reinterpret_cast<const float4*>(it.w + (int64_t)n * it.ld + k0 + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = it.w[(int64_t)(k0 + j) * it.ld + n];
    }
    uint4 p0, p1, p2;
    split_pair(v[0], v[1], p0.x, p1.x, p2.x);
    split_pair(v[2], v[3], p0.y, p1.y, p2.y);
    split_pair(v[4], v[5], p0.z, p1.z, p2.z);
    split_pair(v[6], v[7], p0.w, p1.w, p2.w);
    uint4* d = it.dst + ((int64_t)(tile * it.ksteps_total + it.kstep0 + kstep) * 3) * 64 + lane;
    d[0] = p0;
    d[64] = p1;
    d[128] = p2;
}

// ---- the A image: a panel's rows as bf16 planes in LDS ---------------------------------------------------------------------
// [plane 3][kstep KS][slot 64] x 16 bytes; the fragment of (plane, kstep) is 1 KB, lane (fh, fr) reads slot
// fh * 32 + (fr ^ swz), swz = ((kstep & 3) << 1) | fh: the xor keeps the 8-byte row-wise writes below (a 16-lane group of a
// wave-per-row store covers four K steps x two halves of ONE row) on 32 distinct banks, and a ds_read_b128 of a fragment stays
// a permutation of its 64 slots inside each hardware lane group.
template <int KS>
__device__ __forceinline__ int a_slot(int kstep, int fh, int row) {
    return kstep * 64 + fh * 32 + (row ^ (((kstep & 3) << 1) | fh));
}

// lane l of a wave-per-row holder has v = row[4 l .. 4 l + 3] (k = 4 l + kbase): its 8 bytes of each plane
template <int KS>
__device__ __forceinline__ void a_put(uint4* __restrict__ img, int row, int k4, const float4& v) {
    const int kstep = k4 >> 2, fh = (k4 >> 1) & 1, half = k4 & 1;
    uint32_t a0, a1, a2, b0, b1, b2;
    split_pair(v.x, v.y, a0, a1, a2);
    split_pair(v.z, v.w, b0, b1, b2);
    uint2* d = reinterpret_cast<uint2*>(img) + a_slot<KS>(kstep, fh, row) * 2 + half;
    d[0] = make_uint2(a0, b0);
    d[KS * 64 * 2] = make_uint2(a1, b1);
    d[KS * 64 * 4] = make_uint2(a2, b2);
}

// ---- the product: acc[g][j] (+)= A image . W image g for the column tiles wave + 4 j ---------------------------------------
// NG products share the A image (conv.py:172,176: X feeds W1's first Linear and the node half of W2's); the weight stream
// runs on across them, PN_PF K steps ahead of the MFMAs.
template <int KS, int NTW, int NG = 1>
struct WStream {
    uint4 q[PN_PF][NTW][3];
    const uint4* base[NG][NTW];
    __device__ __forceinline__ void init(int g, const uint4* __restrict__ w, int wave, int lane, int n_tiles = 1 << 30) {
#pragma unroll
        for (int j = 0; j < NTW; ++j) {       // (a tile past the image re-reads tile 0; its product is not staged)
            const int tile = wave + 4 * j < n_tiles ? wave + 4 * j : 0;
            base[g][j] = w + (int64_t)(tile * KS) * 3 * 64 + lane;
        }
    }
    __device__ __forceinline__ void fetch(int slot, int kk) {      // kk = product * KS + kstep (compile-time after unrolling)
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) q[slot][j][p] = base[kk / KS][j][((kk % KS) * 3 + p) * 64];
    }
    __device__ __forceinline__ void prime() {
#pragma unroll
        for (int s = 0; s < PN_PF; ++s)
            if (s < KS * NG) fetch(s, s);
    }
};

template <int KS, int NTW, int NG>
__device__ __forceinline__ void panel_mma(const uint4* __restrict__ img, WStream<KS, NTW, NG>& ws, f32x16 (&acc)[NG][NTW], int lane) {
    const int fh = lane >> 5, fr = lane & 31;
    uint4 af[2][3];            // the A fragments of a K step are requested during the step before
    auto a_read = [&](int kk) {
        const uint4* ap = img + a_slot<KS>(kk % KS, fh, fr);
        af[kk & 1][0] = ap[0];
        af[kk & 1][1] = ap[KS * 64];
        af[kk & 1][2] = ap[KS * 128];
    };
    a_read(0);
#pragma unroll
    for (int kk = 0; kk < KS * NG; ++kk) {
        const int slot = kk % PN_PF, g = kk / KS;
        if (kk + 1 < KS * NG) a_read(kk + 1);
        const bf16x8 a0 = __builtin_bit_cast(bf16x8, af[kk & 1][0]);
        const bf16x8 a1 = __builtin_bit_cast(bf16x8, af[kk & 1][1]);
        const bf16x8 a2 = __builtin_bit_cast(bf16x8, af[kk & 1][2]);
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const bf16x8 b0 = __builtin_bit_cast(bf16x8, ws.q[slot][j][0]);
            const bf16x8 b1 = __builtin_bit_cast(bf16x8, ws.q[slot][j][1]);
            const bf16x8 b2 = __builtin_bit_cast(bf16x8, ws.q[slot][j][2]);
            // smallest terms first, as gemm_x6.hip: a1 b1, a0 b2, a2 b0, a0 b1, a1 b0, a0 b0
            acc[g][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a1, acc[g][j], 0, 0, 0);
            acc[g][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b2, a0, acc[g][j], 0, 0, 0);
            acc[g][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a2, acc[g][j], 0, 0, 0);
            acc[g][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a0, acc[g][j], 0, 0, 0);
            acc[g][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a1, acc[g][j], 0, 0, 0);
            acc[g][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a0, acc[g][j], 0, 0, 0);
        }
        if (kk + PN_PF < KS * NG) ws.fetch(slot, kk + PN_PF);
        // (the scheduler otherwise sinks the fetches next to their uses -- registers it thinks it saves -- and the stream
        // runs one load deep)
        __builtin_amdgcn_sched_barrier(0);
    }
}

// accumulators (C^T layout: lane (fh, fr) holds row fr, columns 8 g + 4 fh .. + 3 of its tiles) -> the fp32 staging tile
template <int NTW, int LD = PN_STG_LD>
__device__ __forceinline__ void acc_to_staging(float* __restrict__ stg, const f32x16 (&acc)[NTW], int wave, int lane, int n_tiles = 1 << 30) {
    const int fh = lane >> 5, fr = lane & 31;
#pragma unroll
    for (int j = 0; j < NTW; ++j)
        if (wave + 4 * j < n_tiles) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(stg + fr * LD + (wave + 4 * j) * 32 + 8 * g + 4 * fh) =
                    make_float4(acc[j][4 * g + 0], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]);
        }
}

template <int NG, int NTW>
__device__ __forceinline__ void acc_zero(f32x16 (&acc)[NG][NTW]) {
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[g][j][i] = 0.f;
}

// ---- plain product: C = act(alpha A W + beta D + bias) ----------------------------------------------------------------------
struct PanelPlain {
    const float* A;
    int64_t lda;
    int rows;
    const uint4* W;
    float alpha, beta;
    const float* D;
    int64_t ldd;
    const float* bias;
    int relu;
    float* Cout;
    int64_t ldc;
};

template <int C>
__global__ void __launch_bounds__(PN_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) k_panel_plain(const PanelPlain p) {
    constexpr int KS = C / 16, NT = C / 32, NTW = (NT + 3) / 4;
    __shared__ uint4 s_img[3 * KS * 64];
    __shared__ float s_stg[PN_ROWS * PN_STG_LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = (int)blockIdx.x * PN_ROWS;
    const bool mul = NT >= 4 || wave < NT;        // (C = 64: two column tiles, wavefronts 2 and 3 only move rows)
    const int c4 = lane * 4;

    PN_STAMP(0);
    float4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int row = r0 + wave * 8 + i;
        row = row < p.rows ? row : p.rows - 1;
        v[i] = c4 < C ? *reinterpret_cast<const float4*>(p.A + (int64_t)row * p.lda + c4) : f4_zero();
    }
    WStream<KS, NTW, 1> ws;
    ws.init(0, p.W, mul ? wave : 0, lane);
    ws.prime();          // unconditional: loads inside a branch make every later wait conservative (idle waves re-read tile 0)
    __builtin_amdgcn_sched_barrier(0);
    if (c4 < C) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a_put<KS>(s_img, wave * 8 + i, lane, v[i]);
    }
    PN_STAMP(1);
    __syncthreads();
    PN_STAMP(2);
    f32x16 acc[1][NTW];
    acc_zero<1, NTW>(acc);
    if (mul) {
        panel_mma<KS, NTW, 1>(s_img, ws, acc, lane);
        PN_STAMP(3);
        acc_to_staging<NTW>(s_stg, acc[0], wave, lane);
    }
    __syncthreads();
    PN_STAMP(4);
    if (c4 < C) {
        const float4 bv = p.bias ? *reinterpret_cast<const float4*>(p.bias + c4) : f4_zero();
        float4 a[8], d[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {          // all of the wavefront's rows requested before the first is used
            const int lr = wave * 8 + i, row = r0 + lr;
            a[i] = *reinterpret_cast<const float4*>(s_stg + lr * PN_STG_LD + c4);
            d[i] = p.D ? *reinterpret_cast<const float4*>(p.D + (int64_t)(row < p.rows ? row : p.rows - 1) * p.ldd + c4) : f4_zero();
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = r0 + wave * 8 + i;
            float4 o = make_float4(p.alpha * a[i].x, p.alpha * a[i].y, p.alpha * a[i].z, p.alpha * a[i].w);
            if (p.D) { o.x = fmaf(p.beta, d[i].x, o.x); o.y = fmaf(p.beta, d[i].y, o.y); o.z = fmaf(p.beta, d[i].z, o.z); o.w = fmaf(p.beta, d[i].w, o.w); }
            o.x += bv.x; o.y += bv.y; o.z += bv.z; o.w += bv.w;
            if (p.relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
            if (row < p.rows) *reinterpret_cast<float4*>(p.Cout + (int64_t)row * p.ldc + c4) = o;
        }
    }
    PN_STAMP(5);
}

// =============================================================================================================================
// The merged MHNNSConv application (conv.py:169-182 after layers.MHNNSConv._prepare_merged) on panels.
//   forward  F1: h1 = X W1a^T (stored raw, the LayerNorm backward recomputes from it), h1n = LN1(relu(h1 + b1a)), pa = X W2v^T
//            F2: hbar[e] = mean_{v in e} h1n[v]  (prologue: gathered mean over the hyperedge's nodes, conv.py:172-173)
//                qb = hbar w12^T + b12
//            (s = k_inc_fwd_col(pa, qb): incidence.hip, the HBM-bound aggregation, stays its own launch)
//            F3: u = scale * (s w23^T) + cw,  x3 = LN3(relu(u + b3a)),  Xn = act(x3 W3b^T + b3b)   [+ F1 of the next application]
//   backward B3: g = dXn * [Xn > 0],  dx3 = g W3b,  dpre = LN3bwd(u + b3a; dx3),  ds = scale * dpre w23
//            (dpa, dqb = k_inc_bwd_both(ds));  B2 = the plain product dhbar = dqb w12
//            B1: dh1[v] = LN1bwd(h1[v] + b1a; sum_{e of v} dhbar[e] / deg e)  (prologue),  dX = [dh1 | dpa] . [W1a ; W2v]
//                [+ B3 of the previous application]
// Row-wise work runs on ROW TILES: a wavefront holds its eight rows of the panel at once, eight lanes per row -- lane
// (r = lane >> 3, c = lane & 7) has the float4 at columns 32 j + 4 c, j < C / 32 -- so a LayerNorm statistic is a sum over a
// lane's own registers plus three DPP steps, the sqrt / division of a row is computed once for all eight rows, and a
// wave-instruction still moves whole 128-byte row segments.  (One wavefront per row, as the stand-alone row kernels have it,
// spends ~150 instructions per row on 64-lane reductions and per-row scalars: 5.5 us of a 16.7 us F3 launch.)  The
// arithmetic per element is that of rowln.h (two-pass mean / variance, correctly rounded sqrt and division); only the order
// of the row sums differs.  The weight / bias / LayerNorm-vector gradients are formed outside from the stored rows (batched
// weight-gradient launch, column sums) and from the per-workgroup slabs.
// =============================================================================================================================
template <int C> struct PnShape {
    static constexpr int KS = C / 16, NT = C / 32, NTW = (NT + 3) / 4, NJ = C / 32;
};

template <int C> struct RowTile { float4 v[C / 32]; };

// sum over the eight lanes of a row (every lane gets the total): quad butterflies, then the other quad through the half-row mirror
__device__ __forceinline__ float row8_sum(float v) {
    v += dpp_move<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);   // row_half_mirror: lane i <- lane 7 - i of its group of eight
    return v;
}

template <int C>
__device__ __forceinline__ void rt_load(RowTile<C>& t, const float* __restrict__ base, int64_t ld, int row, int c4) {
#pragma unroll
    for (int j = 0; j < C / 32; ++j) t.v[j] = *reinterpret_cast<const float4*>(base + (int64_t)row * ld + 32 * j + c4);
}
template <int C>
__device__ __forceinline__ void rt_store(const RowTile<C>& t, float* __restrict__ base, int64_t ld, int row, int c4) {
#pragma unroll
    for (int j = 0; j < C / 32; ++j) *reinterpret_cast<float4*>(base + (int64_t)row * ld + 32 * j + c4) = t.v[j];
}
template <int C>
__device__ __forceinline__ void rt_load_vec(RowTile<C>& t, const float* __restrict__ vec, int c4) {
#pragma unroll
    for (int j = 0; j < C / 32; ++j) t.v[j] = *reinterpret_cast<const float4*>(vec + 32 * j + c4);
}
// the tile's rows -> the A image (K offset kb4 = k / 4 of the tile's first column)
template <int C, int KS>
__device__ __forceinline__ void rt_a_put(const RowTile<C>& t, uint4* __restrict__ img, int lrow, int c8, int kb4 = 0) {
#pragma unroll
    for (int j = 0; j < C / 32; ++j) a_put<KS>(img, lrow, kb4 + 8 * j + c8, t.v[j]);
}

// y = gamma * xhat(relu(pre + bias)) + beta per row (mlp.py:93-97)
template <int C>
__device__ __forceinline__ void rt_ln_fwd(const RowTile<C>& pre, const RowTile<C>& bias, const RowTile<C>& gam, const RowTile<C>& bet,
                                          float eps, RowTile<C>& y) {
    constexpr int NJ = C / 32;
    const float inv_c = 1.0f / (float)C;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        float4 h = make_float4(pre.v[j].x + bias.v[j].x, pre.v[j].y + bias.v[j].y, pre.v[j].z + bias.v[j].z, pre.v[j].w + bias.v[j].w);
        h.x = fmaxf(h.x, 0.f); h.y = fmaxf(h.y, 0.f); h.z = fmaxf(h.z, 0.f); h.w = fmaxf(h.w, 0.f);
        y.v[j] = h;
        s += (h.x + h.y) + (h.z + h.w);
    }
    const float mu = row8_sum(s) * inv_c;
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        float4 d = y.v[j];
        d.x -= mu; d.y -= mu; d.z -= mu; d.w -= mu;
        y.v[j] = d;
        ss += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
    }
    const float r = 1.0f / sqrtf(row8_sum(ss) * inv_c + eps);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        y.v[j].x = fmaf(gam.v[j].x, y.v[j].x * r, bet.v[j].x); y.v[j].y = fmaf(gam.v[j].y, y.v[j].y * r, bet.v[j].y);
        y.v[j].z = fmaf(gam.v[j].z, y.v[j].z * r, bet.v[j].z); y.v[j].w = fmaf(gam.v[j].w, y.v[j].w * r, bet.v[j].w);
    }
}

// dpre = gradient of the pre-activation given dy = d LN output (the formulas of k_rowln_bwd); the lane's terms of d bias,
// d gamma, d beta are ADDED to a_db / a_dg / a_dbeta when `count` (rows past the end of the matrix are not counted)
template <int C>
__device__ __forceinline__ void rt_ln_bwd(const RowTile<C>& pre, const RowTile<C>& bias, const RowTile<C>& gam, const RowTile<C>& dy,
                                          float eps, bool count, RowTile<C>& dpre, RowTile<C>& a_db, RowTile<C>& a_dg, RowTile<C>& a_dbeta) {
    constexpr int NJ = C / 32;
    const float inv_c = 1.0f / (float)C;
    RowTile<C> x;
    unsigned pos = 0u;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        float4 h = make_float4(pre.v[j].x + bias.v[j].x, pre.v[j].y + bias.v[j].y, pre.v[j].z + bias.v[j].z, pre.v[j].w + bias.v[j].w);
        pos |= (((h.x > 0.f) ? 1u : 0u) | ((h.y > 0.f) ? 2u : 0u) | ((h.z > 0.f) ? 4u : 0u) | ((h.w > 0.f) ? 8u : 0u)) << (4 * j);
        h.x = fmaxf(h.x, 0.f); h.y = fmaxf(h.y, 0.f); h.z = fmaxf(h.z, 0.f); h.w = fmaxf(h.w, 0.f);
        x.v[j] = h;
        s += (h.x + h.y) + (h.z + h.w);
    }
    const float mu = row8_sum(s) * inv_c;
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        float4 d = x.v[j];
        d.x -= mu; d.y -= mu; d.z -= mu; d.w -= mu;
        x.v[j] = d;
        ss += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
    }
    const float rstd = 1.0f / sqrtf(row8_sum(ss) * inv_c + eps);
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        float4 xh = x.v[j];
        xh.x *= rstd; xh.y *= rstd; xh.z *= rstd; xh.w *= rstd;
        x.v[j] = xh;
        float4 d = dy.v[j];
        if (count) {
            f4_add(a_dbeta.v[j], d);
            a_dg.v[j].x = fmaf(d.x, xh.x, a_dg.v[j].x); a_dg.v[j].y = fmaf(d.y, xh.y, a_dg.v[j].y);
            a_dg.v[j].z = fmaf(d.z, xh.z, a_dg.v[j].z); a_dg.v[j].w = fmaf(d.w, xh.w, a_dg.v[j].w);
        }
        d.x *= gam.v[j].x; d.y *= gam.v[j].y; d.z *= gam.v[j].z; d.w *= gam.v[j].w;
        dpre.v[j] = d;
        m1 += (d.x + d.y) + (d.z + d.w);
        m2 += (d.x * xh.x + d.y * xh.y) + (d.z * xh.z + d.w * xh.w);
    }
    m1 = row8_sum(m1) * inv_c;
    m2 = row8_sum(m2) * inv_c;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const unsigned b = pos >> (4 * j);
        const float4 d = dpre.v[j], xh = x.v[j];
        float4 dx;
        dx.x = (b & 1u) ? rstd * (d.x - m1 - xh.x * m2) : 0.f;
        dx.y = (b & 2u) ? rstd * (d.y - m1 - xh.y * m2) : 0.f;
        dx.z = (b & 4u) ? rstd * (d.z - m1 - xh.z * m2) : 0.f;
        dx.w = (b & 8u) ? rstd * (d.w - m1 - xh.w * m2) : 0.f;
        dpre.v[j] = dx;
        if (count) f4_add(a_db.v[j], dx);
    }
}

template <int C>
__device__ __forceinline__ void rt_zero(RowTile<C>& t) {
#pragma unroll
    for (int j = 0; j < C / 32; ++j) t.v[j] = f4_zero();
}

// The workgroup's [d bias | d gamma | d beta]: every lane holds its row's terms; the 32 rows meet through the staging tile and
// are summed per column in row order (fixed order: bitwise reproducible) -> slab[3][C].  All threads call; the staging tile is
// free again afterwards.
template <int C>
__device__ __forceinline__ void write_slab(float* __restrict__ s_stg, float* __restrict__ slab, const RowTile<C>& a_db, const RowTile<C>& a_dg,
                                           const RowTile<C>& a_dbeta, int lrow, int c4) {
#pragma unroll
    for (int which = 0; which < 3; ++which) {
        const RowTile<C>& t = which == 0 ? a_db : (which == 1 ? a_dg : a_dbeta);
        __syncthreads();
        rt_store<C>(t, s_stg, PN_STG_LD, lrow, c4);
        __syncthreads();
        if ((int)threadIdx.x < C) {
            float acc = 0.f;
#pragma unroll 8
            for (int r = 0; r < PN_ROWS; ++r) acc += s_stg[r * PN_STG_LD + threadIdx.x];
            slab[which * C + threadIdx.x] = acc;
        }
    }
    __syncthreads();
}

// A wavefront's eight consecutive CSR rows [s_beg, s_end): sum_q w[q] * src[col[q]] per row, in the wave-per-row layout
// (lane l: columns 4 l .. 4 l + 3; rows of 1 KB are what a gather should move), rows handed to `sink(i, sum, deg)`.  One
// chain fetches the row ends and up to 64 entries of the range (a lane each); four gathered rows in flight (the walk of
// k_gather_ln_bwd, incidence.hip).
template <int C>
__device__ __forceinline__ float4 ld_row(const float* __restrict__ base, int64_t ld, int row, int lane) {
    return (lane * 4 < C) ? *reinterpret_cast<const float4*>(base + (int64_t)row * ld + lane * 4) : f4_zero();
}
template <int C, typename Sink>
__device__ __forceinline__ void gather_range(const float* __restrict__ src, const int* __restrict__ rowptr, const int* __restrict__ col,
                                             const float* __restrict__ wq, int s_beg, int s_end, int lane, Sink&& sink) {
    if (s_beg >= s_end) return;
    const int p_beg = rowptr[s_beg];
    const int my_rend = (s_beg + lane < s_end) ? rowptr[s_beg + lane + 1] : 0;   // lane i: end of row s_beg + i
    const int p_end = rowptr[s_end];
    // The range's entries are walked in order, EIGHT gathered rows in flight at a time whatever row they belong to (a
    // hyperedge has 2-3 nodes, a node 2-3 hyperedges: one round trip per CSR row -- eight of them in sequence per wavefront --
    // was what a panel's prologue waited for); a row is handed to the sink when the walk passes its end.
    int row = s_beg, rbeg = p_beg;
    int rend = __builtin_amdgcn_readlane(my_rend, 0);
    float4 sum = f4_zero();
    auto flush = [&]() {
        sink(row - s_beg, sum, rend - rbeg);
        sum = f4_zero();
        rbeg = rend;
        ++row;
        if (row < s_end) rend = __builtin_amdgcn_readlane(my_rend, row - s_beg);
    };
    for (int q0 = p_beg; q0 < p_end; q0 += 64) {
        const int cnt = (p_end - q0 < 64) ? (p_end - q0) : 64;
        const int my_c = (lane < cnt) ? col[q0 + lane] : 0;
        const float my_w = (wq && lane < cnt) ? wq[q0 + lane] : 1.0f;
        for (int j0 = 0; j0 < cnt; j0 += 8) {
            float4 d[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {       // entries past the chunk re-read its last one and are not added
                const int jj = (j0 + t < cnt) ? j0 + t : cnt - 1;
                d[t] = ld_row<C>(src, C, __builtin_amdgcn_readlane(my_c, jj), lane);
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (j0 + t < cnt) {
                    const int q = q0 + j0 + t;
                    while (q >= rend) flush();          // (rows without entries are handed out empty)
                    f4_fma(sum, d[t], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_w), j0 + t)));
                }
            }
        }
    }
    while (row < s_end) flush();
}

struct ConvPanelArgs {
    int rows;                       // rows of this stage's panels (nodes, or hyperedges for F2)
    float eps, scale;
    int relu, acc_first, tail;
    // operands / products (meaning per stage, see the kernels)
    const float* in0; const float* in1; const float* in2; const float* in3;
    int64_t ld0;
    const int* rowptr; const int* col; const float* wq;
    const uint4* w0; const uint4* w1; const uint4* w2; const uint4* w3;
    const float* b0; const float* g0; const float* be0;      // bias / gamma / beta of the first LayerNorm of the stage
    const float* b1; const float* g1; const float* be1;      // ... of the tail's
    const float* bias_out;                                    // bias of a plain Linear output
    float* out0; float* out1; float* out2; float* out3; float* out4; float* out5;
    float* slab; float* slab2;
    float* acc_out;
};

// lane -> (its row of the wavefront's eight, its column quad); rows past the end of the matrix are clamped for loads
struct RtPos {
    int lrow, c8, c4, row, rowc;
    bool live;
    __device__ __forceinline__ RtPos(int r0, int rows, int wave, int lane) {
        lrow = wave * 8 + (lane >> 3);
        c8 = lane & 7;
        c4 = c8 * 4;
        row = r0 + lrow;
        live = row < rows;
        rowc = live ? row : rows - 1;
    }
};

// ---- F1: X -> h1 (raw), h1n = LN1(relu(h1 + b1a)), pa -------------------------------------------------------------------------
// the A image holds X's panel; w_a = W1a image (x W^T), w_b = W2v image
template <int C>
__device__ __forceinline__ void stage_f1(const ConvPanelArgs& p, uint4* __restrict__ s_img, float* __restrict__ s_stg,
                                         float* __restrict__ s_stg2, const uint4* w_a, const uint4* w_b, const float* b1a,
                                         const float* g1, const float* be1, float* h1, float* h1n, float* pa, const RtPos& P, int wave,
                                         int lane, bool mul) {
    using S = PnShape<C>;
    WStream<S::KS, S::NTW, 2> ws;
    ws.init(0, w_a, mul ? wave : 0, lane);
    ws.init(1, w_b, mul ? wave : 0, lane);
    ws.prime();
    __builtin_amdgcn_sched_barrier(0);
    f32x16 acc[2][S::NTW];
    acc_zero<2, S::NTW>(acc);
    if (mul) {
        panel_mma<S::KS, S::NTW, 2>(s_img, ws, acc, lane);
        acc_to_staging<S::NTW>(s_stg, acc[0], wave, lane);
        acc_to_staging<S::NTW>(s_stg2, acc[1], wave, lane);
    }
    __syncthreads();
    RowTile<C> a, b, bv, gv, bev, y;
    rt_load<C>(a, s_stg, PN_STG_LD, P.lrow, P.c4);
    rt_load<C>(b, s_stg2, PN_STG_LD, P.lrow, P.c4);
    rt_load_vec<C>(bv, b1a, P.c4);
    rt_load_vec<C>(gv, g1, P.c4);
    rt_load_vec<C>(bev, be1, P.c4);
    if (P.live) {
        rt_store<C>(a, h1, C, P.row, P.c4);
        rt_store<C>(b, pa, C, P.row, P.c4);
    }
    rt_ln_fwd<C>(a, bv, gv, bev, p.eps, y);
    if (P.live) rt_store<C>(y, h1n, C, P.row, P.c4);
}

template <int C>
__global__ void __launch_bounds__(PN_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) k_conv_f1(const ConvPanelArgs p) {
    using S = PnShape<C>;
    __shared__ uint4 s_img[3 * S::KS * 64];
    __shared__ float s_stg[PN_ROWS * PN_STG_LD];
    __shared__ float s_stg2[PN_ROWS * PN_STG_LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const RtPos P((int)blockIdx.x * PN_ROWS, p.rows, wave, lane);
    const bool mul = S::NT >= 4 || wave < S::NT;
    RowTile<C> x;
    rt_load<C>(x, p.in0, p.ld0, P.rowc, P.c4);
    rt_a_put<C, S::KS>(x, s_img, P.lrow, P.c8);
    __syncthreads();
    stage_f1<C>(p, s_img, s_stg, s_stg2, p.w0, p.w1, p.b0, p.g0, p.be0, p.out0, p.out1, p.out2, P, wave, lane, mul);
}

// ---- F2: hbar[e] = mean over the hyperedge's nodes of h1n, qb = hbar w12^T + b12 ---------------------------------------------
// in0 = h1n [N, C], rowptr / col = the incidence CSR by hyperedge, w0 = w12 image, bias_out = b12; out0 = hbar, out1 = qb
template <int C>
__global__ void __launch_bounds__(PN_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) k_conv_f2(const ConvPanelArgs p) {
    using S = PnShape<C>;
    __shared__ uint4 s_img[3 * S::KS * 64];
    __shared__ float s_stg[PN_ROWS * PN_STG_LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = (int)blockIdx.x * PN_ROWS;
    const RtPos P(r0, p.rows, wave, lane);
    const bool mul = S::NT >= 4 || wave < S::NT;
    WStream<S::KS, S::NTW, 1> ws;
    ws.init(0, p.w0, mul ? wave : 0, lane);
    ws.prime();
    __builtin_amdgcn_sched_barrier(0);
    const int s_beg = min(r0 + wave * 8, p.rows), s_end = min(s_beg + 8, p.rows);
    gather_range<C>(p.in0, p.rowptr, p.col, nullptr, s_beg, s_end, lane, [&](int i, const float4& sum, int deg) {
        const float den = deg > 1 ? (float)deg : 1.0f;
        const float4 m = make_float4(sum.x / den, sum.y / den, sum.z / den, sum.w / den);
        if (lane * 4 < C) {
            *reinterpret_cast<float4*>(p.out0 + (int64_t)(s_beg + i) * C + lane * 4) = m;
            a_put<S::KS>(s_img, wave * 8 + i, lane, m);
        }
    });
    for (int i = s_end - s_beg; i < 8; ++i)                 // rows past the end of the matrix: zeros for the MFMA
        if (lane * 4 < C) a_put<S::KS>(s_img, wave * 8 + i, lane, f4_zero());
    __syncthreads();
    f32x16 acc[1][S::NTW];
    acc_zero<1, S::NTW>(acc);
    if (mul) {
        panel_mma<S::KS, S::NTW, 1>(s_img, ws, acc, lane);
        acc_to_staging<S::NTW>(s_stg, acc[0], wave, lane);
    }
    __syncthreads();
    RowTile<C> a, bv;
    rt_load<C>(a, s_stg, PN_STG_LD, P.lrow, P.c4);
    rt_load_vec<C>(bv, p.bias_out, P.c4);
#pragma unroll
    for (int j = 0; j < S::NJ; ++j) f4_add(a.v[j], bv.v[j]);
    if (P.live) rt_store<C>(a, p.out1, C, P.row, P.c4);
}

// ---- F3: s -> u = scale * (s w23^T) + cw, x3 = LN3(relu(u + b3a)), Xn = act(x3 W3b^T + b3b)  [tail: F1 on Xn] ----------------
// in0 = s, in1 = cw, w0 = w23 image, b0/g0/be0 = b3a, gamma3, beta3, w1 = W3b image, bias_out = b3b, relu;
// out0 = u, out1 = x3, out2 = Xn;  tail: w2 = W1a image, w3 = W2v image, b1/g1/be1 = b1a, gamma1, beta1, out3 = h1, out4 = h1n, out5 = pa
template <int C>
__global__ void __launch_bounds__(PN_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) k_conv_f3(const ConvPanelArgs p) {
    using S = PnShape<C>;
    __shared__ uint4 s_img[3 * S::KS * 64];
    __shared__ float s_stg[PN_ROWS * PN_STG_LD];
    __shared__ float s_stg2[PN_ROWS * PN_STG_LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const RtPos P((int)blockIdx.x * PN_ROWS, p.rows, wave, lane);
    const bool mul = S::NT >= 4 || wave < S::NT;
    PN_STAMP(0);
    RowTile<C> t, cw;
    rt_load<C>(t, p.in0, C, P.rowc, P.c4);
    WStream<S::KS, S::NTW, 1> ws;
    ws.init(0, p.w0, mul ? wave : 0, lane);
    ws.prime();
    rt_load<C>(cw, p.in1, C, P.rowc, P.c4);
    __builtin_amdgcn_sched_barrier(0);
    rt_a_put<C, S::KS>(t, s_img, P.lrow, P.c8);
    PN_STAMP(1);
    __syncthreads();
    PN_STAMP(2);
    f32x16 acc[1][S::NTW];
    acc_zero<1, S::NTW>(acc);
    if (mul) {
        panel_mma<S::KS, S::NTW, 1>(s_img, ws, acc, lane);
        PN_STAMP(3);
        acc_to_staging<S::NTW>(s_stg, acc[0], wave, lane);
    }
    // the next product's weight stream starts now: its first fragments arrive while the rows are normalised
    ws.init(0, p.w1, mul ? wave : 0, lane);
    ws.prime();
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    PN_STAMP(4);
    {
        RowTile<C> bv, gv, bev, x3;
        rt_load<C>(t, s_stg, PN_STG_LD, P.lrow, P.c4);
        rt_load_vec<C>(bv, p.b0, P.c4);
        rt_load_vec<C>(gv, p.g0, P.c4);
        rt_load_vec<C>(bev, p.be0, P.c4);
#pragma unroll
        for (int j = 0; j < S::NJ; ++j)
            t.v[j] = make_float4(fmaf(p.scale, t.v[j].x, cw.v[j].x), fmaf(p.scale, t.v[j].y, cw.v[j].y), fmaf(p.scale, t.v[j].z, cw.v[j].z),
                                 fmaf(p.scale, t.v[j].w, cw.v[j].w));
        rt_ln_fwd<C>(t, bv, gv, bev, p.eps, x3);
        if (P.live) {
            rt_store<C>(t, p.out0, C, P.row, P.c4);
            rt_store<C>(x3, p.out1, C, P.row, P.c4);
        }
        rt_a_put<C, S::KS>(x3, s_img, P.lrow, P.c8);       // (every wavefront has left the MFMA loop: barrier above)
    }
    PN_STAMP(5);
    __syncthreads();
    PN_STAMP(6);
    acc_zero<1, S::NTW>(acc);
    if (mul) {
        panel_mma<S::KS, S::NTW, 1>(s_img, ws, acc, lane);
        acc_to_staging<S::NTW>(s_stg, acc[0], wave, lane);
    }
    PN_STAMP(7);
    __syncthreads();
    {
        RowTile<C> bv;
        rt_load<C>(t, s_stg, PN_STG_LD, P.lrow, P.c4);
        rt_load_vec<C>(bv, p.bias_out, P.c4);
#pragma unroll
        for (int j = 0; j < S::NJ; ++j) {
            f4_add(t.v[j], bv.v[j]);
            if (p.relu) { t.v[j].x = fmaxf(t.v[j].x, 0.f); t.v[j].y = fmaxf(t.v[j].y, 0.f); t.v[j].z = fmaxf(t.v[j].z, 0.f); t.v[j].w = fmaxf(t.v[j].w, 0.f); }
        }
        if (P.live) rt_store<C>(t, p.out2, C, P.row, P.c4);
        if (p.tail) rt_a_put<C, S::KS>(t, s_img, P.lrow, P.c8);
    }
    if (!p.tail) return;
    __syncthreads();
    stage_f1<C>(p, s_img, s_stg, s_stg2, p.w2, p.w3, p.b1, p.g1, p.be1, p.out3, p.out4, p.out5, P, wave, lane, mul);
}

// ---- B3: dXn -> g = dXn * [Xn > 0], dx3 = g W3b, dpre = LN3bwd(u + b3a; dx3), ds = scale * dpre w23 --------------------------
// Shared by k_conv_b3 (rows from memory) and the tail of k_conv_b1 (rows = the dX it has just formed, in the staging tile).
// w_a = W3b image (dy W), w_b = w23 image (dy W); slab = [d b3a | d gamma3 | d beta3] of this workgroup; acc_out += dpre
template <int C, bool FROM_STAGING>
__device__ __forceinline__ void stage_b3(const ConvPanelArgs& p, uint4* __restrict__ s_img, float* __restrict__ s_stg,
                                         const float* dxn, int64_t ld_dxn, const float* xmask, const uint4* w_a, const uint4* w_b,
                                         const float* u_pre, const float* b3a, const float* g3, float* g_out, float* dpre_out,
                                         float* ds_out, float* slab, float* acc_out, int acc_first, const RtPos& P, int wave, int lane,
                                         bool mul) {
    using S = PnShape<C>;
    RowTile<C> t, upre;
    if constexpr (FROM_STAGING) rt_load<C>(t, s_stg, PN_STG_LD, P.lrow, P.c4);
    else rt_load<C>(t, dxn, ld_dxn, P.rowc, P.c4);
    if (xmask) {
        RowTile<C> m;
        rt_load<C>(m, xmask, C, P.rowc, P.c4);
#pragma unroll
        for (int j = 0; j < S::NJ; ++j) {
            t.v[j].x = m.v[j].x > 0.f ? t.v[j].x : 0.f; t.v[j].y = m.v[j].y > 0.f ? t.v[j].y : 0.f;
            t.v[j].z = m.v[j].z > 0.f ? t.v[j].z : 0.f; t.v[j].w = m.v[j].w > 0.f ? t.v[j].w : 0.f;
        }
    }
    WStream<S::KS, S::NTW, 1> ws;
    ws.init(0, w_a, mul ? wave : 0, lane);
    ws.prime();
    rt_load<C>(upre, u_pre, C, P.rowc, P.c4);
    __builtin_amdgcn_sched_barrier(0);
    if (g_out && P.live) rt_store<C>(t, g_out, C, P.row, P.c4);
    // (FROM_STAGING: a wavefront reads its own rows of the staging tile above and the tile is next written after the barrier
    // below; the A image was last read before the barriers of the caller's slab reduction)
    rt_a_put<C, S::KS>(t, s_img, P.lrow, P.c8);
    __syncthreads();
    f32x16 acc[1][S::NTW];
    acc_zero<1, S::NTW>(acc);
    if (mul) {
        panel_mma<S::KS, S::NTW, 1>(s_img, ws, acc, lane);
        acc_to_staging<S::NTW>(s_stg, acc[0], wave, lane);
    }
    ws.init(0, w_b, mul ? wave : 0, lane);
    ws.prime();
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    RowTile<C> a_db, a_dg, a_dbeta;
    rt_zero<C>(a_db); rt_zero<C>(a_dg); rt_zero<C>(a_dbeta);
    {
        RowTile<C> bv, gv, dpre;
        rt_load<C>(t, s_stg, PN_STG_LD, P.lrow, P.c4);
        rt_load_vec<C>(bv, b3a, P.c4);
        rt_load_vec<C>(gv, g3, P.c4);
        rt_ln_bwd<C>(upre, bv, gv, t, p.eps, P.live, dpre, a_db, a_dg, a_dbeta);
        if (P.live) {
            rt_store<C>(dpre, dpre_out, C, P.row, P.c4);
            if (acc_out) {
                if (!acc_first) {
                    RowTile<C> o;
                    rt_load<C>(o, acc_out, C, P.row, P.c4);
#pragma unroll
                    for (int j = 0; j < S::NJ; ++j) f4_add(o.v[j], dpre.v[j]);
                    rt_store<C>(o, acc_out, C, P.row, P.c4);
                } else {
                    rt_store<C>(dpre, acc_out, C, P.row, P.c4);
                }
            }
        }
        rt_a_put<C, S::KS>(dpre, s_img, P.lrow, P.c8);
    }
    __syncthreads();
    acc_zero<1, S::NTW>(acc);
    if (mul) {
        panel_mma<S::KS, S::NTW, 1>(s_img, ws, acc, lane);
        acc_to_staging<S::NTW>(s_stg, acc[0], wave, lane);
    }
    __syncthreads();
    rt_load<C>(t, s_stg, PN_STG_LD, P.lrow, P.c4);
#pragma unroll
    for (int j = 0; j < S::NJ; ++j) { t.v[j].x *= p.scale; t.v[j].y *= p.scale; t.v[j].z *= p.scale; t.v[j].w *= p.scale; }
    if (P.live) rt_store<C>(t, ds_out, C, P.row, P.c4);
    write_slab<C>(s_stg, slab + (int64_t)blockIdx.x * 3 * C, a_db, a_dg, a_dbeta, P.lrow, P.c4);
}

// in0 = dXn (ld0), in1 = Xn or null, w0 = W3b image, w1 = w23 image, in2 = u, b0/g0 = b3a, gamma3;
// out0 = g (when in1), out1 = dpre, out2 = ds, slab, acc_out (+= dpre; overwritten when acc_first)
template <int C>
__global__ void __launch_bounds__(PN_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) k_conv_b3(const ConvPanelArgs p) {
    using S = PnShape<C>;
    __shared__ uint4 s_img[3 * S::KS * 64];
    __shared__ float s_stg[PN_ROWS * PN_STG_LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const RtPos P((int)blockIdx.x * PN_ROWS, p.rows, wave, lane);
    const bool mul = S::NT >= 4 || wave < S::NT;
    stage_b3<C, false>(p, s_img, s_stg, p.in0, p.ld0, p.in1, p.w0, p.w1, p.in2, p.b0, p.g0, p.out0, p.out1, p.out2, p.slab,
                       p.acc_out, p.acc_first, P, wave, lane, mul);
}

// ---- B1: dh1[v] = LN1bwd(h1[v] + b1a; sum_e dhbar[e] / deg e), dX = [dh1 | dpa] . [W1a ; W2v]  [tail: B3 of the application before]
// in0 = dhbar [M, C] (or, with w3 = w12 image (dy W), dqb: B2 folded in), rowptr / col / wq = incidence CSR by node + entry
// weights, in1 = h1, b0/g0 = b1a, gamma1, in2 = dpa, w0 = stacked image [W1a ; W2v] (dy W, K = 2 C); out0 = dh1, out1 = dX,
// slab = [d b1a | d gamma1 | d beta1];
// tail: in3 = X of this application = Xn of the one before (mask), w1 = W3b image, w2 = w23 image, out5 = its u (read),
//       b1/g1 = b3a, gamma3; out2 = g, out3 = dpre, out4 = ds, slab2, acc_out
template <int C>
__global__ void __launch_bounds__(PN_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) k_conv_b1(const ConvPanelArgs p) {
    using S = PnShape<C>;
    constexpr int KS2 = 2 * S::KS;
    __shared__ uint4 s_img[3 * KS2 * 64];
    __shared__ float s_stg[PN_ROWS * PN_STG_LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = (int)blockIdx.x * PN_ROWS;
    const RtPos P(r0, p.rows, wave, lane);
    const bool mul = S::NT >= 4 || wave < S::NT;
    // w3 != null: B2 folded in -- in0 is dqb and the gathered sums are multiplied by w12 here (the gathered mean is linear:
    // sum_e w_e (dqb[e] w12) = (sum_e w_e dqb[e]) w12), so dhbar never exists and its launch is gone
    PN_STAMP(0);
    WStream<S::KS, S::NTW, 1> ws12;
    if (p.w3) {
        ws12.init(0, p.w3, mul ? wave : 0, lane);
        ws12.prime();
    }
    RowTile<C> h, dpa;
    rt_load<C>(h, p.in1, C, P.rowc, P.c4);
    rt_load<C>(dpa, p.in2, C, P.rowc, P.c4);
    __builtin_amdgcn_sched_barrier(0);
    // the gathered sums of the wavefront's eight rows go through ITS rows of the staging tile into the row-tile layout
    const int s_beg = min(r0 + wave * 8, p.rows), s_end = min(s_beg + 8, p.rows);
    gather_range<C>(p.in0, p.rowptr, p.col, p.wq, s_beg, s_end, lane, [&](int i, const float4& dsum, int) {
        if (lane * 4 < C) *reinterpret_cast<float4*>(s_stg + (wave * 8 + i) * PN_STG_LD + lane * 4) = dsum;
    });
    for (int i = s_end - s_beg; i < 8; ++i)
        if (lane * 4 < C) *reinterpret_cast<float4*>(s_stg + (wave * 8 + i) * PN_STG_LD + lane * 4) = f4_zero();
    PN_STAMP(1);
    if (p.w3) {
        RowTile<C> dq;
        rt_load<C>(dq, s_stg, PN_STG_LD, P.lrow, P.c4);
        rt_a_put<C, S::KS>(dq, s_img, P.lrow, P.c8);
        __syncthreads();
        f32x16 acc12[1][S::NTW];
        acc_zero<1, S::NTW>(acc12);
        if (mul) {
            panel_mma<S::KS, S::NTW, 1>(s_img, ws12, acc12, lane);
            acc_to_staging<S::NTW>(s_stg, acc12[0], wave, lane);
        }
        __syncthreads();
    }
    PN_STAMP(2);
    WStream<KS2, S::NTW, 1> ws;
    ws.init(0, p.w0, mul ? wave : 0, lane);
    ws.prime();
    __builtin_amdgcn_sched_barrier(0);
    RowTile<C> a_db, a_dg, a_dbeta;
    rt_zero<C>(a_db); rt_zero<C>(a_dg); rt_zero<C>(a_dbeta);
    {
        RowTile<C> dsum, bv, gv, dh;
        rt_load<C>(dsum, s_stg, PN_STG_LD, P.lrow, P.c4);      // (the wavefront's own writes: program order, no barrier)
        rt_load_vec<C>(bv, p.b0, P.c4);
        rt_load_vec<C>(gv, p.g0, P.c4);
        rt_ln_bwd<C>(h, bv, gv, dsum, p.eps, P.live, dh, a_db, a_dg, a_dbeta);
        if (P.live) rt_store<C>(dh, p.out0, C, P.row, P.c4);
        rt_a_put<C, KS2>(dh, s_img, P.lrow, P.c8, 0);
        rt_a_put<C, KS2>(dpa, s_img, P.lrow, P.c8, C / 4);
    }
    PN_STAMP(3);
    __syncthreads();
    f32x16 acc[1][S::NTW];
    acc_zero<1, S::NTW>(acc);
    if (mul) {
        panel_mma<KS2, S::NTW, 1>(s_img, ws, acc, lane);
        acc_to_staging<S::NTW>(s_stg, acc[0], wave, lane);
    }
    PN_STAMP(4);
    __syncthreads();
    RowTile<C> dx;
    rt_load<C>(dx, s_stg, PN_STG_LD, P.lrow, P.c4);
    if (p.out1 && P.live) rt_store<C>(dx, p.out1, C, P.row, P.c4);     // (with the tail only the masked gradient g is needed afterwards)
    write_slab<C>(s_stg, p.slab + (int64_t)blockIdx.x * 3 * C, a_db, a_dg, a_dbeta, P.lrow, P.c4);
    PN_STAMP(5);
    if (!p.tail) return;
    // the tail starts from dX, which the slab reduction has overwritten in the staging tile: back from registers
    rt_store<C>(dx, s_stg, PN_STG_LD, P.lrow, P.c4);
    stage_b3<C, true>(p, s_img, s_stg, nullptr, 0, p.in3, p.w1, p.w2, p.out5, p.b1, p.g1, p.out2, p.out3, p.out4, p.slab2,
                      p.acc_out, p.acc_first, P, wave, lane, mul);
    PN_STAMP(6);
}

// =============================================================================================================================
// The EGNN node update (egnn_layer.py:360-362 with node_mlp = Linear(C + 16, 2 C) -> SiLU -> Linear(2 C, C), :180-187):
//   forward   node_in = [normed | m_i],  hpre = node_in W0^T + b0,  hid = silu(hpre),  out = hid W3^T + b3 + feats
//   backward  dhid = dout W3,  dpre = dhid * silu'(hpre),  dnode_in = dpre W0   (= [d normed | d m_i])
// as two panel launches instead of cat + GEMM + SiLU + GEMM + add (and their five backward launches).  The 2 C-wide hidden row
// goes through the staging tile in two halves of C columns; the C + 16 columns of dnode_in are nine column tiles (the image is
// zero-padded to C + 32).
// =============================================================================================================================
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }
__device__ __forceinline__ float silu_grad_f(float x) {
    const float sg = 1.0f / (1.0f + expf(-x));
    return sg * (1.0f + x * (1.0f - sg));
}

// in0 = normed [N, C], in1 = m_i [N, 16], in2 = feats (residual); w0 / w1 = W0 T image, output columns [0, C) / [C, 2 C) (K = C + 16);
// w2 = W3 T image (K = 2 C); b0 = bias of W0 [2 C], bias_out = bias of W3 [C];
// out0 = node_in [N, C + 16], out1 = hpre [N, 2 C], out2 = hid [N, 2 C], out3 = out [N, C]
template <int C>
__global__ void __launch_bounds__(PN_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) k_node_f(const ConvPanelArgs p) {
    using S = PnShape<C>;
    constexpr int KS1 = C / 16 + 1, KS2 = C / 8;
    __shared__ uint4 s_img[3 * KS2 * 64];
    __shared__ float s_stg[PN_ROWS * PN_STG_LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const RtPos P((int)blockIdx.x * PN_ROWS, p.rows, wave, lane);
    const bool mul = S::NT >= 4 || wave < S::NT;
    RowTile<C> x, res;
    rt_load<C>(x, p.in0, C, P.rowc, P.c4);
    const float4 mi = P.c8 < 4 ? *reinterpret_cast<const float4*>(p.in1 + (int64_t)P.rowc * 16 + P.c4) : f4_zero();
    WStream<KS1, S::NTW, 2> ws;
    ws.init(0, p.w0, mul ? wave : 0, lane);
    ws.init(1, p.w1, mul ? wave : 0, lane);
    ws.prime();
    rt_load<C>(res, p.in2, C, P.rowc, P.c4);
    __builtin_amdgcn_sched_barrier(0);
    rt_a_put<C, KS1>(x, s_img, P.lrow, P.c8);
    if (P.c8 < 4) a_put<KS1>(s_img, P.lrow, C / 4 + P.c8, mi);
    if (P.live) {
        rt_store<C>(x, p.out0, C + 16, P.row, P.c4);
        if (P.c8 < 4) *reinterpret_cast<float4*>(p.out0 + (int64_t)P.row * (C + 16) + C + P.c4) = mi;
    }
    __syncthreads();
    f32x16 acc[2][S::NTW];
    acc_zero<2, S::NTW>(acc);
    if (mul) {
        panel_mma<KS1, S::NTW, 2>(s_img, ws, acc, lane);
        acc_to_staging<S::NTW>(s_stg, acc[0], wave, lane);
    }
    WStream<KS2, S::NTW, 1> ws2;
    ws2.init(0, p.w2, mul ? wave : 0, lane);
    ws2.prime();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __syncthreads();                       // (half 0: every wavefront is out of the MFMA loop -- the image may be rewritten)
        RowTile<C> t, bv;
        rt_load<C>(t, s_stg, PN_STG_LD, P.lrow, P.c4);
        rt_load_vec<C>(bv, p.b0 + half * C, P.c4);
#pragma unroll
        for (int j = 0; j < S::NJ; ++j) f4_add(t.v[j], bv.v[j]);
        if (P.live) rt_store<C>(t, p.out1 + half * C, 2 * C, P.row, P.c4);
#pragma unroll
        for (int j = 0; j < S::NJ; ++j)
            t.v[j] = make_float4(silu_f(t.v[j].x), silu_f(t.v[j].y), silu_f(t.v[j].z), silu_f(t.v[j].w));
        if (P.live) rt_store<C>(t, p.out2 + half * C, 2 * C, P.row, P.c4);
        rt_a_put<C, KS2>(t, s_img, P.lrow, P.c8, half * (C / 4));
        if (half == 0) {
            __syncthreads();                   // the staging tile's rows have been read: second half of the product
            if (mul) acc_to_staging<S::NTW>(s_stg, acc[1], wave, lane);
        }
    }
    __syncthreads();
    f32x16 acc2[1][S::NTW];
    acc_zero<1, S::NTW>(acc2);
    if (mul) {
        panel_mma<KS2, S::NTW, 1>(s_img, ws2, acc2, lane);
        acc_to_staging<S::NTW>(s_stg, acc2[0], wave, lane);
    }
    __syncthreads();
    RowTile<C> t, bv;
    rt_load<C>(t, s_stg, PN_STG_LD, P.lrow, P.c4);
    rt_load_vec<C>(bv, p.bias_out, P.c4);
#pragma unroll
    for (int j = 0; j < S::NJ; ++j) { f4_add(t.v[j], bv.v[j]); f4_add(t.v[j], res.v[j]); }
    if (P.live) rt_store<C>(t, p.out3, C, P.row, P.c4);
}

// in0 = dout [N, C] (ld0), in1 = hpre [N, 2 C]; w0 / w1 = W3 N image, output columns [0, C) / [C, 2 C) (K = C);
// w2 = W0 N image (K = 2 C, N = C + 16 zero-padded to C + 32); out0 = dpre [N, 2 C], out1 = dnode_in [N, C + 16]
template <int C>
__global__ void __launch_bounds__(PN_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) k_node_b(const ConvPanelArgs p) {
    using S = PnShape<C>;
    constexpr int KS2 = C / 8, NT2 = C / 32 + 1, NTW2 = (NT2 + 3) / 4, LD2 = C + 32 + 4;
    __shared__ uint4 s_img[3 * KS2 * 64];
    __shared__ float s_stg[PN_ROWS * LD2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const RtPos P((int)blockIdx.x * PN_ROWS, p.rows, wave, lane);
    const bool mul = S::NT >= 4 || wave < S::NT;
    RowTile<C> d, hp[2];
    rt_load<C>(d, p.in0, p.ld0, P.rowc, P.c4);
    WStream<S::KS, S::NTW, 2> ws;
    ws.init(0, p.w0, mul ? wave : 0, lane);
    ws.init(1, p.w1, mul ? wave : 0, lane);
    ws.prime();
    rt_load<C>(hp[0], p.in1, 2 * C, P.rowc, P.c4);
    rt_load<C>(hp[1], p.in1 + C, 2 * C, P.rowc, P.c4);
    __builtin_amdgcn_sched_barrier(0);
    rt_a_put<C, S::KS>(d, s_img, P.lrow, P.c8);
    __syncthreads();
    f32x16 acc[2][S::NTW];
    acc_zero<2, S::NTW>(acc);
    if (mul) {
        panel_mma<S::KS, S::NTW, 2>(s_img, ws, acc, lane);
        acc_to_staging<S::NTW, LD2>(s_stg, acc[0], wave, lane);
    }
    const bool mul2 = NT2 >= 4 || wave < NT2;
    WStream<KS2, NTW2, 1> ws2;
    ws2.init(0, p.w2, mul2 ? wave : 0, lane, NT2);
    ws2.prime();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __syncthreads();
        RowTile<C> t;
        rt_load<C>(t, s_stg, LD2, P.lrow, P.c4);
#pragma unroll
        for (int j = 0; j < S::NJ; ++j) {
            const float4 h = hp[half].v[j];
            t.v[j].x *= silu_grad_f(h.x); t.v[j].y *= silu_grad_f(h.y); t.v[j].z *= silu_grad_f(h.z); t.v[j].w *= silu_grad_f(h.w);
        }
        if (P.live) rt_store<C>(t, p.out0 + half * C, 2 * C, P.row, P.c4);
        rt_a_put<C, KS2>(t, s_img, P.lrow, P.c8, half * (C / 4));
        if (half == 0) {
            __syncthreads();
            if (mul) acc_to_staging<S::NTW, LD2>(s_stg, acc[1], wave, lane);
        }
    }
    __syncthreads();
    f32x16 acc2[1][NTW2];
    acc_zero<1, NTW2>(acc2);
    if (mul2) {
        panel_mma<KS2, NTW2, 1>(s_img, ws2, acc2, lane);
        acc_to_staging<NTW2, LD2>(s_stg, acc2[0], wave, lane, NT2);
    }
    __syncthreads();
    RowTile<C> t;
    rt_load<C>(t, s_stg, LD2, P.lrow, P.c4);
    if (P.live) {
        rt_store<C>(t, p.out1, C + 16, P.row, P.c4);
        if (P.c8 < 4)
            *reinterpret_cast<float4*>(p.out1 + (int64_t)P.row * (C + 16) + C + P.c4) =
                *reinterpret_cast<const float4*>(s_stg + P.lrow * LD2 + C + P.c4);
    }
}

inline bool pn_width_ok(int C) { return C == 64 || C == 128 || C == 256; }

}  // namespace

extern "C" size_t hg_conv_panel_slab_bytes(int64_t rows, int32_t C) {
    if (rows < 0 || C <= 0) return 0;
    return (size_t)((rows + PN_ROWS - 1) / PN_ROWS) * 3 * (size_t)C * sizeof(float);
}

extern "C" int hg_conv_panel(int32_t stage, const HgConvPanel* q, void* stream_) {
    if (!q) return EQH_ERR_ARG;
    const int C = q->C;
    if (!pn_width_ok(C) || q->rows < 0) return EQH_ERR_ARG;
    if (q->rows >= ((int64_t)1 << 31) - 64) return EQH_ERR_RANGE;
    if (q->rows == 0) return EQH_OK;
    const void* ptrs[] = {q->in0, q->in1, q->in2, q->in3, q->w0, q->w1, q->w2, q->w3, q->b0, q->g0, q->be0, q->b1, q->g1, q->be1,
                          q->bias_out, q->out0, q->out1, q->out2, q->out3, q->out4, q->out5, q->slab, q->slab2, q->acc_out, q->wq};
    for (const void* x : ptrs)
        if (!eqh_aligned16(x)) return EQH_ERR_ALIGN;
    ConvPanelArgs a{};
    a.rows = (int)q->rows; a.eps = q->eps; a.scale = q->scale; a.relu = q->relu; a.acc_first = q->acc_first; a.tail = q->tail;
    a.in0 = q->in0; a.in1 = q->in1; a.in2 = q->in2; a.in3 = q->in3; a.ld0 = q->ld0 > 0 ? q->ld0 : C;
    a.rowptr = q->rowptr; a.col = q->col; a.wq = q->wq;
    a.w0 = static_cast<const uint4*>(q->w0); a.w1 = static_cast<const uint4*>(q->w1);
    a.w2 = static_cast<const uint4*>(q->w2); a.w3 = static_cast<const uint4*>(q->w3);
    a.b0 = q->b0; a.g0 = q->g0; a.be0 = q->be0; a.b1 = q->b1; a.g1 = q->g1; a.be1 = q->be1; a.bias_out = q->bias_out;
    a.out0 = q->out0; a.out1 = q->out1; a.out2 = q->out2; a.out3 = q->out3; a.out4 = q->out4; a.out5 = q->out5;
    a.slab = q->slab; a.slab2 = q->slab2; a.acc_out = q->acc_out;
    if (a.ld0 & 3) return EQH_ERR_ALIGN;
    const int blocks = (int)((q->rows + PN_ROWS - 1) / PN_ROWS);
    const dim3 grid(blocks), block(PN_THREADS);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    auto need = [](std::initializer_list<const void*> l) { for (const void* x : l) if (!x) return false; return true; };
    auto f1_ok = [&](const void* wa, const void* wb, const void* b, const void* g, const void* be, const void* o0, const void* o1, const void* o2) {
        return need({wa, wb, b, g, be, o0, o1, o2});
    };
#define PN_LAUNCH(K)                                                                                   \
    do {                                                                                               \
        if (C == 256) hipLaunchKernelGGL(K<256>, grid, block, 0, stream, a);                           \
        else if (C == 128) hipLaunchKernelGGL(K<128>, grid, block, 0, stream, a);                      \
        else hipLaunchKernelGGL(K<64>, grid, block, 0, stream, a);                                     \
        EQH_CHECK_LAUNCH();                                                                            \
    } while (0)
    switch (stage) {
        case HG_CONV_F1:
            if (!q->in0 || !f1_ok(q->w0, q->w1, q->b0, q->g0, q->be0, q->out0, q->out1, q->out2)) return EQH_ERR_ARG;
            PN_LAUNCH(k_conv_f1);
            return EQH_OK;
        case HG_CONV_F2:
            if (!need({q->in0, q->rowptr, q->col, q->w0, q->bias_out, q->out0, q->out1})) return EQH_ERR_ARG;
            PN_LAUNCH(k_conv_f2);
            return EQH_OK;
        case HG_CONV_F3:
            if (!need({q->in0, q->in1, q->w0, q->b0, q->g0, q->be0, q->w1, q->bias_out, q->out0, q->out1, q->out2})) return EQH_ERR_ARG;
            if (q->tail && !f1_ok(q->w2, q->w3, q->b1, q->g1, q->be1, q->out3, q->out4, q->out5)) return EQH_ERR_ARG;
            PN_LAUNCH(k_conv_f3);
            return EQH_OK;
        case HG_CONV_B3: {
            if (!need({q->in0, q->w0, q->w1, q->in2, q->b0, q->g0, q->out1, q->out2, q->slab, q->dbias, q->dgamma, q->dbeta})) return EQH_ERR_ARG;
            if (q->in1 && !q->out0) return EQH_ERR_ARG;
            PN_LAUNCH(k_conv_b3);
            return eqh_reduce_slabs3_async(q->slab, blocks, 3 * (int64_t)C, q->dbias, q->dgamma, q->dbeta, C, C, q->accumulate, stream);
        }
        case HG_CONV_B1: {
            if (!need({q->in0, q->rowptr, q->col, q->in1, q->in2, q->b0, q->g0, q->w0, q->out0, q->slab, q->dbias, q->dgamma, q->dbeta}))
                return EQH_ERR_ARG;
            if (!q->tail && !q->out1) return EQH_ERR_ARG;
            if (q->tail && !need({q->in3, q->w1, q->w2, q->out5, q->b1, q->g1, q->out2, q->out3, q->out4, q->slab2, q->dbias2,
                                  q->dgamma2, q->dbeta2}))
                return EQH_ERR_ARG;
            PN_LAUNCH(k_conv_b1);
            int rc = eqh_reduce_slabs3_async(q->slab, blocks, 3 * (int64_t)C, q->dbias, q->dgamma, q->dbeta, C, C, q->accumulate, stream);
            if (rc || !q->tail) return rc;
            return eqh_reduce_slabs3_async(q->slab2, blocks, 3 * (int64_t)C, q->dbias2, q->dgamma2, q->dbeta2, C, C, q->accumulate, stream);
        }
        case HG_EGNN_NODE_F:
            if (!need({q->in0, q->in1, q->in2, q->w0, q->w1, q->w2, q->b0, q->bias_out, q->out0, q->out1, q->out2, q->out3})) return EQH_ERR_ARG;
            PN_LAUNCH(k_node_f);
            return EQH_OK;
        case HG_EGNN_NODE_B:
            if (!need({q->in0, q->in1, q->w0, q->w1, q->w2, q->out0, q->out1})) return EQH_ERR_ARG;
            PN_LAUNCH(k_node_b);
            return EQH_OK;
        default:
            return EQH_ERR_ARG;
    }
#undef PN_LAUNCH
}

#ifdef PN_STAMPS
extern "C" int hg_panel_debug_stamps(void* buf) {
    unsigned long long* q = static_cast<unsigned long long*>(buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(pn_stamp_buf), &q, sizeof(q)) == hipSuccess ? EQH_OK : EQH_ERR_ARG;
}
#endif

extern "C" size_t hg_panel_pack_bytes(int32_t K, int32_t N) {
    if (K <= 0 || N <= 0 || (K & 15) || (N & 31)) return 0;
    return (size_t)K * (size_t)N * 6;
}

extern "C" int hg_panel_pack(int32_t n_items, const HgPanelPack* items, void* stream_) {
    if (n_items <= 0 || !items) return EQH_ERR_ARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    for (int i0 = 0; i0 < n_items; i0 += PN_MAXPACK) {
        PackBatch b;
        b.n = (n_items - i0 < PN_MAXPACK) ? n_items - i0 : PN_MAXPACK;
        b.first[0] = 0;
        for (int i = 0; i < b.n; ++i) {
            const HgPanelPack& q = items[i0 + i];
            if (!q.w || !q.dst || q.K <= 0 || q.N <= 0 || q.kstep0 < 0) return EQH_ERR_ARG;
            if ((q.K & 15) || (q.N & 31) || (q.ld & 3) || !eqh_aligned16(q.w) || !eqh_aligned16(q.dst)) return EQH_ERR_ALIGN;
            const int total = q.ksteps_total > 0 ? q.ksteps_total : q.K / 16;
            if (q.kstep0 + q.K / 16 > total) return EQH_ERR_ARG;
            const int n_valid = (q.n_valid > 0 && q.n_valid < q.N) ? q.n_valid : q.N;
            b.it[i] = PackItem{q.w, q.ld, static_cast<uint4*>(q.dst), q.K, q.N, q.trans ? 1 : 0, q.kstep0, total, n_valid};
            b.first[i + 1] = b.first[i] + (q.K / 16) * (q.N / 32);
        }
        for (int i = b.n; i < PN_MAXPACK; ++i) { b.it[i] = b.it[0]; b.first[i + 1] = b.first[b.n]; }
        const int units = b.first[b.n];
        hipLaunchKernelGGL(k_panel_pack, dim3((units + 3) / 4), dim3(256), 0, stream, b);
        EQH_CHECK_LAUNCH();
    }
    return EQH_OK;
}

extern "C" int hg_panel_gemm_f32(const float* a, int64_t lda, int64_t rows, int32_t C, const void* wpack, float alpha,
                                 const float* d, int64_t ldd, float beta, const float* bias, int32_t relu, float* c, int64_t ldc,
                                 void* stream_) {
    if (rows < 0 || !a || !wpack || !c) return EQH_ERR_ARG;
    if (!pn_width_ok(C)) return EQH_ERR_ARG;
    if ((lda & 3) || (ldc & 3) || (d && (ldd & 3)) || !eqh_aligned16(a) || !eqh_aligned16(c) || !eqh_aligned16(d) ||
        !eqh_aligned16(bias) || !eqh_aligned16(wpack))
        return EQH_ERR_ALIGN;
    if (rows >= ((int64_t)1 << 31) - 64) return EQH_ERR_RANGE;
    if (rows == 0) return EQH_OK;
    PanelPlain p{a, lda, (int)rows, static_cast<const uint4*>(wpack), alpha, beta, d, ldd, bias, relu, c, ldc};
    const dim3 grid((unsigned)((rows + PN_ROWS - 1) / PN_ROWS)), block(PN_THREADS);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (C == 256) hipLaunchKernelGGL(k_panel_plain<256>, grid, block, 0, stream, p);
    else if (C == 128) hipLaunchKernelGGL(k_panel_plain<128>, grid, block, 0, stream, p);
    else hipLaunchKernelGGL(k_panel_plain<64>, grid, block, 0, stream, p);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}
