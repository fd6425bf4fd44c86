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
//   * the panel's rows are produced by a row PROLOGUE (plain rows, a gathered mean over a CSR row, the per-incidence hidden
//     layer + mean of conv.py:175-177, a LayerNorm backward of a gathered sum), split (bf16x3.h) and laid into LDS as the A
//     image; they are 48 KB for all of K = 256, shared by the workgroup's wavefronts;
//   * products are the six bf16 MFMAs of gemm_x6.hip (fp32-grade results, tests compare with float64);
//   * the accumulators go through an fp32 LDS staging tile into ROW TILES (below), where the row EPILOGUE runs (scale, addend,
//     bias, ReLU, LayerNorm and its backward, SiLU), stores whole row segments, and -- in the chained forms -- lays the next
//     product's A image without leaving the workgroup.
// A panel needs the whole [C x C] weight (384 KB of planes at C = 256) through its CU's vector memory path: that, equal to the
// MFMA time of 192 MFMAs per SIMD (2.6 us at 2.4 GHz), is what bounds a product; with ~150 panels on 256 CUs the launch is one
// round.  MFMA 32 x 32 x 16 with the operands swapped (the accumulator holds C^T: a lane owns 4 consecutive columns of one row).
//
// Round 5: the number of wavefronts per workgroup is a template parameter NW (8 by default, 4 = round 4's geometry, kept for
// A/B runs through EQH_PANEL_WAVES).  A panel's life was 15.7 k cycles per product of which the MFMA loop was 7.7 k: the rest --
// A-image build, LayerNorm row phases, staging, slab reductions -- ran on ONE wavefront per SIMD, where every VALU instruction
// issues in 4 cycles instead of 2 and every latency is exposed.  With eight wavefronts the matrix work per SIMD is unchanged (two
// wavefronts, one column tile each, alternate on the pipe) and every row phase has twice the issue slots; the gathers run in the
// row-tile layout (below) instead of wave-per-row + a sink through LDS, and the column sums of a slab are formed in registers
// (v_permlane16/32_swap) with ONE pass through LDS instead of three staged tiles and seven barriers.
#include <cstdlib>
#include <initializer_list>

#include "common.h"
#include "bf16x3.h"
#include "rowln.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int PN_ROWS = 32;        // rows per panel = one MFMA row tile
constexpr int PN_STG_LD = 260;     // floats per staged row: 256 + 4 keeps the accumulators' 16-byte stores conflict-free
constexpr int PN_PF = 3;           // K steps (of 16) of weight fragments in flight per wavefront

// Geometry of a workgroup of NW wavefronts.  Matrix work: wavefront w multiplies the column tiles w, w + NW, ...  Row work
// runs on ROW TILES: a wavefront holds RPW = 32 / NW rows of the panel at once, LPR = 64 / RPW lanes per row -- lane
// (r = lane / LPR, c = lane % LPR) has the float4 at columns 4 (LPR j + c), j < C / (4 LPR) -- so a LayerNorm statistic is a
// sum over a lane's own registers plus log2(LPR) DPP steps, the sqrt / division of a row is computed once for all RPW rows, and a
// wave-instruction still moves whole 128-byte (256-byte at NW = 8) row segments.
template <int NW> struct Geo {
    static_assert(NW == 4 || NW == 8, "4 or 8 wavefronts per panel");
    static constexpr int THREADS = 64 * NW, RPW = PN_ROWS / NW, LPR = 64 / RPW;
};

#ifdef PN_STAMPS   // diagnostic build only (tools/panel_stamps*.py): per-wavefront s_memtime stamps of the phases
__device__ unsigned long long* pn_stamp_buf = nullptr;
#define PN_STAMP(slot)                                                                                               \
    do {                                                                                                             \
        if (pn_stamp_buf && (threadIdx.x & 63) == 0)                                                                  \
            pn_stamp_buf[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + (slot)] = __builtin_amdgcn_s_memtime(); \
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
    int k_major;         // image[k / 32][tile][k half][plane][lane] (gemm_x6.hip's pre-split B: a K step of ALL tiles contiguous)
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
    const int kg = it.kstep0 + kstep;
    uint4* d = it.k_major ? it.dst + ((int64_t)(((kg >> 1) * (it.N >> 5) + tile) * 2 + (kg & 1)) * 3) * 64 + lane
                          : it.dst + ((int64_t)(tile * it.ksteps_total + kg) * 3) * 64 + lane;
    d[0] = p0;
    d[64] = p1;
    d[128] = p2;
}

// ---- the A image: a panel's rows as bf16 planes in LDS ---------------------------------------------------------------------
// [plane 3][kstep KS][slot 64] x 16 bytes; the fragment of (plane, kstep) is 1 KB, lane (fh, fr) reads slot
// fh * 32 + (fr ^ swz), swz = ((kstep & 3) << 1) | fh: the xor keeps the 8-byte row-wise writes below (sixteen lanes of a row
// tile cover four K steps x two halves of ONE row) on 32 distinct banks, and a ds_read_b128 of a fragment stays a permutation of
// its 64 slots inside each hardware lane group.
template <int KS>
__device__ __forceinline__ int a_slot(int kstep, int fh, int row) {
    return kstep * 64 + fh * 32 + (row ^ (((kstep & 3) << 1) | fh));
}

// a lane with v = row[4 k4 .. 4 k4 + 3]: its 8 bytes of each plane
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

// ---- the product: acc[g][j] (+)= A image . W image g for the column tiles wave + NW j --------------------------------------
// NG products share the A image (conv.py:172,176: X feeds W1's first Linear and the node half of W2's); the weight stream
// runs on across them, PN_PF K steps ahead of the MFMAs.
template <int KS, int NTW, int NG, int NW>
struct WStream {
    uint4 q[PN_PF][NTW][3];
    const uint4* base[NG][NTW];
    __device__ __forceinline__ void init(int g, const uint4* __restrict__ w, int wave, int lane, int n_tiles = 1 << 30) {
#pragma unroll
        for (int j = 0; j < NTW; ++j) {       // (a tile past the image re-reads tile 0; its product is not staged)
            const int tile = wave + NW * j < n_tiles ? wave + NW * j : 0;
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

template <int KS, int NTW, int NG, int NW>
__device__ __forceinline__ void panel_mma(const uint4* __restrict__ img, WStream<KS, NTW, NG, NW>& ws, f32x16 (&acc)[NG][NTW], int lane) {
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
// Product G of a SUM of products over different A images (k_panel_sum): K steps [G KS, (G + 1) KS) of the weight stream, which
// runs on across the products as in panel_mma, against the image of product G; every product adds into the same accumulators.
template <int KS, int NTW, int NG, int NW, int G>
__device__ __forceinline__ void panel_mma_part(const uint4* __restrict__ img, WStream<KS, NTW, NG, NW>& ws, f32x16 (&acc)[1][NTW], int lane) {
    const int fh = lane >> 5, fr = lane & 31;
    uint4 af[2][3];
    auto a_read = [&](int ks) {
        const uint4* ap = img + a_slot<KS>(ks, fh, fr);
        af[ks & 1][0] = ap[0];
        af[ks & 1][1] = ap[KS * 64];
        af[ks & 1][2] = ap[KS * 128];
    };
    a_read(0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int kk = G * KS + ks, slot = kk % PN_PF;
        if (ks + 1 < KS) a_read(ks + 1);
        const bf16x8 a0 = __builtin_bit_cast(bf16x8, af[ks & 1][0]);
        const bf16x8 a1 = __builtin_bit_cast(bf16x8, af[ks & 1][1]);
        const bf16x8 a2 = __builtin_bit_cast(bf16x8, af[ks & 1][2]);
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const bf16x8 b0 = __builtin_bit_cast(bf16x8, ws.q[slot][j][0]);
            const bf16x8 b1 = __builtin_bit_cast(bf16x8, ws.q[slot][j][1]);
            const bf16x8 b2 = __builtin_bit_cast(bf16x8, ws.q[slot][j][2]);
            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a1, acc[0][j], 0, 0, 0);
            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b2, a0, acc[0][j], 0, 0, 0);
            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a2, acc[0][j], 0, 0, 0);
            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a0, acc[0][j], 0, 0, 0);
            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a1, acc[0][j], 0, 0, 0);
            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a0, acc[0][j], 0, 0, 0);
        }
        if (kk + PN_PF < KS * NG) ws.fetch(slot, kk + PN_PF);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int NTW, int NW, int LD = PN_STG_LD>
__device__ __forceinline__ void acc_to_staging(float* __restrict__ stg, const f32x16 (&acc)[NTW], int wave, int lane, int n_tiles = 1 << 30) {
    const int fh = lane >> 5, fr = lane & 31;
#pragma unroll
    for (int j = 0; j < NTW; ++j)
        if (wave + NW * j < n_tiles) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(stg + fr * LD + (wave + NW * j) * 32 + 8 * g + 4 * fh) =
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

// ---- row tiles --------------------------------------------------------------------------------------------------------------
template <int C, int NW> struct PnShape {
    using G = Geo<NW>;
    static constexpr int KS = C / 16, NT = C / 32, NTW = (NT + NW - 1) / NW, NJ = C / (4 * G::LPR);
    static_assert(NJ >= 1, "C too narrow for this geometry");
};

template <int C, int NW> struct RowTile { float4 v[PnShape<C, NW>::NJ]; };

// lane -> (its row of the wavefront's RPW, its column quad); rows past the end of the matrix are clamped for loads
template <int NW>
struct RtPos {
    int lrow, c, c4, row, rowc;
    bool live;
    __device__ __forceinline__ RtPos(int r0, int rows, int wave, int lane) {
        using G = Geo<NW>;
        lrow = wave * G::RPW + lane / G::LPR;
        c = lane % G::LPR;
        c4 = c * 4;
        row = r0 + lrow;
        live = row < rows;
        rowc = live ? row : rows - 1;
    }
};

// sum over the LPR lanes of a row (every lane gets the total)
template <int LPR>
__device__ __forceinline__ float row_sum(float v) {
    v += dpp_move<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);   // row_half_mirror: lane i <- lane 7 - i of its group of eight
    if constexpr (LPR == 16) v += dpp_move<0x140>(v);   // row_mirror: lane i <- lane 15 - i of its row of sixteen
    return v;
}

template <int C, int NW>
__device__ __forceinline__ void rt_load(RowTile<C, NW>& t, const float* __restrict__ base, int64_t ld, int row, int c4) {
#pragma unroll
    for (int j = 0; j < PnShape<C, NW>::NJ; ++j)
        t.v[j] = *reinterpret_cast<const float4*>(base + (int64_t)row * ld + 4 * Geo<NW>::LPR * j + c4);
}
template <int C, int NW>
__device__ __forceinline__ void rt_store(const RowTile<C, NW>& t, float* __restrict__ base, int64_t ld, int row, int c4) {
#pragma unroll
    for (int j = 0; j < PnShape<C, NW>::NJ; ++j)
        *reinterpret_cast<float4*>(base + (int64_t)row * ld + 4 * Geo<NW>::LPR * j + c4) = t.v[j];
}
template <int C, int NW>
__device__ __forceinline__ void rt_load_vec(RowTile<C, NW>& t, const float* __restrict__ vec, int c4) {
#pragma unroll
    for (int j = 0; j < PnShape<C, NW>::NJ; ++j) t.v[j] = *reinterpret_cast<const float4*>(vec + 4 * Geo<NW>::LPR * j + c4);
}
// the tile's rows -> the A image (K offset kb4 = k / 4 of the tile's first column)
template <int C, int NW, int KS>
__device__ __forceinline__ void rt_a_put(const RowTile<C, NW>& t, uint4* __restrict__ img, int lrow, int c, int kb4 = 0) {
#pragma unroll
    for (int j = 0; j < PnShape<C, NW>::NJ; ++j) a_put<KS>(img, lrow, kb4 + Geo<NW>::LPR * j + c, t.v[j]);
}
template <int C, int NW>
__device__ __forceinline__ void rt_zero(RowTile<C, NW>& t) {
#pragma unroll
    for (int j = 0; j < PnShape<C, NW>::NJ; ++j) t.v[j] = f4_zero();
}

// xhat(relu(pre + bias)) of a row: the shared first half of the LayerNorm forward and backward (mlp.py:93-97): two-pass mean /
// variance, correctly rounded sqrt and division (the arithmetic per element of rowln.h; only the order of the row sums differs)
template <int C, int NW>
__device__ __forceinline__ void rt_xhat(const RowTile<C, NW>& pre, const RowTile<C, NW>& bias, float eps, RowTile<C, NW>& x, unsigned& pos,
                                        float& rstd) {
    constexpr int NJ = PnShape<C, NW>::NJ, LPR = Geo<NW>::LPR;
    const float inv_c = 1.0f / (float)C;
    pos = 0u;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        float4 h = make_float4(pre.v[j].x + bias.v[j].x, pre.v[j].y + bias.v[j].y, pre.v[j].z + bias.v[j].z, pre.v[j].w + bias.v[j].w);
        pos |= (((h.x > 0.f) ? 1u : 0u) | ((h.y > 0.f) ? 2u : 0u) | ((h.z > 0.f) ? 4u : 0u) | ((h.w > 0.f) ? 8u : 0u)) << (4 * j);
        h.x = fmaxf(h.x, 0.f); h.y = fmaxf(h.y, 0.f); h.z = fmaxf(h.z, 0.f); h.w = fmaxf(h.w, 0.f);
        x.v[j] = h;
        s += (h.x + h.y) + (h.z + h.w);
    }
    const float mu = row_sum<LPR>(s) * inv_c;
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        float4 d = x.v[j];
        d.x -= mu; d.y -= mu; d.z -= mu; d.w -= mu;
        x.v[j] = d;
        ss += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
    }
    rstd = 1.0f / sqrtf(row_sum<LPR>(ss) * inv_c + eps);
#pragma unroll
    for (int j = 0; j < NJ; ++j) { x.v[j].x *= rstd; x.v[j].y *= rstd; x.v[j].z *= rstd; x.v[j].w *= rstd; }
}

// y = gamma * xhat(relu(pre + bias)) + beta per row (mlp.py:93-97)
template <int C, int NW>
__device__ __forceinline__ void rt_ln_fwd(const RowTile<C, NW>& pre, const RowTile<C, NW>& bias, const RowTile<C, NW>& gam,
                                          const RowTile<C, NW>& bet, float eps, RowTile<C, NW>& y) {
    unsigned pos;
    float rstd;
    rt_xhat<C, NW>(pre, bias, eps, y, pos, rstd);
#pragma unroll
    for (int j = 0; j < PnShape<C, NW>::NJ; ++j) {
        y.v[j].x = fmaf(gam.v[j].x, y.v[j].x, bet.v[j].x); y.v[j].y = fmaf(gam.v[j].y, y.v[j].y, bet.v[j].y);
        y.v[j].z = fmaf(gam.v[j].z, y.v[j].z, bet.v[j].z); y.v[j].w = fmaf(gam.v[j].w, y.v[j].w, bet.v[j].w);
    }
}

// dpre = gradient of the pre-activation given dy = d LN output (the formulas of k_rowln_bwd); the lane's terms of d bias,
// d gamma, d beta are ADDED to a_db / a_dg / a_dbeta when `count` (rows past the end of the matrix are not counted)
template <int C, int NW>
__device__ __forceinline__ void rt_ln_bwd(const RowTile<C, NW>& pre, const RowTile<C, NW>& bias, const RowTile<C, NW>& gam,
                                          const RowTile<C, NW>& dy, float eps, bool count, RowTile<C, NW>& dpre, RowTile<C, NW>& a_db,
                                          RowTile<C, NW>& a_dg, RowTile<C, NW>& a_dbeta) {
    constexpr int NJ = PnShape<C, NW>::NJ, LPR = Geo<NW>::LPR;
    const float inv_c = 1.0f / (float)C;
    RowTile<C, NW> x;
    unsigned pos;
    float rstd;
    rt_xhat<C, NW>(pre, bias, eps, x, pos, rstd);
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const float4 xh = x.v[j];
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
    m1 = row_sum<LPR>(m1) * inv_c;
    m2 = row_sum<LPR>(m2) * inv_c;
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

// plain LayerNorm of a row (no bias, no ReLU: nn.LayerNorm on node rows, egnn_layer.py:192,360): xhat and 1 / std
template <int C, int NW>
__device__ __forceinline__ void rt_xhat_plain(const RowTile<C, NW>& in, float eps, RowTile<C, NW>& x, float& rstd) {
    constexpr int NJ = PnShape<C, NW>::NJ, LPR = Geo<NW>::LPR;
    const float inv_c = 1.0f / (float)C;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) s += (in.v[j].x + in.v[j].y) + (in.v[j].z + in.v[j].w);
    const float mu = row_sum<LPR>(s) * inv_c;
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        float4 d = in.v[j];
        d.x -= mu; d.y -= mu; d.z -= mu; d.w -= mu;
        x.v[j] = d;
        ss += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
    }
    rstd = 1.0f / sqrtf(row_sum<LPR>(ss) * inv_c + eps);
#pragma unroll
    for (int j = 0; j < NJ; ++j) { x.v[j].x *= rstd; x.v[j].y *= rstd; x.v[j].z *= rstd; x.v[j].w *= rstd; }
}
// ... and its backward: dx = rstd (g - mean(g) - xhat mean(g xhat)), g = dy gamma; the lane's terms of d gamma / d beta are ADDED
// to a_dg / a_dbeta when `count`
template <int C, int NW>
__device__ __forceinline__ void rt_ln_plain_bwd(const RowTile<C, NW>& xh, float rstd, const RowTile<C, NW>& gam, const RowTile<C, NW>& dy,
                                                bool count, RowTile<C, NW>& dx, RowTile<C, NW>& a_dg, RowTile<C, NW>& a_dbeta) {
    constexpr int NJ = PnShape<C, NW>::NJ, LPR = Geo<NW>::LPR;
    const float inv_c = 1.0f / (float)C;
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const float4 x = xh.v[j];
        float4 d = dy.v[j];
        if (count) {
            f4_add(a_dbeta.v[j], d);
            a_dg.v[j].x = fmaf(d.x, x.x, a_dg.v[j].x); a_dg.v[j].y = fmaf(d.y, x.y, a_dg.v[j].y);
            a_dg.v[j].z = fmaf(d.z, x.z, a_dg.v[j].z); a_dg.v[j].w = fmaf(d.w, x.w, a_dg.v[j].w);
        }
        d.x *= gam.v[j].x; d.y *= gam.v[j].y; d.z *= gam.v[j].z; d.w *= gam.v[j].w;
        dx.v[j] = d;
        m1 += (d.x + d.y) + (d.z + d.w);
        m2 += (d.x * x.x + d.y * x.y) + (d.z * x.z + d.w * x.w);
    }
    m1 = row_sum<LPR>(m1) * inv_c;
    m2 = row_sum<LPR>(m2) * inv_c;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const float4 d = dx.v[j], x = xh.v[j];
        dx.v[j] = make_float4(rstd * (d.x - m1 - x.x * m2), rstd * (d.y - m1 - x.y * m2), rstd * (d.z - m1 - x.z * m2), rstd * (d.w - m1 - x.w * m2));
    }
}

// sum of v over the RPW rows a wavefront holds (lanes with the same c): every lane gets the total.  The rows of a DPP row of
// sixteen lanes meet by a rotation, the DPP rows by v_permlane16_swap / v_permlane32_swap (gfx950: VALU, no LDS).
template <int NW>
__device__ __forceinline__ float wave_rows_sum(float v) {
    if constexpr (Geo<NW>::LPR == 8) v += dpp_move<0x128>(v);   // row_ror:8 -- the two rows of a DPP row
    {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);       // DPP rows 0 + 1, 2 + 3
    }
    {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);       // ... + the other half of the wavefront
    }
    return v;
}

// The workgroup's [d bias | d gamma | d beta]: every lane holds its row's terms; a wavefront sums its rows in registers, the NW
// partial vectors meet in LDS (`red`: NW x 3 x C floats, may alias the staging tile) and are summed per column in wavefront
// order (fixed order: bitwise reproducible) -> slab[3][C].  All threads call; `red` is free again afterwards.
template <int C, int NW>
__device__ __forceinline__ void write_slab(float* __restrict__ red, float* __restrict__ slab, const RowTile<C, NW>& a_db,
                                           const RowTile<C, NW>& a_dg, const RowTile<C, NW>& a_dbeta, int wave, int lane) {
    constexpr int NJ = PnShape<C, NW>::NJ, LPR = Geo<NW>::LPR;
    __syncthreads();                      // (the staging tile's last readers are done)
#pragma unroll
    for (int which = 0; which < 3; ++which) {
        const RowTile<C, NW>& t = which == 0 ? a_db : (which == 1 ? a_dg : a_dbeta);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float4 s;
            s.x = wave_rows_sum<NW>(t.v[j].x); s.y = wave_rows_sum<NW>(t.v[j].y);
            s.z = wave_rows_sum<NW>(t.v[j].z); s.w = wave_rows_sum<NW>(t.v[j].w);
            if (lane < LPR) *reinterpret_cast<float4*>(red + (wave * 3 + which) * C + 4 * (LPR * j + lane)) = s;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * C; i += Geo<NW>::THREADS) {
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) acc += red[w * 3 * C + i];
        slab[i] = acc;
    }
    __syncthreads();
}

// sum_q w[q] * src[col[q]] over the CSR row of every row of the tile, in the row-tile layout: the RPW rows of a wavefront walk
// their entries together, slot by slot (a hyperedge has 2-3 nodes, a node 2-3 hyperedges), U slots -- U * NJ 16-byte loads per
// lane -- in flight; entries are added in CSR order (the order of the wave-per-row kernels: same sums bit for bit).
template <int C, int NW, bool WEIGHTED>
__device__ __forceinline__ void rt_gather_sum(const float* __restrict__ src, const int* __restrict__ rowptr, const int* __restrict__ col,
                                              const float* __restrict__ wq, const RtPos<NW>& P, RowTile<C, NW>& sum, int& deg) {
    constexpr int NJ = PnShape<C, NW>::NJ, LPR = Geo<NW>::LPR, U = 4;
    int beg = 0, end = 0;
    if (P.live) {
        beg = rowptr[P.row];
        end = rowptr[P.row + 1];
    }
    deg = end - beg;
    rt_zero<C, NW>(sum);
    for (int i0 = 0; __builtin_amdgcn_ballot_w64(beg + i0 < end) != 0ull; i0 += U) {
        int idx[U];
        float w[U];
        float4 d[U][NJ];
#pragma unroll
        for (int t = 0; t < U; ++t) {
            const int q = beg + i0 + t;
            idx[t] = -1;
            w[t] = 0.f;
            if (q < end) {
                idx[t] = col[q];
                if constexpr (WEIGHTED) w[t] = wq[q];
            }
        }
#pragma unroll
        for (int t = 0; t < U; ++t)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                d[t][j] = idx[t] >= 0 ? *reinterpret_cast<const float4*>(src + (int64_t)idx[t] * C + 4 * LPR * j + P.c4) : f4_zero();
#pragma unroll
        for (int t = 0; t < U; ++t)
            if (idx[t] >= 0) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if constexpr (WEIGHTED) f4_fma(sum.v[j], d[t][j], w[t]);
                    else f4_add(sum.v[j], d[t][j]);
                }
            }
    }
}

// s[v] = gamma * mean_{e of v} xhat(relu(pa[v] + qb[e])) + beta * [deg v > 0]  (conv.py:175-177 after moving W2's last Linear
// behind the mean: the per-incidence hidden layer of W2 and the hyperedge -> node mean; the arithmetic of k_inc_fwd_col,
// incidence.hip) for the rows of the tile, in the row-tile layout: a LayerNorm per incidence costs a lane NJ float4 of work
// and 2 log2(LPR) DPP steps, for all RPW rows of the wavefront at once.
template <int C, int NW>
__device__ __forceinline__ void rt_incidence_mean(const float* __restrict__ pa, const float* __restrict__ qb, const int* __restrict__ rowptr,
                                                  const int* __restrict__ col, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                  float eps, const RtPos<NW>& P, RowTile<C, NW>& s) {
    constexpr int NJ = PnShape<C, NW>::NJ, LPR = Geo<NW>::LPR, U = 2;
    int beg = 0, end = 0;
    if (P.live) {
        beg = rowptr[P.row];
        end = rowptr[P.row + 1];
    }
    RowTile<C, NW> own, acc;
    rt_load<C, NW>(own, pa, C, P.rowc, P.c4);
    rt_zero<C, NW>(acc);
    for (int i0 = 0; __builtin_amdgcn_ballot_w64(beg + i0 < end) != 0ull; i0 += U) {
        int idx[U];
        RowTile<C, NW> d[U];
#pragma unroll
        for (int t = 0; t < U; ++t) {
            const int q = beg + i0 + t;
            idx[t] = q < end ? col[q] : -1;
        }
#pragma unroll
        for (int t = 0; t < U; ++t)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                d[t].v[j] = idx[t] >= 0 ? *reinterpret_cast<const float4*>(qb + (int64_t)idx[t] * C + 4 * LPR * j + P.c4) : f4_zero();
#pragma unroll
        for (int t = 0; t < U; ++t) {
            RowTile<C, NW> x;
            unsigned pos;
            float rstd;
            rt_xhat<C, NW>(own, d[t], eps, x, pos, rstd);        // (every lane of the wavefront takes part in the DPP sums)
            if (idx[t] >= 0) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) f4_add(acc.v[j], x.v[j]);
            }
        }
    }
    const int deg = end - beg;
    const float den = deg > 1 ? (float)deg : 1.0f;
    const float bscale = deg > 0 ? 1.0f : 0.0f;
    RowTile<C, NW> gv, bv;
    rt_load_vec<C, NW>(gv, gamma, P.c4);
    rt_load_vec<C, NW>(bv, beta, P.c4);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        s.v[j].x = fmaf(gv.v[j].x, acc.v[j].x / den, bv.v[j].x * bscale);
        s.v[j].y = fmaf(gv.v[j].y, acc.v[j].y / den, bv.v[j].y * bscale);
        s.v[j].z = fmaf(gv.v[j].z, acc.v[j].z / den, bv.v[j].z * bscale);
        s.v[j].w = fmaf(gv.v[j].w, acc.v[j].w / den, bv.v[j].w * bscale);
    }
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

#define PN_KERNEL(NW_) __global__ void __launch_bounds__(64 * NW_) __attribute__((amdgpu_waves_per_eu(NW_ / 4, NW_ / 4)))

template <int C, int NW>
PN_KERNEL(NW) k_panel_plain(const PanelPlain p) {
    using S = PnShape<C, NW>;
    __shared__ uint4 s_img[3 * S::KS * 64];
    __shared__ float s_stg[PN_ROWS * PN_STG_LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const RtPos<NW> P((int)blockIdx.x * PN_ROWS, p.rows, wave, lane);
    const bool mul = wave < S::NT;        // (C = 64: two column tiles; the other wavefronts only move rows)
    PN_STAMP(0);
    RowTile<C, NW> x, d;
    rt_load<C, NW>(x, p.A, p.lda, P.rowc, P.c4);
    WStream<S::KS, S::NTW, 1, NW> ws;
    ws.init(0, p.W, mul ? wave : 0, lane);
    ws.prime();          // unconditional: loads inside a branch make every later wait conservative (idle waves re-read tile 0)
    if (p.D) rt_load<C, NW>(d, p.D, p.ldd, P.rowc, P.c4);
    else rt_zero<C, NW>(d);
    __builtin_amdgcn_sched_barrier(0);
    rt_a_put<C, NW, S::KS>(x, s_img, P.lrow, P.c);
    PN_STAMP(1);
    __syncthreads();
    PN_STAMP(2);
    f32x16 acc[1][S::NTW];
    acc_zero<1, S::NTW>(acc);
    if (mul) {
        panel_mma<S::KS, S::NTW, 1, NW>(s_img, ws, acc, lane);
        PN_STAMP(3);
        acc_to_staging<S::NTW, NW>(s_stg, acc[0], wave, lane);
    }
    __syncthreads();
    PN_STAMP(4);
    RowTile<C, NW> a, bv;
    rt_load<C, NW>(a, s_stg, PN_STG_LD, P.lrow, P.c4);
    if (p.bias) rt_load_vec<C, NW>(bv, p.bias, P.c4);
    else rt_zero<C, NW>(bv);
#pragma unroll
    for (int j = 0; j < S::NJ; ++j) {
        float4 o = make_float4(p.alpha * a.v[j].x, p.alpha * a.v[j].y, p.alpha * a.v[j].z, p.alpha * a.v[j].w);
        if (p.D) {
            o.x = fmaf(p.beta, d.v[j].x, o.x); o.y = fmaf(p.beta, d.v[j].y, o.y);
            o.z = fmaf(p.beta, d.v[j].z, o.z); o.w = fmaf(p.beta, d.v[j].w, o.w);
        }
        f4_add(o, bv.v[j]);
        if (p.relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        a.v[j] = o;
    }
    if (P.live) rt_store<C, NW>(a, p.Cout, p.ldc, P.row, P.c4);
    PN_STAMP(5);
}

// ---- the STREAMING product: C = act(alpha A W + beta D + bias) for MANY rows, A [rows, K], W a packed K x N image -------------
// (round 6; K, N <= 256: FAFormer's frame / edge Linears, [246 k x 256] . [256 x 256] and [1.97 M x 128] . [128 x 256] at the
// Molecule3D batch, fa_former_layer.py:61-120,241-289 -- 53 % of that step ran in the tiled x6 GEMM at 0.3-0.4 of its peak, its
// staging wavefronts and its multiplying wavefronts taking turns on the same SIMDs.)  One PERSISTENT workgroup per CU walks its
// panels; the weight image stays in the XCD's L2 and streams into registers exactly as in the one-panel kernels, and what made
// those cost 6.3 us per panel against 2.6 us of MFMAs -- the row prologue and epilogue, exposed once per launch -- is hidden:
// TWO A images, the next panel's rows requested before the MFMA loop and split into the idle image right behind it, so a panel
// costs its MFMA loop + one image build (~100 VALU instructions per wavefront) + the epilogue's LDS round trip.
// Two ROLES, as in gemm_x6.hip -- on gfx950 a wavefront's loads AND stores retire in order behind one counter (vmcnt), so a
// multiplying wavefront that also fetched the next panel's rows or stored the last panel's results would wait for HBM before
// every weight fragment from L2 (the first version of this kernel did: 8.7 us per panel).  Wavefronts 0-7 only stream weight
// fragments and multiply; wavefronts 8-15 (the ROW wavefronts) run the previous panel's epilogue out of the staging tile and
// build the next panel's A image in the meantime, with their rows requested two panels ahead.  Two barriers per panel.
#ifdef PN_STAMPS
__device__ int pn_debug_flags = 0;     // diagnostic build only: 1 = no result stores, 2 = every panel re-reads the first panel's rows
#define PS_FLAG(bit) (pn_debug_flags & (bit))
#else
#define PS_FLAG(bit) 0
#endif
#ifdef PN_STAMPS
#define PS_STAMP(slot)                                                                                                  \
    do {                                                                                                                \
        if (pn_stamp_buf && (threadIdx.x & 63) == 0 && it == 3)                                                          \
            pn_stamp_buf[((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 16 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define PS_STAMP(slot) do { } while (0)
#endif
template <int K, int N>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) k_panel_stream(const PanelPlain p, int n_panels) {
    constexpr int NW = 8, KS = K / 16, NT = N / 32, LDS_LD = N + 4;
    static_assert(NT >= 1 && NT <= NW && K % 64 == 0 && N % 64 == 0, "one column tile per multiplying wavefront; whole row-tile quads");
    __shared__ uint4 s_img[2][3 * KS * 64];
    __shared__ float s_stg[PN_ROWS * LDS_LD];
    const int lane = threadIdx.x & 63, wave_all = threadIdx.x >> 6;
    const bool row_role = wave_all >= NW;                 // wavefront-uniform
    const int wave = wave_all & (NW - 1);
    const int step = (int)gridDim.x, first = (int)blockIdx.x;
    if (row_role) {
        // the row wavefronts' few instructions go in front of the multipliers' MFMA streams at the issue port (priority, then
        // age: at equal priority a panel's image build took 6.7 k cycles for ~100 VALU instructions per wavefront)
        __builtin_amdgcn_s_setprio(3);
        const RtPos<NW> L(0, PN_ROWS, wave, lane);        // lane constants (local row, column quad) of the row-tile layout
        auto rowc = [&](int pnl) {                        // this lane's (clamped) row of panel pnl; panels past the end re-read the last
            if (PS_FLAG(2)) pnl = first;
            const int r = (pnl < n_panels ? pnl : n_panels - 1) * PN_ROWS + L.lrow;
            return r < p.rows ? r : p.rows - 1;
        };
        // the epilogue of one panel out of the staging tile, one float4 column group at a time (the row role shares the
        // kernel's 128-register budget with the multiplying role: whole row tiles of a, d and the bias spilled)
        // the epilogue of one panel out of the staging tile.  The bias is loaded ONCE (the same columns for every panel: fetched
        // per panel and per column group, each load sat in front of its use -- 4 x an L2 round trip per panel); the addend's
        // rows, when there is one, are requested before the staging tile is read
        RowTile<N, NW> bv;
        if (p.bias) rt_load_vec<N, NW>(bv, p.bias, L.c4);
        else rt_zero<N, NW>(bv);
        auto epilogue = [&](int pnl, const RtPos<NW>& L) {
            const int prow = pnl * PN_ROWS + L.lrow;
            const bool live = prow < p.rows;
            const int prc = live ? prow : p.rows - 1;
            RowTile<N, NW> a;
            if (p.D) {
                RowTile<N, NW> d;
                rt_load<N, NW>(d, p.D, p.ldd, prc, L.c4);
                rt_load<N, NW>(a, s_stg, LDS_LD, L.lrow, L.c4);
#pragma unroll
                for (int j = 0; j < PnShape<N, NW>::NJ; ++j) {
                    a.v[j].x = fmaf(p.beta, d.v[j].x, p.alpha * a.v[j].x); a.v[j].y = fmaf(p.beta, d.v[j].y, p.alpha * a.v[j].y);
                    a.v[j].z = fmaf(p.beta, d.v[j].z, p.alpha * a.v[j].z); a.v[j].w = fmaf(p.beta, d.v[j].w, p.alpha * a.v[j].w);
                }
            } else {
                rt_load<N, NW>(a, s_stg, LDS_LD, L.lrow, L.c4);
#pragma unroll
                for (int j = 0; j < PnShape<N, NW>::NJ; ++j) {
                    a.v[j].x *= p.alpha; a.v[j].y *= p.alpha; a.v[j].z *= p.alpha; a.v[j].w *= p.alpha;
                }
            }
#pragma unroll
            for (int j = 0; j < PnShape<N, NW>::NJ; ++j) {
                f4_add(a.v[j], bv.v[j]);
                if (p.relu) { a.v[j].x = fmaxf(a.v[j].x, 0.f); a.v[j].y = fmaxf(a.v[j].y, 0.f); a.v[j].z = fmaxf(a.v[j].z, 0.f); a.v[j].w = fmaxf(a.v[j].w, 0.f); }
            }
            if (live && !PS_FLAG(1)) rt_store<N, NW>(a, p.Cout, p.ldc, prow, L.c4);
        };
        RowTile<K, NW> xa, xb;                            // rows of the panels at distance 1 and 2
        rt_load<K, NW>(xa, p.A, p.lda, rowc(first), L.c4);
        rt_load<K, NW>(xb, p.A, p.lda, rowc(first + step), L.c4);
        rt_a_put<K, NW, KS>(xa, s_img[0], L.lrow, L.c);
        xa = xb;
        rt_load<K, NW>(xb, p.A, p.lda, rowc(first + 2 * step), L.c4);
        __syncthreads();                                  // (A) image of the first panel is in place
        int cur = 0, last = first;
        [[maybe_unused]] int it = -1;
        for (int pnl = first; pnl < n_panels; pnl += step) {
            ++it;
            PS_STAMP(0);
            // (the lane index is laundered once per panel: the swizzled LDS addresses of a panel's life depend on the lane only,
            // and hoisted out of this loop -- dozens of them -- they overflowed the 128-register budget into scratch)
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const RtPos<NW> L(0, PN_ROWS, wave, ln);
            // while the multipliers work on panel pnl: the NEXT panel's image, then the PREVIOUS panel's epilogue
            if (pnl + step < n_panels) rt_a_put<K, NW, KS>(xa, s_img[cur ^ 1], L.lrow, L.c);
            PS_STAMP(1);
            xa = xb;
            rt_load<K, NW>(xb, p.A, p.lda, rowc(pnl + 3 * step), L.c4);
            if (pnl != first) epilogue(pnl - step, L);
            PS_STAMP(2);
            __syncthreads();                              // (B) staging read, next image written | MFMA loop done
            PS_STAMP(3);
            __syncthreads();                              // (A) staging of panel pnl written
            PS_STAMP(4);
            cur ^= 1;
            last = pnl;
        }
        epilogue(last, L);
        return;
    }
    // ---- multiplying wavefronts
    const bool mul = wave < NT;
    WStream<KS, 1, 1, NW> ws;
    ws.init(0, p.W, mul ? wave : 0, lane);
    __syncthreads();                                      // (A)
    int cur = 0;
    [[maybe_unused]] int it = -1;
    for (int pnl = first; pnl < n_panels; pnl += step) {
        ++it;
        PS_STAMP(0);
        int ln = lane;                                    // (laundered per panel: see the row role)
        asm volatile("" : "+v"(ln));
        // ... and the weight stream's base: the 48 fragment addresses of a panel (1 KB apart: beyond the immediate offset of a
        // load) are loop-invariant, and hoisted they are 96 registers -- spilled, each reload then sat, with a vmcnt(0), in
        // front of its fetch: 12.9 us per panel)
        asm volatile("" : "+v"(ws.base[0][0]));
        ws.prime();
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc[1][1];
        acc_zero<1, 1>(acc);
        if (mul) panel_mma<KS, 1, 1, NW>(s_img[cur], ws, acc, ln);
        PS_STAMP(2);
        __syncthreads();                                  // (B)
        PS_STAMP(3);
        if (mul) acc_to_staging<1, NW, LDS_LD>(s_stg, acc[0], wave, ln);
        __syncthreads();                                  // (A)
        PS_STAMP(4);
        cur ^= 1;
    }
}

// Measurement aid (EQH_PANEL_PAIR=1, C = 256): the plain product as a COLUMN-SPLIT PAIR -- two workgroups of four
// wavefronts per 32-row panel, blockIdx.y owning 128 of the 256 output columns: half the weight image (192 KB) and half the
// MFMAs (96 per SIMD) per workgroup, twice the workgroups.  Each half still needs the panel's whole rows as its A image (K =
// 256), so the row prologue -- load, split into bf16 planes, LDS image -- is done twice, by half as many wavefronts each.
template <int C>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) k_panel_plain_pair(const PanelPlain p) {
    constexpr int NW = 4;
    using S = PnShape<C, NW>;
    constexpr int HT = S::NT / 2;          // column tiles per half (4 at C = 256): one per wavefront
    static_assert(HT == NW, "one column tile per wavefront");
    __shared__ uint4 s_img[3 * S::KS * 64];
    __shared__ float s_stg[PN_ROWS * (C / 2 + 4)];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = blockIdx.y;
    const RtPos<NW> P((int)blockIdx.x * PN_ROWS, p.rows, wave, lane);
    RowTile<C, NW> x;
    rt_load<C, NW>(x, p.A, p.lda, P.rowc, P.c4);
    WStream<S::KS, 1, 1, NW> ws;
    ws.init(0, p.W + (int64_t)(half * HT * S::KS) * 3 * 64, wave, lane);
    ws.prime();
    __builtin_amdgcn_sched_barrier(0);
    rt_a_put<C, NW, S::KS>(x, s_img, P.lrow, P.c);
    __syncthreads();
    f32x16 acc[1][1];
    acc_zero<1, 1>(acc);
    panel_mma<S::KS, 1, 1, NW>(s_img, ws, acc, lane);
    acc_to_staging<1, NW, C / 2 + 4>(s_stg, acc[0], wave, lane);
    __syncthreads();
    // epilogue on this half's 32 x 128 block: 8 lanes per row, four float4 each
    const int row = threadIdx.x >> 3, c = threadIdx.x & 7, grow = (int)blockIdx.x * PN_ROWS + row;
    if (grow < p.rows) {
#pragma unroll
        for (int j = 0; j < C / 64; ++j) {
            const int col = 4 * (8 * j + c);
            const float4 a = *reinterpret_cast<const float4*>(s_stg + row * (C / 2 + 4) + col);
            float4 o = make_float4(p.alpha * a.x, p.alpha * a.y, p.alpha * a.z, p.alpha * a.w);
            const int gcol = half * (C / 2) + col;
            if (p.D) {
                const float4 d = *reinterpret_cast<const float4*>(p.D + (int64_t)grow * p.ldd + gcol);
                o.x = fmaf(p.beta, d.x, o.x); o.y = fmaf(p.beta, d.y, o.y); o.z = fmaf(p.beta, d.z, o.z); o.w = fmaf(p.beta, d.w, o.w);
            }
            if (p.bias) f4_add(o, *reinterpret_cast<const float4*>(p.bias + gcol));
            if (p.relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
            *reinterpret_cast<float4*>(p.Cout + (int64_t)grow * p.ldc + gcol) = o;
        }
    }
}

// ---- several products of ONE row block: out_g = A W_g + rw_g[row] * bias_g + D_g, g < NG <= 3 --------------------------------
// (MHNNConv, conv.py:87-101 after the split of every first Linear by input block: X feeds W1's and W3's node halves and W4's own
// half, E feeds W1's hyperedge half and W2's own half -- one A image, NG weight streams.  rw: per-row weight of the bias, the
// [row has an incidence] factor that a mean over an empty row leaves on the bias of the Linear behind it.)
struct PanelMulti {
    const float* A;
    int64_t lda;
    int rows;
    const uint4* W[3];
    const float* bias[3];
    const float* rw[3];
    const float* D[3];
    int64_t ldd[3];
    float* out[3];
    int64_t ldo[3];
};

template <int C, int NW, int NG>
PN_KERNEL(NW) k_panel_multi(const PanelMulti p) {
    using S = PnShape<C, NW>;
    __shared__ uint4 s_img[3 * S::KS * 64];
    __shared__ float s_stg[NG][PN_ROWS * PN_STG_LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const RtPos<NW> P((int)blockIdx.x * PN_ROWS, p.rows, wave, lane);
    const bool mul = wave < S::NT;
    RowTile<C, NW> x;
    rt_load<C, NW>(x, p.A, p.lda, P.rowc, P.c4);
    WStream<S::KS, S::NTW, NG, NW> ws;
#pragma unroll
    for (int g = 0; g < NG; ++g) ws.init(g, p.W[g], mul ? wave : 0, lane);
    ws.prime();
    __builtin_amdgcn_sched_barrier(0);
    rt_a_put<C, NW, S::KS>(x, s_img, P.lrow, P.c);
    __syncthreads();
    f32x16 acc[NG][S::NTW];
    acc_zero<NG, S::NTW>(acc);
    if (mul) {
        panel_mma<S::KS, S::NTW, NG, NW>(s_img, ws, acc, lane);
#pragma unroll
        for (int g = 0; g < NG; ++g) acc_to_staging<S::NTW, NW>(s_stg[g], acc[g], wave, lane);
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        RowTile<C, NW> a, bv, d;
        rt_load<C, NW>(a, s_stg[g], PN_STG_LD, P.lrow, P.c4);
        if (p.bias[g]) rt_load_vec<C, NW>(bv, p.bias[g], P.c4);
        else rt_zero<C, NW>(bv);
        const float w = (p.bias[g] && p.rw[g]) ? p.rw[g][P.rowc] : 1.0f;
        if (p.D[g]) rt_load<C, NW>(d, p.D[g], p.ldd[g], P.rowc, P.c4);
        else rt_zero<C, NW>(d);
#pragma unroll
        for (int j = 0; j < S::NJ; ++j) {
            a.v[j].x = fmaf(w, bv.v[j].x, a.v[j].x) + d.v[j].x; a.v[j].y = fmaf(w, bv.v[j].y, a.v[j].y) + d.v[j].y;
            a.v[j].z = fmaf(w, bv.v[j].z, a.v[j].z) + d.v[j].z; a.v[j].w = fmaf(w, bv.v[j].w, a.v[j].w) + d.v[j].w;
        }
        if (P.live) rt_store<C, NW>(a, p.out[g], p.ldo[g], P.row, P.c4);
    }
}

// =============================================================================================================================
// The merged MHNNSConv application (conv.py:169-182 after layers.MHNNSConv._prepare_merged) on panels.
//   forward  F1: h1 = X W1a^T (stored raw, the LayerNorm backward recomputes from it), h1n = LN1(relu(h1 + b1a)), pa = X W2v^T
//            F2: hbar[e] = mean_{v in e} h1n[v]  (prologue: gathered mean over the hyperedge's nodes, conv.py:172-173)
//                qb = hbar w12^T + b12
//            F3: s[v] = gamma2 mean_{e of v} xhat(relu(pa[v] + qb[e])) + beta2   (prologue, round 5: the per-incidence hidden
//                layer + hyperedge -> node mean of conv.py:175-177, k_inc_fwd_col's arithmetic; or s read from memory),
//                u = scale * (s w23^T) + cw,  x3 = LN3(relu(u + b3a)),  Xn = act(x3 W3b^T + b3b)   [+ F1 of the next application]
//   backward B3: g = dXn * [Xn > 0],  dx3 = g W3b,  dpre = LN3bwd(u + b3a; dx3),  ds = scale * dpre w23
//            (dpa, dqb = k_inc_bwd_both(ds): incidence.hip, the HBM-bound aggregation backward, stays its own launch)
//            B1: dh1[v] = LN1bwd(h1[v] + b1a; sum_{e of v} (dqb[e] w12) / deg e)  (prologue + the folded product B2),
//                dX = [dh1 | dpa] . [W1a ; W2v]   [+ B3 of the previous application]
// The weight / bias / LayerNorm-vector gradients are formed outside from the stored rows (batched weight-gradient launch,
// column sums) and from the per-workgroup slabs.
// =============================================================================================================================
struct ConvPanelArgs {
    int rows;                       // rows of this stage's panels (nodes, or hyperedges for F2)
    float eps, scale, eps_inc;
    int relu, acc_first, tail;
    // operands / products (meaning per stage, see the kernels)
    const float* in0; const float* in1; const float* in2; const float* in3;
    int64_t ld0;
    const int* rowptr; const int* col; const float* wq;
    const uint4* w0; const uint4* w1; const uint4* w2; const uint4* w3;
    const float* b0; const float* g0; const float* be0;      // bias / gamma / beta of the first LayerNorm of the stage
    const float* b1; const float* g1; const float* be1;      // ... of the tail's
    const float* g_inc; const float* be_inc;                  // gamma / beta of the incidence LayerNorm (F3's prologue)
    const float* bias_out;                                    // bias of a plain Linear output
    float* out0; float* out1; float* out2; float* out3; float* out4; float* out5; float* out6;
    float* slab; float* slab2;
    float* acc_out;
    int* signal;                    // F2: a device counter the first thread of the launch bumps (eqh_signal_post folded in), or null
};

// ---- F1: X -> h1 (raw), h1n = LN1(relu(h1 + b1a)), pa -------------------------------------------------------------------------
// the A image holds X's panel; w_a = W1a image (x W^T), w_b = W2v image
template <int C, int NW>
__device__ __forceinline__ void stage_f1(const ConvPanelArgs& p, uint4* __restrict__ s_img, float* __restrict__ s_stg,
                                         float* __restrict__ s_stg2, const uint4* w_a, const uint4* w_b, const float* b1a,
                                         const float* g1, const float* be1, float* h1, float* h1n, float* pa, const RtPos<NW>& P, int wave,
                                         int lane, bool mul) {
    using S = PnShape<C, NW>;
    WStream<S::KS, S::NTW, 2, NW> ws;
    ws.init(0, w_a, mul ? wave : 0, lane);
    ws.init(1, w_b, mul ? wave : 0, lane);
    ws.prime();
    __builtin_amdgcn_sched_barrier(0);
    f32x16 acc[2][S::NTW];
    acc_zero<2, S::NTW>(acc);
    if (mul) {
        panel_mma<S::KS, S::NTW, 2, NW>(s_img, ws, acc, lane);
        acc_to_staging<S::NTW, NW>(s_stg, acc[0], wave, lane);
        acc_to_staging<S::NTW, NW>(s_stg2, acc[1], wave, lane);
    }
    __syncthreads();
    RowTile<C, NW> a, b, bv, gv, bev, y;
    rt_load<C, NW>(a, s_stg, PN_STG_LD, P.lrow, P.c4);
    rt_load<C, NW>(b, s_stg2, PN_STG_LD, P.lrow, P.c4);
    rt_load_vec<C, NW>(bv, b1a, P.c4);
    rt_load_vec<C, NW>(gv, g1, P.c4);
    rt_load_vec<C, NW>(bev, be1, P.c4);
    if (P.live) {
        rt_store<C, NW>(a, h1, C, P.row, P.c4);
        rt_store<C, NW>(b, pa, C, P.row, P.c4);
    }
    rt_ln_fwd<C, NW>(a, bv, gv, bev, p.eps, y);
    if (P.live) rt_store<C, NW>(y, h1n, C, P.row, P.c4);
}

template <int C, int NW>
PN_KERNEL(NW) k_conv_f1(const ConvPanelArgs p) {
    using S = PnShape<C, NW>;
    __shared__ uint4 s_img[3 * S::KS * 64];
    __shared__ float s_stg[PN_ROWS * PN_STG_LD];
    __shared__ float s_stg2[PN_ROWS * PN_STG_LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const RtPos<NW> P((int)blockIdx.x * PN_ROWS, p.rows, wave, lane);
    const bool mul = wave < S::NT;
    RowTile<C, NW> x;
    rt_load<C, NW>(x, p.in0, p.ld0, P.rowc, P.c4);
    rt_a_put<C, NW, S::KS>(x, s_img, P.lrow, P.c);
    __syncthreads();
    stage_f1<C, NW>(p, s_img, s_stg, s_stg2, p.w0, p.w1, p.b0, p.g0, p.be0, p.out0, p.out1, p.out2, P, wave, lane, mul);
}

// ---- F2: hbar[e] = mean over the hyperedge's nodes of h1n, qb = hbar w12^T + b12 ---------------------------------------------
// in0 = h1n [N, C], rowptr / col = the incidence CSR by hyperedge, w0 = w12 image, bias_out = b12; out0 = hbar, out1 = qb
template <int C, int NW>
PN_KERNEL(NW) k_conv_f2(const ConvPanelArgs p) {
    using S = PnShape<C, NW>;
    __shared__ uint4 s_img[3 * S::KS * 64];
    __shared__ float s_stg[PN_ROWS * PN_STG_LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const RtPos<NW> P((int)blockIdx.x * PN_ROWS, p.rows, wave, lane);
    const bool mul = wave < S::NT;
    PN_STAMP(0);
    // (round 6) the post that releases the next batch's index build on the trainer's side stream, folded into this launch: one
    // launch slot (~4.7 us in a replayed graph) less than a one-thread kernel of its own
    if (p.signal && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_fetch_add(p.signal, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    WStream<S::KS, S::NTW, 1, NW> ws;
    ws.init(0, p.w0, mul ? wave : 0, lane);
    ws.prime();
    __builtin_amdgcn_sched_barrier(0);
    RowTile<C, NW> m;
    int deg;
    rt_gather_sum<C, NW, false>(p.in0, p.rowptr, p.col, nullptr, P, m, deg);
    const float den = deg > 1 ? (float)deg : 1.0f;
#pragma unroll
    for (int j = 0; j < S::NJ; ++j) { m.v[j].x /= den; m.v[j].y /= den; m.v[j].z /= den; m.v[j].w /= den; }
    if (P.live) rt_store<C, NW>(m, p.out0, C, P.row, P.c4);
    rt_a_put<C, NW, S::KS>(m, s_img, P.lrow, P.c);        // (rows past the end of the matrix: zeros for the MFMA)
    PN_STAMP(1);
    __syncthreads();
    PN_STAMP(2);
    f32x16 acc[1][S::NTW];
    acc_zero<1, S::NTW>(acc);
    if (mul) {
        panel_mma<S::KS, S::NTW, 1, NW>(s_img, ws, acc, lane);
        PN_STAMP(3);
        acc_to_staging<S::NTW, NW>(s_stg, acc[0], wave, lane);
    }
    __syncthreads();
    PN_STAMP(4);
    RowTile<C, NW> a, bv;
    rt_load<C, NW>(a, s_stg, PN_STG_LD, P.lrow, P.c4);
    rt_load_vec<C, NW>(bv, p.bias_out, P.c4);
#pragma unroll
    for (int j = 0; j < S::NJ; ++j) f4_add(a.v[j], bv.v[j]);
    if (P.live) rt_store<C, NW>(a, p.out1, C, P.row, P.c4);
    PN_STAMP(5);
}

// ---- F3: s -> u = scale * (s w23^T) + cw, x3 = LN3(relu(u + b3a)), Xn = act(x3 W3b^T + b3b)  [tail: F1 on Xn] ----------------
// in0 = s (read), or with rowptr: in0 = pa, in2 = qb, rowptr / col = the incidence CSR by node, g_inc / be_inc / eps_inc = the
// incidence LayerNorm, out6 = s (written); in1 = cw, w0 = w23 image, b0/g0/be0 = b3a, gamma3, beta3, w1 = W3b image,
// bias_out = b3b, relu; out0 = u, out1 = x3, out2 = Xn;
// tail: w2 = W1a image, w3 = W2v image, b1/g1/be1 = b1a, gamma1, beta1, out3 = h1, out4 = h1n, out5 = pa
template <int C, int NW>
PN_KERNEL(NW) k_conv_f3(const ConvPanelArgs p) {
    using S = PnShape<C, NW>;
    __shared__ uint4 s_img[3 * S::KS * 64];
    __shared__ float s_stg[PN_ROWS * PN_STG_LD];
    __shared__ float s_stg2[PN_ROWS * PN_STG_LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const RtPos<NW> P((int)blockIdx.x * PN_ROWS, p.rows, wave, lane);
    const bool mul = wave < S::NT;
    PN_STAMP(0);
    WStream<S::KS, S::NTW, 1, NW> ws;
    ws.init(0, p.w0, mul ? wave : 0, lane);
    ws.prime();
    RowTile<C, NW> t, cw;
    rt_load<C, NW>(cw, p.in1, C, P.rowc, P.c4);
    __builtin_amdgcn_sched_barrier(0);
    if (p.rowptr) {
        rt_incidence_mean<C, NW>(p.in0, p.in2, p.rowptr, p.col, p.g_inc, p.be_inc, p.eps_inc, P, t);
        if (P.live) rt_store<C, NW>(t, p.out6, C, P.row, P.c4);
    } else {
        rt_load<C, NW>(t, p.in0, C, P.rowc, P.c4);
    }
    rt_a_put<C, NW, S::KS>(t, s_img, P.lrow, P.c);
    PN_STAMP(1);
    __syncthreads();
    PN_STAMP(2);
    f32x16 acc[1][S::NTW];
    acc_zero<1, S::NTW>(acc);
    if (mul) {
        panel_mma<S::KS, S::NTW, 1, NW>(s_img, ws, acc, lane);
        PN_STAMP(3);
        acc_to_staging<S::NTW, NW>(s_stg, acc[0], wave, lane);
    }
    // the next product's weight stream starts now: its first fragments arrive while the rows are normalised
    ws.init(0, p.w1, mul ? wave : 0, lane);
    ws.prime();
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    PN_STAMP(4);
    {
        RowTile<C, NW> bv, gv, bev, x3;
        rt_load<C, NW>(t, s_stg, PN_STG_LD, P.lrow, P.c4);
        rt_load_vec<C, NW>(bv, p.b0, P.c4);
        rt_load_vec<C, NW>(gv, p.g0, P.c4);
        rt_load_vec<C, NW>(bev, p.be0, P.c4);
#pragma unroll
        for (int j = 0; j < S::NJ; ++j)
            t.v[j] = make_float4(fmaf(p.scale, t.v[j].x, cw.v[j].x), fmaf(p.scale, t.v[j].y, cw.v[j].y), fmaf(p.scale, t.v[j].z, cw.v[j].z),
                                 fmaf(p.scale, t.v[j].w, cw.v[j].w));
        rt_ln_fwd<C, NW>(t, bv, gv, bev, p.eps, x3);
        if (P.live) {
            rt_store<C, NW>(t, p.out0, C, P.row, P.c4);
            rt_store<C, NW>(x3, p.out1, C, P.row, P.c4);
        }
        rt_a_put<C, NW, S::KS>(x3, s_img, P.lrow, P.c);       // (every wavefront has left the MFMA loop: barrier above)
    }
    PN_STAMP(5);
    __syncthreads();
    PN_STAMP(6);
    acc_zero<1, S::NTW>(acc);
    if (mul) {
        panel_mma<S::KS, S::NTW, 1, NW>(s_img, ws, acc, lane);
        acc_to_staging<S::NTW, NW>(s_stg, acc[0], wave, lane);
    }
    PN_STAMP(7);
    __syncthreads();
    {
        RowTile<C, NW> bv;
        rt_load<C, NW>(t, s_stg, PN_STG_LD, P.lrow, P.c4);
        rt_load_vec<C, NW>(bv, p.bias_out, P.c4);
#pragma unroll
        for (int j = 0; j < S::NJ; ++j) {
            f4_add(t.v[j], bv.v[j]);
            if (p.relu) { t.v[j].x = fmaxf(t.v[j].x, 0.f); t.v[j].y = fmaxf(t.v[j].y, 0.f); t.v[j].z = fmaxf(t.v[j].z, 0.f); t.v[j].w = fmaxf(t.v[j].w, 0.f); }
        }
        if (P.live) rt_store<C, NW>(t, p.out2, C, P.row, P.c4);
        if (p.tail) rt_a_put<C, NW, S::KS>(t, s_img, P.lrow, P.c);
    }
    PN_STAMP(8);
    if (!p.tail) return;
    __syncthreads();
    stage_f1<C, NW>(p, s_img, s_stg, s_stg2, p.w2, p.w3, p.b1, p.g1, p.be1, p.out3, p.out4, p.out5, P, wave, lane, mul);
    PN_STAMP(9);
}

// ---- B3: dXn -> g = dXn * [Xn > 0], dx3 = g W3b, dpre = LN3bwd(u + b3a; dx3), ds = scale * dpre w23 --------------------------
// Shared by k_conv_b3 (rows from memory) and the tail of k_conv_b1 (rows = the dX it has just formed, in registers).
// w_a = W3b image (dy W), w_b = w23 image (dy W); slab = [d b3a | d gamma3 | d beta3] of this workgroup; acc_out += dpre
template <int C, int NW>
__device__ __forceinline__ void stage_b3(const ConvPanelArgs& p, uint4* __restrict__ s_img, float* __restrict__ s_stg, RowTile<C, NW>& t,
                                         const float* xmask, const uint4* w_a, const uint4* w_b, const float* u_pre, const float* b3a,
                                         const float* g3, float* g_out, float* dpre_out, float* ds_out, float* slab, float* acc_out,
                                         int acc_first, const RtPos<NW>& P, int wave, int lane, bool mul) {
    using S = PnShape<C, NW>;
    RowTile<C, NW> upre;
    WStream<S::KS, S::NTW, 1, NW> ws;
    ws.init(0, w_a, mul ? wave : 0, lane);
    ws.prime();
    rt_load<C, NW>(upre, u_pre, C, P.rowc, P.c4);
    if (xmask) {
        RowTile<C, NW> m;
        rt_load<C, NW>(m, xmask, C, P.rowc, P.c4);
#pragma unroll
        for (int j = 0; j < S::NJ; ++j) {
            t.v[j].x = m.v[j].x > 0.f ? t.v[j].x : 0.f; t.v[j].y = m.v[j].y > 0.f ? t.v[j].y : 0.f;
            t.v[j].z = m.v[j].z > 0.f ? t.v[j].z : 0.f; t.v[j].w = m.v[j].w > 0.f ? t.v[j].w : 0.f;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (g_out && P.live) rt_store<C, NW>(t, g_out, C, P.row, P.c4);
    // (the A image was last read before the barriers of the caller's slab reduction)
    rt_a_put<C, NW, S::KS>(t, s_img, P.lrow, P.c);
    PN_STAMP(8);
    __syncthreads();
    f32x16 acc[1][S::NTW];
    acc_zero<1, S::NTW>(acc);
    if (mul) {
        panel_mma<S::KS, S::NTW, 1, NW>(s_img, ws, acc, lane);
        acc_to_staging<S::NTW, NW>(s_stg, acc[0], wave, lane);
    }
    PN_STAMP(9);
    ws.init(0, w_b, mul ? wave : 0, lane);
    ws.prime();
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    PN_STAMP(10);
    RowTile<C, NW> a_db, a_dg, a_dbeta;
    rt_zero<C, NW>(a_db); rt_zero<C, NW>(a_dg); rt_zero<C, NW>(a_dbeta);
    {
        RowTile<C, NW> bv, gv, dpre;
        rt_load<C, NW>(t, s_stg, PN_STG_LD, P.lrow, P.c4);
        rt_load_vec<C, NW>(bv, b3a, P.c4);
        rt_load_vec<C, NW>(gv, g3, P.c4);
        rt_ln_bwd<C, NW>(upre, bv, gv, t, p.eps, P.live, dpre, a_db, a_dg, a_dbeta);
        if (P.live) {
            rt_store<C, NW>(dpre, dpre_out, C, P.row, P.c4);
            if (acc_out) {
                if (!acc_first) {
                    RowTile<C, NW> o;
                    rt_load<C, NW>(o, acc_out, C, P.row, P.c4);
#pragma unroll
                    for (int j = 0; j < S::NJ; ++j) f4_add(o.v[j], dpre.v[j]);
                    rt_store<C, NW>(o, acc_out, C, P.row, P.c4);
                } else {
                    rt_store<C, NW>(dpre, acc_out, C, P.row, P.c4);
                }
            }
        }
        rt_a_put<C, NW, S::KS>(dpre, s_img, P.lrow, P.c);
    }
    PN_STAMP(11);
    __syncthreads();
    acc_zero<1, S::NTW>(acc);
    if (mul) {
        panel_mma<S::KS, S::NTW, 1, NW>(s_img, ws, acc, lane);
        acc_to_staging<S::NTW, NW>(s_stg, acc[0], wave, lane);
    }
    PN_STAMP(12);
    __syncthreads();
    rt_load<C, NW>(t, s_stg, PN_STG_LD, P.lrow, P.c4);
#pragma unroll
    for (int j = 0; j < S::NJ; ++j) { t.v[j].x *= p.scale; t.v[j].y *= p.scale; t.v[j].z *= p.scale; t.v[j].w *= p.scale; }
    if (P.live) rt_store<C, NW>(t, ds_out, C, P.row, P.c4);
    PN_STAMP(13);
    write_slab<C, NW>(s_stg, slab + (int64_t)blockIdx.x * 3 * C, a_db, a_dg, a_dbeta, wave, lane);
}

// in0 = dXn (ld0), in1 = Xn or null, w0 = W3b image, w1 = w23 image, in2 = u, b0/g0 = b3a, gamma3;
// out0 = g (when in1), out1 = dpre, out2 = ds, slab, acc_out (+= dpre; overwritten when acc_first)
template <int C, int NW>
PN_KERNEL(NW) k_conv_b3(const ConvPanelArgs p) {
    using S = PnShape<C, NW>;
    __shared__ uint4 s_img[3 * S::KS * 64];
    __shared__ float s_stg[PN_ROWS * PN_STG_LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const RtPos<NW> P((int)blockIdx.x * PN_ROWS, p.rows, wave, lane);
    const bool mul = wave < S::NT;
    RowTile<C, NW> t;
    rt_load<C, NW>(t, p.in0, p.ld0, P.rowc, P.c4);
    stage_b3<C, NW>(p, s_img, s_stg, t, p.in1, p.w0, p.w1, p.in2, p.b0, p.g0, p.out0, p.out1, p.out2, p.slab, p.acc_out, p.acc_first, P,
                    wave, lane, mul);
}

// ---- B1: dh1[v] = LN1bwd(h1[v] + b1a; sum_e dhbar[e] / deg e), dX = [dh1 | dpa] . [W1a ; W2v]  [tail: B3 of the application before]
// in0 = dhbar [M, C] (or, with w3 = w12 image (dy W), dqb: B2 folded in), rowptr / col / wq = incidence CSR by node + entry
// weights, in1 = h1, b0/g0 = b1a, gamma1, in2 = dpa, w0 = stacked image [W1a ; W2v] (dy W, K = 2 C); out0 = dh1, out1 = dX,
// slab = [d b1a | d gamma1 | d beta1];
// tail: in3 = X of this application = Xn of the one before (mask), w1 = W3b image, w2 = w23 image, out5 = its u (read),
//       b1/g1 = b3a, gamma3; out2 = g, out3 = dpre, out4 = ds, slab2, acc_out
template <int C, int NW>
PN_KERNEL(NW) k_conv_b1(const ConvPanelArgs p) {
    using S = PnShape<C, NW>;
    constexpr int KS2 = 2 * S::KS;
    __shared__ uint4 s_img[3 * KS2 * 64];
    __shared__ float s_stg[PN_ROWS * PN_STG_LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const RtPos<NW> P((int)blockIdx.x * PN_ROWS, p.rows, wave, lane);
    const bool mul = wave < S::NT;
    // w3 != null: B2 folded in -- in0 is dqb and the gathered sums are multiplied by w12 here (the gathered mean is linear:
    // sum_e w_e (dqb[e] w12) = (sum_e w_e dqb[e]) w12), so dhbar never exists and its launch is gone
    PN_STAMP(0);
    WStream<S::KS, S::NTW, 1, NW> ws12;
    if (p.w3) {
        ws12.init(0, p.w3, mul ? wave : 0, lane);
        ws12.prime();
    }
    RowTile<C, NW> h, dpa, dsum;
    rt_load<C, NW>(h, p.in1, C, P.rowc, P.c4);
    rt_load<C, NW>(dpa, p.in2, C, P.rowc, P.c4);
    __builtin_amdgcn_sched_barrier(0);
    int deg;
    if (p.wq) rt_gather_sum<C, NW, true>(p.in0, p.rowptr, p.col, p.wq, P, dsum, deg);
    else rt_gather_sum<C, NW, false>(p.in0, p.rowptr, p.col, nullptr, P, dsum, deg);
    PN_STAMP(1);
    if (p.w3) {
        rt_a_put<C, NW, S::KS>(dsum, s_img, P.lrow, P.c);
        __syncthreads();
        f32x16 acc12[1][S::NTW];
        acc_zero<1, S::NTW>(acc12);
        if (mul) {
            panel_mma<S::KS, S::NTW, 1, NW>(s_img, ws12, acc12, lane);
            acc_to_staging<S::NTW, NW>(s_stg, acc12[0], wave, lane);
        }
        __syncthreads();
        rt_load<C, NW>(dsum, s_stg, PN_STG_LD, P.lrow, P.c4);
    }
    PN_STAMP(2);
    WStream<KS2, S::NTW, 1, NW> ws;
    ws.init(0, p.w0, mul ? wave : 0, lane);
    ws.prime();
    __builtin_amdgcn_sched_barrier(0);
    RowTile<C, NW> a_db, a_dg, a_dbeta;
    rt_zero<C, NW>(a_db); rt_zero<C, NW>(a_dg); rt_zero<C, NW>(a_dbeta);
    {
        RowTile<C, NW> bv, gv, dh;
        rt_load_vec<C, NW>(bv, p.b0, P.c4);
        rt_load_vec<C, NW>(gv, p.g0, P.c4);
        rt_ln_bwd<C, NW>(h, bv, gv, dsum, p.eps, P.live, dh, a_db, a_dg, a_dbeta);
        if (P.live) rt_store<C, NW>(dh, p.out0, C, P.row, P.c4);
        // (every wavefront is past the barrier behind the w12 product: its image may be overwritten, in the K = 2 C layout)
        rt_a_put<C, NW, KS2>(dh, s_img, P.lrow, P.c, 0);
        rt_a_put<C, NW, KS2>(dpa, s_img, P.lrow, P.c, C / 4);
    }
    PN_STAMP(3);
    __syncthreads();
    f32x16 acc[1][S::NTW];
    acc_zero<1, S::NTW>(acc);
    if (mul) {
        panel_mma<KS2, S::NTW, 1, NW>(s_img, ws, acc, lane);
        acc_to_staging<S::NTW, NW>(s_stg, acc[0], wave, lane);
    }
    PN_STAMP(4);
    __syncthreads();
    RowTile<C, NW> dx;
    rt_load<C, NW>(dx, s_stg, PN_STG_LD, P.lrow, P.c4);
    if (p.out1 && P.live) rt_store<C, NW>(dx, p.out1, C, P.row, P.c4);     // (with the tail only the masked gradient g is needed afterwards)
    write_slab<C, NW>(s_stg, p.slab + (int64_t)blockIdx.x * 3 * C, a_db, a_dg, a_dbeta, wave, lane);
    PN_STAMP(5);
    if (!p.tail) return;
    stage_b3<C, NW>(p, s_img, s_stg, dx, p.in3, p.w1, p.w2, p.out5, p.b1, p.g1, p.out2, p.out3, p.out4, p.slab2, p.acc_out, p.acc_first, P,
                    wave, lane, mul);
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
// out0 = node_in [N, C + 16], out1 = hpre [N, 2 C], out2 = hid [N, 2 C], out3 = out [N, C].
// g0 != null (round 6): in0 = feats and normed = LayerNorm(feats; g0, be0, eps) is formed HERE (node_norm, egnn_layer.py:192,360:
// it was a launch of its own each way), the residual is in0 itself (in2 unused)
template <int C, int NW>
PN_KERNEL(NW) k_node_f(const ConvPanelArgs p) {
    using S = PnShape<C, NW>;
    constexpr int KS1 = C / 16 + 1, KS2 = C / 8;
    __shared__ uint4 s_img[3 * KS2 * 64];
    __shared__ float s_stg[PN_ROWS * PN_STG_LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const RtPos<NW> P((int)blockIdx.x * PN_ROWS, p.rows, wave, lane);
    const bool mul = wave < S::NT;
    RowTile<C, NW> x, res;
    rt_load<C, NW>(x, p.in0, C, P.rowc, P.c4);
    const float4 mi = P.c < 4 ? *reinterpret_cast<const float4*>(p.in1 + (int64_t)P.rowc * 16 + P.c4) : f4_zero();
    WStream<KS1, S::NTW, 2, NW> ws;
    ws.init(0, p.w0, mul ? wave : 0, lane);
    ws.init(1, p.w1, mul ? wave : 0, lane);
    ws.prime();
    if (p.g0) {
        res = x;
        RowTile<C, NW> gv, bv, xh;
        rt_load_vec<C, NW>(gv, p.g0, P.c4);
        rt_load_vec<C, NW>(bv, p.be0, P.c4);
        float rstd;
        rt_xhat_plain<C, NW>(x, p.eps, xh, rstd);
#pragma unroll
        for (int j = 0; j < S::NJ; ++j) {
            x.v[j].x = fmaf(gv.v[j].x, xh.v[j].x, bv.v[j].x); x.v[j].y = fmaf(gv.v[j].y, xh.v[j].y, bv.v[j].y);
            x.v[j].z = fmaf(gv.v[j].z, xh.v[j].z, bv.v[j].z); x.v[j].w = fmaf(gv.v[j].w, xh.v[j].w, bv.v[j].w);
        }
    } else {
        rt_load<C, NW>(res, p.in2, C, P.rowc, P.c4);
    }
    __builtin_amdgcn_sched_barrier(0);
    rt_a_put<C, NW, KS1>(x, s_img, P.lrow, P.c);
    if (P.c < 4) a_put<KS1>(s_img, P.lrow, C / 4 + P.c, mi);
    if (P.live) {
        rt_store<C, NW>(x, p.out0, C + 16, P.row, P.c4);
        if (P.c < 4) *reinterpret_cast<float4*>(p.out0 + (int64_t)P.row * (C + 16) + C + P.c4) = mi;
    }
    __syncthreads();
    f32x16 acc[2][S::NTW];
    acc_zero<2, S::NTW>(acc);
    if (mul) {
        panel_mma<KS1, S::NTW, 2, NW>(s_img, ws, acc, lane);
        acc_to_staging<S::NTW, NW>(s_stg, acc[0], wave, lane);
    }
    WStream<KS2, S::NTW, 1, NW> ws2;
    ws2.init(0, p.w2, mul ? wave : 0, lane);
    ws2.prime();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __syncthreads();                       // (half 0: every wavefront is out of the MFMA loop -- the image may be rewritten)
        RowTile<C, NW> t, bv;
        rt_load<C, NW>(t, s_stg, PN_STG_LD, P.lrow, P.c4);
        rt_load_vec<C, NW>(bv, p.b0 + half * C, P.c4);
#pragma unroll
        for (int j = 0; j < S::NJ; ++j) f4_add(t.v[j], bv.v[j]);
        if (P.live) rt_store<C, NW>(t, p.out1 + half * C, 2 * C, P.row, P.c4);
#pragma unroll
        for (int j = 0; j < S::NJ; ++j)
            t.v[j] = make_float4(silu_f(t.v[j].x), silu_f(t.v[j].y), silu_f(t.v[j].z), silu_f(t.v[j].w));
        if (P.live) rt_store<C, NW>(t, p.out2 + half * C, 2 * C, P.row, P.c4);
        rt_a_put<C, NW, KS2>(t, s_img, P.lrow, P.c, half * (C / 4));
        if (half == 0) {
            __syncthreads();                   // the staging tile's rows have been read: second half of the product
            if (mul) acc_to_staging<S::NTW, NW>(s_stg, acc[1], wave, lane);
        }
    }
    __syncthreads();
    f32x16 acc2[1][S::NTW];
    acc_zero<1, S::NTW>(acc2);
    if (mul) {
        panel_mma<KS2, S::NTW, 1, NW>(s_img, ws2, acc2, lane);
        acc_to_staging<S::NTW, NW>(s_stg, acc2[0], wave, lane);
    }
    __syncthreads();
    RowTile<C, NW> t, bv;
    rt_load<C, NW>(t, s_stg, PN_STG_LD, P.lrow, P.c4);
    rt_load_vec<C, NW>(bv, p.bias_out, P.c4);
#pragma unroll
    for (int j = 0; j < S::NJ; ++j) { f4_add(t.v[j], bv.v[j]); f4_add(t.v[j], res.v[j]); }
    if (P.live) rt_store<C, NW>(t, p.out3, C, P.row, P.c4);
}

// in0 = dout [N, C] (ld0), in1 = hpre [N, 2 C]; w0 / w1 = W3 N image, output columns [0, C) / [C, 2 C) (K = C);
// w2 = W0 N image (K = 2 C, N = C + 16 zero-padded to C + 32); out0 = dpre [N, 2 C], out1 = dnode_in [N, C + 16].
// g0 != null (round 6, with k_node_f's LayerNorm): in3 = feats; out1 = d feats [N, C] = LNbwd(d normed) + dout (the residual's
// gradient rides along), out2 = d m_i [N, 16], slab = [unused | d gamma | d beta] partial sums of this workgroup
template <int C, int NW>
PN_KERNEL(NW) k_node_b(const ConvPanelArgs p) {
    using S = PnShape<C, NW>;
    constexpr int KS2 = C / 8, NT2 = C / 32 + 1, NTW2 = (NT2 + NW - 1) / NW, LD2 = C + 32 + 4;
    __shared__ uint4 s_img[3 * KS2 * 64];
    __shared__ float s_stg[PN_ROWS * LD2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const RtPos<NW> P((int)blockIdx.x * PN_ROWS, p.rows, wave, lane);
    const bool mul = wave < S::NT;
    RowTile<C, NW> d, hp[2];
    rt_load<C, NW>(d, p.in0, p.ld0, P.rowc, P.c4);
    WStream<S::KS, S::NTW, 2, NW> ws;
    ws.init(0, p.w0, mul ? wave : 0, lane);
    ws.init(1, p.w1, mul ? wave : 0, lane);
    ws.prime();
    rt_load<C, NW>(hp[0], p.in1, 2 * C, P.rowc, P.c4);
    rt_load<C, NW>(hp[1], p.in1 + C, 2 * C, P.rowc, P.c4);
    __builtin_amdgcn_sched_barrier(0);
    rt_a_put<C, NW, S::KS>(d, s_img, P.lrow, P.c);
    __syncthreads();
    f32x16 acc[2][S::NTW];
    acc_zero<2, S::NTW>(acc);
    if (mul) {
        panel_mma<S::KS, S::NTW, 2, NW>(s_img, ws, acc, lane);
        acc_to_staging<S::NTW, NW, LD2>(s_stg, acc[0], wave, lane);
    }
    const bool mul2 = wave < NT2;
    WStream<KS2, NTW2, 1, NW> ws2;
    ws2.init(0, p.w2, mul2 ? wave : 0, lane, NT2);
    ws2.prime();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __syncthreads();
        RowTile<C, NW> t;
        rt_load<C, NW>(t, s_stg, LD2, P.lrow, P.c4);
#pragma unroll
        for (int j = 0; j < S::NJ; ++j) {
            const float4 h = hp[half].v[j];
            t.v[j].x *= silu_grad_f(h.x); t.v[j].y *= silu_grad_f(h.y); t.v[j].z *= silu_grad_f(h.z); t.v[j].w *= silu_grad_f(h.w);
        }
        if (P.live) rt_store<C, NW>(t, p.out0 + half * C, 2 * C, P.row, P.c4);
        rt_a_put<C, NW, KS2>(t, s_img, P.lrow, P.c, half * (C / 4));
        if (half == 0) {
            __syncthreads();
            if (mul) acc_to_staging<S::NTW, NW, LD2>(s_stg, acc[1], wave, lane);
        }
    }
    __syncthreads();
    f32x16 acc2[1][NTW2];
    acc_zero<1, NTW2>(acc2);
    if (mul2) {
        panel_mma<KS2, NTW2, 1, NW>(s_img, ws2, acc2, lane);
        acc_to_staging<NTW2, NW, LD2>(s_stg, acc2[0], wave, lane, NT2);
    }
    __syncthreads();
    RowTile<C, NW> t;
    rt_load<C, NW>(t, s_stg, LD2, P.lrow, P.c4);
    if (p.g0) {
        if (P.live && P.c < 4)
            *reinterpret_cast<float4*>(p.out2 + (int64_t)P.row * 16 + P.c4) = *reinterpret_cast<const float4*>(s_stg + P.lrow * LD2 + C + P.c4);
        RowTile<C, NW> xr, gv, xh, dx, a_db, a_dg, a_dbeta;
        rt_load<C, NW>(xr, p.in3, C, P.rowc, P.c4);
        rt_load_vec<C, NW>(gv, p.g0, P.c4);
        rt_zero<C, NW>(a_db); rt_zero<C, NW>(a_dg); rt_zero<C, NW>(a_dbeta);
        float rstd;
        rt_xhat_plain<C, NW>(xr, p.eps, xh, rstd);
        rt_ln_plain_bwd<C, NW>(xh, rstd, gv, t, P.live, dx, a_dg, a_dbeta);
#pragma unroll
        for (int j = 0; j < S::NJ; ++j) f4_add(dx.v[j], d.v[j]);
        if (P.live) rt_store<C, NW>(dx, p.out1, C, P.row, P.c4);
        write_slab<C, NW>(s_stg, p.slab + (int64_t)blockIdx.x * 3 * C, a_db, a_dg, a_dbeta, wave, lane);
        return;
    }
    if (P.live) {
        rt_store<C, NW>(t, p.out1, C + 16, P.row, P.c4);
        if (P.c < 4)
            *reinterpret_cast<float4*>(p.out1 + (int64_t)P.row * (C + 16) + C + P.c4) =
                *reinterpret_cast<const float4*>(s_stg + P.lrow * LD2 + C + P.c4);
    }
}

inline bool pn_width_ok(int C) { return C == 64 || C == 128 || C == 256; }

// wavefronts per panel: 8 (default); EQH_PANEL_WAVES=4 selects round 4's geometry for same-box A/B runs
inline int pn_waves() {
    static const int nw = [] {
        const char* e = std::getenv("EQH_PANEL_WAVES");
        return (e && e[0] == '4') ? 4 : 8;
    }();
    return nw;
}

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
                          q->bias_out, q->out0, q->out1, q->out2, q->out3, q->out4, q->out5, q->slab, q->slab2, q->acc_out, q->wq,
                          q->g_inc, q->be_inc, q->out6};
    for (const void* x : ptrs)
        if (!eqh_aligned16(x)) return EQH_ERR_ALIGN;
    ConvPanelArgs a{};
    a.rows = (int)q->rows; a.eps = q->eps; a.scale = q->scale; a.eps_inc = q->eps_inc;
    a.relu = q->relu; a.acc_first = q->acc_first; a.tail = q->tail;
    a.in0 = q->in0; a.in1 = q->in1; a.in2 = q->in2; a.in3 = q->in3; a.ld0 = q->ld0 > 0 ? q->ld0 : C;
    a.rowptr = q->rowptr; a.col = q->col; a.wq = q->wq;
    a.w0 = static_cast<const uint4*>(q->w0); a.w1 = static_cast<const uint4*>(q->w1);
    a.w2 = static_cast<const uint4*>(q->w2); a.w3 = static_cast<const uint4*>(q->w3);
    a.b0 = q->b0; a.g0 = q->g0; a.be0 = q->be0; a.b1 = q->b1; a.g1 = q->g1; a.be1 = q->be1; a.bias_out = q->bias_out;
    a.g_inc = q->g_inc; a.be_inc = q->be_inc;
    a.out0 = q->out0; a.out1 = q->out1; a.out2 = q->out2; a.out3 = q->out3; a.out4 = q->out4; a.out5 = q->out5; a.out6 = q->out6;
    a.slab = q->slab; a.slab2 = q->slab2; a.acc_out = q->acc_out;
    a.signal = stage == HG_CONV_F2 ? q->signal : nullptr;
    if (a.ld0 & 3) return EQH_ERR_ALIGN;
    const int blocks = (int)((q->rows + PN_ROWS - 1) / PN_ROWS);
    const int nw = pn_waves();
    const dim3 grid(blocks), block(64 * nw);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    auto need = [](std::initializer_list<const void*> l) { for (const void* x : l) if (!x) return false; return true; };
    auto f1_ok = [&](const void* wa, const void* wb, const void* b, const void* g, const void* be, const void* o0, const void* o1, const void* o2) {
        return need({wa, wb, b, g, be, o0, o1, o2});
    };
#define PN_LAUNCH_W(K, NW_)                                                                               \
    do {                                                                                                  \
        if (C == 256) hipLaunchKernelGGL((K<256, NW_>), grid, block, 0, stream, a);                       \
        else if (C == 128) hipLaunchKernelGGL((K<128, NW_>), grid, block, 0, stream, a);                  \
        else hipLaunchKernelGGL((K<64, NW_>), grid, block, 0, stream, a);                                 \
    } while (0)
#define PN_LAUNCH(K)                                                                                      \
    do {                                                                                                  \
        if (nw == 8) PN_LAUNCH_W(K, 8);                                                                   \
        else PN_LAUNCH_W(K, 4);                                                                           \
        EQH_CHECK_LAUNCH();                                                                               \
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
            if (q->rowptr && !need({q->in2, q->col, q->g_inc, q->be_inc, q->out6})) return EQH_ERR_ARG;
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
            if (q->tail && !need({q->w1, q->w2, q->out5, q->b1, q->g1, q->out3, q->out4, q->slab2, q->dbias2,
                                  q->dgamma2, q->dbeta2}))
                return EQH_ERR_ARG;
            if (q->tail && q->in3 && !q->out2) return EQH_ERR_ARG;
            PN_LAUNCH(k_conv_b1);
            int rc = eqh_reduce_slabs3_async(q->slab, blocks, 3 * (int64_t)C, q->dbias, q->dgamma, q->dbeta, C, C, q->accumulate, stream);
            if (rc || !q->tail) return rc;
            return eqh_reduce_slabs3_async(q->slab2, blocks, 3 * (int64_t)C, q->dbias2, q->dgamma2, q->dbeta2, C, C, q->accumulate, stream);
        }
        case HG_EGNN_NODE_F:
            if (!need({q->in0, q->in1, q->w0, q->w1, q->w2, q->b0, q->bias_out, q->out0, q->out1, q->out2, q->out3})) return EQH_ERR_ARG;
            if (q->g0 ? !q->be0 : !q->in2) return EQH_ERR_ARG;      // LayerNorm of in0 inside (g0, be0) -- or normed rows + the residual in2
            PN_LAUNCH(k_node_f);
            return EQH_OK;
        case HG_EGNN_NODE_B:
            if (!need({q->in0, q->in1, q->w0, q->w1, q->w2, q->out0, q->out1})) return EQH_ERR_ARG;
            if (q->g0 && !need({q->in3, q->out2, q->slab, q->dbias, q->dgamma, q->dbeta})) return EQH_ERR_ARG;
            PN_LAUNCH(k_node_b);
            if (!q->g0) return EQH_OK;
            // (dbias: a [C] scratch row -- a plain LayerNorm has no bias in front of it; the slab's first third is zeros)
            return eqh_reduce_slabs3_async(q->slab, blocks, 3 * (int64_t)C, q->dbias, q->dgamma, q->dbeta, C, C, q->accumulate, stream);
        default:
            return EQH_ERR_ARG;
    }
#undef PN_LAUNCH
#undef PN_LAUNCH_W
}

// ---- a SUM of products over different row blocks of the same rows: out = sum_g A_g W_g + D, g < NG <= 3 -------------------------
// (MHNNConv's input gradients, autograd of conv.py:87-101 with every first Linear split by input block: dX = dpre_v W4a_X +
// dpa3 W3a_x + dpa1 W1a_x, dE = dpre_e W2a_E + dqb1 W1a_e -- three / two single-product launches chained through their addend
// until round 6; at 19-148 panels a launch is its ~7 us slot, not its work.)  Two A images take turns: image g + 1 is built while
// product g runs; the weight stream runs on across the products.
struct PanelSum {
    const float* A[3];
    int64_t lda[3];
    int rows;
    const uint4* W[3];
    const float* D;
    int64_t ldd;
    float* out;
    int64_t ldo;
};

template <int C, int NW, int NG>
PN_KERNEL(NW) k_panel_sum(const PanelSum p) {
    using S = PnShape<C, NW>;
    __shared__ uint4 s_img[2][3 * S::KS * 64];
    __shared__ float s_stg[PN_ROWS * PN_STG_LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const RtPos<NW> P((int)blockIdx.x * PN_ROWS, p.rows, wave, lane);
    const bool mul = wave < S::NT;
    RowTile<C, NW> x;
    rt_load<C, NW>(x, p.A[0], p.lda[0], P.rowc, P.c4);
    WStream<S::KS, S::NTW, NG, NW> ws;
#pragma unroll
    for (int g = 0; g < NG; ++g) ws.init(g, p.W[g], mul ? wave : 0, lane);
    ws.prime();
    __builtin_amdgcn_sched_barrier(0);
    rt_a_put<C, NW, S::KS>(x, s_img[0], P.lrow, P.c);
    if (NG > 1) rt_load<C, NW>(x, p.A[1], p.lda[1], P.rowc, P.c4);        // (in flight during product 0)
    __syncthreads();
    f32x16 acc[1][S::NTW];
    acc_zero<1, S::NTW>(acc);
    if (mul) panel_mma_part<S::KS, S::NTW, NG, NW, 0>(s_img[0], ws, acc, lane);
    if (NG > 1) {
        rt_a_put<C, NW, S::KS>(x, s_img[1], P.lrow, P.c);                  // (nobody reads image 1 yet)
        if (NG > 2) rt_load<C, NW>(x, p.A[2], p.lda[2], P.rowc, P.c4);
        __syncthreads();                                                   // image 1 complete; image 0 free
        if (mul) panel_mma_part<S::KS, S::NTW, NG, NW, (NG > 1 ? 1 : 0)>(s_img[1], ws, acc, lane);
    }
    if (NG > 2) {
        rt_a_put<C, NW, S::KS>(x, s_img[0], P.lrow, P.c);
        __syncthreads();
        if (mul) panel_mma_part<S::KS, S::NTW, NG, NW, (NG > 2 ? 2 : 0)>(s_img[0], ws, acc, lane);
    }
    if (mul) acc_to_staging<S::NTW, NW>(s_stg, acc[0], wave, lane);
    __syncthreads();
    RowTile<C, NW> a, d;
    rt_load<C, NW>(a, s_stg, PN_STG_LD, P.lrow, P.c4);
    if (p.D) {
        rt_load<C, NW>(d, p.D, p.ldd, P.rowc, P.c4);
#pragma unroll
        for (int j = 0; j < S::NJ; ++j) {
            a.v[j].x += d.v[j].x; a.v[j].y += d.v[j].y; a.v[j].z += d.v[j].z; a.v[j].w += d.v[j].w;
        }
    }
    if (P.live) rt_store<C, NW>(a, p.out, p.ldo, P.row, P.c4);
}

extern "C" int hg_panel_sum(const HgPanelSum* q, void* stream_) {
    if (!q || q->rows < 0 || q->n < 1 || q->n > 3 || !q->out) return EQH_ERR_ARG;
    const int C = q->C;
    if (!pn_width_ok(C)) return EQH_ERR_ARG;
    if (q->rows >= ((int64_t)1 << 31) - 64) return EQH_ERR_RANGE;
    if (!eqh_aligned16(q->out) || (q->ldo & 3) || !eqh_aligned16(q->d) || (q->d && (q->ldd & 3))) return EQH_ERR_ALIGN;
    PanelSum p{};
    p.rows = (int)q->rows; p.D = q->d; p.ldd = q->ldd; p.out = q->out; p.ldo = q->ldo;
    for (int g = 0; g < q->n; ++g) {
        if (!q->a[g] || !q->w[g]) return EQH_ERR_ARG;
        if (!eqh_aligned16(q->a[g]) || (q->lda[g] & 3) || !eqh_aligned16(q->w[g])) return EQH_ERR_ALIGN;
        p.A[g] = q->a[g]; p.lda[g] = q->lda[g]; p.W[g] = static_cast<const uint4*>(q->w[g]);
    }
    if (q->rows == 0) return EQH_OK;
    const int nw = pn_waves();
    const dim3 grid((unsigned)((q->rows + PN_ROWS - 1) / PN_ROWS)), block(64 * nw);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
#define PN_SUM_C(NW_, NG_)                                                                                     \
    do {                                                                                                       \
        if (C == 256) hipLaunchKernelGGL((k_panel_sum<256, NW_, NG_>), grid, block, 0, stream, p);             \
        else if (C == 128) hipLaunchKernelGGL((k_panel_sum<128, NW_, NG_>), grid, block, 0, stream, p);        \
        else hipLaunchKernelGGL((k_panel_sum<64, NW_, NG_>), grid, block, 0, stream, p);                       \
    } while (0)
#define PN_SUM(NG_)                                                                                            \
    do {                                                                                                       \
        if (nw == 8) PN_SUM_C(8, NG_);                                                                         \
        else PN_SUM_C(4, NG_);                                                                                 \
    } while (0)
    if (q->n == 1) PN_SUM(1);
    else if (q->n == 2) PN_SUM(2);
    else PN_SUM(3);
#undef PN_SUM
#undef PN_SUM_C
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}

extern "C" int hg_panel_multi(const HgPanelMulti* q, void* stream_) {
    if (!q || q->rows < 0 || !q->a || q->n < 1 || q->n > 3) return EQH_ERR_ARG;
    const int C = q->C;
    if (!pn_width_ok(C)) return EQH_ERR_ARG;
    if (q->rows >= ((int64_t)1 << 31) - 64) return EQH_ERR_RANGE;
    if ((q->lda & 3) || !eqh_aligned16(q->a)) return EQH_ERR_ALIGN;
    PanelMulti p{};
    p.A = q->a; p.lda = q->lda; p.rows = (int)q->rows;
    for (int g = 0; g < q->n; ++g) {
        if (!q->w[g] || !q->out[g]) return EQH_ERR_ARG;
        if (!eqh_aligned16(q->w[g]) || !eqh_aligned16(q->out[g]) || !eqh_aligned16(q->bias[g]) || !eqh_aligned16(q->d[g]) ||
            (q->ldo[g] & 3) || (q->d[g] && (q->ldd[g] & 3)))
            return EQH_ERR_ALIGN;
        p.W[g] = static_cast<const uint4*>(q->w[g]); p.bias[g] = q->bias[g]; p.rw[g] = q->rw[g]; p.D[g] = q->d[g];
        p.ldd[g] = q->ldd[g]; p.out[g] = q->out[g]; p.ldo[g] = q->ldo[g];
    }
    if (q->rows == 0) return EQH_OK;
    const int nw = pn_waves();
    const dim3 grid((unsigned)((q->rows + PN_ROWS - 1) / PN_ROWS)), block(64 * nw);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
#define PN_MULTI_C(NW_, NG_)                                                                                   \
    do {                                                                                                       \
        if (C == 256) hipLaunchKernelGGL((k_panel_multi<256, NW_, NG_>), grid, block, 0, stream, p);           \
        else if (C == 128) hipLaunchKernelGGL((k_panel_multi<128, NW_, NG_>), grid, block, 0, stream, p);      \
        else hipLaunchKernelGGL((k_panel_multi<64, NW_, NG_>), grid, block, 0, stream, p);                     \
    } while (0)
#define PN_MULTI(NG_)                                                                                          \
    do {                                                                                                       \
        if (nw == 8) PN_MULTI_C(8, NG_);                                                                       \
        else PN_MULTI_C(4, NG_);                                                                               \
    } while (0)
    if (q->n == 1) PN_MULTI(1);
    else if (q->n == 2) PN_MULTI(2);
    else PN_MULTI(3);
#undef PN_MULTI
#undef PN_MULTI_C
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}

extern "C" int32_t hg_panel_waves(void) { return pn_waves(); }

#ifdef PN_STAMPS
extern "C" int hg_panel_debug_stamps(void* buf) {
    unsigned long long* q = static_cast<unsigned long long*>(buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(pn_stamp_buf), &q, sizeof(q)) == hipSuccess ? EQH_OK : EQH_ERR_ARG;
}
extern "C" int hg_panel_debug_flags(int flags) {
    return hipMemcpyToSymbol(HIP_SYMBOL(pn_debug_flags), &flags, sizeof(flags)) == hipSuccess ? EQH_OK : EQH_ERR_ARG;
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
            b.it[i] = PackItem{q.w, q.ld, static_cast<uint4*>(q.dst), q.K, q.N, q.trans ? 1 : 0, q.kstep0, total, n_valid, q.k_major ? 1 : 0};
            b.first[i + 1] = b.first[i] + (q.K / 16) * (q.N / 32);
        }
        for (int i = b.n; i < PN_MAXPACK; ++i) { b.it[i] = b.it[0]; b.first[i + 1] = b.first[b.n]; }
        const int units = b.first[b.n];
        hipLaunchKernelGGL(k_panel_pack, dim3((units + 3) / 4), dim3(256), 0, stream, b);
        EQH_CHECK_LAUNCH();
    }
    return EQH_OK;
}

extern "C" int hg_panel_stream_supported(int32_t K, int32_t N) {
    return (K == 64 || K == 128 || K == 256) && (N == 128 || N == 256);
}

extern "C" int hg_panel_stream_gemm_f32(const float* a, int64_t lda, int64_t rows, int32_t K, int32_t N, const void* wpack, float alpha,
                                        const float* d, int64_t ldd, float beta, const float* bias, int32_t relu, float* c, int64_t ldc,
                                        void* stream_) {
    if (rows < 0 || !a || !wpack || !c) return EQH_ERR_ARG;
    if (!hg_panel_stream_supported(K, N)) return EQH_ERR_ARG;
    if ((lda & 3) || (ldc & 3) || (d && (ldd & 3)) || !eqh_aligned16(a) || !eqh_aligned16(c) || !eqh_aligned16(d) ||
        !eqh_aligned16(bias) || !eqh_aligned16(wpack))
        return EQH_ERR_ALIGN;
    if (rows >= ((int64_t)1 << 31) - 64) return EQH_ERR_RANGE;
    if (rows == 0) return EQH_OK;
    PanelPlain p{a, lda, (int)rows, static_cast<const uint4*>(wpack), alpha, beta, d, ldd, bias, relu, c, ldc};
    const int n_panels = (int)((rows + PN_ROWS - 1) / PN_ROWS);
    static const int n_cu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        return n;
    }();
    const dim3 grid((unsigned)(n_panels < n_cu ? n_panels : n_cu)), block(1024);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
#define PN_STREAM(K_, N_) if (K == K_ && N == N_) hipLaunchKernelGGL((k_panel_stream<K_, N_>), grid, block, 0, stream, p, n_panels)
    PN_STREAM(256, 256); else PN_STREAM(128, 256); else PN_STREAM(64, 256);
    else PN_STREAM(256, 128); else PN_STREAM(128, 128); else PN_STREAM(64, 128);
#undef PN_STREAM
    EQH_CHECK_LAUNCH();
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
    const int nw = pn_waves();
    const dim3 grid((unsigned)((rows + PN_ROWS - 1) / PN_ROWS)), block(64 * nw);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    static const bool pair = [] { const char* e = getenv("EQH_PANEL_PAIR"); return e && e[0] == '1'; }();
    if (pair && C == 256) {      // measurement aid: see k_panel_plain_pair
        hipLaunchKernelGGL((k_panel_plain_pair<256>), dim3(grid.x, 2), dim3(256), 0, stream, p);
        EQH_CHECK_LAUNCH();
        return EQH_OK;
    }
#define PN_PLAIN(NW_)                                                                                       \
    do {                                                                                                    \
        if (C == 256) hipLaunchKernelGGL((k_panel_plain<256, NW_>), grid, block, 0, stream, p);            \
        else if (C == 128) hipLaunchKernelGGL((k_panel_plain<128, NW_>), grid, block, 0, stream, p);       \
        else hipLaunchKernelGGL((k_panel_plain<64, NW_>), grid, block, 0, stream, p);                      \
    } while (0)
    if (nw == 8) PN_PLAIN(8);
    else PN_PLAIN(4);
#undef PN_PLAIN
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}
