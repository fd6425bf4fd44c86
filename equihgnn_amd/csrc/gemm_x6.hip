// fp32 GEMM on the bf16 matrix cores with fp32-grade results ("x6": six bf16 products per fp32 product):
//
//     C[M,N] = act( alpha * op(A)[M,K] . op(B)[K,N]  (+ beta * D[M,N])  (+ bias[N]) )
//
// Stands where the reference has nn.Linear / F.linear and autograd's input-gradient products (mlp.py:91-99,
// conv.py:169-182, egnn_layer.py:180-208,298-310,360-362, equiformer_layer.py:376-383, fa_former_layer.py:241-289): the
// "dense per-type linear mixes" of the north star.
//
// Why.  gfx950 has no TF32-like mode; its fp32-input MFMA (v_mfma_f32_16x16x4_f32) runs at the fp32 VALU rate,
// 64 FLOP/clk/SIMD = 157 TFLOP/s, and the tuned library reaches 100-129 TFLOP/s of it on this model's shapes.  The
// bf16 MFMA (v_mfma_f32_16x16x32_bf16) is 16x faster.  An fp32 number splits EXACTLY into three bf16 numbers,
//     a = a0 + a1 + a2,   a0 = top 16 bits of a (truncated: 8 significand bits),  a1 = top 16 bits of (a - a0),
//                         a2 = a - a0 - a1  (at most 8 significant bits are left: exactly a bf16 number),
// so a.b = sum_ij ai.bj, and the six products with i + j <= 2 (a0b0; a0b1, a1b0; a0b2, a2b0, a1b1) carry everything
// down to 2^-16 of |a||b|; the three dropped ones are below 2^-22 |a||b| (typically 2^-24: one fp32 rounding of the
// product).  Every bf16 x bf16 product is exact in fp32 and the MFMA accumulates in fp32, so the result has the error
// of an fp32 dot product (tests/test_hip_kernels.py compares it with the fp32-MFMA library GEMM against float64: the
// same error or less) at 16 / 6 = 2.7x the fp32 matrix rate.  No fp16 variant: its exponent range would need scaling.
//
// Structure.  256 threads = 4 wavefronts as 2 x 2; block tile 64 x 64 (wave tile 32 x 32) or 128 x 128 (64 x 64);
// K walked 32 at a time (one MFMA K).  Per K step the A and B tiles go global (fp32, coalesced 128-B row segments)
// -> registers (issued before the MFMAs of the previous step) -> split into the three bf16 planes (5.5 VALU
// operations per element: and, sub, and, sub and three byte permutes per element pair) -> LDS in MFMA OPERAND ORDER:
// the 16 rows x 32 k of one operand fragment are 1 KiB, lane (q = k / 8, r = row) owns 16 bytes at slot
// q * 16 + (r ^ 2q) -- the xor keeps both the 8-byte split writes (a 16-lane group covers two rows x four q) and the
// ds_read_b128 fragment reads (hardware lane groups {0-3, 12-15, 20-27}, ...) free of bank conflicts.  A wavefront
// reads its fragments with one ds_read_b128 each and issues 6 MFMAs per output tile and K step.  Operands whose K index
// is the slow one in memory (B of an input gradient dY . W, both operands of a weight gradient) are transposed in
// registers on the way in (a thread loads a 4 x RR block and packs along k).
//
// The MFMA is issued with the operands swapped (D' = Bfrag . Afrag = C^T tile): a lane then holds four CONSECUTIVE
// columns of one output row, so the epilogue reads the addend / bias and stores the result as float4.
//
// Results are bitwise reproducible (fixed k order, no atomics, no split-K).  Up to 8 problems share a launch.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "bf16x3.h"
#include "drop_hash.h"

#ifdef GX_ABLATE_SPLIT   // diagnostic build: the stagers store raw bits (no split arithmetic); results are garbage
#define split_pair(x, y, a, b, c) ((a) = __float_as_uint(x), (b) = __float_as_uint(y), (c) = (a) ^ (b))
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int GX_STAGERS = 512;      // eight wavefronts load, split and stage (two groups of four), behind the multiplying wavefronts (2 x 2; 2 x 4 for the 128 x 256 tile)
constexpr int GX_BK = 32;
constexpr int GX_MAXP = 8;

enum : int { GX_TRANS_A = 1, GX_TRANS_B = 2, GX_RELU = 4, GX_MEAN8 = 8 };

struct GxProb {
    const float* A;      // !TRANS_A: [M, K] (lda);  TRANS_A: [K, M] (lda)
    const float* B;      // TRANS_B:  [N, K] (ldb) -- an nn.Linear weight used as x W^T;  !TRANS_B: [K, N] (ldb)
    const float* D;      // optional addend [M, N] (ldd), may alias C
    const float* bias;   // optional [N]
    float* C;            // [M, N] (ldc)
    int64_t lda, ldb, ldc, ldd;
    int M, N, K;
    float alpha, beta;
    int flags;
    int first_tile, tiles_n, n_tiles;   // n_tiles = row tiles x column tiles x splits
    int tiles_mn;                       // row tiles x column tiles
    int chunk_steps;                    // K steps (of 32) per split
    const uint4* Bimg;                  // B pre-split into bf16 planes by hg_panel_pack, k_major form (image[k / 32][tile n / 32][k half][plane][lane]), or null
    int b_ksteps, b_tiles;              //   its K / 16 and N / 32 (N padded up to a multiple of 32)
    float* slab;                        // splits > 1: [splits][M][N] partial products (alpha applied), else null
    const int64_t* drop_seed;           // frame-mean epilogue: c[m / 8, :] = mean over the 8 rows of dropout_p(alpha a b + bias)
    uint32_t drop_threshold;            //   p * 2^32 (0: no dropout)
    float drop_inv_keep;                //   1 / (1 - p)
};

struct GxBatch {
    GxProb p[GX_MAXP];
    int n;
    int total_tiles;
};

#ifdef GX_STAMPS   // diagnostic build only (tools/gemm_stamps.py): per-wavefront s_memtime stamps of the pipeline phases
__device__ unsigned long long* gx_stamp_buf = nullptr;
#define GX_STAMP(slot)                                                                                  \
    do {                                                                                                \
        if (gx_stamp_buf && (threadIdx.x & 63) == 0 && (slot) < 64)                                      \
            gx_stamp_buf[((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 64 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#define GX_STAMP_CLOCK(slot)                                                                            \
    do {                                                                                                \
        if (gx_stamp_buf && (threadIdx.x & 63) == 0) {                                                   \
            unsigned long long* q_ = gx_stamp_buf + ((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 64 + (slot);            \
            q_[0] = __builtin_amdgcn_s_memtime();                                                        \
            q_[1] = __builtin_amdgcn_s_memrealtime();     /* 100 MHz */                                  \
        }                                                                                               \
    } while (0)
#define GX_STAMP_SIMD()                                                                                 \
    do {                                                                                                \
        if (gx_stamp_buf && (threadIdx.x & 63) == 0)                                                     \
            gx_stamp_buf[((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 64 + 33] = __builtin_amdgcn_s_getreg(2308) + 1;   /* HW_ID.SIMD_ID + 1 */ \
    } while (0)
#else
#define GX_STAMP(slot) do { } while (0)
#define GX_STAMP_SIMD() do { } while (0)
#define GX_STAMP_CLOCK(slot) do { } while (0)
#endif

// LDS image of one operand tile (R rows x 32 k): [plane 3][row block R / 32][k half 2][slot 64] x 16 bytes -- the
// 1 KiB of one v_mfma_f32_32x32x16_bf16 operand fragment (32 rows x 16 k) is contiguous, lane (h = k / 8, r = row)
// owns 16 bytes at slot h * 32 + (r ^ 2q), q = 2 * (k half) + h.  `k4` = k / 4 in 0 .. 7.
template <int R>
__device__ __forceinline__ int slot_index(int plane, int row, int k4) {
    const int q = k4 >> 1;                      // 2 * khalf + h
    return ((plane * (R / 32) + (row >> 5)) * 2 + (k4 >> 2)) * 64 + (q & 1) * 32 + ((row & 31) ^ (2 * q));
}

// A tile's global loads never branch: rows / columns past the matrix edge are CLAMPED to the last valid one (their
// products only reach output rows / columns that the epilogue does not store), and only the last, partial K step of a
// K that is not a multiple of 32 zero-fills (an unconditional load from a clamped address, then a select).
// (Measured and not kept: two tiles per wavefront in flight -- three register sets in rotation, or inline-asm loads
// with hand-counted vmcnt.  hipcc drains every outstanding load before the first use of the older tile, and with the
// waits counted by hand the stager step did not get shorter either: what bounds a step is LDS bandwidth, 6 bytes
// written and 12 read per staged element, not memory latency.)

// ---- operand whose K index is CONTIGUOUS in memory: rows [row0, row0 + R) of src[rows_total][K] -------------------
// a wavefront-load covers 8 rows x 128 bytes (whole cache lines)
template <int R>
struct LoadKC {
    static constexpr int NL = R / 32;
    const float* p[NL];
    __device__ __forceinline__ void init(const float* __restrict__ src, int64_t ld, int row0, int rows_total) {
        const int t = threadIdx.x & 255;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            int row = row0 + (t >> 3) + 32 * i;
            row = row < rows_total ? row : rows_total - 1;
            p[i] = src + (int64_t)row * ld + (t & 7) * 4;
        }
    }
    struct Regs { float4 v[NL]; };
    __device__ __forceinline__ void fetch(Regs& r, int kt) const {
#pragma unroll
        for (int i = 0; i < NL; ++i) r.v[i] = *reinterpret_cast<const float4*>(p[i] + kt * GX_BK);
    }
    __device__ __forceinline__ void fetch_tail(Regs& r, int kt, int rem) const {   // rem = K - kt * 32 in (0, 32)
        const int koff = (int)(threadIdx.x & 7) * 4;
        const bool ok = koff < rem;
#pragma unroll
        for (int i = 0; i < NL; ++i) {       // lanes past K re-read the step's first columns, then select zero
            const float4 x = *reinterpret_cast<const float4*>(p[i] + kt * GX_BK - (ok ? 0 : koff));
            r.v[i] = ok ? x : f4_zero();
        }
    }
    static __device__ __forceinline__ void store(const Regs& r, uint4* __restrict__ s) {
        const int t = threadIdx.x & 255, k4 = t & 7;
        uint2* d = reinterpret_cast<uint2*>(s);
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int row = (t >> 3) + 32 * i;
            uint32_t a0, a1, a2, b0, b1, b2;
            split_pair(r.v[i].x, r.v[i].y, a0, a1, a2);
            split_pair(r.v[i].z, r.v[i].w, b0, b1, b2);
            d[slot_index<R>(0, row, k4) * 2 + (k4 & 1)] = make_uint2(a0, b0);
            d[slot_index<R>(1, row, k4) * 2 + (k4 & 1)] = make_uint2(a1, b1);
            d[slot_index<R>(2, row, k4) * 2 + (k4 & 1)] = make_uint2(a2, b2);
        }
    }
};

// ---- operand whose K index is the SLOW one: columns [col0, col0 + R) of src[K][cols_total] ----------------------------
// a thread takes 4 (k) x 2 (columns) per pass of 64 columns; the 4 k values of a column are packed along k on the way
// into LDS (a wavefront-load covers 2 k rows x 256 bytes)
template <int R>
struct LoadKS {
    static constexpr int NP = R / 64;
    const float* p[NP];
    int64_t ld;
    __device__ __forceinline__ void init(const float* __restrict__ src, int64_t ld_, int col0, int cols_total) {
        const int t = threadIdx.x & 255;
        ld = ld_;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            int c = col0 + 64 * i + (t & 31) * 2;
            c = c < cols_total ? c : cols_total - 2;          // (cols_total is a multiple of 4)
            p[i] = src + (int64_t)((t >> 5) * 4) * ld_ + c;
        }
    }
    struct Regs { float2 v[NP][4]; };
    __device__ __forceinline__ void fetch(Regs& r, int kt) const {
#pragma unroll
        for (int i = 0; i < NP; ++i)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                r.v[i][kk] = *reinterpret_cast<const float2*>(p[i] + ((int64_t)kt * GX_BK + kk) * ld);
    }
    __device__ __forceinline__ void fetch_tail(Regs& r, int kt, int rem) const {
        const int kb4 = (int)((threadIdx.x & 255) >> 5) * 4, n_ok = rem - kb4;
#pragma unroll
        for (int i = 0; i < NP; ++i)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {   // rows past K re-read the step's first row, then select zero
                const bool ok = kk < n_ok;
                const float2 x = *reinterpret_cast<const float2*>(p[i] + ((int64_t)kt * GX_BK + (ok ? kk : -kb4)) * ld);
                r.v[i][kk] = ok ? x : make_float2(0.f, 0.f);
            }
    }
    static __device__ __forceinline__ void store(const Regs& r, uint4* __restrict__ s) {
        const int t = threadIdx.x & 255, k4 = t >> 5;
        uint2* d = reinterpret_cast<uint2*>(s);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            uint32_t a0, a1, a2, b0, b1, b2;
            const int row = 64 * i + (t & 31) * 2;
            split_pair(r.v[i][0].x, r.v[i][1].x, a0, a1, a2);
            split_pair(r.v[i][2].x, r.v[i][3].x, b0, b1, b2);
            d[slot_index<R>(0, row, k4) * 2 + (k4 & 1)] = make_uint2(a0, b0);
            d[slot_index<R>(1, row, k4) * 2 + (k4 & 1)] = make_uint2(a1, b1);
            d[slot_index<R>(2, row, k4) * 2 + (k4 & 1)] = make_uint2(a2, b2);
            split_pair(r.v[i][0].y, r.v[i][1].y, a0, a1, a2);
            split_pair(r.v[i][2].y, r.v[i][3].y, b0, b1, b2);
            d[slot_index<R>(0, row + 1, k4) * 2 + (k4 & 1)] = make_uint2(a0, b0);
            d[slot_index<R>(1, row + 1, k4) * 2 + (k4 & 1)] = make_uint2(a1, b1);
            d[slot_index<R>(2, row + 1, k4) * 2 + (k4 & 1)] = make_uint2(a2, b2);
        }
    }
};

// ---- operand ALREADY split into its three bf16 planes, in MFMA operand order (hg_panel_pack: the weight of a Linear, split once
// per call instead of once per output-row tile -- two thirds of a stager's VALU work at the 128 x 256 tile): in the k_major
// form of the image a K step of 32 is contiguous over all column tiles ([k / 32][tile][k half][plane][lane] x 16 bytes; with the
// panel kernels' tile-major order the eight tiles of a step lay 48 KB apart -- one L2 channel group -- and the kernel ran at half
// speed); the stagers copy it into the ring stage with the slot swizzle of slot_index (lane (h, r) of fragment (tile, k half)
// -> slot h * 32 + (r ^ 2 q)).
template <int R>
struct LoadPre {
    static constexpr int NB = R / 32, PER = NB * 384 / 256;
    static_assert((NB * 384) % 256 == 0, "a K step of the image splits evenly over a stager group");
    const uint4* base;
    int step_stride;       // uint4 between two K steps of 32 in the image: all its column tiles x (2 k halves x 3 planes x 64 lanes)
    int off[PER];          // this thread's elements of a K step: offset in the image at k step 0 (uint4) ...
    int dst[PER];          // ... and slot in the ring stage
    __device__ __forceinline__ void init(const uint4* img, int ksteps, int tiles, int col0) {
        const int t = threadIdx.x & 255, tile0 = col0 >> 5;
        base = img;
        step_stride = tiles * 384;
        (void)ksteps;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int id = t + 256 * i;
            const int tile = id / 384, rem = id - tile * 384;       // rem = (k half * 3 + plane) * 64 + lane
            const int kh = rem / 192, pl = (rem - kh * 192) >> 6, lane = rem & 63;
            int gt = tile0 + tile;
            gt = gt < tiles ? gt : tiles - 1;                        // tiles past N: products the epilogue does not store
            off[i] = ((gt * 2 + kh) * 3 + pl) * 64 + lane;
            const int h = lane >> 5, fr = lane & 31, q = 2 * kh + h;
            dst[i] = ((pl * NB + tile) * 2 + kh) * 64 + h * 32 + (fr ^ (2 * q));
        }
    }
    // (held as float4 like the other loaders' registers: with `uint4 v[PER]` -- or twelve named uint4 -- the compiler kept the
    // struct in scratch memory, a scratch store behind every load, and the kernel ran at half the speed of the splitting stagers)
    struct Regs { float4 v[PER]; };
    __device__ __forceinline__ void fetch(Regs& r, int kt) const {
        const float4* __restrict__ p = reinterpret_cast<const float4*>(base + (int64_t)kt * step_stride);
#pragma unroll
        for (int i = 0; i < PER; ++i) r.v[i] = p[off[i]];
    }
    __device__ __forceinline__ void fetch_tail(Regs& r, int kt, int) const { fetch(r, kt); }    // (K % 32 == 0 on this path)
    __device__ __forceinline__ void store(const Regs& r, uint4* __restrict__ s) const {
        float4* __restrict__ d = reinterpret_cast<float4*>(s);
#pragma unroll
        for (int i = 0; i < PER; ++i) d[dst[i]] = r.v[i];
    }
};

template <int R, bool KS> struct Loader;
template <int R> struct Loader<R, false> : LoadKC<R> {};
template <int R> struct Loader<R, true> : LoadKS<R> {};

// block id -> logical tile such that consecutive logical tiles (the column tiles of one row tile, which share the A
// rows) run on ONE XCD's L2 (blocks are dealt round-robin over the 8 XCDs; bijective for any grid size)
__device__ __forceinline__ int xcd_remap(int b, int n) {
    const int xcd = b & 7, q = n >> 3, r = n & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

// the tile `bid` of the launch: its problem (a COPY in registers: a reference is re-read from the kernarg segment after every
// global store -- possible alias -- behind vmcnt(0)), its place in the output and its K steps [kt0, kt0 + NS) (split-K: with more
// than one split the partial tile goes to a slab that a fixed-order reduction sums afterwards)
#define GX_TILE(bid)                                                                                               \
    const int lt = xcd_remap(bid, batch.total_tiles);                                                          \
    int pi = 0;                                                                                                \
    _Pragma("unroll")                                                                                          \
    for (int i = 1; i < GX_MAXP; ++i)                                                                          \
        if (i < batch.n && lt >= batch.p[i].first_tile) pi = i;                                                \
    const GxProb P = batch.p[pi];                                                                              \
    const int local = lt - P.first_tile;                                                                       \
    const int tile = local % P.tiles_mn, split = local / P.tiles_mn;                                           \
    const int m0 = (tile / P.tiles_n) * BM, n0 = (tile % P.tiles_n) * BN;                                      \
    const int M = P.M, N = P.N, K = P.K;                                                                       \
    const int KT = K / GX_BK, rem = K - KT * GX_BK;                                                            \
    const int kt0 = split * P.chunk_steps;                                                                     \
    const int NS = min(KT + (rem > 0 ? 1 : 0) - kt0, P.chunk_steps);

// MT x NT: 32 x 32 MFMA tiles per multiplying wavefront; the four of them sit 2 x 2: block tile 64 MT x 64 NT
// WR x WC: the multiplying wavefronts' grid (2 x 2; 2 x 4 and 4 x 2 for the 128 x 256 / 256 x 128 tiles, one workgroup per CU)
//
// A workgroup walks the tiles blockIdx.x, blockIdx.x + gridDim.x, ...  The one-per-CU configurations (MINW <= 4: their ring
// leaves no room for a second workgroup) are launched with ONE WORKGROUP PER CU (round 6): nothing else on the CU could cover a
// tile's prologue -- the stagers' first loads, an HBM latency plus a split, 6-8 k cycles of a 57 k-cycle tile at K = 256 -- so
// the SAME workgroup covers it: its stagers request and split the next tile's first stages while its multipliers are in the
// epilogue of this one (whose LDS pieces lie in the ring stage the stagers reach last).  Same box: [246 k x 256].[256 x 256]
// 204 against 212 us, [1.97 M x 128].[128 x 256] 941 against 1033, the 128 x 128 tile 213 against 238; FAFormer's step 19.0
// against 19.5-19.6 ms.  (EQH_X6_TILES_PER_WG=n caps a workgroup's tiles for A/B runs: 1 is the old launch; 4 loses to both --
// 304 workgroups of 4 tiles on 256 CUs are two rounds.)  The two-per-CU configurations keep one tile per workgroup: their CU
// partner covers the prologue.
template <int MT, int NT, int S, int MINW, bool A_KS, bool B_KS, int WR = 2, int WC = 2, bool B_PRE = false>
__global__ void __launch_bounds__(64 * WR * WC + GX_STAGERS, MINW)
k_gemm_x6(const GxBatch batch) {
    constexpr int BM = 32 * WR * MT, BN = 32 * WC * NT;
    constexpr int NMT = 64 * WR * WC;                                              // multiplying threads
    constexpr int SA = 3 * (BM / 32) * 2 * 64, SB = 3 * (BN / 32) * 2 * 64;       // uint4 per stage
    constexpr bool PERSIST = MINW <= 4;
    constexpr int RING = S * (SA + SB);
    __shared__ uint4 s_mem[RING];                                                  // ring of stages, each [A | B]

    // Two roles, one barrier per K step (and one before a tile's first).  During step k the four MULTIPLIERS (wavefronts 0-3)
    // read the fragments of tile k from ring stage k % S and issue its MFMAs; the STAGERS (wavefronts 4-11, two groups of four)
    // keep the ring S - 1 tiles ahead: the group whose turn it is splits tile k + S - 1 -- which it requested from global memory
    // during the PREVIOUS step -- into stage (k + S - 1) % S, the stage the multipliers left at the last barrier, while
    // the other group requests tile k + S.  A stager is plain synchronous code (load, wait, split, store): the two
    // groups alternating hide the memory latency, not a register pipeline, so the compiler's own waits are exact.
    // A SIMD hosts one multiplier and two stagers per resident block; the hardware interleaves the stagers' VALU work
    // with the multipliers' MFMAs (the 32 x 32 x 16 MFMA holds the vector issue port for 8 of its 32 cycles).
    GX_STAMP_SIMD();
    if (threadIdx.x >= NMT) {
#ifdef GX_PRIO_STAGERS
        __builtin_amdgcn_s_setprio(GX_PRIO_STAGERS);
#endif
        int bid = (int)blockIdx.x;
        do {
        GX_TILE(bid)
        GX_STAMP(0);
        using LA = Loader<BM, A_KS>;
        using LB = std::conditional_t<B_PRE, LoadPre<BN>, Loader<BN, B_KS>>;
        LA la;
        LB lb;
        la.init(P.A, P.lda, m0, M);
        if constexpr (B_PRE) lb.init(P.Bimg, P.b_ksteps, P.b_tiles, n0);
        else lb.init(P.B, P.ldb, n0, N);
        typename LA::Regs ra;
        typename LB::Regs rb;
        const int grp = ((int)threadIdx.x - NMT) >> 8;      // 0 or 1 (wavefront-uniform)
        auto fetch = [&](int w) {                           // w: step within this split
            if (w >= NS) return;
#ifdef GX_ABLATE_STAGE
            return;
#endif
            if (kt0 + w < KT) {
                la.fetch(ra, kt0 + w);
                if constexpr (!B_PRE) lb.fetch(rb, kt0 + w);     // (the pre-split planes are fetched where they are staged: below)
            } else {
                la.fetch_tail(ra, kt0 + w, rem);
                lb.fetch_tail(rb, kt0 + w, rem);
            }
        };
        auto stage = [&](int w) {
            if (w >= NS) return;
#ifdef GX_ABLATE_STAGE
            return;
#endif
            uint4* st = s_mem + (w % S) * (SA + SB);
            if constexpr (B_PRE) {
                // planes from L2 straight through registers into the stage (held across the barrier, as the fp32 operands are, the
                // twelve registers went to scratch memory -- a scratch store behind every load -- and the kernel ran at half speed);
                // requested BEFORE the A tile is split, stored behind it
                typename LB::Regs tb;
                lb.fetch(tb, kt0 + w);
                LA::store(ra, st);
                lb.store(tb, st + SA);
            } else {
                LA::store(ra, st);
                LB::store(rb, st + SA);
            }
        };
        // tiles 0 .. S-2 before the first barrier (even tiles by group 0, odd ones by group 1), and the request for
        // tile S - 1, which is split during step 0
#pragma unroll
        for (int w = 0; w < S - 1; ++w)
            if ((w & 1) == grp) { fetch(w); stage(w); }
        if (((S - 1) & 1) == grp) fetch(S - 1);
        GX_STAMP(1);
        __syncthreads();
        GX_STAMP(2);
        for (int k = 0; k < NS; ++k) {
            const int w = k + S - 1;
            if ((w & 1) == grp) stage(w);
            else fetch(w + 1);
            GX_STAMP(k < 14 ? 3 + 2 * k : 63);
            __syncthreads();
            GX_STAMP(k < 14 ? 4 + 2 * k : 63);
        }
        GX_STAMP(31);
        } while (PERSIST && (bid += (int)gridDim.x) < batch.total_tiles);
        return;
    }

#ifdef GX_PRIO_MULT
    __builtin_amdgcn_s_setprio(GX_PRIO_MULT);
#endif
    GX_STAMP_CLOCK(34);
    int bid = (int)blockIdx.x;
    do {
    GX_TILE(bid)
    GX_STAMP(0);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WC, wn = wave % WC;
    const int fh = lane >> 5, fr = lane & 31;
    const int frag_off0 = fh * 32 + (fr ^ (2 * fh)), frag_off1 = 64 + fh * 32 + (fr ^ (2 * (2 + fh)));   // k halves

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

    // one K step from ring stage k % S is two 16-deep halves, each 3 x (MT + NT) fragment reads and 6 x MT x NT MFMAs (smallest
    // terms first).  (Measured and not kept, round 6: with two multiplying wavefronts per SIMD, the second one running half a step
    // late -- a step's second-half fragments held in registers across the barrier and multiplied while the first one reads:
    // [246 k x 256].[256 x 256] 232 against 205 us; the two in step, reading together and multiplying together, are the faster form.)
    bf16x8 fa[3][MT], fb[3][NT];
    auto rd = [&](const uint4* __restrict__ sa, const uint4* __restrict__ sb, int off) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
                fa[p][m] = __builtin_bit_cast(bf16x8, sa[(p * (BM / 32) + wm * MT + m) * 128 + off]);
#pragma unroll
            for (int n = 0; n < NT; ++n)
                fb[p][n] = __builtin_bit_cast(bf16x8, sb[(p * (BN / 32) + wn * NT + n) * 128 + off]);
        }
    };
#define GX_MM(PA, PB)                                                                                              \
        _Pragma("unroll") for (int m = 0; m < MT; ++m)                                                             \
        _Pragma("unroll") for (int n = 0; n < NT; ++n)                                                             \
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[PB][n], fa[PA][m], acc[m][n], 0, 0, 0);
    auto mm = [&]() {
#ifndef GX_ABLATE_MFMA
        GX_MM(1, 1) GX_MM(0, 2) GX_MM(2, 0) GX_MM(0, 1) GX_MM(1, 0) GX_MM(0, 0)
#else
        _Pragma("unroll") for (int p = 0; p < 3; ++p) {
            _Pragma("unroll") for (int m = 0; m < MT; ++m) { asm volatile("" :: "v"(fa[p][m])); }
            _Pragma("unroll") for (int n = 0; n < NT; ++n) { asm volatile("" :: "v"(fb[p][n])); }
        }
#endif
    };
    GX_STAMP(1);
    __syncthreads();
    GX_STAMP(2);
    // (Measured and not kept, round 6: on the three-stage 128 x 128 tile -- stage k + 1 is complete one barrier early there -- the
    // fragments a step multiplies first read during the previous step: 230-238 against 209-212 us for the two-stage tile.)
    for (int k = 0; k < NS; ++k) {
        const uint4* __restrict__ sa = s_mem + (k % S) * (SA + SB);
        const uint4* __restrict__ sb = sa + SA;
        rd(sa, sb, frag_off0);
        mm();
        rd(sa, sb, frag_off1);
        mm();
        GX_STAMP(k < 14 ? 3 + 2 * k : 63);
        __syncthreads();
        GX_STAMP(k < 14 ? 4 + 2 * k : 63);
    }
#undef GX_MM
    GX_STAMP(32);

    // epilogue.  The MFMA ran with the operands swapped (D' = C^T tile): lane (fr, fh) holds row fr of the 32 x 32 tile
    // and, in register group g, the four consecutive columns 8 g + 4 fh .. + 3.  Stored like that a wavefront-store would
    // touch 32 rows x 32 bytes (quarter cache lines: the write path, not the MFMA, then bounds short-K products), so
    // the tile goes through the wavefront's own piece of LDS (the ring is dead after the last barrier) and comes back
    // as 8 rows x 128 bytes per instruction: whole lines for the addend, the bias and the result.
    const float alpha = P.alpha, beta = P.beta;
    const bool relu = (P.flags & GX_RELU) != 0;
    const float* __restrict__ D = P.D;
    const float* __restrict__ bias = P.bias;
    // The pieces lie in ring stage S - 1: the stagers, who after a tile's last barrier go on to the next tile of the workgroup, write
    // its first S - 1 stages (0 .. S - 2) before the barrier that the multipliers reach only after this epilogue.
    constexpr int EP_LD = 36;                                  // floats per staged row (32 + 4: conflict-free b128 writes)
    static_assert((NMT / 64) * 32 * EP_LD * 4 <= (SA + SB) * 16, "the epilogue pieces fit one ring stage");
    float* ep = reinterpret_cast<float*>(s_mem + (S - 1) * (SA + SB)) + wave * (32 * EP_LD);
    int le = lane;
    if constexpr (PERSIST) asm volatile("" : "+v"(le));       // (the epilogue's lane arithmetic stays in the epilogue: hoisted out of the
                                                               // tile loop it cost 34 spilled registers, reloaded from scratch memory here)
    const int er = le >> 3, ec = (le & 7) * 4;                 // read-back: row er + 8 i, columns ec .. ec + 3
    const int pr = le & 31, ph = le >> 5;                      // (fr, fh again)
    // (the addend / bias loads are unconditional, from clamped addresses, in their own instantiation: loads inside
    // branches make the compiler wait vmcnt(0) -- i.e. for every earlier STORE too -- before each use)
    auto finish = [&](auto has_d, auto has_bias) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(ep + pr * EP_LD + 8 * g + 4 * ph) =
                        make_float4(acc[m][n][4 * g + 0], acc[m][n][4 * g + 1], acc[m][n][4 * g + 2], acc[m][n][4 * g + 3]);
                const int col = n0 + (wn * NT + n) * 32 + ec;
                const int colc = col < N ? col : N - 4;
                const int row0 = m0 + (wm * MT + m) * 32 + er;
                float4 bv = f4_zero(), dv[4];
                if constexpr (decltype(has_bias)::value) bv = *reinterpret_cast<const float4*>(bias + colc);
                if constexpr (decltype(has_d)::value) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = row0 + 8 * i;
                        dv[i] = *reinterpret_cast<const float4*>(D + (int64_t)(row < M ? row : M - 1) * P.ldd + colc);
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = row0 + 8 * i;
                    const float4 a = *reinterpret_cast<const float4*>(ep + (er + 8 * i) * EP_LD + ec);
                    float4 o = make_float4(alpha * a.x, alpha * a.y, alpha * a.z, alpha * a.w);
                    if constexpr (decltype(has_d)::value) {
                        o.x = fmaf(beta, dv[i].x, o.x); o.y = fmaf(beta, dv[i].y, o.y);
                        o.z = fmaf(beta, dv[i].z, o.z); o.w = fmaf(beta, dv[i].w, o.w);
                    }
                    o.x += bv.x; o.y += bv.y; o.z += bv.z; o.w += bv.w;
                    if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
                    if (row < M && col < N) *reinterpret_cast<float4*>(P.C + (int64_t)row * P.ldc + col) = o;
                }
            }
        }
    };
    // Frame-mean epilogue (FAFormer's frame MLP, fa_former_layer.py:61-120: fc2 -> dropout -> mean over the 8 sign frames):
    // rows 8 e .. 8 e + 7 of the product are the frames of row e of the output, so the [M, N] product never reaches
    // memory -- c[e, :] = 1/8 sum_f keep(8 e + f, :) * (alpha ab + bias)[8 e + f, :], the keep decisions being the hash of
    // (seed, element of the virtual [M, N] tensor) that faf_dropout_mean_fwd / _bwd use.  A lane takes four consecutive
    // rows of a frame group and four columns; its partner (lane ^ 8) has the other four rows.
    auto finish_mean = [&](auto has_bias) {
        const uint32_t thr = P.drop_threshold;
        const float inv_keep = P.drop_inv_keep;
        const DropKey key = drop_key(thr ? (uint64_t)*P.drop_seed : 0);
        const int grp = le >> 4, part = (le >> 3) & 1;              // frame group 0..3 of the 32 rows, its lower / upper half
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(ep + pr * EP_LD + 8 * g + 4 * ph) =
                        make_float4(acc[m][n][4 * g + 0], acc[m][n][4 * g + 1], acc[m][n][4 * g + 2], acc[m][n][4 * g + 3]);
                const int col = n0 + (wn * NT + n) * 32 + ec;
                const int colc = col < N ? col : N - 4;
                float4 bv = f4_zero();
                if constexpr (decltype(has_bias)::value) bv = *reinterpret_cast<const float4*>(bias + colc);
                const int rloc = grp * 8 + part * 4;
                const int64_t row0 = (int64_t)m0 + (wm * MT + m) * 32 + rloc;
                float4 sum = f4_zero();
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 a = *reinterpret_cast<const float4*>(ep + (rloc + i) * EP_LD + ec);
                    float4 o = make_float4(fmaf(alpha, a.x, bv.x), fmaf(alpha, a.y, bv.y), fmaf(alpha, a.z, bv.z), fmaf(alpha, a.w, bv.w));
                    if (thr) {
                        const uint64_t e = (uint64_t)((row0 + i) * N + col);
                        keep_scale4(key, e, thr, inv_keep, o);
                    }
                    if (row0 + i < M) f4_add(sum, o);
                }
                // rows in the order of the unfused pass (f = 0 .. 7): lower half first, then the partner's upper half
                float4 other;
                other.x = __shfl_xor(sum.x, 8); other.y = __shfl_xor(sum.y, 8);
                other.z = __shfl_xor(sum.z, 8); other.w = __shfl_xor(sum.w, 8);
                if (part == 0 && row0 < M && col < N) {
                    f4_add(sum, other);
                    sum.x *= 0.125f; sum.y *= 0.125f; sum.z *= 0.125f; sum.w *= 0.125f;
                    *reinterpret_cast<float4*>(P.C + (row0 >> 3) * P.ldc + col) = sum;
                }
            }
        }
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    if (P.flags & GX_MEAN8) { if (bias) finish_mean(T_{}); else finish_mean(F_{}); }
    else if (P.slab) {                       // split-K partial: alpha * acc, no addend / bias / activation (the reduction adds them)
        float* __restrict__ sl = P.slab + (int64_t)split * M * N;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(ep + pr * EP_LD + 8 * g + 4 * ph) =
                        make_float4(acc[m][n][4 * g + 0], acc[m][n][4 * g + 1], acc[m][n][4 * g + 2], acc[m][n][4 * g + 3]);
                const int col = n0 + (wn * NT + n) * 32 + ec;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = m0 + (wm * MT + m) * 32 + er + 8 * i;
                    const float4 a = *reinterpret_cast<const float4*>(ep + (er + 8 * i) * EP_LD + ec);
                    if (row < M && col < N)
                        *reinterpret_cast<float4*>(sl + (int64_t)row * N + col) = make_float4(alpha * a.x, alpha * a.y, alpha * a.z, alpha * a.w);
                }
            }
        }
    } else if (D) { if (bias) finish(T_{}, T_{}); else finish(T_{}, F_{}); }
    else   { if (bias) finish(F_{}, T_{}); else finish(F_{}, F_{}); }
    GX_STAMP(31);
    } while (PERSIST && (bid += (int)gridDim.x) < batch.total_tiles);   // the tiles of this workgroup
    GX_STAMP_CLOCK(36);
}

template <int MT, int NT, int S, int MINW, int WR = 2, int WC = 2>
int launch(const GxBatch& b, bool a_ks, bool b_ks, hipStream_t stream, bool b_pre = false) {
    // the one-per-CU configurations: one workgroup per CU walks its share of the tiles (a multiple of 8 workgroups: a workgroup's
    // tiles stay on its XCD's L2)
    static const int per_wg = [] { const char* e = getenv("EQH_X6_TILES_PER_WG"); const int v = e ? atoi(e) : 0; return v > 0 ? v : (1 << 20); }();
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n & ~7;
    }();
    int n_wg = b.total_tiles;
    if (MINW <= 4 && per_wg > 1 && b.total_tiles > cus) {       // (never fewer workgroups than CUs)
        n_wg = (((b.total_tiles + per_wg - 1) / per_wg) + 7) & ~7;
        if (n_wg < cus) n_wg = cus;
    }
    const dim3 grid(n_wg), block(64 * WR * WC + GX_STAGERS);
    if (b_pre) {
        if (a_ks) return EQH_ERR_ARG;
        hipLaunchKernelGGL((k_gemm_x6<MT, NT, S, MINW, false, false, WR, WC, true>), grid, block, 0, stream, b);
        EQH_CHECK_LAUNCH();
        return EQH_OK;
    }
    if (!a_ks && !b_ks) hipLaunchKernelGGL((k_gemm_x6<MT, NT, S, MINW, false, false, WR, WC>), grid, block, 0, stream, b);
    else if (!a_ks && b_ks) hipLaunchKernelGGL((k_gemm_x6<MT, NT, S, MINW, false, true, WR, WC>), grid, block, 0, stream, b);
    else if (a_ks && b_ks) hipLaunchKernelGGL((k_gemm_x6<MT, NT, S, MINW, true, true, WR, WC>), grid, block, 0, stream, b);
    else hipLaunchKernelGGL((k_gemm_x6<MT, NT, S, MINW, true, false, WR, WC>), grid, block, 0, stream, b);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}

}  // namespace

#ifdef GX_STAMPS
extern "C" int hg_gemm_x6_debug_stamps(void* buf) {
    unsigned long long* p = static_cast<unsigned long long*>(buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(gx_stamp_buf), &p, sizeof(p)) == hipSuccess ? EQH_OK : EQH_ERR_ARG;
}
#endif

// The five tile configurations and what a workgroup of each costs, from tools/gemm_bench.py on MI355X (microseconds per
// 32-deep K step and per tile for prologue + epilogue, with every CU holding its `slots / CUs` workgroups):
//   64 x 64 and 128 x 64 run two workgroups per CU (512 slots), the others one (256 slots)
struct GxCfg {
    int id, TM, TN, slots;
    float t_step, t_fixed;
};
static const GxCfg GX_CFGS[5] = {{64, 64, 64, 512, 0.95f, 2.0f},
                                 {128, 128, 64, 512, 1.55f, 2.3f},
                                 {256, 128, 128, 256, 1.25f, 5.0f},
                                 {512, 128, 256, 256, 2.41f, 7.2f},     // eight multiplying wavefronts
                                 {513, 256, 128, 256, 2.41f, 7.2f}};
static inline const GxCfg& gx_cfg(int id) {
    for (const GxCfg& c : GX_CFGS)
        if (c.id == id) return c;
    return GX_CFGS[0];
}

// split-K plan of one problem: splits > 1 when the output tiles alone leave most of the chip idle and K is long
static inline void gx_plan(int64_t m, int n, int k, const GxCfg& c, int* splits, int* chunk_steps) {
    const int64_t tiles = ((m + c.TM - 1) / c.TM) * ((n + c.TN - 1) / c.TN);
    const int steps = (k + GX_BK - 1) / GX_BK;
    int sp = 1;
    if (tiles > 0 && tiles < c.slots / 2 && steps >= 64) {     // (an empty problem, m == 0, has no tiles: no plan)
        const int64_t want = (2 * c.slots + tiles - 1) / tiles;   // two rounds of workgroups in all
        sp = (int)(want < steps / 16 ? want : steps / 16);        // at least 16 steps per split
        if (sp < 1) sp = 1;
        if (sp > 4096) sp = 4096;
    }
    int ch = (steps + sp - 1) / sp;
    sp = (steps + ch - 1) / ch;
    *splits = sp;
    *chunk_steps = ch;
}

// split-K sums slabs into c: only the plain product (d absent) or the accumulating form (d == c, beta == 1) can take it;
// the plan, the cost model, the workspace query and the launch all ask this one predicate
static inline bool gx_may_split(const HgGemmProblem& q) {
    return q.m > 0 && !q.bias && !q.relu && !q.mean_rows && (q.d == nullptr || (q.d == q.c && q.beta == 1.f));
}

// Estimated time of a launch with configuration c (microseconds): the workgroups' times summed over the slots, never
// less than the longest workgroup; a CU that holds ONE workgroup of a two-per-CU configuration runs it ~1.5 x faster;
// split-K adds the slab traffic (written, then read by the reduction; mostly cache hits).
static double gx_cost(int32_t n_problems, const HgGemmProblem* pr, const GxCfg& c, bool has_ws) {
    double work = 0, longest = 0, extra = 0;
    int64_t blocks = 0;
    for (int i = 0; i < n_problems; ++i) {
        const HgGemmProblem& q = pr[i];
        if (q.m <= 0) continue;                                   // an empty problem costs nothing
        int sp = 1, ch = (q.k + GX_BK - 1) / GX_BK;
        if (has_ws && gx_may_split(q)) gx_plan(q.m, q.n, q.k, c, &sp, &ch);
        const int64_t tiles = ((q.m + c.TM - 1) / c.TM) * ((q.n + c.TN - 1) / c.TN);
        // (a K-slow operand -- a weight gradient's dY^T, a weight stored [k, n] -- costs the four-multiplier 128 x 128 tile
        // more than the eight-multiplier ones: [256 x 128] over k = 1.97 M 864 against 793 us)
        const double ts = (c.id == 256 && (q.trans_a || !q.trans_b)) ? 1.15 * c.t_step : c.t_step;
        const double bt = ch * ts + c.t_fixed;
        work += (double)(tiles * sp) * bt;
        blocks += tiles * sp;
        if (bt > longest) longest = bt;
        if (sp > 1) extra += 2.0 * sp * (double)q.m * q.n * 4.0 / 6.0e6 + 4.0;
    }
    const int cus = c.slots == 512 ? 256 : c.slots;
    double t = work / c.slots;
    if (c.slots == 512 && blocks <= cus) t = 0.65 * longest;          // one workgroup per CU, alone
    else if (blocks <= c.slots) t = longest;
    else if (blocks <= 3 * c.slots) {                                 // a few rounds: whole ones ([15 k x 768]: 363 tiles of
        const int64_t rounds = (blocks + c.slots - 1) / c.slots;      // 128 x 256 are two rounds, 46 against 42 us)
        const double whole = rounds * (work / blocks);
        if (whole > t) t = whole;
    }
    if (t < longest) t = longest;
    return t + extra;
}

// tile id (64, 128, 256, 512, 513) the launch takes for these problems
static int gx_choose(int32_t n_problems, const HgGemmProblem* pr, bool has_ws) {
    int best = 64;
    double best_t = 1e30;
    for (const GxCfg& c : GX_CFGS) {
        const double t = gx_cost(n_problems, pr, c, has_ws);
        if (t < best_t) { best_t = t; best = c.id; }
    }
    return best;
}

extern "C" int32_t hg_gemm_x6_choose_tile(int32_t n_problems, const HgGemmProblem* pr, int32_t with_workspace) {
    if (n_problems <= 0 || n_problems > GX_MAXP || !pr) return 0;
    return gx_choose(n_problems, pr, with_workspace != 0);
}

extern "C" size_t hg_gemm_x6_workspace_bytes(int32_t n_problems, const HgGemmProblem* pr, int32_t tile) {
    if (n_problems <= 0 || n_problems > GX_MAXP || !pr) return 0;
    const GxCfg& c = gx_cfg(tile != 0 ? tile : gx_choose(n_problems, pr, true));
    size_t total = 0;
    for (int i = 0; i < n_problems; ++i) {
        int sp, ch;
        if (!gx_may_split(pr[i])) continue;
        gx_plan(pr[i].m, pr[i].n, pr[i].k, c, &sp, &ch);
        if (sp > 1) total += (size_t)sp * (size_t)pr[i].m * (size_t)pr[i].n * sizeof(float);
    }
    return total;
}

extern "C" int hg_gemm_x6_batch(int32_t n_problems, const HgGemmProblem* pr, int32_t tile, void* workspace, size_t workspace_bytes,
                                void* stream_) {
    if (n_problems <= 0 || n_problems > GX_MAXP || !pr) return EQH_ERR_ARG;
    GxBatch b;
    b.n = n_problems;
    const bool a_ks = pr[0].trans_a != 0, b_ks = pr[0].trans_b == 0;
    const bool b_pre = pr[0].b_packed != nullptr;
    for (int i = 0; i < n_problems; ++i) {
        const HgGemmProblem& q = pr[i];
        if (q.m < 0 || q.n <= 0 || q.k <= 0 || !q.a || (!q.b && !q.b_packed) || !q.c) return EQH_ERR_ARG;
        if ((q.b_packed != nullptr) != b_pre) return EQH_ERR_ARG;                       // one B form per launch
        if (b_pre && ((q.k & 31) || q.trans_a || !eqh_aligned16(q.b_packed))) return EQH_ERR_ARG;   // whole K steps of 32
        if ((q.trans_a != 0) != a_ks || (q.trans_b == 0) != b_ks) return EQH_ERR_ARG;   // one operand layout per launch
        if (q.m >= (1ll << 31) - 256) return EQH_ERR_RANGE;
        if (q.mean_rows != 0 && (q.mean_rows != 8 || (q.m & 7) || q.d || q.relu || q.trans_a || !(q.drop_p >= 0.f) || !(q.drop_p < 1.f) ||
                                 (q.drop_p > 0.f && !q.drop_seed)))
            return EQH_ERR_ARG;
        // float4 / float2 accesses: the contiguous extents and the row strides are multiples of 4 floats
        if ((q.n & 3) || (q.lda & 3) || (q.ldb & 3) || (q.ldc & 3) || (q.d && (q.ldd & 3))) return EQH_ERR_ALIGN;
        if ((!a_ks && (q.k & 3)) || (a_ks && (q.m & 3)) || (!b_pre && !b_ks && (q.k & 3))) return EQH_ERR_ALIGN;
        if (!eqh_aligned16(q.a) || (!b_pre && !eqh_aligned16(q.b)) || !eqh_aligned16(q.c) || !eqh_aligned16(q.d) || !eqh_aligned16(q.bias))
            return EQH_ERR_ALIGN;
    }
    // the tile: 64 x 64 fills the chip at ~5 k-row batches; 128 x 64 amortises the operand split better; 128 x 128 (one
    // workgroup per CU) has the least operand traffic per MFMA among the four-multiplier tiles and takes the deep split-K
    // products; 128 x 256 / 256 x 128 run EIGHT multiplying wavefronts (two per SIMD): half the operand traffic of 128 x 64
    // and a second multiplier to fill each SIMD's MFMA pipe -- [246 k x 256].[256 x 256] 199 against 220 us, [1.97 M x
    // 128].[128 x 256] 943 against 1122 us, [256 x 128] over k = 1.97 M 793 against 864 us.  gx_cost estimates each.
    const int id = tile != 0 ? tile : gx_choose(n_problems, pr, workspace != nullptr);
    if (id != 64 && id != 128 && id != 256 && id != 512 && id != 513) return EQH_ERR_ARG;
    const GxCfg& cfg = gx_cfg(id);
    const int TM = cfg.TM, TN = cfg.TN;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    int64_t first = 0;
    size_t ws_used = 0;
    for (int i = 0; i < n_problems; ++i) {
        const HgGemmProblem& q = pr[i];
        GxProb& p = b.p[i];
        p.A = q.a; p.B = q.b; p.D = q.d; p.bias = q.bias; p.C = q.c;
        p.Bimg = static_cast<const uint4*>(q.b_packed);
        p.b_ksteps = q.k / 16;
        p.b_tiles = (q.n + 31) / 32;
        p.lda = q.lda; p.ldb = q.ldb; p.ldc = q.ldc; p.ldd = q.ldd;
        p.M = (int)q.m; p.N = q.n; p.K = q.k;
        p.alpha = q.alpha; p.beta = q.beta;
        p.flags = (q.trans_a ? GX_TRANS_A : 0) | (q.trans_b ? GX_TRANS_B : 0) | (q.relu ? GX_RELU : 0) | (q.mean_rows ? GX_MEAN8 : 0);
        p.drop_seed = q.drop_seed;
        p.drop_threshold = drop_threshold(q.drop_p);
        p.drop_inv_keep = drop_inv_keep(q.drop_p);
        p.tiles_n = (q.n + TN - 1) / TN;
        p.tiles_mn = (int)(((q.m + TM - 1) / TM) * p.tiles_n);
        int sp = 1, ch = (q.k + GX_BK - 1) / GX_BK;
        p.slab = nullptr;
        if (gx_may_split(q)) {
            gx_plan(q.m, q.n, q.k, cfg, &sp, &ch);
            if (sp > 1) {
                const size_t need = (size_t)sp * (size_t)q.m * (size_t)q.n * sizeof(float);
                if (!workspace || ws_used + need > workspace_bytes) {   // no room: one pass over the whole K
                    sp = 1;
                    ch = (q.k + GX_BK - 1) / GX_BK;
                } else {
                    p.slab = reinterpret_cast<float*>(static_cast<char*>(workspace) + ws_used);
                    ws_used += need;
                }
            }
        }
        p.chunk_steps = ch;
        p.n_tiles = p.tiles_mn * sp;
        p.first_tile = (int)first;
        first += p.n_tiles;
        if (first >= (1ll << 31) - 1) return EQH_ERR_RANGE;
    }
    for (int i = n_problems; i < GX_MAXP; ++i) b.p[i] = b.p[0], b.p[i].first_tile = 0x7fffffff;
    b.total_tiles = (int)first;
    if (first == 0) return EQH_OK;
    int rc;
    if (id == 512) rc = launch<2, 2, 2, 4, 2, 4>(b, a_ks, b_ks, stream, b_pre);
    else if (id == 513) rc = launch<2, 2, 2, 4, 4, 2>(b, a_ks, b_ks, stream, b_pre);
    else if (id == 256) {
        static const bool deep = [] { const char* e = getenv("EQH_X6_DEEP"); return e && e[0] == '1'; }();   // three ring stages (A/B runs)
        rc = deep ? launch<2, 2, 3, 3>(b, a_ks, b_ks, stream, b_pre) : launch<2, 2, 2, 3>(b, a_ks, b_ks, stream, b_pre);
    }
    else if (id == 128) rc = launch<2, 1, 2, 6>(b, a_ks, b_ks, stream, b_pre);   // 6 waves / SIMD = 2 blocks / CU
    else rc = launch<1, 1, 3, 6>(b, a_ks, b_ks, stream, b_pre);
    if (rc) return rc;
    // split-K problems: c = beta * d + sum of the slabs, in slab order (bitwise reproducible); beta * d with d == c and
    // beta == 1 is the accumulating form the weight gradients use
    for (int i = 0; i < n_problems; ++i) {
        const GxProb& p = b.p[i];
        if (!p.slab) continue;
        const int sp = p.n_tiles / p.tiles_mn;
        const bool acc = p.D == p.C && p.beta == 1.0f;           // (gx_may_split admits no other addend)
        rc = eqh_reduce_slabs2d_async(p.slab, sp, p.M, p.N, p.C, p.ldc, acc ? 1 : 0, stream);
        if (rc) return rc;
    }
    return EQH_OK;
}
