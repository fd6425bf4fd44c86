// Weight gradient of a Linear:  dW[o][i] (+)= alpha * sum_k dy[k][o] * x[k][i]   (dy [K,O], x [K,I], K = rows).
//
// Replaces the library GEMM autograd runs for `grad_output.t() @ input` on every nn.Linear of the path
// (mlp.py:91-99, conv.py:90-97,172-180).  At the BASELINE batch these are [256 x K] . [K x 256] products
// with K = 4608 .. 9728: 256 output tiles and a long reduction, for which hipBLASLt's best kernel
// (MT16x16x256, one wavefront per tile, no split of K) reaches 35 TFLOP/s -- 17.4 us each, 21 of them
// per step, a third of all GEMM time.
//
// Here K is split: grid = (64 x 64 output tiles) x (S chunks of K), four wavefronts per workgroup each
// walking a quarter of the chunk for the same output tile.  Both operands are K-major in memory, which is
// exactly the MFMA operand order (k in the high lane bits, m / n in the low ones), so they go from global
// memory straight into the operand registers: lane (r, q) reads ONE float4 of dy[k0+q][o0+4r ..] and one
// of x[k0+q][i0+4r ..] per 4 rows of K, and component j of the float4 belongs to the j-th of four
// interleaved 16-row MFMA tiles (rows o0 + 4r + j) -- two fully coalesced 1 KB loads feed sixteen
// v_mfma_f32_16x16x4_f32.  Loads run three register stages (24 rows of K) ahead.  The four wavefronts' 64 x 64
// accumulators meet in LDS (wavefront order), each workgroup writes one slab, and the slabs are reduced
// in chunk order by the common slab reducer -- which can add into the destination (a weight shared by L
// layer applications) and is deferred to the one batched launch of eqh_defer_flush when active.
// No atomics: bitwise reproducible.
#include <cstdlib>

#include "common.h"
#include "bf16x3.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int THREADS = 256;
constexpr int WAVES = 4;
constexpr int AHEAD = 2;  // k-steps (of 4 rows) per register stage

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

struct Stage {
    float4 a[AHEAD], b[AHEAD];
};

// rows k = k0 + 4 t + q, t < AHEAD; rows at or past k_end contribute zeros (address clamped, value masked)
__device__ __forceinline__ void load_stage(Stage& st, const float* __restrict__ pa, const float* __restrict__ pb,
                                           int64_t k0, int64_t k_end, int64_t k_clamp, int64_t O, int64_t I, int q) {   // O, I: row strides
#pragma unroll
    for (int t = 0; t < AHEAD; ++t) {
        const int64_t k = k0 + 4 * t + q;
        const int64_t kc = k < k_clamp ? k : k_clamp;
        float4 a = *reinterpret_cast<const float4*>(pa + kc * O);
        float4 b = *reinterpret_cast<const float4*>(pb + kc * I);
        if (k >= k_end) { a = make_float4(0.f, 0.f, 0.f, 0.f); b = a; }
        st.a[t] = a;
        st.b[t] = b;
    }
}

// one 64 x 64 output tile over the rows [chunk * k_chunk, (chunk + 1) * k_chunk) of K, by one workgroup
// (ldy, ldx: row strides of dy and x -- O and I for whole matrices, wider when the operands are column blocks)
__device__ __forceinline__ void wgrad_tile(const float* __restrict__ dy, const float* __restrict__ x, int64_t K,
                                           int O, int I, float* __restrict__ out, int64_t out_ld, int o0, int i0,
                                           int chunk, int64_t k_chunk, int direct, int accumulate, float alpha,
                                           float (*s_acc)[64 * 64], int64_t ldy, int64_t ldx) {   // s_acc[2][64 * 64]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    // this wavefront's rows of K: a quarter of the chunk (multiples of 4)
    const int64_t k_quarter = k_chunk / WAVES;
    const int64_t k_beg = (int64_t)chunk * k_chunk + wave * k_quarter;
    int64_t k_end = k_beg + k_quarter;
    if (k_end > K) k_end = K;
    const float* __restrict__ pa = dy + o0 + 4 * r;
    const float* __restrict__ pb = x + i0 + 4 * r;
    f32x4 acc[4][4];
#pragma unroll
    for (int ja = 0; ja < 4; ++ja)
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) acc[ja][jb] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (k_beg < k_end) {
        // four register stages in a ring: the loads of stage s+3 are issued before stage s is consumed (a
        // single stage of look-ahead left every stage waiting ~2000 cycles for its rows: 65 cycles per MFMA
        // instead of 32)
        const int64_t k_clamp = K - 1;
        constexpr int64_t STEP = 4 * AHEAD;
        Stage s0, s1, s2, s3;
        load_stage(s0, pa, pb, k_beg, k_end, k_clamp, ldy, ldx, q);
        load_stage(s1, pa, pb, k_beg + STEP, k_end, k_clamp, ldy, ldx, q);
        load_stage(s2, pa, pb, k_beg + 2 * STEP, k_end, k_clamp, ldy, ldx, q);
        auto consume = [&](const Stage& st) {
#pragma unroll
            for (int t = 0; t < AHEAD; ++t) {
                const float av[4] = {st.a[t].x, st.a[t].y, st.a[t].z, st.a[t].w};
                const float bv[4] = {st.b[t].x, st.b[t].y, st.b[t].z, st.b[t].w};
#pragma unroll
                for (int ja = 0; ja < 4; ++ja)
#pragma unroll
                    for (int jb = 0; jb < 4; ++jb) acc[ja][jb] = mfma16(av[ja], bv[jb], acc[ja][jb]);
            }
        };
        for (int64_t k0 = k_beg; k0 < k_end; k0 += 4 * STEP) {
            load_stage(s3, pa, pb, k0 + 3 * STEP, k_end, k_clamp, ldy, ldx, q);
            __builtin_amdgcn_sched_barrier(0);
            consume(s0);
            __builtin_amdgcn_sched_barrier(0);
            load_stage(s0, pa, pb, k0 + 4 * STEP, k_end, k_clamp, ldy, ldx, q);
            __builtin_amdgcn_sched_barrier(0);
            if (k0 + STEP < k_end) consume(s1);
            __builtin_amdgcn_sched_barrier(0);
            load_stage(s1, pa, pb, k0 + 5 * STEP, k_end, k_clamp, ldy, ldx, q);
            __builtin_amdgcn_sched_barrier(0);
            if (k0 + 2 * STEP < k_end) consume(s2);
            __builtin_amdgcn_sched_barrier(0);
            load_stage(s2, pa, pb, k0 + 6 * STEP, k_end, k_clamp, ldy, ldx, q);
            __builtin_amdgcn_sched_barrier(0);
            if (k0 + 3 * STEP < k_end) consume(s3);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // acc[ja][jb][g] = C[o0 + 4 (4q + g) + ja][i0 + 4 r + jb].  Two LDS images ([row][col], 32 KB: three
    // workgroups fit a CU; four images allowed only two): wavefronts 2, 3 park their partials, wavefronts 0, 1
    // add them to their own and park the sums, then all 256 threads add the two images, a float4 each per pass.
    // Order (w0 + w2) + (w1 + w3): fixed.
    auto park = [&](float* img) {
#pragma unroll
        for (int ja = 0; ja < 4; ++ja)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(img + (16 * q + 4 * g + ja) * 64 + 4 * r) =
                    make_float4(acc[ja][0][g], acc[ja][1][g], acc[ja][2][g], acc[ja][3][g]);
    };
    if (wave >= 2) park(s_acc[wave - 2]);
    __syncthreads();
    if (wave < 2) {
#pragma unroll
        for (int ja = 0; ja < 4; ++ja)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 u = *reinterpret_cast<const float4*>(s_acc[wave] + (16 * q + 4 * g + ja) * 64 + 4 * r);
                acc[ja][0][g] += u.x; acc[ja][1][g] += u.y; acc[ja][2][g] += u.z; acc[ja][3][g] += u.w;
            }
    }
    __syncthreads();
    if (wave < 2) park(s_acc[wave]);
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int e = (pass * THREADS + threadIdx.x) * 4;  // element index in the 64 x 64 tile
        const int row = e >> 6, col = e & 63;
        float4 v = *reinterpret_cast<const float4*>(s_acc[0] + e);
        {
            const float4 u = *reinterpret_cast<const float4*>(s_acc[1] + e);
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        v.x *= alpha; v.y *= alpha; v.z *= alpha; v.w *= alpha;
        float* dst;
        if (direct) {  // a single chunk: straight to the destination block
            dst = out + (int64_t)(o0 + row) * out_ld + i0 + col;
            if (accumulate) {
                const float4 old = *reinterpret_cast<const float4*>(dst);
                v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w;
            }
        } else {       // slab [chunk][O][I]
            dst = out + ((int64_t)chunk * O + o0 + row) * I + i0 + col;
        }
        *reinterpret_cast<float4*>(dst) = v;
    }
}

__global__ void __launch_bounds__(THREADS)
k_wgrad(const float* __restrict__ dy, const float* __restrict__ x, int64_t K, int O, int I, float* __restrict__ out,
        int64_t out_ld, int tiles_i, int tiles, int64_t k_chunk, int direct, int accumulate, float alpha) {
    __shared__ __attribute__((aligned(16))) float s_acc[2][64 * 64];
    // Workgroup b runs on XCD b % 8 (round-robin dispatch).  All output tiles of one K-chunk read the same
    // rows of dy and x, so a chunk's tiles go to ONE XCD: each input byte crosses the fabric once and is
    // re-read from that XCD's L2.
    int tile, chunk;
    const int n_chunks = gridDim.x / tiles;
    if ((n_chunks & 7) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        tile = j % tiles;
        chunk = xcd + 8 * (j / tiles);
    } else {
        tile = blockIdx.x % tiles;
        chunk = blockIdx.x / tiles;
    }
    wgrad_tile(dy, x, K, O, I, out, out_ld, 64 * (tile / tiles_i), 64 * (tile % tiles_i), chunk, k_chunk, direct,
               accumulate, alpha, s_acc, O, I);
}

// Many weight gradients of the same [O x I] shape in ONE launch (the backward of a model step defers them to
// its end: 21 products of [256 x K].[K x 256] at the BASELINE batch).  One product alone has too few tiles to
// fill the chip without a deep split of K, and every split costs a slab; together they fill it with a split
// of two.  blockIdx.y = product, blockIdx.x = (tile, half of K).
// (Measured and not kept: a 128 x 128 workgroup tile with eight wavefronts so that quadrant pairs share operand
// rows through L1 -- same 141 us per 21-product batch; a main loop without clamps / masks / 64-bit multiplies --
// 157 us.  The memory side alone takes 63 us of the 140, the MFMAs 86: they overlap poorly at two wavefronts
// per SIMD, which is what the 64 KB LDS epilogue buffer allows.)
constexpr int WG_MAX_BATCH = 64;     // (a WgradBatch is a by-value kernel argument of 64 x 56 bytes)
struct WgradEntry {
    const float* dy;
    const float* x;
    float* slab;      // [splits][O][I]
    int64_t K;
    int64_t ldy, ldx; // row strides (column blocks of wider matrices take part as they are)
    float alpha;
    int pad;
};
struct WgradBatch {
    WgradEntry e[WG_MAX_BATCH];
};
static_assert(sizeof(WgradBatch) <= 4096, "WgradBatch is a by-value kernel argument");

__global__ void __launch_bounds__(THREADS)
k_wgrad_batch(WgradBatch b, int O, int I, int tiles_i, int tiles, int splits) {
    __shared__ __attribute__((aligned(16))) float s_acc[2][64 * 64];
    const WgradEntry& en = b.e[blockIdx.y];
    const int tile = blockIdx.x % tiles, chunk = blockIdx.x / tiles;
    int64_t kc = (en.K + splits - 1) / splits;
    kc = (kc + 16 * AHEAD - 1) / (16 * AHEAD) * (16 * AHEAD);
    wgrad_tile(en.dy, en.x, en.K, O, I, en.slab, (int64_t)I, 64 * (tile / tiles_i), 64 * (tile % tiles_i), chunk, kc,
               0, 0, en.alpha, s_acc, en.ldy, en.ldx);
}

// ------------------------------------------------------------------------------------------------------------------------------
// Round 5: the batched weight gradients on the bf16 matrix cores (fp32-grade: the exact three-way split of bf16x3.h and the six
// products of gemm_x6.hip, smallest first into one fp32 accumulator).  The fp32 MFMA above runs at 1/16 of the bf16 rate; six
// bf16 MFMAs per product are 2.7x faster, and both operands being K-major is STILL the operand order: for v_mfma_f32_32x32x16_bf16
// lane (m = lane & 31, kg = lane >> 5) holds the eight values k = 8 kg .. 8 kg + 7 of column m -- eight dword loads, each of which
// reads two whole 128-byte row segments per wavefront, and split_pair on consecutive rows yields the packed fragment registers
// directly.  No LDS staging, no transposition.
// A workgroup of eight wavefronts owns a 128 x 128 output tile of one product over one chunk of K: wavefront (kh, qi, qj) computes
// the 64 x 64 quadrant (qi, qj) over the kh-th half of the chunk (a quadrant's operand rows are shared with its neighbours through
// L1, so a product's operands cross L2 -> CU twice instead of four times); the two K-halves meet in LDS, one slab per chunk.
// ------------------------------------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int WX_THREADS = 512;

struct WxStage {
    float a[2][8], b[2][8];      // [32-column tile][row 8 kg + i of the 16-row step]
};

// rows 8 kg + i, i < 8, of the two 32-column tiles of each operand: ga / gb are wave-uniform (the operand at row 0 of the
// step, first column of the quadrant), la / lb the lane's offset (8 kg rows down, column m).  FULL: all sixteen rows exist.
// (Measured, 21 products of [256 x 4.7 k].[4.7 k x 256] inside the step: fp32-MFMA kernel 131 us; this kernel with one stage
// of look-ahead 107, with three 91; with the MFMAs compiled out 75 -- the operand side (204 MB of rows written earlier in the
// step, dword loads, the split) bounds it, the matrix pipe needs 31.  Not kept: dy in 16-byte loads, eight of them feeding FOUR
// interleaved-column tiles of a 128 x 32 wavefront tile -- half the load instructions per MFMA, a 40-register stage and
// therefore a ring of three: 98 us.)
template <bool FULL>
__device__ __forceinline__ void wx_load(WxStage& st, const float* __restrict__ ga, const float* __restrict__ gb, unsigned la, unsigned lb,
                                        int64_t ldy, int64_t ldx, int rows_left, int kg) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const bool ok = FULL || (8 * kg + i < rows_left);
        const int64_t ri = ok ? i : 0;                    // (a row past the end re-reads the step's first row: in range, masked)
        const unsigned la_ = ok ? la : (la - (unsigned)(8 * kg * ldy)), lb_ = ok ? lb : (lb - (unsigned)(8 * kg * ldx));
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float va = (ga + ri * ldy + 32 * t)[FULL ? la : la_], vb = (gb + ri * ldx + 32 * t)[FULL ? lb : lb_];
            st.a[t][i] = ok ? va : 0.f;
            st.b[t][i] = ok ? vb : 0.f;
        }
    }
}

__device__ __forceinline__ void wx_split(const float (&v)[8], bf16x8 (&p)[3]) {
    uint4 p0, p1, p2;
    split_pair(v[0], v[1], p0.x, p1.x, p2.x);
    split_pair(v[2], v[3], p0.y, p1.y, p2.y);
    split_pair(v[4], v[5], p0.z, p1.z, p2.z);
    split_pair(v[6], v[7], p0.w, p1.w, p2.w);
    p[0] = __builtin_bit_cast(bf16x8, p0);
    p[1] = __builtin_bit_cast(bf16x8, p1);
    p[2] = __builtin_bit_cast(bf16x8, p2);
}

__global__ void __launch_bounds__(WX_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_wgrad_batch_x3(WgradBatch b, int n_prod, int O, int I, int tiles_i, int tiles, int splits, int use_buf) {
    __shared__ __attribute__((aligned(16))) float s_acc[4][64 * 64];      // the second K-half's quadrants
    // Work unit = (product, chunk of K, 128 x 128 tile).  The tiles of one (product, chunk) read the same operand rows: they go
    // to ONE XCD (blocks are dealt round-robin over the eight XCDs: b and b + 8 share one -- speed only, nothing depends on
    // it), next to each other in its dispatch order, so the rows cross the fabric once and the other tiles hit that XCD's L2.
    const int groups = n_prod * splits;
    const int xcd = blockIdx.x & 7, j_ = blockIdx.x >> 3;
    const int group = xcd + 8 * (j_ / tiles), tile = j_ % tiles;
    if (group >= groups) return;                                          // whole workgroup
    const WgradEntry& en = b.e[group / splits];
    const int chunk = group % splits;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kh = wave >> 2, qd = wave & 3, qi = qd >> 1, qj = qd & 1;
    const int m = lane & 31, kg = lane >> 5;
    const int o0 = 128 * (tile / tiles_i) + 64 * qi, i0 = 128 * (tile % tiles_i) + 64 * qj;
    const bool active = o0 < O && i0 < I;
    int64_t kc = (en.K + splits - 1) / splits;
    kc = (kc + 31) / 32 * 32;                                   // two halves of whole 16-row steps
    const int64_t c_beg = (int64_t)chunk * kc;
    const int64_t k_beg = c_beg + kh * (kc / 2);
    int64_t k_end = k_beg + kc / 2;
    if (k_end > en.K) k_end = en.K;
    f32x16 acc[2][2];
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[ta][tb][i] = 0.f;
    if (active && k_beg < k_end) {
        const int64_t ldy = en.ldy, ldx = en.ldx;
        const float* __restrict__ ga = en.dy + k_beg * ldy + o0;       // wave-uniform
        const float* __restrict__ gb = en.x + k_beg * ldx + i0;
        const unsigned la = (unsigned)(8 * kg * ldy + m), lb = (unsigned)(8 * kg * ldx + m);
        auto consume = [&](const WxStage& st) {
            bf16x8 A[2][3], B[2][3];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                wx_split(st.a[t], A[t]);
                wx_split(st.b[t], B[t]);
            }
#pragma unroll
            for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                for (int tb = 0; tb < 2; ++tb) {
                    // smallest terms first, as gemm_x6.hip: a1 b1, a0 b2, a2 b0, a0 b1, a1 b0, a0 b0
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[ta][1], B[tb][1], acc[ta][tb], 0, 0, 0);
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[ta][0], B[tb][2], acc[ta][tb], 0, 0, 0);
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[ta][2], B[tb][0], acc[ta][tb], 0, 0, 0);
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[ta][0], B[tb][1], acc[ta][tb], 0, 0, 0);
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[ta][1], B[tb][0], acc[ta][tb], 0, 0, 0);
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[ta][0], B[tb][0], acc[ta][tb], 0, 0, 0);
                }
        };
        // whole 16-row steps through a ring of four register stages (three in flight behind the one being multiplied), then
        // the ragged last step with its rows masked
        const int n_full = (int)((k_end - k_beg) / 16), tail = (int)((k_end - k_beg) % 16);
        WxStage s0, s1, s2, s3;
        // whole steps through buffer loads (scalar row offsets: see k_wgrad_batch_x3w)
        const bool buf_ok = use_buf && en.K * ldy < ((int64_t)1 << 29) && en.K * ldx < ((int64_t)1 << 29);
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)en.dy, 0, buf_ok ? (int)(en.K * ldy * 4) : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)en.x, 0, buf_ok ? (int)(en.K * ldx * 4) : 0, 0x00020000);
        const unsigned va = 4u * la, vb = 4u * lb;
        const unsigned ldyb = (unsigned)(4 * ldy), ldxb = (unsigned)(4 * ldx);
        const unsigned sa0 = (unsigned)((k_beg * ldy + o0) * 4), sb0 = (unsigned)((k_beg * ldx + i0) * 4);
        auto fetch = [&](WxStage& st, int step) {
            if (step >= n_full) return;
            if (!buf_ok) {
                wx_load<true>(st, ga + (int64_t)step * 16 * ldy, gb + (int64_t)step * 16 * ldx, la, lb, ldy, ldx, 16, kg);
                return;
            }
            const unsigned sa = sa0 + (unsigned)step * 16u * ldyb, sb = sb0 + (unsigned)step * 16u * ldxb;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    st.a[t][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ra, va + 128u * t, sa + (unsigned)i * ldyb, 0));
                    st.b[t][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rb, vb + 128u * t, sb + (unsigned)i * ldxb, 0));
                }
        };
        fetch(s0, 0);
        fetch(s1, 1);
        fetch(s2, 2);
        for (int step = 0; step < n_full; step += 4) {
            fetch(s3, step + 3);
            __builtin_amdgcn_sched_barrier(0);
            consume(s0);
            __builtin_amdgcn_sched_barrier(0);
            fetch(s0, step + 4);
            __builtin_amdgcn_sched_barrier(0);
            if (step + 1 < n_full) consume(s1);
            __builtin_amdgcn_sched_barrier(0);
            fetch(s1, step + 5);
            __builtin_amdgcn_sched_barrier(0);
            if (step + 2 < n_full) consume(s2);
            __builtin_amdgcn_sched_barrier(0);
            fetch(s2, step + 6);
            __builtin_amdgcn_sched_barrier(0);
            if (step + 3 < n_full) consume(s3);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (tail) {
            wx_load<false>(s0, ga + (int64_t)n_full * 16 * ldy, gb + (int64_t)n_full * 16 * ldx, la, lb, ldy, ldx, tail, kg);
            consume(s0);
        }
    }
    // accumulator register e of tile (ta, tb): row 32 ta + (e & 3) + 8 (e >> 2) + 4 kg, column 32 tb + m of the quadrant
    float* img = s_acc[qd];
    if (kh == 1) {
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
            for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    img[(32 * ta + (e & 3) + 8 * (e >> 2) + 4 * kg) * 64 + 32 * tb + m] = acc[ta][tb][e];
    }
    __syncthreads();
    if (kh == 0 && active) {
        float* __restrict__ out = en.slab + ((int64_t)chunk * O + o0) * I + i0;
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
            for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = 32 * ta + (e & 3) + 8 * (e >> 2) + 4 * kg, col = 32 * tb + m;
                    out[(int64_t)row * I + col] = en.alpha * (acc[ta][tb][e] + img[row * 64 + col]);
                }
    }
}

// ---- the WIDE form: ONE wavefront per SIMD owning a 64 x 128 tile (2 x 4 MFMA tiles, 128 accumulator registers), four
// wavefronts = a 128 x 256 workgroup tile.  Six operand fragments are split for eight tile products (0.75 splits per product
// against 1.0 of the 64 x 64 quadrants above, where the split VALU work equals the MFMA time), an operand row crosses L2 -> CU
// 1.5 times per 128 x 256 outputs instead of twice per 128 x 128, no K-halves to merge.  The price is one wavefront per SIMD
// (~380 registers): the four-stage register ring and the wavefront's own MFMAs are what hides the loads.
struct WxStageW {
    float a[2][8], b[4][8];
};

template <bool FULL>
__device__ __forceinline__ void wxw_load(WxStageW& st, const float* __restrict__ ga, const float* __restrict__ gb, unsigned la,
                                         unsigned lb, int64_t ldy, int64_t ldx, int rows_left, int kg) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const bool ok = FULL || (8 * kg + i < rows_left);
        const int64_t ri = ok ? i : 0;
        const unsigned la_ = ok ? la : (la - (unsigned)(8 * kg * ldy)), lb_ = ok ? lb : (lb - (unsigned)(8 * kg * ldx));
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float va = (ga + ri * ldy + 32 * t)[FULL ? la : la_];
            st.a[t][i] = ok ? va : 0.f;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float vb = (gb + ri * ldx + 32 * t)[FULL ? lb : lb_];
            st.b[t][i] = ok ? vb : 0.f;
        }
    }
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_wgrad_batch_x3w(WgradBatch b, int n_prod, int O, int I, int tiles_i, int tiles, int splits, int use_buf) {
    const int groups = n_prod * splits;
    const int xcd = blockIdx.x & 7, j_ = blockIdx.x >> 3;
    const int group = xcd + 8 * (j_ / tiles), tile = j_ % tiles;
    if (group >= groups) return;                                          // whole workgroup
    const WgradEntry& en = b.e[group / splits];
    const int chunk = group % splits;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wi = wave >> 1, wj = wave & 1;
    const int m = lane & 31, kg = lane >> 5;
    const int o0 = 128 * (tile / tiles_i) + 64 * wi, i0 = 256 * (tile % tiles_i) + 128 * wj;
    if (o0 >= O || i0 >= I) return;                                       // (whole wavefront; no barrier below)
    int64_t kc = (en.K + splits - 1) / splits;
    kc = (kc + 15) / 16 * 16;
    const int64_t k_beg = (int64_t)chunk * kc;
    int64_t k_end = k_beg + kc;
    if (k_end > en.K) k_end = en.K;
    f32x16 acc[2][4];
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[ta][tb][i] = 0.f;
    if (k_beg < k_end) {
        const int64_t ldy = en.ldy, ldx = en.ldx;
        const float* __restrict__ ga = en.dy + k_beg * ldy + o0;
        const float* __restrict__ gb = en.x + k_beg * ldx + i0;
        const unsigned la = (unsigned)(8 * kg * ldy + m), lb = (unsigned)(8 * kg * ldx + m);
        const bool buf_ok = use_buf && en.K * ldy < ((int64_t)1 << 29) && en.K * ldx < ((int64_t)1 << 29);   // byte offsets below 2 GB
        auto consume = [&](const WxStageW& st) {
            bf16x8 A[2][3];
            wx_split(st.a[0], A[0]);
            wx_split(st.a[1], A[1]);
#pragma unroll
            for (int tb = 0; tb < 4; ++tb) {
                bf16x8 B[3];
                wx_split(st.b[tb], B);
#pragma unroll
                for (int ta = 0; ta < 2; ++ta) {
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[ta][1], B[1], acc[ta][tb], 0, 0, 0);
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[ta][0], B[2], acc[ta][tb], 0, 0, 0);
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[ta][2], B[0], acc[ta][tb], 0, 0, 0);
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[ta][0], B[1], acc[ta][tb], 0, 0, 0);
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[ta][1], B[0], acc[ta][tb], 0, 0, 0);
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[ta][0], B[0], acc[ta][tb], 0, 0, 0);
                }
            }
        };
        const int n_full = (int)((k_end - k_beg) / 16), tail = (int)((k_end - k_beg) % 16);
        WxStageW s0, s1, s2, s3;
        // Whole steps through BUFFER loads: the row of a load is a scalar offset (an SGPR operand), the lane's part one VGPR for
        // the whole kernel -- no vector address arithmetic (with 64-bit global addresses it was ~100 of a step's ~400 VALU
        // instructions, and the VALU issue is what bounds this kernel).  Operands of 4 GB and more keep the global loads.
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)en.dy, 0, buf_ok ? (int)(en.K * ldy * 4) : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)en.x, 0, buf_ok ? (int)(en.K * ldx * 4) : 0, 0x00020000);
        const unsigned va = 4u * la, vb = 4u * lb;
        const unsigned ldyb = (unsigned)(4 * ldy), ldxb = (unsigned)(4 * ldx);
        const unsigned sa0 = (unsigned)((k_beg * ldy + o0) * 4), sb0 = (unsigned)((k_beg * ldx + i0) * 4);
        auto fetch = [&](WxStageW& st, int step) {
            if (step >= n_full) return;
            if (!buf_ok) {
                wxw_load<true>(st, ga + (int64_t)step * 16 * ldy, gb + (int64_t)step * 16 * ldx, la, lb, ldy, ldx, 16, kg);
                return;
            }
            const unsigned sa = sa0 + (unsigned)step * 16u * ldyb, sb = sb0 + (unsigned)step * 16u * ldxb;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    st.a[t][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ra, va + 128u * t, sa + (unsigned)i * ldyb, 0));
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    st.b[t][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rb, vb + 128u * t, sb + (unsigned)i * ldxb, 0));
            }
        };
        fetch(s0, 0);
        fetch(s1, 1);
        fetch(s2, 2);
        for (int step = 0; step < n_full; step += 4) {
            fetch(s3, step + 3);
            __builtin_amdgcn_sched_barrier(0);
            consume(s0);
            __builtin_amdgcn_sched_barrier(0);
            fetch(s0, step + 4);
            __builtin_amdgcn_sched_barrier(0);
            if (step + 1 < n_full) consume(s1);
            __builtin_amdgcn_sched_barrier(0);
            fetch(s1, step + 5);
            __builtin_amdgcn_sched_barrier(0);
            if (step + 2 < n_full) consume(s2);
            __builtin_amdgcn_sched_barrier(0);
            fetch(s2, step + 6);
            __builtin_amdgcn_sched_barrier(0);
            if (step + 3 < n_full) consume(s3);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (tail) {
            wxw_load<false>(s0, ga + (int64_t)n_full * 16 * ldy, gb + (int64_t)n_full * 16 * ldx, la, lb, ldy, ldx, tail, kg);
            consume(s0);
        }
    }
    // accumulator register e of tile (ta, tb): row 32 ta + (e & 3) + 8 (e >> 2) + 4 kg, column 32 tb + m of the wavefront's tile
    float* __restrict__ out = en.slab + ((int64_t)chunk * O + o0) * I + i0;
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = 32 * ta + (e & 3) + 8 * (e >> 2) + 4 * kg, col = 32 * tb + m;
                out[(int64_t)row * I + col] = en.alpha * acc[ta][tb][e];
            }
}

// chunks of K: enough workgroups to fill the chip, rows per chunk a whole number of register stages
inline void plan(int64_t K, int O, int I, int* chunks, int64_t* k_chunk) {
    const int tiles = (O / 64) * (I / 64);
    int64_t c = 256 / tiles;   // one workgroup per CU (two per CU measured slower: 13.4 vs 12.0 us)
    if (c < 1) c = 1;
    int64_t kc = (K + c - 1) / c;
    kc = (kc + 16 * AHEAD - 1) / (16 * AHEAD) * (16 * AHEAD);   // whole register stages per wavefront
    if (kc < 64) kc = 64;
    c = (K + kc - 1) / kc;
    *chunks = (int)(c < 1 ? 1 : c);
    *k_chunk = kc;
}

}  // namespace

extern "C" size_t hg_wgrad_workspace_bytes(int64_t K, int32_t O, int32_t I) {
    if (K <= 0 || O <= 0 || I <= 0 || (O & 63) || (I & 63)) return 0;
    int chunks;
    int64_t kc;
    plan(K, O, I, &chunks, &kc);
    return chunks > 1 ? (size_t)chunks * (size_t)O * (size_t)I * sizeof(float) : 16;
}

// ---- skinny weight gradient: dw[O x J] (+)= dy^T [O x K] . x [K x J] with J <= 16 ----------------------------------------------
// (the m_i block of the EGNN node MLP's first Linear, egnn_layer.py:180-187: [512 x 16] from ~4.7 k rows.  The library ran it
// as a 1-workgroup split-K product + a post-reduction, 19 us of a 1.1 ms step for 75 MFLOP.)  A workgroup owns 64 columns of dy
// and a chunk of SK_ROWS rows; a lane owns one column and keeps the J sums in registers; the rows of x are the same for every
// lane (scalar loads); the four wavefronts' sums meet in LDS in wavefront order, the chunks' partial results in a slab that the
// step's (deferred) fixed-order reduction adds into dw: no atomics, bitwise reproducible.
namespace {
constexpr int SK_ROWS = 64, SK_J = 16;     // (256-row chunks: 152 workgroups at the BASELINE batch, 16 us; 64: 600 workgroups)
__global__ void __launch_bounds__(256) k_wgrad_skinny(const float* __restrict__ dy, int64_t ld_dy, const float* __restrict__ x, int64_t ld_x,
                                                      int64_t K, int O, int J, float alpha, float* __restrict__ slab) {
    __shared__ float s_part[3][SK_J][64];
    __shared__ __attribute__((aligned(16))) float s_x[SK_ROWS][SK_J];     // the chunk's rows of x (zero beyond J / K): read as broadcasts
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int o = blockIdx.x * 64 + lane;
    const int64_t r0 = (int64_t)blockIdx.y * SK_ROWS;
    const int64_t r1 = r0 + SK_ROWS < K ? r0 + SK_ROWS : K;
    const int nr = (int)(r1 - r0);
    for (int i = threadIdx.x; i < SK_ROWS * SK_J; i += 256) {
        const int r = i / SK_J, j = i - r * SK_J;
        s_x[r][j] = (r < nr && j < J) ? x[(r0 + r) * ld_x + j] : 0.f;
    }
    float acc[SK_J];
#pragma unroll
    for (int j = 0; j < SK_J; ++j) acc[j] = 0.f;
    constexpr int U = 8;                                   // rows in flight per wavefront
    float a[U];
    auto fetch = [&](int rl) {
#pragma unroll
        for (int u = 0; u < U; ++u) a[u] = (rl + u < nr && o < O) ? dy[(r0 + rl + u) * ld_dy + o] : 0.f;
    };
    fetch(wave * U);
    __syncthreads();
    for (int rl = wave * U; rl < nr; rl += 4 * U) {
        float c[U];
#pragma unroll
        for (int u = 0; u < U; ++u) c[u] = a[u];
        fetch(rl + 4 * U);                                 // the next eight rows, behind this round's arithmetic
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (rl + u < nr) {                             // wavefront-uniform
#pragma unroll
                for (int j4 = 0; j4 < SK_J / 4; ++j4) {
                    const float4 xv = *reinterpret_cast<const float4*>(&s_x[rl + u][4 * j4]);
                    acc[4 * j4 + 0] = fmaf(c[u], xv.x, acc[4 * j4 + 0]); acc[4 * j4 + 1] = fmaf(c[u], xv.y, acc[4 * j4 + 1]);
                    acc[4 * j4 + 2] = fmaf(c[u], xv.z, acc[4 * j4 + 2]); acc[4 * j4 + 3] = fmaf(c[u], xv.w, acc[4 * j4 + 3]);
                }
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int j = 0; j < SK_J; ++j) s_part[wave - 1][j][lane] = acc[j];
    }
    __syncthreads();
    if (wave == 0 && o < O) {
        float* out = slab + ((int64_t)blockIdx.y * O + o) * J;
        for (int j = 0; j < J; ++j)
            out[j] = alpha * (((acc[j] + s_part[0][j][lane]) + s_part[1][j][lane]) + s_part[2][j][lane]);
    }
}
}  // namespace

extern "C" size_t hg_wgrad_skinny_workspace_bytes(int64_t K, int32_t O, int32_t J) {
    if (K <= 0 || O <= 0 || J <= 0) return 0;
    return (size_t)((K + SK_ROWS - 1) / SK_ROWS) * (size_t)O * (size_t)J * sizeof(float);
}

extern "C" int hg_wgrad_skinny_f32(const float* dy, int64_t ld_dy, const float* x, int64_t ld_x, int64_t K, int32_t O, int32_t J,
                                   float alpha, float* dw, int64_t ldw, int32_t accumulate, void* workspace, size_t workspace_bytes,
                                   void* stream_) {
    if (K < 0 || O <= 0 || J <= 0 || J > SK_J || !dw || ldw < J || ld_dy < O || ld_x < J) return EQH_ERR_ARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (K == 0) return EQH_OK;                             // (nothing to add; a non-accumulating caller clears dw itself)
    if (!dy || !x || !workspace) return EQH_ERR_ARG;
    if (workspace_bytes < hg_wgrad_skinny_workspace_bytes(K, O, J)) return EQH_ERR_ARG;
    const int chunks = (int)((K + SK_ROWS - 1) / SK_ROWS);
    if (chunks > 65535) return EQH_ERR_RANGE;
    float* slab = static_cast<float*>(workspace);
    hipLaunchKernelGGL(k_wgrad_skinny, dim3((O + 63) / 64, chunks), dim3(256), 0, stream, dy, ld_dy, x, ld_x, K, (int)O, (int)J, alpha, slab);
    EQH_CHECK_LAUNCH();
    return eqh_reduce_slabs2d_async(slab, chunks, O, J, dw, ldw, accumulate, stream);
}

extern "C" int hg_wgrad_f32(const float* dy, const float* x, int64_t K, int32_t O, int32_t I, float alpha,
                            float* dw, int64_t ldw, int32_t accumulate, void* workspace, size_t workspace_bytes,
                            void* stream_) {
    if (K < 0 || O <= 0 || I <= 0 || !dw || ldw < I) return EQH_ERR_ARG;
    if ((O & 63) || (I & 63) || (ldw & 3)) return EQH_ERR_ALIGN;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (K == 0) {
        if (accumulate) return EQH_OK;
        for (int o = 0; o < O; ++o)
            if (eqh_zero_async(dw + (int64_t)o * ldw, I, stream)) return EQH_ERR_LAUNCH;
        return EQH_OK;
    }
    if (!dy || !x) return EQH_ERR_ARG;
    if (!eqh_aligned16(dy) || !eqh_aligned16(x) || !eqh_aligned16(dw)) return EQH_ERR_ALIGN;
    int chunks;
    int64_t kc;
    plan(K, O, I, &chunks, &kc);
    const int tiles_i = I / 64, tiles = (O / 64) * tiles_i;
    if (chunks == 1) {
        hipLaunchKernelGGL(k_wgrad, dim3(tiles), dim3(THREADS), 0, stream, dy, x, K, (int)O, (int)I, dw, ldw, tiles_i,
                           tiles, kc, 1, (int)accumulate, alpha);
        EQH_CHECK_LAUNCH();
        return EQH_OK;
    }
    if (!workspace || !eqh_aligned16(workspace)) return EQH_ERR_ARG;
    if (workspace_bytes < hg_wgrad_workspace_bytes(K, O, I)) return EQH_ERR_ARG;
    float* slab = static_cast<float*>(workspace);
    hipLaunchKernelGGL(k_wgrad, dim3(tiles * chunks), dim3(THREADS), 0, stream, dy, x, K, (int)O, (int)I, slab,
                       (int64_t)I, tiles_i, tiles, kc, 0, 0, alpha);
    EQH_CHECK_LAUNCH();
    return eqh_reduce_slabs2d_async(slab, chunks, O, I, dw, ldw, accumulate, stream);
}

/* count products of one shape in one launch: dw[i] (+)= alpha[i] * dy[i].T @ x[i].  Entries that share a
 * destination must be adjacent (they are summed in array order by one reduction); the workspace holds
 * count * (chunks of K) slabs of O x I floats (hg_wgrad_batch_workspace_bytes).  ld_dy / ld_x (NULL: O / I): row strides of the operands, so that column blocks of
 * wider matrices (the two halves of a [rows x 2 C] hidden activation) join a batch of [C x C] products. */
// A launch takes at most WG_MAX_BATCH products; a longer list is dealt out evenly over the launches it needs.
static inline int wg_per_launch(int32_t count) {
    const int n_launch = (count + WG_MAX_BATCH - 1) / WG_MAX_BATCH;
    return n_launch > 0 ? (count + n_launch - 1) / n_launch : 1;
}
// The 128 x 256 workgroup tiles (one wavefront per SIMD) where a launch fills the chip with at most five chunks of K per product
// (measured: mhnnm's launch 174.6 -> 135.3 us; the 21 products of the BASELINE egnn_equihnns step would need 6 chunks of 49
// steps each and gain nothing, 100.0 vs 100.8 us: they keep the 128 x 128 tiles).  EQH_WGRAD_WIDE=0 / =1 force either.
static inline bool wg_wide(int32_t per_launch, int32_t O, int32_t I) {
    static const int mode = [] { const char* e = std::getenv("EQH_WGRAD_WIDE"); return e ? (e[0] == '0' ? 0 : 1) : -1; }();
    if (mode == 0 || (O % 128) || (I % 256)) return false;
    if (mode == 1) return true;
    return (int64_t)(O / 128) * (I / 256) * per_launch * 5 >= 256;
}
// chunks of K per product of a batched launch: three where that fills the chip (21 products x four 128 x 128 tiles), more for
// a few large products (the [2176 x 4.7 k] . [4.7 k x 256] gradient of the EGNN's first edge Linear alone: 34 tiles x 8)
static inline int wg_batch_splits(int32_t count, int32_t O, int32_t I) {
    const int per = wg_per_launch(count);
    if (wg_wide(per, O, I)) {        // one workgroup per CU: as many chunks of K as fit 256 workgroups
        const int64_t t = (int64_t)(O / 128) * (I / 256) * per;
        int64_t s = 256 / (t > 0 ? t : 1);
        return (int)(s < 1 ? 1 : (s > 8 ? 8 : s));
    }
    const int64_t tiles = (int64_t)((O + 127) / 128) * ((I + 127) / 128) * per;
    if (tiles * 3 >= 200) return 3;
    int64_t s = (256 + tiles - 1) / tiles;
    return (int)(s < 3 ? 3 : (s > 8 ? 8 : s));
}

extern "C" size_t hg_wgrad_batch_workspace_bytes(int32_t count, int32_t O, int32_t I) {
    if (count <= 0 || O <= 0 || I <= 0) return 0;
    return (size_t)count * wg_batch_splits(count, O, I) * (size_t)O * (size_t)I * sizeof(float);
}

extern "C" int hg_wgrad_batch_f32(int32_t count, const float* const* dy, const float* const* x, const int64_t* K,
                                  int32_t O, int32_t I, const float* alpha, float* const* dw, const int64_t* ldw,
                                  int32_t accumulate, void* workspace, size_t workspace_bytes, void* stream_,
                                  const int64_t* ld_dy, const int64_t* ld_x) {
    if (count < 0 || O <= 0 || I <= 0) return EQH_ERR_ARG;
    if (count == 0) return EQH_OK;
    if (!dy || !x || !K || !alpha || !dw || !ldw || !workspace) return EQH_ERR_ARG;
    if ((O & 63) || (I & 63) || !eqh_aligned16(workspace)) return EQH_ERR_ALIGN;
    if (workspace_bytes < hg_wgrad_batch_workspace_bytes(count, O, I)) return EQH_ERR_ARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int SPLITS = wg_batch_splits(count, O, I);  // (three for the conv-sized batches: measured flat between 2 and 6)
    const int tiles_i = I / 64, tiles = (O / 64) * tiles_i;
    const size_t slab_elems = (size_t)O * I;
    float* ws = static_cast<float*>(workspace);
    const int per = wg_per_launch(count);
    static const bool debug = std::getenv("EQH_WGRAD_DEBUG") != nullptr;
    if (debug) fprintf(stderr, "wgrad batch: %d products of %d x %d, %d per launch, %s tiles, %d chunks of K\n", (int)count, (int)O, (int)I,
                       per, wg_wide(per, O, I) ? "128 x 256" : "128 x 128", SPLITS);
    for (int i0 = 0; i0 < count; i0 += per) {
        WgradBatch b;
        const int m = (count - i0 < per) ? count - i0 : per;
        for (int i = 0; i < m; ++i) {
            const int j = i0 + i;
            if (K[j] <= 0 || !dy[j] || !x[j] || !dw[j] || ldw[j] < I || (ldw[j] & 3)) return EQH_ERR_ARG;
            if (!eqh_aligned16(dy[j]) || !eqh_aligned16(x[j]) || !eqh_aligned16(dw[j])) return EQH_ERR_ALIGN;
            const int64_t ldy = ld_dy ? ld_dy[j] : O, ldx = ld_x ? ld_x[j] : I;
            if (ldy < O || ldx < I || (ldy & 3) || (ldx & 3)) return EQH_ERR_ALIGN;
            b.e[i] = WgradEntry{dy[j], x[j], ws + (size_t)j * SPLITS * slab_elems, K[j], ldy, ldx, alpha[j], 0};
        }
        // round 5: the bf16 x 3 kernel (128 x 128 workgroup tiles); EQH_WGRAD_F32=1 keeps the fp32-MFMA kernel for same-box A/B runs
        static const bool use_f32 = [] { const char* e = std::getenv("EQH_WGRAD_F32"); return e && e[0] == '1'; }();
        // (EQH_WGRAD_GLOBAL=1: 64-bit global loads instead of buffer loads, for same-box A/B runs)
        static const int use_buf = [] { const char* e = std::getenv("EQH_WGRAD_GLOBAL"); return (e && e[0] == '1') ? 0 : 1; }();
        if (use_f32) {
            hipLaunchKernelGGL(k_wgrad_batch, dim3(tiles * SPLITS, m), dim3(THREADS), 0, stream, b, (int)O, (int)I, tiles_i,
                               tiles, SPLITS);
        } else if (wg_wide(per, O, I)) {
            const int t_i = I / 256, t_all = (O / 128) * t_i;
            const int groups = m * SPLITS;
            hipLaunchKernelGGL(k_wgrad_batch_x3w, dim3(8 * ((groups + 7) / 8) * t_all), dim3(256), 0, stream, b, m, (int)O, (int)I,
                               t_i, t_all, SPLITS, use_buf);
        } else {
            const int t_i = (I + 127) / 128, t_all = ((O + 127) / 128) * t_i;
            const int groups = m * SPLITS;
            hipLaunchKernelGGL(k_wgrad_batch_x3, dim3(8 * ((groups + 7) / 8) * t_all), dim3(WX_THREADS), 0, stream, b, m, (int)O,
                               (int)I, t_i, t_all, SPLITS, use_buf);
        }
        EQH_CHECK_LAUNCH();
    }
    // one reduction per destination: runs of equal (dw, ldw) are adjacent, their slabs contiguous
    for (int j = 0; j < count;) {
        int j1 = j + 1;
        while (j1 < count && dw[j1] == dw[j] && ldw[j1] == ldw[j]) ++j1;
        const int rc = eqh_reduce_slabs2d_async(ws + (size_t)j * SPLITS * slab_elems, (j1 - j) * SPLITS, O, I, dw[j],
                                                ldw[j], accumulate, stream);
        if (rc) return rc;
        j = j1;
    }
    return EQH_OK;
}
