// CSR-grouped small GEMMs: every entry (edge) e multiplies its feature row z[e, :Kd] with the
// [Kd, L] matrix OWNED BY ITS CSR ROW (a node):   out[e, :] (+)= z[e, :] · w[row(e)].
//
// This is how the Equiformer's radial tensor product is evaluated without ever forming the
// per-edge radial weights R[e, lo, li] (equiformer_layer.py:451-479 builds them with a
// Linear(64 -> lo*li) per edge — 262 KB per edge at C=256 — and contracts them at :383):
//   out[e, lo] = sum_li R[e, lo, li] x[e, li],  R[e] = reshape(W3 z_e + b3),  x[e] = xj[j] + xi[i]
//             = sum_k z_e[k] (P[j, k, lo] + Q[i, k, lo]) + ...,   P[n] = sum_li W3[lo, li, k] xj[n, li]
// P and Q are node-level library GEMMs; what is left per edge is z_e · P[sender] (+ z_e ·
// Q[receiver]): a [deg x Kd] x [Kd x L] product per node, grouped by the sender (transposed
// neighbour CSR) or by the receiver.  fp32 MFMA 16x16x4, 16 entries per tile on the M axis; the
// node matrix is read once per 16 entries instead of once per edge.
//
// Backward: dz[e] = dout[e] · w[row]ᵀ and dw[row] = sum_e z[e]ᵀ ⊗ dout[e]; every dw[row] is written
// by exactly one wavefront (no atomics, reproducible).
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int THREADS = 256;
constexpr int WAVES = THREADS / 64;

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ int entry_at(const int* __restrict__ perm, int pos, int end) {
    if (pos >= end) return -1;
    return perm ? perm[pos] : pos;
}

// grid.x strides rows (one wavefront per row), grid.y splits the L/16 column tiles
__global__ void __launch_bounds__(THREADS)
k_rowgemm_fwd(const float* __restrict__ z, const float* __restrict__ w, const int* __restrict__ rowptr,
              const int* __restrict__ perm, int R, int Kd, int L, float* __restrict__ out, int accumulate) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r_ = lane & 15, q = lane >> 4;
    const int ctiles = L >> 4, ksteps = Kd >> 4;
    for (int row = blockIdx.x * WAVES + wave; row < R; row += gridDim.x * WAVES) {
        const int beg = rowptr[row], end = rowptr[row + 1];
        const float* __restrict__ wr = w + (int64_t)row * Kd * L;
        for (int g0 = beg; g0 < end; g0 += 16) {
            const int e_r = entry_at(perm, g0 + r_, end);
            int e_g[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) e_g[g] = __shfl(e_r, 4 * q + g, 64);
            for (int ct = blockIdx.y; ct < ctiles; ct += gridDim.y) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                for (int t = 0; t < ksteps; ++t) {
                    float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (e_r >= 0) z4 = *reinterpret_cast<const float4*>(z + (int64_t)e_r * Kd + 16 * t + 4 * q);
                    const float* __restrict__ wb = wr + (int64_t)(16 * t + 4 * q) * L + ct * 16 + r_;
                    acc = mfma16(z4.x, wb[0], acc);
                    acc = mfma16(z4.y, wb[L], acc);
                    acc = mfma16(z4.z, wb[2 * L], acc);
                    acc = mfma16(z4.w, wb[3 * L], acc);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (e_g[g] >= 0) {
                        float* o = out + (int64_t)e_g[g] * L + ct * 16 + r_;
                        *o = accumulate ? (*o + acc[g]) : acc[g];
                    }
                }
            }
        }
    }
}

// dz[e, k] (+)= sum_l dout[e, l] * w[row][k][l];   grid.y splits the Kd/16 k-tiles
__global__ void __launch_bounds__(THREADS)
k_rowgemm_bwd_z(const float* __restrict__ dout, const float* __restrict__ w,
                const int* __restrict__ rowptr, const int* __restrict__ perm, int R, int Kd, int L,
                float* __restrict__ dz, int accumulate) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r_ = lane & 15, q = lane >> 4;
    const int ktiles = Kd >> 4, lsteps = L >> 4;
    for (int row = blockIdx.x * WAVES + wave; row < R; row += gridDim.x * WAVES) {
        const int beg = rowptr[row], end = rowptr[row + 1];
        const float* __restrict__ wr = w + (int64_t)row * Kd * L;
        for (int g0 = beg; g0 < end; g0 += 16) {
            const int e_r = entry_at(perm, g0 + r_, end);
            int e_g[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) e_g[g] = __shfl(e_r, 4 * q + g, 64);
            for (int kt = blockIdx.y; kt < ktiles; kt += gridDim.y) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                const float* __restrict__ wk = wr + (int64_t)(kt * 16 + r_) * L + 4 * q;
                for (int t = 0; t < lsteps; ++t) {
                    float4 d4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (e_r >= 0) d4 = *reinterpret_cast<const float4*>(dout + (int64_t)e_r * L + 16 * t + 4 * q);
                    const float4 w4 = *reinterpret_cast<const float4*>(wk + 16 * t);
                    acc = mfma16(d4.x, w4.x, acc);
                    acc = mfma16(d4.y, w4.y, acc);
                    acc = mfma16(d4.z, w4.z, acc);
                    acc = mfma16(d4.w, w4.w, acc);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (e_g[g] >= 0) {
                        float* o = dz + (int64_t)e_g[g] * Kd + kt * 16 + r_;
                        *o = accumulate ? (*o + acc[g]) : acc[g];
                    }
                }
            }
        }
    }
}

// dw[row][k][l] = sum_{e in row} z[e, k] * dout[e, l];   grid.y splits the (Kd/16)*(L/16) tiles
__global__ void __launch_bounds__(THREADS)
k_rowgemm_bwd_w(const float* __restrict__ z, const float* __restrict__ dout,
                const int* __restrict__ rowptr, const int* __restrict__ perm, int R, int Kd, int L,
                float* __restrict__ dw) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r_ = lane & 15, q = lane >> 4;
    const int ktiles = Kd >> 4, ltiles = L >> 4;
    for (int row = blockIdx.x * WAVES + wave; row < R; row += gridDim.x * WAVES) {
        const int beg = rowptr[row], end = rowptr[row + 1];
        float* __restrict__ dwr = dw + (int64_t)row * Kd * L;
        for (int tile = blockIdx.y; tile < ktiles * ltiles; tile += gridDim.y) {
            const int kt = tile / ltiles, lt = tile - kt * ltiles;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int g0 = beg; g0 < end; g0 += 16) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int e = entry_at(perm, g0 + 4 * q + c, end);
                    float a = 0.f, b = 0.f;
                    if (e >= 0) {
                        a = z[(int64_t)e * Kd + kt * 16 + r_];
                        b = dout[(int64_t)e * L + lt * 16 + r_];
                    }
                    acc = mfma16(a, b, acc);
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) dwr[(int64_t)(kt * 16 + 4 * q + g) * L + lt * 16 + r_] = acc[g];
        }
    }
}

// ---- workgroup-per-row variants: the row's [Kd, L] matrix is staged into LDS ONCE with coalesced float4 loads
// (padded stride L + 4: the four k-groups of an MFMA operand land in different banks) and shared by the four
// wavefronts, which split the output tiles.  The first versions above read it straight from global memory, one
// 64-byte segment per operand fetch and again for every column tile: 72 us per call at the BASELINE batch
// (2.4 k nodes x 64 KB) against the ~35 us of reading the matrices once.
__device__ __forceinline__ void stage_matrix(const float* __restrict__ wr, float* __restrict__ s_w, int Kd, int L) {
    const int ld = L + 4;
    const int n4 = (Kd * L) >> 2;
    // eight float4 requests per thread in flight before the first LDS store (the loop is the kernel's only
    // long-latency part: one request at a time left the workgroup waiting a memory round trip per 4 KB)
    for (int f0 = threadIdx.x; f0 < n4; f0 += 8 * THREADS) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int f = f0 + u * THREADS;
            v[u] = f < n4 ? *reinterpret_cast<const float4*>(wr + ((int64_t)f << 2)) : f4_zero();
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int f = f0 + u * THREADS;
            if (f < n4) {
                const int i = f << 2;
                const int row = i / L, col = i - row * L;
                *reinterpret_cast<float4*>(s_w + row * ld + col) = v[u];
            }
        }
    }
}

template <int KSTEPS>   // Kd / 16
__global__ void __launch_bounds__(THREADS)
k_rowgemm_fwd_lds(const float* __restrict__ z, const float* __restrict__ w, const int* __restrict__ rowptr,
                  const int* __restrict__ perm, int R, int L, float* __restrict__ out, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) float s_w[];
    constexpr int Kd = KSTEPS * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r_ = lane & 15, q = lane >> 4;
    const int ctiles = L >> 4, ld = L + 4;
    for (int row = blockIdx.x; row < R; row += gridDim.x) {
        const int beg = rowptr[row], end = rowptr[row + 1];
        if (beg == end) continue;                     // block-uniform
        __syncthreads();                              // the previous row's readers are done
        stage_matrix(w + (int64_t)row * Kd * L, s_w, Kd, L);
        __syncthreads();
        for (int g0 = beg; g0 < end; g0 += 16) {
            const int e_r = entry_at(perm, g0 + r_, end);
            int e_g[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) e_g[g] = __shfl(e_r, 4 * q + g, 64);
            float4 z4[KSTEPS];
#pragma unroll
            for (int t = 0; t < KSTEPS; ++t)
                z4[t] = e_r >= 0 ? *reinterpret_cast<const float4*>(z + (int64_t)e_r * Kd + 16 * t + 4 * q) : f4_zero();
            for (int ct = wave; ct < ctiles; ct += WAVES) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                const float* __restrict__ sb = s_w + (4 * q) * ld + ct * 16 + r_;
#pragma unroll
                for (int t = 0; t < KSTEPS; ++t) {
                    acc = mfma16(z4[t].x, sb[(16 * t + 0) * ld], acc);
                    acc = mfma16(z4[t].y, sb[(16 * t + 1) * ld], acc);
                    acc = mfma16(z4[t].z, sb[(16 * t + 2) * ld], acc);
                    acc = mfma16(z4[t].w, sb[(16 * t + 3) * ld], acc);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (e_g[g] >= 0) {
                        float* o = out + (int64_t)e_g[g] * L + ct * 16 + r_;
                        *o = accumulate ? (*o + acc[g]) : acc[g];
                    }
                }
            }
        }
    }
}

template <int KSTEPS>
__global__ void __launch_bounds__(THREADS)
k_rowgemm_bwd_z_lds(const float* __restrict__ dout, const float* __restrict__ w, const int* __restrict__ rowptr,
                    const int* __restrict__ perm, int R, int L, float* __restrict__ dz, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) float s_w[];
    constexpr int Kd = KSTEPS * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r_ = lane & 15, q = lane >> 4;
    const int lsteps = L >> 4, ld = L + 4;
    for (int row = blockIdx.x; row < R; row += gridDim.x) {
        const int beg = rowptr[row], end = rowptr[row + 1];
        if (beg == end) continue;
        __syncthreads();
        stage_matrix(w + (int64_t)row * Kd * L, s_w, Kd, L);
        __syncthreads();
        for (int g0 = beg; g0 < end; g0 += 16) {
            const int e_r = entry_at(perm, g0 + r_, end);
            int e_g[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) e_g[g] = __shfl(e_r, 4 * q + g, 64);
            for (int kt = wave; kt < KSTEPS; kt += WAVES) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                const float* __restrict__ sk = s_w + (kt * 16 + r_) * ld + 4 * q;
                const float* __restrict__ dr = dout + (int64_t)(e_r >= 0 ? e_r : 0) * L + 4 * q;
                for (int t = 0; t < lsteps; ++t) {
                    float4 d4 = *reinterpret_cast<const float4*>(dr + 16 * t);
                    if (e_r < 0) d4 = f4_zero();
                    const float4 w4 = *reinterpret_cast<const float4*>(sk + 16 * t);
                    acc = mfma16(d4.x, w4.x, acc);
                    acc = mfma16(d4.y, w4.y, acc);
                    acc = mfma16(d4.z, w4.z, acc);
                    acc = mfma16(d4.w, w4.w, acc);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (e_g[g] >= 0) {
                        float* o = dz + (int64_t)e_g[g] * Kd + kt * 16 + r_;
                        *o = accumulate ? (*o + acc[g]) : acc[g];
                    }
                }
            }
        }
    }
}

// dw[row] = sum_{e in row} z[e]^T (x) dout[e]: the rows of z and dout of up to 32 entries are staged into LDS
// (coalesced float4 loads; the first version re-read 64-byte segments of them from global memory for each of the
// (Kd/16) x (L/16) output tiles), every wavefront keeps its share of the output tiles in registers across chunks.
constexpr int BW_CH = 32;
template <int KSTEPS, int LTILES>
__global__ void __launch_bounds__(THREADS)
k_rowgemm_bwd_w_lds(const float* __restrict__ z, const float* __restrict__ dout, const int* __restrict__ rowptr,
                    const int* __restrict__ perm, int R, float* __restrict__ dw) {
    constexpr int Kd = KSTEPS * 16, L = LTILES * 16;
    constexpr int LZ = Kd + 16, LD = L + 16;                 // row strides = 16 mod 64 banks
    constexpr int TPW = KSTEPS * LTILES / WAVES;             // output tiles per wavefront
    static_assert(KSTEPS * LTILES % WAVES == 0, "tiles split evenly");
    __shared__ __attribute__((aligned(16))) float s_z[BW_CH * LZ];
    __shared__ __attribute__((aligned(16))) float s_d[BW_CH * LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r_ = lane & 15, q = lane >> 4;
    for (int row = blockIdx.x; row < R; row += gridDim.x) {
        const int beg = rowptr[row], end = rowptr[row + 1];
        f32x4 acc[TPW];
#pragma unroll
        for (int i = 0; i < TPW; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int c0 = beg; c0 < end; c0 += BW_CH) {
            __syncthreads();
            constexpr int Z4 = Kd / 4, D4 = L / 4;
            for (int f = threadIdx.x; f < BW_CH * (Z4 + D4); f += THREADS) {
                const int ent = f / (Z4 + D4), j = f - ent * (Z4 + D4);
                const int e = entry_at(perm, c0 + ent, end);
                float4 v = f4_zero();
                if (j < Z4) {
                    if (e >= 0) v = *reinterpret_cast<const float4*>(z + (int64_t)e * Kd + 4 * j);
                    *reinterpret_cast<float4*>(s_z + ent * LZ + 4 * j) = v;
                } else {
                    if (e >= 0) v = *reinterpret_cast<const float4*>(dout + (int64_t)e * L + 4 * (j - Z4));
                    *reinterpret_cast<float4*>(s_d + ent * LD + 4 * (j - Z4)) = v;
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                const int tile = wave * TPW + i;
                const int kt = tile / LTILES, lt = tile - kt * LTILES;
#pragma unroll
                for (int s4 = 0; s4 < BW_CH / 4; ++s4)
                    acc[i] = mfma16(s_z[(4 * s4 + q) * LZ + kt * 16 + r_], s_d[(4 * s4 + q) * LD + lt * 16 + r_], acc[i]);
            }
        }
        float* __restrict__ dwr = dw + (int64_t)row * Kd * L;
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int tile = wave * TPW + i;
            const int kt = tile / LTILES, lt = tile - kt * LTILES;
#pragma unroll
            for (int g = 0; g < 4; ++g) dwr[(int64_t)(kt * 16 + 4 * q + g) * L + lt * 16 + r_] = acc[i][g];
        }
    }
}

// ---- streaming variants (round 5): ONE WAVEFRONT per row, no LDS and no barrier.  The row's [Kd, L] matrix is 48-64 KB that is
// used once per 16 entries, so these kernels are a stream over 115-160 MB with a few MFMAs riding along; the workgroup-per-row
// kernels above put a load -> barrier -> multiply -> barrier chain around every matrix and reached 2 TB/s.  Here every
// wavefront keeps a ring of RING 16-byte loads in flight straight into the registers the MFMA reads:
//   * B operand by float4: lane (r, q) of a 16x16x4 MFMA supplies B[k = q][n = r]; a float4 at columns 64 u + 4 r .. + 3 of
//     matrix row k(q) is that lane's value for FOUR column tiles (tile (u, j) holds the columns 64 u + 4 r + j), so one
//     instruction reads 4 matrix rows x 256 contiguous bytes and feeds 4 MFMAs; the output tiles come back as float4 over j.
//   * the k index of MFMA step (t, i) is 16 t + 4 q + i, so that the A operand z[e][16 t + 4 q ..] is a float4 as well.
// L is a multiple of 64 (LU = L / 64), Kd = 16 KT.
#ifndef ROW_STREAM_WAVES
#define ROW_STREAM_WAVES 1      // wavefronts per workgroup of the streaming kernels: ONE, so that the ~10 rows per CU are dealt out
#endif                          // one by one (four-wavefront workgroups: 3 against 2 of them per CU, the launch as slow as the 3)
constexpr int SW = ROW_STREAM_WAVES, STHREADS = 64 * SW;

template <int KT, int LU>
__global__ void __launch_bounds__(STHREADS) __attribute__((amdgpu_waves_per_eu(3, 3)))
k_rowgemm_fwd_stream(const float* __restrict__ z, const float* __restrict__ w, const int* __restrict__ rowptr,
                     const int* __restrict__ perm, int R, int Lrt, float* __restrict__ out, int accumulate,
                     const float* __restrict__ rowbias, const float* __restrict__ coef, int MB, int zfac) {
    // a ROUND is 16 loads = 16 KB of the matrix per wavefront: TB steps of 16 k-values (one for L = 256, four for L = 64);
    // the loop over rounds is not unrolled, so the loads of round n + 1 are what is in flight while round n multiplies
    // L: the matrix' row pitch = its width; 256 for LU = 4, any multiple of 4 up to 64 for LU = 1 (52 columns of the attention's
    // pairs: lanes whose four columns lie beyond L load zeros and store nothing)
    constexpr int Kd = KT * 16, TB = LU == 4 ? 1 : 4, PER = TB * 4 * LU, NR = KT / TB;
    static_assert(KT % TB == 0 && PER == 16, "rounds of 16 loads");
    const int L = LU == 4 ? 256 : Lrt;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r_ = lane & 15, q = lane >> 4;
    const bool act = LU == 4 || 4 * r_ < L;
    const int row = blockIdx.x * SW + wave;
    if (row >= R) return;
    const int beg = rowptr[row], end = rowptr[row + 1];
    const float* wr = w + (int64_t)row * Kd * L + (int64_t)(4 * q) * L + 4 * r_;
    // load n of round rd -> (tt, i, u): matrix row 16 (rd TB + tt) + 4 q + i, columns 64 u + 4 r ..
    auto wload = [&](int rd, int n) {      // (one base address per round; the 16 loads differ by constants)
        const int tt = n / (4 * LU), i = (n / LU) & 3, u = n % LU;
        const float* __restrict__ wb = wr + (int64_t)rd * (16 * TB * L);
        return act ? *reinterpret_cast<const float4*>(wb + (16 * tt + i) * L + 64 * u) : f4_zero();
    };
#ifdef ROW_ROTATE
    const int rot = row % NR;
#else
    constexpr int rot = 0;
#endif
    auto rnd = [&](int rd) { const int x = rd + rot; return x >= NR ? x - NR : x; };
    for (int g0 = beg; g0 < end; g0 += 16) {
        // (the matrix is the same for every group of the row: without this the first round's loads are hoisted out of the
        // loop and kept -- spilled -- across it)
        asm volatile("" : "+v"(wr));
        const int e_r = entry_at(perm, g0 + r_, end);
        int e_g[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) e_g[g] = __shfl(e_r, 4 * q + g, 64);
        // zfac (LU = 1: a round is 64 k-values): z is [E, 64] and column (m, k) of the A operand is coef[e, m] z[e, k] -- the
        // (1 -> 0) pair's r_hat (x) z, equiformer_layer.py:376-383, never formed
        const float* __restrict__ zr = z + (int64_t)(e_r >= 0 ? e_r : 0) * (zfac ? 64 : Kd) + 4 * q;
        const float* __restrict__ cfr = coef + (int64_t)(e_r >= 0 ? e_r : 0) * MB;
        auto zload = [&](int rd, int tt) {
            if (e_r < 0) return f4_zero();
            if (LU == 1 && zfac) {
                float4 v = *reinterpret_cast<const float4*>(zr + 16 * tt);
                const float cf = cfr[rd];
                v.x *= cf; v.y *= cf; v.z *= cf; v.w *= cf;
                return v;
            }
            return *reinterpret_cast<const float4*>(zr + 16 * (rd * TB + tt));
        };
        float4 ring[PER], zc[TB];
#pragma unroll
        for (int n = 0; n < PER; ++n) ring[n] = wload(rnd(0), n);
#pragma unroll
        for (int tt = 0; tt < TB; ++tt) zc[tt] = zload(rnd(0), tt);
        f32x4 acc[4 * LU];
#pragma unroll
        for (int c = 0; c < 4 * LU; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int rd = 0; rd < NR; ++rd) {
            const bool more = rd + 1 < NR;
            float4 zn[TB];
#pragma unroll
            for (int tt = 0; tt < TB; ++tt)
                zn[tt] = more ? zload(rnd(rd + 1), tt) : f4_zero();
            // (the scheduling barriers keep the refill of a half behind its last use: hoisted, the 16 loads of the next round
            // need 64 registers of their own and the kernel spills)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int n = 8 * h; n < 8 * h + 8; ++n) {
                    const int tt = n / (4 * LU), i = (n / LU) & 3, u = n % LU;
                    const float4 b = ring[n];
                    const float a = i == 0 ? zc[tt].x : i == 1 ? zc[tt].y : i == 2 ? zc[tt].z : zc[tt].w;
                    acc[4 * u + 0] = mfma16(a, b.x, acc[4 * u + 0]);
                    acc[4 * u + 1] = mfma16(a, b.y, acc[4 * u + 1]);
                    acc[4 * u + 2] = mfma16(a, b.z, acc[4 * u + 2]);
                    acc[4 * u + 3] = mfma16(a, b.w, acc[4 * u + 3]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (more) {
#pragma unroll
                    for (int n = 8 * h; n < 8 * h + 8; ++n) ring[n] = wload(rnd(rd + 1), n);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int tt = 0; tt < TB; ++tt) zc[tt] = zn[tt];
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (e_g[g] < 0 || !act) continue;
            float* __restrict__ o = out + (int64_t)e_g[g] * L + 4 * r_;
#pragma unroll
            for (int u = 0; u < LU; ++u) {
                float4 v = make_float4(acc[4 * u][g], acc[4 * u + 1][g], acc[4 * u + 2][g], acc[4 * u + 3][g]);
                if (accumulate) {
                    const float4 p = *reinterpret_cast<const float4*>(o + 64 * u);
                    v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
                }
                if (rowbias) {          // + sum_m coef[e, m] rowbias[row, m, :]  (the row's bias block; L1 hits after the first entry)
                    for (int m = 0; m < MB; ++m) {
                        const float cf = coef ? coef[(int64_t)e_g[g] * MB + m] : 1.f;
                        const float4 rb = *reinterpret_cast<const float4*>(rowbias + ((int64_t)row * MB + m) * L + 64 * u + 4 * r_);
                        v.x = fmaf(cf, rb.x, v.x); v.y = fmaf(cf, rb.y, v.y); v.z = fmaf(cf, rb.z, v.z); v.w = fmaf(cf, rb.w, v.w);
                    }
                }
                *reinterpret_cast<float4*>(o + 64 * u) = v;
            }
            __builtin_amdgcn_sched_barrier(0);      // (one entry's read-add-write at a time: 16 registers, not 64)
        }
    }
}

// dz[e, k] (+)= sum_l dout[e, l] w[row][k][l]: the contraction runs along the matrix rows, so lane (r, q) reads a float4 of
// matrix row 16 kt + r at columns 16 t + 4 q .. (64-byte pieces of 16 rows per instruction; consecutive t complete the
// 128-byte lines while they are still in L1).  A round is TB tile rows (16 loads), stored as soon as it is summed.
template <int KT, int LU>
__global__ void __launch_bounds__(STHREADS) __attribute__((amdgpu_waves_per_eu(3, 3)))
k_rowgemm_bwd_z_stream(const float* __restrict__ dout, const float* __restrict__ w, const int* __restrict__ rowptr,
                       const int* __restrict__ perm, int R, int Lrt, float* __restrict__ dz, int accumulate,
                       const float* __restrict__ coef, int MB, int zfac) {
    constexpr int Kd = KT * 16, LT = 4 * LU, TB = LU == 4 ? 1 : 4, PER = TB * LT, NR = KT / TB;
    static_assert(KT % TB == 0 && PER == 16, "rounds of 16 loads");
    const int L = LU == 4 ? 256 : Lrt;        // (as in the forward kernel; here the columns are the contraction index)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r_ = lane & 15, q = lane >> 4;
    const int row = blockIdx.x * SW + wave;
    if (row >= R) return;
    const int beg = rowptr[row], end = rowptr[row + 1];
    const float* wr = w + (int64_t)row * Kd * L + (int64_t)r_ * L + 4 * q;
    auto wload = [&](int rd, int n) {       // n -> (kk, t)
        const int kk = n / LT, t = n % LT;
        const float* __restrict__ wb = wr + (int64_t)rd * (16 * TB * L);
        return (LU == 4 || 16 * t + 4 * q < L) ? *reinterpret_cast<const float4*>(wb + 16 * kk * L + 16 * t) : f4_zero();
    };
    for (int g0 = beg; g0 < end; g0 += 16) {
        asm volatile("" : "+v"(wr));        // (as in the forward kernel)
        const int e_r = entry_at(perm, g0 + r_, end);
        int e_g[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) e_g[g] = __shfl(e_r, 4 * q + g, 64);
        float4 ring[PER];
#pragma unroll
        for (int n = 0; n < PER; ++n) ring[n] = wload(0, n);
        float4 d4[LT];
#pragma unroll
        for (int t = 0; t < LT; ++t)
            d4[t] = (e_r >= 0 && (LU == 4 || 16 * t + 4 * q < L)) ? *reinterpret_cast<const float4*>(dout + (int64_t)e_r * L + 16 * t + 4 * q)
                                                                  : f4_zero();
        float dzf[TB][4];
#pragma unroll
        for (int kk = 0; kk < TB; ++kk)
#pragma unroll
            for (int g = 0; g < 4; ++g) dzf[kk][g] = 0.f;
#pragma unroll 1
        for (int rd = 0; rd < NR; ++rd) {
            const bool more = rd + 1 < NR;
            // (two accumulators per tile row: a chain of dependent MFMAs waits out the 8 passes of its predecessor)
            f32x4 acc[TB], acc2[TB];
#pragma unroll
            for (int kk = 0; kk < TB; ++kk) acc[kk] = acc2[kk] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int n = 8 * h; n < 8 * h + 8; ++n) {
                    const int kk = n / LT, t = n % LT;
                    const float4 b = ring[n];
                    acc[kk] = mfma16(d4[t].x, b.x, acc[kk]);
                    acc2[kk] = mfma16(d4[t].y, b.y, acc2[kk]);
                    acc[kk] = mfma16(d4[t].z, b.z, acc[kk]);
                    acc2[kk] = mfma16(d4[t].w, b.w, acc2[kk]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (more) {
#pragma unroll
                    for (int n = 8 * h; n < 8 * h + 8; ++n) ring[n] = wload(rd + 1, n);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (LU == 1 && zfac) {          // dz[e, k] = sum_m coef[e, m] dz'[e, (m, k)]: the round IS m
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float cf = e_g[g] >= 0 ? coef[(int64_t)e_g[g] * MB + rd] : 0.f;
#pragma unroll
                    for (int kk = 0; kk < TB; ++kk) dzf[kk][g] = fmaf(cf, acc[kk][g] + acc2[kk][g], dzf[kk][g]);
                }
                continue;
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (e_g[g] < 0) continue;
                float* __restrict__ o = dz + (int64_t)e_g[g] * Kd + 16 * (rd * TB) + r_;
#pragma unroll
                for (int kk = 0; kk < TB; ++kk) {
                    const float v = acc[kk][g] + acc2[kk][g];
                    o[16 * kk] = accumulate ? (o[16 * kk] + v) : v;
                }
            }
        }
        if (LU == 1 && zfac) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (e_g[g] < 0) continue;
                float* __restrict__ o = dz + (int64_t)e_g[g] * 64 + r_;
#pragma unroll
                for (int kk = 0; kk < TB; ++kk) o[16 * kk] = accumulate ? (o[16 * kk] + dzf[kk][g]) : dzf[kk][g];
            }
        }
    }
}

// dw[row][k][l] = sum_{e in row} z[e, k] dout[e, l]: the entries are the MFMA's K axis (16 per pass of 4 MFMAs per output tile);
// the rows of dout are held in registers across the KT tile rows when the row has one group of 16 entries (every receiver row;
// most sender rows) and re-read from L2 per tile row otherwise.  Rows without entries are written as zeros.
template <int KT, int LU>
__global__ void __launch_bounds__(STHREADS) __attribute__((amdgpu_waves_per_eu(3, 3)))
k_rowgemm_bwd_w_stream(const float* __restrict__ z, const float* __restrict__ dout, const int* __restrict__ rowptr,
                       const int* __restrict__ perm, int R, int Lrt, float* __restrict__ dw, const float* __restrict__ coef, int MB,
                       float* __restrict__ drowbias, int zfac) {
    constexpr int Kd = KT * 16;
    const int L = LU == 4 ? 256 : Lrt;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r_ = lane & 15, q = lane >> 4;
    const bool act = LU == 4 || 4 * r_ < L;
    const int row = blockIdx.x * SW + wave;
    if (row >= R) return;
    const int beg = rowptr[row], end = rowptr[row + 1];
    const bool one = end - beg <= 16;
    float* __restrict__ dwr = dw + (int64_t)row * Kd * L + 4 * r_;
    float4 b4[4][LU];
    int ent[4];
    auto load_group = [&](int g0) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            ent[s] = entry_at(perm, g0 + 4 * s + q, end);
#pragma unroll
            for (int u = 0; u < LU; ++u)
                b4[s][u] = (ent[s] >= 0 && act) ? *reinterpret_cast<const float4*>(dout + (int64_t)ent[s] * L + 64 * u + 4 * r_) : f4_zero();
        }
    };
    if (one) load_group(beg);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        f32x4 acc[4 * LU];
#pragma unroll
        for (int c = 0; c < 4 * LU; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int g0 = beg; g0 < end; g0 += 16) {
            if (!one) load_group(g0);
            float a[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (ent[s] < 0) a[s] = 0.f;
                else if (zfac) a[s] = coef[(int64_t)ent[s] * MB + (kt >> 2)] * z[(int64_t)ent[s] * 64 + 16 * (kt & 3) + r_];
                else a[s] = z[(int64_t)ent[s] * Kd + 16 * kt + r_];
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int u = 0; u < LU; ++u) {
                    acc[4 * u + 0] = mfma16(a[s], b4[s][u].x, acc[4 * u + 0]);
                    acc[4 * u + 1] = mfma16(a[s], b4[s][u].y, acc[4 * u + 1]);
                    acc[4 * u + 2] = mfma16(a[s], b4[s][u].z, acc[4 * u + 2]);
                    acc[4 * u + 3] = mfma16(a[s], b4[s][u].w, acc[4 * u + 3]);
                }
        }
        if (act) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int u = 0; u < LU; ++u)
                    *reinterpret_cast<float4*>(dwr + (int64_t)(16 * kt + 4 * q + g) * L + 64 * u) =
                        make_float4(acc[4 * u][g], acc[4 * u + 1][g], acc[4 * u + 2][g], acc[4 * u + 3][g]);
        }
    }
    if (drowbias) {     // d rowbias[row, m, :] = sum_e coef[e, m] dout[e, :]: one more tile row whose "z" is the coefficient block
        f32x4 acc[4 * LU];
#pragma unroll
        for (int c = 0; c < 4 * LU; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int g0 = beg; g0 < end; g0 += 16) {
            if (!one) load_group(g0);
            float a[4];
#pragma unroll
            for (int s = 0; s < 4; ++s)
                a[s] = (ent[s] >= 0 && r_ < MB) ? (coef ? coef[(int64_t)ent[s] * MB + r_] : 1.f) : 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int u = 0; u < LU; ++u) {
                    acc[4 * u + 0] = mfma16(a[s], b4[s][u].x, acc[4 * u + 0]);
                    acc[4 * u + 1] = mfma16(a[s], b4[s][u].y, acc[4 * u + 1]);
                    acc[4 * u + 2] = mfma16(a[s], b4[s][u].z, acc[4 * u + 2]);
                    acc[4 * u + 3] = mfma16(a[s], b4[s][u].w, acc[4 * u + 3]);
                }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (4 * q + g >= MB || !act) continue;
#pragma unroll
            for (int u = 0; u < LU; ++u)
                *reinterpret_cast<float4*>(drowbias + ((int64_t)row * MB + 4 * q + g) * L + 64 * u + 4 * r_) =
                    make_float4(acc[4 * u][g], acc[4 * u + 1][g], acc[4 * u + 2][g], acc[4 * u + 3][g]);
        }
    }
}

// shapes of the streaming kernels: (Kd, L) = (64, 256) radial pairs into a 256-wide fiber; (64, L), (192, L) with L <= 64 a
// multiple of 4 the attention's pairs (52 columns, unpadded); (256, 64) the pooled form.  EQH_ROWGEMM_LDS=1 selects the
// workgroup-per-row kernels (which want L a multiple of 16).
inline bool stream_on() {
    static const bool on = [] { const char* e = getenv("EQH_ROWGEMM_LDS"); return !(e && e[0] == '1'); }();
    return on;
}
inline int stream_shape(int Kd, int L) {
    if (!stream_on()) return 0;
    if (Kd == 64 && L == 256) return 1;
    if (L > 64 || L < 4 || (L & 3)) return 0;
    if (Kd == 64) return 2;
    if (Kd == 192) return 3;
    if (Kd == 256) return 4;
    return 0;
}
#define ROW_STREAM_LAUNCH(KERNEL, shape, ...)                                                                                  \
    do {                                                                                                                       \
        const dim3 grid_((unsigned)((R + SW - 1) / SW));                                                                 \
        if (shape == 1) hipLaunchKernelGGL((KERNEL<4, 4>), grid_, dim3(STHREADS), 0, stream, __VA_ARGS__);                      \
        else if (shape == 2) hipLaunchKernelGGL((KERNEL<4, 1>), grid_, dim3(STHREADS), 0, stream, __VA_ARGS__);                 \
        else if (shape == 3) hipLaunchKernelGGL((KERNEL<12, 1>), grid_, dim3(STHREADS), 0, stream, __VA_ARGS__);                \
        else hipLaunchKernelGGL((KERNEL<16, 1>), grid_, dim3(STHREADS), 0, stream, __VA_ARGS__);                                \
        EQH_CHECK_LAUNCH();                                                                                                    \
    } while (0)

constexpr size_t ROW_LDS_MAX = 80 * 1024;   // two workgroups per CU

inline size_t row_lds_bytes(int Kd, int L) { return (size_t)Kd * (size_t)(L + 4) * sizeof(float); }

template <typename K>
int row_lds_attr(K kernel, bool* done) {
    if (*done) return EQH_OK;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)ROW_LDS_MAX) != hipSuccess)
        return EQH_ERR_LAUNCH;
    *done = true;
    return EQH_OK;
}

int check(int64_t R, int Kd, int L) {
    if (R < 0 || Kd <= 0 || L <= 0) return EQH_ERR_ARG;
    if ((Kd & 15) || ((L & 15) && !stream_shape(Kd, L))) return EQH_ERR_ALIGN;
    if (R >= ((int64_t)1 << 31) - 1) return EQH_ERR_RANGE;
    return EQH_OK;
}

}  // namespace

static int zfac_check(int32_t Kd, int32_t L, const float* coef, int32_t MB, int32_t zfac) {
    if (!zfac) return EQH_OK;
    // z [E, 64], A column (m, k) = coef[e, m] z[e, k]: the 64-column streaming kernels, a round of 64 k-values per m
    if (!coef || MB < 1 || Kd != 64 * MB || L > 64 || !stream_shape(Kd, L)) return EQH_ERR_ARG;
    return EQH_OK;
}

extern "C" int hg_rowgemm_fwd_bias(const float* z, const float* w, const int32_t* rowptr, const int32_t* perm, int64_t R,
                                   int32_t Kd, int32_t L, float* out, int32_t accumulate, const float* rowbias,
                                   const float* coef, int32_t MB, int32_t z_factored, void* stream_) {
    int rc = check(R, Kd, L);
    if (rc) return rc;
    if (R == 0) return EQH_OK;
    if (!z || !w || !rowptr || !out) return EQH_ERR_ARG;
    if (!eqh_aligned16(z) || !eqh_aligned16(w)) return EQH_ERR_ALIGN;
    if (rowbias && (MB < 1 || MB > 16 || !eqh_aligned16(rowbias))) return EQH_ERR_ARG;
    if (!rowbias && coef && !z_factored) return EQH_ERR_ARG;
    if (zfac_check(Kd, L, coef, MB, z_factored)) return EQH_ERR_ARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (const int shape = stream_shape(Kd, L)) {
        if (!eqh_aligned16(out)) return EQH_ERR_ALIGN;
        ROW_STREAM_LAUNCH(k_rowgemm_fwd_stream, shape, z, w, rowptr, perm, (int)R, (int)L, out, (int)accumulate, rowbias, coef, (int)MB,
                          (int)z_factored);
        return EQH_OK;
    }
    if (rowbias) return EQH_ERR_ARG;          // (the bias block rides the streaming kernels only: see hg_rowgemm_bias_supported)
    const size_t lds = row_lds_bytes(Kd, L);
    if ((Kd == 64 || Kd == 192 || Kd == 256) && lds <= ROW_LDS_MAX) {   // the widths of the radial contraction (mid, 3 * mid; li of the pooled form)
        static bool a4 = false, a12 = false, a16 = false;
        const int blocks = eqh_grid_for(R, 1, 2048);
        if (Kd == 64) {
            if (row_lds_attr(k_rowgemm_fwd_lds<4>, &a4)) return EQH_ERR_LAUNCH;
            hipLaunchKernelGGL(k_rowgemm_fwd_lds<4>, dim3(blocks), dim3(THREADS), lds, stream, z, w, rowptr, perm, (int)R,
                               (int)L, out, (int)accumulate);
        } else if (Kd == 256) {
            if (row_lds_attr(k_rowgemm_fwd_lds<16>, &a16)) return EQH_ERR_LAUNCH;
            hipLaunchKernelGGL(k_rowgemm_fwd_lds<16>, dim3(blocks), dim3(THREADS), lds, stream, z, w, rowptr, perm, (int)R,
                               (int)L, out, (int)accumulate);
        } else {
            if (row_lds_attr(k_rowgemm_fwd_lds<12>, &a12)) return EQH_ERR_LAUNCH;
            hipLaunchKernelGGL(k_rowgemm_fwd_lds<12>, dim3(blocks), dim3(THREADS), lds, stream, z, w, rowptr, perm, (int)R,
                               (int)L, out, (int)accumulate);
        }
        EQH_CHECK_LAUNCH();
        return EQH_OK;
    }
    const int gy = (L >> 4) < 4 ? (L >> 4) : 4;
    dim3 grid(eqh_grid_for(R, WAVES, 4096), gy);
    hipLaunchKernelGGL(k_rowgemm_fwd, grid, dim3(THREADS), 0, stream, z, w, rowptr, perm, (int)R, (int)Kd,
                       (int)L, out, (int)accumulate);
    EQH_CHECK_LAUNCH();
    return EQH_OK;
}

extern "C" int hg_rowgemm_fwd(const float* z, const float* w, const int32_t* rowptr, const int32_t* perm, int64_t R,
                              int32_t Kd, int32_t L, float* out, int32_t accumulate, void* stream_) {
    return hg_rowgemm_fwd_bias(z, w, rowptr, perm, R, Kd, L, out, accumulate, nullptr, nullptr, 0, 0, stream_);
}

extern "C" int hg_rowgemm_bias_supported(int32_t Kd, int32_t L) { return stream_shape(Kd, L) != 0; }

extern "C" int hg_rowgemm_bwd_bias(const float* z, const float* w, const float* dout, const int32_t* rowptr,
                                   const int32_t* perm, int64_t R, int32_t Kd, int32_t L, float* dz, int32_t accumulate_dz,
                                   float* dw, const float* coef, int32_t MB, float* drowbias, int32_t z_factored, void* stream_) {
    int rc = check(R, Kd, L);
    if (rc) return rc;
    if (R == 0) return EQH_OK;
    if (!z || !dout || !rowptr || (dz && !w)) return EQH_ERR_ARG;     // (w is read for dz only)
    if (!eqh_aligned16(dout) || (dz && (!eqh_aligned16(w) || !eqh_aligned16(dz))) || !eqh_aligned16(z)) return EQH_ERR_ALIGN;
    if (drowbias && (!dw || MB < 1 || MB > 16 || !eqh_aligned16(drowbias))) return EQH_ERR_ARG;   // (it rides the dw launch)
    if (zfac_check(Kd, L, coef, MB, z_factored)) return EQH_ERR_ARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (const int shape = stream_shape(Kd, L)) {
        if (dz) ROW_STREAM_LAUNCH(k_rowgemm_bwd_z_stream, shape, dout, w, rowptr, perm, (int)R, (int)L, dz, (int)accumulate_dz, coef,
                                  (int)MB, (int)z_factored);
        if (dw) {
            if (!eqh_aligned16(dw)) return EQH_ERR_ALIGN;
            ROW_STREAM_LAUNCH(k_rowgemm_bwd_w_stream, shape, z, dout, rowptr, perm, (int)R, (int)L, dw, coef, (int)MB, drowbias,
                              (int)z_factored);
        }
        return EQH_OK;
    }
    if (drowbias) return EQH_ERR_ARG;
    const size_t lds = row_lds_bytes(Kd, L);
    if (dz && (Kd == 64 || Kd == 192 || Kd == 256) && lds <= ROW_LDS_MAX) {
        static bool a4 = false, a12 = false, a16 = false;
        const int blocks = eqh_grid_for(R, 1, 2048);
        if (Kd == 64) {
            if (row_lds_attr(k_rowgemm_bwd_z_lds<4>, &a4)) return EQH_ERR_LAUNCH;
            hipLaunchKernelGGL(k_rowgemm_bwd_z_lds<4>, dim3(blocks), dim3(THREADS), lds, stream, dout, w, rowptr, perm,
                               (int)R, (int)L, dz, (int)accumulate_dz);
        } else if (Kd == 256) {
            if (row_lds_attr(k_rowgemm_bwd_z_lds<16>, &a16)) return EQH_ERR_LAUNCH;
            hipLaunchKernelGGL(k_rowgemm_bwd_z_lds<16>, dim3(blocks), dim3(THREADS), lds, stream, dout, w, rowptr, perm,
                               (int)R, (int)L, dz, (int)accumulate_dz);
        } else {
            if (row_lds_attr(k_rowgemm_bwd_z_lds<12>, &a12)) return EQH_ERR_LAUNCH;
            hipLaunchKernelGGL(k_rowgemm_bwd_z_lds<12>, dim3(blocks), dim3(THREADS), lds, stream, dout, w, rowptr, perm,
                               (int)R, (int)L, dz, (int)accumulate_dz);
        }
        EQH_CHECK_LAUNCH();
    } else if (dz) {
        const int gy = (Kd >> 4) < 4 ? (Kd >> 4) : 4;
        dim3 grid(eqh_grid_for(R, WAVES, 4096), gy);
        hipLaunchKernelGGL(k_rowgemm_bwd_z, grid, dim3(THREADS), 0, stream, dout, w, rowptr, perm, (int)R,
                           (int)Kd, (int)L, dz, (int)accumulate_dz);
        EQH_CHECK_LAUNCH();
    }
    if (dw && ((Kd == 64 && (L == 256 || L == 64)) || ((Kd == 192 || Kd == 256) && L == 64))) {
        const int blocks = eqh_grid_for(R, 1, 2048);
        if (Kd == 256)
            hipLaunchKernelGGL((k_rowgemm_bwd_w_lds<16, 4>), dim3(blocks), dim3(THREADS), 0, stream, z, dout, rowptr, perm,
                               (int)R, dw);
        else if (Kd == 64 && L == 256)
            hipLaunchKernelGGL((k_rowgemm_bwd_w_lds<4, 16>), dim3(blocks), dim3(THREADS), 0, stream, z, dout, rowptr, perm,
                               (int)R, dw);
        else if (Kd == 64)
            hipLaunchKernelGGL((k_rowgemm_bwd_w_lds<4, 4>), dim3(blocks), dim3(THREADS), 0, stream, z, dout, rowptr, perm,
                               (int)R, dw);
        else
            hipLaunchKernelGGL((k_rowgemm_bwd_w_lds<12, 4>), dim3(blocks), dim3(THREADS), 0, stream, z, dout, rowptr, perm,
                               (int)R, dw);
        EQH_CHECK_LAUNCH();
    } else if (dw) {
        const int tiles = (Kd >> 4) * (L >> 4);
        dim3 grid(eqh_grid_for(R, WAVES, 2048), tiles < 8 ? tiles : 8);
        hipLaunchKernelGGL(k_rowgemm_bwd_w, grid, dim3(THREADS), 0, stream, z, dout, rowptr, perm, (int)R,
                           (int)Kd, (int)L, dw);
        EQH_CHECK_LAUNCH();
    }
    return EQH_OK;
}

extern "C" int hg_rowgemm_bwd(const float* z, const float* w, const float* dout, const int32_t* rowptr,
                              const int32_t* perm, int64_t R, int32_t Kd, int32_t L, float* dz,
                              int32_t accumulate_dz, float* dw, void* stream_) {
    return hg_rowgemm_bwd_bias(z, w, dout, rowptr, perm, R, Kd, L, dz, accumulate_dz, dw, nullptr, 0, nullptr, 0, stream_);
}
