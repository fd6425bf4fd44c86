/*
 * equihgnn_hip.h — C ABI of libequihgnn_hip.so (gfx950 / MI355X).
 *
 * The reference (HySonLab/EquiHGNN) is pure Python; the only native code its hot path reaches
 * is third-party (torch_scatter, ATen, PyG).  Each entry point below replaces one such operator
 * call site (cited as reference file:line, relative to the reference checkout).  The Python host
 * side (equihgnn_amd/hip.py, used by equihgnn_amd/ops/) binds these with ctypes; INTEGRATION.md shows the stub a
 * maintainer of the reference would add.
 *
 * Conventions (every function):
 *   - extern "C", plain pointers and sizes, no torch types;
 *   - all pointers are DEVICE pointers unless the name ends in _host;
 *   - returns 0 on success, a negative EQH_ERR_* otherwise; never throws, never exits;
 *   - enqueues work on `stream` (a hipStream_t passed as void*) and returns without
 *     synchronising; never allocates or frees — scratch memory is passed in, sized by the
 *     matching *_workspace_bytes query; safe to capture in a hipGraph;
 *   - matrices are row-major, contiguous, fp32; row length C must be a multiple of 4 and
 *     rows 16-byte aligned; index arrays produced by this library are int32.
 */
#ifndef EQUIHGNN_HIP_H
#define EQUIHGNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EQH_OK 0
#define EQH_ERR_ARG (-1)     /* null pointer / negative size / unsupported shape */
#define EQH_ERR_ALIGN (-2)   /* C not a multiple of 4 or pointer not 16-byte aligned */
#define EQH_ERR_RANGE (-3)   /* size exceeds what int32 indexing supports */
#define EQH_ERR_LAUNCH (-4)  /* hipGetLastError() after launch was not hipSuccess */

int eqh_version(void);
const char* eqh_error_string(int code);

/* Optimiser-side helpers of a training step over flat parameter / gradient buffers (main.py:137-140:
 * torch.optim.Adam).  eqh_adam_step: p, exp_avg, exp_avg_sq updated in place from grad * grad_scale
 * (+ weight_decay * p); `lr` is a DEVICE float and `state` a 16-byte zero-initialised device block holding
 * the step counter (advanced by the kernel), so a captured hipGraph follows a learning-rate schedule; zero_grad != 0
 * clears `grad` after reading it (optimizer.zero_grad() of the next step without a fill launch) and, with zero_also, a
 * second buffer of zero_also_n floats (a multiple of 4; accumulators that are not parameter gradients).
 * eqh_copy_many: count device-to-device float copies (n[i] elements each) in one launch. */
int eqh_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                  const float* lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                  void* state, int32_t zero_grad, float* zero_also, int64_t zero_also_n, void* stream);
int eqh_copy_many(int32_t count, const float* const* src, float* const* dst, const int64_t* n, void* stream);
/* Mean-squared-error loss of main.py:36,49-63 and its gradient in one launch: loss[0] = mean((pred - target)^2)
 * over n values, grad[i] = 2 (pred[i] - target[i]) / n.  n <= 65536 (one workgroup; a batch of molecules). */
int eqh_mse_fwd_bwd(const float* pred, const float* target, int32_t n, float* loss, float* grad, void* stream);

/* Deferred gradient reductions.  Several backward kernels end in a fixed-order reduction of per-workgroup
 * partial slabs into a parameter gradient; with accumulate != 0 that gradient is only read by the
 * optimiser.  Between eqh_defer_begin(stream) and eqh_defer_flush(stream) such accumulating reductions
 * issued on `stream` are recorded instead of launched, and the flush performs all of them in ONE launch
 * (a training step of egnn_equihnns has ~25 of them).  The workspaces passed to the deferred calls must
 * stay allocated and untouched until the flush.  Without begin/flush every call reduces at once. */
int eqh_defer_begin(void* stream);
int eqh_defer_flush(void* stream);

/* ---------------------------------------------------------------------------------------------
 * fp32 GEMM with fp32-grade results on the bf16 matrix cores (csrc/gemm_x6.hip):
 *
 *     c[m, n] = act( alpha * op(a)[m, k] . op(b)[k, n]  (+ beta * d[m, n])  (+ bias[n]) )
 *
 * Replaces the library GEMMs behind nn.Linear / F.linear and autograd's input- and weight-gradient products
 * (mlp.py:91-99, conv.py:169-182, egnn_layer.py:180-208,360-362, equiformer_layer.py:376-383,
 * fa_former_layer.py:241-289).  Every fp32 operand element is split exactly into three bf16 numbers while its tile is
 * staged into LDS and the six significant bf16 x bf16 products are accumulated in fp32 by v_mfma_f32_16x16x32_bf16:
 * the error against exact arithmetic is that of an fp32 dot product (no larger than the fp32-input MFMA's), at 2.7x
 * its rate.  All matrices fp32 row-major:
 *   trans_a == 0: a is [m, k] (row stride lda);   trans_a != 0: a is [k, m]  (a weight gradient dY^T . X)
 *   trans_b != 0: b is [n, k] (row stride ldb) -- an nn.Linear weight used as x W^T;   trans_b == 0: b is [k, n]
 *   d: optional addend [m, n] (row stride ldd; may be c itself: accumulate), bias: optional [n], relu != 0: max(., 0).
 * n, the contiguous extents of a and b (k, or m for trans_a) and all row strides are multiples of 4 floats; pointers
 * 16-byte aligned.  Up to 8 problems with the same (trans_a, trans_b) share one launch.  tile: 0 = choose, 64, 128, 256, 512
 * or 513 (block tile 64 x 64, 128 x 64, 128 x 128, 128 x 256, 256 x 128).  Products with few output tiles and a long k (weight gradients: k = rows of the
 * batch) are split along k: partial tiles go to `workspace` (hg_gemm_x6_workspace_bytes; caller-owned, may be NULL: no
 * split) and are summed in a fixed order (d must then be NULL or c itself with beta = 1, and no bias / relu).  Bitwise
 * reproducible; capturable.
 * ------------------------------------------------------------------------------------------- */
typedef struct HgGemmProblem {
    const float* a;
    int64_t lda;
    const float* b;
    int64_t ldb;
    const float* d;
    int64_t ldd;
    const float* bias;
    float* c;
    int64_t ldc;
    int64_t m;
    int32_t n, k;
    int32_t trans_a, trans_b, relu;
    float alpha, beta;
    /* mean_rows == 8: the frame-mean epilogue of FAFormer's frame MLP (fa_former_layer.py:61-120: fc2 -> dropout ->
     * mean over the 8 sign frames).  c is [m / 8, n]:  c[e, :] = 1/8 sum_{f < 8} dropout_p(alpha a b + bias)[8 e + f, :],
     * with the keep decisions of faf_dropout_mean_fwd / _bwd on the virtual [m, n] tensor (hash of (*drop_seed, element));
     * the [m, n] product is never written.  Needs m % 8 == 0, trans_a == 0, no d, no relu; 0 = ordinary epilogue. */
    const int64_t* drop_seed;
    float drop_p;
    int32_t mean_rows;
    /* b_packed != NULL: op(b) [k, n] ALREADY split into its bf16 planes by hg_panel_pack (K = k, N = n rounded up to a multiple
     * of 32; trans as trans_b) -- the weight of a Linear, split once per call instead of once per row tile of the output.  b may
     * then be NULL; needs k % 32 == 0 and trans_a == 0; every problem of a launch in the same form. */
    const void* b_packed;
} HgGemmProblem;
size_t hg_gemm_x6_workspace_bytes(int32_t n_problems, const HgGemmProblem* problems, int32_t tile);
/* the block tile hg_gemm_x6_batch takes for these problems with tile == 0 (64, 128, 256: 64 x 64, 128 x 64, 128 x 128;
 * 512, 513: 128 x 256, 256 x 128 with eight multiplying wavefronts), from its cost estimate of the five configurations */
int32_t hg_gemm_x6_choose_tile(int32_t n_problems, const HgGemmProblem* problems, int32_t with_workspace);
int hg_gemm_x6_batch(int32_t n_problems, const HgGemmProblem* problems, int32_t tile, void* workspace, size_t workspace_bytes,
                     void* stream);

/* ---------------------------------------------------------------------------------------------
 * Batched small matrix products with riders (csrc/small_mm.hip), one launch for up to 8 problems:
 *     c[m, n] (+)= alpha * sum_k A(m, k) B(k, n) (+ u[m] v[n])      A(m, k) = a[m * a_rs + k * a_cs]
 *     y[m]    (+)= sum_k A(m, k) x[k] (+ z[m])                       B(k, n) = b[k * b_rs + n * b_cs]
 *     w[m]     += u[m]
 * The weight-level products of two merged Linears and their backward (layers.MHNNSConv._prepare_merged; conv.py:172-181):
 * any transposition through the strides; u/v, x/z/y, w optional (NULL).  fp32 FMA, fixed order: reproducible.
 * ------------------------------------------------------------------------------------------- */
typedef struct HgSmallMM {
    const float* a;
    int64_t a_rs, a_cs;
    const float* b;
    int64_t b_rs, b_cs;
    float* c;
    int64_t ldc;
    const float* u;
    const float* v;
    const float* x;
    const float* z;
    float* y;
    float* w;
    int32_t m, n, k;
    float alpha;
    int32_t accumulate_c, accumulate_y;
} HgSmallMM;
int hg_small_mm_batch(int32_t n_problems, const HgSmallMM* problems, void* stream);

/* Row-panel kernels (csrc/panel.hip) for the conv-sized dense products [rows x C] . [C x C], C = MLP_hidden in {64, 128,
 * 256}: mlp.py:91-99 as used by conv.py:169-182 (torch.nn.functional.linear + its input gradient in the reference).
 * hg_panel_pack splits weights ONCE into their three bf16 planes in MFMA operand order (fp32-grade products from six
 * bf16 MFMAs, as hg_gemm_x6_batch): item i lays B[k][n] = w[n * ld + k] (trans != 0: an nn.Linear weight used as x W^T)
 * or B[k][n] = w[k * ld + n] (trans == 0: the same weight used as dY W), K x N (K % 16 == 0, N % 32 == 0), into the image
 * at dst, hg_panel_pack_bytes(K_total, N) bytes; kstep0 / ksteps_total (in units of 16 k; 0 = this item alone) stack
 * several weights along K in one image.  All items in ONE launch (per 32).
 * hg_panel_gemm_f32: c = act(alpha * a . B + beta * d + bias), a [rows, C] (lda), B the packed C x C image. */
typedef struct {
    const float* w;
    int64_t ld;
    void* dst;
    int32_t K, N, trans, kstep0, ksteps_total;
    int32_t n_valid;     /* 0 or N: all columns; else columns n >= n_valid of B are zero (N padded to a multiple of 32) */
    int32_t k_major;     /* != 0: image[k / 32][tile n / 32][k half][plane][lane] -- the form HgGemmProblem.b_packed takes (a K step
                            of 32 of ALL column tiles contiguous; needs (kstep0 + K / 16) even); 0: image[tile][k / 16][plane][lane] */
} HgPanelPack;
size_t hg_panel_pack_bytes(int32_t K, int32_t N);
int hg_panel_pack(int32_t n_items, const HgPanelPack* items, void* stream);
int hg_panel_gemm_f32(const float* a, int64_t lda, int64_t rows, int32_t C, const void* wpack, float alpha,
                      const float* d, int64_t ldd, float beta, const float* bias, int32_t relu, float* c, int64_t ldc,
                      void* stream);
/* hg_panel_stream_gemm_f32: the same product for MANY rows and a rectangular weight -- a [rows, K] (lda), B the packed K x N
 * image (hg_panel_pack), K in {64, 128, 256}, N in {128, 256} (hg_panel_stream_supported): one persistent workgroup per CU walks
 * 32-row panels with two A images in LDS (the next panel's rows are fetched and split behind the current panel's MFMA loop).
 * Stands where the reference has F.linear on ~10^5 .. 10^6 rows (fa_former_layer.py:61-120,241-289) and its input gradient. */
int hg_panel_stream_supported(int32_t K, int32_t N);
int hg_panel_stream_gemm_f32(const float* a, int64_t lda, int64_t rows, int32_t K, int32_t N, const void* wpack, float alpha,
                             const float* d, int64_t ldd, float beta, const float* bias, int32_t relu, float* c, int64_t ldc,
                             void* stream);

/* One application of the merged MHNNSConv (conv.py:169-182; layers.MHNNSConv._forward_merged) as panel stages: each launch
 * takes panels of 32 rows through one to four [C x C] products with the row-wise work between them (bias, ReLU, LayerNorm and
 * its backward, the gathered means of conv.py:172-173 and their backward) in the same workgroup.  Weight images from
 * hg_panel_pack ("T": trans != 0, x W^T; "N": trans == 0, dy W).  All row tensors contiguous [rows, C] fp32 unless ld0 says
 * otherwise for in0.  Stages and their operands:
 *  HG_CONV_F1  in0 = X (ld0); w0 = W1a T, w1 = W2v T; b0 / g0 / be0 = b1a, gamma1, beta1.
 *              out0 = h1 = X W1a^T, out1 = h1n = LN1(relu(h1 + b1a)), out2 = pa = X W2v^T.
 *  HG_CONV_F2  rows = hyperedges; in0 = h1n; rowptr / col = incidence CSR by hyperedge; w0 = w12 T; bias_out = b12.
 *              out0 = hbar = mean over the hyperedge's nodes of h1n, out1 = qb = hbar w12^T + b12.
 *  HG_CONV_F3  in0 = s, in1 = cw; scale; w0 = w23 T; b0 / g0 / be0 = b3a, gamma3, beta3; w1 = W3b T; bias_out = b3b; relu.
 *              out0 = u = scale s w23^T + cw, out1 = x3 = LN3(relu(u + b3a)), out2 = Xn = act(x3 W3b^T + b3b).
 *              tail != 0: F1 of the next application on Xn (w2 = W1a T, w3 = W2v T, b1 / g1 / be1, out3 = h1, out4 = h1n, out5 = pa).
 *  HG_CONV_B3  in0 = dXn (ld0), in1 = Xn or NULL (ReLU mask); w0 = W3b N, w1 = w23 N; in2 = u; b0 / g0 = b3a, gamma3; scale.
 *              out0 = g = dXn [Xn > 0] (with in1), out1 = dpre, out2 = ds = scale dpre w23; acc_out (+)= dpre (acc_first: =);
 *              dbias / dgamma / dbeta (+)= the LayerNorm's vector gradients (slab: hg_conv_panel_slab_bytes, kept until
 *              eqh_defer_flush when reductions are deferred).
 *  HG_CONV_B1  in0 = dhbar (or dqb with w3 = w12 N: the product dhbar = dqb w12 is then formed here, after the gather, and
 *              needs no launch of its own); rowptr / col / wq = incidence CSR by node and its entries' 1 / deg(hyperedge); in1 = h1;
 *              b0 / g0 = b1a, gamma1; in2 = dpa; w0 = [W1a ; W2v] N stacked along K.
 *              out0 = dh1, out1 = dX = dh1 W1a + dpa W2v (may be NULL with tail); dbias / dgamma / dbeta, slab as B3.
 *              tail != 0: B3 of the application before on dX (in3 = its Xn = this X, out5 = its u (read), w1 = W3b N, w2 = w23 N,
 *              b1 / g1 = b3a, gamma3, out2 = g, out3 = dpre, out4 = ds, acc_out, slab2, dbias2 / dgamma2 / dbeta2).
 *  The EGNN node update (egnn_layer.py:180-187,360-362: node_mlp = Linear(C + 16, 2 C) -> SiLU -> Linear(2 C, C), + feats):
 *  HG_EGNN_NODE_F  in0 = normed [N, C], in1 = m_i [N, 16], in2 = feats; w0 / w1 = W0 T for output columns [0, C) / [C, 2 C)
 *              (K = C + 16), w2 = W3 T (K = 2 C); b0 = W0's bias [2 C], bias_out = W3's bias.
 *              out0 = node_in = [normed | m_i] [N, C + 16], out1 = hpre = node_in W0^T + b0 [N, 2 C], out2 = hid = silu(hpre),
 *              out3 = hid W3^T + b3 + feats.
 *  HG_EGNN_NODE_B  in0 = dout (ld0), in1 = hpre; w0 / w1 = W3 N for output columns [0, C) / [C, 2 C) (K = C), w2 = W0 N
 *              (K = 2 C, N = C + 16 packed with n_valid = C + 16 into C + 32 columns).
 *              out0 = dpre = (dout W3) silu'(hpre) [N, 2 C], out1 = dnode_in = dpre W0 [N, C + 16]. */
enum { HG_CONV_F1 = 1, HG_CONV_F2 = 2, HG_CONV_F3 = 3, HG_CONV_B3 = 4, HG_CONV_B1 = 5, HG_EGNN_NODE_F = 6, HG_EGNN_NODE_B = 7 };
typedef struct {
    int64_t rows;
    int32_t C;
    float eps, scale;
    int32_t relu, acc_first, tail, accumulate;
    const float *in0, *in1, *in2, *in3;
    int64_t ld0;
    const int32_t *rowptr, *col;
    const float* wq;
    const void *w0, *w1, *w2, *w3;
    const float *b0, *g0, *be0, *b1, *g1, *be1, *bias_out;
    float *out0, *out1, *out2, *out3, *out4, *out5;
    float *slab, *slab2, *acc_out;
    float *dbias, *dgamma, *dbeta, *dbias2, *dgamma2, *dbeta2;
    /* HG_CONV_F3 with rowptr != NULL (round 5): the per-incidence hidden layer + hyperedge -> node mean of conv.py:175-177
     * (hg_incidence_ln_reduce_fwd_col's arithmetic) runs as the stage's prologue: in0 = pa [N, C], in2 = qb [M, C],
     * rowptr / col = incidence CSR by node, g_inc / be_inc / eps_inc = that LayerNorm; out6 = s [N, C] (written). */
    const float *g_inc, *be_inc;
    float eps_inc;
    float* out6;
    /* HG_CONV_F2 (round 6): an int32 device counter that the launch's first thread adds 1 to (eqh_signal_post folded into the
     * stage: the trainer's index-prefetch stream waits on it with eqh_signal_wait), or NULL. */
    int32_t* signal;
} HgConvPanel;
size_t hg_conv_panel_slab_bytes(int64_t rows, int32_t C);
int hg_conv_panel(int32_t stage, const HgConvPanel* args, void* stream);
/* Up to three products of ONE row block in one launch: out[g] = a W_g + rw[g][row] * bias[g] + d[g], g < n <= 3 (bias, rw, d
 * may be NULL; rw: per-row weight of the bias).  MHNNConv (conv.py:87-101) with every first Linear split by input block: X
 * feeds the node halves of W1 and W3 and W4's own half, E the hyperedge half of W1 and W2's own half -- the library GEMMs of
 * `F.linear` on [rows, C] x [C, C] for the `mhnn` / `mhnnm` / `egnn_equihnn(m)` methods. */
typedef struct {
    const float* a;
    int64_t lda, rows;
    int32_t C, n;
    const void* w[3];
    const float* bias[3];
    const float* rw[3];
    const float* d[3];
    int64_t ldd[3];
    float* out[3];
    int64_t ldo[3];
} HgPanelMulti;
int hg_panel_multi(const HgPanelMulti* args, void* stream);
/* A SUM of up to three products over different row blocks of the same rows, one launch: out = sum_g a[g] W_g + d (d may be
 * NULL; out may alias d).  MHNNConv's input gradients (autograd of conv.py:87-101): dX = dpre_v W4a_X + dpa3 W3a_x + dpa1 W1a_x
 * and dE = dpre_e W2a_E + dqb1 W1a_e -- the dY W products of `F.linear`'s backward summed by autograd's AccumulateGrad in the
 * reference.  w[g]: images of hg_panel_pack. */
typedef struct {
    const float* a[3];
    int64_t lda[3];
    int64_t rows;
    int32_t C, n;
    const void* w[3];
    const float* d;
    int64_t ldd;
    float* out;
    int64_t ldo;
} HgPanelSum;
int hg_panel_sum(const HgPanelSum* args, void* stream);
/* wavefronts per panel workgroup the library launches (8; 4 with EQH_PANEL_WAVES=4 in the environment: round 4's geometry,
 * kept for same-box A/B measurements) */
int32_t hg_panel_waves(void);

/* Host-side batch assembly (no device work, no stream): molecules idx[0..B) of a structure-of-arrays dataset -- concatenated
 * fields plus per-molecule offsets [n_mols + 1], as batch.MolStore and a PyG InMemoryDataset file hold them -- written as one
 * batch with the HData.__inc__ offsets (data/utils.py:172-178; what main.py:227-229's DataLoader collate produces): node rows
 * x [., 9] int64 / pos [., 3] f32, incidences v / e (LOCAL node / hyperedge ids, shifted here), hyperedge rows edge_attr [., 1] /
 * e_order, y per molecule.  padded != 0: outputs have PN / PM / PZ rows and B + 1 molecules; one dummy molecule owns the
 * padding (atoms 10 A apart on a line 10^4 A away, incidences -1: dropped by hg_csr_build) exactly as batch.pad_batch does;
 * PN > N, PM > M, PZ >= Z required.  out_counts[3] = N, M, Z of the real molecules.  Replaces the per-molecule Python collate
 * of torch_geometric's Batch.from_data_list on the loader thread of every rank. */
typedef struct {
    int64_t B, n_mols;
    const int64_t* idx;
    const int64_t *node_off, *he_off, *inc_off;
    const int64_t* x;
    const float* pos;
    const int64_t *v, *e, *edge_attr, *e_order;
    const float* y;
    int64_t PN, PM, PZ;
    int32_t padded;
    int64_t* out_x;
    float* out_pos;
    int64_t *out_edge_index0, *out_edge_index1, *out_edge_attr, *out_n_e, *out_e_order, *out_batch;
    float* out_y;
    int64_t* out_counts;
} HbCollate;
int hb_collate(const HbCollate* args);

/* Measurement aid (bench.py, not used by the models): eqh_stamp stores the device's constant-rate wall clock into
 * *slot (uint64, device memory) from a one-thread kernel on `stream` -- capturable, so two stamps around a launch
 * time it INSIDE a replayed hipGraph; eqh_wall_clock_khz is that clock's rate. */
int eqh_stamp(void* slot, void* stream);
int64_t eqh_wall_clock_khz(void);
/* eqh_clock_probe: one wavefront waits spin_us (1..10000) microseconds on the constant-rate clock and stores
 * out[0] = shader cycles (s_memtime) and out[1] = constant-rate ticks (s_memrealtime) that passed, two uint64: the shader
 * clock the chip holds at that moment is out[0] / out[1] x eqh_wall_clock_khz (bench.py prints it beside each timed block). */
int eqh_clock_probe(void* out, int32_t spin_us, void* stream);
/* Cross-stream trigger from inside a replayed hipGraph (trainer.GraphedTrainStep's index prefetch; the reference has no
 * counterpart: it builds every index implicitly inside its scatter / topk calls, conv.py:90-98, egnn_layer.py:253-288).
 * eqh_signal_post: a one-thread kernel, capturable, that adds 1 to *counter (int32, device memory).  eqh_signal_wait: a
 * one-thread kernel that returns once *counter - target >= 0 (wrap-around safe) or after timeout_us (1..1000000)
 * microseconds, whichever comes first -- it never outlives its timeout. */
int eqh_signal_post(int32_t* counter, void* stream);
int eqh_signal_wait(const int32_t* counter, int32_t target, int32_t timeout_us, void* stream);
/* dst[0 .. n) += src[0 .. n) (fp32).  Between eqh_defer_begin and eqh_defer_flush on `stream` the addition is recorded and runs
 * in the flush's one batched launch (src must stay alive until then), otherwise at once.  The reference's counterpart is
 * autograd's AccumulateGrad (`p.grad += g`) for the 1-D parameters of its normalisation layers. */
int eqh_accumulate(const float* src, float* dst, int64_t n, void* stream);
/* Events that order two streams of one device and nothing else (hipEventDisableTiming | hipEventDisableSystemFence: recording
 * one does not write back / invalidate the L2, which a default event does -- at the head of every training step in the trainer's
 * index prefetch).  eqh_event_wait makes `stream` wait for the event's latest record.  Not for host synchronisation. */
int eqh_event_create(void** event);
int eqh_event_record(void* event, void* stream);
int eqh_event_wait(void* event, void* stream);
int eqh_event_destroy(void* event);

/* ---------------------------------------------------------------------------------------------
 * Incidence CSR.  Replaces the implicit "unsorted int64 index" contract of
 * torch_scatter.scatter (conv.py:91-93,97,173,177): the COO incidence list is sorted ONCE per
 * batch into a CSR so that every later aggregation is an atomic-free segmented reduction.
 *
 *   key[nnz]  int64  row id of every entry (e.g. edge_index1 for the by-hyperedge CSR)
 *   other     int64  optional second coordinate (e.g. edge_index0); may be NULL
 *   rowptr[n_rows+1], perm[nnz], col[nnz]  int32 outputs:
 *     perm  = entry ids, grouped by key, ascending entry id inside a row (stable => results
 *             are bitwise reproducible run to run);
 *     col   = other[perm] if other != NULL, else perm / col_div (col_div >= 1); may be NULL.
 *   Entries with key outside [0, n_rows) are dropped (torch_scatter leaves this undefined).
 * ------------------------------------------------------------------------------------------- */
size_t hg_csr_build_workspace_bytes(int64_t nnz, int64_t n_rows);
int hg_csr_build(const int64_t* key, const int64_t* other, int64_t nnz, int64_t n_rows,
                 int32_t col_div, int32_t* rowptr, int32_t* perm, int32_t* col,
                 void* workspace, size_t workspace_bytes, void* stream);
/* the same with int32 keys (neighbour lists as geo_knn writes them) */
int hg_csr_build_i32(const int32_t* key, const int64_t* other, int64_t nnz, int64_t n_rows, int32_t col_div,
                     int32_t* rowptr, int32_t* perm, int32_t* col, void* workspace, size_t workspace_bytes,
                     void* stream);
/* the same when the histogram of the keys exists already: counts [n_rows + 2] = occurrences of every key followed by two
 * zeros, i.e. an array that was zeroed (hg_index_aux's zero_buf) before the producer of the keys counted into it
 * (geo_knn_counted).  The clear and histogram launches are skipped; counts is consumed (it becomes the fill cursors). */
int hg_csr_build_i32_counted(const int32_t* key, const int64_t* other, int64_t nnz, int64_t n_rows, int32_t col_div,
                             int32_t* rowptr, int32_t* perm, int32_t* col, int32_t* counts, void* workspace,
                             size_t workspace_bytes, void* stream);
/* n independent builds at once (arrays of n pointers / sizes, same meaning as above): the three CSRs a
 * model step derives from the batch structure cost 3 launches instead of 18. */
/* The per-batch index vectors the layers read besides the CSRs, in one launch: int32 copies of the
 * incidence coordinates (null incidences -- either coordinate out of range -- become -1 in both) and of `batch`
 * (batch32 may be NULL), and
 * the float 0/1 masks "row has at least one incidence" of the two CSRs (conv.py's mean leaves such rows
 * at zero, so the bias of the last Linear must not reach them).  Optional (NULL to skip): col_v / col_e, the `col`
 * arrays of the two CSRs, and ew_v / ew_e, which receive the per-entry mean weights of each CSR with respect to the
 * other's rows (ew_v[q] = 1 / max(deg_e(col_v[q]), 1)), as hg_entry_weights would.  zero_buf [zero_n] ints (may be NULL
 * with zero_n 0) are cleared by the same launch: the counters of the neighbour search that follows (geo_knn_counted). */
int hg_index_aux(const int64_t* vertex, const int64_t* edges, int64_t nnz, const int64_t* batch,
                 int64_t n_nodes, int64_t n_edges, const int32_t* rowptr_v, const int32_t* rowptr_e,
                 int32_t* v32, int32_t* e32, int32_t* batch32, float* has_v, float* has_e,
                 const int32_t* col_v, const int32_t* col_e, float* ew_v, float* ew_e, int32_t* zero_buf, int64_t zero_n,
                 void* stream);
size_t hg_csr_build_batch_workspace_bytes(int32_t n, const int64_t* nnz, const int64_t* n_rows);
int hg_csr_build_batch(int32_t n, const int64_t* const* key, const int64_t* const* other, const int64_t* nnz,
                       const int64_t* n_rows, const int32_t* col_div, int32_t* const* rowptr,
                       int32_t* const* perm, int32_t* const* col, void* workspace, size_t workspace_bytes,
                       void* stream);

/* ---------------------------------------------------------------------------------------------
 * Segmented reduction / row gather — torch_scatter.scatter(src, index, dim=-2, reduce) at
 * conv.py:91-93,97,173,177, the advanced-index gathers X[..., idx, :] at conv.py:90,96,172,
 * 175,176 (and their backward passes), and global_add_pool at equihnn_egnn.py:167, mhnn.py:216,
 * equihnn_equiformer.py:91.
 *
 *   out[r, :] = s(r) * sum_{q in [rowptr[r], rowptr[r+1])}  w(idx[q]) * src[idx[q], :]
 *
 *   rowptr == NULL : one source per output row (q = r): a pure row gather;
 *   idx    == NULL : identity (q-th source row);
 *   mean != 0      : s(r) = 1 / max(rowptr[r+1]-rowptr[r], 1)   (torch_scatter "mean");
 *   src_wptr != NULL : w(j) = 1 / max(src_wptr[j+1]-src_wptr[j], 1), the mean-weights of the
 *                      source rows in ANOTHER CSR — this is what the backward of a gathered
 *                      mean needs; NULL => w = 1.
 * Rows with no entry are written as zeros.  Summation order inside a row is the CSR order.
 * A negative idx[q] is a null entry and contributes a zero row (padded incidences of a static-shape batch).
 * ------------------------------------------------------------------------------------------- */
int hg_segment_reduce_f32(const float* src, const int32_t* idx, const int32_t* rowptr,
                          const int32_t* src_wptr, float* out, int64_t n_out_rows, int32_t C,
                          int32_t mean, void* stream);
/* Per-entry form of the mean weights: hg_entry_weights writes w[q] = 1 / max(wptr[idx[q]+1] - wptr[idx[q]], 1) (0 for a
 * null entry) once per batch; hg_segment_reduce_w_f32 is hg_segment_reduce_f32 with w(idx[q]) read as w[q] (sum, no
 * per-row scale): the backward of a gathered mean without the dependent rowptr lookups. */
int hg_entry_weights(const int32_t* idx, const int32_t* wptr, int64_t nnz, float* w, void* stream);
int hg_segment_reduce_w_f32(const float* src, const int32_t* idx, const int32_t* rowptr, const float* entry_w, float* out,
                            int64_t n_out_rows, int32_t C, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Embedding-sum — ogb AtomEncoder (equihnn_egnn.py:121,157; mhnn.py:164,201) and
 * nn.Embedding(6, C) for bond types (mhnn.py:165,202).
 *   out[n,:] = sum_{f<F} table[off_host[f] + x[n,f], :]      (f ascending, as ogb does)
 * ------------------------------------------------------------------------------------------- */
int hg_embed_sum_fwd(const int64_t* x, const float* table, const int32_t* off_host, int32_t F,
                     int64_t N, int32_t C, int64_t table_rows, float* out, void* stream);

/* Backward of the embedding-sum: dtable[g,:] = sum_{n,f : off[f]+x[n,f] == g} dout[n,:]
 * (autograd of the nn.Embedding lookups above).  Two passes over node chunks, no atomics,
 * bitwise reproducible.  dtable is fully overwritten. */
size_t hg_embed_sum_bwd_workspace_bytes(int64_t N, int32_t C, int64_t table_rows);
int hg_embed_sum_bwd(const int64_t* x, const float* dout, const int32_t* off_host, int32_t F,
                     int64_t N, int32_t C, int64_t table_rows, float* dtable, int32_t accumulate,
                     void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * k nearest neighbours over the whole batch point cloud.
 *   mode 0 — EGNN (egnn_layer.py:253-288): key = squared distance ((ci-cj)^2).sum(-1), self
 *            INCLUDED; requires N >= k.
 *   mode 1 — Equiformer (equiformer_layer.py:1216-1346): key = sqrt of the same sum, self
 *            EXCLUDED; requires N-1 >= k.
 * Outputs are sorted by (key, index) ascending: nbr int32 [N,k], dist fp32 [N,k] (the key).
 * Distance arithmetic is (dx*dx + dy*dy) + dz*dz without fused multiply-add, matching the
 * fp32 reduction order of the reference's CPU path.
 * ------------------------------------------------------------------------------------------- */
int geo_knn(const float* pos, int64_t N, int32_t k, int32_t mode, int32_t* nbr, float* dist,
            void* stream);
/* geo_knn that also counts how often every point is listed: counts[j] += 1 per list entry j (counts [N] zeroed by the
 * caller): the row lengths of the transposed neighbour graph, which hg_csr_build_i32_counted starts from. */
int geo_knn_counted(const float* pos, int64_t N, int32_t k, int32_t mode, int32_t* nbr, float* dist, int32_t* counts,
                    void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused EGNN edge update for k = 16 neighbours, m_dim = 16 — egnn_layer.py:298-310,357-358
 * (neighbour feature gather, cat(f_i, f_j, d2), edge_mlp = Linear -> SiLU -> Linear -> SiLU,
 * sum over neighbours) and its autograd.
 *   ab   [N, 2*Hp]  node-level pre-activations: ab[i, :Hp] = W1_i f_i + b1, ab[j, Hp:] = W1_j f_j
 *                   (Hp = hidden width 2(2C+1) zero-padded to a multiple of 16)
 *   wd   [Hp]       column of W1 that multiplies the squared distance
 *   w2   [16, Hp], b2 [16]   second edge Linear
 *   nbr  [N,16] int32, d2 [N,16]   geo_knn(mode 0) outputs
 * fwd:  m[i,:] = sum_j silu(W2 silu(ab[i,:Hp] + ab[nbr_ij,Hp:] + wd*d2_ij) + b2);  pre2 [N,16,16]
 *       (pre-activation of the second SiLU) is saved for the backward.
 * bwd:  given dm [N,16] (row stride dm_ld floats, >= 16 and a multiple of 4): dab [N,2*Hp], dwd [Hp], dw2 [16,Hp], dpre2 [N,16,16], and db2 [16] = the sum
 *       of dpre2 over its first two axes (NULL: not wanted; overwritten, or added to with
 *       db2_accumulate != 0; dwd / dw2 likewise with dw_accumulate != 0).  t_rowptr / t_perm: CSR of the transposed neighbour graph
 *       (hg_csr_build with key = nbr flattened, n_rows = N; entries are i*16+slot).
 * ------------------------------------------------------------------------------------------- */
int egnn_edge_fwd(const float* ab, const float* wd, const float* w2, const float* b2,
                  const int32_t* nbr, const float* d2, int64_t N, int32_t Hp, float* m, float* pre2,
                  void* stream);
size_t egnn_edge_bwd_workspace_bytes(int64_t N, int32_t Hp);
int egnn_edge_bwd(const float* ab, const float* wd, const float* w2, const int32_t* nbr,
                  const float* d2, const float* pre2, const float* dm, int64_t dm_ld, const int32_t* t_rowptr,
                  const int32_t* t_perm, int64_t N, int32_t Hp, float* dab, float* dwd, float* dw2,
                  float* dpre2, float* db2, int32_t db2_accumulate, int32_t dw_accumulate, void* workspace,
                  size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * CSR-grouped small GEMMs — the Equiformer's radial tensor product (equiformer_layer.py:376-383:
 * R = Radial(d) [lo, li] per edge, out = R x) re-associated so that the per-edge radial weights
 * are never formed:  out[e, :] (+)= z[e, :Kd] . w[row(e)]  with one [Kd, L] matrix per CSR row
 * (node), entries (edges) listed by rowptr / perm (perm NULL = identity: entry q belongs to the
 * row whose range contains q).  Kd a multiple of 16; L a multiple of 16, or -- for Kd of {64, 192, 256} -- any multiple
 * of 4 up to 64 (hg_rowgemm_bias_supported(Kd, L) reports those shapes).
 * bwd: dz[e, :] (+)= dout[e, :] . w[row(e)]^T (skipped if dz NULL);
 *      dw[r] = sum_{e in row r} z[e]^T (x) dout[e]  (fully overwritten; skipped if dw NULL; w may be NULL when dz is:
 *      this half alone is the POOLED form of the product, sum_e w_e R_e x_e = reshape(W3)[lo, (li, k)] . sum_e x_e (x) w_e z_e).
 * ------------------------------------------------------------------------------------------- */
int hg_rowgemm_fwd(const float* z, const float* w, const int32_t* rowptr, const int32_t* perm,
                   int64_t R, int32_t Kd, int32_t L, float* out, int32_t accumulate, void* stream);
int hg_rowgemm_bwd(const float* z, const float* w, const float* dout, const int32_t* rowptr,
                   const int32_t* perm, int64_t R, int32_t Kd, int32_t L, float* dz,
                   int32_t accumulate_dz, float* dw, void* stream);
/* The same with the row's BIAS BLOCK riding along (the b3 term of the radial network's last Linear, equiformer_layer.py:
 * 451-479: reshape(b3)[lo, li] x[li] per node):  out[e, :] += sum_{m < MB} coef[e, m] rowbias[row(e), m, :]
 * (rowbias [R, MB, L]; coef [E, MB] or NULL = 1; MB <= 16), and in the backward pass
 * drowbias[r, m, :] = sum_{e in row r} coef[e, m] dout[e, :] beside dw (drowbias needs dw).  Only for the shapes
 * hg_rowgemm_bias_supported(Kd, L) reports (L a multiple of 64, Kd of {64, 192, 256}); coef carries no gradient.
 * z_factored != 0 (Kd = 64 MB, L <= 64): z is [E, 64] and column (m, k) of the row operand is coef[e, m] z[e, k] -- the
 * (1 -> 0) pair's D[e, m] z_e[k] (equiformer_layer.py:376-383) without the [E, 3 * 64] product; dz is [E, 64] then. */
int hg_rowgemm_bias_supported(int32_t Kd, int32_t L);
int hg_rowgemm_fwd_bias(const float* z, const float* w, const int32_t* rowptr, const int32_t* perm, int64_t R, int32_t Kd,
                        int32_t L, float* out, int32_t accumulate, const float* rowbias, const float* coef, int32_t MB,
                        int32_t z_factored, void* stream);
int hg_rowgemm_bwd_bias(const float* z, const float* w, const float* dout, const int32_t* rowptr, const int32_t* perm,
                        int64_t R, int32_t Kd, int32_t L, float* dz, int32_t accumulate_dz, float* dw, const float* coef,
                        int32_t MB, float* drowbias, int32_t z_factored, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused per-incidence hidden layer + aggregation — conv.py:90-93,96-97,175-177 with the MLP of
 * mlp.py:91-99 (Linear -> ReLU -> LayerNorm -> Linear) after splitting the first Linear into
 * node/hyperedge-level products (pa = X Wa^T, qb = E Wb^T + b) and moving the last Linear behind
 * the (linear) aggregation:
 *   out[r,:] = s(r) * sum_{q in row r} LN(relu(pa[ia[p]] + qb[ib[p]]))  with p = perm[q]
 *            = gamma * (s(r) * sum xhat_p) + beta * (mean ? [deg>0] : deg)
 * (rowptr, perm) is the CSR of the OUTPUT rows over incidences; ia/ib are the per-incidence row
 * indices (int32) into pa/qb.  C <= 1024, multiple of 4.
 * bwd: (a_rowptr,a_perm) / (b_rowptr,b_perm) are the CSRs of the incidences keyed by ia / ib;
 * okey[p] is the output row of incidence p and orowptr the forward rowptr.  Produces dpa
 * [n_a_rows,C], dqb [n_b_rows,C] and dgamma [C] (added to dgamma if accumulate != 0: a layer applied
 * L times per step sums its parameter gradients in place); d beta is a row-weighted column sum of ds
 * (hg_colsum_f32 with orowptr).  Recomputes the LayerNorm statistics; nothing is saved by the forward.
 * ------------------------------------------------------------------------------------------- */
int hg_incidence_ln_reduce_fwd(const float* pa, const float* qb, const int32_t* ia, const int32_t* ib,
                               const int32_t* rowptr, const int32_t* perm, const float* gamma,
                               const float* beta, int64_t n_rows, int32_t C, int32_t mean, float eps,
                               float* out, void* stream);
/* the same forward when the output row IS the index of one operand (row_is_a != 0: ia[p] = r, ib[p] = col[q];
 * else ib[p] = r, ia[p] = col[q]) -- (rowptr, col) is then the whole description of the incidences of a row */
int hg_incidence_ln_reduce_fwd_col(const float* pa, const float* qb, const int32_t* rowptr, const int32_t* col,
                                   int32_t row_is_a, const float* gamma, const float* beta, int64_t n_rows, int32_t C,
                                   int32_t mean, float eps, float* out, void* stream);
size_t hg_incidence_ln_reduce_bwd_workspace_bytes(int64_t n_a_rows, int32_t C);
int hg_incidence_ln_reduce_bwd(const float* pa, const float* qb, const int32_t* ia, const int32_t* ib,
                               const int32_t* a_rowptr, const int32_t* a_perm, int64_t n_a_rows,
                               const int32_t* b_rowptr, const int32_t* b_perm, int64_t n_b_rows,
                               const int32_t* okey, const int32_t* orowptr, const float* ds,
                               const float* gamma, int32_t C, int32_t mean, float eps, float* dpa,
                               float* dqb, float* dgamma, int32_t accumulate, void* workspace,
                               size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Dense-row hidden layer of mlp.py:91-99: out = LayerNorm(relu(h + bias)) with h = x W^T computed
 * by a bias-free library GEMM.  bwd: dh = d relu . d LayerNorm (recomputed statistics) and, from
 * the same pass, dbias, dgamma, dbeta (C floats each, three separate destinations; dbias = column
 * sums of dh is the bias gradient of the preceding Linear), overwritten or, with accumulate != 0,
 * added to.  C <= 1024, multiple of 4.
 * ------------------------------------------------------------------------------------------- */
int hg_bias_relu_ln_fwd(const float* h, const float* bias, const float* gamma, const float* beta,
                        int64_t n_rows, int32_t C, float eps, float* out, void* stream);
size_t hg_bias_relu_ln_bwd_workspace_bytes(int64_t n_rows, int32_t C);
int hg_bias_relu_ln_bwd(const float* h, const float* bias, const float* gamma, const float* dy,
                        int64_t n_rows, int32_t C, float eps, float* dh, float* dbias, float* dgamma,
                        float* dbeta, int32_t accumulate, void* workspace, size_t workspace_bytes,
                        void* stream);
/* generalised form: pre-activation = h_scale * h + pre_add[r] + bias (pre_add [n_rows, C] may be NULL with h_scale 1): the
 * beta = 1 addend and the alpha of the GEMM that produced h, applied here (conv.py:179-180 ahead of W3, mlp.py:91-99);
 * bwd: dh is the gradient of the pre-activation, also summed into acc_out [n_rows, C] when given (overwritten if
 * acc_first != 0, else added to) -- the gradient of pre_add over several applications of the layer */
int hg_bias_relu_ln_fwd_ex(const float* h, float h_scale, const float* pre_add, const float* bias, const float* gamma,
                           const float* beta, int64_t n_rows, int32_t C, float eps, float* out, void* stream);
int hg_bias_relu_ln_bwd_ex(const float* h, float h_scale, const float* pre_add, const float* bias, const float* gamma,
                           const float* dy, int64_t n_rows, int32_t C, float eps, float* dh, float* dbias, float* dgamma,
                           float* dbeta, int32_t accumulate, void* workspace, size_t workspace_bytes, float* acc_out,
                           int32_t acc_first, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The same hidden layer when its output is consumed ONLY through a gathered reduction -- conv.py:172-173,
 * scatter(W1(X)[..., vertex, :], edges, reduce) with W1's last Linear moved behind the (linear) reduction:
 *   out[r] = gamma * reduce_{q in row r} xhat(relu(h[col[q]] + bias)) + beta * (mean ? [deg r > 0] : deg r)
 * One launch instead of hg_bias_relu_ln_fwd + hg_segment_reduce_f32; the [rows of h, C] normalised tensor
 * is never written.  bwd: (t_rowptr, t_col) is the TRANSPOSED CSR (its rows are the rows of h), t_w the
 * per-entry weights of hg_entry_weights for the mean (NULL: sum); dh [n_src_rows, C], dbias / dgamma / dbeta
 * as for hg_bias_relu_ln_bwd.  Replaces torch_scatter.scatter + mlp.py:93-97 and their autograd.
 * ------------------------------------------------------------------------------------------- */
int hg_gather_ln_reduce_fwd(const float* h, const float* bias, const float* gamma, const float* beta,
                            const int32_t* rowptr, const int32_t* col, int64_t n_rows, int32_t C, int32_t mean,
                            float eps, float* out, void* stream);
size_t hg_gather_ln_reduce_bwd_workspace_bytes(int64_t n_src_rows, int32_t C);
int hg_gather_ln_reduce_bwd(const float* h, const float* bias, const float* gamma, const float* dout,
                            const int32_t* t_rowptr, const int32_t* t_col, const float* t_w, int64_t n_src_rows,
                            int32_t C, float eps, float* dh, float* dbias, float* dgamma, float* dbeta,
                            int32_t accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Training-mode nn.BatchNorm1d over node rows (mhnn.py:182,206; equihnn_egnn.py egnn_equihnnm) whose batch statistics
 * count only the rows with row_mask[i] > 0 (NULL: all rows) -- the real atoms of a padded static-shape batch:
 *   y = (x - mean) * rstd * gamma + beta,  mean / var over the masked rows (biased var for the output, unbiased into
 *   running_var), running buffers updated with `momentum`, *num_batches_tracked += 1 (both optional: NULL).
 * Two launches each way (float64 partial column sums over row chunks, then every workgroup sums the partials of its columns
 * in chunk order and writes its rows: bitwise reproducible, nothing atomic); save_mean / save_rstd [C] feed the backward,
 * which returns dx [R, C], dgamma, dbeta [C] (overwritten).  C % 4 == 0; workspace (8-byte aligned) of
 * hg_batch_norm_rows_workspace_bytes(R, C) for the partials.  relu != 0: y = max(0, .) in the same pass (the activation
 * mhnn.py:208-214 applies behind the normalisation); the backward then takes beta too and gates dy by the sign the forward saw.
 * ------------------------------------------------------------------------------------------- */
size_t hg_batch_norm_rows_workspace_bytes(int64_t R, int32_t C);
int hg_batch_norm_rows_fwd(const float* x, const float* row_mask, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum, float eps,
                           int64_t R, int32_t C, float* y, float* save_mean, float* save_rstd, int32_t relu, void* workspace,
                           size_t workspace_bytes, void* stream);
int hg_batch_norm_rows_bwd(const float* x, const float* dy, const float* row_mask, const float* gamma,
                           const float* save_mean, const float* save_rstd, int64_t R, int32_t C, float* dx, float* dgamma,
                           float* dbeta, const float* beta, int32_t relu, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Plain LayerNorm over dense rows — nn.LayerNorm(dim) applied to node features at
 * egnn_layer.py:192 (node_norm) — one wave per row.  bwd: dx and, from the same pass, dgamma and
 * dbeta (C floats each; overwritten or, with accumulate != 0, added to); add [n_rows, C] (may be NULL) is
 * added to dx — a second gradient of the same input (the residual of egnn_layer.py:362) that would
 * otherwise cost an add kernel.  dy_ld: row stride of dy in floats (>= C, multiple of 4: dy may be a column
 * block of a wider matrix).  C <= 1024, multiple of 4.
 * ------------------------------------------------------------------------------------------- */
int hg_layer_norm_fwd(const float* x, const float* gamma, const float* beta, int64_t n_rows, int32_t C,
                      float eps, float* out, void* stream);
size_t hg_layer_norm_bwd_workspace_bytes(int64_t n_rows, int32_t C);
int hg_layer_norm_bwd(const float* x, const float* gamma, const float* dy, int64_t dy_ld, const float* add,
                      int64_t n_rows, int32_t C, float eps, float* dx, float* dgamma, float* dbeta, int32_t accumulate,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * geo_knn through a uniform cell grid: identical outputs (neighbours ordered by (distance, index), same
 * distance arithmetic, same modes and error behaviour as geo_knn above), O(N) distance evaluations.
 * n_box: device int32 or NULL — the bounding box of the grid is taken over the first *n_box points only
 * (a padded batch passes its number of real atoms: padding atoms are parked far away and are clamped
 * into the boundary cells); NULL = all N.  N <= geo_knn_grid_max_points().
 * ------------------------------------------------------------------------------------------- */
int64_t geo_knn_grid_max_points(void);
size_t geo_knn_grid_workspace_bytes(int64_t N);
int geo_knn_grid(const float* pos, int64_t N, int32_t k, int32_t mode, const int32_t* n_box, int32_t* nbr,
                 float* dist, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Attention-weighted neighbour sum of FAFormer (fa_former_layer.py:497-506, einsum "nhm,nmhd->nhd"):
 *   out[n, c] = sum_{m < K} attn[n, c / D, m] * x[n, m, c],   attn [N, H, K], x [N, K, H*D], out [N, H*D]
 * bwd: dx [N, K, H*D] and dattn [N, H, K] from dout [N, H*D].  K <= 16; H*D / 4 and D / 4 powers of two, H*D <= 256.
 * ------------------------------------------------------------------------------------------- */
int faf_attn_sum_fwd(const float* attn, const float* x, int64_t N, int32_t K, int32_t H, int32_t D, float* out,
                     void* stream);
int faf_attn_sum_bwd(const float* attn, const float* x, const float* dout, int64_t N, int32_t K, int32_t H, int32_t D,
                     float* dx, float* dattn, void* stream);
/* The same sum over rows gathered on the fly -- fa_former_layer.py:497-506 over batched_index_select(values, neighbours):
 *   out[n, c] = sum_m attn[n, c / D, m] * x[nbr[n, m], c];  x [NS, H*D] node rows with row stride x_ld floats (a column
 *   block of the qkv product), nbr [N, K] int32.  The [N, K, H*D] gathered tensor is never formed.
 * bwd: dattn [N, H, K]; dx [NS, H*D] (may be NULL) through the transposed neighbour CSR (rowptr [NS + 1], perm = entry
 * ids n * K + m ascending inside a row: fixed summation order). */
int faf_attn_gather_sum_fwd(const float* attn, const float* x, int64_t x_ld, const int32_t* nbr, int64_t N, int32_t K,
                            int32_t H, int32_t D, float* out, void* stream);
int faf_attn_gather_sum_bwd(const float* attn, const float* x, int64_t x_ld, const int32_t* nbr, const float* dout,
                            const int32_t* rowptr, const int32_t* perm, int64_t N, int64_t NS, int32_t K, int32_t H,
                            int32_t D, float* dattn, float* dx, void* stream);

/* ---------------------------------------------------------------------------------------------
 * FAFormer's small geometric steps, one launch each (+ one partial-sum pass where the step reduces over the whole
 * cloud; float64 partial sums in a fixed order: bitwise reproducible).  csrc/faformer_geom.hip.
 * `row_mask` [N] float (> 0: a real atom of a padded batch) or NULL; workspace >= faf_moments_workspace_bytes(N).
 *
 * centre mix -- fa_former_layer.py:552-571 as the frame-0 gather leaves it (the geometric context is the centroid):
 *   out[i] = c g_i + geo[i] (1 - g_i), g = sigmoid(logit), c = sum_valid geo / #valid (float64, rounded once)
 *   aux [4] = (c, #valid), kept for the backward; bwd: dgeo [N,3], dlogit [N] from dout [N,3]
 * cloud frame -- create_frame on one point set, fa_former_layer.py:86-113 (called at :310-318):
 *   y = (x - c m) V, c = masked centroid (count clamped at 1), V = eigenvectors (geo_eigh3's convention) of the
 *   masked covariance about c; no gradient through V (:98-99).  aux [13] = (V row-major, c, count)
 * edge frame -- the same per atom over its K <= 16 neighbour offsets, fa_former_layer.py:357-372:
 *   rel_k = geo[i] - gj[i,k,:3] (gj [N,K,4]: the neighbours' padded coordinates), d2 = |rel|^2, mask [N,K] bytes,
 *   y [N,K,3], V [N,9]; bwd: dgeo [N,3] (receiver side) and dgj [N,K,4] from dy / dd2 (either may be NULL)
 * attention logits -- fa_former_layer.py:483-496: logits a_q[i] + a_k[j] + l_e, masked_fill(~mask, -1e9), softmax
 *   over the K <= 16 slots, dropout (keep decisions of drop_hash.h).  qa [N,4] = (a_q of the H <= 2 heads, a_k of the
 *   H heads), qan [N,K,4] = qa gathered by neighbour, le [N,K,H]; prob / attn [N,H,K] (prob only written when p > 0);
 *   bwd: dqa [N,4] (the a_q columns), dqan [N,K,4] (the a_k columns), dle [N,K,H]
 * ------------------------------------------------------------------------------------------- */
size_t faf_moments_workspace_bytes(int64_t N);
int faf_centre_mix_fwd(const float* geo, const float* logit, const float* row_mask, int64_t N, float* out, float* aux,
                       void* workspace, size_t workspace_bytes, void* stream);
int faf_centre_mix_bwd(const float* dout, const float* geo, const float* logit, const float* row_mask, const float* aux,
                       int64_t N, float* dgeo, float* dlogit, void* workspace, size_t workspace_bytes, void* stream);
int faf_cloud_frame_fwd(const float* x, const float* row_mask, int64_t N, float* y, float* aux, void* workspace,
                        size_t workspace_bytes, void* stream);
int faf_cloud_frame_bwd(const float* dy, const float* row_mask, const float* aux, int64_t N, float* dx, void* workspace,
                        size_t workspace_bytes, void* stream);
int faf_edge_frame_fwd(const float* geo, const float* gj, const uint8_t* mask, int64_t N, int32_t K, float* y, float* d2,
                       float* V, void* stream);
int faf_edge_frame_bwd(const float* geo, const float* gj, const uint8_t* mask, const float* V, const float* dy,
                       const float* dd2, int64_t N, int32_t K, float* dgeo, float* dgj, void* stream);
int faf_attn_logits_fwd(const float* qa, const float* qan, const float* le, const uint8_t* mask, int64_t N, int32_t K,
                        int32_t H, float p, const int64_t* seed, float* prob, float* attn, void* stream);
/* out = res + dropout_p(x) over n floats (n % 4 == 0; res may be NULL): the residual behind FAFormer's MLPs
 * (fa_former_layer.py:289 with :508 and :606); keep decisions = the hash of (seed, element index) (drop_hash.h).  Its
 * backward for x is the same call on dout with res = NULL. */
int faf_dropout_add(const float* x, const float* res, int64_t n, float p, const int64_t* seed, float* out, void* stream);
/* LayerNorm over dense rows with J <= 2 row-wise dot products of its output riding along (csrc/ln_rowdot.hip) --
 * fa_former_layer.py:436-441 (the LayerNorm in front of the edge Linear) with :483-489 (the per-head edge logits, linear
 * in the normalised edge features):  out = LayerNorm(x) [R, C],  le[r, j] = out[r, :] . U[j, :] + cb[j]  (cb may be NULL).
 * bwd: dx [R, C] = LNbwd(dy + sum_j dle[r, j] U[j, :]) + add;  dy [R, C] (row stride dy_ld), dle [R, J], add [R, C] may be
 * NULL;  dU [J, C] overwritten;  dgamma / dbeta [C] overwritten or (accumulate != 0) added to.  C <= 1024, multiple of 4. */
int faf_ln_rowdot_fwd(const float* x, const float* gamma, const float* beta, const float* U, const float* cb, int64_t R,
                      int32_t C, int32_t J, float eps, float* out, float* le, void* stream);
size_t faf_ln_rowdot_bwd_workspace_bytes(int64_t R, int32_t C, int32_t J);
int faf_ln_rowdot_bwd(const float* x, const float* gamma, const float* beta, const float* U, const float* dy, int64_t dy_ld,
                      const float* dle, const float* add, int64_t R, int32_t C, int32_t J, float eps, float* dx, float* dgamma,
                      float* dbeta, int32_t accumulate, float* dU, void* workspace, size_t workspace_bytes, void* stream);
/* The edge logits' folded weights -- fa_former_layer.py:483-489, Linear(deh, 1) of the edge query (itself the first de rows
 * of a Linear(de, 2 de) of the edge features), folded at weight level:
 *   u[h, j] = sum_d we[d] W[h * deh + d, j],  c[h] = sum_d we[d] b[h * deh + d];   W [H * deh, de] (row stride ldw), we [deh]
 * bwd: dW [H * deh, de] (row stride lddw), db [H * deh], dwe [deh] from du [H, de] / dc [H] (either may be NULL), each
 * overwritten or (acc_* != 0) added to. */
int faf_edge_logit_weights_fwd(const float* W, int64_t ldw, const float* b, const float* we, int32_t H, int32_t deh, int32_t de,
                               float* u, float* c, void* stream);
int faf_edge_logit_weights_bwd(const float* W, int64_t ldw, const float* b, const float* we, const float* du, const float* dc,
                               int32_t H, int32_t deh, int32_t de, float* dW, int64_t lddw, float* db, float* dwe,
                               int32_t acc_w, int32_t acc_b, int32_t acc_e, void* stream);
int faf_attn_logits_bwd(const float* prob, const float* dattn, const uint8_t* mask, int64_t N, int32_t K, int32_t H, float p,
                        const int64_t* seed, float* dqa, float* dqan, float* dle, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Per-edge geometry of the Equiformer front-end -- equiformer_layer.py:1250-1252,1317-1346 (rel_pos = x_i - x_j of
 * the selected neighbours, neighbor_mask = dist <= radius), equiformer/basis.py:194-215 with :169-191 and
 * irr_repr.py:105-118,23-32 (D[1] of the rotation taking r_ij onto y, built in float64 with the |x+y|^2 >= 1e-6
 * clamp), equiformer/utils.py:71-82 (masked-mean weights).
 *   pos [N,3]; nbr [N,K] int32 and dist [N,K] from geo_knn(mode 1); K <= 16
 *   rhat        [N*K,3]  D[:, m=0], the only column of D the type-0 output depends on: r_hat for generic edges,
 *                        y for coincident atoms, the clamped construction within 1e-3 rad of -y
 *   maskf       [N,K]    1.0 where dist <= radius
 *   mean_w      [N,K]    maskf / max(count, 1)  (all zero for a node without an in-radius neighbour)
 *   mean_w_rhat [N,K,3]  mean_w * rhat
 *   dmat        [N*K,3,3] or NULL: the whole D[1] = Z(a) J Z(b) J Z(c) (degree-1 outputs, depth > 1)
 * No gradient (positions are data; the reference builds D under no_grad).
 * ------------------------------------------------------------------------------------------- */
int eqf_edge_geometry(const float* pos, const int32_t* nbr, const float* dist, int64_t N, int32_t K, float radius,
                      float* rhat, float* maskf, float* mean_w, float* mean_w_rhat, float* dmat, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Pooled (0 -> 1) pair: dst[n, m, c] = sum_k w3[n, k, m] src[n, k, c] for m < 3 (component-major) -- masked_mean over the K neighbour slots
 * (equiformer/utils.py:71-82) of out[e, c] D[e, m] (equiformer_layer.py:432-436), w3 = mean_w_rhat of eqf_edge_geometry.
 * backward != 0: the same map turned around, dst[n, k, c] = sum_m w3[n, k, m] src[n, m, c] (src = the output's gradient).
 * C a multiple of 4; src, dst 16-byte aligned.
 * ------------------------------------------------------------------------------------------- */
int eqf_pool3(const float* src, const float* w3, int64_t N, int32_t K, int32_t C, int32_t backward, float* dst, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Degree-0 branch of the Equiformer's `Norm` (equiformer_layer.py:194-225): out = t / max(rms, eps) * g with
 * rms = ||t||_2 * scale (scale = C^-1/2 as fp32), g = transforms.0 [C].  bwd: dx and dg (overwritten, or added to
 * with accumulate != 0); the clamp passes the gradient where rms >= eps.  C <= 1024, multiple of 4.
 * ------------------------------------------------------------------------------------------- */
int eqf_rms_norm_fwd(const float* x, const float* g, int64_t n_rows, int32_t C, float scale, float eps, float* out,
                     void* stream);
size_t eqf_rms_norm_bwd_workspace_bytes(int64_t n_rows, int32_t C);
int eqf_rms_norm_bwd(const float* x, const float* g, const float* dy, int64_t n_rows, int32_t C, float scale,
                     float eps, float* dx, float* dg, int32_t accumulate, void* workspace, size_t workspace_bytes,
                     void* stream);

/* ---------------------------------------------------------------------------------------------
 * Tail of the Equiformer's MLP attention (equiformer_layer.py:871-955, one head): per node n with its own row
 * me[n, :D] (slot 0, always valid, :877-878) and its K = 16 edge rows edge[n*K + s, :D] (valid where
 * mask[n, s] != 0):  logit_s = scale * w_logit . LeakyReLU_slope(x_s[0:4]);  attn = softmax over the valid slots
 * (masked ones filled with -max, :912-915);  out[n, :V] = sum_s attn_s * (SiLU(x_s[v_off : v_off+V]) @ wv).
 * attn [N, K+1] is saved for the backward.  bwd: dme / dedge (every column written: zeros outside the two
 * column blocks that are read), dw_logit [4], dwv [V, V] (overwritten, or added to with accumulate != 0).
 * K = 16, V = 48; D, v_off multiples of 4.
 * ------------------------------------------------------------------------------------------- */
int eqf_attn_pool_fwd(const float* me, const float* edge, const float* mask, const float* w_logit, const float* wv,
                      int64_t N, int32_t K, int32_t D, int32_t v_off, int32_t V, float scale, float slope, float* out,
                      float* attn, void* stream);
size_t eqf_attn_pool_bwd_workspace_bytes(int64_t N);
int eqf_attn_pool_bwd(const float* me, const float* edge, const float* mask, const float* w_logit, const float* wv,
                      const float* attn, const float* dout, int64_t N, int32_t K, int32_t D, int32_t v_off, int32_t V,
                      float scale, float slope, float* dme, float* dedge, float* dw_logit, float* dwv,
                      int32_t accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Trunk of the Equiformer radial network, `Radial.rp[0..5]` of equiformer_layer.py:451-479, per edge:
 * Linear(1,64) -> SiLU -> LayerNorm -> Linear(64,64) -> SiLU -> LayerNorm (the local LayerNorm of
 * :158-165: learnable gamma, beta a zero buffer).  dist [E]; params[8] = {w0 [64] (= rp.0.weight[:,0]),
 * b0, gamma1, beta1, W1 [64,64], b1, gamma2, beta2}; out [E,64].  bwd recomputes the forward from dist
 * (nothing saved) and yields dparams[6] = {dw0, db0, dgamma1, dW1, db1, dgamma2} (overwritten, or added to
 * with accumulate != 0); dist gets no gradient (positions are data, basis.py:194).  M must be 64.
 * ------------------------------------------------------------------------------------------- */
int eqf_radial_trunk_fwd(const float* dist, const float* const* params, int64_t E, int32_t M, float eps,
                         float* out, void* stream);
size_t eqf_radial_trunk_bwd_workspace_bytes(int64_t E);
int eqf_radial_trunk_bwd(const float* dist, const float* const* params, const float* dh, int64_t E, int32_t M,
                         float eps, float* const* dparams, int32_t accumulate, void* workspace,
                         size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Readout head, forward + loss + complete backward in one launch: global_add_pool over the sorted node
 * rows of each molecule (equihnn_egnn.py:167, mhnn.py:216, equihnn_equiformer.py:91), the output MLP
 * C -> H -> H -> 1 with LayerNorm after ReLU (mlp.py:91-99 as built at equihnn_egnn.py:139-149 with
 * output_num_layers = 3, Normalization "ln"), y.view(-1), and F.mse_loss over the first n_real molecules
 * (main.py:49-63; the rest of the n_graphs rows are padding).
 *   x [n_nodes, C]; rowptr int32 [n_graphs + 1] (nodes of molecule b = rows rowptr[b] .. rowptr[b+1]-1);
 *   weights[10] = {W1 [H,C], b1, gamma1, beta1, W2 [H,H], b2, gamma2, beta2, w3 [1,H], b3 [1]};
 *   y [n_graphs] predictions (rows >= n_real are not meaningful).
 * target == NULL: predictions only.  Otherwise also loss [1], dx [n_nodes, C] = d loss / d x (written for
 * every node row), and the ten parameter gradients dweights[10] (same shapes; overwritten or, with
 * accumulate != 0, added to).  state: one int32 in device memory, zero before the first launch (the kernel
 * leaves it zero); calls sharing one state must be stream-ordered.  Supported: C in {64,128,256}, H in {64,128}.
 * ------------------------------------------------------------------------------------------- */
int hg_readout_mse_supported(int32_t C, int32_t H);
size_t hg_readout_mse_workspace_bytes(int32_t n_graphs, int32_t C, int32_t H);
int hg_readout_mse_f32(const float* x, const int32_t* rowptr, int32_t n_graphs, int32_t n_real, int32_t C,
                       int32_t H, const float* const* weights, float eps, const float* target, float* y,
                       float* loss, float* dx, float* const* dweights, int32_t accumulate, void* workspace,
                       size_t workspace_bytes, int32_t* state, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Elementwise passes of FAFormer's SwiGLU MLP (fa_former_layer.py:241-289) and its 8-frame average (:61-120):
 *   faf_swiglu_dropout: out [R, H] = dropout_p(SiLU(pre[:, :H]) * pre[:, H:]),  pre [R, 2H];
 *   faf_dropout_mean:   out [R, C] = mean over the F consecutive rows r*F .. r*F+F-1 of dropout_p(x),  x [R*F, C].
 * p in [0, 1): 0 = no dropout (seed may be NULL); otherwise element i is kept (and scaled by 1/(1-p)) by a hash of
 * (*seed, i), *seed an int64 in device memory; the backward recomputes the decisions from the same seed, nothing
 * is stored.  H, C multiples of 4.
 * ------------------------------------------------------------------------------------------- */
int faf_swiglu_dropout_fwd(const float* pre, int64_t R, int32_t H, float p, const int64_t* seed, float* out,
                           void* stream);
int faf_swiglu_dropout_bwd(const float* pre, const float* dout, int64_t R, int32_t H, float p, const int64_t* seed,
                           float* dpre, void* stream);
int faf_dropout_mean_fwd(const float* x, int64_t R, int32_t F, int32_t C, float p, const int64_t* seed, float* out,
                         void* stream);
int faf_dropout_mean_bwd(const float* dout, int64_t R, int32_t F, int32_t C, float p, const int64_t* seed, float* dx,
                         void* stream);
/* faf_dropout_mean_bwd with a rider: colsum[c] (+)= sum over the R*F rows of dx[:, c] -- the bias gradient
 * (grad_output.sum(0)) of the nn.Linear whose output the forward consumed (fa_former_layer.py:61-120: fc2 of the frame
 * MLP), taken from the values this pass writes instead of by another pass over the [R*F, C] tensor.  C a power of two,
 * 4 .. 1024; `workspace` of faf_dropout_mean_bwd_colsum_workspace_bytes(R, F, C) bytes holds the per-workgroup partial
 * sums, reduced in fixed order (deferred with the other accumulating reductions between eqh_defer_begin / flush). */
size_t faf_dropout_mean_bwd_colsum_workspace_bytes(int64_t R, int32_t F, int32_t C);
int faf_dropout_mean_bwd_colsum(const float* dout, int64_t R, int32_t F, int32_t C, float p, const int64_t* seed, float* dx,
                                float* colsum, int32_t accumulate, void* workspace, size_t workspace_bytes, void* stream);
/* First Linear of the frame-averaged MLP: out[e, f, :] = w3 (y[e] * s_f) + base[e, :] for the 8 sign frames
 * s_f = ((f&4 ? +1 : -1), (f&2 ? +1 : -1), (f&1 ? +1 : -1)) (fa_former_layer.py:70-84 order); y [E,3], w3 [H,3]
 * (= fc1.weight[:, :3], contiguous), base [E,H], out [E,8,H].  bwd reads dpre [E,8,H] once: dy [E,3], dbase [E,H],
 * dw3 [H,3] (overwritten, or added to with accumulate != 0).  H must be 256. */
int faf_frame_pre_fwd(const float* y, const float* w3, const float* base, int64_t E, int32_t H, float* out, void* stream);
size_t faf_frame_pre_bwd_workspace_bytes(int64_t E, int32_t H);
int faf_frame_pre_bwd(const float* y, const float* w3, const float* dpre, int64_t E, int32_t H, float* dy, float* dbase,
                      float* dw3, int32_t accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* The hidden layer of the frame-averaged MLP in ONE launch each way (fa_former_layer.py:61-120 with :241-289):
 *   hn[e, f, :] = LayerNorm( dropout_p( SiLU(a_f) * b_f ) ),  [a_f | b_f] = w3 (y_e * s_f) + base_e,  256 -> 128 channels
 * = faf_frame_pre_fwd + faf_swiglu_dropout_fwd + hg_layer_norm_fwd without the [E * 8, 256] and [E * 8, 128]
 * intermediates (same dropout hash over the [E * 8, 128] tensor as faf_swiglu_dropout_*).  base: [E, 256] (base_ld 256)
 * or one row broadcast (base_ld 0).  w_ld = 3: w3 [256, 3] packed (and wx [256] packed); w_ld = 4: w3 is fc1.weight
 * [256, 4] read in place (columns 0-2) and wx, when given, must be w3 + 3 (its column 3).  bwd: dy [E, 3], dbase [E, 256],
 * dw3 [256, 3] and dwx [256] (packed in either case), dgamma / dbeta [128]. */
int faf_frame_hidden_fwd(const float* y, const float* w3, const float* base, int64_t base_ld, const float* extra,
                         const float* wx, const float* gamma, const float* beta, int64_t E, float p, const int64_t* seed,
                         float eps, float* out, int32_t w_ld, void* stream);
size_t faf_frame_hidden_bwd_workspace_bytes(int64_t E);
/* vector form (wx != NULL; base = fc1's bias vector, base_ld 0; the point's row is bias + extra[e] * wx, extra may be
 * NULL): dbase receives d bias [256], dwx d wx [256], dextra [E] the gradient of extra; row form: dbase [E, 256]. */
int faf_frame_hidden_bwd(const float* y, const float* w3, const float* base, int64_t base_ld, const float* extra,
                         const float* wx, const float* gamma, const float* dhn, int64_t E, float p, const int64_t* seed,
                         float eps, float* dy, float* dbase, float* dwx, float* dextra, float* dw3, float* dgamma,
                         float* dbeta, int32_t accumulate, int32_t w_ld, void* workspace, size_t workspace_bytes,
                         void* stream);

/* Row-wise dot products -- nn.Linear(C, J) with J <= 4 outputs on edge rows (fa_former_layer.py:340-400 att_mlp,
 * :483-489 the per-head edge logits): y [R, J] = x [R, C] . U [J, C]^T + bias [J] (may be NULL), a wavefront per row.
 * bwd: dx [R, C] = dx_add (may be NULL: a second gradient of x that rides along) + dy U; dU [J, C] overwritten or
 * accumulated.  C % 4 == 0, C <= 1024. */
int faf_rowdot_fwd(const float* x, const float* U, const float* bias, int64_t R, int32_t C, int32_t J, float* y, void* stream);
size_t faf_rowdot_bwd_workspace_bytes(int64_t R, int32_t C, int32_t J);
int faf_rowdot_bwd(const float* x, const float* U, const float* dy, const float* dx_add, int64_t R, int32_t C, int32_t J,
                   float* dx, float* dU, int32_t accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* The sigmoid gate of EdgeModule (fa_former_layer.py:340-400: pair * att_mlp(pair)) with the dropout in front of it and the
 * residual behind it:  out = res (may be NULL) + xd * sigmoid(xd . w + b),  xd = dropout_p(x)  (hash of (seed, element)).
 * bwd: dx [R, C], dw [C], db [1] (overwritten or accumulated); the gradient of res is dout itself.  dx_colsum [C] (may be
 * NULL): the column sums of dx from the same pass (overwritten, or added to with accumulate_colsum != 0) -- the bias
 * gradient `grad_output.sum(0)` of the Linear that produced x (`edge_mlp.fc2`), which then needs no pass of its own. */
int faf_gate_fwd(const float* x, const float* w, const float* b, const float* res, int64_t R, int32_t C, float p,
                 const int64_t* seed, float* out, void* stream);
size_t faf_gate_bwd_workspace_bytes(int64_t R, int32_t C);
int faf_gate_bwd(const float* x, const float* w, const float* b, const float* dout, int64_t R, int32_t C, float p,
                 const int64_t* seed, float* dx, float* dw, float* db, int32_t accumulate, float* dx_colsum,
                 int32_t accumulate_colsum, void* workspace, size_t workspace_bytes, void* stream);

/* Hidden layer of EdgeModule's edge MLP on the kNN edges (fa_former_layer.py:340-400 with :241-289), first Linear split by
 * input block:  hn[i, k, :] = LayerNorm(dropout_p(SiLU(a) * b)),  [a | b] = A[i] + B[nbr[i, k]] + Cf[i, k]  (256 -> 128).
 * One launch each way instead of gather + two adds + SwiGLU + LayerNorm.  bwd: dpre [N * K, 256] (= d Cf; d B is its
 * reduction over the transposed neighbour CSR), dA [N, 256], dgamma / dbeta [128]. */
int faf_edge_hidden_fwd(const float* A, const float* B, const float* Cf, const int32_t* nbr, const float* gamma,
                        const float* beta, int64_t N, int32_t K, float p, const int64_t* seed, float eps, float* out,
                        void* stream);
size_t faf_edge_hidden_bwd_workspace_bytes(int64_t N);
int faf_edge_hidden_bwd(const float* A, const float* B, const float* Cf, const int32_t* nbr, const float* gamma,
                        const float* dhn, int64_t N, int32_t K, float p, const int64_t* seed, float eps, float* dpre, float* dA,
                        float* dgamma, float* dbeta, int32_t accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* Batched transposition with free strides: dst[b * dB + x * dX + y] = src[b * sB + y * sY + x] (x < nX, y < nY, b < nB;
 * the source is contiguous along x, the destination along y), zero for nY <= y < nY_pad.  Re-lays the radial network's
 * last weight nn.Linear(64, lo * li).weight [(lo, li), k] (equiformer_layer.py:451-479) out as [li, (k, lo_pad)] for the
 * node-level GEMM, and its gradient back. */
int eqh_permute_tiles_f32(const float* src, float* dst, int32_t nX, int32_t nY, int32_t nY_pad, int32_t nB, int64_t sY,
                          int64_t sB, int64_t dX, int64_t dB, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Batched symmetric 3x3 eigen-decomposition — torch.linalg.eigh(C, UPLO="U") at
 * fa_former_layer.py:100 (frame averaging).  a [B,3,3] (upper triangle read), w [B,3] ascending
 * (may be NULL), v [B,3,3] eigenvectors in columns, largest component of each column positive.
 * ------------------------------------------------------------------------------------------- */
int geo_eigh3(const float* a, int64_t B, float* w, float* v, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Dense helpers around the library GEMMs.
 * hg_colsum_f32: out[c] (+)= sum_r w_r x[r,c] — the bias gradient of an nn.Linear (autograd's
 *   grad_output.sum(0)); two passes, fixed order.  weight_mode 0: w_r = 1 (rowptr may be NULL);
 *   1: w_r = [row r of the CSR rowptr is non-empty]; 2: w_r = length of row r (the bias of a Linear
 *   applied before a mean / sum over incidences, conv.py:91-97,175-177).  The sums are multiplied by
 *   ``scale``.  accumulate != 0 adds to out.
 * hg_residual_mix_f32: out[r,:] = alpha * x0[r,:] + (1 - alpha) * w_r * bias[:] (w_r by weight_mode 1 / 2 as
 *   above) — the layer-independent half of the residual mix (1-a) * x_v + a * X0 of conv.py:179-180 with the
 *   bias of the Linear that produced x_v folded in; its bias gradient is hg_colsum_f32 with scale = 1 - alpha.
 * egnn_pack_weights_fwd/bwd: layout change of the EGNN edge-MLP weights (egnn_layer.py:180-186) into
 *   what egnn_edge_fwd consumes: w1 [H, 2C+1], b1 [H], w2 [16, H]  ->  w_cat [2*Hp, C]
 *   (= [w1[:, :C] ; w1[:, C:2C]], zero rows from H to Hp), b_cat [2*Hp] (= [b1 ; 0]),
 *   wd [Hp] (= w1[:, 2C]), w2p [16, Hp]; bwd is the exact adjoint (overwriting dw1 / db1 / dw2, or adding to
 *   them with accumulate != 0).
 * ------------------------------------------------------------------------------------------- */
/* Weight gradient of a Linear, dw[o*ldw + i] (+)= alpha * sum_k dy[k*O + o] * x[k*I + i]  (autograd's
 * grad_output.t() @ input, mlp.py:91-99 / conv.py:90-97,172-180): split-K fp32 MFMA, fixed summation
 * order.  O and I multiples of 64; dw may be a column block of a wider matrix (ldw >= I, multiple of 4). */
size_t hg_wgrad_workspace_bytes(int64_t K, int32_t O, int32_t I);
int hg_wgrad_f32(const float* dy, const float* x, int64_t K, int32_t O, int32_t I, float alpha, float* dw,
                 int64_t ldw, int32_t accumulate, void* workspace, size_t workspace_bytes, void* stream);
/* hg_wgrad_skinny_f32: dw [O x J] (ldw) (+)= alpha * dy^T x for dy [K, O] (ld_dy), x [K, J] (ld_x), J <= 16 -- the 16-wide m_i
 * block of the EGNN node MLP's first Linear (egnn_layer.py:180-187).  Partial sums per 64-row chunk in `workspace`
 * (hg_wgrad_skinny_workspace_bytes), added by the fixed-order slab reduction (deferred inside eqh_defer_begin / _flush). */
size_t hg_wgrad_skinny_workspace_bytes(int64_t K, int32_t O, int32_t J);
int hg_wgrad_skinny_f32(const float* dy, int64_t ld_dy, const float* x, int64_t ld_x, int64_t K, int32_t O, int32_t J, float alpha,
                        float* dw, int64_t ldw, int32_t accumulate, void* workspace, size_t workspace_bytes, void* stream);
/* count such products of one [O x I] shape in one launch (a training step defers the weight gradients of its
 * backward pass to the end: 21 products at the BASELINE batch, which fill the chip together where one alone
 * cannot).  Entries with the same destination must be adjacent; they are added in array order. */
size_t hg_wgrad_batch_workspace_bytes(int32_t count, int32_t O, int32_t I);
int hg_wgrad_batch_f32(int32_t count, const float* const* dy, const float* const* x, const int64_t* K, int32_t O,
                       int32_t I, const float* alpha, float* const* dw, const int64_t* ldw, int32_t accumulate,
                       void* workspace, size_t workspace_bytes, void* stream,
                       const int64_t* ld_dy, const int64_t* ld_x);
size_t hg_colsum_workspace_bytes(int64_t R, int32_t C);
int hg_colsum_f32(const float* x, const int32_t* rowptr, int32_t weight_mode, float scale, int64_t R, int32_t C,
                  int32_t accumulate, float* out, void* workspace, size_t workspace_bytes, void* stream);
int hg_residual_mix_f32(const float* x0, const float* bias, const int32_t* rowptr, int32_t weight_mode,
                        float alpha, int64_t R, int32_t C, float* out, void* stream);
/* count row-weighted column sums in one launch, each ADDED to its out[i] (the bias gradients of a backward
 * pass, deferred to its end like the weight gradients).  scale may be NULL (all 1).  ld[i] (NULL: C[i]): floats between
 * consecutive rows of x[i], a multiple of 4 and >= C[i] -- an entry may be the leading C columns of a wider matrix. */
size_t hg_colsum_batch_workspace_bytes(int32_t count, const int64_t* R, const int32_t* C);
int hg_colsum_batch_f32(int32_t count, const float* const* x, const int64_t* ld, const int32_t* const* rowptr,
                        const int32_t* weight_mode, const float* scale, const int64_t* R, const int32_t* C,
                        float* const* out, void* workspace, size_t workspace_bytes, void* stream);
int egnn_pack_weights_fwd(const float* w1, const float* b1, const float* w2, int32_t H, int32_t Hp,
                          int32_t C, float* w_cat, float* b_cat, float* wd, float* w2p, void* stream);
int egnn_pack_weights_bwd(const float* dw_cat, const float* db_cat, const float* dwd, const float* dw2p,
                          int32_t H, int32_t Hp, int32_t C, float* dw1, float* db1, float* dw2,
                          int32_t accumulate, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EQUIHGNN_HIP_H */
