"""ctypes binding of libequihgnn_hip.so — the C ABI declared in include/equihgnn_hip.h.

There is NO fallback: if the shared library is missing or a call returns an error code this
module raises.  The product path never routes through a CPU implementation.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int32, c_int64, c_size_t, c_void_p

_PKG = os.path.dirname(os.path.abspath(__file__))
# (EQH_LIB_PATH: another build of the same sources -- how two variants of a kernel are timed on ONE box, whose clocks differ
# from the next box's by more than most optimisations gain)
LIB_PATH = os.environ.get("EQH_LIB_PATH") or os.path.join(_PKG, "libequihgnn_hip.so")

class HgSmallMM(ctypes.Structure):
    """HgSmallMM of include/equihgnn_hip.h (one problem of hg_small_mm_batch)."""
    _fields_ = [("a", c_void_p), ("a_rs", c_int64), ("a_cs", c_int64), ("b", c_void_p), ("b_rs", c_int64), ("b_cs", c_int64),
                ("c", c_void_p), ("ldc", c_int64), ("u", c_void_p), ("v", c_void_p), ("x", c_void_p), ("z", c_void_p),
                ("y", c_void_p), ("w", c_void_p), ("m", c_int32), ("n", c_int32), ("k", c_int32), ("alpha", c_float),
                ("accumulate_c", c_int32), ("accumulate_y", c_int32)]


class HgGemmProblem(ctypes.Structure):
    """HgGemmProblem of include/equihgnn_hip.h (one problem of hg_gemm_x6_batch)."""
    _fields_ = [("a", c_void_p), ("lda", c_int64), ("b", c_void_p), ("ldb", c_int64), ("d", c_void_p), ("ldd", c_int64),
                ("bias", c_void_p), ("c", c_void_p), ("ldc", c_int64), ("m", c_int64), ("n", c_int32), ("k", c_int32),
                ("trans_a", c_int32), ("trans_b", c_int32), ("relu", c_int32), ("alpha", c_float), ("beta", c_float),
                ("drop_seed", c_void_p), ("drop_p", c_float), ("mean_rows", c_int32), ("b_packed", c_void_p)]


class HgPanelPack(ctypes.Structure):
    """HgPanelPack of include/equihgnn_hip.h (one weight of hg_panel_pack)."""
    _fields_ = [("w", c_void_p), ("ld", c_int64), ("dst", c_void_p), ("K", c_int32), ("N", c_int32), ("trans", c_int32),
                ("kstep0", c_int32), ("ksteps_total", c_int32), ("n_valid", c_int32), ("k_major", c_int32)]


class HgConvPanel(ctypes.Structure):
    """HgConvPanel of include/equihgnn_hip.h (operands of one hg_conv_panel stage)."""
    _fields_ = ([("rows", c_int64), ("C", c_int32), ("eps", c_float), ("scale", c_float), ("relu", c_int32),
                 ("acc_first", c_int32), ("tail", c_int32), ("accumulate", c_int32)]
                + [(n, c_void_p) for n in ("in0", "in1", "in2", "in3")] + [("ld0", c_int64)]
                + [(n, c_void_p) for n in ("rowptr", "col", "wq", "w0", "w1", "w2", "w3", "b0", "g0", "be0", "b1", "g1", "be1",
                                           "bias_out", "out0", "out1", "out2", "out3", "out4", "out5", "slab", "slab2",
                                           "acc_out", "dbias", "dgamma", "dbeta", "dbias2", "dgamma2", "dbeta2")]
                + [("g_inc", c_void_p), ("be_inc", c_void_p), ("eps_inc", c_float), ("out6", c_void_p), ("signal", c_void_p)])


class HgPanelMulti(ctypes.Structure):
    """HgPanelMulti of include/equihgnn_hip.h (up to three products of one row block)."""
    _fields_ = [("a", c_void_p), ("lda", c_int64), ("rows", c_int64), ("C", c_int32), ("n", c_int32), ("w", c_void_p * 3),
                ("bias", c_void_p * 3), ("rw", c_void_p * 3), ("d", c_void_p * 3), ("ldd", c_int64 * 3), ("out", c_void_p * 3),
                ("ldo", c_int64 * 3)]


class HgPanelSum(ctypes.Structure):
    """HgPanelSum of include/equihgnn_hip.h (a sum of up to three products over different row blocks)."""
    _fields_ = [("a", c_void_p * 3), ("lda", c_int64 * 3), ("rows", c_int64), ("C", c_int32), ("n", c_int32), ("w", c_void_p * 3),
                ("d", c_void_p), ("ldd", c_int64), ("out", c_void_p), ("ldo", c_int64)]


class HbCollate(ctypes.Structure):
    """HbCollate of include/equihgnn_hip.h (operands of hb_collate, the host-side batch assembly)."""
    _fields_ = ([("B", c_int64), ("n_mols", c_int64)]
                + [(n, c_void_p) for n in ("idx", "node_off", "he_off", "inc_off", "x", "pos", "v", "e", "edge_attr", "e_order", "y")]
                + [("PN", c_int64), ("PM", c_int64), ("PZ", c_int64), ("padded", c_int32)]
                + [(n, c_void_p) for n in ("out_x", "out_pos", "out_edge_index0", "out_edge_index1", "out_edge_attr", "out_n_e",
                                           "out_e_order", "out_batch", "out_y", "out_counts")])


HG_CONV_F1, HG_CONV_F2, HG_CONV_F3, HG_CONV_B3, HG_CONV_B1, HG_EGNN_NODE_F, HG_EGNN_NODE_B = 1, 2, 3, 4, 5, 6, 7

# name -> (restype, argtypes); mirrors include/equihgnn_hip.h one to one
SIGNATURES = {
    "hg_conv_panel_slab_bytes": (c_size_t, [c_int64, c_int32]),
    "hg_conv_panel": (c_int32, [c_int32, ctypes.POINTER(HgConvPanel), c_void_p]),
    "hg_panel_waves": (c_int32, []),
    "hg_panel_multi": (c_int32, [ctypes.POINTER(HgPanelMulti), c_void_p]),
    "hg_panel_sum": (c_int32, [ctypes.POINTER(HgPanelSum), c_void_p]),
    "hb_collate": (c_int32, [ctypes.POINTER(HbCollate)]),
    "hg_panel_pack_bytes": (c_size_t, [c_int32, c_int32]),
    "hg_panel_pack": (c_int32, [c_int32, ctypes.POINTER(HgPanelPack), c_void_p]),
    "hg_panel_gemm_f32": (c_int32, [c_void_p, c_int64, c_int64, c_int32, c_void_p, c_float, c_void_p, c_int64, c_float,
                                    c_void_p, c_int32, c_void_p, c_int64, c_void_p]),
    "hg_panel_stream_supported": (c_int32, [c_int32, c_int32]),
    "hg_panel_stream_gemm_f32": (c_int32, [c_void_p, c_int64, c_int64, c_int32, c_int32, c_void_p, c_float, c_void_p, c_int64, c_float,
                                           c_void_p, c_int32, c_void_p, c_int64, c_void_p]),
    "hg_small_mm_batch": (c_int32, [c_int32, ctypes.POINTER(HgSmallMM), c_void_p]),
    "hg_gemm_x6_workspace_bytes": (c_size_t, [c_int32, ctypes.POINTER(HgGemmProblem), c_int32]),
    "hg_gemm_x6_batch": (c_int32, [c_int32, ctypes.POINTER(HgGemmProblem), c_int32, c_void_p, c_size_t, c_void_p]),
    "hg_gemm_x6_choose_tile": (c_int32, [c_int32, ctypes.POINTER(HgGemmProblem), c_int32]),
    "eqh_version": (c_int32, []),
    "eqh_error_string": (c_char_p, [c_int32]),
    "hg_csr_build_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "hg_csr_build": (c_int32, [c_void_p, c_void_p, c_int64, c_int64, c_int32, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_size_t, c_void_p]),
    "hg_csr_build_i32": (c_int32, [c_void_p, c_void_p, c_int64, c_int64, c_int32, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_size_t, c_void_p]),
    "hg_segment_reduce_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                        c_int32, c_int32, c_void_p]),
    "hg_entry_weights": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "hg_segment_reduce_w_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p]),
    "hg_embed_sum_fwd": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int32, c_int64,
                                   c_void_p, c_void_p]),
    "hg_embed_sum_bwd_workspace_bytes": (c_size_t, [c_int64, c_int32, c_int64]),
    "hg_embed_sum_bwd": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int32, c_int64, c_void_p, c_int32,
                                   c_void_p, c_size_t, c_void_p]),
    "geo_knn": (c_int32, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "geo_knn_counted": (c_int32, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "hg_csr_build_i32_counted": (c_int32, [c_void_p, c_void_p, c_int64, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_size_t, c_void_p]),
    "egnn_edge_fwd": (c_int32, [c_void_p] * 6 + [c_int64, c_int32, c_void_p, c_void_p, c_void_p]),
    "egnn_edge_bwd_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "egnn_edge_bwd": (c_int32, [c_void_p] * 7 + [c_int64] + [c_void_p] * 2 + [c_int64, c_int32] + [c_void_p] * 5
                      + [c_int32, c_int32, c_void_p, c_size_t, c_void_p]),
    "hg_rowgemm_bias_supported": (c_int32, [c_int32, c_int32]),
    "hg_rowgemm_fwd_bias": (c_int32, [c_void_p] * 4 + [c_int64, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_int32,
                                      c_int32, c_void_p]),
    "hg_rowgemm_bwd_bias": (c_int32, [c_void_p] * 5 + [c_int64, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_int32,
                                      c_void_p, c_int32, c_void_p]),
    "hg_rowgemm_fwd": (c_int32, [c_void_p] * 4 + [c_int64, c_int32, c_int32, c_void_p, c_int32, c_void_p]),
    "hg_incidence_ln_reduce_fwd": (c_int32, [c_void_p] * 8 + [c_int64, c_int32, c_int32, c_float, c_void_p, c_void_p]),
    "hg_incidence_ln_reduce_fwd_col": (c_int32, [c_void_p] * 4 + [c_int32, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_float,
                                                 c_void_p, c_void_p]),
    "hg_incidence_ln_reduce_bwd_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "hg_incidence_ln_reduce_bwd": (c_int32, [c_void_p] * 6 + [c_int64, c_void_p, c_void_p, c_int64] + [c_void_p] * 4
                                   + [c_int32, c_int32, c_float] + [c_void_p] * 3 + [c_int32, c_void_p, c_size_t,
                                                                                    c_void_p]),
    "hg_layer_norm_fwd": (c_int32, [c_void_p] * 3 + [c_int64, c_int32, c_float, c_void_p, c_void_p]),
    "hg_layer_norm_bwd_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "hg_layer_norm_bwd": (c_int32, [c_void_p] * 3 + [c_int64, c_void_p] + [c_int64, c_int32, c_float] + [c_void_p] * 3
                          + [c_int32, c_void_p, c_size_t, c_void_p]),
    "eqf_radial_trunk_fwd": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_float, c_void_p, c_void_p]),
    "eqf_radial_trunk_bwd_workspace_bytes": (c_size_t, [c_int64]),
    "eqf_radial_trunk_bwd": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_float, c_void_p, c_int32,
                                       c_void_p, c_size_t, c_void_p]),
    "geo_knn_grid_max_points": (c_int64, []),
    "geo_knn_grid_workspace_bytes": (c_size_t, [c_int64]),
    "geo_knn_grid": (c_int32, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                               c_void_p]),
    "eqf_pool3": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "eqf_edge_geometry": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_float] + [c_void_p] * 6),
    "eqf_rms_norm_fwd": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_float, c_float, c_void_p, c_void_p]),
    "eqf_rms_norm_bwd_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "eqf_rms_norm_bwd": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_float, c_float, c_void_p, c_void_p,
                                   c_int32, c_void_p, c_size_t, c_void_p]),
    "eqf_attn_pool_fwd": (c_int32, [c_void_p] * 5 + [c_int64, c_int32, c_int32, c_int32, c_int32, c_float, c_float, c_void_p,
                                    c_void_p, c_void_p]),
    "eqf_attn_pool_bwd_workspace_bytes": (c_size_t, [c_int64]),
    "eqf_attn_pool_bwd": (c_int32, [c_void_p] * 7 + [c_int64, c_int32, c_int32, c_int32, c_int32, c_float, c_float]
                          + [c_void_p] * 4 + [c_int32, c_void_p, c_size_t, c_void_p]),
    "faf_swiglu_dropout_fwd": (c_int32, [c_void_p, c_int64, c_int32, c_float, c_void_p, c_void_p, c_void_p]),
    "faf_swiglu_dropout_bwd": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_float, c_void_p, c_void_p, c_void_p]),
    "faf_dropout_mean_fwd": (c_int32, [c_void_p, c_int64, c_int32, c_int32, c_float, c_void_p, c_void_p, c_void_p]),
    "faf_dropout_mean_bwd": (c_int32, [c_void_p, c_int64, c_int32, c_int32, c_float, c_void_p, c_void_p, c_void_p]),
    "faf_dropout_mean_bwd_colsum_workspace_bytes": (c_size_t, [c_int64, c_int32, c_int32]),
    "faf_dropout_mean_bwd_colsum": (c_int32, [c_void_p, c_int64, c_int32, c_int32, c_float, c_void_p, c_void_p, c_void_p,
                                              c_int32, c_void_p, c_size_t, c_void_p]),
    "faf_attn_sum_fwd": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "faf_attn_sum_bwd": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                                   c_void_p]),
    "faf_moments_workspace_bytes": (c_size_t, [c_int64]),
    "faf_centre_mix_fwd": (c_int32, [c_void_p] * 3 + [c_int64, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "faf_centre_mix_bwd": (c_int32, [c_void_p] * 5 + [c_int64, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "faf_cloud_frame_fwd": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "faf_cloud_frame_bwd": (c_int32, [c_void_p] * 3 + [c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "faf_edge_frame_fwd": (c_int32, [c_void_p] * 3 + [c_int64, c_int32] + [c_void_p] * 4),
    "faf_edge_frame_bwd": (c_int32, [c_void_p] * 6 + [c_int64, c_int32] + [c_void_p] * 3),
    "faf_attn_logits_fwd": (c_int32, [c_void_p] * 4 + [c_int64, c_int32, c_int32, c_float] + [c_void_p] * 4),
    "faf_attn_logits_bwd": (c_int32, [c_void_p] * 3 + [c_int64, c_int32, c_int32, c_float] + [c_void_p] * 5),
    "faf_attn_gather_sum_fwd": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p,
                                          c_void_p]),
    "faf_attn_gather_sum_bwd": (c_int32, [c_void_p, c_void_p, c_int64] + [c_void_p] * 4 + [c_int64, c_int64, c_int32, c_int32,
                                                                                         c_int32, c_void_p, c_void_p, c_void_p]),
    "faf_dropout_add": (c_int32, [c_void_p, c_void_p, c_int64, c_float, c_void_p, c_void_p, c_void_p]),
    "faf_ln_rowdot_fwd": (c_int32, [c_void_p] * 5 + [c_int64, c_int32, c_int32, c_float, c_void_p, c_void_p, c_void_p]),
    "faf_ln_rowdot_bwd_workspace_bytes": (c_size_t, [c_int64, c_int32, c_int32]),
    "faf_ln_rowdot_bwd": (c_int32, [c_void_p] * 5 + [c_int64, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_float]
                          + [c_void_p] * 3 + [c_int32, c_void_p, c_void_p, c_size_t, c_void_p]),
    "faf_edge_logit_weights_fwd": (c_int32, [c_void_p, c_int64, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                                             c_void_p]),
    "faf_edge_logit_weights_bwd": (c_int32, [c_void_p, c_int64] + [c_void_p] * 4 + [c_int32] * 3 + [c_void_p, c_int64, c_void_p,
                                                                                                  c_void_p] + [c_int32] * 3
                                   + [c_void_p]),
    "faf_frame_pre_fwd": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p]),
    "faf_frame_pre_bwd_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "faf_frame_pre_bwd": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_int32,
                                    c_void_p, c_size_t, c_void_p]),
    "faf_frame_hidden_fwd": (c_int32, [c_void_p] * 3 + [c_int64] + [c_void_p] * 4 + [c_int64, c_float, c_void_p, c_float,
                                                                                   c_void_p, c_int32, c_void_p]),
    "faf_frame_hidden_bwd_workspace_bytes": (c_size_t, [c_int64]),
    "faf_frame_hidden_bwd": (c_int32, [c_void_p] * 3 + [c_int64] + [c_void_p] * 4 + [c_int64, c_float, c_void_p, c_float]
                             + [c_void_p] * 7 + [c_int32, c_int32, c_void_p, c_size_t, c_void_p]),
    "faf_rowdot_fwd": (c_int32, [c_void_p] * 3 + [c_int64, c_int32, c_int32, c_void_p, c_void_p]),
    "faf_rowdot_bwd_workspace_bytes": (c_size_t, [c_int64, c_int32, c_int32]),
    "faf_rowdot_bwd": (c_int32, [c_void_p] * 4 + [c_int64, c_int32, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_size_t,
                                                  c_void_p]),
    "faf_gate_fwd": (c_int32, [c_void_p] * 4 + [c_int64, c_int32, c_float, c_void_p, c_void_p, c_void_p]),
    "faf_gate_bwd_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "faf_gate_bwd": (c_int32, [c_void_p] * 4 + [c_int64, c_int32, c_float] + [c_void_p] * 4 + [c_int32, c_void_p, c_int32,
                                                                                               c_void_p, c_size_t, c_void_p]),
    "faf_edge_hidden_fwd": (c_int32, [c_void_p] * 6 + [c_int64, c_int32, c_float, c_void_p, c_float, c_void_p, c_void_p]),
    "faf_edge_hidden_bwd_workspace_bytes": (c_size_t, [c_int64]),
    "faf_edge_hidden_bwd": (c_int32, [c_void_p] * 6 + [c_int64, c_int32, c_float, c_void_p, c_float] + [c_void_p] * 4
                            + [c_int32, c_void_p, c_size_t, c_void_p]),
    "eqh_permute_tiles_f32": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int64, c_int64, c_int64, c_int64,
                                        c_void_p]),
    "hg_readout_mse_supported": (c_int32, [c_int32, c_int32]),
    "hg_readout_mse_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "hg_readout_mse_f32": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_float,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_size_t,
                                     c_void_p, c_void_p]),
    "hg_bias_relu_ln_fwd": (c_int32, [c_void_p] * 4 + [c_int64, c_int32, c_float, c_void_p, c_void_p]),
    "hg_gather_ln_reduce_fwd": (c_int32, [c_void_p] * 6 + [c_int64, c_int32, c_int32, c_float, c_void_p, c_void_p]),
    "hg_gather_ln_reduce_bwd_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "hg_gather_ln_reduce_bwd": (c_int32, [c_void_p] * 7 + [c_int64, c_int32, c_float] + [c_void_p] * 4
                                + [c_int32, c_void_p, c_size_t, c_void_p]),
    "hg_batch_norm_rows_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "hg_batch_norm_rows_fwd": (c_int32, [c_void_p] * 7 + [c_float, c_float, c_int64, c_int32] + [c_void_p] * 3 + [c_int32, c_void_p, c_size_t,
                                                                                                              c_void_p]),
    "hg_batch_norm_rows_bwd": (c_int32, [c_void_p] * 6 + [c_int64, c_int32] + [c_void_p] * 4 + [c_int32, c_void_p, c_size_t, c_void_p]),
    "hg_bias_relu_ln_bwd_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "hg_bias_relu_ln_bwd": (c_int32, [c_void_p] * 4 + [c_int64, c_int32, c_float] + [c_void_p] * 4
                            + [c_int32, c_void_p, c_size_t, c_void_p]),
    "hg_bias_relu_ln_fwd_ex": (c_int32, [c_void_p, c_float] + [c_void_p] * 4 + [c_int64, c_int32, c_float, c_void_p, c_void_p]),
    "hg_bias_relu_ln_bwd_ex": (c_int32, [c_void_p, c_float] + [c_void_p] * 4 + [c_int64, c_int32, c_float] + [c_void_p] * 4
                               + [c_int32, c_void_p, c_size_t, c_void_p, c_int32, c_void_p]),
    "hg_csr_build_batch_workspace_bytes": (c_size_t, [c_int32, c_void_p, c_void_p]),
    "hg_csr_build_batch": (c_int32, [c_int32] + [c_void_p] * 8 + [c_void_p, c_size_t, c_void_p]),
    "hg_index_aux": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64] + [c_void_p] * 12 + [c_int64, c_void_p]),
    "eqh_adam_step": (c_int32, [c_void_p] * 4 + [c_int64, c_void_p] + [c_float] * 5 + [c_void_p, c_int32, c_void_p, c_int64, c_void_p]),
    "eqh_copy_many": (c_int32, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "eqh_mse_fwd_bwd": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    "eqh_stamp": (c_int32, [c_void_p, c_void_p]),
    "eqh_wall_clock_khz": (c_int64, []),
    "eqh_clock_probe": (c_int32, [c_void_p, c_int32, c_void_p]),
    "eqh_signal_post": (c_int32, [c_void_p, c_void_p]),
    "eqh_signal_wait": (c_int32, [c_void_p, c_int32, c_int32, c_void_p]),
    "eqh_accumulate": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p]),
    "eqh_event_create": (c_int32, [ctypes.POINTER(c_void_p)]),
    "eqh_event_record": (c_int32, [c_void_p, c_void_p]),
    "eqh_event_wait": (c_int32, [c_void_p, c_void_p]),
    "eqh_event_destroy": (c_int32, [c_void_p]),
    "eqh_defer_begin": (c_int32, [c_void_p]),
    "eqh_defer_flush": (c_int32, [c_void_p]),
    "hg_wgrad_workspace_bytes": (c_size_t, [c_int64, c_int32, c_int32]),
    "hg_wgrad_skinny_workspace_bytes": (c_size_t, [c_int64, c_int32, c_int32]),
    "hg_wgrad_skinny_f32": (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int32, c_int32, c_float, c_void_p, c_int64,
                                      c_int32, c_void_p, c_size_t, c_void_p]),
    "hg_wgrad_f32": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_float, c_void_p, c_int64, c_int32,
                               c_void_p, c_size_t, c_void_p]),
    "hg_wgrad_batch_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "hg_wgrad_batch_f32": (c_int32, [c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p,
                                     c_void_p, c_int32, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "hg_colsum_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "hg_colsum_batch_workspace_bytes": (c_size_t, [c_int32, c_void_p, c_void_p]),
    "hg_colsum_batch_f32": (c_int32, [c_int32] + [c_void_p] * 8 + [c_void_p, c_size_t, c_void_p]),
    "hg_colsum_f32": (c_int32, [c_void_p, c_void_p, c_int32, c_float, c_int64, c_int32, c_int32, c_void_p, c_void_p,
                                c_size_t, c_void_p]),
    "hg_residual_mix_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_float, c_int64, c_int32, c_void_p,
                                      c_void_p]),
    "egnn_pack_weights_fwd": (c_int32, [c_void_p] * 3 + [c_int32] * 3 + [c_void_p] * 5),
    "egnn_pack_weights_bwd": (c_int32, [c_void_p] * 4 + [c_int32] * 3 + [c_void_p] * 3 + [c_int32, c_void_p]),
    "geo_eigh3": (c_int32, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "hg_rowgemm_bwd": (c_int32, [c_void_p] * 5 + [c_int64, c_int32, c_int32, c_void_p, c_int32, c_void_p,
                                                c_void_p]),
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    """Load (once) and return the shared library; raise if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} is missing: build it with `python -m equihgnn_amd.build` "
            "(hipcc --offload-arch=gfx950).  equihgnn_amd has no CPU fallback.")
    try:
        handle = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise HipLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as e:
            raise HipLibraryError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    _lib = handle
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().eqh_error_string(rc)
        raise HipLibraryError(f"{what} failed with code {rc}: {msg.decode() if msg else '?'}")

