"""Host-side operators over libequihgnn_hip.so: torch.autograd.Functions whose forward and
backward are C-ABI kernel launches on the current HIP stream.

PyTorch supplies device memory, streams and the autograd tape; all gather / scatter /
neighbour-search / embedding arithmetic runs in the hand-written gfx950 kernels.  Nothing here
has a CPU fallback: tensors must live on a HIP device.

One module per subsystem; everything is re-exported here, so callers keep writing ``ops.linear``:

  _base    pointers / streams for the C ABI, in-graph Timeline, accumulator registry, deferred-scratch bookkeeping
  aggregate CSR builds, gather / segmented reduce (torch_scatter.scatter of conv.py), embedding sums, kNN
  products x6 GEMM, fp32-MFMA dense batch, dispatch against the library, weight-level small products
  grads    persistent gradient accumulators, deferred weight / bias gradients and slab reductions, gradient fan-in
  linears  nn.Linear-shaped autograd nodes, merged consecutive Linears, matmul fans
  rows     per-incidence hidden layer + reduce, bias/ReLU/LayerNorm rows, BatchNorm / LayerNorm rows, residual mix
  egnn     EGNN edge kernels, feature node, weight packing
  readout  pooled readout head + MSE in one launch
  se3      Equiformer: row GEMMs, radial trunk, attention pooling, RMS norm, edge geometry
  frames   FAFormer: frame-averaged SwiGLU pieces, edge hidden layer, row dots, gates, attention sums, eigh3

Module-level switches (``ops.GEMM_TILE``, ``ops.TIMELINE``, ``ops.KNN_GRID_MIN_POINTS``, ...) are read by the
submodule that owns them; assigning them on this package forwards the value there (_OpsModule below).
"""
import importlib as _importlib
import sys as _sys
import types as _types

from ._base import (  # noqa: F401
    _c_void_p, _ptr, _stream, _require_gpu, _f32c, Timeline, TIMELINE, timed, StepSignal, StreamEvent, SIGNAL, signal_point, _row_view, _as2d, _contiguous_run,
    _stacked_view, _rows_ld, _DEFER, _workspace, _acc_target, _hand_out, _note_acc, LINEAR_PARAMS, ACC_PARAMS,
)
from .aggregate import (  # noqa: F401
    CSR, csr_build, csr_build_batch, index_aux, segment_reduce_bytes, _segment_reduce, entry_weights,
    _segment_reduce_w, _ReduceGathered, _ReduceEntries, _GatherRows, _EmbedSum, reduce_gathered, reduce_entries,
    gather_rows, embed_sum, KNN_GRID_MIN_POINTS, knn, scatter,
)
from .products import (  # noqa: F401
    GemmProblem, GEMM_TILE, gemm_supported, gemm_batch, gemm,
    X6_MIN_OUTPUTS, X6_MAX_K, X6_DEEP_ROWS, X6_WGRAD_OUTPUTS, X6_WGRAD_ROWS, USE_X6, _x6_ok, mm_nt, mm_nn, small_mm_batch,
)
from .panel import (  # noqa: F401
    PANEL_WIDTHS, panel_supported, panel_pack, panel_pack_bytes, panel_gemm, panel_stream_gemm, panel_stream_supported,
    conv_panel, conv_panel_slab,
)
from .grads import (  # noqa: F401
    GradFan, _FanSource, fanout, WGRAD_ON_SIDE_STREAM, _WGRAD_STREAMS, wgrad_stream, join_wgrad_stream,
    defer_begin, wgrad_batch, colsum_batch, defer_flush, copy_many, colsum, USE_WGRAD_KERNEL, DEFER_WGRAD,
    _wgrad_shape_ok, _wgrad_ok, _wgrad_deferred, wgrad, _linear_weight_grad, MergedScratch, _merged_acc,
)
from .linears import (  # noqa: F401
    _MergedWeight, _MergedWeights, merged_weights, merged_weight, _Linear, _Linear2, _LinearAddC, linear, linear2,
    linear_add, _MatmulFan, matmul_fan, matmul,
)
from .conv_stack import (  # noqa: F401
    conv_stack_supported, merged_conv_stack, _MergedConvStack,
)
from . import conv_stack  # noqa: F401
from .mhnn_panel import (mhnn_conv_panel, mhnn_panel_supported, panel_multi, panel_sum, merged_scope)  # noqa: F401
from . import mhnn_panel  # noqa: F401
from .rows import (  # noqa: F401
    _IncidenceLnReduce, _BiasReluLn, _LinearAddReluLn, linear_add_relu_ln, _GatherLnReduce, gather_ln_reduce,
    _BatchNormRows, batch_norm_rows, batch_norm_rows_supported, _LayerNormRows, incidence_ln_reduce, bias_relu_ln,
    _ResidualMix, residual_mix, layer_norm_rows,
)
from .egnn import (  # noqa: F401
    _EgnnEdge, _EgnnFeats, egnn_feats, _EgnnPackWeights, egnn_pack_weights, egnn_edge, _EgnnNodeMlp, egnn_node_mlp,
    egnn_node_mlp_supported, egnn_node_mlp_ln, NODE_LN_FOLD,
)
from .readout import (  # noqa: F401
    _MseLoss, mse_loss, _READOUT_STATE, _readout_state, _ReadoutMse, readout_mse_supported, readout_mse,
)
from .se3 import (  # noqa: F401
    _RowGemm, _RowGemm2, _RadialWeightLayout, radial_weight_layout, _AttnPool, attn_pool_supported, attn_pool,
    _RmsNormRows, rms_norm_rows, _RadialTrunk, radial_trunk_supported, radial_trunk, rowgemm, rowgemm2, rowgemm_bias_supported,
    edge_geometry, _RowOuter, row_outer, _PooledRadial, pooled_radial, _Pool3, pool3,
)
from .frames import (  # noqa: F401
    _dropout_seed, _SwigluDropout, _DropoutMean, _FramePre, _FrameHidden, _EdgeHidden, edge_hidden, _RowDot,
    rowdot, rowdot_supported, _GateRows, gate_rows, _AttnSum, attn_sum_supported, attn_sum, frame_pre,
    frame_hidden, swiglu_dropout, dropout_mean, eigh3, linear_dropout_mean, linear_dropout_mean_supported,
    _CentreMix, centre_mix, _CloudFrame, cloud_frame, _EdgeFrame, edge_frame, edge_frame_supported, _AttnLogits,
    attn_logits, attn_logits_supported, geom_supported, dropout_seeds, _AttnGatherSum, attn_gather_sum,
    attn_gather_sum_supported, _EdgeLogitWeights, edge_logit_weights, _LnRowDot, ln_rowdot, ln_rowdot_supported,
    _DropoutAdd, dropout_add, dropout_add_supported,
)

# switches that tests / tools / bench.py set as ``ops.NAME = value``: owner module of each
_SWITCH_OWNER = {
    "DEFER_WGRAD": "grads",
    "GEMM_TILE": "products",
    "KNN_GRID_MIN_POINTS": "aggregate",
    "TIMELINE": "_base",
    "SIGNAL": "_base",
    "NODE_LN_FOLD": "egnn",
    "USE_GEOM": "frames",
    "USE_WGRAD_KERNEL": "grads",
    "USE_X6": "products",
    "WGRAD_ON_SIDE_STREAM": "grads",
    "X6_DEEP_ROWS": "products",
    "X6_MAX_K": "products",
    "X6_MIN_OUTPUTS": "products",
    "X6_WGRAD_OUTPUTS": "products",
    "X6_WGRAD_ROWS": "products",
}


# (module objects by name: the package attributes ``linear`` etc. are the re-exported FUNCTIONS)
_SUBMODULES = tuple(_importlib.import_module(f"{__name__}.{_n}") for _n in (
    "_base", "aggregate", "products", "grads", "linears", "rows", "egnn", "readout", "se3", "frames"))


class _OpsModule(_types.ModuleType):
    """``ops.NAME = value`` for a switch reaches the submodule whose functions read it."""

    def __setattr__(self, name, value):
        if name in _SWITCH_OWNER:       # the owner and every submodule that imported the name
            for mod in _SUBMODULES:
                if hasattr(mod, name):
                    setattr(mod, name, value)
        super().__setattr__(name, value)


_sys.modules[__name__].__class__ = _OpsModule

import os as _os  # noqa: E402

USE_NODE_PANEL = not _os.environ.get("EQH_NO_NODE_PANEL")     # the EGNN node update on the panel kernels (tests switch it off to compare with the per-operator path)
