"""Layers of the hot path, MI355X-native, with the reference's parameter names and shapes.

The arithmetic is re-derived for the device rather than transcribed:

* A Linear applied to ``cat(a[v], b[e])`` per incidence is split by columns,
  ``W·cat(a,b) = W_a·a + W_b·b``, and evaluated at node / hyperedge level (N + M rows through
  MFMA instead of 2·nnz rows); the per-incidence work becomes a row gather-add.
* The last Linear of a per-incidence MLP commutes with the (linear) mean aggregation, so it runs
  on the aggregated rows: ``mean_r(W·h_p + b) = W·mean_r(h_p) + b·[deg(r) > 0]``.
* All gathers / scatters go through the CSR kernels of libequihgnn_hip.so (ops/aggregate.py).

Dense Linears are plain library GEMMs (fp32 MFMA through hipBLASLt via torch); they are the
"dense per-type linear mixes" of the north star, not the hand-written part.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .batch import ATOM_FEATURE_DIMS
from .index import HyperIndex


def _make_norm(kind: str, width: int) -> nn.Module:
    if kind == "ln":
        return nn.LayerNorm(width)
    if kind == "bn":
        return nn.BatchNorm1d(width)
    if kind == "None":
        return nn.Identity()
    raise AssertionError(f"Normalization must be bn/ln/None, got {kind}")  # mlp.py:27


class AtomEncoder(nn.Module):
    """ogb AtomEncoder (call sites equihnn_egnn.py:121,157): ``atom_embedding_list.{0..8}``."""

    def __init__(self, emb_dim: int):
        super().__init__()
        self.atom_embedding_list = nn.ModuleList()
        offs, run = [], 0
        for d in ATOM_FEATURE_DIMS:
            emb = nn.Embedding(d, emb_dim)
            nn.init.xavier_uniform_(emb.weight.data)
            self.atom_embedding_list.append(emb)
            offs.append(run)
            run += d
        self.offsets = tuple(offs)

    def forward(self, x):
        # the nine weights go in as they are: back to back in the trainer's flat parameter buffer their
        # concatenation is a view, and the backward adds into their (equally contiguous) accumulators
        return ops.embed_sum(x, [e.weight for e in self.atom_embedding_list], self.offsets)


class BondEncoder(nn.Embedding):
    """nn.Embedding(6, C) on ``edge_attr`` (mhnn.py:165,202), through the same kernel."""

    def forward(self, idx):
        return ops.embed_sum(idx.reshape(-1, 1), self.weight, (0,))


class MLP(nn.Module):
    """mlp.py:9-99 — same ``lins`` / ``normalizations`` lists; forward mlp.py:91-99:
    norm0 -> [Linear -> ReLU -> norm -> dropout]* -> Linear (the norm comes after ReLU)."""

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers,
                 dropout=0.5, Normalization="bn", InputNorm=False):
        super().__init__()
        self.lins = nn.ModuleList()
        self.normalizations = nn.ModuleList()
        self.InputNorm = InputNorm
        self.normalizations.append(_make_norm(Normalization, in_channels) if InputNorm else nn.Identity())
        widths = [in_channels] + [hidden_channels] * (num_layers - 1) + [out_channels]
        for i in range(num_layers):
            self.lins.append(nn.Linear(widths[i], widths[i + 1]))
            if i < num_layers - 1:
                self.normalizations.append(_make_norm(Normalization, widths[i + 1]))
        self.dropout = dropout

    def reset_parameters(self):
        for lin in self.lins:
            lin.reset_parameters()
        for n in self.normalizations:
            if not isinstance(n, nn.Identity):
                n.reset_parameters()

    @staticmethod
    def _norm(norm, h, mask):
        """A normalisation of mlp.py:17-60 on rows ``h``; ``mask`` ([rows, 1], 1 = real row) keeps the padded rows of a
        static-shape batch out of BatchNorm's TRAINING statistics (LayerNorm / Identity are row-wise: no mask needed)."""
        if isinstance(norm, nn.BatchNorm1d) and h.dim() == 2:
            return batch_norm_rows(norm, h, mask)
        return norm(h)

    @property
    def has_batch_norm(self) -> bool:
        return any(isinstance(n, nn.BatchNorm1d) for n in self.normalizations)

    def hidden(self, h, start: int, mask=None):
        """ReLU -> norm -> dropout after Linear ``start``, then the hidden Linears up to (not
        including) the last one."""
        last = len(self.lins) - 1
        for i in range(start, last):
            if i > start:
                h = ops.linear(h, self.lins[i].weight, self.lins[i].bias) if h.is_cuda else self.lins[i](h)
            h = F.relu(h)
            h = self._norm(self.normalizations[i + 1], h, mask)
            h = F.dropout(h, p=self.dropout, training=self.training)
        return h

    def takes_first(self, x) -> bool:
        """Whether forward(x, first=...) uses a precomputed first GEMM (hidden layers on the fused path)."""
        return len(self.lins) > 1 and not self.InputNorm and self._fusable(x)

    def _fusable(self, x):
        return (x.is_cuda and x.dim() == 2 and not (self.training and self.dropout > 0)
                and all(isinstance(n, nn.LayerNorm) for n in self.normalizations[1:])
                and all(l.out_features % 4 == 0 and l.out_features <= 1024 for l in self.lins[:-1]))

    def forward(self, x, first=None, mask=None):
        """``first``: x @ lins[0].weight.T already computed by the caller (bias-free; only on the fused path,
        see ``takes_first``).  ``mask``: [rows, 1] real-row mask of a padded batch (BatchNorm statistics, see _norm)."""
        x = self._norm(self.normalizations[0], x, mask)          # InputNorm (mlp.py:31-58): LayerNorm / BatchNorm of the input
        if len(self.lins) == 1:
            return ops.linear(x, self.lins[0].weight, self.lins[0].bias)
        if self._fusable(x):
            # bias-free GEMM, then bias + ReLU + LayerNorm in one launch (its backward also yields
            # the bias gradient, so no separate column-sum kernel runs)
            for i in range(len(self.lins) - 1):
                lin, norm = self.lins[i], self.normalizations[i + 1]
                h = first if (i == 0 and first is not None) else ops.linear(x, lin.weight)
                x = ops.bias_relu_ln(h, lin.bias, norm.weight, norm.bias, norm.eps)
            return ops.linear(x, self.lins[-1].weight, self.lins[-1].bias)
        h = self.hidden(ops.linear(x, self.lins[0].weight, self.lins[0].bias), 0, mask)
        return ops.linear(h, self.lins[-1].weight, self.lins[-1].bias)


def _row_weight(out_csr, has_row, aggr, dtype):
    """Per-row multiplier of the last Linear's bias after the reduction: 1 (mean over a non-empty row),
    0 (empty row) or the row length (sum)."""
    if aggr == "mean":
        return has_row
    return (out_csr.rowptr[1:] - out_csr.rowptr[:-1]).to(dtype)[:, None]


def _pair_message(mlp: MLP, a, b, idx_a32, idx_b32, csr_a, csr_b, out_csr, out_key32, has_row, aggr,
                  residual=None, pa=None, inc_mask=None):
    """reduce_r( mlp(cat(a[idx_a], b[idx_b])) ) over the rows of ``out_csr`` — the
    per-incidence MLP + scatter of conv.py:90-93,96-97,175-177, restructured (module docstring).

    a is indexed by idx_a32 (CSR keyed by that index: csr_a), b by idx_b32 (csr_b)."""
    if mlp.InputNorm or mlp.has_batch_norm:
        # A normalisation of the CONCATENATED per-incidence input (InputNorm: mlp.py:31-58 over 2C channels), or BatchNorm
        # statistics over the incidences: the split-weight restructuring does not apply -- the MLP runs on the [nnz, 2C] rows
        # as conv.py:90,96,176 writes it (null incidences of a padded batch are zero rows, kept out of the statistics by
        # ``inc_mask`` and out of the reduction by the CSR)
        h = torch.cat((ops.gather_rows(a, idx_a32, csr_a), ops.gather_rows(b, idx_b32, csr_b)), -1)
        out = ops.reduce_entries(mlp(h, mask=inc_mask), out_csr, out_key32, aggr)
        return out if residual is None else residual[0] * out + residual[2]
    lin0 = mlp.lins[0]
    ca = a.shape[-1]
    cin = lin0.weight.shape[1]
    if pa is None:                                              # (else: computed by the caller with another GEMM of a)
        pa = ops.linear(a, lin0.weight, None, (0, ca))          # rows of a
    qb = ops.linear(b, lin0.weight, lin0.bias, (ca, cin))       # rows of b
    norm = mlp.normalizations[1] if len(mlp.lins) > 1 else None
    fused = (len(mlp.lins) == 2 and isinstance(norm, nn.LayerNorm) and pa.dim() == 2
             and pa.shape[-1] % 4 == 0 and pa.shape[-1] <= 1024 and not (mlp.training and mlp.dropout > 0))
    last = mlp.lins[-1]
    if fused:
        # gather + gather + add + ReLU + LayerNorm + segmented reduce in ONE kernel (incidence.hip)
        s = ops.incidence_ln_reduce(pa, qb, norm.weight, norm.bias, idx_a32, idx_b32, csr_a, csr_b,
                                    out_csr, out_key32, aggr, norm.eps)
        if residual is not None:   # (scale, c): scale * last(s) + c, c already holds the scaled bias
            return ops.linear_add(s, last.weight, residual[1], residual[0])
        return torch.addcmul(ops.linear(s, last.weight), _row_weight(out_csr, has_row, aggr, s.dtype), last.bias)
    h = ops.gather_rows(pa, idx_a32, csr_a) + ops.gather_rows(qb, idx_b32, csr_b)  # [nnz, C]
    if len(mlp.lins) == 1:              # a single Linear: everything is linear in h
        out = ops.reduce_entries(h, out_csr, out_key32, aggr)
        return out if residual is None else residual[0] * out + residual[2]
    h = mlp.hidden(h, 0)
    s = ops.reduce_entries(h, out_csr, out_key32, aggr)
    if residual is not None:
        return ops.linear_add(s, last.weight, residual[1], residual[0])
    return ops.linear(s, last.weight) + last.bias * _row_weight(out_csr, has_row, aggr, s.dtype)


class MHNNConv(nn.Module):
    """conv.py:8-101 (node and hyperedge features, W1..W4)."""

    def __init__(self, hid_dim, mlp1_layers=1, mlp2_layers=1, mlp3_layers=1, mlp4_layers=1,
                 aggr="mean", dropout=0.0, normalization="None", input_norm=False):
        super().__init__()
        # mlpK_layers = 0 (conv.py:33-34,45-46,57-58,69-70): W_k is `lambda X: X[..., hid_dim:]`, the SECOND half of its
        # concatenated input, and owns no parameters -- kept as None here
        mk = lambda n: (MLP(hid_dim * 2, hid_dim, hid_dim, n, dropout=dropout, Normalization=normalization, InputNorm=input_norm)
                        if n > 0 else None)
        self.W1, self.W2, self.W3, self.W4 = mk(mlp1_layers), mk(mlp2_layers), mk(mlp3_layers), mk(mlp4_layers)
        self.aggr = aggr
        self.dropout = dropout

    def reset_parameters(self):
        for w in (self.W1, self.W2, self.W3, self.W4):
            if w is not None:
                w.reset_parameters()

    def forward(self, X, E, index: HyperIndex):
        ix = index
        if ops.mhnn_panel_supported(X, E, self):
            return ops.mhnn_conv_panel(self, X, E, ix)      # the scripts' configuration: one autograd node on the panel kernels
        mk = ix.pad_masks()                 # (node, hyperedge, incidence) real-row masks of a padded batch, or Nones
        if self.W1 is None:                 # the message IS the hyperedge's own row: its mean over the incidences is itself
            m_e = ops.reduce_entries(ops.gather_rows(E, ix.e32, ix.by_e), ix.by_e, ix.e32, self.aggr)
        else:
            m_e = _pair_message(self.W1, X, E, ix.v32, ix.e32, ix.by_v, ix.by_e, ix.by_e, ix.e32,
                                ix.has_e, self.aggr, inc_mask=mk[2])  # conv.py:90-93
        E = m_e if self.W2 is None else self.W2(torch.cat((E, m_e), -1), mask=mk[1])   # conv.py:94
        if self.W3 is None:
            m_v = ops.reduce_gathered(E, ix.by_v, ix.by_e, self.aggr)
        else:
            m_v = _pair_message(self.W3, X, E, ix.v32, ix.e32, ix.by_v, ix.by_e, ix.by_v, ix.v32,
                                ix.has_v, self.aggr, inc_mask=mk[2])  # conv.py:96-97
        X = m_v if self.W4 is None else self.W4(torch.cat((X, m_v), -1), mask=mk[0])   # conv.py:98
        return X, E


def real_row_mask(data, like):
    """[N, 1] float mask of the rows that belong to real molecules of a padded batch (batch.pad_batch gives
    all padded atoms to one extra molecule, id = num_real_graphs), or None for an unpadded batch."""
    nb = getattr(data, "num_real_graphs", None)
    if nb is None or nb >= data.y.shape[0]:
        return None
    return (data.batch < nb).to(like.dtype).unsqueeze(-1)


def batch_norm_rows(bn: nn.BatchNorm1d, x, mask, relu: bool = False):
    """nn.BatchNorm1d over node rows (mhnn.py:182,206) whose TRAINING statistics count the real rows only:
    (``relu``: the ReLU mhnn.py:208-214 applies behind it, in the same launches) with it a padded batch gives the real rows the same outputs, the parameters the same gradients and the
    running buffers the same updates as the unpadded one, so the BatchNorm models can run under hipGraph
    replay too.  Everything stays on the device (the row count is a device scalar: it changes per batch)."""
    if ops.batch_norm_rows_supported(x, bn):       # two launches each way (csrc/bn_rows.hip), masked or not
        return ops.batch_norm_rows(x, mask, bn, relu=relu)
    if relu:
        return torch.relu(batch_norm_rows(bn, x, mask))
    if mask is None or not bn.training or not bn.track_running_stats or bn.momentum is None:
        return bn(x)
    n = mask.sum()
    mean = (x * mask).sum(0) / n
    xc = x - mean
    var = (xc * xc * mask).sum(0) / n                      # biased, as F.batch_norm normalises with
    y = xc * torch.rsqrt(var + bn.eps) * bn.weight + bn.bias
    with torch.no_grad():
        bn.running_mean.lerp_(mean, bn.momentum)
        bn.running_var.lerp_(var * (n / (n - 1.0)), bn.momentum)   # unbiased, as nn.BatchNorm1d stores
        bn.num_batches_tracked.add_(1)
    return y


MERGE_LINEARS = True   # MHNNSConv: fold Linear -> (linear map) -> Linear pairs into one Linear (see _prepare_merged)


class MHNNSConv(nn.Module):
    """conv.py:104-182 (node features only, W1..W3, alpha residual to X0)."""

    def __init__(self, hid_dim, mlp1_layers=1, mlp2_layers=1, mlp3_layers=1, aggr="mean",
                 alpha=0.5, dropout=0.0, normalization="None", input_norm=False):
        super().__init__()
        # mlpK_layers = 0: W1 = Identity (conv.py:128-130), W2 = the second half of its input (:142-143); for W3 the reference
        # sets ``self.W`` (:155-156), not ``self.W3``, so its forward() fails with AttributeError at :180 -- reproduced
        mk = lambda cin, n: (MLP(cin, hid_dim, hid_dim, n, dropout=dropout, Normalization=normalization, InputNorm=input_norm)
                             if n > 0 else None)
        self.W1, self.W2 = mk(hid_dim, mlp1_layers), mk(hid_dim * 2, mlp2_layers)
        if mlp3_layers > 0:
            self.W3 = mk(hid_dim, mlp3_layers)
        else:
            self.W = nn.Identity()
        self.aggr = aggr
        self.alpha = alpha
        self.dropout = dropout

    def _mlps(self):
        return [w for w in (self.W1, self.W2, getattr(self, "W3", None)) if w is not None]

    @property
    def plain(self) -> bool:
        """Configurations outside the scripts' (a missing MLP, InputNorm, BatchNorm inside the MLPs): the layer runs as
        conv.py:169-182 writes it instead of on the restructured / merged / panel paths."""
        ws = (self.W1, self.W2, getattr(self, "W3", None))
        return any(w is None for w in ws) or any(w.InputNorm or w.has_batch_norm for w in ws)

    def reset_parameters(self):
        for w in self._mlps():
            w.reset_parameters()

    def residual(self, X0, index: HyperIndex, passthrough: bool = False):
        """The layer-independent part of conv.py:179-180's mix, built once per forward pass:
        (1-a) * x_v + a * X0 with x_v = W2_last(s) + rows * b  ==  (1-a) * W2_last_nobias(s) + c,
        c = a * X0 + (1-a) * rows * b.  ``c`` enters the last GEMM of W2 as its beta = 1 operand, so
        neither the bias broadcast nor the lerp is a kernel of its own (nor are their backward passes:
        the weights are shared, so c is the same tensor in all L applications of the layer).
        Returns (scale, c, c without the bias) for _pair_message."""
        a = self.alpha
        if X0.is_cuda and X0.dim() == 2 and X0.shape[-1] % 4 == 0 and X0.dtype == torch.float32:
            mode = 1 if self.aggr == "mean" else 2     # one launch; backward: one mul + a batched column sum
            if passthrough:
                c, x0p = ops.residual_mix(X0, self.W2.lins[-1].bias, index.by_v.rowptr, mode, a, passthrough=True)
                return (1.0 - a, c, None, x0p)
            return (1.0 - a, ops.residual_mix(X0, self.W2.lins[-1].bias, index.by_v.rowptr, mode, a), None)
        x0a = X0 * a
        if X0.dim() != 2:
            return (1.0 - a, None, x0a)
        rows = _row_weight(index.by_v, index.has_v, self.aggr, X0.dtype)
        return (1.0 - a, torch.addcmul(x0a, rows, self.W2.lins[-1].bias, value=1.0 - a), x0a)

    def prepare(self, X0, index: HyperIndex):
        """``residual`` for forward() when the fused path applies (call once per model forward)."""
        if self.plain:
            return None
        if X0.is_cuda and X0.dim() == 2 and len(self.W2.lins) > 1:
            if self._mergeable(X0):
                # (X0 is also the first application's input: it reaches that application through the residual's autograd node)
                return self._prepare_merged(self.residual(X0, index, passthrough=True), X0)
            return self.residual(X0, index)
        return None

    # -- merged path ------------------------------------------------------------------------------------------------
    # Two places of conv.py:169-182 have two Linears with only a LINEAR map between them:
    #   W1's last Linear -> mean over the hyperedge's nodes (:173) -> the hyperedge half of W2's first Linear (:176), and
    #   W2's last Linear -> alpha-mix with X0 (:179-180) -> W3's first Linear.
    # Each pair is one Linear with the product of the two weights, formed at weight level once per forward pass (the
    # layer's weights are shared by its L applications): 5 instead of 7 [rows x C] x [C x C] products per application
    # and direction, and the alpha-mix's X0 term goes through W3's first Linear once per forward instead of L times.
    #   qb = mean_e(h1n) (W2b W1b)^T + (W2b b1 + b2)          h3 = (1 - a) s (W3a W2c)^T + c W3a^T,  c = a X0 + (1 - a) w_r b2c
    def _mergeable(self, X) -> bool:
        two = all(len(w.lins) == 2 for w in (self.W1, self.W2, self.W3))
        return (MERGE_LINEARS and two and self.aggr == "mean" and X.is_cuda and X.dim() == 2 and X.dtype == torch.float32
                and self.W1._fusable(X) and self.W2._fusable(X) and self.W3._fusable(X)
                and not (self.W1.InputNorm or self.W2.InputNorm or self.W3.InputNorm))

    def _prepare_merged(self, res, X0):
        c_dim = self.W1.lins[0].weight.shape[1]
        (w12, b12), (w23, _) = ops.merged_weights([
            (self.W2.lins[0].weight, self.W1.lins[1].weight, self.W1.lins[1].bias, self.W2.lins[0].bias, (c_dim, 2 * c_dim)),
            (self.W3.lins[0].weight, self.W2.lins[1].weight, None, None, None)])      # one launch each way
        cw = ops.linear(res[1], self.W3.lins[0].weight)                   # (a X0 + (1 - a) w_r b) W3a^T, layer-independent
        cw, fan = ops.fanout(cw)          # its gradient is summed over the L applications by their LayerNorm backward kernels
        return {"scale": res[0], "w12": w12, "b12": b12, "w23": w23, "cw": cw, "fan": fan, "x0": X0, "x0_pass": res[3]}

    def _forward_merged(self, X, ix: HyperIndex, m, relu_out=False):
        c = X.shape[-1]
        if X is m["x0"]:
            X = m["x0_pass"]
        W1, W2, W3 = self.W1, self.W2, self.W3
        h1, pa = ops.linear2(X, W1.lins[0].weight, None, W2.lins[0].weight, (0, c))
        n1, n2, n3 = W1.normalizations[1], W2.normalizations[1], W3.normalizations[1]
        # W1's hidden layer + the mean over the hyperedge's nodes (conv.py:172-173) in one launch each way
        hbar = ops.gather_ln_reduce(h1, W1.lins[0].bias, n1.weight, n1.bias, ix.by_e, ix.by_v, "mean", n1.eps)
        qb = ops.linear(hbar, m["w12"], m["b12"])
        s = ops.incidence_ln_reduce(pa, qb, n2.weight, n2.bias, ix.v32, ix.e32, ix.by_v, ix.by_e, ix.by_v, ix.v32,
                                    "mean", n2.eps)                                            # conv.py:175-177
        # conv.py:179-180 + W3's first Linear and hidden layer: scale * (s Wd^T) + cw + bias -> ReLU -> LayerNorm
        x = ops.linear_add_relu_ln(s, m["w23"], m["cw"], m["scale"], W3.lins[0].bias, n3.weight, n3.bias, n3.eps, fan=m["fan"])
        return ops.linear(x, W3.lins[1].weight, W3.lins[1].bias, relu=relu_out)

    def stack_supported(self, X, residual) -> bool:
        """Whether forward_stack applies: the merged path (``residual`` from prepare()) at a width the panel kernels take."""
        return (isinstance(residual, dict) and not self.plain and not (self.training and self.dropout > 0)
                and ops.conv_stack_supported(X, self.W1.lins[0].weight.shape[0]))

    def forward_stack(self, X, index: HyperIndex, residual, n_layers: int, relu_out: bool):
        """``n_layers`` applications of the layer (the wrappers' loop, equihnn_egnn.py:160-165: conv -> activation ->
        dropout with p = 0) as one autograd node on the row-panel kernels (ops.merged_conv_stack)."""
        m = residual
        if X is m["x0"]:
            X = m["x0_pass"]
        return ops.merged_conv_stack(X, m["cw"], self.W1, self.W2, self.W3, m["w12"], m["b12"], m["w23"], index, n_layers,
                                     m["scale"], relu_out)

    def forward(self, X, index: HyperIndex, X0, residual=None, relu_out=False):
        """``relu_out``: return relu(output) (the wrappers' activation, fused into the last GEMM); only with the
        dict ``residual`` of the merged path."""
        ix = index
        if isinstance(residual, dict):
            return self._forward_merged(X, ix, residual, relu_out)
        assert not relu_out
        if self.plain:
            mk = ix.pad_masks()
            w1x = X if self.W1 is None else self.W1(X, mask=mk[0])
            x_e = ops.reduce_gathered(w1x, ix.by_e, ix.by_v, self.aggr)              # conv.py:172-173
            if self.W2 is None:             # the message is the hyperedge half of the concatenation (conv.py:142-143)
                x_v = ops.reduce_gathered(x_e, ix.by_v, ix.by_e, self.aggr)
            else:
                x_v = _pair_message(self.W2, X, x_e, ix.v32, ix.e32, ix.by_v, ix.by_e, ix.by_v, ix.v32, ix.has_v, self.aggr,
                                    inc_mask=mk[2])
            return self.W3(torch.lerp(x_v, X0, self.alpha), mask=mk[0])     # (AttributeError without W3: conv.py:155-156,180)
        fused = X.is_cuda and X.dim() == 2 and len(self.W2.lins) > 1
        pa = None
        if fused and self.W1.takes_first(X) and not self.W2.InputNorm:
            # X feeds the first Linear of W1 and the node half of W2's first Linear: issued together, so that
            # their two input gradients meet in one accumulating GEMM instead of an add kernel
            c = X.shape[-1]
            h1, pa = ops.linear2(X, self.W1.lins[0].weight, None, self.W2.lins[0].weight, (0, c))
            w1x = self.W1(X, first=h1)
        else:
            w1x = self.W1(X)
        x_e = ops.reduce_gathered(w1x, ix.by_e, ix.by_v, self.aggr)              # conv.py:172-173
        if fused:
            res = residual if residual is not None else self.residual(X0, ix)
            mixed = _pair_message(self.W2, X, x_e, ix.v32, ix.e32, ix.by_v, ix.by_e, ix.by_v, ix.v32,
                                  ix.has_v, self.aggr, residual=res, pa=pa)      # conv.py:175-180
            return self.W3(mixed)
        x_v = _pair_message(self.W2, X, x_e, ix.v32, ix.e32, ix.by_v, ix.by_e, ix.by_v, ix.v32,
                            ix.has_v, self.aggr)                                 # conv.py:175-177
        return self.W3(torch.lerp(x_v, X0, self.alpha))      # (1-a)*x_v + a*X0, conv.py:179-180


class CoorsNorm(nn.Module):
    """egnn_layer.py:71-81 — parameter kept for state_dict compatibility; the coordinate branch
    is dead in this model (equihnn_egnn.py:158 discards the coordinate output)."""

    def __init__(self, scale_init=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.full((1,), float(scale_init)))


class EGNN(nn.Module):
    """egnn_layer.py:145-366 as configured at equihnn_egnn.py:123-129 and called at :158
    (mask=None, edges=None, k=16 incl. self, m_pool="sum").  ``coors_mlp`` / ``coors_norm`` are
    kept as parameters and never receive a gradient, as in the reference."""

    def __init__(self, dim, m_dim=16, num_nearest_neighbors=16, init_eps=1e-3,
                 norm_coors_scale_init=1e-2):
        super().__init__()
        e_in = 2 * dim + 1
        self.dim, self.m_dim, self.k = dim, m_dim, num_nearest_neighbors
        self.edge_mlp = nn.Sequential(nn.Linear(e_in, 2 * e_in), nn.Identity(), nn.SiLU(),
                                      nn.Linear(2 * e_in, m_dim), nn.SiLU())
        self.node_norm = nn.LayerNorm(dim)
        self.coors_norm = CoorsNorm(norm_coors_scale_init)
        self.node_mlp = nn.Sequential(nn.Linear(dim + m_dim, 2 * dim), nn.Identity(), nn.SiLU(),
                                      nn.Linear(2 * dim, dim))
        self.coors_mlp = nn.Sequential(nn.Linear(m_dim, 4 * m_dim), nn.Identity(), nn.SiLU(),
                                       nn.Linear(4 * m_dim, 1))
        for mod in self.modules():  # egnn_layer.py:227-230
            if type(mod) is nn.Linear:
                nn.init.normal_(mod.weight, std=init_eps)

    def forward(self, feats, coors, index: HyperIndex):
        c = self.dim
        nbr, d2, csr_t = index.knn(coors, self.k, 0)
        lin1, lin2 = self.edge_mlp[0], self.edge_mlp[3]
        # hidden width H = 2(2C+1) is padded with zero rows to a multiple of 64 (the edge kernels walk
        # 64 hidden units per step); silu(0) = 0, so the padding contributes nothing
        hdim = lin1.weight.shape[0]
        hp = hdim + (-hdim) % 64
        # one launch re-lays the edge-MLP weights out as the kernels want them: w_cat = [W1_i ; W1_j],
        # b_cat = [b1 ; 0], wd = W1[:, 2C], W2 padded (ops.egnn_pack_weights, csrc/dense_aux.hip)
        w_cat, b_cat, w_d, w2 = ops.egnn_pack_weights(lin1.weight, lin1.bias, lin2.weight, hp)
        # one node-level GEMM gives both halves: ab[:, :Hp] = W1_i f + b1 (receiver), ab[:, Hp:] = W1_j f.  The
        # features feed three things (this GEMM, node_norm, the residual): one autograd node, so that their
        # gradients are summed inside the LayerNorm backward kernel and an accumulating GEMM (ops._EgnnFeats)
        nn_ = self.node_norm
        n0, n3 = self.node_mlp[0], self.node_mlp[3]
        # node_norm inside the node update's panel launches (round 6: its forward and backward were launches of their own)
        fold = (ops.NODE_LN_FOLD and ops.USE_NODE_PANEL and feats.dim() == 2 and feats.is_cuda and feats.dtype == torch.float32
                and tuple(nn_.normalized_shape) == (c,) and nn_.elementwise_affine and nn_.bias is not None
                and ops.panel_supported(c) and tuple(n0.weight.shape) == (2 * c, c + 16) and tuple(n3.weight.shape) == (c, 2 * c)
                and feats.shape[0] > 0 and self.edge_mlp[3].weight.shape[0] == 16)
        if fold:
            ab, normed, res = ops.egnn_feats(feats, w_cat, b_cat, None)
        elif feats.dim() == 2 and c % 4 == 0 and c <= 1024:
            ab, normed, res = ops.egnn_feats(feats, w_cat, b_cat, nn_)
        else:
            ab, normed, res = ops.linear(feats, w_cat, b_cat), nn_(feats), feats
        # egnn_layer.py:298-310,357-358 fused: gather, +, SiLU, 16 x Hp x 16 MFMA, SiLU, sum over j
        m_i = ops.egnn_edge(ab, w_d, w2, lin2.bias, nbr, d2, csr_t)
        # everything up to here fills the chip; the panel kernels that follow occupy ~150 of 256 CUs: where a graphed
        # trainer lets the next batch's index build start on its side stream (no-op otherwise)
        ops.signal_point()
        if fold:
            return ops.egnn_node_mlp_ln(normed, m_i, n0, n3, nn_)      # (normed is feats here: one launch each way, LayerNorm included)
        if ops.USE_NODE_PANEL and ops.egnn_node_mlp_supported(normed, m_i, n0, n3):
            return ops.egnn_node_mlp(normed, m_i, res, n0, n3)      # one launch each way (csrc/panel.hip)
        node_in = torch.cat((normed, m_i), -1)
        hid = F.silu(ops.linear(node_in, n0.weight, n0.bias))
        return ops.linear(hid, n3.weight, n3.bias) + res       # egnn_layer.py:360-362


def pool_sum(x, index: HyperIndex):
    """global_add_pool (equihnn_egnn.py:167, mhnn.py:216): per-molecule sum over sorted rows."""
    return ops.reduce_entries(x, index.pool, index.batch32, "sum")


def head_loss(out, head):
    """``out`` itself, or with head = (target, n_real[, unit_grad]) the training loss of main.py:49-63:
    F.mse_loss over the first n_real molecules (the rest of a padded batch is padding)."""
    if head is None:
        return out
    nb = head[1] or out.shape[0]
    return ops.mse_loss(out[:nb], head[0][:nb])


def readout(mlp_out, x, index: HyperIndex, taps=None, head=None):
    """The tail of every wrapper: global_add_pool -> output MLP -> .view(-1) (equihnn_egnn.py:167-169,
    mhnn.py:216-218, equihnn_equiformer.py:91-93).  With ``head`` the training loss is returned instead, and
    when the head has the scripts' shape (three Linears, LayerNorm) pool, MLP, loss and the whole backward
    pass of the head are one launch (ops.readout_mse)."""
    if head is not None and taps is None:
        ops.signal_point("readout")
        x2 = x.reshape(-1, x.shape[-1])
        if ops.readout_mse_supported(x2, mlp_out) and head[0].is_cuda:
            loss, _ = ops.readout_mse(x2, index.pool.rowptr, mlp_out, head[0], head[1],
                                      unit_grad=bool(head[2]) if len(head) > 2 else False)
            ops.signal_point("after_readout")
            return loss
    xp = pool_sum(x, index)
    if taps is not None:
        taps["pool"] = xp
    return head_loss(mlp_out(xp, mask=index.pad_masks()[3]).view(-1), head)     # (mask: BatchNorm in the head over real molecules)
