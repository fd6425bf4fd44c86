"""Model classes registered under the reference's ``--method`` names.

Constructor and call contract of main.py:28-34,46-47: ``cls(num_target, args)``;
``model(data) -> float32 Tensor[B]`` on ``data.x.device``.  Parameter / buffer names and shapes are
identical to the reference's, so its checkpoints load with ``strict=True``.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .equiformer import Equiformer
from .faformer import FAFormer
from .index import HyperIndex
from .layers import (EGNN, MLP, AtomEncoder, BondEncoder, MHNNConv, MHNNSConv, batch_norm_rows, head_loss, pool_sum, readout,
                     real_row_mask)
from .registry import registry

_ACT = {"Id": nn.Identity, "relu": nn.ReLU, "prelu": nn.PReLU}


def _conv_layers(model, x, index, x0, res, fuse_act, taps):
    """The wrappers' loop over the shared conv layer (equihnn_egnn.py:160-165, mhnn.py:208-212): conv -> activation ->
    dropout, ``All_num_layers`` times.  With the merged residual, an inactive dropout and the ReLU fused, all applications
    run as ONE autograd node on the row-panel kernels (MHNNSConv.forward_stack)."""
    drop_off = not (model.dropout.training and model.dropout.p > 0)
    if model.nlayer >= 1 and fuse_act and taps is None and drop_off and model.conv.stack_supported(x, res):
        return model.conv.forward_stack(x, index, res, model.nlayer, relu_out=True)
    for i in range(model.nlayer):
        x = model.conv(model.dropout(x), index, x0, res, relu_out=fuse_act)
        if taps is not None:
            taps[f"conv{i}"] = x
        if not fuse_act:
            x = model.act(x)
    return x


@registry.register_model("egnn_equihnns")
class EGNNEquiHNNS(nn.Module):
    """equihnn_egnn.py:98-169: AtomEncoder -> EGNN (once) -> shared MHNNSConv x L -> pool -> head."""

    def __init__(self, num_target, args):
        super().__init__()
        self.act = _ACT[args.activation]()
        self.dropout = nn.Dropout(args.dropout)
        self.mlp1_layers = args.MLP1_num_layers
        self.mlp2_layers = args.MLP2_num_layers
        self.mlp3_layers = args.MLP3_num_layers
        self.nlayer = args.All_num_layers
        self.atom_encoder = AtomEncoder(emb_dim=args.MLP_hidden)
        self.egnn_layer = EGNN(dim=args.MLP_hidden, num_nearest_neighbors=16)
        self.conv = MHNNSConv(args.MLP_hidden, mlp1_layers=self.mlp1_layers,
                              mlp2_layers=self.mlp2_layers, mlp3_layers=self.mlp3_layers,
                              aggr=args.aggregate, dropout=args.dropout,
                              normalization=args.normalization)
        self.mlp_out = MLP(in_channels=args.MLP_hidden, hidden_channels=args.output_hidden,
                           out_channels=num_target, num_layers=args.output_num_layers,
                           dropout=args.dropout, Normalization=args.normalization, InputNorm=False)

    def reset_parameters(self):  # equihnn_egnn.py:151-153
        self.conv.reset_parameters()
        self.mlp_out.reset_parameters()

    def forward(self, data, taps=None, head=None):
        index = HyperIndex.from_batch(data)
        x = self.atom_encoder(data.x)
        if taps is not None:
            taps["atom_encoder"] = x
        x = self.egnn_layer(x, data.pos, index)
        if taps is not None:
            taps["front_end"] = x
        x0 = x
        res = self.conv.prepare(x0, index)   # layer-independent residual term, built once
        fuse_act = taps is None and isinstance(res, dict) and isinstance(self.act, nn.ReLU)   # ReLU in the GEMM epilogue
        x = _conv_layers(self, x, index, x0, res, fuse_act, taps)
        return readout(self.mlp_out, self.dropout(x), index, taps, head)


@registry.register_model("mhnnm")
class MHNNM(nn.Module):
    """mhnn.py:144-218: L unshared MHNNConv layers + BatchNorm1d on node rows.  The host-side
    per-molecule ``.item()`` loop of mhnn.py:196-199 builds a tensor the model never uses; it is
    not reproduced (it costs B device syncs per step)."""

    def __init__(self, num_target, args):
        super().__init__()
        self.act = _ACT[args.activation]()
        self.dropout = nn.Dropout(args.dropout)
        self.mlp1_layers = args.MLP1_num_layers
        self.mlp2_layers = args.MLP2_num_layers
        self.mlp3_layers = args.MLP3_num_layers
        self.mlp4_layers = args.MLP4_num_layers
        self.nlayer = args.All_num_layers
        self.atom_encoder = AtomEncoder(emb_dim=args.MLP_hidden)
        self.bond_encoder = BondEncoder(6, args.MLP_hidden)
        self.layers = nn.ModuleList()
        self.batch_norms = nn.ModuleList()
        for _ in range(self.nlayer):
            self.layers.append(MHNNConv(args.MLP_hidden, mlp1_layers=self.mlp1_layers,
                                        mlp2_layers=self.mlp2_layers, mlp3_layers=self.mlp3_layers,
                                        mlp4_layers=self.mlp4_layers, aggr=args.aggregate,
                                        dropout=args.dropout, normalization=args.normalization))
            self.batch_norms.append(nn.BatchNorm1d(args.MLP_hidden))
        self.mlp_out = MLP(in_channels=args.MLP_hidden, hidden_channels=args.output_hidden,
                           out_channels=num_target, num_layers=args.output_num_layers,
                           dropout=args.dropout, Normalization=args.normalization, InputNorm=False)

    def forward(self, data, taps=None, head=None):
        index = HyperIndex.from_batch(data)
        x = self.atom_encoder(data.x)
        e = self.bond_encoder(data.edge_attr)
        if taps is not None:
            taps["atom_encoder"] = x
        mask = real_row_mask(data, x)   # padded batch: BatchNorm statistics over the real atoms only
        with _merged_scope(self.layers, x, e):      # the layers' weight-level products in one launch each way
            for i, layer in enumerate(self.layers):
                x, e = layer(x, e, index)
                # (the ReLU behind the normalisation rides its launches, unless the pre-activation value is tapped)
                fuse = taps is None and i != self.nlayer - 1 and isinstance(self.act, nn.ReLU)
                x = batch_norm_rows(self.batch_norms[i], x, mask, relu=fuse)
                if taps is not None:
                    taps[f"bn{i}"] = x
                if i != self.nlayer - 1:  # no activation after the last layer, mhnn.py:208-214
                    x, e = (x if fuse else self.act(x)), self.act(e)
                x, e = self.dropout(x), self.dropout(e)
        return readout(self.mlp_out, x, index, taps, head)


def _merged_scope(convs, x, e):
    """ops.merged_scope on the GPU (the panel path of MHNNConv); a no-op context elsewhere."""
    import contextlib
    if x.is_cuda:
        from . import ops
        return ops.merged_scope(convs, x, e)
    return contextlib.nullcontext()


@registry.register_model("equiformer_equihnns")
class EquiformerEquiHNNS(nn.Module):
    """equihnn_equiformer.py:12-93: AtomEncoder -> Equiformer (once, type-0 output) -> shared
    MHNNSConv x L -> pool -> head.  The reference keeps a leading batch dim of 1 through the conv /
    pool / head (equihnn_equiformer.py:82-85) and flattens at the end; values are identical."""

    def __init__(self, num_target, args):
        super().__init__()
        self.act = _ACT[args.activation]()
        self.dropout = nn.Dropout(args.dropout)
        self.mlp1_layers = args.MLP1_num_layers
        self.mlp2_layers = args.MLP2_num_layers
        self.mlp3_layers = args.MLP3_num_layers
        self.nlayer = args.All_num_layers
        self.atom_encoder = AtomEncoder(emb_dim=args.MLP_hidden)
        self.equiformer_layer = Equiformer(dim=args.MLP_hidden, dim_head=48, num_neighbors=16,
                                           valid_radius=5.0)
        self.conv = MHNNSConv(args.MLP_hidden, mlp1_layers=self.mlp1_layers,
                              mlp2_layers=self.mlp2_layers, mlp3_layers=self.mlp3_layers,
                              aggr=args.aggregate, dropout=args.dropout,
                              normalization=args.normalization)
        self.mlp_out = MLP(in_channels=args.MLP_hidden, hidden_channels=args.output_hidden,
                           out_channels=num_target, num_layers=args.output_num_layers,
                           dropout=args.dropout, Normalization=args.normalization, InputNorm=False)

    def reset_parameters(self):
        self.conv.reset_parameters()
        self.mlp_out.reset_parameters()

    def forward(self, data, taps=None, head=None):
        index = HyperIndex.from_batch(data)
        x = self.atom_encoder(data.x)
        if taps is not None:
            taps["atom_encoder"] = x
        x = self.equiformer_layer(x, data.pos, index)
        if taps is not None:
            taps["front_end"] = x
        x0 = x
        res = self.conv.prepare(x0, index)   # layer-independent residual term, built once
        fuse_act = taps is None and isinstance(res, dict) and isinstance(self.act, nn.ReLU)   # ReLU in the GEMM epilogue
        x = _conv_layers(self, x, index, x0, res, fuse_act, taps)
        return readout(self.mlp_out, self.dropout(x), index, taps, head)


class _PairedBase(nn.Module):
    """``mhnn`` (mhnn.py:11-81) and ``egnn_equihnn`` (equihnn_egnn.py:12-95): ONE shared MHNNConv
    applied L times; nodes and hyperedges of order > 2 are pooled per molecule and concatenated.
    The reference sizes the hyperedge pool by ``he_batch.max()+1`` and fails in ``torch.cat`` when the
    last molecules of a batch have no such hyperedge; here they get a zero row."""

    def __init__(self, num_target, args, with_egnn: bool):
        super().__init__()
        self.act = _ACT[args.activation]()
        self.dropout = nn.Dropout(args.dropout)
        self.mlp1_layers = args.MLP1_num_layers
        self.mlp2_layers = args.MLP2_num_layers
        self.mlp3_layers = args.MLP3_num_layers
        self.mlp4_layers = args.MLP4_num_layers
        self.nlayer = args.All_num_layers
        if with_egnn:
            self.egnn_layer = EGNN(dim=args.MLP_hidden, num_nearest_neighbors=16)
        self.atom_encoder = AtomEncoder(emb_dim=args.MLP_hidden)
        self.bond_encoder = BondEncoder(6, args.MLP_hidden)
        self.conv = MHNNConv(args.MLP_hidden, mlp1_layers=self.mlp1_layers, mlp2_layers=self.mlp2_layers,
                             mlp3_layers=self.mlp3_layers, mlp4_layers=self.mlp4_layers,
                             aggr=args.aggregate, dropout=args.dropout, normalization=args.normalization)
        self.mlp_out = MLP(in_channels=args.MLP_hidden * 2, hidden_channels=args.output_hidden * 2,
                           out_channels=num_target, num_layers=args.output_num_layers,
                           dropout=args.dropout, Normalization=args.normalization, InputNorm=False)
        self.with_egnn = with_egnn

    def forward(self, data, taps=None, head=None):
        from . import ops
        index = HyperIndex.from_batch(data)
        x = self.atom_encoder(data.x)
        if self.with_egnn:
            x = self.egnn_layer(x, data.pos, index)
            if taps is not None:
                taps["front_end"] = x
        e = self.bond_encoder(data.edge_attr)
        with _merged_scope([self.conv], x, e):      # the shared layer's weight-level products once per step
            for i in range(self.nlayer):
                x, e = self.conv(x, e, index)
                if i != self.nlayer - 1:
                    x, e = self.act(x), self.act(e)
                x, e = self.dropout(x), self.dropout(e)
        xp = pool_sum(x, index)
        he_csr, he_key = index.hyperedge_pool(data.n_e)
        keep = (data.e_order > 2).to(e.dtype).unsqueeze(-1)            # mhnn.py:58,72
        ep = ops.reduce_entries(e * keep, he_csr, he_key, "sum")
        both = torch.cat((xp, ep), -1)
        if taps is not None:
            taps["pool"] = both
        return head_loss(self.mlp_out(both, mask=index.pad_masks()[3]).view(-1), head)


@registry.register_model("mhnn")
class MHNN(_PairedBase):
    def __init__(self, num_target, args):
        super().__init__(num_target, args, with_egnn=False)


@registry.register_model("egnn_equihnn")
class EGNNEquiHNN(_PairedBase):
    def __init__(self, num_target, args):
        super().__init__(num_target, args, with_egnn=True)


@registry.register_model("mhnns")
class MHNNS(nn.Module):
    """mhnn.py:84-141: shared MHNNSConv x L on atom embeddings (no geometric front-end)."""

    def __init__(self, num_target, args):
        super().__init__()
        self.act = _ACT[args.activation]()
        self.dropout = nn.Dropout(args.dropout)
        self.mlp1_layers = args.MLP1_num_layers
        self.mlp2_layers = args.MLP2_num_layers
        self.mlp3_layers = args.MLP3_num_layers
        self.nlayer = args.All_num_layers
        self.atom_encoder = AtomEncoder(emb_dim=args.MLP_hidden)
        self.conv = MHNNSConv(args.MLP_hidden, mlp1_layers=self.mlp1_layers, mlp2_layers=self.mlp2_layers,
                              mlp3_layers=self.mlp3_layers, aggr=args.aggregate, dropout=args.dropout,
                              normalization=args.normalization)
        self.mlp_out = MLP(in_channels=args.MLP_hidden, hidden_channels=args.output_hidden,
                           out_channels=num_target, num_layers=args.output_num_layers,
                           dropout=args.dropout, Normalization=args.normalization, InputNorm=False)

    def reset_parameters(self):
        self.conv.reset_parameters()
        self.mlp_out.reset_parameters()

    def forward(self, data, taps=None, head=None):
        index = HyperIndex.from_batch(data)
        x = self.atom_encoder(data.x)
        x0 = x
        res = self.conv.prepare(x0, index)   # layer-independent residual term, built once
        fuse_act = taps is None and isinstance(res, dict) and isinstance(self.act, nn.ReLU)   # ReLU in the GEMM epilogue
        x = _conv_layers(self, x, index, x0, res, fuse_act, taps)
        return readout(self.mlp_out, self.dropout(x), index, taps, head)


@registry.register_model("egnn_equihnnm")
class EGNNEquiHNNM(MHNNM):
    """equihnn_egnn.py:172-261: mhnnm with the EGNN front-end applied once to the atom embeddings."""

    def __init__(self, num_target, args):
        super().__init__(num_target, args)
        self.egnn_layer = EGNN(dim=args.MLP_hidden, num_nearest_neighbors=16)

    def forward(self, data, taps=None, head=None):
        index = HyperIndex.from_batch(data)
        x = self.egnn_layer(self.atom_encoder(data.x), data.pos, index)
        if taps is not None:
            taps["front_end"] = x
        e = self.bond_encoder(data.edge_attr)
        mask = real_row_mask(data, x)   # padded batch: BatchNorm statistics over the real atoms only
        for i, layer in enumerate(self.layers):
            x, e = layer(x, e, index)
            fuse = taps is None and i != self.nlayer - 1 and isinstance(self.act, nn.ReLU)
            x = batch_norm_rows(self.batch_norms[i], x, mask, relu=fuse)
            if taps is not None:
                taps[f"bn{i}"] = x
            if i != self.nlayer - 1:
                x, e = (x if fuse else self.act(x)), self.act(e)
            x, e = self.dropout(x), self.dropout(e)
        return readout(self.mlp_out, x, index, taps, head)


@registry.register_model("faformer_equihnns")
class FAFormerEquiHNNS(nn.Module):
    """equihnn_fa_former.py:105-184: AtomEncoder -> FAFormer (once) -> shared MHNNSConv x L -> pool
    -> head.  Note the reference keeps proj_drop = attn_drop = 0.1 inside FAFormer in training mode
    (fa_former_layer.py:20-21), whatever ``--dropout`` says."""

    def __init__(self, num_target, args):
        super().__init__()
        self.act = _ACT[args.activation]()
        self.dropout = nn.Dropout(args.dropout)
        self.mlp1_layers = args.MLP1_num_layers
        self.mlp2_layers = args.MLP2_num_layers
        self.mlp3_layers = args.MLP3_num_layers
        self.nlayer = args.All_num_layers
        self.atom_encoder = AtomEncoder(emb_dim=args.MLP_hidden)
        self.fa_former = FAFormer(args.MLP_hidden, n_layers=2, n_heads=2, n_neighbors=16, valid_radius=5.0)
        self.conv = MHNNSConv(args.MLP_hidden, mlp1_layers=self.mlp1_layers, mlp2_layers=self.mlp2_layers,
                              mlp3_layers=self.mlp3_layers, aggr=args.aggregate, dropout=args.dropout,
                              normalization=args.normalization)
        self.mlp_out = MLP(in_channels=args.MLP_hidden, hidden_channels=args.output_hidden,
                           out_channels=num_target, num_layers=args.output_num_layers,
                           dropout=args.dropout, Normalization=args.normalization, InputNorm=False)

    def reset_parameters(self):
        self.conv.reset_parameters()
        self.mlp_out.reset_parameters()

    def forward(self, data, taps=None, head=None):
        index = HyperIndex.from_batch(data)
        x = self.atom_encoder(data.x)
        if taps is not None:
            taps["atom_encoder"] = x
        x = self.fa_former(x, data.pos, index, real_row_mask(data, x))
        if taps is not None:
            taps["front_end"] = x
        x0 = x
        res = self.conv.prepare(x0, index)   # layer-independent residual term, built once
        fuse_act = taps is None and isinstance(res, dict) and isinstance(self.act, nn.ReLU)   # ReLU in the GEMM epilogue
        x = _conv_layers(self, x, index, x0, res, fuse_act, taps)
        return readout(self.mlp_out, self.dropout(x), index, taps, head)


MODELS = {"egnn_equihnns": EGNNEquiHNNS, "mhnnm": MHNNM, "equiformer_equihnns": EquiformerEquiHNNS,
          "faformer_equihnns": FAFormerEquiHNNS,
          "mhnn": MHNN, "mhnns": MHNNS, "egnn_equihnn": EGNNEquiHNN, "egnn_equihnnm": EGNNEquiHNNM}
