"""ORACLE — test infrastructure, not product code.

CPU restatement (plain PyTorch fp32 ops, no PyG / torch_scatter / ogb) of the
reference's hot path for ``--method {mhnnm, egnn_equihnns}``: AtomEncoder ->
geometric front-end -> node<->hyperedge message passing -> pooling -> head.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this package; ``equihgnn_amd`` never does.

Pinned (parity is NOT unpinned): ``tests/golden/make_golden.py`` imports the reference's
own model files from /root/reference (with container-only stand-ins for the absent
third-party packages), runs them on seeded synthetic batches and
commits inputs + outputs + gradients under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks this restatement against them to 1e-5.

Every module keeps the reference's parameter names and shapes, so a reference
``state_dict`` loads with ``strict=True``.  File:line citations are relative to
/root/reference.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

ATOM_FEATURE_DIMS = (119, 5, 12, 12, 10, 6, 6, 2, 2)  # ogb 1.3.6 features.py


# --------------------------------------------------------------------------------------
# third-party operators restated (source not under /root/reference; semantics from the
# packages' documentation, call sites cited)
# --------------------------------------------------------------------------------------
def segment_reduce(src: torch.Tensor, index: torch.Tensor, dim_size=None, reduce: str = "sum"):
    """torch_scatter.scatter(src, index, dim=-2, reduce=...) — call sites conv.py:91-93,97,
    173,177.  ``dim_size=None`` means ``index.max()+1``; mean divides by max(count, 1);
    rows nobody points at are zero."""
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() > 0 else 0
    shape = list(src.shape)
    shape[-2] = dim_size
    out = torch.zeros(shape, dtype=src.dtype, device=src.device)
    out.index_add_(src.dim() - 2, index, src)
    if reduce == "mean":
        cnt = torch.zeros(dim_size, dtype=src.dtype, device=src.device)
        cnt.index_add_(0, index, torch.ones_like(index, dtype=src.dtype))
        out = out / cnt.clamp(min=1).unsqueeze(-1)
    elif reduce != "sum":
        raise ValueError(reduce)
    return out


def pool_sum(x: torch.Tensor, batch: torch.Tensor, num_graphs=None):
    """torch_geometric.nn.global_add_pool — call sites equihnn_egnn.py:167, mhnn.py:216,
    equihnn_equiformer.py:91.  Reduces along dim -2 (the Equiformer wrapper carries a
    leading 1-dim)."""
    if num_graphs is None:
        num_graphs = int(batch.max()) + 1
    return segment_reduce(x, batch, num_graphs, "sum")


class AtomEncoder(nn.Module):
    """ogb.graphproppred.mol_encoder.AtomEncoder (ogb 1.3.6): nine xavier-uniform
    embedding tables summed in feature order 0..8 — call sites equihnn_egnn.py:121,157."""

    def __init__(self, emb_dim: int):
        super().__init__()
        self.atom_embedding_list = nn.ModuleList()
        for d in ATOM_FEATURE_DIMS:
            emb = nn.Embedding(d, emb_dim)
            nn.init.xavier_uniform_(emb.weight.data)
            self.atom_embedding_list.append(emb)

    def forward(self, x):
        out = 0
        for f in range(x.shape[1]):
            out = out + self.atom_embedding_list[f](x[:, f])
        return out


# --------------------------------------------------------------------------------------
# MLP (mlp.py:9-99)
# --------------------------------------------------------------------------------------
def _make_norm(kind: str, width: int) -> nn.Module:
    if kind == "ln":
        return nn.LayerNorm(width)
    if kind == "bn":
        return nn.BatchNorm1d(width)
    if kind == "None":
        return nn.Identity()
    raise AssertionError(kind)


class MLP(nn.Module):
    """mlp.py:9-99.  ``lins`` / ``normalizations`` lists; forward (mlp.py:91-99) is
    norm0 -> [Linear -> ReLU -> norm -> dropout]* -> Linear: the norm comes AFTER ReLU."""

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers,
                 dropout=0.5, Normalization="bn", InputNorm=False):
        super().__init__()
        self.lins = nn.ModuleList()
        self.normalizations = nn.ModuleList()
        self.normalizations.append(_make_norm(Normalization, in_channels) if InputNorm
                                   else nn.Identity())
        widths = [in_channels] + [hidden_channels] * (num_layers - 1) + [out_channels]
        for i in range(num_layers):
            self.lins.append(nn.Linear(widths[i], widths[i + 1]))
            if i < num_layers - 1:
                self.normalizations.append(_make_norm(Normalization, widths[i + 1]))
        self.dropout = dropout

    def forward(self, x):
        x = self.normalizations[0](x)
        last = len(self.lins) - 1
        for i in range(last):
            x = F.relu(self.lins[i](x))
            x = self.normalizations[i + 1](x)
            x = F.dropout(x, p=self.dropout, training=self.training)
        return self.lins[last](x)


# --------------------------------------------------------------------------------------
# hypergraph convolutions (conv.py)
# --------------------------------------------------------------------------------------
class MHNNConv(nn.Module):
    """conv.py:8-101: node AND hyperedge features, four MLPs on concatenated pairs."""

    def __init__(self, hid_dim, mlp1_layers=1, mlp2_layers=1, mlp3_layers=1, mlp4_layers=1,
                 aggr="mean", dropout=0.0, normalization="None", input_norm=False):
        super().__init__()
        # conv.py:22-70: mlpK_layers == 0 makes W_k `lambda X: X[..., hid_dim:]` (the second half of its input, no parameters)
        half = lambda X: X[..., hid_dim:]
        mk = lambda n: (MLP(hid_dim * 2, hid_dim, hid_dim, n, dropout=dropout, Normalization=normalization, InputNorm=input_norm)
                        if n > 0 else half)
        self.W1, self.W2, self.W3, self.W4 = mk(mlp1_layers), mk(mlp2_layers), mk(mlp3_layers), mk(mlp4_layers)
        self.aggr = aggr

    def forward(self, X, E, vertex, edges):  # conv.py:87-101
        n_nodes = X.shape[-2]
        m_ve = self.W1(torch.cat((X[..., vertex, :], E[..., edges, :]), -1))
        m_e = segment_reduce(m_ve, edges, None, self.aggr)
        E = self.W2(torch.cat((E, m_e), -1))
        m_ev = self.W3(torch.cat((X[..., vertex, :], E[..., edges, :]), -1))
        m_v = segment_reduce(m_ev, vertex, n_nodes, self.aggr)
        X = self.W4(torch.cat((X, m_v), -1))
        return X, E


class MHNNSConv(nn.Module):
    """conv.py:104-182: node features only, three MLPs, alpha-residual to X0."""

    def __init__(self, hid_dim, mlp1_layers=1, mlp2_layers=1, mlp3_layers=1, aggr="mean",
                 alpha=0.5, dropout=0.0, normalization="None", input_norm=False):
        super().__init__()
        # conv.py:118-156: zero layers -> W1 = Identity, W2 = the second half of its input, and for W3 the reference assigns
        # ``self.W`` instead of ``self.W3`` (:155-156), so forward() raises AttributeError at :180 -- restated as it is
        mk = lambda cin, n: MLP(cin, hid_dim, hid_dim, n, dropout=dropout, Normalization=normalization, InputNorm=input_norm)
        self.W1 = mk(hid_dim, mlp1_layers) if mlp1_layers > 0 else nn.Identity()
        self.W2 = mk(hid_dim * 2, mlp2_layers) if mlp2_layers > 0 else (lambda X: X[..., hid_dim:])
        if mlp3_layers > 0:
            self.W3 = mk(hid_dim, mlp3_layers)
        else:
            self.W = nn.Identity()
        self.aggr = aggr
        self.alpha = alpha

    def forward(self, X, vertex, edges, X0):  # conv.py:169-182
        n_nodes = X.shape[-2]
        x_ve = self.W1(X)[..., vertex, :]
        x_e = segment_reduce(x_ve, edges, None, self.aggr)
        x_ev = self.W2(torch.cat((X[..., vertex, :], x_e[..., edges, :]), -1))
        x_v = segment_reduce(x_ev, vertex, n_nodes, self.aggr)
        return self.W3((1 - self.alpha) * x_v + self.alpha * X0)


# --------------------------------------------------------------------------------------
# EGNN front-end (egnn_layer.py:145-366), restricted to how equihnn_egnn.py:123-129,158
# configures and calls it: mask=None, edges=None, adj_mat=None, fourier_features=0,
# m_pool_method="sum", k=16 nearest incl. self over the whole batch cloud.
# --------------------------------------------------------------------------------------
class CoorsNorm(nn.Module):  # egnn_layer.py:71-81 (parameter kept; branch is dead here)
    def __init__(self, scale_init=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.full((1,), float(scale_init)))


def knn_self_included(coors: torch.Tensor, k: int):
    """egnn_layer.py:253-283: squared distance ((ci-cj)**2).sum(-1) over the whole cloud,
    k smallest including self.  Returns (d2[N,k], idx[N,k])."""
    rel = coors[:, None, :] - coors[None, :, :]
    d2 = (rel ** 2).sum(-1)
    return d2.topk(k, dim=-1, largest=False)


class EGNN(nn.Module):
    def __init__(self, dim, m_dim=16, num_nearest_neighbors=16, init_eps=1e-3,
                 norm_coors_scale_init=1e-2):
        super().__init__()
        e_in = 2 * dim + 1
        # egnn_layer.py:180-186 (index 1 is the dropout slot, Identity at p=0)
        self.edge_mlp = nn.Sequential(nn.Linear(e_in, 2 * e_in), nn.Identity(), nn.SiLU(),
                                      nn.Linear(2 * e_in, m_dim), nn.SiLU())
        self.node_norm = nn.LayerNorm(dim)                      # :192
        self.coors_norm = CoorsNorm(norm_coors_scale_init)      # :193-195
        self.node_mlp = nn.Sequential(nn.Linear(dim + m_dim, 2 * dim), nn.Identity(), nn.SiLU(),
                                      nn.Linear(2 * dim, dim))  # :199-208
        self.coors_mlp = nn.Sequential(nn.Linear(m_dim, 4 * m_dim), nn.Identity(), nn.SiLU(),
                                       nn.Linear(4 * m_dim, 1))  # :210-217, dead branch
        self.k = num_nearest_neighbors
        for mod in self.modules():  # :227-230
            if type(mod) is nn.Linear:
                nn.init.normal_(mod.weight, std=init_eps)

    def forward(self, feats, coors):
        """feats [N,C], coors [N,3] -> node_out [N,C] (coordinate output is discarded by the
        wrapper, equihnn_egnn.py:158, so it is not computed)."""
        d2, idx = knn_self_included(coors, self.k)              # :253-288
        n, c = feats.shape
        f_i = feats[:, None, :].expand(n, self.k, c)
        f_j = feats[idx]                                        # :298
        m_ij = self.edge_mlp(torch.cat((f_i, f_j, d2[..., None]), -1))  # :305-310
        m_i = m_ij.sum(-2)                                      # :357-358
        return self.node_mlp(torch.cat((self.node_norm(feats), m_i), -1)) + feats  # :360-362


# --------------------------------------------------------------------------------------
# model wrappers
# --------------------------------------------------------------------------------------
_ACT = {"Id": nn.Identity, "relu": nn.ReLU, "prelu": nn.PReLU}


class EGNNEquiHNNS(nn.Module):
    """equihnn_egnn.py:98-169 (``egnn_equihnns``): one shared MHNNSConv applied L times."""

    def __init__(self, num_target, args):
        super().__init__()
        self.act = _ACT[args.activation]()
        self.dropout = nn.Dropout(args.dropout)
        self.nlayer = args.All_num_layers
        self.atom_encoder = AtomEncoder(args.MLP_hidden)
        self.egnn_layer = EGNN(args.MLP_hidden)
        self.conv = MHNNSConv(args.MLP_hidden, args.MLP1_num_layers, args.MLP2_num_layers,
                              args.MLP3_num_layers, aggr=args.aggregate, dropout=args.dropout,
                              normalization=args.normalization)
        self.mlp_out = MLP(args.MLP_hidden, args.output_hidden, num_target,
                           args.output_num_layers, dropout=args.dropout,
                           Normalization=args.normalization, InputNorm=False)

    def forward(self, data, taps=None):
        V, E = data.edge_index0, data.edge_index1
        x = self.atom_encoder(data.x)
        if taps is not None:
            taps["atom_encoder"] = x
        x = self.egnn_layer(x, data.pos)
        if taps is not None:
            taps["front_end"] = x
        x0 = x
        for i in range(self.nlayer):
            x = self.conv(self.dropout(x), V, E, x0)
            if taps is not None:
                taps[f"conv{i}"] = x  # pre-activation, as a forward hook on conv sees it
            x = self.act(x)
        x = pool_sum(self.dropout(x), data.batch)
        if taps is not None:
            taps["pool"] = x
        return self.mlp_out(x).view(-1)


class MHNNM(nn.Module):
    """mhnn.py:144-218 (``mhnnm``): L unshared MHNNConv layers, BatchNorm1d on node rows."""

    def __init__(self, num_target, args):
        super().__init__()
        self.act = _ACT[args.activation]()
        self.dropout = nn.Dropout(args.dropout)
        self.nlayer = args.All_num_layers
        self.atom_encoder = AtomEncoder(args.MLP_hidden)
        self.bond_encoder = nn.Embedding(6, args.MLP_hidden)
        self.layers = nn.ModuleList()
        self.batch_norms = nn.ModuleList()
        for _ in range(self.nlayer):
            self.layers.append(MHNNConv(args.MLP_hidden, args.MLP1_num_layers, args.MLP2_num_layers,
                                        args.MLP3_num_layers, args.MLP4_num_layers,
                                        aggr=args.aggregate, dropout=args.dropout,
                                        normalization=args.normalization))
            self.batch_norms.append(nn.BatchNorm1d(args.MLP_hidden))
        self.mlp_out = MLP(args.MLP_hidden, args.output_hidden, num_target,
                           args.output_num_layers, dropout=args.dropout,
                           Normalization=args.normalization, InputNorm=False)

    def forward(self, data, taps=None):
        V, E = data.edge_index0, data.edge_index1
        x = self.atom_encoder(data.x)
        e = self.bond_encoder(data.edge_attr.squeeze(-1))  # mhnn.py:202 (.squeeze())
        if taps is not None:
            taps["atom_encoder"] = x
        for i, layer in enumerate(self.layers):  # mhnn.py:204-214
            x, e = layer(x, e, V, E)
            x = self.batch_norms[i](x)
            if taps is not None:
                taps[f"bn{i}"] = x
            if i != self.nlayer - 1:
                x, e = self.act(x), self.act(e)
            x, e = self.dropout(x), self.dropout(e)
        x = pool_sum(x, data.batch)
        if taps is not None:
            taps["pool"] = x
        return self.mlp_out(x).view(-1)


def hyperedge_batch(data):
    """mhnn.py:54-58: molecule id of every hyperedge (from n_e) — built without the reference's
    per-molecule ``.item()`` loop."""
    b = data.n_e.shape[0]
    return torch.repeat_interleave(torch.arange(b, device=data.x.device), data.n_e)


def pool_high_order(e, data):
    """mhnn.py:72: global_add_pool(e[e_order > 2], he_batch).  The reference sizes the result by
    he_batch.max()+1 and then fails in torch.cat when the last molecules have no hyperedge of
    order > 2; here such molecules simply get a zero row (identical whenever the reference runs)."""
    keep = (data.e_order > 2).to(e.dtype).unsqueeze(-1)
    return segment_reduce(e * keep, hyperedge_batch(data), data.n_e.shape[0], "sum")


class _PairedBase(nn.Module):
    """Common part of mhnn.py:11-81 (``mhnn``) and equihnn_egnn.py:12-95 (``egnn_equihnn``): ONE
    shared MHNNConv applied L times, node AND high-order-hyperedge pooling, 2C-wide head."""

    def __init__(self, num_target, args, with_egnn):
        super().__init__()
        self.act = _ACT[args.activation]()
        self.dropout = nn.Dropout(args.dropout)
        self.nlayer = args.All_num_layers
        if with_egnn:
            self.egnn_layer = EGNN(args.MLP_hidden)
        self.atom_encoder = AtomEncoder(args.MLP_hidden)
        self.bond_encoder = nn.Embedding(6, args.MLP_hidden)
        self.conv = MHNNConv(args.MLP_hidden, args.MLP1_num_layers, args.MLP2_num_layers,
                             args.MLP3_num_layers, args.MLP4_num_layers, aggr=args.aggregate,
                             dropout=args.dropout, normalization=args.normalization)
        self.mlp_out = MLP(args.MLP_hidden * 2, args.output_hidden * 2, num_target,
                           args.output_num_layers, dropout=args.dropout,
                           Normalization=args.normalization, InputNorm=False)
        self.with_egnn = with_egnn

    def forward(self, data, taps=None):
        V, E = data.edge_index0, data.edge_index1
        x = self.atom_encoder(data.x)
        if self.with_egnn:
            x = self.egnn_layer(x, data.pos)
            if taps is not None:
                taps["front_end"] = x
        e = self.bond_encoder(data.edge_attr.squeeze(-1))
        for i in range(self.nlayer):
            x, e = self.conv(x, e, V, E)
            if i != self.nlayer - 1:
                x, e = self.act(x), self.act(e)
            x, e = self.dropout(x), self.dropout(e)
        xp = pool_sum(x, data.batch)
        ep = pool_high_order(e, data)
        if taps is not None:
            taps["pool"] = torch.cat((xp, ep), -1)
        return self.mlp_out(torch.cat((xp, ep), -1)).view(-1)


class MHNN(_PairedBase):
    def __init__(self, num_target, args):
        super().__init__(num_target, args, with_egnn=False)


class EGNNEquiHNN(_PairedBase):
    def __init__(self, num_target, args):
        super().__init__(num_target, args, with_egnn=True)


class MHNNS(nn.Module):
    """mhnn.py:84-141 (``mhnns``): egnn_equihnns without the geometric front-end."""

    def __init__(self, num_target, args):
        super().__init__()
        self.act = _ACT[args.activation]()
        self.dropout = nn.Dropout(args.dropout)
        self.nlayer = args.All_num_layers
        self.atom_encoder = AtomEncoder(args.MLP_hidden)
        self.conv = MHNNSConv(args.MLP_hidden, args.MLP1_num_layers, args.MLP2_num_layers,
                              args.MLP3_num_layers, aggr=args.aggregate, dropout=args.dropout,
                              normalization=args.normalization)
        self.mlp_out = MLP(args.MLP_hidden, args.output_hidden, num_target, args.output_num_layers,
                           dropout=args.dropout, Normalization=args.normalization, InputNorm=False)

    def forward(self, data, taps=None):
        V, E = data.edge_index0, data.edge_index1
        x = self.atom_encoder(data.x)
        x0 = x
        for i in range(self.nlayer):
            x = self.conv(self.dropout(x), V, E, x0)
            if taps is not None:
                taps[f"conv{i}"] = x
            x = self.act(x)
        x = pool_sum(self.dropout(x), data.batch)
        if taps is not None:
            taps["pool"] = x
        return self.mlp_out(x).view(-1)


class EGNNEquiHNNM(MHNNM):
    """equihnn_egnn.py:172-261 (``egnn_equihnnm``): mhnnm with the EGNN front-end."""

    def __init__(self, num_target, args):
        super().__init__(num_target, args)
        self.egnn_layer = EGNN(args.MLP_hidden)

    def forward(self, data, taps=None):
        V, E = data.edge_index0, data.edge_index1
        x = self.egnn_layer(self.atom_encoder(data.x), data.pos)
        if taps is not None:
            taps["front_end"] = x
        e = self.bond_encoder(data.edge_attr.squeeze(-1))
        for i, layer in enumerate(self.layers):
            x, e = layer(x, e, V, E)
            x = self.batch_norms[i](x)
            if taps is not None:
                taps[f"bn{i}"] = x
            if i != self.nlayer - 1:
                x, e = self.act(x), self.act(e)
            x, e = self.dropout(x), self.dropout(e)
        x = pool_sum(x, data.batch)
        if taps is not None:
            taps["pool"] = x
        return self.mlp_out(x).view(-1)


MODELS = {"egnn_equihnns": EGNNEquiHNNS, "mhnnm": MHNNM, "mhnn": MHNN, "mhnns": MHNNS,
          "egnn_equihnn": EGNNEquiHNN, "egnn_equihnnm": EGNNEquiHNNM}
