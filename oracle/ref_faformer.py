"""ORACLE — test infrastructure, not product code.

CPU restatement of the FAFormer front-end as ``faformer_equihnns`` configures and calls it
(equihnn_fa_former.py:130-143,170-174: d_model = d_edge_model = C, n_layers=2, n_heads=2,
n_neighbors=16, valid_radius=5, activation "swiglu", n_pos=None; called with a leading batch dim of
1, so ``batch_idx`` is all zeros and every "per-molecule" frame is a frame of the WHOLE batch cloud)
and of the wrapper (equihnn_fa_former.py:105-184).  Same parameter names as the reference.
Pinned by tests/golden/faformer_*.npz (captured in eval() mode: the reference keeps
proj_drop = attn_drop = 0.1 active in training, fa_former_layer.py:20-21).
File:line citations are relative to /root/reference/equihgnn/models/layers/fa_former_layer.py.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .ref_models import MLP as HeadMLP
from .ref_models import AtomEncoder, MHNNSConv, pool_sum


def sign_ops(dim=3):
    """:70-84 — the 2^dim sign patterns, first axis varying slowest."""
    d = torch.tensor([-1.0, 1.0])
    grids = torch.meshgrid(*([d] * dim), indexing="ij")
    return torch.stack(grids, -1).reshape(-1, dim)


def create_frame(x, mask):
    """:86-113.  x [B,P,3], mask [B,P] -> projections [B,8,P,3], F_ops [B,8,3,3], center [B,3]."""
    m = mask.unsqueeze(-1)
    center = (x * m).sum(1) / m.sum(1)
    x = x - center.unsqueeze(1) * m
    xm = x.masked_fill(~m, 0.0)
    cov = torch.bmm(xm.transpose(1, 2), xm).detach()
    _, vec = torch.linalg.eigh(cov, UPLO="U")
    f_ops = sign_ops().to(x)[None, :, None, :] * vec[:, None, :, :]      # [B,8,3,3]
    h = torch.einsum("boij,bpj->bopi", f_ops.transpose(2, 3), x)         # NB: unmasked x, as :108
    return h, f_ops.detach(), center


def invert_frame(x, mask, f_ops, center):
    """:115-121.  x [B,8,P,3] -> [B,P,3]."""
    x = torch.einsum("boij,bopj->bopi", f_ops, x).mean(1) + center.unsqueeze(1)
    return x * mask.unsqueeze(-1)


class SwiGLUMLP(nn.Module):
    """:241-289 (fc1 -> chunk -> silu(x1)*x2 -> drop -> LayerNorm(h/2) -> fc2 -> drop)."""

    def __init__(self, d_in, d_hidden, d_out, drop=0.0):
        super().__init__()
        self.fc1 = nn.Linear(d_in, d_hidden)
        self.act = nn.SiLU()
        self.drop1 = nn.Dropout(drop)
        self.norm = nn.LayerNorm(d_hidden // 2)
        self.fc2 = nn.Linear(d_hidden // 2, d_out)
        self.drop2 = nn.Dropout(drop)

    def forward(self, x):
        a, b = self.fc1(x).chunk(2, dim=-1)
        return self.drop2(self.fc2(self.norm(self.drop1(self.act(a) * b))))


class EdgeModule(nn.Module):
    """:340-400."""

    def __init__(self, d, d_edge, drop):
        super().__init__()
        self.coord_mlp = SwiGLUMLP(4, d_edge, d_edge, drop)
        self.edge_mlp = SwiGLUMLP(2 * d + d_edge, d, d, drop)
        self.att_mlp = nn.Sequential(nn.Linear(d, 1), nn.Sigmoid())

    def forward(self, tok, geo, nbr, mask):
        n, k = nbr.shape
        rel = geo.unsqueeze(1) - geo[nbr]                                  # :373-375
        d2 = (rel ** 2).sum(-1, keepdim=True)
        frames, _, _ = create_frame(rel, mask)                             # [N,8,K,3]
        feats = self.coord_mlp(torch.cat((frames, d2.unsqueeze(1).expand(n, 8, k, 1)), -1)).mean(1)
        pair = torch.cat((tok.unsqueeze(1).expand(n, k, -1), tok[nbr], feats), -1)
        pair = self.edge_mlp(pair)
        return pair * self.att_mlp(pair)


class MLPAttnEdgeAggregation(nn.Module):
    """:403-573 with n_heads > 1 (frame-averaged geometric context)."""

    def __init__(self, d, d_edge, n_heads, proj_drop, attn_drop):
        super().__init__()
        self.h, self.dh, self.deh = n_heads, d // n_heads, d_edge // n_heads
        self.layernorm_qkv = nn.Sequential(nn.LayerNorm(d), nn.Linear(d, 3 * d))
        self.layernorm_qkv_edge = nn.Sequential(nn.LayerNorm(d_edge), nn.Linear(d_edge, 2 * d_edge))
        self.mlp_attn = nn.Linear(self.dh, 1, bias=False)
        self.edge_attn = nn.Linear(self.deh, 1, bias=False)
        self.W_output = SwiGLUMLP(d + d_edge, d, d, proj_drop)
        self.W_gate = nn.Linear(d, 1)
        self.attn_dropout = nn.Dropout(attn_drop)
        self.W_frame_agg = nn.Sequential(nn.Linear(n_heads, 1), nn.SiLU())
        nn.init.constant_(self.W_gate.weight, 0.0)                         # :449-451
        nn.init.constant_(self.W_gate.bias, 1.0)

    def forward(self, tok, geo, edge, nbr, mask):
        n, k = nbr.shape
        h = self.h
        q, kk, v = self.layernorm_qkv(tok).chunk(3, -1)
        q, kk, v = (t.view(n, h, self.dh) for t in (q, kk, v))
        qe, ve = self.layernorm_qkv_edge(edge).chunk(2, -1)
        qe, ve = qe.view(n, k, h, self.deh), ve.view(n, k, h, self.deh)
        gate = self.W_gate(tok).sigmoid()                                  # :480-482
        logits = self.mlp_attn(q.unsqueeze(1) + kk[nbr]).squeeze(-1) + self.edge_attn(qe).squeeze(-1)
        logits = logits.masked_fill(~mask.unsqueeze(-1), -1e9)             # :492
        attn = self.attn_dropout(logits.transpose(1, 2).softmax(-1))       # [N,h,K]
        ctx = torch.einsum("nhm,nmhd->nhd", attn, v[nbr]).reshape(n, -1)
        ectx = torch.einsum("nhm,nmhd->nhd", attn, ve).reshape(n, -1)
        out = self.W_output(torch.cat((ctx, ectx), -1)) + tok              # :508-510
        # frame-averaged geometric context over the whole cloud (:517-571)
        full = torch.ones(1, n, dtype=torch.bool)
        frames, f_ops, center = create_frame(geo.unsqueeze(0), full)       # [1,8,N,3]
        # Reference quirk, reproduced: the frame features are flattened to [8*N, .] but gathered with
        # neighbour ids in [0, N) WITHOUT a per-frame offset (:536-549), so all eight "frames" read
        # frame 0's features.  (Consequence: the eight per-frame contexts are equal, the signed frame
        # average cancels, and the geometric context is the cloud centroid up to rounding.)
        g0 = torch.einsum("nhm,nmd->nhd", attn, frames[0][0][nbr])         # [N,h,3] from frame 0
        gctx = g0.unsqueeze(0).expand(8, -1, -1, -1)                       # [8,N,h,3]
        gctx = self.W_frame_agg(gctx.transpose(3, 2)).squeeze(-1)          # [8,N,3]
        gctx = invert_frame(gctx.unsqueeze(0), full, f_ops, center)[0]
        return out, gctx * gate + geo * (1 - gate)                         # :572


class FAFFN(nn.Module):
    """:293-337."""

    def __init__(self, d, drop):
        super().__init__()
        self.W_frame = SwiGLUMLP(3, d, d, drop)
        self.ffn = SwiGLUMLP(2 * d, 4 * d, d, drop)
        self.ln = nn.LayerNorm(d)

    def forward(self, tok, geo):
        n = tok.shape[0]
        frames, _, _ = create_frame(geo.unsqueeze(0), torch.ones(1, n, dtype=torch.bool))
        g = self.W_frame(frames[0]).mean(0)                                # [N,C]
        return self.ffn(torch.cat((self.ln(tok), g), -1))


class FAFormerEncoderLayer(nn.Module):
    def __init__(self, d, d_edge, n_heads, proj_drop, attn_drop):
        super().__init__()
        self.self_attn = MLPAttnEdgeAggregation(d, d_edge, n_heads, proj_drop, attn_drop)
        self.ffn = FAFFN(d, proj_drop)
        self.edge_module = EdgeModule(d, d_edge, proj_drop)

    def forward(self, tok, geo, edge, nbr, mask, last):
        tok, geo = self.self_attn(tok, geo, edge, nbr, mask)
        if not last:  # the last layer's edge update never reaches the output
            edge = edge + self.edge_module(tok, geo, nbr, mask)            # :602-604
        tok = tok + self.ffn(tok, geo)                                     # :606
        return tok, geo, edge


def build_graph(coords, k, radius):
    """:651-668 with a single-molecule batch_idx: self-excluded (filled with 1e9), true distance."""
    n = coords.shape[0]
    dist = (coords[:, None] - coords[None]).norm(dim=-1).detach()
    dist.masked_fill_(torch.eye(n, dtype=torch.bool), 1e9)
    val, idx = dist.topk(k, dim=-1, largest=False)
    return idx, val <= radius


class FAFormer(nn.Module):
    def __init__(self, d, n_layers=2, n_heads=2, n_neighbors=16, valid_radius=5.0,
                 proj_drop=0.1, attn_drop=0.1):
        super().__init__()
        self.input_transform = nn.Linear(d, d)
        self.edge_module = EdgeModule(d, d, proj_drop)
        self.layers = nn.ModuleList([FAFormerEncoderLayer(d, d, n_heads, proj_drop, attn_drop)
                                     for _ in range(n_layers)])
        self.dropout_module = nn.Dropout(proj_drop)
        self.k, self.radius = n_neighbors, valid_radius

    def forward(self, feats, coords):
        keep = feats.sum(-1) != 0                                          # :673 pad_mask
        assert bool(keep.all()), "rows with an exactly zero feature sum are dropped by the reference"
        tok = self.dropout_module(self.input_transform(feats))
        nbr, mask = build_graph(coords, int(min(self.k, coords.shape[0])), self.radius)
        edge = self.edge_module(tok, coords, nbr, mask)
        geo = coords
        for i, layer in enumerate(self.layers):
            tok, geo, edge = layer(tok, geo, edge, nbr, mask, last=(i == len(self.layers) - 1))
        return tok


_ACT = {"Id": nn.Identity, "relu": nn.ReLU, "prelu": nn.PReLU}


class FAFormerEquiHNNS(nn.Module):
    """equihnn_fa_former.py:105-184."""

    def __init__(self, num_target, args):
        super().__init__()
        self.act = _ACT[args.activation]()
        self.dropout = nn.Dropout(args.dropout)
        self.nlayer = args.All_num_layers
        self.atom_encoder = AtomEncoder(args.MLP_hidden)
        self.fa_former = FAFormer(args.MLP_hidden)
        self.conv = MHNNSConv(args.MLP_hidden, args.MLP1_num_layers, args.MLP2_num_layers,
                              args.MLP3_num_layers, aggr=args.aggregate, dropout=args.dropout,
                              normalization=args.normalization)
        self.mlp_out = HeadMLP(args.MLP_hidden, args.output_hidden, num_target, args.output_num_layers,
                               dropout=args.dropout, Normalization=args.normalization, InputNorm=False)

    def forward(self, data, taps=None):
        V, E = data.edge_index0, data.edge_index1
        x = self.atom_encoder(data.x)
        if taps is not None:
            taps["atom_encoder"] = x
        x = self.fa_former(x, data.pos)
        if taps is not None:
            taps["front_end"] = x
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
