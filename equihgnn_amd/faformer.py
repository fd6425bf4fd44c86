"""FAFormer front-end of ``faformer_equihnns`` (equihnn_fa_former.py:130-143: d_model = d_edge = C,
2 layers, 2 heads, 16 neighbours, radius 5, SwiGLU MLPs; fa_former_layer.py), with the reference's
parameter names.  Round-1 status: correct first — neighbour search and every gather run in the HIP
kernels, the per-edge MLPs are library GEMMs + torch elementwise ops (the fused per-edge kernels of
the EGNN / Equiformer paths are the template for the next round).

Structure used (instead of transcribing the reference's dense-batch code):
* The wrapper calls FAFormer with a leading batch dim of 1, so ``batch_idx`` is all zeros: every
  "per-molecule" frame (:293-337, :517-571) is the frame of the WHOLE batch cloud.
* A frame is F = V·diag(s) for the eigenvectors V of the covariance and the 8 sign patterns s, so the
  8 frame projections of a point are y ⊙ s with y = (x - c)·V (:86-113): one projection, eight sign flips.
* Frame averaging commutes with the (linear) last Linear of each frame MLP.
* Reference quirk reproduced: the attention's geometric context gathers frame features with
  neighbour ids that lack the per-frame offset (:536-549), so all eight frames read frame 0 and the
  signed frame average cancels; the context equals the cloud centroid (to 1e-8).  See
  oracle/ref_faformer.py, which restates the computation literally and is pinned to the reference.
* The last layer's edge update never reaches the output and is skipped (its parameters get
  grad=None in the reference as well).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .index import HyperIndex


def _sign_ops(device, dtype):
    d = torch.arange(2, device=device, dtype=dtype) * 2 - 1              # (-1, +1), built on the device
    g = torch.meshgrid(d, d, d, indexing="ij")
    return torch.stack(g, -1).reshape(8, 3)                       # fa_former_layer.py:70-84


def _layer_norm(mod: nn.LayerNorm, x, passthrough: bool = False):
    """nn.LayerNorm over the last dim through the row kernels (one launch each way, the backward also yields
    d gamma / d beta): torch's own kernels take 2.3 ms forward and 3.3 ms backward on the [E*8, 128] frame
    tensors of a batch-512 step, 4-5x the time the bytes need."""
    c = x.shape[-1]
    if x.is_cuda and x.dtype == torch.float32 and c % 4 == 0 and c <= 1024 and x.numel() > 0:
        if passthrough:     # (LayerNorm(x), x): x's other consumer reads the second output, see ops._LayerNormRows
            y, xp = ops.layer_norm_rows(x.reshape(-1, c), mod.weight, mod.bias, mod.eps, passthrough=True)
            return y.view(x.shape), xp.view(x.shape)
        return ops.layer_norm_rows(x.reshape(-1, c), mod.weight, mod.bias, mod.eps).view(x.shape)
    return (mod(x), x) if passthrough else mod(x)


import os as _os

BIAS_RIDERS = not _os.environ.get("EQH_NO_BIAS_RIDERS")   # bias gradients of edge-level Linears taken by their consumers' backward passes
FUSE_EDGE_HIDDEN = True     # EdgeModule: gather + adds + SwiGLU + dropout + LayerNorm of the edge MLP's hidden layer in one launch
FUSE_FRAME_HIDDEN = True    # SwiGLUMLP.frame_mean: one launch for the hidden layer over the 8 sign frames (ops.frame_hidden)


class SwiGLUMLP(nn.Module):
    """fa_former_layer.py:241-289: fc1 -> (silu(x1) * x2) -> drop -> LayerNorm -> fc2 -> drop."""

    def __init__(self, d_in, d_hidden, d_out, drop=0.0):
        super().__init__()
        self.fc1 = nn.Linear(d_in, d_hidden)
        self.norm = nn.LayerNorm(d_hidden // 2)
        self.fc2 = nn.Linear(d_hidden // 2, d_out)
        self.p = drop

    def hidden(self, pre):
        if pre.is_cuda and pre.dtype == torch.float32 and pre.shape[-1] % 8 == 0:
            # SiLU gate, product and dropout in one pass each way (no mask tensor, no chunk copies)
            return _layer_norm(self.norm, ops.swiglu_dropout(pre, self.p if self.training else 0.0))
        a, b = pre.chunk(2, dim=-1)
        return _layer_norm(self.norm, F.dropout(F.silu(a) * b, self.p, self.training))

    def _fc2(self, h):
        if h.is_cuda:   # weight / bias gradients accumulate in place (ops._Linear), the bias sum is the row kernel
            return ops.linear(h, self.fc2.weight, self.fc2.bias)
        return self.fc2(h)

    def _fc1(self, x):
        if x.is_cuda:   # as _fc2: the weight gradient joins the batched launch at the end of the backward pass
            return ops.linear(x, self.fc1.weight, self.fc1.bias)
        return self.fc1(x)

    def forward(self, x, res=None):
        """``res``: the tensor the caller adds the result to (a residual connection); folded into the dropout pass."""
        y = self._fc2(self.hidden(self._fc1(x)))
        p = self.p if self.training else 0.0
        if p > 0 and ops.dropout_add_supported(y, res):
            return ops.dropout_add(y, res, p)
        y = F.dropout(y, self.p, self.training)
        return y if res is None else y + res

    def frame_mean(self, y, extra=None):
        """mean over the 8 sign frames of MLP(cat(y ⊙ s, extra)); y [..., 3], extra [..., E]."""
        w = self.fc1.weight
        fused = y.is_cuda and y.dtype == torch.float32 and w.shape[0] == 256 and FUSE_FRAME_HIDDEN
        if fused and (extra is None or extra.shape[-1] == 1):
            # first Linear over the sign frames (the squared distance's K = 1 column included) + SwiGLU + dropout + LayerNorm
            # in one launch each way (csrc/faformer_ew.hip)
            if extra is not None and w.shape[1] == 4:      # the whole [256, 4] weight, read in place (column 3 multiplies extra)
                h = ops.frame_hidden(y, w, self.fc1.bias, self.norm.weight, self.norm.bias, self.norm.eps,
                                     self.p if self.training else 0.0, None, extra, None)
            else:
                h = ops.frame_hidden(y, w[:, :3], self.fc1.bias, self.norm.weight, self.norm.bias, self.norm.eps,
                                     self.p if self.training else 0.0, None, extra, None if extra is None else w[:, 3])
            pre = base = None
        elif extra is None:
            base = self.fc1.bias
        elif extra.shape[-1] == 1:      # one extra input (the squared distance): a broadcast multiply-add, not a K = 1 GEMM
            base = torch.addcmul(self.fc1.bias, extra, w[:, 3])
        else:
            base = F.linear(extra, w[:, 3:], self.fc1.bias)
        if base is None:
            pass
        elif fused:
            h = ops.frame_hidden(y, w[:, :3], base, self.norm.weight, self.norm.bias, self.norm.eps,
                                 self.p if self.training else 0.0)
            pre = None
        elif y.is_cuda and y.dtype == torch.float32 and w.shape[0] == 256:
            pre = ops.frame_pre(y, w[:, :3], base)                         # [..., 8, H] in one pass (csrc/faformer_ew.hip)
        else:
            s = _sign_ops(y.device, y.dtype)                               # [8, 3]
            u = y.unsqueeze(-2) * s                                        # [..., 8, 3]
            pre = F.linear(u, w[:, :3])                                    # [..., 8, H]
            pre = pre + (base if extra is None else base.unsqueeze(-2))
        if pre is not None:
            h = self.hidden(pre)                                           # [..., 8, H/2]
        if self.training and self.p > 0:                                   # dropout after fc2 is per frame
            if ops.linear_dropout_mean_supported(h, self.fc2.weight):
                # fc2, the per-frame dropout and the frame average as one GEMM: the [E * 8, C] product is never written
                return ops.linear_dropout_mean(h, self.fc2.weight, self.fc2.bias, self.p)
            if h.is_cuda and h.dtype == torch.float32 and self.fc2.weight.shape[0] % 4 == 0:
                # dropout + frame average in one pass; fc2's bias gradient (the column sums of the [E * 8, C] gradient)
                # rides that pass's backward instead of reading the tensor once more
                out = ops.linear(h, self.fc2.weight, self.fc2.bias, bias_grad=False)
                return ops.dropout_mean(out, self.p, bias=self.fc2.bias)
            return F.dropout(self._fc2(h), self.p, True).mean(-2)
        return self._fc2(h.mean(-2))


def _frame_axes(x, mask=None):
    """Centre and eigenvectors of the (masked) covariance — create_frame, :86-113.  x [B,P,3].
    Returns y = (x - c·mask)·V [B,P,3], V [B,3,3], centre [B,3]; no gradient through V (:98-99)."""
    # A cloud-wide frame (the FFN's: one point set of N atoms) sums over ~15 k points at the Molecule3D batch size, and
    # its eigenvalues lie within 3 % of each other (molecules of all orientations overlap around the origin): fp32
    # accumulation of the centroid / covariance (1e-6 relative) turns the eigenvectors by 1e-4 rad, 5e-5 of the
    # feature scale at batch 512.  Those two reductions are therefore accumulated in float64 (3 x N numbers).
    wide = x.shape[1] > 64
    acc = torch.float64 if wide else x.dtype
    if mask is None:
        center = x.to(acc).mean(1).to(x.dtype)
        xc = x - center.unsqueeze(1)
        xm = xc
    else:
        m = mask.unsqueeze(-1).to(x.dtype)
        keep = m > 0
        zero = x.new_zeros(())
        # a point set with NO valid member (the far-away atoms of a padded batch: every neighbour is beyond the
        # radius) gets centre 0 instead of the reference's 0/0 = NaN, which would leak into the weight
        # gradients through 0 * NaN
        center = (torch.where(keep, x, zero).to(acc).sum(1) / m.to(acc).sum(1).clamp(min=1.0)).to(x.dtype)  # (where, not *: masked rows may
        xc = x - center.unsqueeze(1) * m                                   #  hold anything; they keep raw x, :94)
        xm = torch.where(keep, xc, zero)
    # the 3 x 3 products as broadcast multiply + sum: batched GEMMs of these shapes ([15 k x 3 x 16] . [15 k x 16 x 3], or one
    # [3 x 15 k] . [15 k x 3] in float64) took 0.1-0.4 ms each in the library, 1.5 ms per step together
    with torch.no_grad():
        xa = xm.to(acc)
        cov = (xa.unsqueeze(-1) * xa.unsqueeze(-2)).sum(1)                 # [B, 3, 3] = X^T X
        vec = ops.eigh3(cov.to(x.dtype))                                   # geo_eigh3 (csrc/eigh3.hip)
    y = (xc.unsqueeze(-1) * vec.unsqueeze(1)).sum(-2)                      # [B, P, 3] = xc V
    return y, vec, center


class EdgeGraph:
    """Self-excluded kNN by true distance over the whole cloud, radius mask (:651-668)."""

    def __init__(self, pos, index: HyperIndex, k: int, radius: float):
        n = pos.shape[0]
        # (clouds of <= k points -- a single small molecule, the reference's tests/run_*_3d.sh use --batch_size 1 -- keep k
        # slots, the surplus masked: index._small_cloud_knn)
        self.N, self.K = n, k
        self.nbr, dist, self.csr_t = index.knn(pos, k, 1)
        self.nbr_flat = self.nbr.reshape(-1)
        self.mask = dist <= radius

    def gather(self, t):
        """t[nbr] for t [N, C] -> [N, K, C] through the row-gather kernel."""
        return ops.gather_rows(t, self.nbr_flat, self.csr_t).view(self.N, self.K, -1)


class EdgeModule(nn.Module):
    """:340-400."""

    def __init__(self, d, d_edge, drop):
        super().__init__()
        self.coord_mlp = SwiGLUMLP(4, d_edge, d_edge, drop)
        self.edge_mlp = SwiGLUMLP(2 * d + d_edge, d, d, drop)
        self.att_mlp = nn.Sequential(nn.Linear(d, 1), nn.Sigmoid())
        self.d = d

    def forward(self, tok, geo, g: EdgeGraph, res=None):
        """``res``: the edge features this module's output is added to (:602-604), folded into the gate kernel."""
        gj = g.gather(geo_pad(geo))                                        # x_j (padded rows)  [N,K,4]
        if ops.edge_frame_supported(geo, gj, g.mask):
            # offsets, squared distances, masked centre / covariance, eigenvectors and the projection: one launch each way
            y, d2 = ops.edge_frame(geo, gj, g.mask)
        else:
            rel = geo.unsqueeze(1) - gj[..., :3]                           # x_i - x_j  [N,K,3]
            d2 = (rel ** 2).sum(-1, keepdim=True)
            y, _, _ = _frame_axes(rel, g.mask)
        feats = self.coord_mlp.frame_mean(y, d2)                           # [N,K,C]
        # edge_mlp's first Linear split by input block: token_i / token_j parts at node level
        w, d = self.edge_mlp.fc1.weight, self.d
        lin = ops.linear if tok.is_cuda else (lambda x, w_, b_=None, cols=None: F.linear(x, w_[:, cols[0]:cols[1]], b_))
        a_i = lin(tok, w, self.edge_mlp.fc1.bias, cols=(0, d))             # [N, H]  receiver part (+ bias)
        b_j = lin(tok, w, None, cols=(d, 2 * d))                           # [N, H]  sender part, gathered per edge
        c_ij = lin(feats, w, None, cols=(2 * d, w.shape[1]))               # [N, K, H]
        mlp = self.edge_mlp
        if tok.is_cuda and tok.dtype == torch.float32 and w.shape[0] == 256 and FUSE_EDGE_HIDDEN:
            # gather + the two adds + SwiGLU + dropout + LayerNorm in one launch each way (csrc/faformer_ew.hip)
            hid = ops.edge_hidden(a_i, b_j, c_ij, g.nbr, g.csr_t, mlp.norm.weight, mlp.norm.bias, mlp.norm.eps,
                                  mlp.p if self.training else 0.0)
        else:
            hid = mlp.hidden(a_i.unsqueeze(1) + g.gather(b_j) + c_ij)
        a = self.att_mlp[0]
        ride = BIAS_RIDERS and ops.geom_supported(tok) and mlp.fc2.weight.shape[0] % 4 == 0 and mlp.fc2.weight.shape[0] <= 1024
        # (fc2's bias gradient = the column sums of the gate's input gradient: taken by the gate's backward pass, ops.gate_rows)
        pair = ops.linear(hid, mlp.fc2.weight, mlp.fc2.bias, bias_grad=False) if ride else mlp._fc2(hid)
        # att_mlp = Linear(d, 1) + Sigmoid on ~250 k edge rows: a row-wise dot product -- as a GEMM with ONE output column the
        # library needs 2 ms for it.  The dropout in front of it, the dot product, the gate and the residual add behind it
        # are one pass each way (ops.gate_rows, csrc/faformer_ew.hip)
        p = self.edge_mlp.p if self.training else 0.0
        if pair.is_cuda and pair.dtype == torch.float32 and pair.shape[-1] % 4 == 0 and pair.shape[-1] <= 1024:
            return ops.gate_rows(pair, a.weight, a.bias, res, p, lin_bias=mlp.fc2.bias if ride else None)
        pair = F.dropout(pair, p, self.training)
        out = pair * torch.sigmoid((pair * a.weight.view(-1)).sum(-1, keepdim=True) + a.bias)
        return out if res is None else res + out


_HEAD_SELECT = {}


def _head_select(h, device):
    """[4, 3h] 0/1 matrix: row r < 2h picks block r of the (q heads | k heads | v heads) columns of the qkv product."""
    key = (h, str(device))
    if key not in _HEAD_SELECT:
        sel = torch.zeros(4, 3 * h)
        for r in range(2 * h):
            sel[r, r] = 1.0
        _HEAD_SELECT[key] = sel.to(device)
    return _HEAD_SELECT[key]


def geo_pad(geo):
    """[N,3] -> [N,4] (the row-gather kernel moves 16-byte rows)."""
    return F.pad(geo, (0, 1))


class MLPAttnEdgeAggregation(nn.Module):
    """:403-573."""

    def __init__(self, d, d_edge, n_heads, proj_drop, attn_drop):
        super().__init__()
        self.h, self.dh, self.deh = n_heads, d // n_heads, d_edge // n_heads
        self.layernorm_qkv = nn.Sequential(nn.LayerNorm(d), nn.Linear(d, 3 * d))
        self.layernorm_qkv_edge = nn.Sequential(nn.LayerNorm(d_edge), nn.Linear(d_edge, 2 * d_edge))
        self.mlp_attn = nn.Linear(self.dh, 1, bias=False)
        self.edge_attn = nn.Linear(self.deh, 1, bias=False)
        self.W_output = SwiGLUMLP(d + d_edge, d, d, proj_drop)
        self.W_gate = nn.Linear(d, 1)
        self.W_frame_agg = nn.Sequential(nn.Linear(n_heads, 1), nn.SiLU())  # parameters only, see below
        self.attn_drop = attn_drop
        nn.init.constant_(self.W_gate.weight, 0.0)
        nn.init.constant_(self.W_gate.bias, 1.0)

    def forward(self, tok, geo, edge, g: EdgeGraph, row_mask=None, edge_passthrough: bool = False):
        """-> (tokens, coordinates, edge features for the residual of the edge update that follows: ``edge`` itself, or
        with ``edge_passthrough`` its alias out of the LayerNorm node, whose backward then adds the residual's gradient
        inside the LayerNorm kernel -- an autograd add over [E, C] otherwise, 130 us at the Molecule3D batch)."""
        n, k, h = g.N, g.K, self.h
        d, de = self.h * self.dh, self.h * self.deh
        qkv_in, qkv_lin = _layer_norm(self.layernorm_qkv[0], tok), self.layernorm_qkv[1]
        qkv = ops.linear(qkv_in, qkv_lin.weight, qkv_lin.bias) if tok.is_cuda else qkv_lin(qkv_in)
        fused = ops.geom_supported(tok) and ops.rowdot_supported(qkv, 4) and h <= 2 and k <= 16
        # The logits are LINEAR in their inputs (:483-489: Linear(dh, 1) of q_i + k_j, plus Linear(deh, 1) of the edge
        # query), so they are evaluated where the inputs live instead of on [N, K, h, dh] edge tensors:
        #   w . (q_i + k_j)      = a_q[i] + a_k[j]                  two dot products per ATOM and head; a_k is gathered
        #                                                           as 16-byte rows beside v
        #   w_e . (Wq_h x_e + b) = (Wq_h^T w_e) . x_e + w_e . b_h   one dot product per edge and head with a vector
        #                                                           folded at weight level: the query half of
        #                                                           layernorm_qkv_edge's Linear (32 GFLOP at the
        #                                                           Molecule3D batch) is never formed
        # (as GEMMs with one or two output columns these took 2 ms each in round 1).
        w = self.mlp_attn.weight.view(-1)
        if fused:
            # a_q and a_k of all heads as ONE row-dot pass over the [N, 3d] product (weight rows: w in the head's
            # block, zeros elsewhere); its backward writes the q / k columns of the product's gradient and adds v's
            U = (_head_select(h, tok.device)[:, :, None] * w).reshape(4, 3 * d)
            qa, qkv = ops.rowdot(qkv, U, None, passthrough=True)                              # [N, 4] = (a_q | a_k)
            qa_n = g.gather(qa)                                                               # [N, K, 4]
            v = qkv[:, 2 * d:]
        else:
            q, kk, v = qkv.chunk(3, -1)
            a_q = (q.reshape(n, h, self.dh) * w).sum(-1)                                      # [N, h]
            a_k = (kk.reshape(n, h, self.dh) * w).sum(-1)
            ak_n = g.gather(F.pad(a_k, (0, 4 - h)))[..., :h]                                  # [N, K, h] (16-byte rows)
        gsum = tok.is_cuda and ops.attn_gather_sum_supported(h, v, g.nbr, g.csr_t)
        v_n = None if gsum else g.gather(v)                                                   # [N, K, d]
        lin_e = self.layernorm_qkv_edge[1]
        folded = ops.geom_supported(tok) and lin_e.weight.shape == (2 * de, de)
        if folded:
            u, c = ops.edge_logit_weights(lin_e.weight, lin_e.bias, self.edge_attn.weight, h)  # [h, de], [h]: one launch each way
        else:
            w_e = self.edge_attn.weight.view(-1)
            u = (lin_e.weight[:de].reshape(h, self.deh, de) * w_e[None, :, None]).sum(1)      # [h, de]
            c = (lin_e.bias[:de].reshape(h, self.deh) * w_e).sum(-1)                          # [h]
        ln_e = self.layernorm_qkv_edge[0]
        if folded and ops.ln_rowdot_supported(edge, h):
            # LayerNorm, the per-head edge logits and (backward) the residual's gradient in one pass over the edge rows
            xe, le, edge_alias = ops.ln_rowdot(edge, ln_e.weight, ln_e.bias, u, c, ln_e.eps)  # [N,K,de], [N,K,h]
            edge = edge_alias if edge_passthrough else edge
        else:
            if edge_passthrough:
                xe, edge = _layer_norm(ln_e, edge, passthrough=True)
            else:
                xe = _layer_norm(ln_e, edge)                                                  # [N, K, de]
            if ops.rowdot_supported(xe, h):   # one pass over xe each way; xe's second gradient (from ve) rides along
                le, xe = ops.rowdot(xe, u, c, passthrough=True)                               # [N, K, h]
            else:
                le = torch.stack([(xe * u[i]).sum(-1) for i in range(h)], -1) + c
        ve = ops.linear(xe, lin_e.weight, lin_e.bias, rows=(de, 2 * de))                      # value half only
        if ops.rowdot_supported(tok, 1):   # Linear(d, 1) on the atom rows: a row-wise dot product, not a one-column GEMM
            gate_logit = ops.rowdot(tok, self.W_gate.weight, self.W_gate.bias)
        else:
            gate_logit = self.W_gate(tok)
        if fused and ops.attn_logits_supported(qa, le, g.mask):
            # the three logit terms, radius mask, softmax over the slots and the dropout: one launch each way
            attn = ops.attn_logits(qa, qa_n, le, g.mask, self.attn_drop if self.training else 0.0)   # [N,h,K]
        else:
            if fused:
                a_q, ak_n = qa[:, :h], qa_n[..., h:2 * h]
            logits = a_q.unsqueeze(1) + ak_n + le
            logits = logits.masked_fill(~g.mask.unsqueeze(-1), -1e9)
            attn = F.dropout(logits.transpose(1, 2).softmax(-1), self.attn_drop, self.training)  # [N,h,K]
        if ops.attn_sum_supported(attn, ve):   # one pass over the values each (csrc/faformer_ew.hip::k_attn_sum)
            # (the neighbours' values are read through the neighbour list: the gathered [N, K, d] tensor is never formed)
            ctx = ops.attn_gather_sum(attn, v, g.nbr, g.csr_t) if gsum else ops.attn_sum(attn, v_n)
            ectx = ops.attn_sum(attn, ve.reshape(n, k, -1))
        else:
            v_n = g.gather(v) if v_n is None else v_n
            ctx = torch.einsum("nhm,nmhd->nhd", attn, v_n.reshape(n, k, h, self.dh)).reshape(n, -1)
            ectx = torch.einsum("nhm,nmhd->nhd", attn, ve.reshape(n, k, h, self.deh)).reshape(n, -1)
        out = self.W_output(torch.cat((ctx, ectx), -1), res=tok)
        # geometric context: with the reference's frame-0 gather (module docstring) the signed frame
        # average cancels and what is left is the centroid of the cloud, for every atom
        if ops.geom_supported(geo):
            # centroid (float64 partial sums), gate and mix in two launches each way; W_frame_agg gets its zero gradient
            agg = self.W_frame_agg[0]
            return out, ops.centre_mix(geo, gate_logit, row_mask, (agg.weight, agg.bias)), edge
        gate = torch.sigmoid(gate_logit)
        if row_mask is None:
            centre = geo.double().mean(0, keepdim=True).to(geo.dtype)       # (float64 accumulation over the cloud, see _frame_axes)
        else:   # padded batch (hipGraph replay): the centroid of the real atoms
            centre = (torch.where(row_mask > 0, geo, geo.new_zeros(())).double().sum(0, keepdim=True)
                      / row_mask.double().sum()).to(geo.dtype)
        # W_frame_agg multiplies the cancelled term: its gradient is rounding noise in the reference;
        # keep it in the autograd graph with an exactly-zero contribution so it gets a (zero) gradient
        # like there, instead of None
        centre = centre + 0.0 * (self.W_frame_agg[0].weight.sum() + self.W_frame_agg[0].bias.sum())
        return out, centre * gate + geo * (1 - gate), edge


class FAFFN(nn.Module):
    """:293-337."""

    def __init__(self, d, drop):
        super().__init__()
        self.W_frame = SwiGLUMLP(3, d, d, drop)
        self.ffn = SwiGLUMLP(2 * d, 4 * d, d, drop)
        self.ln = nn.LayerNorm(d)

    def forward(self, tok, geo, row_mask=None):
        if ops.geom_supported(geo):      # centroid, covariance (float64 partial sums), eigenvectors, projection: two launches
            y = ops.cloud_frame(geo, row_mask)
        else:
            y = _frame_axes(geo.unsqueeze(0), None if row_mask is None else row_mask.view(1, -1))[0][0]
        gfeat = self.W_frame.frame_mean(y)                                  # [N,C]
        return self.ffn(torch.cat((_layer_norm(self.ln, tok), gfeat), -1), res=tok)     # tok + ffn(...), :606


class FAFormerEncoderLayer(nn.Module):
    def __init__(self, d, d_edge, n_heads, proj_drop, attn_drop):
        super().__init__()
        self.self_attn = MLPAttnEdgeAggregation(d, d_edge, n_heads, proj_drop, attn_drop)
        self.ffn = FAFFN(d, proj_drop)
        self.edge_module = EdgeModule(d, d_edge, proj_drop)

    def forward(self, tok, geo, edge, g, last, row_mask=None):
        tok, geo, edge = self.self_attn(tok, geo, edge, g, row_mask, edge_passthrough=not last)
        if not last:
            edge = self.edge_module(tok, geo, g, res=edge)                  # edge + edge_module(...), :602-604
        return self.ffn(tok, geo, row_mask), geo, edge                      # tok + ffn(...) (:606): the residual is added in FAFFN


class FAFormer(nn.Module):
    def __init__(self, d, n_layers=2, n_heads=2, n_neighbors=16, valid_radius=5.0, proj_drop=0.1,
                 attn_drop=0.1):
        super().__init__()
        self.input_transform = nn.Linear(d, d)
        self.edge_module = EdgeModule(d, d, proj_drop)
        self.layers = nn.ModuleList([FAFormerEncoderLayer(d, d, n_heads, proj_drop, attn_drop)
                                     for _ in range(n_layers)])
        self.k, self.radius, self.p = n_neighbors, float(valid_radius), proj_drop

    def forward(self, feats, coords, index: HyperIndex, row_mask=None):
        """``row_mask`` [N,1] float: 1 for the atoms of real molecules of a padded batch (the two cloud-wide
        statistics -- the centroid in the attention and the frame of the FFN -- then skip the padding)."""
        # (one generator launch for the seeds of all dropout sites of the pass, ops.dropout_seeds)
        with ops.dropout_seeds(feats.device, 64, enabled=self.training and feats.is_cuda):
            tok = self.input_transform(feats)
            if self.training and self.p > 0 and ops.dropout_add_supported(tok, None):
                tok = ops.dropout_add(tok, None, self.p)
            else:
                tok = F.dropout(tok, self.p, self.training)
            g = EdgeGraph(coords, index, self.k, self.radius)
            edge = self.edge_module(tok, coords, g)
            geo = coords
            for i, layer in enumerate(self.layers):
                tok, geo, edge = layer(tok, geo, edge, g, last=(i == len(self.layers) - 1), row_mask=row_mask)
        return tok
