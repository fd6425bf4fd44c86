"""ORACLE — test infrastructure, not product code.

CPU restatement of the Equiformer layer with BOTH output degrees and any depth: the attention
block's degree-1 outputs, the (0->1) and (1->1) pairs of its tensor product with the (1,1) basis contraction, the
degree-1 feed-forward and the final degree-1 norm -- everything ``oracle/ref_equiformer.py`` leaves out because
``equiformer_equihnns`` (depth 1, type-0 readout) never reads it.  With depth > 1 those paths become live: block 2's
(1->0) pair consumes block 1's degree-1 output.  Same parameter / buffer names as the reference's ``Equiformer``
(equiformer_layer.py:961-1398), so its state_dict loads with strict=True; pinned by
tests/golden/equiformer_layer_depth2_c*.npz, captured from the reference's own class at depth 2.

Conventions: type-0 features [N, d]; type-1 features [N, d, 3]; per-edge tensors [N, K, ...]; ``D`` [N, K, 3, 3] is
D[1] of the rotation taking r_ij onto the y axis (equiformer/basis.py:194-215), ``rotate in`` is
x'[l, m2] = sum_m1 D[m1, m2] x[l, m1] (:364-366) and ``rotate out`` y'[l, m2] = sum_m1 y[l, m1] D[m2, m1] (:416-418).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .ref_equiformer import (FiberLinear, FiberNorm, Radial, masked_mean, neighbours_self_excluded, wigner_d1_to_y)


def split2(n: int):
    """split_num_into_groups(n, 2), equiformer_layer.py:84-94."""
    return (n + 1) // 2, n // 2


class DTPFull(nn.Module):
    """equiformer_layer.py:260-448 for fibers of at most two degrees."""

    def __init__(self, fiber_in, fiber_out, pool: bool, mid: int = 64):
        super().__init__()
        self.fiber_in, self.fiber_out, self.pool = tuple(fiber_in), tuple(fiber_out), pool
        self.to_xi = FiberLinear(fiber_in, fiber_in)
        self.to_xj = FiberLinear(fiber_in, fiber_in)
        self.kernel_unary = nn.ModuleDict()
        for d_out, dim_out in enumerate(fiber_out):                                   # :289-308
            parts = split2(dim_out) if len(fiber_in) == 2 else (dim_out,)
            for d_in, (dim_in, lo) in enumerate(zip(fiber_in, parts)):
                self.kernel_unary[f"({d_in},{d_out})"] = Radial(dim_in, lo, mid)
        self.self_interact = FiberLinear(fiber_in, fiber_out)
        self.to_out = FiberLinear(fiber_out, fiber_out)

    def forward(self, inp, basis11, idx, dist, mask, D):
        """inp = {0: [N,d0], 1: [N,d1,3] (optional)} -> {0: ..., 1: ...}: pooled [N, e] / [N, e, 3], or unpooled with the
        self-interaction as slot 0: [N, 1+K, e] / [N, 1+K, e, 3]."""
        d = dist[..., None]
        x = {0: self.to_xj.mix(0, inp[0])[idx] + self.to_xi.mix(0, inp[0])[:, None]}                  # :333-360
        if 1 in inp and len(self.fiber_in) > 1:
            x1 = self.to_xj.mix(1, inp[1])[idx] + self.to_xi.mix(1, inp[1])[:, None]
            x[1] = torch.einsum("nkab,nkla->nklb", D, x1)                                              # rotate in, :364-366
        outs = {}
        for d_out in range(len(self.fiber_out)):
            chunks = []
            for d_in in range(len(self.fiber_in)):
                if d_in not in x:
                    continue
                R = self.kernel_unary[f"({d_in},{d_out})"](d)                                          # [N,K,lo,li]
                if d_in == 0 and d_out == 0:
                    chunks.append(torch.einsum("nkol,nkl->nko", R, x[0]))
                elif d_in == 1 and d_out == 0:            # centre component m = 1 of the rotated input, :371-374
                    chunks.append(torch.einsum("nkol,nkl->nko", R, x[1][..., 1]))
                elif d_in == 0 and d_out == 1:            # result sits at m = 1, zero-padded to 3, :407-409
                    o = torch.einsum("nkol,nkl->nko", R, x[0])
                    chunks.append(F.pad(o[..., None], (1, 1)))
                else:                                     # (1,1): basis contraction, :385-404
                    xr = x[1]                             # [N,K,li,3]
                    xf = torch.stack((xr, xr.flip(-1), xr), -1)                                        # [..., m, f]
                    chunks.append(torch.einsum("nkoi,mf,nkimf->nkom", R, basis11, xf))
            out = torch.cat(chunks, dim=2)
            if d_out == 1:
                out = torch.einsum("nklm,nkam->nkla", out, D)                                          # rotate out, :416-418
            outs[d_out] = masked_mean(out, mask) if self.pool else out
        outs = {k: self.to_out.mix(k, v) for k, v in outs.items()}                                     # :430
        si = {k: self.self_interact.mix(k, inp[k]) for k in outs if k in inp and k < len(self.fiber_in)}
        if self.pool:
            return {k: (v + si[k] if k in si else v) for k, v in outs.items()}                         # residual_fn :436
        return {k: torch.cat((si[k][:, None], v), 1) for k, v in outs.items()}                         # :438-447


class MLPAttentionFull(nn.Module):
    """equiformer_layer.py:743-955, heads (1, 1), dim_head (48, 48), attend_self."""

    def __init__(self, c: int, dim_head: int = 48, mid: int = 64):
        super().__init__()
        self.dh, self.scale = dim_head, dim_head ** -0.5
        self.prenorm = FiberNorm((c, c))
        self.to_attn_and_v = DTPFull((c, c), (4 + 4 + 2 * dim_head, dim_head), pool=False, mid=mid)
        self.to_attn_logits = nn.ModuleList([nn.Sequential(nn.LeakyReLU(0.1), nn.Linear(4, 1, bias=False))
                                             for _ in range(2)])
        self.to_values = nn.Sequential(nn.Identity(), FiberLinear((dim_head, dim_head), (dim_head, dim_head)))
        self.attn_head_gates = nn.Sequential(nn.Identity(), nn.Linear(c, 2), nn.Sigmoid(), nn.Identity())
        self.to_out = FiberLinear((dim_head, dim_head), (c, c))

    def forward(self, x0, x1, basis11, idx, dist, mask, D):
        f0, f1 = self.prenorm.norm0(x0), self.prenorm.norm1(x1)
        inter = self.to_attn_and_v({0: f0, 1: f1}, basis11, idx, dist, mask, D)        # {0: [N,17,104], 1: [N,17,48,3]}
        a0, a1, val0 = inter[0][..., :4], inter[0][..., 4:8], inter[0][..., 8:]        # :888-890
        full = F.pad(mask, (1, 0), value=True)[..., None]
        neg = -torch.finfo(x0.dtype).max
        attn = [(self.to_attn_logits[i](a) * self.scale).masked_fill(~full, neg).softmax(dim=1)
                for i, a in enumerate((a0, a1))]                                       # [N,17,1] each
        # Gate((96, 48)): the first 48 type-0 channels gate degree 1, the rest get SiLU (:228-257)
        v0 = self.to_values[1].mix(0, F.silu(val0[..., self.dh:]))
        v1 = self.to_values[1].mix(1, inter[1] * torch.sigmoid(val0[..., :self.dh])[..., None])
        gates = self.attn_head_gates[2](self.attn_head_gates[1](f0))                   # [N, 2]
        o0 = (attn[0] * v0).sum(1) * gates[:, :1]
        o1 = (attn[1][..., None] * v1).sum(1) * gates[:, 1:2, None]
        return self.to_out.mix(0, o0), self.to_out.mix(1, o1)


class FeedForwardFull(nn.Module):
    """equiformer_layer.py:485-529 with include_htype_norms=False, mult=4."""

    def __init__(self, c: int, mult: int = 4):
        super().__init__()
        self.c, self.mult = c, mult
        self.prenorm = FiberNorm((c, c))
        self.project_in = FiberLinear((c, c), (2 * mult * c, mult * c))
        self.project_out = FiberLinear((mult * c, mult * c), (c, c))

    def forward(self, x0, x1):
        h0 = self.project_in.mix(0, self.prenorm.norm0(x0))
        h1 = self.project_in.mix(1, self.prenorm.norm1(x1))
        m = self.mult * self.c
        g0 = F.silu(h0[..., m:])                                                       # Gate((8C, 4C))
        g1 = h1 * torch.sigmoid(h0[..., :m])[..., None]
        return self.project_out.mix(0, g0), self.project_out.mix(1, g1)


class _Blocks(nn.Module):
    def __init__(self, blocks):
        super().__init__()
        self.blocks = nn.ModuleList([nn.ModuleList([a, f]) for a, f in blocks])


class EquiformerFull(nn.Module):
    def __init__(self, dim: int, depth: int = 1, dim_head: int = 48, num_neighbors: int = 16, valid_radius: float = 5.0,
                 radial_hidden_dim: int = 64):
        super().__init__()
        self.k, self.radius = num_neighbors, valid_radius
        self.register_buffer("basis:(1,1)", torch.tensor([[0.57735027, 0.40824829, 0.18257419],
                                                            [0.57735027, 0.0, -0.36514837],
                                                            [0.57735027, -0.40824829, 0.18257419]]))
        self.tp_in = DTPFull((dim,), (dim, dim), pool=True, mid=radial_hidden_dim)
        self.layers = _Blocks([(MLPAttentionFull(dim, dim_head, radial_hidden_dim), FeedForwardFull(dim))
                               for _ in range(depth)])
        self.norm = FiberNorm((dim, dim))

    def forward(self, feats, coors):
        """feats [N,C], coors [N,3] -> (type0 [N,C], type1 [N,C,3])."""
        feats = 0.5 * feats + 0.5 * feats.detach()                                     # :1183-1186
        idx, dist, rel_pos, mask = neighbours_self_excluded(coors, self.k, self.radius)
        D = wigner_d1_to_y(rel_pos)
        b11 = getattr(self, "basis:(1,1)")
        x = self.tp_in({0: feats}, b11, idx, dist, mask, D)
        x0, x1 = x[0], x[1]
        for attn, ff in self.layers.blocks:                                            # reversible.py:251-257
            a0, a1 = attn(x0, x1, b11, idx, dist, mask, D)
            x0, x1 = x0 + a0, x1 + a1
            f0, f1 = ff(x0, x1)
            x0, x1 = x0 + f0, x1 + f1
        return self.norm.norm0(x0), self.norm.norm1(x1)                                # :1378,1392-1398
