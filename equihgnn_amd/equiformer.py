"""Equiformer front-end of ``equiformer_equihnns`` (equihnn_equiformer.py:37-49: dim=C, heads=1,
depth=1, dim_head=48, num_degrees=2, 16 neighbours, radius 5, MLPAttention, attend_self),
MI355X-native, with the reference's parameter / buffer names (equiformer_layer.py).

Only the computation that reaches the type-0 output is evaluated: the attention
block's degree-1 outputs, the degree-1 feed-forward, the final degree-1 norm and the (1,1) basis
path never reach the loss in the reference either (their parameters get grad=None there too).

How the radial tensor product is evaluated (the reference's dominant cost — it materialises
R[e, lo, li], 262 KB per edge at C=256, equiformer_layer.py:376-383,451-479):
    out[e, lo] = sum_li R[e,lo,li] x[e,li],  R[e] = reshape(W3 z_e + b3),  x[e] = xj[j] + xi[i]
               = z_e · (P[j] + Q[i])[:, lo] + (PB[j] + QB[i])[lo]
with P[n] = [sum_li W3[lo,li,k] xj[n,li]]_{k,lo} a NODE-level library GEMM.  What remains per edge is
a [deg x 64] x [64 x lo] product per node (ops.rowgemm, fp32 MFMA, csrc/rowgemm.hip), grouped by
sender through the transposed neighbour CSR and by receiver through the (contiguous) forward
lists.  For degree-1 inputs, (1->0): x[e,li] = r_hat_e · (xj1[j,li,:] + xi1[i,li,:]) folds r_hat
into the edge row: z'[e,(m,k)] = r_hat[e,m] z[e,k] against P1[n] = [(m,k), lo].
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .index import HyperIndex

import os as _os

# the POOLED (0,0) pair of tp_in through one product per pass instead of two (radial_contract_pooled; off: per-edge outputs, then the mean)
POOLED = not _os.environ.get("EQH_NO_EQF_POOLED")
# the b3 term of the radial networks inside the row-product launches (off: gathered and added per edge by ATen kernels)
BIAS_FOLD = not _os.environ.get("EQH_NO_EQF_BIAS_FOLD")
UNPADDED = not _os.environ.get("EQH_NO_EQF_UNPADDED")      # node matrices [mid, lo] at the attention's lo = 52 instead of 64 columns
FUSE_SMALL = not _os.environ.get("EQH_NO_EQF_FUSE")   # degree-1 Norm and masked means on the row kernels (off: the torch expressions, for same-box timing)


def _component_major(t) -> bool:
    """A degree-1 feature [N, d, 3] stored component-major ([N, 3, d] in memory, seen through .transpose(1, 2)): what the
    pooled (0 -> 1) pair produces (ops.pool3), so that every per-channel product downstream reads [3 N, d] rows in place."""
    return t.dim() == 3 and t.shape[2] == 3 and t.stride() == (3 * t.shape[1], 1, t.shape[1])


def _mix_cm(t, *ws):
    """(einsum("ndm,de->nem", t, w) for w in ws) for a component-major t, component-major results: one fan of [3 N, d]
    products, no transposed copies."""
    n, d, _ = t.shape
    outs = _fan(t.transpose(1, 2).reshape(3 * n, d), *ws)
    return tuple(o.view(n, 3, -1).transpose(1, 2) for o in outs)


class FiberLinear(nn.Module):
    """equiformer_layer.py:168-191 — ``weights.{i}`` of shape [d_in, d_out] per shared degree."""

    def __init__(self, fiber_in, fiber_out):
        super().__init__()
        self.weights = nn.ParameterList()
        self.degrees = []
        for deg, d_in in enumerate(fiber_in):
            if deg < len(fiber_out):
                self.weights.append(nn.Parameter(torch.randn(d_in, fiber_out[deg]) / math.sqrt(d_in)))
                self.degrees.append(deg)

    def w(self, deg: int) -> torch.Tensor:
        return self.weights[self.degrees.index(deg)]

    def init_zero_(self):
        for p in self.weights:
            p.data.zero_()


class FiberNorm(nn.Module):
    """equiformer_layer.py:194-225 — ``transforms.{deg}`` of shape [d, 1]."""

    def __init__(self, fiber, eps=1e-12):
        super().__init__()
        self.eps = eps
        self.transforms = nn.ParameterList([nn.Parameter(torch.ones(d, 1)) for d in fiber])

    def norm0(self, t):  # [N, d]
        if t.is_cuda and t.dim() == 2 and t.dtype == torch.float32 and t.shape[-1] % 4 == 0 and t.shape[-1] <= 1024:
            return ops.rms_norm_rows(t, self.transforms[0], self.eps)       # one launch each way (csrc/rmsnorm.hip)
        rms = t.norm(dim=-1, keepdim=True) * (t.shape[-1] ** -0.5)
        return t / rms.clamp(min=self.eps) * self.transforms[0][:, 0]

    def norm1(self, t):  # [N, d, 3]
        if (FUSE_SMALL and t.is_cuda and t.dtype == torch.float32 and _component_major(t) and t.shape[1] * 3 <= 1024
                and (t.shape[1] * 3) % 4 == 0):
            n, d, _ = t.shape
            out = ops.rms_norm_rows(t.transpose(1, 2).reshape(n, 3 * d), self.transforms[1], self.eps, rep=3, tiled=True)
            return out.view(n, 3, d).transpose(1, 2)
        if (FUSE_SMALL and t.is_cuda and t.dim() == 3 and t.dtype == torch.float32 and t.shape[-2] * 3 <= 1024
                and (t.shape[-2] * 3) % 4 == 0):
            # the same row kernel over the flattened [d, 3] block: one launch each way instead of 6 / 20 elementwise ones
            return ops.rms_norm_rows(t.flatten(-2), self.transforms[1], self.eps, rep=3).view_as(t)
        rms = t.flatten(-2).norm(dim=-1, keepdim=True)[..., None] * (t.shape[-2] ** -0.5)
        return t / rms.clamp(min=self.eps) * self.transforms[1]


class GammaLayerNorm(nn.Module):
    """equiformer_layer.py:158-165 — learnable ``gamma``, zero ``beta`` buffer."""

    def __init__(self, dim):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(dim))
        self.register_buffer("beta", torch.zeros(dim))

    def forward(self, x):
        return F.layer_norm(x, x.shape[-1:], self.gamma, self.beta)


class Radial(nn.Module):
    """equiformer_layer.py:451-479 — ``rp.{0,2,3,5,6}``; ``trunk`` is rp[0..5], the last Linear is
    never applied per edge (module docstring)."""

    def __init__(self, nc_in, nc_out, mid=64):
        super().__init__()
        self.nc_in, self.nc_out, self.mid = nc_in, nc_out, mid
        self.rp = nn.Sequential(nn.Linear(1, mid), nn.SiLU(), GammaLayerNorm(mid),
                                nn.Linear(mid, mid), nn.SiLU(), GammaLayerNorm(mid),
                                nn.Linear(mid, nc_in * nc_out))

    def trunk(self, dist_flat):  # [E, 1] -> [E, mid]
        rp = self.rp
        if ops.radial_trunk_supported(dist_flat, rp[0], rp[2], rp[3], rp[5]):
            return ops.radial_trunk(dist_flat, rp[0], rp[2], rp[3], rp[5])   # one launch each way (csrc/radial.hip)
        h = dist_flat
        for i in range(6):
            h = self.rp[i](h)
        return h

    def node_weights(self):
        """W3 as [li, mid * lo_p] (columns ordered (k, lo), lo zero-padded to a multiple of 16 where the row-product kernels
        need it) and b3 as [lo, li]."""
        lo, li, mid = self.nc_out, self.nc_in, self.mid
        lo_p = -(-lo // 16) * 16
        gpu = self.rp[6].weight.is_cuda and self.rp[6].weight.dtype == torch.float32
        if UNPADDED and gpu and lo_p != lo and ops.rowgemm_bias_supported(mid, lo):
            lo_p = lo      # the streaming row products take any width that is a multiple of 4 (the attention's 52): no padded columns
        if gpu:
            # one tiled transposition each way (16 MB at hidden 256; torch's strided copy moves it at 0.7 TB/s)
            return ops.radial_weight_layout(self.rp[6].weight, lo, li, mid, lo_p), self.rp[6].bias.view(lo, li), lo_p
        w = self.rp[6].weight.view(lo, li, mid).permute(1, 2, 0)
        if lo_p != lo:
            w = F.pad(w, (0, lo_p - lo))
        return w.reshape(li, mid * lo_p), self.rp[6].bias.view(lo, li), lo_p


class EdgeGeometry:
    """Neighbour lists and per-edge geometry, built once per batch (no gradient: positions are
    data, and the reference builds D under no_grad, equiformer/basis.py:194)."""

    def __init__(self, pos, index: HyperIndex, k: int, radius: float, full_d: bool = False):
        n = pos.shape[0]
        self.N, self.K = n, int(min(k, n - 1))
        nbr, dist, csr_t = index.knn(pos, self.K, 1)
        self.nbr, self.csr_t = nbr, csr_t
        self.nbr_flat = nbr.reshape(-1)
        # rel_pos, D[:, m=0] (with the reference's clamped rotation near -y and for coincident atoms), radius mask and
        # masked-mean weights: one launch (csrc/edge_geom.hip) instead of ~10 elementwise ones
        geo = ops.edge_geometry(pos, nbr, dist, radius, full_d)
        self.rhat, self.maskf, mean_w, self.mean_w_rhat = geo[:4]
        self.D = geo[4].view(n, self.K, 3, 3) if full_d else None        # whole D[1], for the degree-1 outputs
        self.dist = dist.reshape(-1, 1)                                   # [E, 1] true distance
        self.mask = self.maskf > 0                                        # [N, K]   :1339 (only the unfused attention reads it)
        self.recv_rowptr = torch.arange(0, (n + 1) * self.K, self.K, dtype=torch.int32, device=pos.device)
        self.mean_w = mean_w[:, None, :]                                  # [N, 1, K]  masked mean as a batched product

    def masked_mean(self, t):
        """equiformer/utils.py:71-82 over the K neighbour slots; t is [E, ...]."""
        shape = t.shape[1:]
        t3 = t.view(self.N, self.K, -1)
        if FUSE_SMALL and ops.attn_sum_supported(self.mean_w, t3):
            # one pass over t each way (the attention-sum kernel with one head): the batched [1 x K] . [K x C] products and
            # the contiguous copy of [N, K, C] in front of their backward took 0.1 ms per call in the library
            return ops.attn_sum(self.mean_w, t3).view(self.N, *shape)
        return torch.bmm(self.mean_w, t3).view(self.N, *shape)

    def masked_mean_times_rhat(self, t):
        """masked_mean(t[:, :, None] * r_hat[:, None, :]) for t [E, C] -> [N, C, 3] without the [E, C, 3]
        intermediate: one batched [C x K] . [K x 3] product per node."""
        t3 = t.view(self.N, self.K, -1)
        if FUSE_SMALL and t3.is_cuda and t3.dtype == torch.float32 and t3.shape[-1] % 4 == 0:
            # one pass each way (component-major result behind a transposed view); the gradient comes back as a contiguous [E, C]
            return ops.pool3(t3, self.mean_w_rhat).transpose(1, 2)
        return torch.bmm(t3.transpose(1, 2), self.mean_w_rhat)


def _fan(x, *ws):
    """(x @ w for w in ws) as one autograd node on the GPU (ops.matmul_fan: the input's gradient without add kernels, leaf
    weight gradients batched); plain matmuls elsewhere."""
    if x.is_cuda and x.dim() == 2 and x.dtype == torch.float32:
        return ops.matmul_fan(x, *ws)
    return tuple(x @ w for w in ws)


def radial_contract(radial: Radial, z, xj, xi, geo: EdgeGeometry, zscale=None):
    """out[e, lo] = sum_li R_e[lo, li] (xj[j_e, li] + xi[i_e, li]) without forming R_e.
    xj / xi: [N, li] (degree 0) or [N, li, 3] (degree 1, with ``zscale`` [E, 3]: the input vectors are contracted with it
    per edge -- r_hat for the (1 -> 0) pair).  ``zscale`` may be a LIST of [E, 3] tensors: the node-level products are
    formed once and one output per entry is returned (the three components of a (1 -> 1) pair)."""
    w, b3, lo_p = radial.node_weights()
    n, lo, mid = geo.N, radial.nc_out, radial.mid
    kd = mid if zscale is None else 3 * mid
    fold = BIAS_FOLD and z.is_cuda and z.dtype == torch.float32 and ops.rowgemm_bias_supported(kd, lo_p)
    b3t = b3.t()
    if fold and lo_p != lo:
        b3t = F.pad(b3t, (0, lo_p - lo))                                            # bias blocks as wide as the node matrices
    if zscale is None:
        p, pb = _fan(xj, w, b3t)                                                    # [N, mid*lo_p], [N, lo]
        q, qb = _fan(xi, w, b3t)
        p, q = p.view(n, mid, lo_p), q.view(n, mid, lo_p)
        if fold:    # out[e] = z_e . (P[j] + Q[i]) + pb[j] + qb[i] in the two row-product launches
            out = ops.rowgemm2(z, p, geo.csr_t.rowptr, geo.csr_t.perm, q, geo.recv_rowptr, None, pb.view(n, 1, lo_p),
                               qb.view(n, 1, lo_p))
            return out if lo_p == lo else out[:, :lo]
        bias = ops.gather_rows(pb, geo.nbr_flat, geo.csr_t).view(n, geo.K, lo) + qb[:, None, :]
        # sender rows (transposed neighbour CSR) and receiver rows both list every edge: one buffer, second pass adds
        out = ops.rowgemm2(z, p, geo.csr_t.rowptr, geo.csr_t.perm, q, geo.recv_rowptr, None)
        return out[:, :lo] + bias.reshape(-1, lo)
    # (one autograd node per input, as for degree 0: the products take the x6 / library dispatch of ops.matmul_fan and
    # their input gradients meet in accumulating GEMMs)
    li = xj.shape[1]
    p, pb = _fan(xj.transpose(1, 2).reshape(-1, li), w, b3t)                        # [3N, mid*lo_p], [3N, lo]
    q, qb = _fan(xi.transpose(1, 2).reshape(-1, li), w, b3t)
    p, q = p.view(n, 3 * mid, lo_p), q.view(n, 3 * mid, lo_p)                       # [(m,k), lo]
    if not fold:
        pb, qb = pb.view(n, 3 * lo), qb.view(n, 3, lo)                              # [N, (m,lo)], [N, 3, lo]
        g = (ops.gather_rows(pb, geo.nbr_flat, geo.csr_t).view(n, geo.K, 3, lo) + qb[:, None]).reshape(-1, 3, lo)
    outs = []
    for zs in (zscale if isinstance(zscale, (list, tuple)) else (zscale,)):
        if fold and mid == 64 and lo_p <= 64:
            # + sum_m zs[e, m] (pb[j, m] + qb[i, m]) inside the launches, and the row operand zs (x) z formed in their registers
            out = ops.rowgemm2(z, p, geo.csr_t.rowptr, geo.csr_t.perm, q, geo.recv_rowptr, None, pb.view(n, 3, lo_p),
                               qb.view(n, 3, lo_p), zs, z_factored=True)
            outs.append(out if lo_p == lo else out[:, :lo])
            continue
        ze = (zs[:, :, None] * z[:, None, :]).reshape(-1, 3 * mid)                  # [E, (m,k)]
        if fold:    # + sum_m zs[e, m] (pb[j, m] + qb[i, m]) inside the launches
            out = ops.rowgemm2(ze, p, geo.csr_t.rowptr, geo.csr_t.perm, q, geo.recv_rowptr, None, pb.view(n, 3, lo_p),
                               qb.view(n, 3, lo_p), zs)
            outs.append(out if lo_p == lo else out[:, :lo])
        else:
            out = ops.rowgemm2(ze, p, geo.csr_t.rowptr, geo.csr_t.perm, q, geo.recv_rowptr, None)
            outs.append(out[:, :lo] + (g * zs[:, :, None]).sum(1))
    return outs if isinstance(zscale, (list, tuple)) else outs[0]


def radial_contract_pooled(radial: Radial, z, xj, xi, geo: EdgeGeometry):
    """masked_mean over the receiver's neighbour slots of radial_contract(radial, z, xj, xi, geo) -- the pooled output of
    equiformer_layer.py:383,432-436 -- WITHOUT the per-edge outputs.  The mean is linear, so with w_e its weights
        p[i, lo] = sum_e w_e sum_li R_e[lo, li] x_e[li]                       x_e = xj[j_e] + xi[i],  R_e = reshape(W3 z_e + b3)
                 = sum_(li,k) W3[lo, li, k] Y[i, li, k] + sum_li b3[lo, li] xbar[i, li],
        Y[i] = sum_e x_e^T (x) (w_e z_e)   (a [li x 16] . [16 x mid] product per node, ops.row_outer),   xbar[i] = sum_e w_e x_e:
    ONE node-level [N x li*mid] . [li*mid x lo] product where the per-edge form needs two ([N x li] . [li x mid*lo] for the
    senders' and the receivers' node matrices), and two instead of four in the backward pass; W3 is used in the parameter's
    own layout (no re-laid copy of the weight or of its gradient)."""
    n, k = geo.N, geo.K
    xe = (ops.gather_rows(xj, geo.nbr_flat, geo.csr_t).view(n, k, -1) + xi[:, None, :])          # [N, K, li]
    zw = (z.view(n, k, -1) * geo.mean_w.view(n, k, 1)).reshape(n * k, -1)                        # [E, mid]
    y = ops.row_outer(xe.reshape(n * k, -1), zw, geo.recv_rowptr)                                 # [N, li, mid]
    xbar = geo.masked_mean(xe.reshape(n * k, -1))                                                 # [N, li]
    return ops.pooled_radial(y, xbar, radial.rp[6].weight, radial.rp[6].bias, radial.nc_out)


def pooled_supported(radial: Radial, z) -> bool:
    return (POOLED and z.is_cuda and z.dtype == torch.float32 and radial.mid == 64 and radial.nc_in == 256
            and radial.nc_out % 4 == 0)


class DTPIn(nn.Module):
    """tp_in = DTP((C,), (C, C)), pooled (equiformer_layer.py:260-448, built at :1081-1086)."""

    def __init__(self, c, mid=64):
        super().__init__()
        self.to_xi = FiberLinear((c,), (c,))
        self.to_xj = FiberLinear((c,), (c,))
        self.kernel_unary = nn.ModuleDict({"(0,0)": Radial(c, c, mid), "(0,1)": Radial(c, c, mid)})
        self.self_interact = FiberLinear((c,), (c, c))
        self.to_out = FiberLinear((c, c), (c, c))

    def forward(self, x0, geo: EdgeGeometry):
        xi, xj, si = _fan(x0, self.to_xi.w(0), self.to_xj.w(0), self.self_interact.w(0))
        r00, r01 = self.kernel_unary["(0,0)"], self.kernel_unary["(0,1)"]
        z00 = r00.trunk(geo.dist)
        if pooled_supported(r00, z00):
            p0 = radial_contract_pooled(r00, z00, xj, xi, geo)                      # [N, C]
        else:
            p0 = geo.masked_mean(radial_contract(r00, z00, xj, xi, geo))            # [E, C] -> [N, C]
        o1 = radial_contract(r01, r01.trunk(geo.dist), xj, xi, geo)                 # [E, C]
        p1 = geo.masked_mean_times_rhat(o1)                                         # [N, C, 3]
        out0 = _fan(p0, self.to_out.w(0))[0] + si
        if p1.is_cuda and _component_major(p1):
            out1 = _mix_cm(p1, self.to_out.w(1))[0]
        else:
            out1 = torch.einsum("ndm,de->nem", p1, self.to_out.w(1))
        return out0, out1


class DTPAttn(nn.Module):
    """to_attn_and_v = DTP((C,C), (104,48), pool=False, self-interaction as slot 0); the (·,1) radial
    nets exist as parameters only (dead for the type-0 output)."""

    def __init__(self, c, d0=104, d1=48, mid=64):
        super().__init__()
        self.to_xi = FiberLinear((c, c), (c, c))
        self.to_xj = FiberLinear((c, c), (c, c))
        self.kernel_unary = nn.ModuleDict({
            "(0,0)": Radial(c, (d0 + 1) // 2, mid), "(1,0)": Radial(c, d0 // 2, mid),
            "(0,1)": Radial(c, (d1 + 1) // 2, mid), "(1,1)": Radial(c, d1 // 2, mid)})
        self.self_interact = FiberLinear((c, c), (d0, d1))
        self.to_out = FiberLinear((d0, d1), (d0, d1))

    def forward(self, f0, f1, geo: EdgeGeometry, joined: bool = True):
        """[N, 1+K, 104] (self-interaction as slot 0), or with ``joined=False`` its two parts: the self rows
        [N, 104] and the edge rows [E, 104]."""
        xi0, xj0, me = _fan(f0, self.to_xi.w(0), self.to_xj.w(0), self.self_interact.w(0))
        if f1.is_cuda and _component_major(f1):
            xi1, xj1 = _mix_cm(f1, self.to_xi.w(1), self.to_xj.w(1))
        else:
            xi1 = torch.einsum("ndm,de->nem", f1, self.to_xi.w(1))
            xj1 = torch.einsum("ndm,de->nem", f1, self.to_xj.w(1))
        r00, r10 = self.kernel_unary["(0,0)"], self.kernel_unary["(1,0)"]
        o00 = radial_contract(r00, r00.trunk(geo.dist), xj0, xi0, geo)
        o10 = radial_contract(r10, r10.trunk(geo.dist), xj1, xi1, geo, zscale=geo.rhat)
        out = _fan(torch.cat((o00, o10), -1), self.to_out.w(0))[0]                  # [E, 104]
        if not joined:
            return me, out
        return torch.cat((me[:, None, :], out.view(geo.N, geo.K, -1)), 1)           # [N, 1+K, 104]


class MLPAttention(nn.Module):
    """equiformer_layer.py:743-955 (heads=1, dim_head=48); degree-0 output."""

    def __init__(self, c, dim_head=48, mid=64):
        super().__init__()
        self.dh, self.scale = dim_head, dim_head ** -0.5
        self.prenorm = FiberNorm((c, c))
        self.to_attn_and_v = DTPAttn(c, 8 + 2 * dim_head, dim_head, mid)
        self.to_attn_logits = nn.ModuleList([nn.Sequential(nn.LeakyReLU(0.1), nn.Linear(4, 1, bias=False))
                                             for _ in range(2)])
        self.to_values = nn.Sequential(nn.Identity(), FiberLinear((dim_head, dim_head), (dim_head, dim_head)))
        self.attn_head_gates = nn.Sequential(nn.Identity(), nn.Linear(c, 2), nn.Sigmoid(), nn.Identity())
        self.to_out = FiberLinear((dim_head, dim_head), (c, c))
        self.to_out.init_zero_()                                                    # :865-868

    def forward(self, x0, x1, geo: EdgeGeometry):
        f0, f1 = self.prenorm.norm0(x0), self.prenorm.norm1(x1)
        wl, wv = self.to_attn_logits[0][1].weight, self.to_values[1].w(0)
        if geo.K == 16 and f0.is_cuda:
            me, edge = self.to_attn_and_v(f0, f1, geo, joined=False)
            if ops.attn_pool_supported(me, edge, geo.maskf, wl, wv, 8 + self.dh):
                # logits, masked softmax over the 17 slots, SiLU gate + value Linear and the weighted sum in one
                # launch each way; the [N, 17, 104] concatenation is never built (csrc/attn_pool.hip)
                out = ops.attn_pool(me, edge, geo.maskf, wl, wv, 8 + self.dh, self.scale, 0.1)
                gate = torch.sigmoid(self.attn_head_gates[1](f0))[:, :1]
                return _fan(out * gate, self.to_out.w(0))[0]
            inter = torch.cat((me[:, None, :], edge.view(geo.N, geo.K, -1)), 1)
        else:
            inter = self.to_attn_and_v(f0, f1, geo)                                 # [N, 1+K, 104]
        logits = self.to_attn_logits[0](inter[..., :4]) * self.scale                # [N, 1+K, 1]
        keep = F.pad(geo.mask, (1, 0), value=True)[..., None]                       # self always valid
        attn = logits.masked_fill(~keep, -torch.finfo(logits.dtype).max).softmax(dim=1)
        v = F.silu(inter[..., 8 + self.dh:]) @ self.to_values[1].w(0)               # Gate + Linear
        out = (attn * v).sum(1)
        gate = torch.sigmoid(self.attn_head_gates[1](f0))[:, :1]
        return (out * gate) @ self.to_out.w(0)


class FeedForward(nn.Module):
    """equiformer_layer.py:485-529 (include_htype_norms=False, mult=4); degree-0 output."""

    def __init__(self, c, mult=4):
        super().__init__()
        self.c, self.mult = c, mult
        self.prenorm = FiberNorm((c, c))
        self.project_in = FiberLinear((c, c), (2 * mult * c, mult * c))
        self.project_out = FiberLinear((mult * c, mult * c), (c, c))
        self.project_out.init_zero_()                                               # :514-515

    def forward(self, x0):
        # only the SiLU half of project_in reaches the type-0 output (the other 4c columns gate the degree-1 features, :517-529):
        # the product is formed for those columns alone (their weight gradient comes back zero-padded, as the reference's is)
        w = self.project_in.w(0)[:, self.mult * self.c:]
        if x0.is_cuda:
            w = w.contiguous()
        h = _fan(self.prenorm.norm0(x0), w)[0]
        return _fan(F.silu(h), self.project_out.w(0))[0]


# ---------------------------------------------------------------------------------------------------------------
# Degree-1 outputs.  The reference's wrapper builds the layer with depth 1 and reads type 0 only, so
# everything below is dead there; with depth > 1 block t+1's (1 -> 0) pair consumes block t's degree-1 output and it
# all becomes live.  No registry name reaches this configuration.  The radial tensor products of the degree-1 outputs
# run on the same row-GEMM kernels as the degree-0 ones (round 3: no per-edge radial weights are formed); the small
# per-edge rotations, gates and the degree-1 attention are torch operations on the device tensors.
# tests/test_equiformer_layer.py pins it to the reference's layer at depth 1-3.
# ---------------------------------------------------------------------------------------------------------------
def _mix1(lin: FiberLinear, t):
    """Degree-1 channel mix [..., d, 3] -> [..., e, 3] (equiformer_layer.py:186-189)."""
    return torch.einsum("...dm,de->...em", t, lin.w(1))


def attention_degree1(att: "MLPAttention", f0, f1, me, edge, geo: EdgeGeometry, basis11):
    """Degree-1 output of the attention block (equiformer_layer.py:871-955 with the (0->1) / (1->1) pairs of its DTP,
    :385-418): [N, C, 3].  ``me`` [N, 104] / ``edge`` [E, 104] are the degree-0 intermediates (self rows, edge rows)."""
    n, k, dh = geo.N, geo.K, att.dh
    dtp = att.to_attn_and_v
    D = geo.D
    # The (0 -> 1) and (1 -> 1) pairs through the SAME re-association as the degree-0 ones: the per-edge
    # radial weights R_e[lo, li] (262 KB per edge at C = 256) are never formed.  (0 -> 1) is a plain radial contraction of
    # the degree-0 inputs.  For (1 -> 1), :364-366 rotates the input into the edge frame (x1r = D^T x1), :389-404 combines
    # its components with the 3 x 3 basis (component m of the result reads x1r[m] and x1r[2 - m]), so output component m
    # is the radial contraction of the UNROTATED node vectors with the per-edge 3-vector
    #     T_e[m, a] = (basis[m, 0] + basis[m, 2]) D_e[a, m] + basis[m, 1] D_e[a, 2 - m]
    # folded into the edge row, exactly as r_hat is for the (1 -> 0) pair: three row-GEMM passes over one pair of
    # node-level products.
    r01, r11 = dtp.kernel_unary["(0,1)"], dtp.kernel_unary["(1,1)"]
    dflat = geo.dist
    o01 = radial_contract(r01, r01.trunk(dflat), f0 @ dtp.to_xj.w(0), f0 @ dtp.to_xi.w(0), geo)  # [E, lo01]
    o01 = F.pad(o01.view(n, k, -1)[..., None], (1, 1))                                           # result at m = 1, :407-409
    Df = D.reshape(n * k, 3, 3)
    c0, c1 = basis11[:, 0] + basis11[:, 2], basis11[:, 1]
    T = [c0[m] * Df[:, :, m] + c1[m] * Df[:, :, 2 - m] for m in range(3)]                        # 3 x [E, 3]
    o11 = radial_contract(r11, r11.trunk(dflat), _mix1(dtp.to_xj, f1), _mix1(dtp.to_xi, f1), geo, zscale=T)
    o11 = torch.stack(o11, -1).view(n, k, -1, 3)                                                 # [N,K,lo11,3]
    out1 = torch.einsum("nklm,nkam->nkla", torch.cat((o01, o11), 2), D)                          # rotate out, :416-418
    inter1 = torch.cat((_mix1(dtp.self_interact, f1)[:, None], _mix1(dtp.to_out, out1)), 1)      # [N,1+K,48,3]
    inter0 = torch.cat((me[:, None], edge.view(n, k, -1)), 1)                                    # [N,1+K,104]
    logits = att.to_attn_logits[1](inter0[..., 4:8]) * att.scale                                 # degree-1 logits, :905-909
    keep = F.pad(geo.mask, (1, 0), value=True)[..., None]
    attn = logits.masked_fill(~keep, -torch.finfo(logits.dtype).max).softmax(dim=1)              # [N,1+K,1]
    v1 = _mix1(att.to_values[1], inter1 * torch.sigmoid(inter0[..., 8:8 + dh])[..., None])      # Gate + Linear, :246,922
    gate = torch.sigmoid(att.attn_head_gates[1](f0))[:, 1:2, None]
    return _mix1(att.to_out, (attn[..., None] * v1).sum(1) * gate)


def feed_forward_degree1(ff: "FeedForward", x0, x1):
    """Degree-1 output of the feed-forward (equiformer_layer.py:517-529, Gate :228-257): [N, C, 3]."""
    m = ff.mult * ff.c
    h0 = ff.prenorm.norm0(x0) @ ff.project_in.w(0)
    h1 = _mix1(ff.project_in, ff.prenorm.norm1(x1))
    return _mix1(ff.project_out, h1 * torch.sigmoid(h0[..., :m])[..., None])


class _HalfGrad(torch.autograd.Function):
    """x -> x with d/dx = 0.5 (see Equiformer.forward)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g * 0.5


class _Blocks(nn.Module):
    def __init__(self, blocks):
        super().__init__()
        self.blocks = nn.ModuleList([nn.ModuleList([attn, ff]) for attn, ff in blocks])


class Equiformer(nn.Module):
    """equiformer_layer.py:961-1398 as equihnn_equiformer.py:37-49 configures it.  ``depth`` = 1 and ``type1`` = False
    is that wrapper's use (type-0 output, the fused live path); ``type1`` = True returns (type0, type1), and any
    ``depth`` > 1 evaluates the degree-1 paths between the blocks."""

    def __init__(self, dim, dim_head=48, num_neighbors=16, valid_radius=5.0, radial_hidden_dim=64, depth=1, type1=False):
        super().__init__()
        self.k, self.radius = num_neighbors, float(valid_radius)
        self.depth, self.type1 = int(depth), bool(type1)
        # the (1,1) basis of equiformer/basis.py:116-163 is a constant that only the (dead) (1,1)
        # path reads; it is carried as data for state_dict compatibility, never recomputed
        self.register_buffer("basis:(1,1)", torch.tensor([[0.57735027, 0.40824829, 0.18257419],
                                                            [0.57735027, 0.0, -0.36514837],
                                                            [0.57735027, -0.40824829, 0.18257419]]))
        self.tp_in = DTPIn(dim, radial_hidden_dim)
        self.layers = _Blocks([(MLPAttention(dim, dim_head, radial_hidden_dim), FeedForward(dim)) for _ in range(self.depth)])
        self.norm = FiberNorm((dim, dim))

    def forward(self, feats, coors, index: HyperIndex = None):
        if index is None:           # the layer on its own (tests): a one-molecule index over the cloud
            one = torch.zeros(1, dtype=torch.int64, device=feats.device)
            index = HyperIndex(one, one, feats.shape[0], 1)
        # :1183-1186 `0.5 * feats + 0.5 * feats.detach()`: the VALUE is feats exactly (0.5 f is exact in binary floating point and
        # so is the sum of the two halves), the gradient is halved -- one scaling in the backward pass instead of three
        # elementwise launches forward and one backward
        feats = _HalfGrad.apply(feats)
        full = self.type1 or self.depth > 1
        geo = EdgeGeometry(coors, index, self.k, self.radius, full_d=full)
        x0, x1 = self.tp_in(feats, geo)
        basis11 = getattr(self, "basis:(1,1)")
        for i, (attn, ff) in enumerate(self.layers.blocks):                          # reversible.py:251-257
            need1 = self.type1 or i + 1 < self.depth          # is this block's degree-1 output read by anything?
            if need1:
                f0, f1 = attn.prenorm.norm0(x0), attn.prenorm.norm1(x1)
                me, edge = attn.to_attn_and_v(f0, f1, geo, joined=False)
                a1 = attention_degree1(attn, f0, f1, me, edge, geo, basis11)
            x0n = x0 + attn(x0, x1, geo)
            if need1:
                x1 = x1 + a1
            x0 = x0n
            if need1:
                x1n = x1 + feed_forward_degree1(ff, x0, x1)
            x0 = x0 + ff(x0)
            if need1:
                x1 = x1n
        if self.type1:
            return self.norm.norm0(x0), self.norm.norm1(x1)                          # :1378,1392-1398
        return self.norm.norm0(x0)
