"""GPU parity of FAFormer's fused geometric steps (csrc/faformer_geom.hip) against the torch expression of the same
step -- the expression the model evaluated before, itself pinned to the reference by the FAFormer fixtures
(tests/test_hip_models.py) -- in float64 where the step has no eigenvectors in it, forward and every gradient."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _ops():
    from equihgnn_amd import ops
    return ops


def _rel(a, r):
    return float((a.detach().cpu().double() - r.detach().cpu().double()).abs().max() / r.detach().abs().max().clamp(min=1e-9))


@pytest.mark.parametrize("N,masked", [(1, False), (300, False), (15744, False), (15744, True), (70000, True)])
def test_centre_mix_matches_float64(N, masked):
    """faf_centre_mix_fwd / _bwd: c g + x (1 - g) with the (masked) centroid; the parameters of the cancelled term get an
    exactly-zero gradient; two runs are bitwise equal (fixed-order float64 partial sums)."""
    ops = _ops()
    g = torch.Generator().manual_seed(N)
    geo, logit, wo = 3 * torch.randn(N, 3, generator=g) + 1.5, torch.randn(N, 1, generator=g), torch.randn(N, 3, generator=g)
    mask = (torch.rand(N, 1, generator=g) > 0.3).float() if masked else None
    if masked:
        geo = torch.where(mask > 0, geo, torch.full_like(geo, 1e4))       # padding rows may hold anything
    t = [a.double().requires_grad_(True) for a in (geo, logit)]
    m64 = mask.double() if masked else torch.ones(N, 1, dtype=torch.float64)
    c = (torch.where(m64 > 0, t[0], torch.zeros((), dtype=torch.float64)).sum(0, keepdim=True) / m64.sum()).float().double()
    gate = torch.sigmoid(t[1])
    ref = c * gate + t[0] * (1 - gate)
    # (the centroid is ROUNDED to fp32 in the op, as in the model; its gradient passes through unchanged)
    c_true = torch.where(m64 > 0, t[0], torch.zeros((), dtype=torch.float64)).sum(0, keepdim=True) / m64.sum()
    ((c_true * gate + t[0] * (1 - gate)) * wo.double()).sum().backward()
    w, b = torch.nn.Parameter(torch.randn(1, 2, device=DEV)), torch.nn.Parameter(torch.randn(1, device=DEV))
    runs = []
    for _ in range(2):
        d = [a.to(DEV).requires_grad_(True) for a in (geo, logit)]
        w.grad = b.grad = None
        out = ops.centre_mix(d[0], d[1], mask.to(DEV) if masked else None, (w, b))
        (out * wo.to(DEV)).sum().backward()
        runs.append((out.detach().clone(), d[0].grad.clone(), d[1].grad.clone()))
    out, dgeo, dlogit = runs[0]
    assert all(torch.equal(a, b_) for a, b_ in zip(runs[0], runs[1]))
    assert float(w.grad.abs().max()) == 0.0 and float(b.grad.abs().max()) == 0.0
    keep = (mask > 0).expand(-1, 3) if masked else torch.ones(N, 3, dtype=torch.bool)
    assert _rel(out[keep.to(DEV)], ref[keep]) < 2e-6
    assert _rel(dgeo, t[0].grad) < 5e-6, _rel(dgeo, t[0].grad)
    assert _rel(dlogit, t[1].grad) < 5e-6


@pytest.mark.parametrize("N,masked", [(5, False), (1000, False), (15744, True), (70000, False)])
def test_cloud_frame_matches_the_unfused_frame(N, masked):
    """faf_cloud_frame_fwd / _bwd against faformer._frame_axes (float64 centroid / covariance, geo_eigh3, projection) on
    an anisotropic cloud (well separated eigenvalues: the comparison is then not one of eigenvector conditioning)."""
    from equihgnn_amd import faformer
    ops = _ops()
    g = torch.Generator().manual_seed(N + 1)
    x = (torch.randn(N, 3, generator=g) * torch.tensor([1.0, 2.0, 3.5]) + torch.tensor([0.3, -1.0, 2.0]))
    rot = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
    x = (x @ rot).to(DEV)
    mask = (torch.rand(N, 1, generator=g) > 0.25).float().to(DEV) if masked else None
    wo = torch.randn(N, 3, generator=g).to(DEV)
    xa = x.clone().requires_grad_(True)
    y_ref, vec, centre = faformer._frame_axes(xa.unsqueeze(0), None if mask is None else mask.view(1, -1))
    (y_ref[0] * wo).sum().backward()
    xb = x.clone().requires_grad_(True)
    y = ops.cloud_frame(xb, mask)
    (y * wo).sum().backward()
    scale = float(y_ref.detach().abs().max())
    assert float((y - y_ref[0]).abs().max()) < 2e-5 * scale, float((y - y_ref[0]).abs().max()) / scale
    assert float((xb.grad - xa.grad).abs().max()) < 2e-5 * float(xa.grad.abs().max())
    y2 = ops.cloud_frame(x, mask)
    assert torch.equal(y2, y.detach())


def _edge_case(N, K, seed, frac_masked=0.3):
    g = torch.Generator().manual_seed(seed)
    geo = 2.0 * torch.randn(N, 3, generator=g)
    nbr = torch.randint(0, N, (N, K), generator=g)
    mask = torch.rand(N, K, generator=g) > frac_masked
    mask[::7] = False                                   # atoms without any neighbour inside the radius
    if N > 3:
        mask[3, :] = False
        mask[3, 0] = True                               # exactly one
    return geo.to(DEV), nbr.to(DEV), mask.to(DEV), g


@pytest.mark.parametrize("N,K", [(1, 16), (50, 16), (4000, 16), (333, 5)])
def test_edge_frame_matches_the_unfused_frame(N, K):
    """faf_edge_frame_fwd / _bwd against offsets + |.|^2 + faformer._frame_axes per atom; both gradient paths of the
    coordinates (receiver side directly, sender side through the gathered rows)."""
    from equihgnn_amd import faformer
    ops = _ops()
    geo, nbr, mask, g = _edge_case(N, K, 100 + N)
    wy, wd = torch.randn(N, K, 3, generator=g).to(DEV), torch.randn(N, K, 1, generator=g).to(DEV)

    def unfused(x):
        gj = F.pad(x, (0, 1))[nbr]
        rel = x.unsqueeze(1) - gj[..., :3]
        d2 = (rel ** 2).sum(-1, keepdim=True)
        return faformer._frame_axes(rel, mask)[0], d2

    def fused(x):
        gj = F.pad(x, (0, 1))[nbr]
        assert ops.edge_frame_supported(x, gj, mask)
        return ops.edge_frame(x, gj, mask)

    res = []
    for fn in (unfused, fused):
        x = geo.clone().requires_grad_(True)
        y, d2 = fn(x)
        ((y * wy).sum() + (d2 * wd).sum()).backward()
        res.append((y.detach(), d2.detach(), x.grad.clone()))
    (y0, d0, g0), (y1, d1, g1) = res
    np.testing.assert_allclose(d1.cpu().numpy(), d0.cpu().numpy(), rtol=1e-6, atol=1e-6)
    # eigenvectors of a 16-point covariance: a handful of atoms have close eigenvalues, where a last-bit difference of the
    # covariance sum turns the axes visibly; bound those and require everything else to agree to rounding
    for a, b in ((y1, y0), (g1, g0)):
        err = (a - b).abs() / b.abs().max().clamp(min=1.0 if N == 1 else 1e-9)   # (N = 1: the atom is its own neighbour and
        #                                                                          everything cancels: compare absolutely)
        assert float((err > 2e-5).float().mean()) < 2e-3, float((err > 2e-5).float().mean())
        assert float(err.max()) < 5e-2, float(err.max())
    # only d2's gradient (no frame in it): exact to rounding
    x = geo.clone().requires_grad_(True)
    (fused(x)[1] * wd).sum().backward()
    x0 = geo.clone().requires_grad_(True)
    (unfused(x0)[1] * wd).sum().backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), x0.grad.cpu().numpy(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("N,K,H,p", [(1, 16, 2, 0.0), (500, 16, 2, 0.0), (15744, 16, 2, 0.0), (300, 7, 1, 0.0), (2000, 16, 2, 0.25)])
def test_attn_logits_matches_float64(N, K, H, p):
    """faf_attn_logits_fwd / _bwd: a_q[i] + a_k[j] + l_e -> masked_fill(-1e9) -> softmax over the slots -> dropout, in
    float64 with the op's own keep decisions (read off its output)."""
    ops = _ops()
    geo, nbr, mask, g = _edge_case(N, K, 200 + N)
    qa = torch.randn(N, 4, generator=g)
    le = torch.randn(N, K, H, generator=g)
    wo = torch.randn(N, H, K, generator=g)
    seed = torch.tensor([13572468], dtype=torch.int64, device=DEV)
    d = [a.to(DEV).requires_grad_(True) for a in (qa, le)]
    qan = d[0][nbr]
    assert ops.attn_logits_supported(d[0], d[1], mask)
    attn = ops.attn_logits(d[0], qan, d[1], mask, p, seed)
    (attn * wo.to(DEV)).sum().backward()
    t = [a.double().requires_grad_(True) for a in (qa, le)]
    logits = t[0][:, None, :H] + t[0][nbr.cpu()][..., H:2 * H] + t[1]
    logits = logits.masked_fill(~mask.cpu().unsqueeze(-1), -1e9)
    prob = logits.transpose(1, 2).softmax(-1)
    if p > 0:
        a0 = ops.attn_logits(d[0].detach(), qan.detach(), d[1].detach(), mask, 0.0)
        live = a0 > 0                                    # (masked slots are exactly 0 before the dropout)
        keep = torch.where((attn.detach() != 0) | ~live, torch.full_like(a0, 1.0 / (1.0 - p)), torch.zeros_like(a0)).cpu().double()
        frac = float((keep[live.cpu()] == 0).double().mean())
        assert abs(frac - p) < 0.02, frac
        assert torch.equal(attn.detach(), ops.attn_logits(d[0].detach(), qan.detach(), d[1].detach(), mask, p, seed))
    else:
        keep = torch.ones_like(prob)
    ref = prob * keep
    (ref * wo.double()).sum().backward()
    np.testing.assert_allclose(attn.detach().cpu().numpy(), ref.detach().numpy(), atol=2e-6, rtol=1e-5)
    assert _rel(d[1].grad, t[1].grad) < 2e-5
    assert _rel(d[0].grad, t[0].grad) < 2e-5
    assert float(d[0].grad[:, 2 * H:].abs().max()) == 0.0 if 2 * H < 4 else True


@pytest.mark.parametrize("masked", [False, True])
def test_faformer_with_fused_geometry_matches_the_unfused_model(masked):
    """The whole front end with and without csrc/faformer_geom.hip (ops.USE_GEOM): output and every gradient."""
    from equihgnn_amd import faformer
    from equihgnn_amd.index import HyperIndex
    ops = _ops()
    torch.manual_seed(3)
    n, c = 300, 64
    model = faformer.FAFormer(c, proj_drop=0.0, attn_drop=0.0).to(DEV)
    pos = (2.5 * torch.randn(n, 3)).to(DEV)
    feats, wo = torch.randn(n, c, device=DEV), torch.randn(n, c, device=DEV)
    row_mask = None
    if masked:
        row_mask = torch.ones(n, 1, device=DEV)
        row_mask[-40:] = 0.0
        pos[-40:] = 1e3 + 50.0 * torch.arange(40, device=DEV)[:, None]       # padding atoms far away from everything
    one = torch.zeros(1, dtype=torch.int64, device=DEV)
    res = []
    for on in (False, True):
        ops.USE_GEOM = on
        try:
            model.zero_grad(set_to_none=True)
            out = model(feats, pos, HyperIndex(one, one, n, 1), row_mask)
            keep = slice(0, n - 40) if masked else slice(0, n)
            (out[keep] * wo[keep]).sum().backward()
            res.append((out[keep].detach().clone(), {k: v.grad.clone() for k, v in model.named_parameters() if v.grad is not None}))
        finally:
            ops.USE_GEOM = True
    (o0, g0), (o1, g1) = res
    assert float((o1 - o0).abs().max()) < 2e-5 * float(o0.abs().max())
    assert g0.keys() == g1.keys()
    for k in g0:
        assert float((g1[k] - g0[k]).abs().max()) <= 2e-4 * float(g0[k].abs().max()) + 1e-7, k


@pytest.mark.parametrize("N,K,H,D,block", [(1, 16, 2, 128, False), (700, 16, 2, 128, True), (15744, 16, 2, 128, True),
                                           (333, 5, 1, 64, False), (900, 16, 2, 32, True)])
def test_attn_gather_sum_matches_float64(N, K, H, D, block):
    """faf_attn_gather_sum_fwd / _bwd against the einsum of fa_former_layer.py:497-506 over explicitly gathered rows in
    float64: output, dattn and dx (through the transposed neighbour CSR); ``block``: x is a column block of a wider
    matrix, read in place; two runs are bitwise equal."""
    ops = _ops()
    g = torch.Generator().manual_seed(N + K + D)
    C = H * D
    nbr = torch.randint(0, N, (N, K), generator=g).to(torch.int32)
    attn = torch.randn(N, H, K, generator=g).softmax(-1)
    wide = torch.randn(N, 3 * C if block else C, generator=g)
    w = torch.randn(N, C, generator=g)
    csr_t = ops.csr_build(nbr.reshape(-1).to(DEV), None, N)
    runs = []
    for _ in range(2):
        a, xw = attn.to(DEV).requires_grad_(True), wide.to(DEV).requires_grad_(True)
        x = xw[:, 2 * C:] if block else xw
        assert ops.attn_gather_sum_supported(H, x, nbr.to(DEV), csr_t)
        out = ops.attn_gather_sum(a, x, nbr.to(DEV), csr_t)
        (out * w.to(DEV)).sum().backward()
        runs.append((out.detach().clone(), a.grad.clone(), xw.grad.clone()))
    assert all(torch.equal(p, q) for p, q in zip(runs[0], runs[1]))
    a64, x64 = attn.double().requires_grad_(True), wide.double().requires_grad_(True)
    xs = x64[:, 2 * C:] if block else x64
    ref = torch.einsum("nhm,nmhd->nhd", a64, xs[nbr.long()].view(N, K, H, D)).reshape(N, -1)
    (ref * w.double()).sum().backward()
    out, da, dx = runs[0]
    np.testing.assert_allclose(out.cpu().numpy(), ref.detach().numpy(), atol=2e-6, rtol=1e-6)
    np.testing.assert_allclose(da.cpu().numpy(), a64.grad.numpy(), atol=2e-5, rtol=1e-5)
    assert _rel(dx, x64.grad) < 5e-6


@pytest.mark.parametrize("acc", [False, True])
def test_edge_logit_weights_match_float64(acc):
    """faf_edge_logit_weights_fwd / _bwd against the torch expression of the fold in float64; with persistent accumulators
    (the graphed trainer's) the gradients are added in place and autograd gets None."""
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    h, deh = 2, 128
    de = h * deh
    W = torch.nn.Parameter((0.1 * torch.randn(2 * de, de, generator=g)).to(DEV))
    b = torch.nn.Parameter(torch.randn(2 * de, generator=g).to(DEV))
    we = torch.nn.Parameter(torch.randn(1, deh, generator=g).to(DEV))
    wu, wc = torch.randn(h, de, generator=g), torch.randn(h, generator=g)
    t = [p.detach().cpu().double().requires_grad_(True) for p in (W, b, we)]
    u64 = (t[0][:de].reshape(h, deh, de) * t[2].view(-1)[None, :, None]).sum(1)
    c64 = (t[1][:de].reshape(h, deh) * t[2].view(-1)).sum(-1)
    ((u64 * wu.double()).sum() + (c64 * wc.double()).sum()).backward()
    base = [torch.randn_like(p) for p in (W, b, we)]
    if acc:
        for p, a in zip((W, b, we), base):
            p._eqh_gbuf = a.clone()
    try:
        u, c = ops.edge_logit_weights(W, b, we, h)
        ((u * wu.to(DEV)).sum() + (c * wc.to(DEV)).sum()).backward()
        np.testing.assert_allclose(u.detach().cpu().numpy(), u64.detach().numpy(), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(c.detach().cpu().numpy(), c64.detach().numpy(), rtol=1e-5, atol=1e-5)
        for p, r, a in zip((W, b, we), t, base):
            if acc:
                assert p.grad is None
                got = p._eqh_gbuf - a
            else:
                got = p.grad
            assert _rel(got, r.grad.view_as(got)) < 1e-5
    finally:
        for p in (W, b, we):
            if hasattr(p, "_eqh_gbuf"):
                del p._eqh_gbuf


@pytest.mark.parametrize("R,C,J,alias", [(1, 256, 2, True), (777, 256, 2, True), (20000, 256, 2, False), (301, 64, 1, True),
                                         (129, 512, 2, True), (70, 1024, 1, False)])
def test_ln_rowdot_matches_float64(R, C, J, alias):
    """faf_ln_rowdot_fwd / _bwd: (LayerNorm(x), its row dots with U, x) against float64 -- every gradient, with the
    alias's gradient added in the same pass and the upstream gradient of xe a column block of a wider matrix."""
    ops = _ops()
    g = torch.Generator().manual_seed(R + C + J)
    x = torch.randn(R, C, generator=g) * 2 + 0.5
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    U, cb = torch.randn(J, C, generator=g), torch.randn(J, generator=g)
    wx, wl, wa = torch.randn(R, 2 * C, generator=g), torch.randn(R, J, generator=g), torch.randn(R, C, generator=g)
    t = [a.double().requires_grad_(True) for a in (x, gamma, beta, U, cb)]
    xe64 = torch.nn.functional.layer_norm(t[0], (C,), t[1], t[2], 1e-5)
    le64 = xe64 @ t[3].T + t[4]
    loss = (torch.cat((xe64, xe64), 1) * wx.double()).sum() + (le64 * wl.double()).sum()
    if alias:
        loss = loss + (t[0] * wa.double()).sum()
    loss.backward()
    d = [a.to(DEV).requires_grad_(True) for a in (x, gamma, beta, U, cb)]
    xe, le, xa = ops.ln_rowdot(d[0], d[1], d[2], d[3], d[4], 1e-5)
    loss = (torch.cat((xe, xe), 1) * wx.to(DEV)).sum() + (le * wl.to(DEV)).sum()
    if alias:
        loss = loss + (xa * wa.to(DEV)).sum()
    loss.backward()
    np.testing.assert_allclose(xe.detach().cpu().numpy(), xe64.detach().numpy(), atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(le.detach().cpu().numpy(), le64.detach().numpy(), atol=3e-4, rtol=1e-5)
    for name, a, r in zip(("dx", "dgamma", "dbeta", "dU", "dcb"), d, t):
        assert _rel(a.grad, r.grad) < 3e-5, (name, _rel(a.grad, r.grad))
    # the LayerNorm itself is the row kernels' (bitwise)
    assert torch.equal(xe.detach(), ops.layer_norm_rows(d[0].detach(), d[1].detach(), d[2].detach(), 1e-5))


@pytest.mark.parametrize("shape,with_res", [((4,), False), ((777, 256), True), ((15744, 256), True), ((300, 64), False)])
def test_dropout_add_is_a_consistent_dropout(shape, with_res):
    """faf_dropout_add: out = res + x * keep with keep in {0, 1 / (1 - p)} from the hash of (seed, element); the backward
    applies the SAME keep pattern to dout and passes dout through to res; p = 0 is exactly res + x."""
    ops = _ops()
    g = torch.Generator().manual_seed(sum(shape))
    x = (torch.rand(*shape, generator=g) * 2 + 1).to(DEV)         # (in [1, 3]: the keep pattern is read off the output)
    res = torch.randn(*shape, generator=g).to(DEV) if with_res else None
    w = torch.randn(*shape, generator=g).to(DEV)
    seed = torch.tensor([424242], dtype=torch.int64, device=DEV)
    p = 0.2
    xa = x.clone().requires_grad_(True)
    ra = res.clone().requires_grad_(True) if with_res else None
    assert ops.dropout_add_supported(xa, ra)
    out = ops.dropout_add(xa, ra, p, seed)
    (out * w).sum().backward()
    keep = ((out.detach() - (res if with_res else 0.0)) / x)
    k0, k1 = keep.abs() < 1e-6, (keep - 1.0 / (1.0 - p)).abs() < 1e-4
    assert bool((k0 | k1).all())
    if x.numel() > 10000:
        assert abs(float(k0.float().mean()) - p) < 0.01
    np.testing.assert_allclose(xa.grad.cpu().numpy(), (w * torch.where(k0, 0.0, 1.0 / (1.0 - p))).cpu().numpy(), rtol=1e-6, atol=1e-6)
    if with_res:
        assert torch.equal(ra.grad, w)
    assert torch.equal(out.detach(), ops.dropout_add(x, res, p, seed))
    assert torch.equal(ops.dropout_add(x, res, 0.0), x + res if with_res else x)
