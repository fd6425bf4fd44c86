"""Host-side algebra of the product path that needs no GPU: the weight-level products that replace pairs of Linears
(ops.merged_weight, layers.MHNNSConv._prepare_merged) against the layer-by-layer formulation of conv.py:169-182."""
import torch

from equihgnn_amd import ops


def test_merged_weight_equals_two_linears_and_backpropagates_to_all_four_parameters():
    g = torch.Generator().manual_seed(0)
    C = 12
    A = torch.randn(C, 2 * C, generator=g, dtype=torch.float64, requires_grad=True)    # W2.lins[0].weight [C, 2C]
    B = torch.randn(C, C, generator=g, dtype=torch.float64, requires_grad=True)        # W1.lins[1].weight
    bb = torch.randn(C, generator=g, dtype=torch.float64, requires_grad=True)          # W1.lins[1].bias
    bo = torch.randn(C, generator=g, dtype=torch.float64, requires_grad=True)          # W2.lins[0].bias
    x = torch.randn(7, C, generator=g, dtype=torch.float64)
    w = torch.randn(7, C, generator=g, dtype=torch.float64)
    # y = A[:, C:2C] (B x + bb) + bo, the two Linears applied one after the other
    ref = torch.nn.functional.linear(torch.nn.functional.linear(x, B, bb), A[:, C:], bo)
    (ref * w).sum().backward()
    g_ref = [t.grad.clone() for t in (A, B, bb, bo)]
    for t in (A, B, bb, bo):
        t.grad = None
    wc, bc = ops.merged_weight(A, B, bb, bo, cols=(C, 2 * C))
    out = torch.nn.functional.linear(x, wc, bc)
    torch.testing.assert_close(out, ref, rtol=1e-12, atol=1e-12)
    (out * w).sum().backward()
    for t, r in zip((A, B, bb, bo), g_ref):
        torch.testing.assert_close(t.grad, r, rtol=1e-10, atol=1e-12)
    assert float(A.grad[:, :C].abs().max()) == 0.0          # only the hyperedge half of W2's first Linear is involved


def test_merged_weight_without_biases_and_without_a_column_block():
    g = torch.Generator().manual_seed(1)
    A = torch.randn(5, 6, generator=g, dtype=torch.float64, requires_grad=True)
    B = torch.randn(6, 4, generator=g, dtype=torch.float64, requires_grad=True)
    wc, bc = ops.merged_weight(A, B)
    assert bc is None
    torch.testing.assert_close(wc, A @ B)
    wc.sum().backward()
    torch.testing.assert_close(A.grad, torch.ones(5, 4, dtype=torch.float64) @ B.detach().t())
    torch.testing.assert_close(B.grad, A.detach().t() @ torch.ones(5, 4, dtype=torch.float64))
