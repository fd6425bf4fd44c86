"""GPU parity of the drop-in model classes: against the golden vectors captured from the
reference itself, and against the oracle at the BASELINE batch sizes (forward outputs within
1e-5 fp32, the north_star tolerance)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from common import batch_from_case, fill_state_dict, golden_args, load_case  # noqa: E402
from test_oracle_golden import CASES, check_against_case  # noqa: E402

from oracle import ref_models as O  # noqa: E402

DEV = "cuda:0"
TOL = 1e-5


def _models():
    from equihgnn_amd import models
    return models.MODELS


@pytest.mark.parametrize("name", CASES)
def test_hip_model_matches_reference_golden(name):
    case = load_case(name)
    method = str(case["meta_method"])
    model = _models()[method](1, golden_args(method, int(case["meta_hidden"])))
    fill_state_dict(model, int(case["meta_seed"]))
    model.train(bool(int(case["meta_train"])))
    model.to(DEV)
    data = batch_from_case(case).to(DEV)
    check_against_case(model, case, data)


@pytest.mark.parametrize("method,bs,seed", [("mhnnm", 32, 1000), ("egnn_equihnns", 32, 2001),
                                            ("egnn_equihnns", 256, 2000)])
def test_hip_model_matches_oracle_at_baseline_sizes(method, bs, seed):
    from equihgnn_amd.batch import synth_batch
    from equihgnn_amd.registry import default_args
    torch.manual_seed(0)
    args = default_args(method=method)
    ref = O.MODELS[method](1, args)
    fill_state_dict(ref, seed)
    mine = _models()[method](1, args)
    mine.load_state_dict(ref.state_dict(), strict=True)
    mine.to(DEV)
    data = synth_batch(bs, seed)
    out_ref = ref(data)
    loss_ref = torch.nn.functional.mse_loss(out_ref, data.y)
    loss_ref.backward()
    d = data.to(DEV)
    out = mine(d)
    loss = torch.nn.functional.mse_loss(out, d.y)
    loss.backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), out_ref.detach().numpy(), atol=TOL * 2, rtol=0)
    gref = dict(ref.named_parameters())
    gmax = max(float(p.grad.abs().max()) for p in gref.values() if p.grad is not None)
    for n, p in mine.named_parameters():
        r = gref[n].grad
        if r is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
            continue
        np.testing.assert_allclose(p.grad.cpu().numpy(), r.numpy(), atol=2e-5 * gmax + 1e-7, rtol=1e-4, err_msg=n)


def test_rigid_motion_invariance():
    """the reference's outputs are invariant to a rotation+translation of pos."""
    from equihgnn_amd.batch import synth_batch
    from equihgnn_amd.registry import default_args
    args = default_args(method="egnn_equihnns", MLP_hidden=64, output_hidden=32)
    m = _models()["egnn_equihnns"](1, args)
    fill_state_dict(m, 5)
    m.to(DEV).eval()
    d = synth_batch(16, 77).to(DEV)
    q, _ = torch.linalg.qr(torch.randn(3, 3, generator=torch.Generator().manual_seed(0)))
    with torch.no_grad():
        a = m(d)
        d2 = synth_batch(16, 77).to(DEV)
        d2.pos = d2.pos @ q.to(DEV) + torch.tensor([1.0, -2.0, 0.5], device=DEV)
        b = m(d2)
    np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), atol=2e-4)
