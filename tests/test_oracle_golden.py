"""The oracle (oracle/ref_models.py) against the golden vectors captured from the reference
itself (tests/golden/make_golden.py).  CPU only.  Tolerance: 1e-5 absolute on O(1) outputs
(BASELINE.json north_star), gradients 1e-5 relative to the largest gradient entry."""
import numpy as np
import pytest
import torch

from common import (CASE_TABLE, F64_FULL_LIMIT, F64_MIN_MARGIN, F64_TABLE, assert_close, batch_from_case, case_spec,
                    f64_sample_indices, f64_spec, fill_state_dict, golden_args, load_case, zero_dropouts)

import oracle  # noqa: F401  (registers the Equiformer oracle in ref_models.MODELS)
from oracle import ref_models

TOL = 1e-5

CASES = list(CASE_TABLE)


def golden_neighbour_ids(case):
    """The Equiformer's topk runs over the self-excluded candidate list (equiformer_layer.py:
    1254-1257,1303): position p of row i is node p + (p >= i)."""
    pos = case["knn_idx"]
    return pos + (pos >= np.arange(pos.shape[0])[:, None])


def build(case, models=ref_models.MODELS):
    method = str(case["meta_method"])
    name = str(case.get("meta_name", ""))
    spec = case_spec(name) if name in CASE_TABLE else {"dropout0": False, "args": {}}
    model = models[method](1, golden_args(method, int(case["meta_hidden"]), **spec["args"]))
    fill_state_dict(model, int(case["meta_seed"]))
    model.train(bool(int(case["meta_train"])))
    if spec["dropout0"]:
        zero_dropouts(model)
    return model


def check_against_case(model, case, data, tol=TOL, taps=True, grad_rtol=1e-4):
    tp = {} if taps else None
    out = model(data, taps=tp) if taps else model(data)
    # 1e-5 absolute wherever |out| <= 1 (north_star); entries larger than 1 get the same RELATIVE budget
    if "out_f64" in case:
        # train-mode BatchNorm cases: against the reference's float64 evaluation at 1e-5; against its float32 capture at 1e-5
        # PLUS that capture's own distance from the float64 value, read off the fixture (3e-6 .. 7e-6 with BatchNorm between
        # the layers, 2e-5 with --normalization bn, where every MLP -- the head over a handful of molecules included -- has one)
        assert_close(out.detach().cpu().numpy(), case["out_f64"], tol, "out (reference in float64)")
        ref32, ref64 = case["out"].astype(np.float64), case["out_f64"]
        own = float((np.abs(ref32 - ref64) / np.maximum(1.0, np.abs(ref64))).max())
        assert_close(out.detach().cpu().numpy(), case["out"], tol + own, "out")
    else:
        assert_close(out.detach().cpu().numpy(), case["out"], tol, "out")
    if taps:
        for k, v in tp.items():
            key = "tap_" + k
            if key in case:
                # intermediate tensors (a pooled sum of O(10) rows cancels to O(0.1)): 1e-5 of the tensor's scale
                ref = case[key]
                got = v.detach().cpu().numpy().reshape(ref.shape)
                scale = max(1.0, float(np.abs(ref).max()))
                np.testing.assert_allclose(got, ref, atol=tol * scale, rtol=0, err_msg=k)
    loss = torch.nn.functional.mse_loss(out, data.y)
    np.testing.assert_allclose(float(loss.detach()), float(case["loss"]), atol=tol * max(1.0, float(case["loss"])))
    loss.backward()
    names = [str(n) for n in case["grad_names"]]
    present = case["grad_present"]
    stats = case["grad_stats"]
    params = dict(model.named_parameters())
    assert sorted(params) == sorted(names)
    # gradients that are analytically zero (e.g. a bias in front of a train-mode BatchNorm) are
    # rounding noise on both sides: floor every tolerance at 1e-3 of the largest gradient norm
    floor = 1e-3 * float(np.max(stats[:, 2]))
    for n, has, st in zip(names, present, stats):
        g = params[n].grad
        if not has:
            assert g is None or float(g.abs().max()) == 0.0, f"{n}: reference leaves grad None"
            continue
        assert g is not None, f"{n}: reference has a gradient"
        g = g.detach().cpu()
        if "grad_" + n in case:
            ref = case["grad_" + n]
            scale = max(floor, float(np.abs(ref).max()))
            np.testing.assert_allclose(g.numpy(), ref, atol=grad_rtol * scale, rtol=0, err_msg=n)
        else:
            ref = case["gradhead_" + n]
            scale = max(floor, float(st[2]))
            np.testing.assert_allclose(g.reshape(-1)[: ref.size].numpy(), ref, atol=grad_rtol * scale,
                                       rtol=0, err_msg=n)
            np.testing.assert_allclose(float(g.norm()), st[2], rtol=10 * grad_rtol, atol=grad_rtol * floor,
                                       err_msg=n)


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference(name):
    case = load_case(name)
    model = build(case)
    data = batch_from_case(case)
    check_against_case(model, case, data)
    if "knn_idx" in case and str(case["meta_method"]).startswith("egnn_"):
        d2, idx = ref_models.knn_self_included(data.pos, 16)
        assert np.array_equal(np.sort(idx.numpy(), -1), np.sort(case["knn_idx"], -1))
    if "knn_idx" in case and str(case["meta_method"]) == "equiformer_equihnns":
        from oracle.ref_equiformer import neighbours_self_excluded
        idx, dist, _, _ = neighbours_self_excluded(data.pos, 16, 5.0)
        assert np.array_equal(np.sort(idx.numpy(), -1), np.sort(golden_neighbour_ids(case), -1))
        np.testing.assert_allclose(dist.numpy(), case["knn_val"], rtol=1e-6)


def test_state_dict_names_match_reference():
    """grad_names in the fixture are the reference's named_parameters()."""
    for name in ("mhnnm_c64_train", "egnn_equihnns_c64"):
        case = load_case(name)
        model = build(case)
        assert [n for n, _ in model.named_parameters()] == [str(n) for n in case["grad_names"]]


def test_bn_running_stats_update():
    case = load_case("mhnnm_c64_train")
    model = build(case)
    model(batch_from_case(case))
    for k, v in model.state_dict().items():
        if "running_" in k:
            np.testing.assert_allclose(v.numpy(), case["buf_" + k], atol=1e-5, rtol=1e-5)


def test_oracle_wigner_d_matches_reference_on_degenerate_directions():
    """equiformer_D.npz: the reference's get_D_to_from_z_axis (equiformer/basis.py:194-215) on crafted rel_pos rows --
    generic, axis-aligned, exactly -y, inside the |x_hat + y_hat|^2 < 1e-6 clamp (:187-190), zero, tiny, huge."""
    from oracle.ref_equiformer import wigner_d1_to_y
    case = load_case("equiformer_D")
    d = wigner_d1_to_y(torch.from_numpy(case["rel_pos"]))
    np.testing.assert_allclose(d.numpy(), case["D1"], atol=2e-6, rtol=0)
    # the column the live path reads (m = 0) is NOT r_hat inside the clamp: the fixture must contain such rows
    rel = case["rel_pos"].astype(np.float64)
    nrm = np.linalg.norm(rel, axis=-1, keepdims=True)
    rhat = np.divide(rel, nrm, out=np.zeros_like(rel), where=nrm > 0)
    dev = np.abs(case["D1"][:, :, 1] - rhat).max(-1)
    assert (dev > 1e-4).sum() >= 5 and (dev < 1e-6).sum() >= 20


# ---- gradients pinned to the reference itself in float64 (tests/golden/*_f64.npz) ---------------------------------------------
F64_CASES = list(F64_TABLE)


def build_f64(case, models=ref_models.MODELS):
    """The model of an *_f64 fixture with its float32 parameter values (what the float64 reference was built from)."""
    name = str(case["meta_name"])
    spec = f64_spec(name)
    model = models[spec["method"]](1, golden_args(spec["method"], spec["hidden"]))
    fill_state_dict(model, spec["seed"])
    model.train(spec["train"])
    if spec["dropout0"]:
        zero_dropouts(model)
    return model


def check_grads_against_f64(params, case, tol):
    """Every stored gradient entry (whole tensors up to F64_FULL_LIMIT entries, the evenly spread sample above) within
    `tol` of the LARGEST gradient entry of the model; returns the worst ratio."""
    names = [str(n) for n in case["grad_names"]]
    assert sorted(params) == sorted(names)
    gmax = float(case["grad_absmax"].max())
    worst = (0.0, "")
    for n, has in zip(names, case["grad_present"]):
        g = params[n].grad
        if not has:
            assert g is None or float(g.abs().max()) == 0.0, f"{n}: reference leaves grad None"
            continue
        assert g is not None, f"{n}: reference has a gradient"
        g = g.detach().cpu().double().numpy()
        if "g64_" + n in case:
            ref = case["g64_" + n].astype(np.float64)
            assert g.size <= F64_FULL_LIMIT
        else:
            ref = case["g64s_" + n].astype(np.float64)
            g = g.reshape(-1)[f64_sample_indices(g.size)]
        err = float(np.abs(g - ref).max()) / gmax
        worst = max(worst, (err, n))
        assert err <= tol, f"{n}: |grad - reference float64| = {err:.2e} of the largest gradient entry (> {tol:g})"
    return worst


@pytest.mark.parametrize("name", F64_CASES)
def test_oracle_in_float64_matches_the_reference_in_float64(name):
    """The CPU restatement evaluated in float64 against the REFERENCE's own model evaluated in float64 (fixtures generated
    by make_golden.run_case_f64 from /root/reference): no rounding on either side, so the two agree to 1e-9 -- the restated
    algorithm IS the reference's, gradients included, at hidden 256 and for FAFormer where float32 fixtures only pin 1e-2."""
    case = load_case(name)
    assert float(case["relu_margin"]) >= F64_MIN_MARGIN
    model = build_f64(case).double()
    data = batch_from_case(case)
    data.pos, data.y = data.pos.double(), data.y.double()
    out = model(data)
    np.testing.assert_allclose(out.detach().numpy(), case["out64"], rtol=1e-9, atol=1e-9)
    loss = torch.nn.functional.mse_loss(out, data.y)
    np.testing.assert_allclose(float(loss.detach()), float(case["loss64"]), rtol=1e-9)
    loss.backward()
    # (stored as float32: 6e-8 relative of each entry)
    check_grads_against_f64(dict(model.named_parameters()), case, 2e-7)
