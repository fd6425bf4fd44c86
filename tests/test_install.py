"""registry.install_into_reference(): the drop-in really swaps this package's classes into the REFERENCE's registry,
so that the reference's own ``create_model`` (equihgnn/utils/create.py:5-10) -- the call LitModel makes at
main.py:28-34 -- hands back the MI355X classes under the same ``--method`` names.

Needs the reference checkout (present in the build container only; the GPU box has none) and imports it with the golden
generator's recipe (tests/golden/make_golden.py::import_reference: stand-ins for the absent third-party packages), in a
child process so that those stand-ins never enter the test session's ``sys.modules``."""
import os
import subprocess
import sys
import textwrap

import pytest

from common import GOLDEN_DIR

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent('''
    import importlib, os, sys, types
    ROOT, GOLD, REF = sys.argv[1:4]
    sys.path[:0] = [ROOT, GOLD]
    import make_golden as mg
    ref_registry = mg.import_reference(("equihnn_egnn", "mhnn"))
    # the reference's create_model, loaded as equihgnn.utils.create without equihgnn/utils/__init__.py (whose
    # data_split import pulls in the dataset stack)
    pkg = types.ModuleType("equihgnn.utils")
    pkg.__path__ = [os.path.join(REF, "equihgnn", "utils")]
    sys.modules["equihgnn.utils"] = pkg
    create = importlib.import_module("equihgnn.utils.create")
    assert os.path.realpath(create.__file__).startswith(os.path.realpath(REF))

    stock = {n: create.create_model(n) for n in ("egnn_equihnns", "mhnnm", "mhnn", "mhnns", "egnn_equihnn", "egnn_equihnnm")}
    assert all(c.__module__.startswith("equihgnn.models") for c in stock.values())

    from equihgnn_amd import models as M
    from equihgnn_amd.registry import default_args, install_into_reference, registry as mine

    kept = install_into_reference(override=False)          # names the reference already has stay the reference's
    assert not (set(kept) & set(stock)) and create.create_model("egnn_equihnns") is stock["egnn_equihnns"]
    for n in kept:                                          # (names its registry did not hold -- their model files were
        assert create.create_model(n) is M.MODELS[n]        #  not imported here -- are added)

    done = install_into_reference()
    assert sorted(done) == sorted(M.MODELS) == sorted(mine.mapping["model_name_mapping"])
    for n in done:
        assert create.create_model(n) is M.MODELS[n], n
        assert ref_registry.get_model_class(n) is M.MODELS[n], n
    try:
        create.create_model("no_such_method")
    except ValueError as e:
        assert "not found" in str(e)
    else:
        raise AssertionError("unknown names must still raise")

    # main.py:28-34: model_cls = create_model(hparams.method); model_cls(1, hparams) -- same state_dict as the stock class
    for n, klass in stock.items():
        args = default_args(method=n, MLP_hidden=32, output_hidden=16)
        swapped = create.create_model(n)
        assert swapped.__name__ != "GNN_2D"
        a, b = klass(1, args).state_dict(), swapped(1, args).state_dict()
        assert sorted(a) == sorted(b), n          # (load_state_dict matches by name: the registration order is free)
        assert all(a[k].shape == b[k].shape and a[k].dtype == b[k].dtype for k in a), n
    print("INSTALL-OK", len(done))
''')


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "equihgnn")), reason="needs the reference checkout (build container)")
def test_install_into_reference_swaps_the_registry_entries():
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT, GOLDEN_DIR, REF], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "INSTALL-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
