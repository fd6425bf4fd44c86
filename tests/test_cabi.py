"""CPU-side checks of the drop-in boundary: the shared library loads and exports every symbol
include/equihgnn_hip.h declares, the ctypes table matches the header, argument validation
works without a GPU, and the product refuses to run on CPU tensors (no fallback)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "equihgnn_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b([a-z_0-9]+)\s*\([^;{]*\)\s*;", text)))


def test_header_declares_functions():
    names = declared_functions()
    assert "hg_segment_reduce_f32" in names and "geo_knn" in names and len(names) >= 8


def test_library_exports_every_declared_symbol():
    from equihgnn_amd import build, hip

    build.build(verbose=False)
    handle = ctypes.CDLL(hip.LIB_PATH)
    for name in declared_functions():
        assert hasattr(handle, name), f"{name} declared in the header but not exported"
    assert sorted(hip.SIGNATURES) == declared_functions()
    assert hip.lib().eqh_version() >= 1


def test_argument_validation_without_gpu():
    from equihgnn_amd import hip

    L = hip.lib()
    assert L.hg_csr_build_workspace_bytes(100, 10) > 0
    assert L.hg_csr_build_workspace_bytes(-1, 10) == 0
    # null pointers / bad shapes are rejected before anything is launched
    assert L.hg_segment_reduce_f32(None, None, None, None, None, 4, 64, 0, None) == -1
    assert L.geo_knn(None, 10, 0, 0, None, None, None) == -1
    assert L.geo_knn(None, 10, 16, 5, None, None, None) == -1
    assert b"argument" in L.eqh_error_string(-1)
    with pytest.raises(hip.HipLibraryError):
        hip.check(-2, "demo")


def test_no_cpu_fallback():
    from equihgnn_amd import hip, ops

    with pytest.raises(hip.HipLibraryError):
        ops.csr_build(torch.zeros(4, dtype=torch.int64), None, 2)
    with pytest.raises(hip.HipLibraryError):
        ops.knn(torch.zeros(20, 3), 16, 0)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "equihgnn_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), fn


def test_registry_contract():
    from equihgnn_amd import models  # noqa: F401  (registers)
    from equihgnn_amd.registry import create_model, registry

    assert registry.get_model_class("egnn_equihnns") is models.EGNNEquiHNNS
    assert registry.get_model_class("nope") is None
    with pytest.raises(ValueError):
        create_model("nope")
    with pytest.raises(ValueError):
        registry.register_model("mhnnm")(object)


def test_state_dict_matches_reference_names():
    """The fixture's grad_names are the reference's named_parameters()."""
    from common import golden_args, load_case

    from equihgnn_amd import models

    for name in ("mhnnm_c64_train", "egnn_equihnns_c64", "equiformer_equihnns_c64", "mhnn_c64", "mhnns_c64",
                 "egnn_equihnn_c64", "egnn_equihnnm_c64", "faformer_equihnns_c64"):
        case = load_case(name)
        method = str(case["meta_method"])
        m = models.MODELS[method](1, golden_args(method, int(case["meta_hidden"])))
        assert sorted(n for n, _ in m.named_parameters()) == sorted(str(n) for n in case["grad_names"])


def test_gemm_x6_tile_choice_is_a_valid_configuration_and_sizes_its_workspace():
    """hg_gemm_x6_choose_tile (the cost estimate behind tile = 0) on host only: a valid tile id for any shape, the big
    configurations for the shapes they were built for, and hg_gemm_x6_workspace_bytes(tile = 0) equal to the chosen
    configuration's own need (the launch re-derives the same choice)."""
    import random

    from equihgnn_amd import hip
    L = hip.lib()
    rng = random.Random(7)

    def prob(m, n, k, ta=0, tb=1):
        pr = (hip.HgGemmProblem * 1)()
        q = pr[0]
        q.m, q.n, q.k, q.trans_a, q.trans_b = m, n, k, ta, tb
        return pr

    for _ in range(300):
        m, n, k = rng.choice([1, 77, 256, 4736, 31232, 250000, 1971840]), 4 * rng.randint(1, 4200), 4 * rng.randint(1, 20000)
        ta = rng.randint(0, 1)
        pr = prob(m if not ta else 4 * ((m + 3) // 4), n, k, ta, rng.randint(0, 1))
        tile = L.hg_gemm_x6_choose_tile(1, pr, 1)
        assert tile in (64, 128, 256, 512, 513), (m, n, k, tile)
        assert L.hg_gemm_x6_workspace_bytes(1, pr, 0) == L.hg_gemm_x6_workspace_bytes(1, pr, tile)
        assert L.hg_gemm_x6_choose_tile(1, pr, 0) in (64, 128, 256, 512, 513)
    assert L.hg_gemm_x6_choose_tile(1, prob(245760, 256, 256), 1) == 512          # [246 k x 256].[256 x 256]: 128 x 256 tiles
    assert L.hg_gemm_x6_choose_tile(1, prob(1971840, 128, 256, 0, 0), 1) == 513    # N = 128: 256 x 128
    assert L.hg_gemm_x6_choose_tile(1, prob(77, 132, 36, 0, 0), 1) == 64
    assert L.hg_gemm_x6_workspace_bytes(1, prob(256, 256, 245760, 1, 0), 0) > 0     # a deep weight gradient is split along k
    assert L.hg_gemm_x6_workspace_bytes(1, prob(245760, 256, 256), 0) == 0
