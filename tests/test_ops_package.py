"""equihgnn_amd.ops is a package of one module per subsystem; callers use the flat ``ops.NAME`` namespace."""
import importlib


def test_switches_set_on_the_package_reach_the_modules_that_read_them():
    """``ops.GEMM_TILE = 64`` (tests, tools/gemm_bench.py), ``ops.TIMELINE = tl`` (bench.py) and
    ``ops.KNN_GRID_MIN_POINTS`` are read inside submodules; the package forwards the assignment, also to a submodule
    that imported the name from its owner."""
    from equihgnn_amd import ops
    mod = {n: importlib.import_module("equihgnn_amd.ops." + n) for n in ("_base", "aggregate", "products", "grads")}
    saved = (ops.GEMM_TILE, ops.TIMELINE, ops.KNN_GRID_MIN_POINTS, ops.X6_DEEP_ROWS)
    try:
        ops.GEMM_TILE, ops.TIMELINE, ops.KNN_GRID_MIN_POINTS, ops.X6_DEEP_ROWS = 64, "tl", 7, 123
        assert mod["products"].GEMM_TILE == 64 and mod["_base"].TIMELINE == "tl"
        assert mod["aggregate"].KNN_GRID_MIN_POINTS == 7
        assert mod["products"].X6_DEEP_ROWS == 123 and mod["grads"].X6_DEEP_ROWS == 123     # owner and importer
        assert ops.GEMM_TILE == 64
    finally:
        ops.GEMM_TILE, ops.TIMELINE, ops.KNN_GRID_MIN_POINTS, ops.X6_DEEP_ROWS = saved
    assert mod["products"].GEMM_TILE == saved[0] and mod["_base"].TIMELINE is saved[1]


def test_flat_namespace_exports_every_operator():
    from equihgnn_amd import ops
    for name in ("linear", "linear2", "gemm", "gemm_batch", "mm_nt", "scatter", "csr_build", "knn", "readout_mse",
                 "egnn_edge", "rowgemm2", "frame_hidden", "defer_begin", "defer_flush", "merged_weights", "CSR",
                 "Timeline", "incidence_ln_reduce", "gather_ln_reduce", "small_mm_batch", "_ptr", "_stream"):
        assert hasattr(ops, name), name
    assert callable(ops.linear) and callable(ops.gemm) and callable(ops.scatter)     # functions, not the submodules
