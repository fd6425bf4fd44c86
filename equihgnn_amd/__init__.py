"""equihgnn_amd: the MI355X-native training hot path of HySonLab/EquiHGNN."""
import os


def enable_tuned_gemms(tuning: bool = False) -> bool:
    """Let PyTorch's TunableOp use the GEMM selections committed for gfx950 (``tuned/tunableop_gfx950.csv``): the library's
    default heuristic is erratic for some fp32 shapes of this path (a 256^3 product: 63 us against 5 us tuned, measured
    inside the replayed step).  ``bench.py`` does the same through the environment; call this before the first GEMM of a
    training run.  ``tuning=True`` also tunes shapes the file does not hold (during the first steps).  Returns whether
    the selections were read."""
    import torch
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned", "tunableop_gfx950.csv")
    t = torch.cuda.tunable
    t.enable(True)
    t.tuning_enable(bool(tuning))
    return bool(t.read_file(path)) if os.path.exists(path) else False
