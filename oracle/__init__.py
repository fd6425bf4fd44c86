"""ORACLE package — test infrastructure only (see ref_models.py).  Never imported by equihgnn_amd."""
from . import ref_equiformer, ref_faformer, ref_models

ref_models.MODELS["equiformer_equihnns"] = ref_equiformer.EquiformerEquiHNNS
ref_models.MODELS["faformer_equihnns"] = ref_faformer.FAFormerEquiHNNS
MODELS = ref_models.MODELS
