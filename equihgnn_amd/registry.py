"""Model registry with the reference's plugin contract.

Mirrors equihgnn/common/registry.py:1-41 (``register_model`` raises ``ValueError`` on
a duplicate name, ``get_model_class`` returns the class or ``None``) and
equihgnn/utils/create.py:5-10 (``create_model`` raises ``ValueError`` on an unknown
name).  main.py:28-34 instantiates ``model_cls(1, hparams)`` and calls
``model(data) -> Tensor[B]``; every class registered here keeps that signature.

``install_into_reference()`` additionally registers the classes under the same names
in the reference's own registry when ``equihgnn`` is importable, which is how the
drop-in replaces the stock models under main.py.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Callable, Dict, Optional


class Registry:
    mapping: Dict[str, Dict[str, type]] = {"model_name_mapping": {}}

    @classmethod
    def register_model(cls, name: str) -> Callable[[type], type]:
        def wrap(model_cls: type) -> type:
            if name in cls.mapping["model_name_mapping"]:
                raise ValueError(f"Class with name {name} already registered.")
            cls.mapping["model_name_mapping"][name] = model_cls
            return model_cls

        return wrap

    @classmethod
    def get_model_class(cls, name: str) -> Optional[type]:
        return cls.mapping["model_name_mapping"].get(name, None)

    @classmethod
    def list_out(cls):
        return cls.mapping


registry = Registry()


def create_model(model_name: str) -> type:
    model_cls = registry.get_model_class(model_name)
    if model_cls is None:
        raise ValueError(f"Model with name {model_name} not found.")
    return model_cls


def default_args(**overrides) -> SimpleNamespace:
    """Hyper-parameters of scripts/run_qm9_3d.sh:10-31 (the config every BASELINE.json
    line is quoted on), as the ``args`` namespace the model constructors read
    (equihnn_egnn.py:113-149)."""
    a = dict(
        method="egnn_equihnns",
        All_num_layers=3,
        MLP1_num_layers=2,
        MLP2_num_layers=2,
        MLP3_num_layers=2,
        MLP4_num_layers=2,
        output_num_layers=3,
        MLP_hidden=256,
        output_hidden=128,
        aggregate="mean",
        normalization="ln",
        activation="relu",
        dropout=0.0,
        lr=1e-4,
        wd=0.0,
        batch_size=256,
    )
    a.update(overrides)
    return SimpleNamespace(**a)


def install_into_reference(override: bool = True) -> list:
    """Register this package's classes in the reference's registry (drop-in)."""
    import importlib

    ref = importlib.import_module("equihgnn.common.registry").registry
    done = []
    for name, klass in registry.mapping["model_name_mapping"].items():
        table = ref.mapping["model_name_mapping"]
        if name in table and not override:
            continue
        table[name] = klass
        done.append(name)
    return done
