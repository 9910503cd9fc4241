"""Drop-in alias: ``import dctn.eps`` etc. resolve to the MI355X-native modules of ``dctn_amd``
(same module paths as the reference package, so its runner / tests import unchanged)."""
import importlib
import sys

_MODULES = (
    "pos2d", "singleton", "align", "utils", "conv_sbs_spec", "contraction_path_cache", "eps",
    "epses_composition", "conv_sbs", "eps_plus_linear", "logmatmulexp", "rank_one_tensor", "evaluation", "training",
)
for _name in _MODULES:
    _mod = importlib.import_module(f"dctn_amd.{_name}")
    sys.modules[f"{__name__}.{_name}"] = _mod
    globals()[_name] = _mod
