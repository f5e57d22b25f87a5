"""Alias package: `import predictive_coding as pc` resolves to the MI355X engine's API mirror (compat/README.md).
Counterpart of /root/reference/predictive_coding/__init__.py:1-2."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.append(_ROOT)          # (behind everything else: only `montecarlopredictivecoding_amd` is wanted from there)

from montecarlopredictivecoding_amd.predictive_coding import PCLayer, PCTrainer                  # noqa: E402,F401
from montecarlopredictivecoding_amd.predictive_coding import pc_layer, pc_trainer, utils         # noqa: E402,F401

# the reference's submodule names (predictive_coding.pc_layer, .pc_trainer, .utils) under this package's name
sys.modules[__name__ + ".pc_layer"] = pc_layer
sys.modules[__name__ + ".pc_trainer"] = pc_trainer
sys.modules[__name__ + ".utils"] = utils

__all__ = ["PCLayer", "PCTrainer"]
