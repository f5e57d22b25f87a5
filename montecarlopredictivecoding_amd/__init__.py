"""montecarlopredictivecoding_amd -- MI355X-native Monte Carlo Predictive Coding inference engine.

One hot path: the Langevin / PC inference loop of
``PCTrainer.train_on_batch`` + ``random_step``, as hand-written HIP for gfx950 behind a C ABI
(``include/mcpc.h`` -> ``libmcpc.so``), with a host-side mirror of the reference's
``predictive_coding`` API in :mod:`montecarlopredictivecoding_amd.predictive_coding`.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
__version__ = "0.1.0"
