"""Alias package: `from utils.model import ...` / `from utils.training_evaluation import ...` resolve to the MI355X engine's
counterparts of /root/reference/utils/model.py and training_evaluation.py; every other submodule (`utils.data`,
`utils.plotting`: out of scope here) is looked up in the other `utils/` directories on sys.path -- the script's own."""
import os
import sys
from pkgutil import extend_path

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.append(_ROOT)

__path__ = extend_path(__path__, __name__)

from montecarlopredictivecoding_amd.utils import model, training_evaluation          # noqa: E402,F401

sys.modules[__name__ + ".model"] = model
sys.modules[__name__ + ".training_evaluation"] = training_evaluation
