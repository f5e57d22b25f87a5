"""The alias packages of compat/ resolve the reference's import lines to this package (CPU: imports only; the script-shaped run on the
GPU is tests/test_gpu_compat.py).  Reference scripts start with `import predictive_coding as pc`, `from utils.model import *`,
`from utils.training_evaluation import *` (/root/reference/figure_2.py:1-22); a script's own `utils/` directory (e.g. utils/plotting.py)
must stay importable beside the aliases."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_import_lines_resolve_to_this_package(tmp_path):
    (tmp_path / "utils").mkdir()
    (tmp_path / "utils" / "plotting.py").write_text("MARK = 'the script\\'s own utils/plotting.py'\n")
    code = (
        "import predictive_coding as pc\n"
        "from utils.model import *\n"
        "from utils.training_evaluation import *\n"
        "import utils.plotting as up\n"
        "assert pc.PCTrainer.__module__ == 'montecarlopredictivecoding_amd.predictive_coding.pc_trainer', pc.PCTrainer.__module__\n"
        "assert pc.PCLayer.__module__ == 'montecarlopredictivecoding_amd.predictive_coding.pc_layer'\n"
        "assert get_model.__module__ == 'montecarlopredictivecoding_amd.utils.model'\n"
        "assert get_mcpc_trainer.__module__ == 'montecarlopredictivecoding_amd.utils.training_evaluation'\n"
        "assert random_step.__module__ == 'montecarlopredictivecoding_amd.utils.model'\n"
        "assert up.MARK.startswith('the script')\n"
        "print('ok')\n")
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "compat"))
    out = subprocess.run([sys.executable, "-c", code], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]
