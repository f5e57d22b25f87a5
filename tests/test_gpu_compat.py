"""north_star: "the PCLayer/PCTrainer constructor and callback API surface is preserved so figure_*.py scripts run unchanged".
compat/ holds alias packages named like the reference's (`predictive_coding`, `utils`); a script shaped like the reference's own
(tests/compat_script/linear_gaussian_posterior.py: it mentions nothing of this repository) is run UNCHANGED with compat/ first on
the module path and must sample the analytic posterior of the figure-2 toy on the engine's fused path."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_alias_packages_resolve_to_the_engine_mirror():
    # (CPU: importing is enough)
    code = ("import predictive_coding as pc, utils.model as um, utils.training_evaluation as te, predictive_coding.pc_trainer as pt;"
            "import montecarlopredictivecoding_amd.predictive_coding as mine;"
            "assert pc.PCTrainer is mine.PCTrainer and pc.PCLayer is mine.PCLayer and pt.PCTrainer is mine.PCTrainer;"
            "assert um.random_step._mcpc['langevin'] and callable(te.get_mcpc_trainer); print('ok')")
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "compat"))
    run = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd="/tmp", timeout=300)
    assert run.returncode == 0 and run.stdout.strip() == "ok", run.stdout + run.stderr


@pytest.mark.gpu
def test_reference_shaped_script_runs_unchanged_on_the_engine():
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "compat"))
    script = os.path.join(ROOT, "tests", "compat_script", "linear_gaussian_posterior.py")
    text = open(script).read()
    assert "montecarlopredictivecoding_amd" not in text.replace('"""', "").split("import json")[1]      # the script names nothing of this repo
    run = subprocess.run([sys.executable, script], capture_output=True, text=True, env=env, cwd="/tmp", timeout=600)
    assert run.returncode == 0, run.stdout + run.stderr
    out = json.loads([ln for ln in run.stdout.splitlines() if ln.startswith("{")][-1])
    # (round 5: compat/sitecustomize.py installs the import finder, so `predictive_coding` IS the engine's package -- no alias package
    # in between any more; with `python -S` the alias package of compat/ would be what is imported)
    assert os.path.join("montecarlopredictivecoding_amd", "predictive_coding") in out["pc_module"] or \
        os.path.join("compat", "predictive_coding") in out["pc_module"]
    assert out["mode"] == "fused"                                   # the MCPC call ran inside mcpc_run, random_step as Philox noise
    assert abs(out["map"] - 0.44) < 2e-3                            # MAP of the posterior (Adam on x)
    # posterior N(0.44, 0.2) (+ O(lr) SGLD discretisation bias on the variance): 256 chains x 1800 steps
    assert abs(out["mean"] - 0.44) < 0.02 and abs(out["var"] - 0.2) < 0.02, out
