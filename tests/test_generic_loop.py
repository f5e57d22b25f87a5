"""g15_*: the reference's keyword surface outside the kernels on the package's generic torch loop
(predictive_coding/generic_loop.py), against fixtures the IMPORTED REFERENCE produced from the same scenario functions
(oracle/gen_golden_generic.py).  The GPU tests are the product behaviour (model on cuda, RuntimeWarning, last_call_mode); the CPU
test checks the loop's arithmetic in the CPU suite through a test-only switch -- a CPU model raises otherwise (no CPU path)."""
import os
import warnings

import numpy as np
import pytest
import torch

from oracle import gen_golden_generic as gg

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _compare(name, got):
    want = np.load(os.path.join(GOLDEN, f"g15_{name}.npz"))
    assert sorted(want.files) == sorted(got), (sorted(want.files), sorted(got))
    for key in want.files:
        scale = max(1.0, float(np.abs(want[key]).max()))
        np.testing.assert_allclose(got[key], want[key], rtol=3e-5, atol=3e-6 * scale, err_msg=f"g15_{name}: {key}")


class _CpuOk:
    """The package's predictive_coding with the test-only switch set on every trainer it builds."""

    def __init__(self):
        import montecarlopredictivecoding_amd.predictive_coding as pc
        self.PCLayer = pc.PCLayer
        self._Trainer = pc.PCTrainer

    def PCTrainer(self, *a, **kw):
        tr = self._Trainer(*a, **kw)
        tr._test_only_generic_on_cpu = True
        return tr


@pytest.mark.parametrize("name", [n for n in gg.SCENARIOS if n != "energy_coefficient"])
def test_generic_loop_matches_reference_fixture_on_cpu(name):
    got, trainer = gg.SCENARIOS[name](_CpuOk(), "cpu")
    assert trainer.last_call_mode == "generic"
    _compare(name, got)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(gg.SCENARIOS))
def test_keyword_surface_on_the_gpu(name):
    import montecarlopredictivecoding_amd.predictive_coding as pc
    got, trainer = gg.SCENARIOS[name](pc, "cuda:0")
    # energy_coefficient is a kernel path (every layer's coefficient scaled); the others leave the engine for the generic loop
    assert trainer.last_call_mode == ("fused" if name == "energy_coefficient" else "generic")
    _compare(name, got)


@pytest.mark.gpu
def test_generic_loop_announces_itself_on_the_models_device():
    import montecarlopredictivecoding_amd.predictive_coding as pc
    import torch.nn as nn
    m = nn.Sequential(nn.Linear(3, 3), pc.PCLayer(M=torch.ones(3, device="cuda:0")), nn.Linear(3, 2)).to("cuda:0")
    m.train()
    tr = pc.PCTrainer(m, T=3, update_p_at="never", plot_progress_at=[])
    with pytest.warns(RuntimeWarning, match="generic torch loop.*Reason: PCLayer 0 uses S/M masks"):
        res = tr.train_on_batch(inputs=torch.zeros(2, 3, device="cuda:0"), is_log_progress=False, is_return_results_every_t=False)
    assert tr.last_call_mode == "generic" and len(res["overall"]) == 1
    # a CPU-built model (round 5): the keyword surface outside the kernels runs on the generic loop where the model lives, announced the
    # same way ("must work, need not be fast"); what the kernels DO express is staged onto the GPU (tests/test_gpu_staging.py)
    m_cpu = nn.Sequential(nn.Linear(3, 3), pc.PCLayer(M=torch.ones(3)), nn.Linear(3, 2))
    m_cpu.train()
    tr_cpu = pc.PCTrainer(m_cpu, T=3, update_p_at="never", plot_progress_at=[])
    with pytest.warns(RuntimeWarning, match="generic torch loop.*Reason: PCLayer 0 uses S/M masks"):
        res = tr_cpu.train_on_batch(inputs=torch.zeros(2, 3), is_log_progress=False, is_return_results_every_t=False)
    assert tr_cpu.last_call_mode == "generic" and len(res["overall"]) == 1 and m_cpu[1].get_x().device.type == "cpu"
