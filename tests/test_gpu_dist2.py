"""World size 2 on REAL engines through the facade (the closest rehearsal of BASELINE's cfg-8GPU one GPU allows).

Two fresh processes (started with subprocess: a child program, never an exec of this GPU-initialised process), both on cuda:0,
one gloo group; each owns an uneven shard of 6000 chains of a cfg-M-width net and runs two learning calls through
PCTrainer.train_on_batch with `set_shard(process_group, chain_base=lo, world_batch=None)`:
    Hebbian sums of the local shard (HIP) -> flat bucket scaled by 1/(n_acc * B_global) -> one sum-all-reduce -> optimizer_p.step().
The reference divides by len(inputs) = the WHOLE batch (pc_trainer.py:905-909), so B_global must be the group's sum.
  * weights after each call: bitwise equal on both ranks, and equal (1e-4 of the largest gradient entry: summation order) to the unsharded 6000-chain run;
  * trajectories: every rank's final state is BITWISE the unsharded run's rows [lo, hi) (global chain ids feed the Philox counter);
  * per-step loss / energy / overall (`reduce_results=True`: one all-reduce of the [T, L + 2] fp64 table per call): both
    ranks report the whole batch's values, equal to the unsharded run's;
  * `world_batch=None`: the local batch rides in the gradient bucket's last float (no second collective, no host sync).
"""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(900)
def test_two_ranks_on_one_gpu_through_the_facade(tmp_path):
    import montecarlopredictivecoding_amd.predictive_coding.pc_trainer as pt
    from tests import dist_facade_case as case
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "tests", "dist_facade_case.py"), str(r), "2", str(port), str(tmp_path)],
                              env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    # the unsharded run, here, while the ranks start up; same Philox step counter as a fresh process
    keep = pt._PHILOX_STEPS[0]
    pt._PHILOX_STEPS[0] = 0
    try:
        full = case.run_case(0, 1)
    finally:
        pt._PHILOX_STEPS[0] = max(keep, pt._PHILOX_STEPS[0])
    logs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        logs.append(out)
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-3000:] for l in logs)
    r = [torch.load(os.path.join(tmp_path, f"rank{k}.pt")) for k in range(2)]
    assert [(x["lo"], x["hi"]) for x in r] == [(0, 3200), (3200, 6000)]
    assert full["modes"] == ["fused", "fused"] and all(x["modes"] == ["fused", "fused"] for x in r)
    for call in range(2):
        # both ranks hold the same weights, bit for bit (same reduced bucket, same optimizer step)
        for a, c in zip(r[0][f"weights{call}"], r[1][f"weights{call}"]):
            assert torch.equal(a, c)
        for a, c in zip(r[0][f"grads{call}"], r[1][f"grads{call}"]):
            assert torch.equal(a, c)
        # ... and the unsharded run's up to the summation order of the bucket (6000 chains in one sum or in 3200 + 2800)
        for j, (a, c) in enumerate(zip(r[0][f"grads{call}"], full[f"grads{call}"])):
            scale = float(c.abs().max())
            if j == 0:
                assert not bool(a.any()) and not bool(c.any())            # dF/dW0 == 0 exactly
            else:
                assert scale > 0
                torch.testing.assert_close(a, c, rtol=0, atol=1e-4 * scale)   # fp32 sums of 528 000 terms per element in another order
        for a, c in zip(r[0][f"weights{call}"], full[f"weights{call}"]):
            torch.testing.assert_close(a, c, rtol=0, atol=(1 + call) * 1e-4 * 0.5 * max(float(g.abs().max()) for g in full[f"grads{call}"]) + 1e-7)
        # trajectories do not depend on the sharding.  Call 0: bitwise.  Call 1 runs on weights that differ in the last bits
        # between the sharded and the unsharded job (bucket summation order), so it is compared in distribution: all but a few elements to 5e-3.
        for l in range(3):
            cat = torch.cat([r[0][f"xs{call}"][l], r[1][f"xs{call}"][l]])
            if call == 0:
                assert torch.equal(cat, full[f"xs{call}"][l])
            else:
                # (a chain that crosses a ReLU kink a step earlier or later amplifies that: all but a handful of elements agree)
                d = (cat - full[f"xs{call}"][l]).abs()
                assert float((d > 5e-3).float().mean()) < 1e-3 and float(d.max()) < 1.0
        # set_shard(reduce_results=True): every rank reports the WHOLE batch's loss / energy / overall, as the reference does
        # (pc_layer.py:295, pc_trainer.py:785-797) -- both ranks the same numbers, equal to the unsharded run's
        for key in ("loss", "energy", "overall"):
            assert torch.equal(r[0][key][call], r[1][key][call]), key
            torch.testing.assert_close(r[0][key][call], full[key][call], rtol=2e-6 if call == 0 else 1e-3, atol=0)
    # the second call really started from other weights than the first
    assert not torch.equal(full["weights0"][2], full["weights1"][2])
