"""Build provenance (next #5).

* the library says what it is (`mcpc_build_info`, ABI 4) and the shipped one is a clean build of the sources beside it;
* every timing-experiment switch the kernel sources mention is known to the ONE umbrella (csrc/mcpc_build.h), a build with one reports
  exp=1 and the binding refuses to load it;
* the compiler-fragile property of the step kernel -- KParams stays in the kernel-argument segment: no scratch traffic, no spilled
  VGPRs (csrc/Makefile: -instcombine-max-copied-from-constant-users) -- is asserted from `make asm`, not left to a comment."""
import glob
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "montecarlopredictivecoding_amd", "csrc")


def test_shipped_library_is_a_clean_build_of_the_tree():
    from montecarlopredictivecoding_amd import _lib
    info = _lib.build_info()
    assert info["abi"] == str(_lib.ABI_VERSION) and info["arch"] == "gfx950"
    assert info["exp"] == "0" and info["stamps"] == "0", info
    assert "MCPC_EXP" not in info["flags"] and "MCPC_HEB_EXP" not in info["flags"]
    assert "-instcombine-max-copied-from-constant-users=4000" in info["flags"] and "-fno-slp-vectorize" in info["flags"]
    assert info["csrc"] == _lib.csrc_sha(), "libmcpc.so was not built from the sources in this tree: run `make -C montecarlopredictivecoding_amd/csrc`"
    assert os.path.basename(info["path"]) == "libmcpc.so"


def test_every_experiment_switch_is_under_the_umbrella():
    umbrella = open(os.path.join(CSRC, "mcpc_build.h")).read()
    listed = set(re.findall(r"defined\((MCPC_[A-Z0-9_]+)\)", umbrella))
    used = set()
    for path in glob.glob(os.path.join(CSRC, "*")):
        if os.path.basename(path) in ("mcpc_build.h", "Makefile"):
            continue
        for line in open(path, errors="replace"):
            if re.match(r"\s*#\s*(if|ifdef|ifndef|elif)\b", line):
                used |= set(re.findall(r"\b(MCPC_EXP_[A-Z0-9_]+|MCPC_HEB_EXP)\b", line))
    assert used, "the scan found no switch at all: the pattern is broken"
    assert used <= listed, f"switches unknown to csrc/mcpc_build.h: {sorted(used - listed)}"


def test_experiment_build_reports_itself_and_is_refused(tmp_path):
    out = os.path.join(ROOT, "scripts", "bin", "libmcpc_provenance_probe.so")
    try:
        subprocess.run(["make", "-C", CSRC, "variant", "VARNAME=provenance_probe", "VARFLAGS=-DMCPC_EXP_NOEPI"], check=True,
                       capture_output=True, timeout=600)
        code = ("from montecarlopredictivecoding_amd import _lib\n"
                "try:\n    _lib.load(); print('LOADED')\n"
                "except _lib.MCPCLibraryError as e:\n    print('REFUSED', 'timing-experiment build' in str(e))\n")
        env = dict(os.environ, MCPC_LIB=out, PYTHONPATH=ROOT)
        env.pop("MCPC_ALLOW_EXP", None)
        run = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
        assert run.stdout.strip() == "REFUSED True", run.stdout + run.stderr
        env["MCPC_ALLOW_EXP"] = "1"
        run = subprocess.run([sys.executable, "-c", code + "print(_lib.build_info()['exp'], 'MCPC_EXP_NOEPI' in _lib.build_info()['flags'])\n"],
                             capture_output=True, text=True, env=env, timeout=300)
        assert run.stdout.split() == ["LOADED", "1", "True"], run.stdout + run.stderr
    finally:
        if os.path.exists(out):
            os.remove(out)


def test_step_kernel_keeps_its_arguments_out_of_scratch(tmp_path):
    asm = str(tmp_path / "mcpc_gfx950.s")
    run = subprocess.run(["make", "-C", CSRC, "asm", f"ASM_OUT={asm}"], capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-3000:]
    # -Rpass-analysis=kernel-resource-usage: one block of remarks per kernel
    usage, name = {}, None
    for line in run.stderr.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            usage[name] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/lane\])?: (\d+)", line)
        if m and name:
            usage[name][m.group(1).strip()] = int(m.group(2))
    step = {k: v for k, v in usage.items() if "mcpc_steps_ws2_kernelILi1E" in k}
    assert len(step) == 2, sorted(usage)
    text = open(asm).read()
    for k, u in step.items():
        assert u["VGPRs Spill"] == 0, (k, u)
        assert u["ScratchSize"] <= 128, (k, u)                   # (KParams in scratch: 1112 bytes per lane on top)
        assert u["VGPRs"] <= 256 and u["Occupancy"] >= 2 if "Occupancy" in u else True
        body = text[text.index(f"\n{k}:"):]
        body = body[:body.index("s_endpgm")]
        n_scratch = len(re.findall(r"^\s*scratch_(load|store)", body, flags=re.M))
        assert n_scratch == 0, f"{k}: {n_scratch} scratch instructions in the step kernel"
    # the unified-wave kernel (round 6): the same demands
    ustep = {k: v for k, v in usage.items() if "mcpc_steps_u_kernel" in k}
    assert len(ustep) == 2, sorted(usage)
    for k, u in ustep.items():
        assert u["VGPRs Spill"] == 0 and u["ScratchSize"] <= 128 and u["VGPRs"] <= 256, (k, u)
        body = text[text.index(f"\n{k}:"):]
        body = body[:body.index("s_endpgm")]
        assert not re.findall(r"^\s*scratch_(load|store)", body, flags=re.M), f"{k}: scratch instructions in the unified-wave kernel"
    # the Hebbian flush: no scratch traffic in any mcpc_heb7_kernel<*>, with ONE known exception that is not in a loop --
    # <17, 2, false> (136 accumulators + the double-buffered split at 256 VGPRs) parks one address register in scratch before its stage
    # loop and reloads it once behind it (`scratch_store_dword` in the prologue, `scratch_load_dword` after the loop's exit label)
    heb = {k: v for k, v in usage.items() if "mcpc_heb7_kernel" in k}
    assert len(heb) >= 6, sorted(usage)
    for k, u in heb.items():
        body = text[text.index(f"\n{k}:"):]
        body = body[:body.index("s_endpgm")]
        lines = body.splitlines()
        scratch = [i for i, ln in enumerate(lines) if re.match(r"\s*scratch_(load|store)", ln)]
        if "ILi17ELi2ELb0E" in k:
            assert u["VGPRs Spill"] <= 1 and len(scratch) <= 2, (k, u, len(scratch))
            # neither of the two sits inside a loop: no backward branch jumps over it
            labels = {m.group(1): i for i, ln in enumerate(lines) for m in [re.match(r"(\.LBB\d+_\d+):", ln)] if m}
            for i, ln in enumerate(lines):
                m = re.match(r"\s*s_cbranch_\w+\s+(\.LBB\d+_\d+)", ln) or re.match(r"\s*s_branch\s+(\.LBB\d+_\d+)", ln)
                if m and m.group(1) in labels and labels[m.group(1)] < i:          # a backward branch: the loop [target, branch]
                    assert not any(labels[m.group(1)] <= sidx <= i for sidx in scratch), (k, ln.strip())
        else:
            assert u["VGPRs Spill"] == 0 and not scratch, (k, u, len(scratch))
