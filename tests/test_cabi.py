"""The C-ABI library loads, exports every symbol include/mcpc.h declares, and the ctypes structs match
the C layout (checked by compiling a probe with gcc against the header).  No compute calls: no GPU here."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mcpc.h")


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mcpc_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from montecarlopredictivecoding_amd import _lib
    lib = _lib.load()
    names = header_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"libmcpc.so does not export {n}"
    assert sorted(_lib.SYMBOLS) == names, "ctypes binding and header disagree on the set of entry points"
    assert lib.mcpc_abi_version() == _lib.ABI_VERSION


def test_error_reporting_without_gpu_calls():
    from montecarlopredictivecoding_amd import _lib
    lib = _lib.load()
    d = _lib.NetDesc()
    d.abi_version = 999
    h = ctypes.c_void_p()
    rc = lib.mcpc_create(ctypes.byref(d), ctypes.byref(h))
    assert rc == -1 and b"ABI version" in lib.mcpc_last_error()
    d.abi_version = _lib.ABI_VERSION
    d.n_latent = 0
    assert lib.mcpc_create(ctypes.byref(d), ctypes.byref(h)) == -1 and b"n_latent" in lib.mcpc_last_error()
    assert lib.mcpc_destroy(None) == 0
    assert lib.mcpc_param_count(None) == 0
    with pytest.raises(_lib.MCPCError):
        _lib.check(lib.mcpc_run(None, None, None))


def test_ctypes_structs_match_c_layout(tmp_path):
    from montecarlopredictivecoding_amd import _lib
    probe = tmp_path / "probe.c"
    fields_net = [f[0] for f in _lib.NetDesc._fields_]
    fields_run = [f[0] for f in _lib.RunDesc._fields_]
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void){",
             'printf("%zu %zu\\n", sizeof(mcpc_net_desc), sizeof(mcpc_run_desc));']
    for f in fields_net:
        lines.append(f'printf("net %s %zu\\n", "{f}", offsetof(mcpc_net_desc, {f}));')
    for f in fields_run:
        lines.append(f'printf("run %s %zu\\n", "{f}", offsetof(mcpc_run_desc, {f}));')
    lines.append("return 0;}")
    probe.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.run(["gcc", "-std=c11", "-o", str(exe), str(probe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    s_net, s_run = map(int, out[0].split())
    assert ctypes.sizeof(_lib.NetDesc) == s_net
    assert ctypes.sizeof(_lib.RunDesc) == s_run
    for line in out[1:]:
        if not line:
            continue
        which, name, off = line.split()
        st = _lib.NetDesc if which == "net" else _lib.RunDesc
        assert getattr(st, name).offset == int(off), (which, name)


def test_missing_library_fails_loudly(monkeypatch):
    from montecarlopredictivecoding_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", os.path.join(ROOT, "does_not_exist.so"))
    with pytest.raises(_lib.MCPCLibraryError):
        _lib.load()
