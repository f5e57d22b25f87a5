"""ctypes binding of libmcpc.so (C ABI: include/mcpc.h).  No torch types cross this boundary.

The library is built in-tree by ``__graft_entry__.build()`` /
``montecarlopredictivecoding_amd/csrc/Makefile``.  There is NO fallback: if the shared object is
missing or does not load, every entry point raises ``MCPCLibraryError``.
"""
import ctypes as C
import os

MAX_LATENT = 6
COMM_ID_BYTES = 128
ENERGY_COLS = MAX_LATENT + 2
ABI_VERSION = 4

ACT_IDENTITY, ACT_RELU, ACT_TANH = 0, 1, 2
LOSS_NONE, LOSS_GAUSSIAN, LOSS_BERNOULLI = 0, 1, 2
XOPT_SGD, XOPT_ADAM = 0, 1
NOISE_NONE, NOISE_PHILOX, NOISE_EXTERNAL = 0, 1, 2
ENERGY_NONE, ENERGY_LAST, ENERGY_ALL = 0, 1, 2

# MCPC_LIB: developer override to load a diagnostic build (e.g. libmcpc_stamps.so) or a `make variant` A/B library; a library that
# reports exp=1 (built with a timing-experiment switch: wrong results on purpose) is refused unless MCPC_ALLOW_EXP=1
LIB_PATH = os.environ.get("MCPC_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmcpc.so")


class MCPCLibraryError(RuntimeError):
    pass


class MCPCError(RuntimeError):
    """A libmcpc call returned a non-zero status."""

    def __init__(self, code, message):
        super().__init__(f"libmcpc error {code}: {message}")
        self.code = code


class NetDesc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("n_latent", C.c_int32),
        ("n_in", C.c_int32),
        ("sizes", C.c_int32 * MAX_LATENT),
        ("acts", C.c_int32 * MAX_LATENT),
        ("ecoef", C.c_float * MAX_LATENT),
        ("n_out", C.c_int32),
        ("batch", C.c_int32),
        ("device", C.c_int32),
        ("spill_budget_bytes", C.c_int64),
        ("tuning", C.c_char_p),
    ]


class RunDesc(C.Structure):
    _fields_ = [
        ("T", C.c_int32), ("t_begin", C.c_int32), ("n_steps", C.c_int32),
        ("loss_kind", C.c_int32), ("mask_start", C.c_int32), ("loss_var", C.c_double),
        ("xopt_kind", C.c_int32), ("adam_step0", C.c_int32), ("lr", C.c_double),
        ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
        ("update_x", C.c_int32),
        ("noise_mode", C.c_int32), ("noise_var", C.c_double),
        ("seed", C.c_uint64), ("step_base", C.c_uint64), ("chain_base", C.c_uint64),
        ("ext_noise", C.c_void_p * MAX_LATENT),
        ("acc_begin", C.c_int32), ("acc_end", C.c_int32), ("acc_reset", C.c_int32),
        ("energy_mode", C.c_int32),
        ("energies_out", C.c_void_p),
        ("rec_begin", C.c_int32), ("rec_stride", C.c_int32), ("rec_count", C.c_int32),
        ("rec_x", C.c_void_p * MAX_LATENT),
        ("rec_out", C.c_void_p),
        ("xgrad", C.c_void_p * MAX_LATENT),
    ]


# every symbol include/mcpc.h declares: (restype, argtypes)
SYMBOLS = {
    "mcpc_abi_version": (C.c_int, []),
    "mcpc_build_info": (C.c_char_p, []),
    "mcpc_last_error": (C.c_char_p, []),
    "mcpc_create": (C.c_int, [C.POINTER(NetDesc), C.POINTER(C.c_void_p)]),
    "mcpc_destroy": (C.c_int, [C.c_void_p]),
    "mcpc_bind_params": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "mcpc_params_changed": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mcpc_bind_inputs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mcpc_bind_target": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mcpc_load_state": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p]),
    "mcpc_store_state": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p]),
    "mcpc_store_adam_state": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p]),
    "mcpc_run": (C.c_int, [C.c_void_p, C.POINTER(RunDesc), C.c_void_p]),
    "mcpc_read_param_grads": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p]),
    "mcpc_read_param_grads_flat": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p]),
    "mcpc_param_count": (C.c_int64, [C.c_void_p]),
    "mcpc_comm_unique_id": (C.c_int, [C.c_void_p]),
    "mcpc_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "mcpc_allreduce_grads": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "mcpc_comm_destroy": (C.c_int, [C.c_void_p]),
    "mcpc_philox_normals": (C.c_int, [C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_uint64, C.c_int, C.c_int,
                                      C.c_void_p, C.c_int, C.c_void_p]),
    "mcpc_query": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                             C.POINTER(C.c_int32)]),
    "mcpc_step_kernel_name": (C.c_char_p, [C.c_void_p]),
    "mcpc_sync_check": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mcpc_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "mcpc_last_step_kernel_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32),
                                           C.POINTER(C.c_int64)]),
    "mcpc_last_shader_clock_ghz": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "mcpc_debug_poison_lds": (C.c_int, [C.c_int, C.c_uint32, C.c_void_p]),
}

_lib = None


def load():
    """Load libmcpc.so once; raise loudly if it is absent (no CPU / eager fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MCPCLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C montecarlopredictivecoding_amd/csrc`. There is no fallback path.")
    # torch must be imported first: its wheel bundles the HIP runtime (SONAME libamdhip64.so.7) and
    # libmcpc.so has to bind to THAT copy so that streams and allocations are shared.  Loading the
    # system copy first leaves the process with two runtimes ("no ROCm-capable device is detected").
    import torch  # noqa: F401
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as exc:   # pragma: no cover - depends on the machine
        raise MCPCLibraryError(f"could not load {LIB_PATH}: {exc}") from exc
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise MCPCLibraryError(f"{LIB_PATH} does not export {name}") from exc
        fn.restype = res
        fn.argtypes = args
    v = lib.mcpc_abi_version()
    if v != ABI_VERSION:
        raise MCPCLibraryError(f"libmcpc ABI {v} != binding ABI {ABI_VERSION}: rebuild the library")
    info = parse_build_info(lib.mcpc_build_info().decode("utf-8", "replace"))
    if info.get("exp") != "0" and os.environ.get("MCPC_ALLOW_EXP") != "1":
        # a timing-experiment build computes wrong results on purpose (csrc/mcpc_build.h): never the product, never by accident
        raise MCPCLibraryError(f"{LIB_PATH} is a timing-experiment build ({info.get('flags')}): it computes wrong results on purpose. "
                               "Set MCPC_ALLOW_EXP=1 to load it for a timing A/B; unset MCPC_LIB to load the product.")
    _lib = lib
    return lib


def parse_build_info(line: str) -> dict:
    """'libmcpc abi=4 arch=gfx950 csrc=... commit=... exp=0 stamps=0 flags=[...]' -> dict (flags: the bracketed string)."""
    head, _, flags = line.partition(" flags=[")
    out = {"flags": flags[:-1] if flags.endswith("]") else flags}
    for word in head.split()[1:]:
        key, _, val = word.partition("=")
        out[key] = val
    return out


def build_info() -> dict:
    """What the loaded libmcpc.so reports about itself (include/mcpc.h: mcpc_build_info), plus the path it was loaded from."""
    info = parse_build_info(load().mcpc_build_info().decode("utf-8", "replace"))
    info["path"] = LIB_PATH
    return info


def csrc_sha(root=None) -> str:
    """sha256[:16] over the kernel sources of THIS tree, as csrc/Makefile computes it for the library (every *.h *.hip *.inc of csrc/
    by name, then include/mcpc.h): `build_info()["csrc"] == csrc_sha()` says the loaded binary is a build of the sources beside it."""
    import glob
    import hashlib
    pkg = os.path.dirname(os.path.abspath(__file__)) if root is None else os.path.join(root, "montecarlopredictivecoding_amd")
    csrc = os.path.join(pkg, "csrc")
    names = sorted(os.path.basename(p) for ext in ("*.h", "*.hip", "*.inc") for p in glob.glob(os.path.join(csrc, ext)))
    h = hashlib.sha256()
    for path in [os.path.join(csrc, n) for n in names] + [os.path.join(os.path.dirname(pkg), "include", "mcpc.h")]:
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def check(code):
    if code != 0:
        raise MCPCError(code, load().mcpc_last_error().decode("utf-8", "replace"))
