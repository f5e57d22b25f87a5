"""Launcher: run one of the reference's scripts UNCHANGED on the MI355X engine.

    python -m montecarlopredictivecoding_amd.run figure_2.py [script arguments ...]

north_star: "the PCLayer/PCTrainer constructor and callback API surface is preserved so figure_*.py scripts run unchanged".  The
reference's scripts reach the hot path through four import lines (/root/reference/figure_2.py:14-21, figure_6.py:11-15):

    import predictive_coding as pc
    from utils.model import ...                     (random_step, get_model, the loss functions, the x initialisers)
    from utils.training_evaluation import ...       (get_pc_trainer, get_mcpc_trainer, sample_pc, the evaluators)
    from utils.data import ... / from utils.plotting import ...        (data loaders, plots: out of this repository's scope)

and they sit in a directory that HOLDS regular packages of those names (`predictive_coding/__init__.py`, `utils/__init__.py`).  Python
puts the script's directory at `sys.path[0]`, ahead of `PYTHONPATH`, so alias packages on `PYTHONPATH` lose to the script's own
(missing #2: the script would silently run the reference's CPU loop).  The module search order cannot fix that; the
import system can: :func:`install` puts a finder FIRST on `sys.meta_path` that answers

    predictive_coding, predictive_coding.pc_layer, predictive_coding.pc_trainer, predictive_coding.utils
    utils.model, utils.training_evaluation

with this package's counterparts and declines everything else -- `utils` itself, `utils.data`, `utils.plotting`, `ResNet9`, ... stay
the script's own (when no `utils` package exists anywhere, an empty one is provided so that `utils.model` has a parent).  The script
is then executed with `runpy.run_path(..., run_name="__main__")`, its directory at `sys.path[0]` and `sys.argv` as if it had been
started directly.

Names the engine-backed modules do not define.  The reference's `utils/training_evaluation.py` also holds helpers that are not on the hot
path and that its scripts import from the same module -- `train`, `test`, `MNIST_LinearClassifier`, `kl_divergence_discrete`
(`figure_2.py:17-19`), `KLdivergence`, `get_paired_stat` (`figure_5.py:14`), `get_fid` (`table_1.py:8`).  They are not restated here: an
attribute this package's module lacks is looked up (PEP 562 `__getattr__`) in the SCRIPT'S OWN module of that name -- found with the
finders behind this one, executed once under a private name with this finder active, so that ITS `import predictive_coding` /
`from utils.model import ...` resolve to the engine too -- and a name neither has raises the usual `ImportError`.

`montecarlopredictivecoding_amd.run.install()` is the same switch for an interpreter that is already running (a notebook, a test).
"""
import importlib
import importlib.abc
import importlib.machinery
import importlib.util
import os
import sys
import types

# reference module name -> this package's module that stands in for it
ALIASES = {
    "predictive_coding": "montecarlopredictivecoding_amd.predictive_coding",
    "predictive_coding.pc_layer": "montecarlopredictivecoding_amd.predictive_coding.pc_layer",
    "predictive_coding.pc_trainer": "montecarlopredictivecoding_amd.predictive_coding.pc_trainer",
    "predictive_coding.utils": "montecarlopredictivecoding_amd.predictive_coding.utils",
    "utils.model": "montecarlopredictivecoding_amd.utils.model",
    "utils.training_evaluation": "montecarlopredictivecoding_amd.utils.training_evaluation",
}


class _AliasLoader(importlib.abc.Loader):
    """Hands the import system an already-imported module of this package under the reference's name."""

    def __init__(self, target):
        self._target = target
        self._own = None

    def create_module(self, spec):
        module = importlib.import_module(self._target)
        # importlib's _init_module_attrs is about to overwrite the module's __spec__ / __loader__ / __package__-related attributes with the
        # ALIAS spec (name 'predictive_coding', this loader) -- for the real montecarlopredictivecoding_amd.* module, process-wide: relative
        # imports inside it would then warn (`__package__ != __spec__.parent`) and resolve against the wrong parent on Pythons that stop
        # consulting __package__, importlib.reload would be a silent no-op.  Remember what the module really is ...
        self._own = {k: getattr(module, k) for k in ("__spec__", "__loader__", "__name__", "__package__", "__file__", "__cached__", "__path__")
                     if hasattr(module, k)}
        return module

    def exec_module(self, module):
        # ... and put it back: the module keeps its own identity, sys.modules merely knows it under a second name
        for k, v in (self._own or {}).items():
            try:
                setattr(module, k, v)
            except (AttributeError, TypeError):
                pass


class _EmptyPackageLoader(importlib.abc.Loader):
    def create_module(self, spec):
        return None

    def exec_module(self, module):
        module.__path__ = []


class EngineFinder(importlib.abc.MetaPathFinder):
    """First on sys.meta_path: the reference's hot-path modules resolve to the engine-backed ones, whatever the script's directory holds."""

    def find_spec(self, fullname, path=None, target=None):
        aliased = ALIASES.get(fullname)
        if aliased is not None:
            is_pkg = fullname == "predictive_coding"
            spec = importlib.machinery.ModuleSpec(fullname, _AliasLoader(aliased), is_package=is_pkg)
            return spec
        if fullname == "utils":
            # the script's own package if there is one (its utils.data / utils.plotting must stay importable) ...
            for finder in sys.meta_path:
                if finder is self or not hasattr(finder, "find_spec"):
                    continue
                spec = finder.find_spec(fullname, path, target)
                if spec is not None:
                    return spec
            # ... else an empty parent for utils.model / utils.training_evaluation
            return importlib.machinery.ModuleSpec(fullname, _EmptyPackageLoader(), is_package=True)
        return None


_FINDER = None
_OWN = {}           # reference module name -> the script's own module of that name (or the exception its import raised)


def script_own_attr(reference_name, attr):
    """`attr` of the SCRIPT'S OWN module `reference_name` (e.g. the reference's utils/training_evaluation.py next to the script) -- what an
    engine-backed module falls back to for a name it does not define.  Raises AttributeError when there is no such module or name."""
    if _FINDER is None or _FINDER not in sys.meta_path:
        raise AttributeError(attr)
    if reference_name not in _OWN:
        own = None
        try:
            parent_name = reference_name.rpartition(".")[0]
            path = None
            if parent_name:
                parent = importlib.import_module(parent_name)
                path = getattr(parent, "__path__", None)
            spec = None
            for finder in sys.meta_path:
                if finder is _FINDER or not hasattr(finder, "find_spec"):
                    continue
                spec = finder.find_spec(reference_name, path, None)
                if spec is not None:
                    break
            if spec is not None and spec.origin and os.path.isfile(spec.origin):
                # executed under a private name (sys.modules keeps the engine-backed module under the reference's), from the same file
                private = importlib.util.spec_from_file_location("_mcpc_script_own." + reference_name, spec.origin)
                own = importlib.util.module_from_spec(private)
                private.loader.exec_module(own)
        except Exception as exc:            # the script's own module does not import here (a missing dependency, a decoy): no fallback
            own = exc
        _OWN[reference_name] = own
    own = _OWN[reference_name]
    if own is None or isinstance(own, Exception):
        why = "there is no such module next to the script" if own is None else f"importing it raised {type(own).__name__}: {own}"
        raise AttributeError(f"{attr!r} is not one of the engine-backed names of {reference_name!r}, and the script's own {reference_name} "
                             f"cannot supply it ({why})")
    try:
        return getattr(own, attr)
    except AttributeError:
        raise AttributeError(f"neither the engine-backed {reference_name!r} nor the script's own defines {attr!r}") from None


def install():
    """Install the finder (idempotent) and drop already-imported modules of the aliased names so that the next import resolves here."""
    global _FINDER
    if _FINDER is None:
        _FINDER = EngineFinder()
    if _FINDER in sys.meta_path:
        sys.meta_path.remove(_FINDER)
    sys.meta_path.insert(0, _FINDER)
    for name in list(sys.modules):
        if name in ALIASES:
            mod = sys.modules[name]
            if getattr(mod, "__name__", name) != ALIASES[name]:
                del sys.modules[name]
    return _FINDER


def uninstall():
    if _FINDER is not None and _FINDER in sys.meta_path:
        sys.meta_path.remove(_FINDER)
    _OWN.clear()


def run_script(path, argv=()):
    """Execute `path` as `__main__` with the finder installed, `sys.argv` = [path, *argv] and the script's directory at sys.path[0]."""
    import runpy
    path = os.path.abspath(path)
    if not os.path.isfile(path):
        raise FileNotFoundError(path)
    install()
    sys.argv = [path] + list(argv)
    script_dir = os.path.dirname(path)
    # `python script.py` puts the script's directory first; `python -m ...` put the current directory (or '') there instead
    if sys.path and sys.path[0] in ("", os.getcwd()):
        sys.path[0] = script_dir
    else:
        sys.path.insert(0, script_dir)
    return runpy.run_path(path, run_name="__main__")


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    if not argv or argv[0] in ("-h", "--help"):
        print(__doc__.strip().split("\n\n")[0], file=sys.stderr)
        return 2
    run_script(argv[0], argv[1:])
    return 0


if __name__ == "__main__":
    # `python -m` executes this file as `__main__`: delegate to the module under its real name, so that there is ONE finder and one
    # table of the script's own modules (script_own_attr is reached through `montecarlopredictivecoding_amd.run`)
    from montecarlopredictivecoding_amd.run import main as _main
    sys.exit(_main())
