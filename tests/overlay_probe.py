"""Helper of tests/test_overlay.py (run as a subprocess; build container only -- needs /root/reference).

Imports the reference's REAL `modules` package from where it lies (nothing is copied), with permissive
stand-ins for the third-party packages this image lacks (zounds, librosa, conjure, ...: none of them is on the
matching-pursuit path), and reports where every name that the reference's callers import from `modules`
resolves to -- with and without mpcore's overlay.

    python tests/overlay_probe.py early      # mpcore.install() before the first `import modules`
    python tests/overlay_probe.py late       # baseline -> install() after everything is imported -> uninstall()

Prints one JSON object on the last line of stdout.
"""
import ast
import importlib
import importlib.abc
import importlib.machinery
import io
import json
import os
import sys
import types
from contextlib import redirect_stdout

REF = os.environ.get("MP_REFERENCE", "/root/reference")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CALLERS = ("iterativedecomposition.py", "mp.py", "modules/multibanddict.py")
MISSING = {"zounds", "librosa", "soundfile", "lmdb", "conjure", "boto3", "botocore", "unittest2", "jax"}


class _Anything(type):
    """A class that can be subclassed, called, subscripted, used as a decorator ... and means nothing."""

    def __getattr__(cls, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _anything(name)

    def __call__(cls, *a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return _anything("instance")

    def __getitem__(cls, k):
        return cls

    def __iter__(cls):
        return iter(())


def _anything(name):
    return _Anything(name, (), {})


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        v = _anything(name)
        setattr(self, name, v)
        return v


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in MISSING:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def imported_names():
    """Every (module, name) that CALLERS import from the `modules` package, by walking their ASTs."""
    out = []
    for rel in CALLERS:
        tree = ast.parse(open(os.path.join(REF, rel)).read())
        for node in ast.walk(tree):
            if isinstance(node, ast.ImportFrom) and node.level == 0 and node.module and (
                    node.module == "modules" or node.module.startswith("modules.")):
                for alias in node.names:
                    out.append((rel, node.lineno, node.module, alias.name))
            elif isinstance(node, ast.Import):
                for alias in node.names:
                    if alias.name == "modules" or alias.name.startswith("modules."):
                        out.append((rel, node.lineno, alias.name, None))
    return out


def resolve(names):
    """-> {"module:name": origin}; origin = "<defining module>.<qualname>" or "ERR <exception type>"."""
    res = {}
    for rel, line, module, name in names:
        key = f"{module}:{name}"
        try:
            mod = importlib.import_module(module)
            obj = mod if name is None else getattr(mod, name)
            origin = f"{getattr(obj, '__module__', type(obj).__module__)}.{getattr(obj, '__qualname__', type(obj).__name__)}"
            if isinstance(obj, types.ModuleType):
                origin = f"module {obj.__name__} @ {os.path.dirname(getattr(obj, '__file__', '') or '')}"
        except Exception as e:  # what the reference itself raises here (e.g. scipy.signal.morlet is gone)
            origin = f"ERR {type(e).__name__}"
        res[key] = origin
    return res


def main(mode):
    sys.meta_path.insert(0, _StubFinder())
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
    os.chdir(REF)
    names = imported_names()
    report = {"mode": mode, "n_names": len(names)}
    noise = io.StringIO()
    with redirect_stdout(noise):  # the reference prints while importing
        import mpcore
        if mode == "early":
            report["install"] = mpcore.install()
            report["overlay"] = resolve(names)
            from mpcore import overlay
            report["status"] = overlay.status()
            # the reference's own wrapper binds sparse_code at import time (multibanddict.py:8): must be mpcore's
            import modules.multibanddict as mb
            report["multibanddict.sparse_code"] = mb.sparse_code.__module__
            report["multibanddict.BandSpec"] = mb.BandSpec.__module__
            import modules.matchingpursuit as m
            report["matchingpursuit.file"] = m.__file__
            report["matchingpursuit.inner_fft_convolve"] = m.fft_convolve.__module__
            report["matchingpursuit.inner_unit_norm"] = m.unit_norm.__module__
        else:
            report["baseline"] = resolve(names)
            import modules.multibanddict as mb
            before = mb.sparse_code
            report["install"] = mpcore.install()
            report["overlay"] = resolve(names)
            report["multibanddict.sparse_code"] = mb.sparse_code.__module__
            mpcore.uninstall()
            report["restored"] = resolve(names)
            report["multibanddict.sparse_code.restored"] = mb.sparse_code is before
    print(json.dumps(report))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "early")
