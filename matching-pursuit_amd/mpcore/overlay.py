"""Drop-in overlay: swap the matching-pursuit hot path of the reference's `modules` package for
mpcore, and leave every other name of that package alone.

The reference has no plugin boundary: callers bind functions by name (`from modules import
iterative_loss, stft, UNet, ...` -- iterativedecomposition.py:12-19; `from modules.matchingpursuit
import dictionary_learning_step, sparse_code` -- mp.py:17, modules/multibanddict.py:8).  A package
called `modules` placed ahead of the reference's on sys.path would shadow all 64 of its files; this
module instead installs an import hook in front of the reference's OWN `modules` package:

    import mpcore; mpcore.install()           # before the first `import modules` (or after: see below)
    python -m mpcore.run iterativedecomposition.py ...      # the same without editing the script

* every module of HOT_PATH is loaded from the reference's file, unmodified, and then the names listed
  for it are rebound to mpcore's implementations -- before any other module can import from it, so
  `modules/__init__.py:9,18-21`'s re-exports and `modules/multibanddict.py:8`'s imports come out
  as mpcore's without being touched;
* modules that were imported before install() are patched in place, and every module already
  holding one of the replaced objects (`from modules.matchingpursuit import sparse_code` executed
  earlier) is rebound by identity;
* everything else (`stft`, `UNet`, `modules.anticausal`, `modules.transfer`, ...) is the reference's;
* uninstall() puts the originals back.

Without a reference checkout on sys.path (the GPU test box) install() registers a small synthetic
`modules` package holding only the hot-path submodules, so code written against the reference's
names (`modules.matchingpursuit.sparse_code`) still resolves.
"""
import importlib
import importlib.abc
import importlib.util
import sys
import types

PACKAGE = "modules"

# reference module -> (mpcore module, names).  Line numbers: /root/reference/modules/<file>.
HOT_PATH = {
    "normalization": ("mpcore.matchingpursuit", ("unit_norm",)),                                  # :4-6
    "conv": ("mpcore.matchingpursuit", ("fft_convolve", "torch_conv")),                           # :4-53
    "sparse": ("mpcore.sparse", ("soft_dirac", "sparsify2")),                                     # :29-89
    "matchingpursuit": ("mpcore.matchingpursuit", (
        "build_scatter_segments", "flatten_atom_dict", "sparse_feature_map", "sparse_coding_loss",
        "sparse_code_to_differentiable_key_points", "sparse_code", "dictionary_learning_step",
        "SparseCodingLoss")),                                                                      # :20-463
    "iterative": ("mpcore.iterative", ("iterative_loss", "sort_channels_descending_norm")),       # :18-74
}
# Optional (install(multiband=True)): the reference's own BandSpec / MultibandDictionaryLearning already run on
# the patched sparse_code / dictionary_learning_step; these are mpcore's mirrors of the wrapper itself.
MULTIBAND = {
    "decompose": ("mpcore.decompose", ("fft_frequency_decompose", "fft_frequency_recompose", "fft_resample")),
    "multibanddict": ("mpcore.multibanddict", (
        "BandEncodingPackage", "BandSpec", "GlobalEventTuple", "LocalEventTuple", "MultibandDictionaryLearning")),
}
# names the reference re-exports from the package itself (modules/__init__.py:2,9,10-12,18-21,26)
PACKAGE_EXPORTS = (
    "unit_norm", "sparsify2", "dictionary_learning_step", "fft_convolve", "build_scatter_segments",
    "flatten_atom_dict", "sparse_feature_map", "sparse_coding_loss", "SparseCodingLoss", "iterative_loss")

_ORIGINALS = "__mpcore_originals__"
_state = {"finder": None, "table": {}, "standalone": False, "replaced": [], "rebound": []}


def _replacement(source, name):
    return getattr(importlib.import_module(source), name)


def _patch_module(module, source, names):
    """Rebind `names` in an executed reference module; remember what they were."""
    saved = module.__dict__.setdefault(_ORIGINALS, {})
    for name in names:
        new = _replacement(source, name)
        old = module.__dict__.get(name)
        if old is new:
            continue
        if name not in saved:
            saved[name] = old
        setattr(module, name, new)
        if old is not None:
            _state["replaced"].append((old, new))


def _rebind_holders(pairs, only_package=False):
    """Every loaded module that holds a replaced object under any name gets the replacement (the effect of
    `from modules.matchingpursuit import sparse_code` having run before the patch)."""
    if not pairs:
        return 0
    by_id = {id(old): new for old, new in pairs}
    n = 0
    for mod_name, mod in list(sys.modules.items()):
        if mod is None or not isinstance(mod, types.ModuleType):
            continue
        if only_package and not (mod_name == PACKAGE or mod_name.startswith(PACKAGE + ".")):
            continue
        if mod_name.startswith("mpcore"):
            continue
        d = getattr(mod, "__dict__", None)
        if not d:
            continue
        for key, val in list(d.items()):
            if key == _ORIGINALS:
                continue
            new = by_id.get(id(val))
            if new is not None and val is not new:
                d[key] = new
                _state["rebound"].append((mod, key, val))
                n += 1
    return n


class _PatchingLoader(importlib.abc.Loader):
    def __init__(self, inner, source, names):
        self.inner, self.source, self.names = inner, source, names

    def create_module(self, spec):
        return self.inner.create_module(spec)

    def exec_module(self, module):
        self.inner.exec_module(module)
        mark = len(_state["replaced"])
        _patch_module(module, self.source, self.names)
        # circular imports: a module that imported from this one while it was still executing
        _rebind_holders(_state["replaced"][mark:], only_package=True)

    def __getattr__(self, name):  # get_source, get_filename, is_package ...: the reference file's own loader
        return getattr(self.inner, name)


class _OverlayFinder(importlib.abc.MetaPathFinder):
    """Finds the reference's module with the ordinary machinery and wraps its loader."""

    def find_spec(self, fullname, path=None, target=None):
        entry = _state["table"].get(fullname)
        if entry is None:
            return None
        for finder in sys.meta_path:
            if finder is self or not hasattr(finder, "find_spec"):
                continue
            spec = finder.find_spec(fullname, path, target)
            if spec is not None:
                break
        else:
            return None
        if spec.loader is None:
            return None
        spec.loader = _PatchingLoader(spec.loader, *entry)
        return spec


def _reference_available():
    if PACKAGE in sys.modules:
        return not getattr(sys.modules[PACKAGE], "__mpcore_standalone__", False)
    try:
        return importlib.util.find_spec(PACKAGE) is not None
    except (ImportError, ValueError):
        return False


def _install_standalone(table):
    """No reference on sys.path: a synthetic `modules` package with the hot-path submodules only."""
    pkg = types.ModuleType(PACKAGE)
    pkg.__path__ = []
    pkg.__mpcore_standalone__ = True
    pkg.__doc__ = "mpcore stand-alone `modules` package (hot-path names only; no reference checkout found)"
    sys.modules[PACKAGE] = pkg
    for fullname, (source, names) in table.items():
        sub = types.ModuleType(fullname)
        sub.__package__ = PACKAGE
        sub.__mpcore_standalone__ = True
        for name in names:
            setattr(sub, name, _replacement(source, name))
        sys.modules[fullname] = sub
        setattr(pkg, fullname.split(".", 1)[1], sub)
    mp = sys.modules[PACKAGE + ".matchingpursuit"]
    for name in ("fft_convolve", "unit_norm"):  # names matchingpursuit.py:4-5 imports into its namespace
        setattr(mp, name, _replacement("mpcore.matchingpursuit", name))
    sys.modules[PACKAGE + ".iterative"].TensorTransform = _replacement("mpcore.iterative", "TensorTransform")
    for name in PACKAGE_EXPORTS:
        for sub in table:
            if hasattr(sys.modules[sub], name):
                setattr(pkg, name, getattr(sys.modules[sub], name))
    _state["standalone"] = True


def install(multiband=False, standalone=None):
    """Put the overlay in place (idempotent).  multiband=True also swaps the multiband wrapper classes.
    standalone: None = only if no reference `modules` package is importable; True = always; False = never
    (raise ImportError instead).  Returns "overlay" or "standalone"."""
    from . import _native
    _native.lib()  # fail loudly, now, if libmpcore.so has not been built: there is no other implementation
    table = {f"{PACKAGE}.{k}": v for k, v in HOT_PATH.items()}
    if multiband:
        table.update({f"{PACKAGE}.{k}": v for k, v in MULTIBAND.items()})
    have_reference = _reference_available()
    if standalone is True or (standalone is None and not have_reference):
        if have_reference and PACKAGE in sys.modules:
            raise ImportError("mpcore.install(standalone=True): the reference's `modules` package is already imported")
        _install_standalone(table)
        return "standalone"
    if not have_reference:
        raise ImportError(f"mpcore.install(standalone=False): no `{PACKAGE}` package on sys.path")
    _state["table"] = table
    if _state["finder"] is None:
        _state["finder"] = _OverlayFinder()
        sys.meta_path.insert(0, _state["finder"])
    # modules imported before install(): patch in place, then rebind whoever already holds the old objects
    mark = len(_state["replaced"])
    for fullname, (source, names) in table.items():
        mod = sys.modules.get(fullname)
        if mod is not None:
            _patch_module(mod, source, names)
    _rebind_holders(_state["replaced"][mark:])
    return "overlay"


def uninstall():
    """Remove the hook and restore every patched reference module (holders rebound by identity)."""
    if _state["finder"] is not None:
        try:
            sys.meta_path.remove(_state["finder"])
        except ValueError:
            pass
        _state["finder"] = None
    if _state["standalone"]:
        for name in [n for n, m in sys.modules.items() if getattr(m, "__mpcore_standalone__", False)]:
            del sys.modules[name]
        _state["standalone"] = False
    for mod in list(sys.modules.values()):
        saved = getattr(mod, "__dict__", {}).get(_ORIGINALS) if isinstance(mod, types.ModuleType) else None
        if saved:
            for name, old in saved.items():
                if old is None:
                    mod.__dict__.pop(name, None)
                else:
                    setattr(mod, name, old)
            del mod.__dict__[_ORIGINALS]
    for mod, key, old in reversed(_state["rebound"]):
        mod.__dict__[key] = old
    _state["rebound"] = []
    _state["replaced"] = []
    _state["table"] = {}


def status():
    """-> {"mode": "overlay" | "standalone" | None, "patched": {module: [names]}} for diagnostics and tests."""
    patched = {}
    for name, mod in sys.modules.items():
        saved = getattr(mod, "__dict__", {}).get(_ORIGINALS) if isinstance(mod, types.ModuleType) else None
        if saved:
            patched[name] = sorted(saved)
    mode = "standalone" if _state["standalone"] else ("overlay" if _state["finder"] is not None else None)
    return {"mode": mode, "patched": patched}
