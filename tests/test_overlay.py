"""The drop-in overlay (mpcore/overlay.py) against the reference's REAL `modules` package.

Build container only: the reference does not travel to the GPU box (the tests skip there).  Every
`from modules... import name` / `import modules...` statement of the three callers SURVEY.md section 8(b) names
-- iterativedecomposition.py:12-19, mp.py:9-17, modules/multibanddict.py:5-9 -- is found by walking their ASTs
and resolved twice, with and without the overlay (tests/overlay_probe.py, a subprocess: it replaces
sys.modules wholesale):

* hot-path names resolve to mpcore's objects,
* every other name resolves to exactly the object the reference alone resolves it to -- including the
  names the reference itself cannot import in this image (`modules.UNet` is commented out of
  modules/__init__.py:1; `scipy.signal.morlet` is gone from this scipy): same failure, not a new one.
"""
import json
import os
import subprocess
import sys

import pytest

REF = "/root/reference"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(REPO, "tests", "overlay_probe.py")

needs_reference = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "modules")),
                                reason="needs the reference checkout (build container only)")

HOT = {
    "modules:iterative_loss": "mpcore.iterative.iterative_loss",
    "modules:unit_norm": "mpcore.matchingpursuit.unit_norm",
    "modules:sparsify2": "mpcore.sparse.sparsify2",
    "modules.matchingpursuit:dictionary_learning_step": "mpcore.matchingpursuit.dictionary_learning_step",
    "modules.matchingpursuit:sparse_code": "mpcore.matchingpursuit.sparse_code",
    "modules.matchingpursuit:build_scatter_segments": "mpcore.matchingpursuit.build_scatter_segments",
    "modules.normalization:unit_norm": "mpcore.matchingpursuit.unit_norm",
}


def _probe(mode):
    out = subprocess.run([sys.executable, PROBE, mode], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.fixture(scope="module")
def late():
    return _probe("late")


@pytest.fixture(scope="module")
def early():
    return _probe("early")


@needs_reference
def test_every_caller_import_resolves_as_in_the_reference(late, early):
    base = late["baseline"]
    assert late["n_names"] >= 40 and len(base) >= 35
    for report in (late, early):
        assert report["install"] == "overlay"
        over = report["overlay"]
        assert set(over) == set(base)
        for key, origin in base.items():
            if key in HOT:
                assert origin.startswith("modules."), (key, origin)     # the reference alone: its own function
                assert over[key] == HOT[key], (key, over[key])          # under the overlay: mpcore's
            else:
                assert over[key] == origin, (key, origin, over[key])    # untouched, failures included
                assert not over[key].startswith("mpcore"), key
    # names that must have resolved to the reference's real objects (not errors) in this image
    for key in ("modules:stft", "modules:sparsify", "modules:max_norm", "modules:flattened_multiband_spectrogram",
                "modules.anticausal:AntiCausalAnalysis", "modules:gammatone_filter_bank", "modules:HyperNetworkLayer",
                "modules.transfer:fft_convolve", "modules.decompose:fft_resample", "modules.stft:stft"):
        assert base[key].startswith("modules."), (key, base[key])
    # every hot-path name the callers import is covered by the table above
    assert {k for k, v in early["overlay"].items() if v.startswith("mpcore")} == set(HOT)


@needs_reference
def test_reference_modules_are_loaded_from_the_reference_files_and_only_names_are_swapped(early):
    assert early["matchingpursuit.file"] == os.path.join(REF, "modules", "matchingpursuit.py")
    patched = early["status"]["patched"]
    assert set(patched) == {"modules.normalization", "modules.sparse", "modules.conv", "modules.matchingpursuit",
                            "modules.iterative"}
    assert "sparse_code" in patched["modules.matchingpursuit"] and patched["modules.conv"] == ["fft_convolve", "torch_conv"]
    # names that matchingpursuit.py:4-5 pulls into its own namespace come out as mpcore's too
    assert early["matchingpursuit.inner_fft_convolve"] == "mpcore.matchingpursuit"
    assert early["matchingpursuit.inner_unit_norm"] == "mpcore.matchingpursuit"
    # the reference's own multiband wrapper stays, bound (multibanddict.py:8) to mpcore's encoder
    assert early["multibanddict.BandSpec"] == "modules.multibanddict"
    assert early["multibanddict.sparse_code"] == "mpcore.matchingpursuit"


@needs_reference
def test_late_install_rebinds_holders_and_uninstall_restores(late):
    assert late["multibanddict.sparse_code"] == "mpcore.matchingpursuit"   # bound before install(), rebound by identity
    assert late["restored"] == late["baseline"]
    assert late["multibanddict.sparse_code.restored"] is True


def test_run_module_executes_a_script_under_the_overlay(tmp_path):
    """python -m mpcore.run script.py: the script's own directory leads sys.path, as the interpreter does it."""
    (tmp_path / "modules").mkdir()
    (tmp_path / "modules" / "__init__.py").write_text("from .normalization import unit_norm, max_norm\n")
    (tmp_path / "modules" / "normalization.py").write_text(
        "def unit_norm(x, dim=-1, epsilon=1e-8):\n    return 'theirs'\n\ndef max_norm(x):\n    return 'theirs'\n")
    (tmp_path / "caller.py").write_text(
        "import sys\nfrom modules import unit_norm, max_norm\n"
        "print('RESULT', unit_norm.__module__, max_norm.__module__, sys.argv[1:])\n")
    env = dict(os.environ, PYTHONPATH=os.path.join(REPO, "matching-pursuit_amd"))
    out = subprocess.run([sys.executable, "-m", "mpcore.run", str(tmp_path / "caller.py"), "--flag", "7"],
                         capture_output=True, text=True, env=env, cwd="/tmp", timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")][0]
    assert line == "RESULT mpcore.matchingpursuit modules.normalization ['--flag', '7']"
