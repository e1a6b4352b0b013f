import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "matching-pursuit_amd")
for p in (PKG, REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, "tests", "golden")

# The reference's names (`modules.matchingpursuit.sparse_code`, ...) as the tests use them.  No reference
# checkout is on sys.path here or on the GPU box, so this registers mpcore's stand-alone `modules` package
# (mpcore/overlay.py); the overlay over the REAL package is exercised by tests/test_overlay.py in a subprocess.
import mpcore  # noqa: E402

assert mpcore.install(multiband=True) == "standalone"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/mp_oracle.py): test infrastructure, never the product."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import mp_oracle
    mp_oracle.build()
    return mp_oracle


@pytest.fixture(autouse=True)
def _library_knobs_back_to_their_defaults():
    """mp_tune's knobs are process-wide (include/mpcore.h); tests that turn one share a process with every other test.
    Whatever a test leaves behind -- by a failed assertion ahead of its own clean-up, too -- is reset here, so that no
    test's result depends on which test ran before it."""
    yield
    from mpcore import _native as nat
    if not os.path.exists(nat.LIB_PATH):
        return
    for key, default in ((nat.MP_TUNE_TAU, 0), (nat.MP_TUNE_SCREEN_PPS, 0), (nat.MP_TUNE_GROUPS, 4), (nat.MP_TUNE_AUDIT, 0),
                         (nat.MP_TUNE_PERSIST_SHARDS, 0), (nat.MP_TUNE_PERSIST_WORKERS, 0), (nat.MP_TUNE_PERSIST_SELECTS, 0),
                         (nat.MP_TUNE_LAZY_MARGIN, 0), (nat.MP_TUNE_LAZY_REUSE, 0), (nat.MP_TUNE_LAZY_RADIUS, 0),
                         (nat.MP_TUNE_PERSIST_PRESCAN, 1), (nat.MP_TUNE_CLEAR_MEMSET, 0), (nat.MP_TUNE_LAZY_FORCE, 0),
                         (nat.MP_TUNE_LAZY_COMPACT, 1), (nat.MP_TUNE_PERSIST_FINE, 0)):
        nat.tune(key, default)
