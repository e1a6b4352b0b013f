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
