import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_sessionstart(session):
    """Build the in-tree native pieces if a fresh checkout has not done so yet (hipcc cross-compiles
    gfx950 without a GPU; the oracle needs gcc).  Failures surface in the tests that need them."""
    try:
        from lmcache_amd import native
        if not os.path.exists(native.SO_PATH):
            native.build()
        from oracle import lmc_oracle
        lmc_oracle.build()
    except Exception as e:  # pragma: no cover
        print(f"[conftest] native build skipped: {e!r}")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; never imported by the product)."""
    from oracle import lmc_oracle
    lmc_oracle.build()
    return lmc_oracle
