import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def kats():
    with open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): built on demand with gcc."""
    from oracle import oracle as orc
    orc.lib()
    return orc


@pytest.fixture(scope="session")
def mmm():
    """The product package (directory name has a dot, so it is loaded through the root shim)."""
    import mmm_pkg
    return mmm_pkg.load()
