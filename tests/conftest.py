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


@pytest.fixture
def tuning(mmm):
    """`tuning(lda_build="dense", grid_blocks=3, disable=("lda_rows16",), ...)`: mmm_ctx_set_tuning on the default context -- the caller's
    choices for the handles created from now on (include/mmmusig.h mmm_tuning_opts; they used to be environment variables read once per
    process).  Back to the defaults when the test ends; `tuning()` resets in between."""
    ctx = mmm.default_context()
    yield ctx.set_tuning
    ctx.set_tuning()
