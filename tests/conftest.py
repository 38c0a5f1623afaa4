import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "helmholtz-x_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# the library checks every caller-supplied index array on the device before it follows it (include/hx.h: HX_DEBUG_BOUNDS): a
# lifetime bug of a test's temporaries then fails with an error code instead of faulting the GPU of a shared machine
os.environ.setdefault("HX_DEBUG_BOUNDS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def ref_logs():
    import json
    with open(os.path.join(GOLDEN, "reference_logs.json")) as fh:
        return json.load(fh)
