import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "wavesim"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    def rd(name):
        with open(os.path.join(GOLDEN, name), "rb") as f:
            return f.read()
    return rd


@pytest.fixture(scope="session")
def oracle():
    import oracle_py
    oracle_py.build()
    return oracle_py


@pytest.fixture(scope="session")
def ctx():
    """A real rcx context; GPU tests only.  Fails (not skips) if the HIP library or device is missing."""
    import torch  # noqa: F401  (loads the HIP runtime torch ships with first)
    import rust_compress_amd as R
    c = R.Context()
    yield c
    c.close()
