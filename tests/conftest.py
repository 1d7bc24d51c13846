"""pytest configuration: markers, import path, shared fixtures."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import util

    return util.load_oracle()


@pytest.fixture(scope="session")
def golden():
    import util

    return util.load_golden()


@pytest.fixture(scope="session")
def engine():
    """One engine for the whole GPU session, sized for the largest test (2^20)."""
    import webgpu_msm_bls12_377_amd as msm

    eng = msm.MsmEngine(1 << 20)
    yield eng
    eng.close()
