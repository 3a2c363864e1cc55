import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # PCD_TEST_EXPERIMENTS=1: run the suite against the EXPERIMENTS build (make -C com_amd/csrc EXPERIMENTS=1), so that the tests
    # of the measured-slower kernels (ggwin, pconv, the 128-channel window configuration) run instead of skipping
    if os.environ.get("PCD_TEST_EXPERIMENTS") == "1":
        from com_amd import _lib
        _lib.use_experiments_library()
    # PCD_TEST_WINDOW_HALF=<bits>: the suite with the 4-wave window configurations (option "subm_window_half": 2 = 32 channels,
    # 4 = 16 channels) -- set before any plan is built, plans / packs / launches of a width must agree
    if os.environ.get("PCD_TEST_WINDOW_HALF"):
        from com_amd import _lib
        _lib.set_option("subm_window_half", int(os.environ["PCD_TEST_WINDOW_HALF"]))


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    d = os.path.join(ROOT, "tests", "golden")

    def load(name):
        return dict(np.load(os.path.join(d, name + ".npz")))

    return load


@pytest.fixture
def pcd_option():
    """set(key, value): change a process-wide tuning option of the library for ONE test; every option touched is put back
    to the value it had (the table is global: a test that leaves `subm_window` changed flips the pack layouts and kernel
    choices of every test after it)."""
    from com_amd import _lib as L

    saved = {}

    def set_(key, value):
        saved.setdefault(key, L.get_option(key))
        L.set_option(key, value)

    yield set_
    for key, value in saved.items():
        L.set_option(key, value)
