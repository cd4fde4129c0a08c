import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _dropout_epoch_at_zero():
    """Tests name dropout masks by their two seeds.  The library adds the device's dropout epoch counter to the
    seeds at run time, and every captured training step of an earlier test has incremented it: start each test
    from zero."""
    mod = sys.modules.get("asr_amd.dropout")
    if mod is not None:
        for t in mod._EPOCHS.values():
            t.zero_()
    yield
