import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def record_value(name, value):
    """Module-level form of the `record` fixture (for tests that do not take fixtures by name)."""
    path = os.environ.get("DC_TEST_LOG")
    if path:
        with open(path, "a") as fh:
            fh.write(f"{name}\t{value}\n")
    return value


@pytest.fixture(scope="session")
def record():
    """record(name, value): append a measured value (PSNR, rel-L2 ...) to the file named by $DC_TEST_LOG, so that the bars in
    the tests can be kept close to what the hardware measures.  Without the variable it does nothing."""
    path = os.environ.get("DC_TEST_LOG")

    def rec(name, value):
        if path:
            with open(path, "a") as fh:
                fh.write(f"{name}\t{value}\n")
    return rec
