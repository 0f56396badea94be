import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DATA = os.path.join(ROOT, "tests", "golden", "data")

# the shells' on-disk .pvar side-cache stays out of the user's cache directory during tests; the test of the cache
# itself points it at a temporary directory
os.environ.setdefault("PLINKING_PVAR_CACHE", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def data_path(name):
    return os.path.join(DATA, name)


@pytest.fixture(scope="session")
def oracle():
    """The CPU checker (oracle/): compiled on demand, never used by the product."""
    from oracle import oracle as orc
    return orc


@pytest.fixture(scope="session")
def lib():
    import plinking_duck_amd.lib as L
    return L


@pytest.fixture(scope="session")
def gpu_lib(lib):
    if lib.device_count() < 1:
        pytest.fail("GPU test selected but libpgenhip sees no HIP device (there is no CPU fallback)")
    return lib
