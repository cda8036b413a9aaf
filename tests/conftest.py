import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def rta():
    import raytracing_amd

    raytracing_amd.load_libraries()
    return raytracing_amd


@pytest.fixture(scope="session")
def orc():
    import oracle_bindings

    oracle_bindings.lib()
    return oracle_bindings


@pytest.fixture(scope="session")
def gpu(rta):
    if rta.device_count() < 1:
        pytest.fail("GPU test selected but no HIP device is visible (the HIP path has no CPU fallback)")
    return 0
