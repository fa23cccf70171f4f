"""pytest configuration.  `-m "not gpu"`: oracle vs golden vectors, host logic, library/ABI checks (runs without a GPU).
`-m gpu`: parity of the HIP path against the oracle, through the C-ABI (needs an MI355X)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Make sure every native library exists (build() is a no-op when they are up to date)."""
    import __graft_entry__ as g
    g.build()
    return True


@pytest.fixture(scope="session")
def oracle(built):
    """TEST-ONLY checker: CPU restatement of the reference (oracle/liboracle.so)."""
    from llama_go_amd.mlapi import MLLib
    return MLLib(os.path.join(ROOT, "oracle", "liboracle.so"))


@pytest.fixture(scope="session")
def product(built):
    """The product: libllamago.so -> libllamahip.so -> MI355X.  No CPU fallback exists."""
    from llama_go_amd.mlapi import load_product
    lib = load_product()
    lib.lib.llamago_DeviceCount.restype = __import__("ctypes").c_int
    if lib.lib.llamago_DeviceCount() < 1:
        pytest.fail("gpu-marked test but no HIP device is visible")
    return lib
