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


def pytest_runtest_logreport(report):
    # how many GPU tests of this session have passed so far: tests/test_gpu_zz_routes.py only judges the kernel routes of a (nearly) full suite
    if report.when == "call" and report.passed and "gpu" in report.keywords:
        _SESSION["gpu_passed"] = _SESSION.get("gpu_passed", 0) + 1


_SESSION = {}


@pytest.fixture(scope="session")
def gpu_tests_passed():
    return lambda: _SESSION.get("gpu_passed", 0)


@pytest.fixture(scope="session")
def built():
    """Make sure every native library exists (build() is a no-op when they are up to date)."""
    import __graft_entry__ as g
    g.build()
    return True


class _CheckerModelPool:
    """The checker's synthetic models are generated on ONE host core (hundreds of millions of hashed weights for a 7B-shape slice: seconds per
    model) and most GPU tests want the same few: keep the most recent ones alive across tests instead of regenerating them.  Tests only read a
    checker model, except QuantizeQ8, which replaces its weights - a quantised model is therefore a pool entry of its own."""

    def __init__(self, make, keep=5):
        self.make, self.keep, self.models = make, keep, {}

    def get(self, key, hp):
        m = self.models.pop(key, None)
        if m is None:
            m = self.make(hp, *key[5:8])
            if key[-1] == "q8":
                m.QuantizeQ8()
            while len(self.models) >= self.keep:
                self.models.pop(next(iter(self.models))).free()   # least recently used
        self.models[key] = m
        return m

    def close(self):
        for m in self.models.values():
            m.free()
        self.models = {}


class _PooledModel:
    """What oracle.NewSyntheticModel hands out: the pool's model behind the Model interface; free() leaves it to the pool."""

    def __init__(self, pool, key, hp):
        self.__dict__.update(_pool=pool, _key=key, _hp=hp, _m=pool.get(key, hp))

    def QuantizeQ8(self):
        self.__dict__.update(_key=self._key + ("q8",))
        self.__dict__.update(_m=self._pool.get(self._key, self._hp))
        return self

    def free(self):
        pass

    def __getattr__(self, name):
        return getattr(self._m, name)


@pytest.fixture(scope="session")
def oracle(built):
    """TEST-ONLY checker: CPU restatement of the reference (oracle/liboracle.so)."""
    from llama_go_amd.mlapi import MLLib
    lib = MLLib(os.path.join(ROOT, "oracle", "liboracle.so"))
    pool = _CheckerModelPool(lib.NewSyntheticModel)

    def pooled(hp, seed=1234, layer0=0, layer1=0):
        return _PooledModel(pool, (hp.vocabSize, hp.embdSize, hp.multSize, hp.headsCount, hp.layersCount, seed, layer0, layer1), hp)

    lib.NewSyntheticModel = pooled
    yield lib
    pool.close()


@pytest.fixture(scope="session")
def product(built, tmp_path_factory):
    """The product: libllamago.so -> libllamahip.so -> MI355X.  No CPU fallback exists."""
    # route log (include/llamahip.h lh_route_log; tests/test_gpu_zz_routes.py): this process and the worker processes the pipeline tests spawn
    # note the kernel family of every launch; the workers append theirs to this file when they exit (mlapi.load_product)
    os.environ.setdefault("LLAMAHIP_ROUTE_FILE", str(tmp_path_factory.mktemp("routes") / "routes.txt"))
    from llama_go_amd.mlapi import load_product
    lib = load_product()
    lib.lib.llamago_DeviceCount.restype = __import__("ctypes").c_int
    if lib.lib.llamago_DeviceCount() < 1:
        pytest.fail("gpu-marked test but no HIP device is visible")
    return lib
