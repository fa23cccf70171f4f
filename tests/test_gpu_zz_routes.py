"""Route coverage: every __global__ kernel of the product sources must be reached by the GPU parity suite - or be deleted.
The library notes the kernel family of every launch while the route log is on (include/llamahip.h: lh_route_log / lh_route_names; csrc/common.h
LH_LAUNCH); with LLAMAHIP_ROUTE_FILE set (tests/conftest.py) mlapi.load_product switches it on in this process and in the worker processes the
pipeline tests spawn (they append their names to the file when they exit), and this file - last in collection order - compares the
union over the session with the __global__ definitions parsed out of llama.go_amd/csrc.  It only judges a (nearly) complete session: run alone
or with -k it skips."""
import ctypes
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "llama.go_amd", "csrc")

# kernels no parity test can reach, each with the reason
EXEMPT = {
    "k_park": "lh_llama_profile_decode's stream parking (timing tool of bench.py, no arithmetic)",
    "k_read_probe": "lh_hbm_read_probe: the read-only stream yardstick of bench.py, no arithmetic",
}


def product_kernels():
    names = set()
    for path in glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.hip")):
        src = open(path).read()
        for m in re.finditer(r"__global__[^;{]*?\bvoid\s+([A-Za-z_][A-Za-z_0-9]*)\s*\(", src):
            names.add(m.group(1))
    return names


def test_parser_sees_the_kernels():
    """(runs without a GPU) the source scan finds the kernels it is meant to guard."""
    ks = product_kernels()
    assert {"k_gemv_sa", "k_stream_b9", "k_stream_dma", "k_attention", "k_gemm_b9", "k_stream_q8b", "g_mul_mat"} <= ks, sorted(ks)
    assert len(ks) >= 50


@pytest.mark.gpu
def test_every_product_kernel_is_reached_by_the_gpu_suite(product, gpu_tests_passed):
    if gpu_tests_passed() < 200:
        pytest.skip(f"only {gpu_tests_passed()} GPU tests ran before this one: the route union of a partial session says nothing")
    from llama_go_amd import LIBLLAMAHIP
    lib = ctypes.CDLL(LIBLLAMAHIP)
    lib.lh_route_names.restype = ctypes.c_int64
    lib.lh_route_names.argtypes = [ctypes.c_char_p, ctypes.c_uint64]
    need = lib.lh_route_names(None, 0)
    buf = ctypes.create_string_buffer(int(need) + 16)
    lib.lh_route_names(buf, len(buf))
    reached = set(buf.value.decode().split())
    route_file = os.environ.get("LLAMAHIP_ROUTE_FILE")
    if route_file and os.path.exists(route_file):   # the worker processes of the pipeline tests (other ranks, bench.py runs)
        reached |= set(open(route_file).read().split())
    kernels = product_kernels()
    missing = sorted(kernels - reached - set(EXEMPT))
    stale = sorted(set(EXEMPT) - kernels)
    unknown = sorted(reached - kernels)
    assert not stale, f"exempt kernels that no longer exist: {stale}"
    assert not unknown, f"launch names that are not __global__ definitions of the product (LH_LAUNCH_AS family wrong?): {unknown}"
    assert not missing, f"{len(missing)} product kernels were launched by no GPU test of this session - cover them with a parity test or delete them: {missing}"
