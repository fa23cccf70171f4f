"""not-gpu: the C-ABI library loads and exports every symbol include/*.h declares; host-side logic that needs no GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b((?:lh|ml|llama)_[A-Za-z0-9_]+)\s*\(", src)))


def test_libllamahip_exports_every_declared_symbol(built):
    import llama_go_amd as pkg
    lib = C.CDLL(pkg.LIBLLAMAHIP)
    names = declared_functions("llamahip.h")
    assert len(names) >= 20
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"libllamahip.so lacks {missing}"
    lib.lh_abi_version.restype = C.c_int
    assert lib.lh_abi_version() == 1


def test_host_library_and_oracle_export_the_mirror_api(built):
    import llama_go_amd as pkg
    C.CDLL(pkg.LIBLLAMAHIP, mode=C.RTLD_GLOBAL)
    names = declared_functions("llamago.h")
    assert len(names) >= 40
    for path in (pkg.LIBLLAMAGO, os.path.join(ROOT, "oracle", "liboracle.so")):
        lib = C.CDLL(path)
        missing = [n for n in names if not hasattr(lib, n)]
        assert not missing, f"{path} lacks {missing}"


def test_no_gpu_means_loud_failure_not_a_cpu_fallback(built):
    """Without a HIP device the product refuses to create a context (no silent CPU path exists)."""
    import llama_go_amd as pkg
    lib = C.CDLL(pkg.LIBLLAMAHIP, mode=C.RTLD_GLOBAL)
    lib.lh_device_count.restype = C.c_int
    if lib.lh_device_count() > 0:
        pytest.skip("a GPU is visible here")
    ctx = C.c_void_p()
    lib.lh_ctx_create.restype = C.c_int
    rc = lib.lh_ctx_create(0, None, C.byref(ctx))
    assert rc == -5  # LH_ENODEVICE
    lib.lh_last_error.restype = C.c_char_p
    assert b"no HIP device" in lib.lh_last_error(None)
    from llama_go_amd.mlapi import load_product, make_hparams, SHAPES, MLError
    prod = load_product()
    with pytest.raises(MLError):
        prod.NewSyntheticModel(make_hparams(**SHAPES["tiny"]), 1)


def test_product_never_references_the_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline leg may touch oracle/ (the oracle is the checker, never the product)."""
    pkg = os.path.join(ROOT, "llama.go_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".go", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in txt.lower(), f"{f} references the oracle"


def test_host_mirror_builds_reference_graph_shapes(built):
    """Graph construction is pure host code: shapes/strides of the reference's builders (ml.go:295-318, 786-845, 601-617)."""
    from llama_go_amd.mlapi import load_product
    import numpy as np
    ml = load_product()
    # constructors that need no device context: pass ctx = None
    a = ml.NewTensor(None, (128, 7, 3))
    b = ml.NewTensor(None, (128, 5, 3))
    mm = ml.MulMat(None, a, b)
    assert ml.shape(mm)[0] == (7, 5, 3, 1)
    p = ml.Permute(None, a, 0, 2, 1, 3)
    ne, nb = ml.shape(p)
    assert ne == (128, 3, 7, 1) and nb == (4, 128 * 7 * 4, 128 * 4, 128 * 7 * 3 * 4)
    v = ml.View1D(None, a, 256, 128)
    assert ml.shape(v)[0] == (256, 1, 1, 1)
    with pytest.raises(Exception):
        ml.View1D(None, a, 128 * 7 * 3, 1)  # past the end of the backing array (Go would panic on the slice)
    g = ml.NewGraph()
    ml.BuildForwardExpand(g, ml.SoftMax(None, ml.Scale(None, mm, ml.NewFP32(None, 0.5))))
    assert ml.graph_ops(g) == ["MUL_MAT", "SCALE", "SOFT_MAX"]
    ml.FreeGraph(g)
