"""llama.go_amd — MI355X-native backend for the llama.go hot path (pkg/ml GraphCompute + pkg/llama Eval).

Layout:
  csrc/        hand-written gfx950 HIP kernels + the C-ABI boundary (include/llamahip.h) -> lib/libllamahip.so
  host/        C++ mirror of the reference's pkg/ml + pkg/llama operator surface (include/llamago.h),
               calling the C-ABI once per GraphCompute -> lib/libllamago.so
  go/          the cgo shim a llama.go maintainer drops into pkg/ml (cannot be compiled here: no Go toolchain)
  mlapi.py     ctypes binding of include/llamago.h (drives the product library; any other library exporting the
               same API can be wrapped by the same class)
  pipeline.py  ctypes view of the C pipeline scheduler (lh_pipeline_schedule / run_hooks) + a gloo transport hook for CPU and shared-GPU tests;
               the schedule, the stages and the RCCL send/recv all live below the C-ABI (csrc/comm.hip)

The product path has no CPU fallback: loading fails loudly if the HIP library is missing.
"""
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(PKG_DIR)
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIBLLAMAHIP = os.path.join(LIB_DIR, "libllamahip.so")
LIBLLAMAGO = os.path.join(LIB_DIR, "libllamago.so")

from .mlapi import MLLib, load_product  # noqa: E402,F401
