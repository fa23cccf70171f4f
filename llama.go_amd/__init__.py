"""llama.go_amd — MI355X-native backend for the llama.go hot path (pkg/ml GraphCompute + pkg/llama Eval).

Layout:
  csrc/        hand-written gfx950 HIP kernels + the C-ABI boundary (include/llamahip.h) -> lib/libllamahip.so
  host/        C++ mirror of the reference's pkg/ml + pkg/llama operator surface (include/llamago.h),
               calling the C-ABI once per GraphCompute -> lib/libllamago.so
  go/          the cgo shim a llama.go maintainer drops into pkg/ml (cannot be compiled here: no Go toolchain)
  mlapi.py     ctypes binding of include/llamago.h (drives the product library; any other library exporting the
               same API can be wrapped by the same class)
  pipeline.py  layer-shard pipeline schedule over torch.distributed (RCCL send/recv of the residual stream)

The product path has no CPU fallback: loading fails loudly if the HIP library is missing.
"""
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(PKG_DIR)
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIBLLAMAHIP = os.path.join(LIB_DIR, "libllamahip.so")
LIBLLAMAGO = os.path.join(LIB_DIR, "libllamago.so")

from .mlapi import MLLib, load_product  # noqa: E402,F401
