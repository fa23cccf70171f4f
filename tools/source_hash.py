"""sha256 over the kernel sources of libllamahip.so (llama.go_amd/csrc/*.h, *.hip and include/llamahip.h, names and bytes, sorted): the stamp that ties
profiles/pmc_traffic.json to the build it was measured on (bench.py drops `roofline.traffic` when it differs).  usage: python tools/source_hash.py"""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_hash():
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "llama.go_amd", "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "llama.go_amd", "csrc", "*.hip"))) + [os.path.join(ROOT, "include", "llamahip.h")]
    for f in files:
        h.update(os.path.relpath(f, ROOT).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    print(kernel_source_hash())
