"""Kernel metadata of a built HIP library: register counts, LDS, scratch / spills per kernel (reads the code object's msgpack notes through
llvm-readelf).  usage: python tools/isa_check.py [lib] [--filter substr]   exit code 1 if any kernel has scratch or spills"""
import re
import subprocess
import sys
import tempfile
import os

lib = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "llama.go_amd", "lib", "libllamahip.so")
flt = sys.argv[sys.argv.index("--filter") + 1] if "--filter" in sys.argv else ""
LLVM = "/opt/rocm/lib/llvm/bin"
notes = ""
with tempfile.TemporaryDirectory() as td:
    # a shared library carries one fat binary per translation unit, back to back in .hip_fatbin
    subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={td}/fat.bin", lib], check=True)
    fat = open(f"{td}/fat.bin", "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), fat)]
    for i, st in enumerate(starts):
        en = starts[i + 1] if i + 1 < len(starts) else len(fat)
        open(f"{td}/b{i}.bin", "wb").write(fat[st:en])
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={td}/b{i}.bin", f"--output={td}/d{i}.co"],
                           capture_output=True, text=True)
        if r.returncode or not os.path.exists(f"{td}/d{i}.co") or os.path.getsize(f"{td}/d{i}.co") == 0:
            continue
        notes += subprocess.run([f"{LLVM}/llvm-readelf", "--notes", f"{td}/d{i}.co"], check=True, capture_output=True, text=True).stdout
kernels = []
for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    kernels.append(dict(name=g("name"), vgpr=g("vgpr_count"), sgpr=g("sgpr_count"), lds=g("group_segment_fixed_size"), scratch=g("private_segment_fixed_size"),
                        vspill=g("vgpr_spill_count"), sspill=g("sgpr_spill_count")))
bad = [k for k in kernels if k["scratch"] not in ("0", "?") or k["vspill"] not in ("0", "?")]
for k in kernels:
    if flt and flt not in k["name"]:
        continue
    print(f'{k["name"][:90]:90s} vgpr {k["vgpr"]:>4s} sgpr {k["sgpr"]:>4s} scratch {k["scratch"]:>5s} vgpr_spill {k["vspill"]:>3s} sgpr_spill {k["sspill"]:>3s}')
print(f"{len(kernels)} kernels, {len(bad)} with scratch or VGPR spills")
sys.exit(1 if bad else 0)
