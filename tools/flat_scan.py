"""Which kernels of the built library still address memory through FLAT instructions (flat_load / flat_store: the address space was lost in a pointer
select or a round trip through an integer; such loads count on the LDS wait counter too) or through scratch.  usage: python tools/flat_scan.py [lib]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "llama.go_amd", "lib", "libllamahip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
counts = collections.OrderedDict()
with tempfile.TemporaryDirectory() as td:
    subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={td}/fat.bin", lib], check=True)
    fat = open(f"{td}/fat.bin", "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), fat)]
    for i, st in enumerate(starts):
        en = starts[i + 1] if i + 1 < len(starts) else len(fat)
        open(f"{td}/b{i}.bin", "wb").write(fat[st:en])
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={td}/b{i}.bin",
                            f"--output={td}/d{i}.co"], capture_output=True, text=True)
        if r.returncode or not os.path.exists(f"{td}/d{i}.co") or os.path.getsize(f"{td}/d{i}.co") == 0:
            continue
        dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", f"{td}/d{i}.co"], capture_output=True, text=True).stdout
        cur = None
        for line in dis.split("\n"):
            m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
            if m:
                cur = m.group(1)
                counts.setdefault(cur, [0, 0, 0])
                continue
            if cur is None:
                continue
            if "flat_load" in line:
                counts[cur][0] += 1
            if "flat_store" in line or "flat_atomic" in line:
                counts[cur][1] += 1
            if "scratch_" in line:
                counts[cur][2] += 1
names = list(counts)
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
tot = 0
for n, d in zip(names, dem):
    fl, fs, sc = counts[n]
    if fl or fs or sc:
        tot += 1
        print(f"{fl:4d} flat_load {fs:3d} flat_store {sc:3d} scratch  {d[:170]}")
print(len(counts), "kernels,", tot, "with flat or scratch accesses")
