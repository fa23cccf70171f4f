"""not-gpu: the C-ABI library loads and exports every symbol include/*.h declares; host-side logic that needs no GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b((?:lh|ml|llama|llamago)_[A-Za-z0-9_]+)\s*\(", src)))


def test_libllamahip_exports_every_declared_symbol(built):
    import llama_go_amd as pkg
    lib = C.CDLL(pkg.LIBLLAMAHIP)
    names = declared_functions("llamahip.h")
    assert len(names) >= 20
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"libllamahip.so lacks {missing}"
    lib.lh_abi_version.restype = C.c_int
    assert lib.lh_abi_version() == 1


def exported(path, prefix_re):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    names = [ln.split()[-1] for ln in out.splitlines() if len(ln.split()) >= 3 and ln.split()[-2] in ("T", "W")]
    return sorted(n for n in names if re.match(prefix_re, n))


def test_headers_and_exports_agree_in_both_directions(built):
    """Every C symbol the libraries export is declared in a header under include/ and vice versa: the hand-written ctypes / cgo
    bindings are written against the headers, so an export without a declaration (or a stale declaration) is a drift nobody sees."""
    import llama_go_amd as pkg
    hip_decl = set(declared_functions("llamahip.h"))
    hip_exp = set(exported(pkg.LIBLLAMAHIP, r"lh_"))
    assert hip_exp == hip_decl, (sorted(hip_exp - hip_decl), sorted(hip_decl - hip_exp))
    go_decl = set(declared_functions("llamago.h")) | set(n for n in declared_functions("llamago_ext.h") if n.startswith("llamago_"))
    go_exp = set(exported(pkg.LIBLLAMAGO, r"(ml|llama|llamago)_"))
    assert go_exp == go_decl, (sorted(go_exp - go_decl), sorted(go_decl - go_exp))
    # the checker exports the mirror API and the [both] part of the extensions; nothing of it is undeclared
    orc_exp = set(exported(os.path.join(ROOT, "oracle", "liboracle.so"), r"(ml|llama|llamago)_"))
    assert orc_exp <= go_decl, sorted(orc_exp - go_decl)
    src = open(os.path.join(ROOT, "include", "llamago_ext.h")).read()
    both = set(re.findall(r"\b(llamago_[A-Za-z0-9_]+)\s*\(", src.split("[product] device plumbing")[0].split("#define LLAMAGO_EXT_H")[1]))
    assert both and both <= orc_exp, sorted(both - orc_exp)
    # mlapi.py binds only declared names
    py = open(os.path.join(ROOT, "llama.go_amd", "mlapi.py")).read()
    bound = set(re.findall(r"(?:lib|L)\.((?:ml|llama|llamago)_[A-Za-z0-9_]+)\b", py)) | set(re.findall(r"sig\(\"((?:ml|llama|llamago)_[A-Za-z0-9_]+)\"", py))
    assert bound <= go_decl, sorted(bound - go_decl)


def _split_args(text):
    """top-level comma split of an argument list (nested parentheses / brackets / braces stay together)"""
    out, depth, cur = [], 0, ""
    for ch in text:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def _call_args(src, start):
    """the text between the parentheses of the call whose '(' is at src[start]"""
    depth, i = 0, start
    while True:
        if src[i] == "(":
            depth += 1
        elif src[i] == ")":
            depth -= 1
            if depth == 0:
                return src[start + 1:i]
        i += 1


GO_FILES = ("ml_hip.go", "ml_hip_pods.go")


def _go_code(name):
    go = open(os.path.join(ROOT, "llama.go_amd", "go", name)).read()
    go = re.sub(r"//[^\n]*", "", go)
    return go.split('import "C"', 1)[1]          # below the cgo preamble


def _c_header():
    return re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "llamahip.h")).read(), flags=re.S)


def _norm_ctype(t):
    """'const float * const *x' -> ('float', 2): base type and pointer depth, qualifiers and the parameter name dropped."""
    t = re.sub(r"\b(const|struct|volatile)\b", " ", t)
    arrays = len(re.findall(r"\[[^\]]*\]", t))          # `uint8_t id[128]` is a pointer parameter
    t = re.sub(r"\[[^\]]*\]", " ", t)
    depth = t.count("*") + arrays
    t = t.replace("*", " ")
    words = t.split()
    if len(words) > 1 and not (len(words) == 2 and words[0] in ("unsigned", "long")):
        words = words[:-1]                      # the parameter name
    return " ".join(words), depth


def _prototypes(hdr):
    protos = {}
    for m in re.finditer(r"\b(lh_[A-Za-z0-9_]+)\s*\(", hdr):
        args = _call_args(hdr, m.end() - 1).strip()
        protos[m.group(1)] = [] if args in ("", "void") else [_norm_ctype(a) for a in _split_args(args)]
    return protos


def _go_arg_ctype(expr):
    """The C type a cgo argument expression visibly has, or None when it is a plain Go variable (whose declaration is checked elsewhere)."""
    e = expr.strip()
    m = re.match(r"^\(\s*(\*+)\s*C\.([A-Za-z0-9_]+)\s*\)\s*\(", e)       # (*C.float)(unsafe.Pointer(...))
    if m:
        return m.group(2), len(m.group(1))
    m = re.match(r"^C\.([A-Za-z0-9_]+)\s*\(", e)                             # C.uint32_t(x)
    if m:
        return m.group(1), 0
    if e.startswith("unsafe.Pointer("):
        return "void", 1
    return None


CGO_NAMES = {"int": "int", "uint": "unsigned", "float": "float", "double": "double", "char": "char", "size_t": "size_t"}


def test_go_shim_calls_only_declared_functions_with_the_declared_arity():
    """The cgo shim cannot be compiled here (no Go toolchain).  What can be checked without one: every C.lh_* it calls is a function
    llamahip.h declares, called with the number of arguments the prototype has; every argument whose C type is visible in the call (an
    explicit conversion C.T(x), a pointer cast (*C.T)(unsafe.Pointer(..)), unsafe.Pointer(..)) has the prototype's base type and pointer depth;
    every C.lh_* / C.LH_* type or constant it names exists; every field of a C struct it writes (composite literals, the lh_tensor records of
    hipGraphCompute) is a field of that struct; ml_hip.go stays the GraphCompute path (pods and pipelines live in ml_hip_pods.go)."""
    hdr = _c_header()
    protos = _prototypes(hdr)
    structs = {m.group(1): set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*(?:\[[^\]]*\])?\s*[,;]", m.group(2)))
               for m in re.finditer(r"typedef struct (lh_[a-z_]+) \{(.*?)\} \1;", hdr, flags=re.S)}
    calls = typed = 0
    for fname in GO_FILES:
        go_code = _go_code(fname)
        for m in re.finditer(r"\bC\.(lh_[A-Za-z0-9_]+)\s*\(", go_code):
            name = m.group(1)
            if name not in protos:
                # a conversion to a C type, e.g. C.lh_buf(x): the type must exist
                assert re.search(r"\b" + name + r"\b", hdr), f"{fname} uses C.{name}, which llamahip.h does not know"
                continue
            args = _call_args(go_code, m.end() - 1).strip()
            got = [] if not args else _split_args(args)
            assert len(got) == len(protos[name]), f"{fname} calls C.{name} with {len(got)} arguments, llamahip.h declares {len(protos[name])}"
            for k, (expr, (base, depth)) in enumerate(zip(got, protos[name])):
                if expr.strip() == "nil":
                    assert depth >= 1, f"{fname}: C.{name} argument {k} is nil but the parameter is {base}"
                    continue
                vis = _go_arg_ctype(expr)
                if vis is None:
                    continue
                gbase, gdepth = CGO_NAMES.get(vis[0], vis[0]), vis[1]
                assert (gbase, gdepth) == (base, depth), f"{fname}: C.{name} argument {k} is {gbase}{'*' * gdepth}, llamahip.h declares {base}{'*' * depth}"
                typed += 1
            calls += 1
        for name in set(re.findall(r"\bC\.((?:lh|LH)_[A-Za-z0-9_]+)\b", go_code)):
            assert re.search(r"\b" + name + r"\b", hdr), f"{fname} names C.{name}, which llamahip.h does not declare"
        for m in re.finditer(r"\bC\.(lh_[a-z_]+)\{([^}]*)\}", go_code):           # composite literals
            for field in re.findall(r"\b([a-z_0-9]+)\s*:", m.group(2)):
                assert field in structs[m.group(1)], f"{fname}: {m.group(1)} has no field {field}"
    assert calls >= 30 and typed >= 25, (calls, typed)
    contract = _go_code("ml_hip.go")
    fields = set(re.findall(r"\bo\.([a-z_0-9]+)", contract))                         # o := &arr[i], a *C.lh_tensor
    assert fields and fields <= structs["lh_tensor"], fields - structs["lh_tensor"]
    assert len(open(os.path.join(ROOT, "llama.go_amd", "go", "ml_hip.go")).read().splitlines()) <= 280
    assert not re.search(r"\bC\.lh_(batch|pipeline|comm|llama)_", contract), "ml_hip.go is the GraphCompute path only"


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


def test_bench_ranks_fail_fast_without_a_gpu(built):
    """`bench.py --gpus 2` where no rank can work (no GPU here): every rank says {"error": ...} and exits non-zero, the self-spawning
    parent takes the siblings down and returns - within seconds, not after a rendezvous timeout (a crashed rank must not
    hold the others)."""
    import json
    import subprocess
    import sys
    import time
    C.CDLL(os.path.join(ROOT, "llama.go_amd", "lib", "libllamahip.so"), mode=C.RTLD_GLOBAL).lh_device_count.restype = C.c_int
    import llama_go_amd as pkg
    lib = C.CDLL(pkg.LIBLLAMAHIP)
    lib.lh_device_count.restype = C.c_int
    if lib.lh_device_count() > 0:
        pytest.skip("a GPU is visible here")
    env = dict(os.environ, BENCH_CONTROL_TIMEOUT_S="30")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--shape", "tiny", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode != 0
    assert time.time() - t0 < 200
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert lines and "error" in json.loads(lines[-1]), r.stdout[-500:]
