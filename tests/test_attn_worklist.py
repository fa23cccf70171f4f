"""Host logic of the prefill attention's work decomposition (llama.go_amd/csrc/attn_worklist.h), compiled with g++ and driven through
ctypes: the list the kernel draws from must cover every (query block, part) exactly once, longest first, with the partial-record
geometry (qb_cut, pmax) the kernel and the combine pass index by.  No GPU."""
import ctypes as C
import os
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAXW = 160


@pytest.fixture(scope="module")
def wl(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("wl") / "libwl.so")
    subprocess.run(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(ROOT, "llama.go_amd", "csrc"), "-o", so,
                    os.path.join(ROOT, "tests", "attn_worklist_shim.cpp")], check=True)
    lib = C.CDLL(so)
    lib.worklist.argtypes = [C.c_uint] * 4 + [C.POINTER(C.c_uint), C.POINTER(C.c_ushort)]
    for f in (lib.steps, lib.parts, lib.part_begin):
        f.restype = C.c_uint
        f.argtypes = [C.c_uint] * 3 if f is not lib.parts else [C.c_uint] * 2
    return lib


def run(lib, n, past, H, slots=512):
    out = (C.c_uint * 6)()
    work = (C.c_ushort * MAXW)()
    nw = lib.worklist(n, past, H, slots, out, work)
    cost = struct.unpack("f", struct.pack("I", out[4]))[0]
    return dict(chunk=out[0], qb_cut=out[1], pmax=out[2], nwork=out[3], cost=cost, nqb=out[5], work=[work[i] for i in range(nw)])


def chain(lib, n, past, H, slots, entries):
    """longest workgroup chain when `entries` [(steps)] x H heads are dealt back and forth"""
    load = [0.0] * slots
    i = 0
    for st in entries:
        for _ in range(H):
            r, b = divmod(i, slots)
            load[slots - 1 - b if r & 1 else b] += st + 0.6
            i += 1
    return max(load)


CASES = [(n, past, H) for H in (8, 32, 40, 52, 64) for (n, past) in
         [(32, 0), (64, 0), (65, 0), (128, 0), (300, 0), (512, 0), (600, 0), (1023, 0), (1024, 0), (2048, 0), (4096, 0), (336, 264), (64, 1984),
          (163, 764), (33, 927), (512, 512), (1024, 1024), (256, 1792), (10240, 0), (10241, 0)]]


@pytest.mark.parametrize("n,past,H", CASES)
def test_list_covers_every_part_once_longest_first(wl, n, past, H):
    r = run(wl, n, past, H)
    nqb = r["nqb"]
    if nqb > MAXW:
        assert r["nwork"] == 0 and r["chunk"] == 0 and r["qb_cut"] == nqb and r["pmax"] == 1   # no list: the kernel's built-in order
        return
    assert 0 < r["nwork"] <= MAXW
    seen, sizes = set(), []
    for code in r["work"]:
        qb, pt = code >> 4, code & 15
        st = wl.steps(past, n, qb)
        np_ = wl.parts(st, r["chunk"])
        assert qb < nqb and pt < np_ <= 16
        assert (qb, pt) not in seen
        seen.add((qb, pt))
        sz = wl.part_begin(st, np_, pt + 1) - wl.part_begin(st, np_, pt)
        assert sz >= 1
        sizes.append(sz)
    assert sizes == sorted(sizes, reverse=True)
    want = {(qb, pt) for qb in range(nqb) for pt in range(wl.parts(wl.steps(past, n, qb), r["chunk"]))}
    assert seen == want
    # the parts of a block tile its steps
    for qb in range(nqb):
        st = wl.steps(past, n, qb)
        np_ = wl.parts(st, r["chunk"])
        assert wl.part_begin(st, np_, 0) == 0 and wl.part_begin(st, np_, np_) == st
    # geometry of the partial records
    cut = [qb for qb in range(nqb) if wl.parts(wl.steps(past, n, qb), r["chunk"]) > 1]
    assert r["qb_cut"] == (min(cut) if cut else nqb)
    assert cut == list(range(r["qb_cut"], nqb))                         # the cut blocks are the last ones: the combine grid is a range
    assert r["pmax"] == max(wl.parts(wl.steps(past, n, qb), r["chunk"]) for qb in range(nqb))
    # steps(qb) is the kernel's own count: ceil(ceil(visible keys / 32) / 2)
    for qb in range(nqb):
        qend = min((qb + 1) * 64, n)
        assert wl.steps(past, n, qb) == ((past + qend + 31) // 32 + 1) // 2


@pytest.mark.parametrize("n,past,H", CASES)
def test_choice_never_costs_more_than_uncut(wl, n, past, H):
    r = run(wl, n, past, H)
    if r["nwork"] == 0:
        return
    uncut = sorted((wl.steps(past, n, qb) for qb in range(r["nqb"])), reverse=True)
    base = chain(wl, n, past, H, 512, uncut)
    assert r["cost"] <= base + 1e-3
    got = chain(wl, n, past, H, 512, [wl.part_begin(wl.steps(past, n, c >> 4), wl.parts(wl.steps(past, n, c >> 4), r["chunk"]), (c & 15) + 1) -
                                     wl.part_begin(wl.steps(past, n, c >> 4), wl.parts(wl.steps(past, n, c >> 4), r["chunk"]), c & 15) for c in r["work"]])
    assert abs(got + (1.0 if r["chunk"] else 0.0) - r["cost"]) < 1e-2


def test_expected_decisions(wl):
    # a chunk of a long conversation: one block of 32 steps on 32 heads -> cut into many parts
    r = run(wl, 64, 1984, 32)
    assert r["chunk"] >= 4 and r["pmax"] >= 4 and r["qb_cut"] == 0 and r["cost"] < 10
    # 13B, N = 1024: the late blocks are cut in two or three, the early ones stay whole
    r = run(wl, 1024, 0, 40)
    assert r["chunk"] != 0 and 0 < r["qb_cut"] < 16 and r["pmax"] in (2, 3) and r["cost"] < 16.6
    # many more items than workgroups: nothing to gain
    r = run(wl, 4096, 0, 32)
    assert r["chunk"] == 0 and r["pmax"] == 1 and r["qb_cut"] == r["nqb"]
    # short prompts: one or two blocks of one or two steps
    r = run(wl, 100, 0, 32)
    assert r["chunk"] == 0 and r["nwork"] == 2 and r["work"] == [1 << 4, 0]


@pytest.mark.parametrize("n,past,H", [(64, 960, 8), (163, 764, 8), (200, 0, 4), (130, 520, 2)])
def test_parts_and_combine_reproduce_softmax_attention(wl, n, past, H):
    """The arithmetic the decomposition relies on, emulated in numpy with the real work list: every part keeps an un-normalised
    (O, m, l) per query over its key range (a query that sees none of the part's keys: O = 0, l = 0, m = -inf), the combine pass
    adds the parts in part order with weights e^(m_p - M); an uncut block normalises directly.  Equal to causal softmax attention
    computed in float64 over all keys (kernels_attn.h k_attn_flash / k_attn_flash_combine)."""
    import numpy as np
    r = run(wl, n, past, H, slots=512)
    rng = np.random.default_rng(n + past)
    hd, T = 16, past + n
    q = rng.standard_normal((n, hd)).astype(np.float32)
    k = rng.standard_normal((T, hd)).astype(np.float32)
    v = rng.standard_normal((T, hd)).astype(np.float32)
    scale = np.float32(1.0 / np.sqrt(hd))
    # reference: float64, key t visible to query j iff t <= past + j
    s = (q.astype(np.float64) @ k.astype(np.float64).T) * float(scale)
    mask = np.arange(T)[None, :] <= (past + np.arange(n))[:, None]
    s = np.where(mask, s, -np.inf)
    p = np.exp(s - s.max(axis=1, keepdims=True))
    want = (p / p.sum(axis=1, keepdims=True)) @ v.astype(np.float64)
    got = np.full((n, hd), np.nan, dtype=np.float32)
    recs = {}
    for code in r["work"]:
        qb, pt = code >> 4, code & 15
        st = wl.steps(past, n, qb)
        np_ = wl.parts(st, r["chunk"])
        k0, k1 = 64 * wl.part_begin(st, np_, pt), min(64 * wl.part_begin(st, np_, pt + 1), T)
        rows = np.arange(qb * 64, min(qb * 64 + 64, n))
        ss = (q[rows] @ k[k0:k1].T).astype(np.float32) * scale
        vis = np.arange(k0, k1)[None, :] <= (past + rows)[:, None]
        ss = np.where(vis, ss, -np.inf).astype(np.float32)
        m = ss.max(axis=1)
        with np.errstate(invalid="ignore"):
            pp = np.where(vis, np.exp((ss - np.where(np.isinf(m), 0, m)[:, None]).astype(np.float32)), 0).astype(np.float32)
        l = pp.sum(axis=1, dtype=np.float32)
        o = (pp @ v[k0:k1]).astype(np.float32)
        if np_ == 1:
            got[rows] = o / l[:, None]
        else:
            recs[(qb, pt)] = (o, m, l, np_, rows)
    for qb in sorted({qb for qb, _ in recs}):
        np_ = recs[(qb, 0)][3]
        rows = recs[(qb, 0)][4]
        M = np.max(np.stack([recs[(qb, pt)][1] for pt in range(np_)]), axis=0)
        assert np.all(np.isfinite(M))                      # part 0 holds key 0
        acc = np.zeros((len(rows), hd), np.float32)
        den = np.zeros(len(rows), np.float32)
        for pt in range(np_):
            o, m, l, _, _ = recs[(qb, pt)]
            w = np.where(np.isinf(m), 0, np.exp((m - M).astype(np.float32))).astype(np.float32)
            acc += o * w[:, None]
            den += l * w
        got[rows] = acc / den[:, None]
    assert not np.isnan(got).any()
    assert np.abs(got - want).max() / np.abs(want).max() < 2e-6
    if (n, past) in ((64, 960), (163, 764), (130, 520)):
        assert r["chunk"] != 0 and r["pmax"] >= 2          # these shapes are cut: the combine path above ran
