"""not-gpu: the ggjt loader against files WRITTEN BY THE REFERENCE's own converter (scripts/convert-pth-to-ggml.py, imported by
tests/golden/make_ggjt_from_reference.py; provenance in tests/golden/ref_ggjt_manifest.json) — not by our writer.

  ref_ggjt_f32.bin          ftype 0, one checkpoint part
  ref_ggjt_f16_2parts.bin   ftype 1, two model-parallel parts reassembled by the converter (split rows / split columns)

Checked here on the CPU side: (1) the committed files are the ones the manifest names; (2) an independent struct-level reader
(this file) finds the layout llama.go:712-976 expects (magic, version, 7 header ints, vocab records, dims reversed, 32-byte
aligned data); (3) the checker's loader (oracle llama_LoadModel) returns exactly the weights of the fixture state dict, rebuilt
here in numpy without torch; (4) its Eval on the loaded model agrees with the independent numpy float64 forward.
tests/test_gpu_llama.py repeats (3)+(4) for the PRODUCT loader (file -> bounce buffer -> HBM)."""
import hashlib
import json
import os
import struct
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
sys.path.insert(0, GOLDEN)
import ggjt_fixture as fx  # noqa: E402

from test_oracle import np_eval, np_weights  # noqa: E402

FILES = ["ref_ggjt_f32.bin", "ref_ggjt_f16_2parts.bin"]


def fixture_weights():
    """name -> float32 [out, in] (the f16 checkpoint values, exactly representable)."""
    return {k: v.astype(np.float32) for k, v in fx.state_dict().items() if not k.endswith("freqs")}


def test_committed_files_are_the_ones_the_reference_converter_wrote():
    man = json.load(open(os.path.join(GOLDEN, "ref_ggjt_manifest.json")))
    assert man["converter"].endswith("scripts/convert-pth-to-ggml.py")
    for f in FILES:
        data = open(os.path.join(GOLDEN, f), "rb").read()
        assert len(data) == man["files"][f]["bytes"]
        assert hashlib.sha256(data).hexdigest() == man["files"][f]["sha256"]
    ref = "/root/reference/scripts/convert-pth-to-ggml.py"
    if os.path.exists(ref):  # build container: the script that is there now is the one that produced the fixtures
        assert hashlib.sha256(open(ref, "rb").read()).hexdigest() == man["converter_sha256"]


@pytest.mark.parametrize("fname", FILES)
def test_struct_level_layout(fname):
    data = open(os.path.join(GOLDEN, fname), "rb").read()
    magic, version, vocab, dim, mult, heads, layers, rot, ftype = struct.unpack_from("<9i", data, 0)
    assert magic == 0x67676A74 and version == 1                      # llama.go:722-739
    assert (vocab, dim, mult, heads, layers) == (fx.VOCAB_SIZE, 128, 64, 4, 2)
    assert rot == dim // heads and ftype == (0 if "f32" in fname else 1)
    off = 36
    kinds = {"empty": 0, "byte": 0, "text": 0}
    for _ in range(vocab):                                           # llama.go:799-811
        (n,) = struct.unpack_from("<i", data, off)
        off += 4
        kinds["empty" if n == 0 else "byte" if n == 1 else "text"] += 1
        off += n + 4
    assert kinds["empty"] == 2 and kinds["byte"] >= 256              # bos/eos control pieces; byte-fallback pieces (+ 1-char pieces)
    want = fixture_weights()
    seen = {}
    while off < len(data):                                           # llama.go:889-969
        n_dims, name_len, dtype = struct.unpack_from("<3i", data, off)
        off += 12
        assert n_dims in (1, 2) and dtype in (0, 1)
        ne = struct.unpack_from(f"<{n_dims}i", data, off)
        off += 4 * n_dims
        name = data[off: off + name_len].decode()
        off += name_len
        off = (off + 31) & ~31                                       # llama.go:926-933
        n = int(np.prod(ne))
        w = want[name]
        assert tuple(reversed(w.shape)) == tuple(ne), name           # dims are written reversed: ne[0] = in-features
        if dtype == 1:
            assert n_dims == 2 and ftype == 1
            arr = np.frombuffer(data, np.float16, n, off).astype(np.float32)
            off += 2 * n
        else:
            arr = np.frombuffer(data, np.float32, n, off)
            off += 4 * n
        assert np.array_equal(arr.reshape(w.shape), w), name         # row-major [out][in], also after the 2-part reassembly
        seen[name] = dtype
    assert set(seen) == set(want) and "rope.freqs" not in seen
    if ftype == 1:
        assert all((dt == 1) == (want[n].ndim == 2) for n, dt in seen.items())  # norms stay f32 (convert-pth-to-ggml.py:153-157)


def _check_loaded_model(lib, reader_ctx=None):
    want = fixture_weights()
    logits = []
    for fname in FILES:
        m = lib.LoadModel(os.path.join(GOLDEN, fname), 32)
        hp = m.hp
        assert (hp.vocabSize, hp.embdSize, hp.multSize, hp.headsCount, hp.layersCount) == (fx.VOCAB_SIZE, 128, 64, 4, 2)
        assert m.ffSize == fx.ff_size(128, 64)
        for name, w in want.items():
            t = m.tensor(name)
            assert t, name
            ne, _ = lib.shape(t)
            got = lib.read(reader_ctx, t).reshape(-1)
            assert tuple(ne[:w.ndim]) == tuple(reversed(w.shape)), name
            assert np.array_equal(got, w.reshape(-1)), name
        logits.append((m, hp))
    return logits


def test_oracle_loader_reads_the_reference_files(oracle):
    models = _check_loaded_model(oracle)
    prompt = [1, 70, 261, 5, 280, 33, 9]
    outs = []
    for m, hp in models:
        W = np_weights(oracle, m)
        c = m.NewContext(32, 2, False)
        L, d, H = hp.layersCount, hp.embdSize, hp.headsCount
        kc = np.zeros((L, 32, H, d // H))
        vc = np.zeros((L, 32, H, d // H))
        got = c.Eval(prompt, 0)
        want = np_eval(W, hp, prompt, 0, kc, vc)
        assert np.abs(got - want).max() / np.abs(want).max() < 2e-6
        tok = int(np.argmax(got))
        assert tok == int(np.argmax(want))
        got2 = c.Eval([tok], len(prompt))
        want2 = np_eval(W, hp, [tok], len(prompt), kc, vc)
        assert np.abs(got2 - want2).max() / np.abs(want2).max() < 2e-6
        outs.append((got, got2))
        c.free()
        m.free()
    # the f32 file and the reassembled two-part f16 file hold the same weights: identical logits
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
