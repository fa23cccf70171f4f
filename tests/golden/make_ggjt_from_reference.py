"""Writes tests/golden/ref_ggjt_f32.bin and ref_ggjt_f16_2parts.bin WITH THE REFERENCE'S OWN CONVERTER.

    python tests/golden/make_ggjt_from_reference.py          (needs /root/reference: run in the build container only)

The one importable piece of the reference that emits bytes the product consumes is scripts/convert-pth-to-ggml.py.  It is
imported here by path (never copied) and its write_header (:109-119), write_tokens (:121-138) and
process_and_write_variables (:140-232) are called on
  - a tiny synthetic LLaMA state dict (tests/golden/ggjt_fixture.py; float16 like Meta's checkpoints, nn.Linear layout), and
  - a sentencepiece tokenizer trained here with byte fallback (so write_tokens takes its unknown / control / byte / piece
    branches); the trained tokenizer.model is committed next to the fixtures.
Two files, the same weights:
  ref_ggjt_f32.bin           ftype 0, one part              (all tensors widened to f32 by the converter)
  ref_ggjt_f16_2parts.bin    ftype 1, TWO model-parallel parts reassembled by the converter's own seek arithmetic
                             (:209-229; split_dim 0 rows / split_dim 1 columns), matrices stay f16, norms f32
These files, not our own writer, are what tests/test_ggjt_reference_fixture.py (CPU: oracle loader) and
tests/test_gpu_llama.py (GPU: product loader) read.  Provenance is recorded in ref_ggjt_manifest.json (sha256 of the files and
of the converter script that produced them).
"""
import hashlib
import importlib.util
import io
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ggjt_fixture as fx  # noqa: E402

REF_SCRIPT = "/root/reference/scripts/convert-pth-to-ggml.py"


def load_converter():
    spec = importlib.util.spec_from_file_location("ref_convert_pth_to_ggml", REF_SCRIPT)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def train_tokenizer(path_prefix):
    import sentencepiece as spm
    corpus = os.path.join(HERE, "_spm_corpus.txt")
    words = ["the", "go", "llama", "tensor", "graph", "compute", "token", "matrix", "vector", "rope", "norm", "cache", "layer", "head", "eval"]
    rng = np.random.RandomState(7)
    with open(corpus, "w") as f:
        for _ in range(400):
            f.write(" ".join(words[i] for i in rng.randint(0, len(words), size=8)) + "\n")
    spm.SentencePieceTrainer.train(input=corpus, model_prefix=path_prefix, vocab_size=fx.VOCAB_SIZE, model_type="bpe", byte_fallback=True,
                                   character_coverage=1.0, unk_id=0, bos_id=1, eos_id=2, pad_id=-1, num_threads=1, minloglevel=2)
    os.remove(corpus)
    os.remove(path_prefix + ".vocab")


def main():
    conv = load_converter()
    tok_path = os.path.join(HERE, "ref_tokenizer")
    if not os.path.exists(tok_path + ".model"):
        train_tokenizer(tok_path)
    tokenizer = conv.SentencePieceProcessor(tok_path + ".model")
    assert tokenizer.vocab_size() == fx.VOCAB_SIZE, tokenizer.vocab_size()
    hparams = dict(fx.PARAMS)
    hparams.update({"vocab_size": tokenizer.vocab_size()})   # as load_hparams_and_tokenizer does (:106)
    sd = fx.state_dict()
    manifest = {"converter": REF_SCRIPT, "converter_sha256": hashlib.sha256(open(REF_SCRIPT, "rb").read()).hexdigest(),
                "tokenizer_sha256": hashlib.sha256(open(tok_path + ".model", "rb").read()).hexdigest(), "hparams": hparams, "files": {}}
    stdout = sys.stdout
    for fname, ftype, n_parts in (("ref_ggjt_f32.bin", 0, 1), ("ref_ggjt_f16_2parts.bin", 1, 2)):
        out = os.path.join(HERE, fname)
        sys.stdout = io.StringIO()  # the converter prints one line per tensor
        try:
            with open(out, "wb") as fout:   # same call sequence as the converter's main() (:262-275)
                conv.write_header(fout, hparams, ftype)
                conv.write_tokens(fout, tokenizer)
                offset_of_tensors = fout.tell()
                for part_id in range(n_parts):
                    fout.seek(offset_of_tensors)
                    model = {k: torch.from_numpy(v.copy()) for k, v in fx.shard(sd, part_id, n_parts).items()}
                    conv.process_and_write_variables(fout, model, ftype, part_id, n_parts)
        finally:
            sys.stdout = stdout
        manifest["files"][fname] = {"ftype": ftype, "n_parts": n_parts, "bytes": os.path.getsize(out), "sha256": hashlib.sha256(open(out, "rb").read()).hexdigest()}
        print(f"wrote {out}: {os.path.getsize(out)} bytes")
    json.dump(manifest, open(os.path.join(HERE, "ref_ggjt_manifest.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
