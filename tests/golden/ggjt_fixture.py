"""Shared description of the ggjt fixture model (tests/golden/ref_ggjt_*.bin): hyper-parameters and the deterministic
state dict, in numpy only, so that tests can rebuild the weights WITHOUT torch / the reference and compare them with what the
loaders read out of the files the reference's converter wrote (make_ggjt_from_reference.py)."""
import numpy as np

PARAMS = {"dim": 128, "multiple_of": 64, "n_heads": 4, "n_layers": 2, "norm_eps": 1e-05, "vocab_size": -1}  # params.json style (Meta)
VOCAB_SIZE = 288          # tokenizer: 3 control pieces + 256 byte-fallback pieces + 29 learned pieces
SEED = 20240923


def ff_size(dim, mult):   # llama.go:761
    return ((2 * (4 * dim) // 3 + mult - 1) // mult) * mult


def state_dict():
    """name -> float16 ndarray in PyTorch nn.Linear layout [out, in], in the order of a Meta consolidated.00.pth
    (matrices first, norms after them per layer, rope.freqs last: the converter must skip that one)."""
    d, L, V = PARAMS["dim"], PARAMS["n_layers"], VOCAB_SIZE
    F = ff_size(d, PARAMS["multiple_of"])
    rng = np.random.RandomState(SEED)

    def mat(rows, cols, scale):
        return (rng.uniform(-1.0, 1.0, size=(rows, cols)) * scale).astype(np.float16)

    def vec(n):
        return (1.0 + 0.1 * rng.uniform(-1.0, 1.0, size=(n,))).astype(np.float16)

    sd = {}
    sd["tok_embeddings.weight"] = mat(V, d, np.sqrt(3.0))
    sd["norm.weight"] = vec(d)
    sd["output.weight"] = mat(V, d, np.sqrt(3.0 / d))
    for i in range(L):
        p = f"layers.{i}."
        sd[p + "attention.wq.weight"] = mat(d, d, np.sqrt(3.0 / d))
        sd[p + "attention.wk.weight"] = mat(d, d, np.sqrt(3.0 / d))
        sd[p + "attention.wv.weight"] = mat(d, d, np.sqrt(3.0 / d))
        sd[p + "attention.wo.weight"] = mat(d, d, np.sqrt(3.0 / d))
        sd[p + "feed_forward.w1.weight"] = mat(F, d, np.sqrt(3.0 / d))
        sd[p + "feed_forward.w2.weight"] = mat(d, F, np.sqrt(3.0 / F))
        sd[p + "feed_forward.w3.weight"] = mat(F, d, np.sqrt(3.0 / d))
        sd[p + "attention_norm.weight"] = vec(d)
        sd[p + "ffn_norm.weight"] = vec(d)
    hd = d // PARAMS["n_heads"]
    sd["rope.freqs"] = (1.0 / (10000.0 ** (np.arange(0, hd, 2)[: hd // 2] / hd))).astype(np.float16)
    return sd


# split dimension of Meta's model-parallel checkpoints (convert-pth-to-ggml.py:161-176 documents the same table)
def split_dim(name):
    if name.endswith("norm.weight") or name.endswith("freqs"):
        return None
    if "tok_embeddings" in name or "attention.wo" in name or "feed_forward.w2" in name:
        return 1
    return 0


def shard(sd, part, n_parts):
    """The part `part` of `n_parts` of a model-parallel checkpoint: matrices cut along their split dimension."""
    out = {}
    for name, a in sd.items():
        sdim = split_dim(name)
        if sdim is None:
            out[name] = a
        else:
            out[name] = np.ascontiguousarray(np.split(a, n_parts, axis=sdim)[part])
    return out
