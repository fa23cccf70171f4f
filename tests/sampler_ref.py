"""Independent numpy derivation of SampleTopPTopK (llama.go:455-707) used to cross-check the C checker; fp32 steps are explicit.
Written from the reference's description of the algorithm, not from oracle.c (different language, different sort, different
membership test), so an error in one is unlikely to be mirrored in the other."""
import numpy as np

M64 = (1 << 64) - 1
f32 = np.float32


def mix64(z):
    z = (z + 0x9E3779B97F4A7C15) & M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def uniform(seed, draw, j):
    key = mix64(seed ^ (((draw + 1) * 0xA24BAED4963EE407) & M64))
    return f32((mix64((key + j) & M64) >> 40)) * f32(1.0 / 16777216.0)


def sample(logits, ring, topK, topP, temp, penalty, seed, draw):
    """-> (token, candidate ids, candidate probs) with candidates in rank order after the topP cut/rescale."""
    l = np.asarray(logits, dtype=f32)
    V = l.size
    scale = f32(1.0) / f32(temp)
    v = l * scale
    member = np.zeros(V, dtype=bool)
    r = np.asarray(list(ring), dtype=np.int64)
    member[r[r < V]] = True
    with np.errstate(over="ignore", invalid="ignore"):
        pv = np.where(l < 0, v * f32(penalty), v / f32(penalty)).astype(f32)
    v = np.where(member, pv, v).astype(f32)
    v = v + f32(0.0)  # -0 -> +0 (they compare equal in the reference)
    order = np.lexsort((np.arange(V), -v.astype(np.float64)))  # value descending, then id ascending
    top = order[:topK]
    vals = v[top]
    with np.errstate(invalid="ignore"):  # -inf - -inf when every candidate is -inf
        d = (vals - vals[0]).astype(f32)
    p64 = np.exp(d.astype(np.float64))
    s = 0.0
    for p in p64:
        s += float(p)
    probs = (p64.astype(f32) / f32(s)).astype(f32)
    keep = topK
    if topP < 1.0:
        c = f32(0.0)
        for i in range(topK):
            c = f32(c + probs[i])
            if c >= f32(topP):
                keep = i + 1
                break
        probs = (probs[:keep] * (f32(1.0) / c)).astype(f32)
    probs = probs[:keep]
    w = np.empty(keep, dtype=f32)
    for i in range(keep):
        f = uniform(seed, draw, i)
        w[i] = f32(f32(f32(probs[i] * probs[i]) * f) * f)
    idx = int(np.argmax(w))  # first maximum
    return int(top[idx]), [int(t) for t in top[:keep]], probs
