"""-m gpu: the pods of one GPU in ONE weight pass (lh_batch), against the oracle.

The reference runs its pods as independent goroutines, each with its own llama.Context over the shared Model
(pkg/server/server.go:88-101, 151); whatever shares a tick here, every stream must decode exactly what it decodes alone:
 - token ids of every pod == the checker's greedy ids for that pod's prompt alone (prompts of different lengths, so the rows
   of a tick stand at different positions of different KV caches), logits of the last tick within 1e-4;
 - a row's result does not depend on its index in the tick or on its neighbours (bit-identical logits for the same prompt in two
   rows of one batch);
 - fp32 and block-int8 weights; row counts on both sides of every kernel boundary (2..16 folded norm, 17..32 / 33..48 / 49..64
   column tiles with K-split wo / w2); contexts beyond 256 (split-T attention per row); shapes the P-row kernels are not built
   for fall back to row-by-row evaluation with the same results;
 - the same through the pipeline scheduler on one rank (groups of streams per tick vs every stream on its own), greedy and with
   the reference's sampler (server.go:201-204) against the solo device loop and the checker's sampler.
"""
import numpy as np
import pytest

from llama_go_amd.mlapi import SHAPES, Batch, Pipeline, make_hparams

pytestmark = pytest.mark.gpu
TOL = 1e-4
SEED_40 = 5003   # (tools/check_test_margins.py --search 13: 4321 leaves a 1.7e-6 near-tie among the 240 steps)


def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / np.abs(b).max())


def make_prompts(rng, vocab, lengths):
    return [[int(t) for t in rng.integers(0, vocab, n)] for n in lengths]


MARGIN = 2.5 * TOL   # a greedy id is only a function of the logits where the top two differ by more than both may move (tolerance each)


class CheckerStreams:
    """Every stream alone on the checker: its ids, the logits of its last step, and the smallest top-2 margin over ALL its steps."""

    def __init__(self, oracle, hp, seed, prompts, n_predict, ctx, int8=False):
        om = oracle.NewSyntheticModel(hp, seed)
        if int8:
            om.QuantizeQ8()
        self.ids, last, self.margin = [], [], np.inf
        for pr in prompts:
            oc = om.NewContext(ctx, 16, False)
            t, lg = oc.GreedyDecode(pr, n_predict)
            oc.free()
            self.ids.append(list(t))
            last.append(lg[-1])
            srt = np.sort(lg, axis=-1)
            self.margin = min(self.margin, float(((srt[:, -1] - srt[:, -2]) / np.abs(lg).max(axis=-1)).min()))
        om.free()
        self.last = np.stack(last)

    def assert_ids(self, got, what=""):
        """Token ids are asserted UNCONDITIONALLY: a configuration whose checker run has a near-tie is a broken test, not a skipped one
        (tools/check_test_margins.py picks the seeds)."""
        if self.margin <= MARGIN:
            pytest.fail(f"the checker's own top-2 margin is {self.margin:.2e} <= {MARGIN:.1e} at some step: pick another seed (tools/check_test_margins.py)")
        assert got == self.ids, what


def oracle_streams(oracle, hp, seed, prompts, n_predict, ctx, int8=False):
    return CheckerStreams(oracle, hp, seed, prompts, n_predict, ctx, int8)


@pytest.mark.parametrize("shape,layers,int8,lengths,seed", [
    ("tiny", None, False, [4, 1, 11, 2, 7], 4321),
    ("tiny", None, True, [4, 1, 11, 2, 7], 4321),
    ("small", None, False, [3, 9], 4321),                               # two rows: the smallest batch (decode stream with two activation rows)
    ("small", None, False, [5, 2, 8], 4321),                            # three rows (four-row instantiation, one idle)
    ("7B", 2, False, [3, 1, 4, 2], 4321),                               # four rows on the 7B launches
    ("small", None, True, [3, 9], 5000),                                # block-int8, two rows (int8 decode stream with two activation rows)
    ("7B", 2, True, [3, 1, 4, 2], 4321),
    ("small", None, True, [3, 9, 1], 4321),
    ("small", None, False, list(range(1, 18)), 4321),                   # 17 rows: two column tiles, K-split wo / w2
    ("small", None, True, list(range(1, 18)), 4321),
    ("7B", 2, False, [8, 3, 1, 5, 2, 9, 4, 6], 4321),                   # the 7B launches, 8 rows (the decode stream with eight activation rows)
    ("7B", 2, True, [8, 3, 1, 5, 2, 9, 4, 6], 5000),                    #   block-int8: the dequantising stream kernel, folded norm
    ("7B", 2, False, [1 + (i % 5) for i in range(24)], 4321),           # 24 rows
    ("7B", 2, False, [1 + (i % 7) for i in range(40)], SEED_40),        # 40 rows: three column tiles
    ("7B", 2, True, [1 + (i % 7) for i in range(40)], 5000),
    ("7B", 2, False, [1 + (i % 3) for i in range(64)], 4321),           # 64 rows: four column tiles
    # both sides of every launch-shape boundary of the P-row kernels (the configurations above sit inside the ranges)
    ("7B", 2, False, [1 + (i % 4) for i in range(5)], 4321),            # 5 rows: first count past the four-row instantiation
    ("7B", 2, False, [1 + (i % 4) for i in range(9)], 4321),            # 9 rows: first count on the MFMA stream kernel (folded norm)
    ("7B", 2, False, [1 + (i % 4) for i in range(16)], 4321),           # 16 / 17 rows: one / two column tiles (folded norm / LDS-DMA loaders)
    ("7B", 2, False, [1 + (i % 4) for i in range(17)], 4321),
    ("7B", 2, False, [1 + (i % 4) for i in range(32)], 5001),           # 32 / 33 rows: two / three column tiles
    ("7B", 2, False, [1 + (i % 4) for i in range(33)], 5000),
    ("7B", 2, False, [1 + (i % 4) for i in range(48)], 5006),           # 48 / 49 rows: three / four column tiles
    ("7B", 2, False, [1 + (i % 4) for i in range(49)], 5010),
    ("7B", 2, True, [1 + (i % 4) for i in range(48)], 4321),            # block-int8 at its last batched row count
])
def test_batched_decode_equals_every_stream_alone(product, oracle, shape, layers, int8, lengths, seed):
    kw = dict(SHAPES[shape])
    if layers:
        kw["layers"] = layers
    n_predict, ctx = 5, 32
    hp = make_hparams(**kw, ctx=ctx)
    rng = np.random.default_rng(len(lengths) * 131 + int(int8))
    prompts = make_prompts(rng, kw["vocab"], lengths)
    m = product.NewSyntheticModel(hp, seed)
    if int8:
        m.QuantizeQ8()
    b = Batch(m, ctx, len(prompts))
    assert b.batched, "these shapes must take the one-pass route"
    ids, lg = b.GreedyDecode(prompts, n_predict, want_logits=True)
    ids2 = b.GreedyDecode(prompts[::-1], n_predict)     # the same batch object again, rows permuted: state fully reset, captured tick reused
    b.free()
    m.free()
    want = oracle_streams(oracle, hp, seed, prompts, n_predict, ctx, int8)
    assert rel(lg, want.last) <= TOL
    want.assert_ids(ids)
    want.assert_ids(ids2[::-1], "rows permuted")


def test_row_results_do_not_depend_on_the_neighbours(product):
    """The same prompt in rows 1 and 4 of one batch, among different neighbours at different positions: bit-identical logits and ids."""
    hp = make_hparams(**SHAPES["small"], ctx=48)
    rng = np.random.default_rng(5)
    same = [int(t) for t in rng.integers(0, hp.vocabSize, 6)]
    others = make_prompts(rng, hp.vocabSize, [2, 9, 13, 1])
    prompts = [others[0], same, others[1], others[2], same, others[3]]
    m = product.NewSyntheticModel(hp, 99)
    b = Batch(m, 48, len(prompts))
    ids, lg = b.GreedyDecode(prompts, 6, want_logits=True)
    b.free()
    # ... and in a batch of another size, next to other streams: still the same ids
    b2 = Batch(m, 48, 3)
    ids_b2 = b2.GreedyDecode([same, others[2], others[0]], 6)
    b2.free()
    m.free()
    assert ids[1] == ids[4] and np.array_equal(lg[1], lg[4])
    assert ids_b2[0] == ids[1]


@pytest.mark.parametrize("lengths,int8", [([6, 2], False), ([6, 2, 9], False), ([1, 7, 3, 5], False), ([6, 2], True), ([6, 2, 9], True), ([1, 7, 3, 5], True),
                                          ([1, 7, 3, 5, 2], False), ([4, 1, 6, 2, 9, 3], False), ([4, 1, 6, 2, 9, 3, 5], False), ([2, 8, 1, 5, 3, 7, 4, 6], False)])
def test_ticks_of_two_to_eight_pods_are_bit_identical_to_solo_decode(product, lengths, int8):
    """2..8 rows (block-int8: 2..4) ride the decode weight stream itself (k_gemv_rows / k_gemv_q8_rows: k_gemv_sa's / k_gemv_q8s' arithmetic per
    activation row): a pod's logits are BIT-identical to those of its solo run (llama.Eval per token on its own context), not merely within
    tolerance."""
    hp = make_hparams(**SHAPES["small"], ctx=48)
    rng = np.random.default_rng(len(lengths))
    prompts = make_prompts(rng, hp.vocabSize, lengths)
    m = product.NewSyntheticModel(hp, 21)
    if int8:
        m.QuantizeQ8()
    b = Batch(m, 48, len(prompts))
    assert b.batched
    ids, lg = b.GreedyDecode(prompts, 6, want_logits=True)
    b.free()
    for i, pr in enumerate(prompts):
        c = m.NewContext(48, 1)
        sid, slg = c.GreedyDecode(pr, 6)
        c.free()
        assert list(sid) == ids[i], i
        assert np.array_equal(slg[-1], lg[i]), (i, float(np.abs(slg[-1] - lg[i]).max()))
    m.free()


def test_batched_decode_long_context_split_attention(product, oracle):
    """Context > 256: the rows' attention runs split over the keys (k_attention_split per row), rows on both sides of a chunk boundary."""
    kw = dict(vocab=515, embd=640, mult=32, heads=5, layers=2)   # 5 heads of 128
    ctx = 320
    hp = make_hparams(**kw, ctx=ctx)
    rng = np.random.default_rng(11)
    prompts = make_prompts(rng, kw["vocab"], [260, 3, 127, 130])
    m = product.NewSyntheticModel(hp, 8)
    b = Batch(m, ctx, len(prompts))
    ids, lg = b.GreedyDecode(prompts, 4, want_logits=True)
    assert b.batched
    b.free()
    m.free()
    want = oracle_streams(oracle, hp, 8, prompts, 4, ctx)
    assert rel(lg, want.last) <= TOL
    want.assert_ids(ids)


@pytest.mark.parametrize("kw", [
    dict(vocab=300, embd=384, mult=32, heads=6, layers=2),      # head dim 64, ff = 1024: one pass, stream kernels
    dict(vocab=300, embd=160, mult=32, heads=5, layers=2),      # embd not a multiple of 128: one pass through the tile GEMM + the separate RoPE / append kernel
    dict(vocab=1000, embd=200, mult=8, heads=25, layers=2),     # embd not a multiple of 32: row by row on the decode kernels
])
def test_batched_decode_odd_shapes(product, oracle, kw):
    ctx = 40
    hp = make_hparams(**kw, ctx=ctx)
    rng = np.random.default_rng(kw["embd"])
    prompts = make_prompts(rng, kw["vocab"], [5, 1, 8])
    m = product.NewSyntheticModel(hp, 3)
    b = Batch(m, ctx, len(prompts))
    ids, lg = b.GreedyDecode(prompts, 5, want_logits=True)
    b.free()
    m.free()
    want = oracle_streams(oracle, hp, 3, prompts, 5, ctx)
    assert rel(lg, want.last) <= TOL
    want.assert_ids(ids)


@pytest.mark.parametrize("int8", [False, True])
def test_pipeline_groups_streams_into_one_weight_pass(product, oracle, int8):
    """The scheduler on one rank: 6 streams as ONE group (one weight pass per tick) and as six groups of one (max_rows = 1): the same ids
    as every stream alone, across two run() calls (state carried over)."""
    hp = make_hparams(**SHAPES["small"], ctx=40)
    rng = np.random.default_rng(21)
    prompts = make_prompts(rng, hp.vocabSize, [5, 1, 8, 2, 12, 3])
    m = product.NewSyntheticModel(hp, 17)
    if int8:
        m.QuantizeQ8()
    got = {}
    for mr in (0, 1, 4):
        pl = Pipeline(m, 40, len(prompts), 0, 1, max_rows=mr)
        assert pl.groups == {0: 1, 1: 6, 4: 2}[mr]
        pl.run(prompts, 2)
        pl.run(None, 3)
        got[mr] = [pl.tokens(i) for i in range(len(prompts))]
        pl.free()
    m.free()
    want = oracle_streams(oracle, hp, 17, prompts, 6, 40, int8)
    for mr in got:
        want.assert_ids(got[mr], mr)
    assert got[0] == got[4]


def test_pipeline_samples_like_the_solo_loop(product, oracle):
    """lh_pipeline_run_sample: SampleTopPTopK after every Eval (server.go:201-204) for every stream of a tick == the solo device loop
    (llama_SampleDecode) and the checker's sampler with the same seed."""
    hp = make_hparams(**SHAPES["small"], ctx=40)
    rng = np.random.default_rng(33)
    prompts = make_prompts(rng, hp.vocabSize, [5, 2, 9, 1])
    smp = dict(topK=40, topP=0.95, temp=0.8, repeatPenalty=1.10, seed=777)
    m = product.NewSyntheticModel(hp, 17)
    pl = Pipeline(m, 40, len(prompts), 0, 1)
    pl.run_sample(prompts, 3, **smp)
    pl.run_sample(None, 2, **smp)
    got = [pl.tokens(i) for i in range(len(prompts))]
    pl.free()
    solo = []
    for pr in prompts:
        c = m.NewContext(40, 1)
        solo.append(c.SampleDecode(pr, 6, **smp))
        c.free()
    m.free()
    assert got == solo
    om = oracle.NewSyntheticModel(hp, 17)
    for i, pr in enumerate(prompts):
        oc = om.NewContext(40, 16, False)
        want = oc.SampleDecode(pr, 6, **smp)
        oc.free()
        assert got[i] == want, i
    om.free()


def test_ticks_never_write_past_the_window(product):
    """lh_batch_stage (BatchHIP.Tick) is an Eval entry point like every other: the tick kernels index the row's KV cache and the RoPE table by
    the row's position, so a row at the window's end is dealt with BEFORE anything is enqueued - a whole-model batch that knows the row's tokens
    swaps its context as server.Do does (server.go:160-172; parity: tests/test_context_swap.py), anything else is an error."""
    from llama_go_amd.mlapi import MLError
    ctx = 12
    hp = make_hparams(**SHAPES["tiny"], ctx=ctx)
    m = product.NewSyntheticModel(hp, 5)
    b = Batch(m, ctx, 3)
    prompts = [[1, 2, 3], [4] * (ctx - 2), [7, 8]]          # row 1 stands at position ctx - 2 behind its prompt
    b.Prompt(prompts)
    for _ in range(3 * ctx):                                 # row 1 swaps at its third tick, the others later; every tick still yields three ids
        assert len(b.Tick()) == 3
    b.free()
    # a prompt may fill the window exactly (Eval's pastCount + N <= CtxSize): the first tick behind it swaps
    b3 = Batch(m, ctx, 2)
    b3.Prompt([[5] * ctx, [6, 7]])
    assert len(b3.Tick()) == 2
    b3.free()
    # KeepCount that leaves no room: an error, not a loop
    b4 = Batch(m, ctx, 2)
    b4.SetKeepCount(ctx)
    b4.Prompt([[5] * ctx, [6, 7]])
    with pytest.raises(MLError, match="KeepCount"):
        b4.Tick()
    b4.free()
    m.free()


def test_batches_come_and_go(product):
    """200 x { lh_batch_create -> prompts -> two ticks -> destroy } over one model: every cycle decodes the ids of the first (the create path once
    raced its own zero fill of the row table: a null KV-cache pointer in roughly one of forty two-rank runs, round 3)."""
    hp = make_hparams(**SHAPES["tiny"], ctx=24)
    m = product.NewSyntheticModel(hp, 77)
    rng = np.random.default_rng(4)
    prompts = make_prompts(rng, hp.vocabSize, [3, 1, 6, 2, 4])
    want = None
    for cycle in range(200):
        b = Batch(m, 24, len(prompts))
        ids = b.GreedyDecode(prompts, 3)
        b.free()
        want = want or ids
        assert ids == want, cycle
    m.free()


def test_batch_argument_errors(product):
    from llama_go_amd.mlapi import MLError
    hp = make_hparams(**SHAPES["tiny"], ctx=16)
    m = product.NewSyntheticModel(hp, 5)
    with pytest.raises(MLError):
        Batch(m, 16, 65)                       # more rows than one pass takes
    b = Batch(m, 16, 2)
    with pytest.raises(MLError):
        b.GreedyDecode([[1, 2], [9999]], 2)    # token id outside the vocabulary: nothing runs
    assert len(b.GreedyDecode([[1, 2], [3]], 30)[0]) == 30   # past the window of 16: the rows swap context like server.Do (tests/test_context_swap.py)
    assert b.GreedyDecode([[1, 2], [3]], 3) == b.GreedyDecode([[1, 2], [3]], 3)
    b.free()
    m.free()
