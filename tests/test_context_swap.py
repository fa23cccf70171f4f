"""The context swap of the reference's generation loop (pkg/server/server.go:160-172; the same lines in main.go:188-200):

    if pastCount+uint32(len(embd)) > Params.CtxSize {
        leftCount := pastCount - Params.KeepCount
        pastCount = Params.KeepCount
        embd = append(llama.ExtractTokens(lastNTokens.Move(-int(leftCount/2)), int(leftCount/2)), embd...)
    }

lastNTokens already holds the pending token when this runs (appendToken precedes `embd = append(embd, id)`, server.go:207-214), so the
re-fed run ends with that token and the token follows once more.  CPU: the checker's loops (oracle.c eval_pending_token) against a loop
written here straight from the Go lines, with container/ring's semantics, over the checker's own llama.Eval.  GPU: the product's loops
(the host loop over ml_GraphCompute, the device-resident greedy and sampling loops, pods in one weight pass) against the checker."""
import collections

import numpy as np
import pytest

from llama_go_amd.mlapi import SHAPES, make_hparams

TOL = 1e-4
MARGIN = 2.5 * TOL


def go_loop(ctx, prompt, n_predict, ctx_size, keep):
    """server.Do, greedy, prompt in one batch: returns (ids, the (tokens, pastCount) of every Eval)."""
    ring = collections.deque([0] * ctx_size, maxlen=ctx_size)     # ring.New(CtxSize), zeroed; appendToken = overwrite the oldest
    past, embd, consumed, remained, out, evals = 0, [], 0, n_predict, [], []
    logits = None
    while remained > 0:
        if embd:
            if past + len(embd) > ctx_size:
                left = past - keep
                past = keep
                embd = list(ring)[ctx_size - left // 2:] + embd    # Move(-n) then n x Next(): the n newest entries, oldest first
            logits = ctx.Eval(embd, past)
            evals.append((list(embd), past))
        past += len(embd)
        embd = []
        if consumed < len(prompt):
            while consumed < len(prompt):
                embd.append(prompt[consumed])
                ring.append(prompt[consumed])
                consumed += 1
        else:
            tok = int(np.argmax(logits))                           # greedy = lowest-index argmax
            ring.append(tok)
            embd.append(tok)
            out.append(tok)
            remained -= 1
    return out, evals


@pytest.mark.parametrize("ctx_size,keep,n_prompt,n_predict", [(16, 0, 5, 40), (16, 3, 5, 40), (12, 0, 12, 20), (9, 8, 4, 15), (8, 7, 3, 10)])
def test_checker_loop_swaps_like_the_go_lines(oracle, ctx_size, keep, n_prompt, n_predict):
    hp = make_hparams(**SHAPES["tiny"], ctx=ctx_size)
    rng = np.random.default_rng(ctx_size * 100 + keep)
    prompt = [int(t) for t in rng.integers(0, hp.vocabSize, n_prompt)]
    m = oracle.NewSyntheticModel(hp, 31)
    c = m.NewContext(ctx_size, 4, False)
    c.SetKeepCount(keep)
    ids, _ = c.GreedyDecode(prompt, n_predict)
    c.free()
    c2 = m.NewContext(ctx_size, 4, False)
    want, evals = go_loop(c2, prompt, n_predict, ctx_size, keep)
    c2.free()
    m.free()
    assert ids == want
    swaps = [e for e in evals[1:] if len(e[0]) != 1 or e[1] == keep]
    assert swaps, "the case never reached the window's end"
    for toks, past in evals:
        assert past + len(toks) <= ctx_size
    # a swap Eval is (ctx - keep) / 2 re-fed tokens + the pending one at position keep, and the pending token appears twice at its end
    toks, past = next(e for e in evals[1:] if e[1] == keep and len(e[0]) == (ctx_size - keep) // 2 + 1)
    assert past == keep and (len(toks) == 1 or toks[-1] == toks[-2])


def _streams(lib, hp, seed, prompts, n_predict, ctx_size, keep, int8=False):
    m = lib.NewSyntheticModel(hp, seed)
    if int8:
        m.QuantizeQ8()
    out, margin = [], np.inf
    for pr in prompts:
        c = m.NewContext(ctx_size, 16, False)
        c.SetKeepCount(keep)
        ids, lg = c.GreedyDecode(pr, n_predict)
        c.free()
        srt = np.sort(lg, axis=-1)
        margin = min(margin, float(((srt[:, -1] - srt[:, -2]) / np.abs(lg).max(axis=-1)).min()))
        out.append((ids, lg))
    m.free()
    return out, margin


@pytest.mark.gpu
@pytest.mark.parametrize("shape,ctx_size,keep,n_prompt,n_predict,seed", [("small", 32, 0, 8, 80, 11), ("small", 32, 5, 8, 80, 11), ("tiny", 16, 0, 16, 40, 3), ("small", 24, 23, 4, 30, 11)])
def test_generation_loops_swap_context_like_the_checker(product, oracle, shape, ctx_size, keep, n_prompt, n_predict, seed):
    """80 tokens through a window of 32: the host loop (llama_GreedyDecode: one ml_GraphCompute per Eval, the Go shim's route), the
    device-resident greedy loop and the device-resident sampling loop against the checker's loops."""
    from llama_go_amd.mlapi import decode_greedy_resident
    hp = make_hparams(**SHAPES[shape], ctx=ctx_size)
    rng = np.random.default_rng(ctx_size + keep)
    prompt = [int(t) for t in rng.integers(0, hp.vocabSize, n_prompt)]
    out, margin = _streams(oracle, hp, seed, [prompt], n_predict, ctx_size, keep)
    want, wlg = out[0]
    if margin <= MARGIN:
        pytest.fail(f"the checker's own top-2 margin is {margin:.2e}: pick another seed")
    m = product.NewSyntheticModel(hp, seed)
    # (1) host loop
    c = m.NewContext(ctx_size, 1)
    c.SetKeepCount(keep)
    ids, lg = c.GreedyDecode(prompt, n_predict)
    c.free()
    assert ids == want
    assert float(np.abs(lg.astype(np.float64) - wlg).max() / np.abs(wlg).max()) <= TOL
    # (2) resident greedy loop: prompt through the context (so it knows the window's tokens), then n_predict - 1 steps from the first id
    c = m.NewContext(ctx_size, 1)
    c.SetKeepCount(keep)
    first = int(np.argmax(c.Eval(prompt, 0)))
    rest, _ = decode_greedy_resident(c, first, n_prompt, n_predict - 1)
    c.free()
    assert [first] + rest == want
    # (3) resident sampling loop == the checker's sampling loop with the same seed (top-k 40, temperature 0.8)
    smp = dict(topK=40, topP=0.95, temp=0.8, repeatPenalty=1.10, seed=99)
    c = m.NewContext(ctx_size, 1)
    c.SetKeepCount(keep)
    got = c.SampleDecode(prompt, n_predict, **smp)
    c.free()
    m.free()
    om = oracle.NewSyntheticModel(hp, seed)
    oc = om.NewContext(ctx_size, 16, False)
    oc.SetKeepCount(keep)
    wsmp = oc.SampleDecode(prompt, n_predict, **smp)
    oc.free()
    om.free()
    assert got == wsmp


@pytest.mark.gpu
def test_resident_loop_without_the_windows_tokens_fails_cleanly(product):
    """A context that never saw the tokens of its window (KV cache filled by someone else) cannot swap: an error at the window's end,
    not a guess."""
    from llama_go_amd.mlapi import MLError, decode_greedy_resident
    hp = make_hparams(**SHAPES["tiny"], ctx=16)
    m = product.NewSyntheticModel(hp, 3)
    c = m.NewContext(16, 1)
    with pytest.raises(MLError, match="not known"):
        decode_greedy_resident(c, 5, 10, 12)      # positions 0..9 were never evaluated through this context
    c.free()
    m.free()


@pytest.mark.gpu
@pytest.mark.parametrize("shape,int8,keep,lengths,seed", [("small", False, 0, [5, 1, 12, 3], 12), ("small", False, 4, [7, 2, 9, 4, 15, 1], 12), ("small", True, 0, [5, 1, 12], 13)])
def test_pods_in_one_weight_pass_swap_context_per_row(product, oracle, shape, int8, keep, lengths, seed):
    """lh_batch: rows reach the end of their windows at different ticks; each swaps on its own cache while the others go on, and every pod
    decodes what it decodes alone on the checker (60 tokens through windows of 24)."""
    from llama_go_amd.mlapi import Batch
    ctx_size, n_predict = 24, 60
    hp = make_hparams(**SHAPES[shape], ctx=ctx_size)
    rng = np.random.default_rng(len(lengths) + keep)
    prompts = [[int(t) for t in rng.integers(0, hp.vocabSize, n)] for n in lengths]
    out, margin = _streams(oracle, hp, seed, prompts, n_predict, ctx_size, keep, int8)
    if margin <= MARGIN:
        pytest.fail(f"the checker's own top-2 margin is {margin:.2e}: pick another seed")
    m = product.NewSyntheticModel(hp, seed)
    if int8:
        m.QuantizeQ8()
    b = Batch(m, ctx_size, len(prompts))
    b.SetKeepCount(keep)
    ids = b.GreedyDecode(prompts, n_predict)
    b.free()
    m.free()
    assert ids == [o[0] for o in out]


@pytest.mark.gpu
def test_sampler_set_mid_stream_keeps_the_swap_history_right(product):
    """lh_batch_set_sampler behind a prompt and a few ticks: it restarts the rows' output lists on the device, so the host's
    mirror of them has to restart too - the context swap reads the tokens of the window through that mirror.  A top-1 sampler picks the
    argmax, so pods that switch to it after three greedy ticks must go on decoding exactly what all-greedy pods decode, through two swaps."""
    from llama_go_amd.mlapi import Batch
    ctx_size, n_ticks = 24, 50
    hp = make_hparams(**SHAPES["small"], ctx=ctx_size)
    rng = np.random.default_rng(77)
    prompts = [[int(t) for t in rng.integers(0, hp.vocabSize, n)] for n in (5, 11, 2)]
    m = product.NewSyntheticModel(hp, 12)
    runs = []
    for switch_at in (None, 3):
        b = Batch(m, ctx_size, len(prompts))
        ids = [b.Prompt(prompts)]
        for t in range(n_ticks):
            if t == switch_at:
                b.SetSampler(topK=1, topP=1.0, temp=1.0, repeatPenalty=1.0, seed=5, ringSize=ctx_size)
            ids.append(b.Tick())
        b.free()
        runs.append(ids)
    m.free()
    assert runs[0] == runs[1]


@pytest.mark.gpu
def test_unsharded_pipeline_swaps_context(product, oracle):
    """The scheduler on one rank (lh_pipeline_run over lh_batch ticks): streams outlive their windows across run() calls."""
    from llama_go_amd.mlapi import Pipeline
    ctx_size, seed = 24, 13
    hp = make_hparams(**SHAPES["small"], ctx=ctx_size)
    rng = np.random.default_rng(77)
    prompts = [[int(t) for t in rng.integers(0, hp.vocabSize, n)] for n in [6, 2, 11]]
    out, margin = _streams(oracle, hp, seed, prompts, 50, ctx_size, 0)
    if margin <= MARGIN:
        pytest.fail(f"the checker's own top-2 margin is {margin:.2e}: pick another seed")
    m = product.NewSyntheticModel(hp, seed)
    pl = Pipeline(m, ctx_size, len(prompts), 0, 1)
    pl.run(prompts, 19)
    pl.run(None, 30)
    got = [pl.tokens(i) for i in range(len(prompts))]
    pl.free()
    m.free()
    assert got == [o[0] for o in out]


@pytest.mark.gpu
@pytest.mark.parametrize("world,keep,sample", [(2, 0, 0), (3, 3, 0), (2, 0, 1)])
def test_sharded_pipeline_swaps_context_across_ranks(product, world, keep, sample):
    """Two / three ranks (layer blocks) on ONE GPU (host-staged p2p), streams of different prompt lengths in two groups, 50 ids through windows of
    24: in the unit where a stream stands at the window's end its re-fed run travels through every stage in front of the tick's rows.  Every stream
    must decode exactly what the single-process loops decode for it (those are checked against the checker above)."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ctx_size, seed = 24, 17
    hp = make_hparams(**SHAPES["small"], ctx=ctx_size)
    rng = np.random.default_rng({(2, 0): 520, (3, 3): 33}[(world, keep)])   # prompt seeds whose checker runs keep a top-2 margin > 5e-4 over all 250 steps
    prompts = [[int(t) for t in rng.integers(0, hp.vocabSize, n)] for n in (5, 1, 9, 3, 2)]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "tests", "pipeline_worker.py"), "small", str(ctx_size), "19", str(sample), json.dumps(prompts), str(keep), "30"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-6000:])
    got = None
    for line in reversed(r.stdout.splitlines()):
        if line.strip().startswith("{") and '"ids"' in line:
            got = json.loads(line)
            break
    assert got is not None, r.stdout[-2000:]
    m = product.NewSyntheticModel(hp, seed)
    smp = dict(topK=40, topP=0.95, temp=0.8, repeatPenalty=1.10, seed=777)
    for i, pr in enumerate(prompts):
        c = m.NewContext(ctx_size, 1)
        c.SetKeepCount(keep)
        want = c.SampleDecode(pr, 50, **smp) if sample else c.GreedyDecode(pr, 50, want_logits=False)[0]
        c.free()
        assert got["ids"][i] == list(want), (i, got["ids"][i], want)
    m.free()
