"""GPU: LRCN_OPT_DETERMINISTIC (include/lrcn.h rev 3).  Four sums on the lossgradient route are float atomics by default (the embedding-
gradient scatter, the slabbed bias column sums, the split-K of the skinny direct-to-LDS contractions, the loss accumulator): fast, but
two identical calls then differ in their last bits at the benchmark's size.  Under the option every sum has a fixed order."""
import numpy as np
import pytest
import torch

import lrcn_amd
from lrcn_amd import _lib
from lrcn_amd import lrcn as L

pytestmark = pytest.mark.gpu


def grads_of(ctx, param, feats, toks, pdrop, seed):
    g, val = L.lossgradient(ctx, param, feats, toks, pdrop=pdrop, seed=seed)
    torch.cuda.synchronize()
    return [L.from_jl(t).copy() for t in g], val


@pytest.mark.parametrize("B,T,E,V", [(256, 11, 1000, 10640), (32, 11, 1000, 10640), (21, 5, 100, 301)])
def test_two_calls_are_bit_identical(B, T, E, V):
    ctx = L.Context(E, E, E, V, max_B=B, max_T=T, lstm_dtype=lrcn_amd.LRCN_BF16)
    ctx.set_option(_lib.LRCN_OPT_DETERMINISTIC, 1)
    param = L.initweights(ctx, seed=42)
    rng = np.random.default_rng(1)
    feats = L.to_jl((rng.standard_normal((B, 4096)) * 0.01).astype(np.float32))
    pz = 1.0 / np.arange(1, V - 3 + 1)
    toks = (rng.choice(V - 3, size=(T, B), p=pz / pz.sum()) + 3).astype(np.int32)   # Zipf ids: many rows share a token
    ga, la = grads_of(ctx, param, feats, toks, 0.4, 9)
    gb, lb = grads_of(ctx, param, feats, toks, 0.4, 9)
    assert la == lb
    for k, (a, b) in enumerate(zip(ga, gb)):
        assert np.array_equal(a, b), "gradient %d differs between two identical calls" % k
    # and it is the same gradient the default (atomic) path computes, to summation-order accuracy
    ctx.set_option(_lib.LRCN_OPT_DETERMINISTIC, 0)
    gc, lc = grads_of(ctx, param, feats, toks, 0.4, 9)
    assert abs(lc - la) <= 1e-9 * abs(la)
    for k, (a, c) in enumerate(zip(ga, gc)):
        den = np.linalg.norm(a.astype(np.float64)) + 1e-30
        assert np.linalg.norm(a.astype(np.float64) - c.astype(np.float64)) / den < 2e-3, k   # bf16 roundings amplify summation order
    ctx.close()


def test_ordered_embedding_sums_match_the_oracle(golden_dir):
    """The sorted segmented sums against the CPU oracle on a golden case with repeated tokens and explicit dropout masks."""
    import os
    from oracle import oracle as orc
    z = np.load(os.path.join(golden_dir, "lstm_tiny_drop.npz"))
    E, H1, H2, V = (int(z[k]) for k in ("E", "H1", "H2", "V"))
    T, B = z["tokens"].shape
    for det in (0, 1):
        ctx = L.Context(E, H1, H2, V, max_B=B, max_T=T, lstm_dtype=lrcn_amd.LRCN_F32)
        ctx.set_option(_lib.LRCN_OPT_DETERMINISTIC, det)
        param = L.model_from_arrays({n: z["p_" + n] for n in orc.PARAM_NAMES})
        g, val = L.lossgradient(ctx, param, L.to_jl(z["feats"]), z["tokens"], norm_B=int(z["norm_B"]), mask1=z["mask1"], mask2=z["mask2"])
        np.testing.assert_allclose(val, float(z["loss"]), rtol=1e-5)
        np.testing.assert_allclose(L.from_jl(g[6]), z["g_Wembed"], rtol=1e-3, atol=1e-5)
        np.testing.assert_allclose(L.from_jl(g[8]), z["g_bout"], rtol=1e-3, atol=1e-5)
        ctx.close()


def test_too_many_rows_for_the_ordered_sums_is_an_error():
    ctx = L.Context(64, 64, 64, 300, max_B=300, max_T=28, lstm_dtype=lrcn_amd.LRCN_BF16)
    ctx.set_option(_lib.LRCN_OPT_DETERMINISTIC, 1)
    param = L.initweights(ctx, seed=1)
    feats = L.to_jl(np.zeros((300, 4096), np.float32))
    toks = np.full((28, 300), 5, np.int32)
    with pytest.raises(L.LrcnError, match="8192"):
        L.lossgradient(ctx, param, feats, toks)
    ctx.sync()
    # the context is still usable (every stream joined on the error path)
    ctx.set_option(_lib.LRCN_OPT_DETERMINISTIC, 0)
    g, val = L.lossgradient(ctx, param, feats, toks)
    assert np.isfinite(val)
    ctx.close()
