"""GPU: entry points that had no device-side test of their own and the ABI's error paths."""
import os

import numpy as np
import pytest
import torch

import lrcn_amd
from lrcn_amd import lrcn as L
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def test_init_weights_distribution_on_device():
    # initweights (lrcn.jl:489-510): xavier-uniform U(-s, s), s = sqrt(2 / (rows + cols)); biases 0 except b[1:H] = 1
    E, H1, H2, V = 96, 128, 64, 1500
    ctx = L.Context(E, H1, H2, V, max_B=2, max_T=1)
    model = L.initweights(ctx, seed=7)
    again = L.initweights(ctx, seed=7)
    other = L.initweights(ctx, seed=8)
    shapes = L.param_shapes(E, H1, H2, V)
    for k, (name, t, shp) in enumerate(zip(L.PARAM_NAMES, model, shapes)):
        a = L.from_jl(t)
        assert a.shape == shp, name
        assert np.array_equal(a, L.from_jl(again[k])), name            # counter-based: the seed fixes the stream
        if name in ("b1", "b2"):
            H = H1 if name == "b1" else H2
            assert (a[0, :H] == 1.0).all() and (a[0, H:] == 0.0).all(), name   # forget gate = block 1 (:501, :531)
        elif name == "bout":
            assert (a == 0.0).all()
        else:
            s = np.sqrt(2.0 / (shp[0] + shp[1]))
            assert np.abs(a).max() <= s * (1 + 1e-6), name
            assert np.abs(a).max() >= 0.98 * s, name
            assert abs(a.std() - s / np.sqrt(3.0)) <= 0.03 * s, (name, a.std(), s / np.sqrt(3.0))
            assert abs(a.mean()) <= 0.03 * s, name
            assert not np.array_equal(a, L.from_jl(other[k])), name
            # no structure along either axis (a counter hash keyed by index): row and column means are all near zero
            if min(shp) >= 64:
                assert np.abs(a.mean(axis=0)).max() <= 6 * s / np.sqrt(3.0 * shp[0]), name
                assert np.abs(a.mean(axis=1)).max() <= 6 * s / np.sqrt(3.0 * shp[1]), name
    ctx.close()


@pytest.mark.parametrize("name", ["lstm_tiny", "lstm_ragged"])
def test_train_step_entry_vs_golden_adam_trajectory(golden_dir, name):
    # lrcn_train_step = body of train1's loop (lrcn.jl:369-394) as ONE call, against the torch-autograd Adam trajectory
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    dims = tuple(int(z[k]) for k in ("E", "H1", "H2", "V"))
    T, B = z["tokens"].shape
    ctx = L.Context(*dims, max_B=B, max_T=T)
    param = L.model_from_arrays({n: z["p_" + n] for n in orc.PARAM_NAMES})
    opt = L.initparams(param)
    grads = L.zeros_like_model(param)
    feats = L.to_jl(z["feats"])
    for k, ref_loss in enumerate(z["adam_losses"]):
        val = L.train_step(ctx, param, opt, grads, feats, z["tokens"], norm_B=int(z["norm_B"]), pdrop=0.0, want_loss=True)
        assert abs(val - ref_loss) <= 2e-5 * abs(ref_loss), (k, val, ref_loss)
    assert opt.t == len(z["adam_losses"])
    for n, p in zip(orc.PARAM_NAMES, param):
        np.testing.assert_allclose(L.from_jl(p), z["a_" + n], rtol=0, atol=5e-6, err_msg=n)
    ctx.close()


def test_average_loss_aggregation_vs_oracle():
    # average_loss (lrcn.jl:407-486): -sum(logp) / sum(B * (T+1)) over batches of unequal size and length; T > 28 skipped (:438)
    rng = np.random.default_rng(3)
    E, H1, H2, V = 32, 48, 32, 211
    m = orc.init_weights(E, H1, H2, V, seed=11)
    ctx = L.Context(E, H1, H2, V, max_B=9, max_T=28)
    param = L.model_from_arrays(m.p)
    batches, tot, cnt = [], 0.0, 0
    for B, T in [(9, 3), (4, 11), (7, 0)]:
        feats = (rng.standard_normal((B, 4096)) * 0.05).astype(np.float32)
        tokens = rng.integers(0, V, size=(T, B)).astype(np.int32)
        batches.append((L.to_jl(feats), tokens))
        tot += orc.loss(m, feats, tokens) * B * (T + 1)
        assert L.avg_loss_batch(ctx, param, batches[-1][0], tokens) == L.loss(ctx, param, batches[-1][0], tokens)   # rev 5: = lrcn_loss(norm_B = B)
        cnt += B * (T + 1)
    # a 29-token batch is skipped by the reference; lrcn_loss itself refuses it (max_T = 28)
    long_tok = rng.integers(0, V, size=(29, 2)).astype(np.int32)
    batches.insert(1, (L.to_jl(np.zeros((2, 4096), np.float32)), long_tok))
    got = L.average_loss(ctx, param, batches)
    assert abs(got - tot / cnt) <= 1e-5 * abs(tot / cnt), (got, tot / cnt)
    with pytest.raises(lrcn_amd.LrcnError):
        L.loss(ctx, param, batches[1][0], long_tok)
    ctx.close()


def test_out_of_range_token_ids_are_reported():
    # The reference raises BoundsError (lrcn.jl:556/569).  The device never faults (ids are clamped to unk) and the next
    # synchronising call returns LRCN_EINVAL -- e.g. a caller that forgot the 1-based -> 0-based shift (id V appears).
    E, H1, H2, V, B, T = 16, 16, 16, 40, 4, 3
    ctx = L.Context(E, H1, H2, V, max_B=B, max_T=T)
    param = L.initweights(ctx, seed=1)
    feats = L.to_jl(np.zeros((B, 4096), np.float32))
    ok = np.full((T, B), 5, np.int32)
    bad = ok.copy()
    bad[1, 2] = V
    with pytest.raises(lrcn_amd.LrcnError, match="token id"):
        L.loss(ctx, param, feats, bad)
    assert np.isfinite(L.loss(ctx, param, feats, ok))       # the flag is cleared once reported
    neg = ok.copy()
    neg[0, 0] = -1
    grads, _ = L.lossgradient(ctx, param, feats, neg, want_loss=False)   # asynchronous: no error yet
    with pytest.raises(lrcn_amd.LrcnError, match="token id"):
        ctx.sync()
    ctx.sync()
    ctx.close()


def test_profile_segments_bracket_the_hbm_bound_parts_of_a_step():
    # lrcn_profile(ctx, 2) + lrcn_profile_segment (rev 5): after three train steps every LSTM-side segment has three brackets (the
    # recurrence: one per layer pass), a positive time, and exactly the algorithmic bytes include/lrcn.h states for it
    E = H = 64
    V, B, T = 301, 8, 5
    rng = np.random.default_rng(2)
    ctx = L.Context(E, H, H, V, max_B=B, max_T=T, lstm_dtype=lrcn_amd.LRCN_BF16)
    param = L.initweights(ctx, seed=3)
    optim = L.initparams(param)
    grads = [L.jl_empty(*t.shape) for t in param]
    feats = L.to_jl((rng.standard_normal((B, 4096)) * 0.05).astype(np.float32))
    tokens = rng.integers(0, V, size=(T, B)).astype(np.int32)
    L.profile(ctx, 2)
    for k in range(3):
        L.train_step(ctx, param, optim, grads, feats, tokens, pdrop=0.4, seed=k)
    seg = L.profile_segments(ctx)
    nparam = sum(int(t.numel()) for t in param)
    M = (T + 1) * B
    want = {"update": (3, 3 * 28.0 * nparam), "rec_fwd": (6, 6 * T * 4.0 * H * H * 2), "rec_bwd": (6, 6 * T * 4.0 * H * H * 2),
            "embed_gather": (3, 3 * 2.0 * M * E * 2), "embed_grad": (3, 3 * (4.0 * M * E + 4.0 * V * E))}
    for name, (n, by) in want.items():
        ms, got_n, got_by = seg[name]
        assert got_n == n and ms > 0 and got_by == by, (name, seg[name], n, by)
    assert seg["preprocess"][1] == 0 and seg["upload"][1] == 0   # no VGG forward, no upload in these steps
    L.profile(ctx, 0)
    L.train_step(ctx, param, optim, grads, feats, tokens, pdrop=0.4, seed=9)
    assert all(v[1] == 0 for v in L.profile_segments(ctx).values())   # switching the level resets, level 0 records nothing
    ctx.close()
