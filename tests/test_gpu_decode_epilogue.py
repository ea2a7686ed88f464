"""GPU: the batched beam decode (generate / beam_search, lrcn.jl:585-678) at the PRODUCTION shape of BASELINE configs[4] -- a chunk of
1024 images x 5 beams = 5120 hypotheses, E = H = 1000, V = 10640 -- with the cell math in the gate GEMM's epilogue (gemm_8p.hip
GEMM_OUT_LSTM_FWD) against the GEMM + cell-kernel decode (LRCN_DECODE_EPI=0) and the bf16-emulating oracle.

Why this file exists: round 5's epilogue wrote h(t) into the h columns of the very [x | h] operand the launch was still
reading.  5120 x 4000 gates are 640 tiles of 256 x 128 at one workgroup per CU = 2.5 rounds, and the XCD renumbering spreads the tiles of
one row block over different rounds, so late tiles contracted against h(t) instead of h(t-1).  Every decode test of round 5 had <= 320
rows (2 row blocks, 64 tiles, one round) and could not see it.  The epilogue now writes to st_h1 / st_h2; these tests pin the shape where
the race lived, alone and beside a concurrent VGG forward (tiles then trickle, as in tools/caption_bench.py)."""
import numpy as np
import pytest
import torch

import lrcn_amd
from lrcn_amd import lrcn as L
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

E = H = 1000
V = 10640
K, NWORD = 5, 8


def decisive_model(seed=4):
    """Random weights scaled until the word distributions are peaky (no near-ties): a decode is then a function of the state arithmetic,
    not of which side of a tie a rounding falls on."""
    rng = np.random.default_rng(seed)
    m = orc.init_weights(E, H, H, V, seed=seed)
    for n in ("W1", "W2", "Wout"):
        m.p[n] *= 2.0
    m.p["Wout"][:] *= 8.0
    m.p["bout"][:] = (rng.standard_normal((1, V)) * 2.0).astype(np.float32)
    m.p["b1"][:] += (rng.standard_normal(m.p["b1"].shape) * 0.5).astype(np.float32)
    return m


def compare(a, b, N):
    same = sum(x[0] == y[0] for x, y in zip(a, b))
    assert same >= N - N // 50, (same, N)   # >= 98 %: a near-tie may fall the other way under another summation order
    for (ta, pa), (tb, pb) in zip(a, b):
        if ta == tb:
            assert abs(pa - pb) <= 2e-2 * abs(pb) + 1e-30, (pa, pb)
        else:
            assert abs(np.log(pa + 1e-300) - np.log(pb + 1e-300)) < 0.3, (pa, pb)
    return same


@pytest.mark.parametrize("N", [1024, 819, 2048])   # 5120 rows = 20 full row blocks; 4095 rows = an M tail in the last of 16; 10240 = the C5 bench's chunk
def test_5120_hypotheses_cell_epilogue_equals_gemm_plus_cell_kernel_and_the_oracle(N, monkeypatch):
    m = decisive_model()
    feats = (np.random.default_rng(N).standard_normal((N, 4096)) * 0.05).astype(np.float32)
    ctx = L.Context(E, H, H, V, max_B=N * K, max_T=2, lstm_dtype=lrcn_amd.LRCN_BF16)
    param = L.model_from_arrays(m.p)
    out = {}
    for knob in ("1", "0", "1"):   # the epilogue decode twice: it must also repeat itself
        monkeypatch.setenv("LRCN_DECODE_EPI", knob)
        r = L.beam_search_batch(ctx, param, L.to_jl(feats), K, NWORD)
        if knob in out:
            assert [t for t, _ in r] == [t for t, _ in out[knob]], "the epilogue decode does not repeat itself"
        out[knob] = r
    compare(out["1"], out["0"], N)
    # hypotheses spread over the row blocks, against the emulating oracle's per-image decode
    agree = 0
    picks = [0, 1, N // 2, N - 2, N - 1]
    for i in picks:
        with orc.emulate_bf16():
            rt, rp = orc.beam_search(m, feats[i], K, NWORD)
        if out["1"][i][0] == list(rt):
            agree += 1
            assert abs(out["1"][i][1] - rp) <= 5e-2 * abs(rp) + 1e-30
    assert agree >= len(picks) - 1, agree
    ctx.close()


def test_5120_hypotheses_beside_a_concurrent_vgg_forward(monkeypatch):
    """tools/caption_bench.py's situation: the VGG forward of the next pass runs on a side stream beside the decode, so the decode's tiles
    trickle onto whatever CUs come free and rounds interleave arbitrarily."""
    N = 1024
    m = decisive_model(seed=5)
    feats = (np.random.default_rng(77).standard_normal((N, 4096)) * 0.05).astype(np.float32)
    ctx = L.Context(E, H, H, V, max_B=N * K, max_T=2, lstm_dtype=lrcn_amd.LRCN_BF16)
    vctx = L.Context(8, 8, 8, 17, max_B=2, max_T=1, vgg_dtype=lrcn_amd.LRCN_BF16, max_images=256)
    L.vgg_load(vctx, *L.synthetic_vgg_weights(seed=1))
    img = torch.as_tensor(np.random.default_rng(3).integers(0, 256, size=(256, 224, 224, 3), dtype=np.uint8)).cuda()
    fbuf = L.jl_empty(256, L.CNNOUT)
    side = torch.cuda.Stream()
    vctx.use_stream(side)
    param = L.model_from_arrays(m.p)
    monkeypatch.setenv("LRCN_DECODE_EPI", "0")
    alone = L.beam_search_batch(ctx, param, L.to_jl(feats), K, NWORD)
    monkeypatch.setenv("LRCN_DECODE_EPI", "1")
    L.beam_search_batch(ctx, param, L.to_jl(feats), K, NWORD)   # once alone: the epilogue route's lazily allocated buffers exist before anything is timed against it
    torch.cuda.synchronize()
    for _ in range(10):   # ~60 ms of convolution launches queued on the side stream: they outlast the decode (~10 ms) several times over
        L.convnet_u8(vctx, img, feats=fbuf)
    beside = L.beam_search_batch(ctx, param, L.to_jl(feats), K, NWORD)
    busy = not side.query()
    torch.cuda.synchronize()
    compare(beside, alone, N)
    assert busy, "the VGG forwards finished before the decode: nothing ran beside it"
    vctx.close()
    ctx.close()


@pytest.mark.parametrize("smax", ["1", "0"])
def test_fused_softmax_topk_follows_probabilities_in_a_tie_group_across_records(smax, monkeypatch):
    """Round 6: from 256 hypotheses the logits GEMM reduces its tiles to per-row records {max, sum exp, 6 best logits} of 128 columns each and
    a merge kernel ranks them (gemm_8p.hip GEMM_OUT_SMAX_TOPK, kernels.hip softmax_topk_merge_kernel) -- the f32 logits are never written.
    The reference ranks float32 PROBABILITIES with a stable sort (lrcn.jl:652-656), so distinct logits whose probabilities round to one
    float form a tie group whose LOWEST column wins.  Wout = 0 makes the logits = bout exactly; columns 300 < 400 < 500 (two different
    128-column records) get x, nextafter(x), nextafter(nextafter(x)): a logit-ranked top-1 returns 500, the reference 300 -- at every step
    and for every one of the 256 images.  LRCN_DECODE_SMAX=0: the same decode through GEMM + softmax_topk_rows_kernel."""
    E = H = 128
    V, K, nword, N = 600, 1, 4, 256
    m = orc.init_weights(E, H, H, V, seed=1)
    m.p["Wout"][:] = 0.0
    bout = np.full(V, -4.0, np.float32)
    bout[0] = -9.0
    x = np.float32(1e-3)
    bout[300] = x
    bout[400] = np.nextafter(x, np.float32(1.0))
    bout[500] = np.nextafter(bout[400], np.float32(1.0))
    m.p["bout"][:] = bout[None, :]
    lse = np.log(np.exp(bout.astype(np.float64)).sum())
    pf = np.exp(bout.astype(np.float64) - lse).astype(np.float32)
    assert bout[300] < bout[400] < bout[500] and pf[300] == pf[400] == pf[500]
    feats = (np.random.default_rng(0).standard_normal((N, 4096)) * 0.05).astype(np.float32)
    ctx = L.Context(E, H, H, V, max_B=N * K, max_T=1, lstm_dtype=lrcn_amd.LRCN_BF16)
    monkeypatch.setenv("LRCN_DECODE_SMAX", smax)
    got = L.beam_search_batch(ctx, L.model_from_arrays(m.p), L.to_jl(feats), K, nword)
    want = [1] + [300] * (nword + 1)
    assert all(t == want for t, _ in got), [t for t, _ in got if t != want][:3]
    ref_p = float(pf[300]) ** (nword + 1)
    assert all(abs(p - ref_p) <= 1e-4 * ref_p for _, p in got)
    ctx.close()


@pytest.mark.parametrize("V,K,layers", [(10640, 5, 2), (7732, 3, 2), (1028, 5, 2), (256, 2, 2), (304, 4, 1), (2052, 5, 1)])
def test_fused_softmax_topk_equals_the_row_kernel(V, K, layers, monkeypatch):
    """The same decode with the logits' softmax / top-K in the GEMM epilogue and through softmax_topk_rows_kernel, on random (not peaky)
    distributions at V = the benchmark's, V = 7732 (a last tile of 52 columns: one record with 52 valid columns, one with none), 1028 (4
    columns in the fifth tile) and 256 (one tile): the first step -- identical inputs on both sides, no state yet -- must return the same K
    tokens per image and probabilities to 1e-5 (sums in another order); nword = 1 so that nothing else enters."""
    E = H = 128
    N = 300 // K + 1
    while N * K < 256:
        N += 1
    m = orc.init_weights(E, H, H, V, seed=V, n_layers=layers)   # layers = 1: LRCN-1f ([embedding | x_cnn | h] operand, logits from h1)
    m.p["Wout"] *= 4.0
    m.p["bout"][:] = (np.random.default_rng(V).standard_normal((1, V)) * 1.5).astype(np.float32)
    feats = (np.random.default_rng(1).standard_normal((N, 4096)) * 0.05).astype(np.float32)
    ctx = L.Context(E, H, H, V, max_B=N * K, max_T=1, lstm_dtype=lrcn_amd.LRCN_BF16, n_layers=layers)
    param = L.model_from_arrays(m.p)
    out = {}
    for knob in ("1", "0"):
        monkeypatch.setenv("LRCN_DECODE_SMAX", knob)
        out[knob] = L.beam_search_batch(ctx, param, L.to_jl(feats), K, 3 if layers == 1 else 1)   # (1f: a few steps, so that parent-indexed states matter)
    for (ta, pa), (tb, pb) in zip(out["1"], out["0"]):
        assert ta == tb, (ta, tb)
        assert abs(pa - pb) <= 1e-5 * abs(pb) + 1e-30, (pa, pb)
    ctx.close()


@pytest.mark.parametrize("N,Kb", [(1024, 5), (300, 1), (87, 3)])
def test_decode_with_input_projection_tables_equals_the_concatenated_operand_form(N, Kb, monkeypatch):
    """Round 6: by default the batched decode's gate GEMMs contract the hidden state only -- h Wh for LSTM-1, [h1 Wproj | h2] for LSTM-2 -- and
    the input-side pre-activations come from tables made once per call: Wembed W1x + b1 per TOKEN, x_cnn W2x + b2 per IMAGE (lrcn.jl:529:
    [x h] W = x Wx + h Wh; :546, :611).  LRCN_DECODE_TABLES=0 is the [x | h] form of round 5.  Same bf16 products, f32 sums in two chains
    instead of one: on decisive distributions the captions agree (a near-tie may fall the other way) and so do the probabilities; the
    table form also agrees with the bf16-emulating oracle's per-image decode.  300 x 1 / 87 x 3: row-block tails, beam widths 1 and 3."""
    m = decisive_model(seed=6)
    feats = (np.random.default_rng(N + Kb).standard_normal((N, 4096)) * 0.05).astype(np.float32)
    ctx = L.Context(E, H, H, V, max_B=N * Kb, max_T=2, lstm_dtype=lrcn_amd.LRCN_BF16)
    param = L.model_from_arrays(m.p)
    out = {}
    for knob in ("1", "0", "1"):
        monkeypatch.setenv("LRCN_DECODE_TABLES", knob)
        r = L.beam_search_batch(ctx, param, L.to_jl(feats), Kb, NWORD)
        if knob in out:
            assert [t for t, _ in r] == [t for t, _ in out[knob]], "the table decode does not repeat itself"
        out[knob] = r
    same = sum(a[0] == b[0] for a, b in zip(out["1"], out["0"]))
    assert same >= N - max(1, N // 50), (same, N)
    for (ta, pa), (tb, pb) in zip(out["1"], out["0"]):
        if ta == tb:
            assert abs(pa - pb) <= 2e-2 * abs(pb) + 1e-30, (pa, pb)
        else:
            assert abs(np.log(pa + 1e-300) - np.log(pb + 1e-300)) < 0.3, (pa, pb)
    agree = 0
    picks = [0, N // 2, N - 1]
    for i in picks:
        with orc.emulate_bf16():
            rt, rp = orc.beam_search(m, feats[i], Kb, NWORD)
        if out["1"][i][0] == list(rt):
            agree += 1
            assert abs(out["1"][i][1] - rp) <= 5e-2 * abs(rp) + 1e-30
    assert agree >= len(picks) - 1, agree
    ctx.close()

