"""CPU: the C oracle (oracle/lrcn_oracle.c) against the committed golden vectors (torch-autograd transcription,
tests/golden/make_golden.py), finite differences and analytic known answers.  No GPU, no HIP library."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as orc

CASES = ["lstm_tiny", "lstm_tiny_drop", "lstm_ragged", "lstm_mid"]


CASES1 = ["lstm1_tiny", "lstm1_drop", "lstm1_mid"]  # LRCN-1f (BASELINE configs[1]; this repo's definition, lrcn_oracle.h)


def load_case(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    E, H1, H2, V = (int(z[k]) for k in ("E", "H1", "H2", "V"))
    nl = int(z["n_layers"]) if "n_layers" in z else 2
    model = orc.Model(E, H1, H2, V, {n: z["p_" + n] for n in orc.PARAM_NAMES}, n_layers=nl)
    return z, model


@pytest.mark.parametrize("name", CASES + CASES1)
def test_loss_and_grads_match_golden(golden_dir, name):
    z, model = load_case(golden_dir, name)
    m1 = z["mask1"] if "mask1" in z else None
    m2 = z["mask2"] if "mask2" in z else None
    val, g = orc.loss(model, z["feats"], z["tokens"], norm_B=int(z["norm_B"]), mask1=m1, mask2=m2, want_grad=True)
    assert abs(val - float(z["loss"])) <= 1e-6 * abs(float(z["loss"]))
    for n in orc.PARAM_NAMES:
        ref = z["g_" + n]
        np.testing.assert_allclose(g.p[n], ref, rtol=1e-4, atol=1e-7, err_msg=n)


@pytest.mark.parametrize("name", ["lstm_tiny", "lstm_mid", "lstm1_tiny", "lstm1_mid"])
def test_logits_match_golden(golden_dir, name):
    z, model = load_case(golden_dir, name)
    got = orc.forward_logits(model, z["feats"], z["tokens"])
    np.testing.assert_allclose(got, z["logits"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", ["lstm_tiny", "lstm_ragged", "lstm1_tiny", "lstm1_drop"])
def test_adam_trajectory_matches_golden(golden_dir, name):
    z, model = load_case(golden_dir, name)
    mom = {n: np.zeros_like(model.p[n]) for n in orc.PARAM_NAMES}
    var = {n: np.zeros_like(model.p[n]) for n in orc.PARAM_NAMES}
    for t, ref_loss in enumerate(z["adam_losses"], start=1):
        val, g = orc.loss(model, z["feats"], z["tokens"], norm_B=int(z["norm_B"]), mask1=z["mask1"] if "mask1" in z else None,
                          mask2=z["mask2"] if "mask2" in z else None, want_grad=True)
        assert abs(val - ref_loss) <= 2e-6 * abs(ref_loss)
        for n in orc.PARAM_NAMES:
            orc.adam(model.p[n], g.p[n], mom[n], var[n], t)
    for n in orc.PARAM_NAMES:
        np.testing.assert_allclose(model.p[n], z["a_" + n], rtol=0, atol=2e-6, err_msg=n)


@pytest.mark.parametrize("name", ["lstm_tiny", "lstm_ragged", "lstm_mid", "lstm1_tiny", "lstm1_mid"])
def test_beam_search_matches_golden(golden_dir, name):
    z, model = load_case(golden_dir, name)
    K, nword = int(z["beam_K"]), int(z["beam_nword"])
    for i, (ref, rp) in enumerate(zip(z["beam_tokens"], z["beam_prob"])):
        seq, p = orc.beam_search(model, z["feats"][i], K, nword)
        ref = ref[ref >= 0]
        assert list(seq) == list(ref), (i, seq, ref)
        assert abs(p - rp) <= 1e-5 * abs(rp)
        assert seq[0] == orc.BOS and len(seq) <= nword + 2


def test_finite_difference_gradient():
    rng = np.random.default_rng(0)
    E, H1, H2, V, B, T = 6, 5, 4, 11, 3, 3
    model = orc.init_weights(E, H1, H2, V, seed=7)
    feats = (rng.standard_normal((B, 4096)) * 0.05).astype(np.float32)
    tokens = rng.integers(0, V, size=(T, B)).astype(np.int32)
    _, g = orc.loss(model, feats, tokens, want_grad=True)
    for n in orc.PARAM_NAMES:
        a = model.p[n]
        flat = a.reshape(-1, order="F")
        for idx in rng.choice(flat.size, size=min(6, flat.size), replace=False):
            i, j = np.unravel_index(idx, a.shape, order="F")
            old = a[i, j]
            eps = 2e-2
            a[i, j] = old + eps
            lp = orc.loss(model, feats, tokens)
            a[i, j] = old - eps
            lm = orc.loss(model, feats, tokens)
            a[i, j] = old
            fd = (lp - lm) / (2 * eps)
            assert abs(fd - g.p[n][i, j]) <= 3e-3 * max(1e-3, abs(fd)) + 2e-5, (n, i, j, fd, g.p[n][i, j])


def test_zero_logit_loss_is_ln_V():
    # Analytic known answer: Wout = bout = 0 -> uniform softmax -> loss = ln V.  The reference's deck plots exactly
    # this at epoch 0 (8.9528 = ln 7730 for Flickr30k, 9.2723 = ln 10640 for COCO).
    V = 173
    model = orc.init_weights(8, 8, 8, V, seed=3)
    model.p["Wout"][:] = 0
    model.p["bout"][:] = 0
    rng = np.random.default_rng(1)
    feats = rng.standard_normal((5, 4096)).astype(np.float32)
    tokens = rng.integers(0, V, size=(4, 5)).astype(np.int32)
    assert abs(orc.loss(model, feats, tokens) - np.log(V)) < 1e-6
    assert abs(np.log(7730) - 8.9528) < 1e-4 and abs(np.log(10640) - 9.2723) < 1e-4


def test_lstm_zero_weights_analytic():
    # W = 0, forget bias 1 (initweights, lrcn.jl:501): c' = c*sigm(1) + sigm(0)*tanh(0) = c*sigm(1); h' = 0.5*tanh(c')
    B, X, H = 3, 4, 5
    W = np.zeros((X + H, 4 * H), np.float32)
    b = np.zeros((1, 4 * H), np.float32)
    b[0, :H] = 1
    rng = np.random.default_rng(2)
    c = rng.standard_normal((B, H)).astype(np.float32)
    h, cn = orc.lstm(W, b, rng.standard_normal((B, X)), rng.standard_normal((B, H)), c)
    s1 = 1 / (1 + np.exp(-1.0))
    np.testing.assert_allclose(cn, c * s1, rtol=1e-6)
    np.testing.assert_allclose(h, 0.5 * np.tanh(c * s1), rtol=1e-6, atol=1e-7)


def test_init_weights_distribution():
    m = orc.init_weights(64, 32, 48, 501, seed=42)
    for n, a in m.p.items():
        if n.startswith("b"):
            continue
        s = np.sqrt(2.0 / sum(a.shape))
        assert np.abs(a).max() <= s * (1 + 1e-6) and abs(a.mean()) < 0.05 * s
        assert abs(a.std() - s / np.sqrt(3)) < 0.05 * s
    assert (m.p["b1"][0, :32] == 1).all() and (m.p["b1"][0, 32:] == 0).all()
    assert (m.p["b2"][0, :48] == 1).all() and (m.p["bout"] == 0).all()


def test_dp_shards_sum_to_full_batch(golden_dir):
    # N shards normalised by the GLOBAL batch, gradients summed == the full-batch step.
    z, model = load_case(golden_dir, "lstm_mid")
    feats, tokens = z["feats"], z["tokens"]
    B = feats.shape[0]
    full, gfull = orc.loss(model, feats, tokens, want_grad=True)
    tot, acc = 0.0, {n: 0 for n in orc.PARAM_NAMES}
    for r in range(4):
        sl = slice(r * B // 4, (r + 1) * B // 4)
        v, g = orc.loss(model, feats[sl], tokens[:, sl], norm_B=B, want_grad=True)
        tot += v
        for n in orc.PARAM_NAMES:
            acc[n] = acc[n] + g.p[n].astype(np.float64)
    assert abs(tot - full) < 1e-6 * abs(full)
    for n in orc.PARAM_NAMES:
        np.testing.assert_allclose(acc[n], gfull.p[n], rtol=1e-4, atol=1e-7)


def test_cnn_ops_match_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "cnn_small.npz"))
    y = orc.conv3x3(z["x"], z["w"], z["b"], relu=True)
    np.testing.assert_allclose(y, z["y"], rtol=1e-5, atol=1e-6)
    yp = orc.pool2(y)
    np.testing.assert_allclose(yp, z["yp"], rtol=1e-5, atol=1e-6)
    N = yp.shape[3]
    f6 = orc.fc(z["w6"], z["b6"], yp.reshape(-1, N, order="F"), relu=False)
    np.testing.assert_allclose(f6, z["f6"], rtol=1e-5, atol=1e-5)


def test_preprocess_u8_layout():
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, size=(2, 8, 8, 3), dtype=np.uint8)
    mean = np.array([123.68, 116.779, 103.939], np.float32)
    out = orc.preprocess_u8(img, mean)
    assert out.shape == (8, 8, 3, 2)
    # out(i,j,c,n) = pixel(row=i, col=j, c) - mean[c]  (lrcn.jl:766-772)
    ref = np.transpose(img.astype(np.float32), (1, 2, 3, 0)) - mean[None, None, :, None]
    np.testing.assert_array_equal(out, ref)


def test_oracle_sanitizer_build_runs_clean():
    # AddressSanitizer + UBSan pass over every oracle entry point (oracle/selftest.c); CPU only by necessity
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    r = subprocess.run(["make", "-s", "-C", here, "-f", os.path.join(here, "Makefile"), "asan"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "oracle selftest ok" in r.stdout


def test_fast_baseline_build_matches_the_checker_build():
    # liblrcn_oracle_f32.so (bench.py's timed cpu_baseline: float accumulation, convolutions as im2col + blocked SGEMM, re-ordered
    # reverse-pass GEMMs) against the double-accumulating checker on the same inputs: same arithmetic, another summation order
    import ctypes as C
    if not orc._has_avx2():
        pytest.skip("no AVX2/FMA: the fast build is not used on this host")
    rng = np.random.default_rng(3)
    for S, Cin, Cout, N in [(14, 64, 72, 2), (10, 3, 64, 1), (30, 40, 130, 1)]:   # pixel-block, channel-block and K-block tails
        x = rng.standard_normal((S, S, Cin, N)).astype(np.float32)
        w = (rng.standard_normal((3, 3, Cin, Cout)) * 0.1).astype(np.float32)
        b = rng.standard_normal(Cout).astype(np.float32)
        ref = orc.conv3x3(x, w, b, relu=True)
        xf, wf, bf = orc.fa(x), orc.fa(w), orc.fa(b)
        y = np.zeros((S, S, Cout, N), np.float32, order="F")
        orc.lib(True).orc_conv3x3(orc._f(xf), S, S, Cin, N, orc._f(wf), orc._f(bf), Cout, 1, orc._f(y))
        assert np.abs(y - ref).max() <= 1e-5 * np.abs(ref).max()
    E, H1, H2, V, B, T = 24, 40, 32, 97, 5, 4
    m = orc.init_weights(E, H1, H2, V, seed=2)
    feats = (rng.standard_normal((B, 4096)) * 0.05).astype(np.float32)
    tokens = rng.integers(0, V, size=(T, B)).astype(np.int32)
    l0, g0 = orc.loss(m, feats, tokens, want_grad=True)
    l1, g1 = orc.loss(m, feats, tokens, want_grad=True, fast=True)
    assert abs(l1 - l0) <= 1e-6 * abs(l0)
    for n in orc.PARAM_NAMES:
        np.testing.assert_allclose(g1.p[n], g0.p[n], rtol=1e-3, atol=1e-6, err_msg=n)


# ---- ORC_EMULATE_BF16 (oracle/lrcn_oracle.h): the rounding points of the HIP library's bf16 arithmetic, restated ----
def test_bf16_round_is_round_to_nearest_even():
    cases = np.array([1.0, 1.00390625, 1.005859375, 1.001953125, 1.998046875, -3.1415927, 1e-40, 65504.0, 3.3895314e38, 0.0, -0.0], np.float32)
    got = orc.bf16_round(cases)
    want = torch.as_tensor(cases).to(torch.bfloat16).to(torch.float32).numpy()
    np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(200000) * np.exp(rng.uniform(-30, 30, 200000))).astype(np.float32)
    np.testing.assert_array_equal(orc.bf16_round(x), torch.as_tensor(x).to(torch.bfloat16).to(torch.float32).numpy())
    for v in (1.0, 1.005859375, -2.71828, 1e-3):
        assert orc.lib().orc_bf16_round(v) == orc.bf16_round(np.float32(v))
    assert np.isnan(orc.bf16_round(np.array([np.nan], np.float32)))[0]


def _np_forward_emulated(m, feats, tokens, mask1, mask2):
    """Forward pass of loss() with the bf16 rounding points, written independently of the C code (numpy, float64 contractions):
    per-step logits (T+1, B, V).  Statement of WHERE the HIP library rounds (lrcn_oracle.h lists the same points)."""
    rb = orc.bf16_round
    f64 = np.float64
    E, H1, H2, V = m.E, m.H1, m.H2, m.V
    hh = H2 // 2
    W = {n: rb(m.p[n]).astype(f64) for n in ("W1", "W2", "Wproj", "Wcnn", "Wembed", "Wout")}
    T, B = tokens.shape
    sig = lambda z: (np.float32(1) / (np.float32(1) + np.exp(-z, dtype=np.float32))).astype(np.float32)
    xcnn = (rb(feats).astype(f64) @ W["Wcnn"]).astype(np.float32)                     # f32, not rounded
    h1 = np.zeros((B, H1), np.float32); c1 = np.zeros((B, H1), np.float32)
    h2 = np.zeros((B, H2), np.float32); c2 = np.zeros((B, H2), np.float32)
    out = []
    for s in range(T + 1):
        inp = np.full(B, orc.BOS) if s == 0 else tokens[s - 1]
        x1 = rb(W["Wembed"][inp].astype(np.float32) * (mask1[s] if mask1 is not None else np.float32(1)))
        g = (np.hstack([x1, rb(h1)]).astype(f64) @ W["W1"]).astype(np.float32) + m.p["b1"]
        f, i, o, ch = sig(g[:, :H1]), sig(g[:, H1:2 * H1]), sig(g[:, 2 * H1:3 * H1]), np.tanh(g[:, 3 * H1:])
        c1 = c1 * f + i * ch
        h1 = o * np.tanh(c1)
        left = rb((rb(h1).astype(f64) @ W["Wproj"]).astype(np.float32))
        x2 = np.hstack([left, xcnn])
        x2 = rb(x2 * (mask2[s] if mask2 is not None else np.float32(1)))
        g = (np.hstack([x2, rb(h2)]).astype(f64) @ W["W2"]).astype(np.float32) + m.p["b2"]
        f, i, o, ch = sig(g[:, :H2]), sig(g[:, H2:2 * H2]), sig(g[:, 2 * H2:3 * H2]), np.tanh(g[:, 3 * H2:])
        c2 = c2 * f + i * ch
        h2 = o * np.tanh(c2)
        out.append((rb(h2).astype(f64) @ W["Wout"]).astype(np.float32) + m.p["bout"])
    return np.stack(out)


def test_emulated_forward_equals_independent_numpy_statement_and_off_is_untouched():
    rng = np.random.default_rng(4)
    E, H1, H2, V, B, T = 40, 48, 32, 97, 6, 4
    m = orc.init_weights(E, H1, H2, V, seed=3)
    for n in ("W1", "W2", "Wout"):
        m.p[n] *= 3.0
    feats = (rng.standard_normal((B, 4096)) * 0.05).astype(np.float32)
    tokens = rng.integers(0, V, size=(T, B)).astype(np.int32)
    plain = orc.forward_logits(m, feats, tokens)
    l_plain, g_plain = orc.loss(m, feats, tokens, want_grad=True)
    with orc.emulate_bf16():
        emu = orc.forward_logits(m, feats, tokens)
        l_emu, g_emu = orc.loss(m, feats, tokens, want_grad=True)
        assert orc.lib().orc_get_emulate_bf16() == 1
    assert orc.lib().orc_get_emulate_bf16() == 0
    np.testing.assert_array_equal(plain, orc.forward_logits(m, feats, tokens))          # the switch leaves no trace
    l_again, g_again = orc.loss(m, feats, tokens, want_grad=True)
    assert l_again == l_plain and all(np.array_equal(g_again.p[n], g_plain.p[n]) for n in orc.PARAM_NAMES)
    want = _np_forward_emulated(m, feats, tokens, None, None)
    np.testing.assert_allclose(emu, want, rtol=2e-6, atol=2e-6)
    d = np.abs(emu - plain).max() / np.abs(plain).max()
    assert 1e-4 < d < 3e-2, d                                                           # bf16 is visible, and small
    assert abs(l_emu - l_plain) <= 2e-2 * abs(l_plain) and l_emu != l_plain
    for n in orc.PARAM_NAMES:                                                           # gradients: same direction, not the same numbers
        a, b = g_emu.p[n].ravel().astype(np.float64), g_plain.p[n].ravel().astype(np.float64)
        assert a @ b / (np.linalg.norm(a) * np.linalg.norm(b)) > 0.999, n
        assert not np.array_equal(g_emu.p[n], g_plain.p[n]), n


def test_emulated_forward_with_dropout_masks_and_gradient_by_finite_differences_of_the_bias():
    # masks: the double rounding bf16(bf16(.) * multiplier) of both halves of LSTM-2's input; d loss / d bout is exact under emulation
    # (bout enters after the last rounding), so finite differences of the EMULATED loss must reproduce the emulated gradient
    rng = np.random.default_rng(9)
    E, H1, H2, V, B, T = 24, 32, 32, 53, 5, 3
    m = orc.init_weights(E, H1, H2, V, seed=8)
    feats = (rng.standard_normal((B, 4096)) * 0.05).astype(np.float32)
    tokens = rng.integers(0, V, size=(T, B)).astype(np.int32)
    mask1 = ((rng.random((T + 1, B, E)) > 0.4) / 0.6).astype(np.float32)
    mask2 = ((rng.random((T + 1, B, H2)) > 0.4) / 0.6).astype(np.float32)
    with orc.emulate_bf16():
        l0, g = orc.loss(m, feats, tokens, mask1=mask1, mask2=mask2, want_grad=True)
        want_logits = _np_forward_emulated(m, feats, tokens, mask1, mask2)
        # loss from the independent logits
        tg = np.vstack([tokens, np.zeros((1, B), np.int32)])
        z = want_logits.astype(np.float64)
        lse = np.log(np.exp(z - z.max(-1, keepdims=True)).sum(-1)) + z.max(-1)
        l_np = -(np.take_along_axis(z, tg[..., None], -1)[..., 0] - lse).sum() / (B * (T + 1))
        assert abs(l0 - l_np) <= 2e-6 * abs(l_np)
        eps = 1e-2
        for j in (0, 7, 31):
            m.p["bout"][0, j] += eps
            lp = orc.loss(m, feats, tokens, mask1=mask1, mask2=mask2)
            m.p["bout"][0, j] -= 2 * eps
            lm = orc.loss(m, feats, tokens, mask1=mask1, mask2=mask2)
            m.p["bout"][0, j] += eps
            # dlogits is stored as bf16 (3 significant digits) before its column sum: the gradient agrees to that
            assert abs((lp - lm) / (2 * eps) - g.p["bout"][0, j]) <= 6e-3 * abs(g.p["bout"][0, j]) + 1e-5


def test_emulated_conv_layer_rounds_operands_and_result():
    rng = np.random.default_rng(2)
    x = rng.standard_normal((6, 6, 5, 2)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 5, 4)) * 0.3).astype(np.float32)
    b = rng.standard_normal(4).astype(np.float32)
    plain = orc.conv3x3(orc.bf16_round(x), orc.bf16_round(w), b, relu=True)   # bf16 operands, exact accumulation, f32 result
    with orc.emulate_bf16():
        emu = orc.conv3x3(x, w, b, relu=True)
    np.testing.assert_array_equal(emu, orc.bf16_round(plain))
    assert not np.array_equal(emu, orc.conv3x3(x, w, b, relu=True))
