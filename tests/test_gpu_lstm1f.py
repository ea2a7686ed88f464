"""GPU: LRCN-1f -- BASELINE configs[1] "VGG-16 + 1-layer LSTM-512 fp32, batch 32" (lrcn_config.n_layers = 1; this repo's
definition, SURVEY 8d: the reference hard-wires two layers) through the C ABI against the torch-autograd golden vectors
(tests/golden/lstm1_*.npz) and the CPU oracle (orc1_*).  fp32 tolerances as for the two-layer model: loss 1e-5 relative,
gradients 1e-5 + 1e-3 |g|; bf16: loss 2e-2."""
import os

import numpy as np
import pytest
import torch

import lrcn_amd
from lrcn_amd import lrcn as L
from oracle import oracle as orc
from parity_util import assert_bf16_matches_emulation, emulated_reference

pytestmark = pytest.mark.gpu

CASES = ["lstm1_tiny", "lstm1_drop", "lstm1_mid"]


def load_case(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    assert int(z["n_layers"]) == 1
    return z, tuple(int(z[k]) for k in ("E", "H1", "H2", "V"))


def make_ctx(dims, B, T, dtype=lrcn_amd.LRCN_F32, **kw):
    E, H1, H2, V = dims
    return L.Context(E, H1, H2, V, max_B=B, max_T=T, lstm_dtype=dtype, n_layers=1, **kw)


def grads_close(got, ref, rtol=1e-3, atol=1e-5):
    for n, g, r in zip(orc.PARAM_NAMES, got, ref):
        if r.size == 0:
            assert g.numel() == 0, n
            continue
        np.testing.assert_allclose(L.from_jl(g), r, rtol=rtol, atol=atol, err_msg=n)


@pytest.mark.parametrize("name", CASES)
def test_loss_and_grads_vs_golden_fp32(golden_dir, name):
    z, dims = load_case(golden_dir, name)
    T, B = z["tokens"].shape
    ctx = make_ctx(dims, B, T)
    param = L.model_from_arrays({n: z["p_" + n] for n in orc.PARAM_NAMES})
    assert [tuple(t.shape) for t in param] == L.param_shapes(*dims, n_layers=1)
    m1 = z["mask1"] if "mask1" in z else None
    feats = L.to_jl(z["feats"])
    val = L.loss(ctx, param, feats, z["tokens"], norm_B=int(z["norm_B"]), mask1=m1)
    assert abs(val - float(z["loss"])) <= 1e-5 * abs(float(z["loss"]))
    grads, val2 = L.lossgradient(ctx, param, feats, z["tokens"], norm_B=int(z["norm_B"]), mask1=m1)
    assert abs(val2 - float(z["loss"])) <= 1e-5 * abs(float(z["loss"]))
    grads_close(grads, [z["g_" + n] for n in orc.PARAM_NAMES])
    got = L.forward_logits(ctx, param, feats, z["tokens"]) if m1 is None else None
    if got is not None:
        np.testing.assert_allclose(got, z["logits"], rtol=1e-4, atol=2e-5)
    ctx.close()


@pytest.mark.parametrize("name", ["lstm1_tiny", "lstm1_mid"])
def test_train_step_adam_trajectory_vs_golden(golden_dir, name):
    z, dims = load_case(golden_dir, name)
    T, B = z["tokens"].shape
    ctx = make_ctx(dims, B, T)
    param = L.model_from_arrays({n: z["p_" + n] for n in orc.PARAM_NAMES})
    opt = L.initparams(param)
    grads = L.zeros_like_model(param)
    for ref_loss in z["adam_losses"]:
        val = L.train_step(ctx, param, opt, grads, L.to_jl(z["feats"]), z["tokens"], norm_B=int(z["norm_B"]), pdrop=0.0, want_loss=True)
        assert abs(val - ref_loss) <= 2e-5 * abs(ref_loss)
    for n, p in zip(orc.PARAM_NAMES, param):
        if p.numel():
            np.testing.assert_allclose(L.from_jl(p), z["a_" + n], rtol=0, atol=5e-6, err_msg=n)
    ctx.close()


@pytest.mark.parametrize("name", ["lstm1_tiny", "lstm1_mid"])
def test_beam_search_vs_golden_single_and_batched(golden_dir, name):
    z, dims = load_case(golden_dir, name)
    K, nword = int(z["beam_K"]), int(z["beam_nword"])
    n = len(z["beam_tokens"])
    ctx = make_ctx(dims, max(n * K, 4), 4)
    param = L.model_from_arrays({n_: z["p_" + n_] for n_ in orc.PARAM_NAMES})
    for i, (ref, rp) in enumerate(zip(z["beam_tokens"], z["beam_prob"])):
        seq, p = L.beam_search(ctx, param, L.to_jl(z["feats"][i:i + 1]), K, nword)
        assert seq == list(ref[ref >= 0]), (i, seq, ref)
        assert abs(p - rp) <= 1e-4 * abs(rp)
    batch = L.beam_search_batch(ctx, param, L.to_jl(z["feats"][:n]), K, nword)
    for i, (ref, rp) in enumerate(zip(z["beam_tokens"], z["beam_prob"])):
        assert batch[i][0] == list(ref[ref >= 0]) and abs(batch[i][1] - rp) <= 1e-4 * abs(rp)
    ctx.close()


def test_single_lstm_and_step_vs_oracle():
    rng = np.random.default_rng(17)
    E, H, V, B = 24, 32, 57, 5
    h = H // 2
    m = orc.init_weights(E, H, H, V, seed=9, n_layers=1)
    ctx = make_ctx((E, H, H, V), B, 2)
    param = L.model_from_arrays(m.p)
    # lstm() entry on the single layer's (X = E + h, H)
    x = rng.standard_normal((B, E + h)).astype(np.float32)
    h0 = rng.standard_normal((B, H)).astype(np.float32) * 0.5
    c0 = rng.standard_normal((B, H)).astype(np.float32) * 0.5
    ho, co = L.lstm(ctx, param[0], param[1], L.to_jl(h0), L.to_jl(c0), L.to_jl(x))
    rh, rc = orc.lstm(m.p["W1"], m.p["b1"], x, h0, c0)
    np.testing.assert_allclose(L.from_jl(ho), rh, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(L.from_jl(co), rc, rtol=1e-5, atol=1e-6)
    # the 1f step with an explicit mask over hcat(x_lstm, x_cnn), state carried across two calls
    state_ref = [orc.fa(rng.standard_normal((B, H)).astype(np.float32) * 0.3) for _ in range(2)]
    state = [L.to_jl(s) for s in state_ref]
    x_cnn = rng.standard_normal((B, h)).astype(np.float32)
    for _ in range(2):
        x_lstm = rng.standard_normal((B, E)).astype(np.float32)
        mk = ((rng.random((B, E + h)) > 0.4) / 0.6).astype(np.float32)
        got = L.lrcn(ctx, param, state, L.to_jl(x_cnn), L.to_jl(x_lstm), L.to_jl(mk), None)
        ref = orc.lrcn_step(m, state_ref, x_cnn, x_lstm, mk)
        np.testing.assert_allclose(L.from_jl(got), ref, rtol=1e-4, atol=1e-5)
        for a, b in zip(state, state_ref):
            np.testing.assert_allclose(L.from_jl(a), b, rtol=1e-4, atol=1e-5)
    assert len(L.initstate(ctx, B)) == 2
    ctx.close()


def test_config2_shape_b32_vs_oracle_fp32_and_bf16():
    # BASELINE configs[1] at its stated size: 1-layer LSTM-512, batch 32, V = 2540, T = 11 (the VGG front is covered by
    # test_config2_end_to_end_1f_fp32_images_to_loss below and by the full-VGG tests); initweights on the device, oracle on its copy
    rng = np.random.default_rng(5)
    E = H = 512
    V, B, T = 2540, 32, 11
    ctx = make_ctx((E, H, H, V), B, T)
    param = L.initweights(ctx, seed=42)
    sizes = [int(np.prod(s)) for s in L.param_shapes(E, H, H, V, n_layers=1)]
    assert sum(sizes) == (E + H // 2 + H) * 4 * H + 4 * H + 4096 * (H // 2) + V * E + H * V + V
    b1 = L.from_jl(param[1])
    assert (b1[0, :H] == 1).all() and (b1[0, H:] == 0).all()
    s0 = np.sqrt(2.0 / (E + H // 2 + H + 4 * H))
    assert abs(L.from_jl(param[0]).std() - s0 / np.sqrt(3)) < 0.02 * s0
    m = orc.Model(E, H, H, V, {n: L.from_jl(t) for n, t in zip(orc.PARAM_NAMES, param)}, n_layers=1)
    feats = (rng.standard_normal((B, 4096)) * 0.01).astype(np.float32)
    tokens = rng.integers(3, V, size=(T, B)).astype(np.int32)
    ref_loss, ref_g = orc.loss(m, feats, tokens, want_grad=True)
    grads, val = L.lossgradient(ctx, param, L.to_jl(feats), tokens)
    assert abs(val - ref_loss) <= 1e-5 * abs(ref_loss)
    grads_close(grads, [ref_g.p[n] for n in orc.PARAM_NAMES])
    ctx.close()
    ctx16 = make_ctx((E, H, H, V), B, T, dtype=lrcn_amd.LRCN_BF16)
    grads16, val16 = L.lossgradient(ctx16, param, L.to_jl(feats), tokens)
    assert abs(val16 - ref_loss) <= 2e-2 * abs(ref_loss)
    emu_loss, emu_g = emulated_reference(m, feats, tokens)  # elementwise against the bf16-emulating oracle (tests/parity_util.py)
    assert_bf16_matches_emulation(val16, [g for g in grads16], emu_loss, emu_g, "LRCN-1f config-2 shape")
    mask = ((rng.random((T + 1, B, E + H // 2)) > 0.4) / 0.6).astype(np.float32)  # one mask over hcat(embedding, x_cnn)
    emu_loss, emu_g = emulated_reference(m, feats, tokens, mask1=mask)
    gm, vm = L.lossgradient(ctx16, param, L.to_jl(feats), tokens, mask1=mask)
    assert_bf16_matches_emulation(vm, [g for g in gm], emu_loss, emu_g, "LRCN-1f config-2 shape, explicit dropout mask")
    # generated dropout: the loss stays finite and differs from the no-dropout loss; same seed -> same loss
    l1 = L.loss(ctx16, param, L.to_jl(feats), tokens, pdrop=0.4, seed=3)
    l2 = L.loss(ctx16, param, L.to_jl(feats), tokens, pdrop=0.4, seed=3)
    assert np.isfinite(l1) and l1 == l2 and l1 != val16 and abs(l1 - val16) < 0.5
    ctx16.close()


def test_config2_end_to_end_1f_fp32_images_to_loss():
    # configs[1] composed end to end on the parity scale: uint8 crops -> VGG-16 fp32 (exact-fp32 MFMA) -> sum-normalised fc7
    # -> LRCN-1f LSTM-512 loss and gradients, against the oracle run on the same crops
    w = L.synthetic_vgg_weights(seed=1, bias_std=0.05)
    host = ([L.from_jl(t) for t in w[0]], [t.cpu().numpy() for t in w[1]], (L.from_jl(w[2][0]), w[2][1].cpu().numpy()),
            (L.from_jl(w[3][0]), w[3][1].cpu().numpy()))
    rng = np.random.default_rng(77)
    B, E, H, V, T = 2, 512, 512, 2540, 11
    img = rng.integers(0, 256, size=(B, 224, 224, 3), dtype=np.uint8)
    ref_f = orc.vgg_forward(host[0], host[1], host[2], host[3], orc.preprocess_u8(img, np.array(L.VGG_MEAN, np.float32)))
    fn = (ref_f / ref_f.sum(axis=1, keepdims=True)).astype(np.float32)
    m = orc.init_weights(E, H, H, V, seed=42, n_layers=1)
    tokens = rng.integers(3, V, size=(T, B)).astype(np.int32)
    ref_loss, ref_g = orc.loss(m, fn, tokens, want_grad=True)
    ctx = make_ctx((E, H, H, V), B, T, vgg_dtype=lrcn_amd.LRCN_F32, max_images=B)
    L.vgg_load(ctx, *w)
    feats = L.from_jl(L.convnet_u8(ctx, torch.as_tensor(img).cuda()))
    assert np.abs(feats - ref_f).max() <= 1e-4 * np.abs(ref_f).max()
    feats = (feats / feats.sum(axis=1, keepdims=True)).astype(np.float32)
    grads, val = L.lossgradient(ctx, L.model_from_arrays(m.p), L.to_jl(feats), tokens)
    assert abs(val - ref_loss) <= 1e-4 * abs(ref_loss), (val, ref_loss)
    for n, g in zip(orc.PARAM_NAMES, grads):
        if g.numel() == 0:
            continue
        a, b = L.from_jl(g).ravel().astype(np.float64), ref_g.p[n].ravel().astype(np.float64)
        assert np.linalg.norm(a - b) <= 2e-3 * np.linalg.norm(b) + 1e-9, n
    ctx.close()


def test_one_layer_config_validation():
    with pytest.raises(lrcn_amd.LrcnError):
        L.Context(16, 16, 32, 40, max_B=2, max_T=2, n_layers=1)  # H1 != H2
    with pytest.raises(lrcn_amd.LrcnError):
        L.Context(16, 16, 16, 40, max_B=2, max_T=2, n_layers=3)
