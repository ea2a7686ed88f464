"""GPU: BASELINE.json configs[2] ("C3": Flickr30k-shaped, VGG-16 bf16 + LSTM-1000 bf16, batch 128, V = 7730, T = 12) at ITS OWN shape
(until round 5 only the forward loss ran at this shape).  The training half -- lossgradient (lrcn.jl:553-583), the
VGG forward at 128 images (lrcn.jl:733-748) and update! (lrcn.jl:394, 399-405) -- against the CPU oracle:

  * lossgradient on a 32-row subset of the 128-row batch, normalised by the GLOBAL batch 128, dropout 0.4 through explicit masks:
    elementwise vs the bf16-emulating oracle AND per tensor vs the plain f32 oracle (tests/parity_util.py);
  * the FULL 128 rows (the route the C3 bench line runs: lstm_fused<4>, two row blocks) through the shard-sum property: its loss and
    gradients equal the sum of the four 32-row shards -- one of which the oracle checked above -- each normalised by 128;
  * the 13-layer + fc6 + fc7 stack at N = 128 images on the bench's capped persistent grids: six images against the oracle's f32 stack;
  * one lrcn_train_step at 128 rows: the parameters it leaves behind equal Knet's Adam formula applied on the host to the
    gradients the same call wrote, every one of the 34 023 730 parameters, to float32 rounding."""
import numpy as np
import pytest
import torch

import lrcn_amd
from lrcn_amd import lrcn as L
from oracle import oracle as orc
from parity_util import assert_bf16_matches_emulation, emulated_reference, worst_vs_f32

pytestmark = pytest.mark.gpu

E = H = 1000
V, B, T = 7730, 128, 12
SHARD = 32


@pytest.fixture(scope="module")
def c3():
    rng = np.random.default_rng(128)
    m = orc.init_weights(E, H, H, V, seed=42)
    feats = (rng.standard_normal((B, 4096)) * 0.05).astype(np.float32)
    pz = 1.0 / np.arange(1, V - 3 + 1)
    tokens = (rng.choice(V - 3, size=(T, B), p=pz / pz.sum()) + 3).astype(np.int32)   # Zipf(1.0) word ids >= 3
    mask1 = ((rng.random((T + 1, B, E)) > 0.4) / 0.6).astype(np.float32)              # dropout 0.4, device-independent
    mask2 = ((rng.random((T + 1, B, H)) > 0.4) / 0.6).astype(np.float32)
    return m, feats, tokens, mask1, mask2


def _shard(c3, r):
    m, feats, tokens, mask1, mask2 = c3
    rows = slice(SHARD * r, SHARD * (r + 1))
    return feats[rows], np.ascontiguousarray(tokens[:, rows]), np.ascontiguousarray(mask1[:, rows]), np.ascontiguousarray(mask2[:, rows])


def test_c3_lossgradient_on_32_of_128_rows_vs_both_oracles(c3):
    m = c3[0]
    f, t, m1, m2 = _shard(c3, 1)
    emu_loss, emu_g = emulated_reference(m, f, t, norm_B=B, mask1=m1, mask2=m2)
    ctx = L.Context(E, H, H, V, max_B=B, max_T=T, lstm_dtype=lrcn_amd.LRCN_BF16)
    grads, val = L.lossgradient(ctx, L.model_from_arrays(m.p), L.to_jl(f), t, norm_B=B, mask1=m1, mask2=m2)
    assert_bf16_matches_emulation(val, grads, emu_loss, emu_g, "C3 dimensions, rows 32..63 of 128")
    print("C3 shard vs plain f32 oracle: worst tensor %s %.3e" % worst_vs_f32["C3 dimensions, rows 32..63 of 128"])
    ctx.close()


def test_c3_full_batch_of_128_equals_the_sum_of_its_row_shards(c3):
    m, feats, tokens, mask1, mask2 = c3
    ctx = L.Context(E, H, H, V, max_B=B, max_T=T, lstm_dtype=lrcn_amd.LRCN_BF16)
    param = L.model_from_arrays(m.p)
    g_full, loss_full = L.lossgradient(ctx, param, L.to_jl(feats), tokens, mask1=mask1, mask2=mask2)
    g_full = [L.from_jl(g).astype(np.float64) for g in g_full]
    acc, loss_sum = [np.zeros_like(g) for g in g_full], 0.0
    for r in range(B // SHARD):
        f, t, m1, m2 = _shard(c3, r)
        g, l = L.lossgradient(ctx, param, L.to_jl(f), t, norm_B=B, mask1=m1, mask2=m2)
        loss_sum += l
        for a, x in zip(acc, g):
            a += L.from_jl(x)
    # the 128-row call runs other kernel instantiations (row blocks of 64, other split-K shapes) than the 32-row calls: bf16 roundings of
    # intermediate values flip with the f32 summation order, 4e-3 each -- the same bound as the 256-row test of test_gpu_fullsize.py
    assert abs(loss_sum - loss_full) <= 1e-3 * abs(loss_full), (loss_sum, loss_full)
    f32_loss = orc.loss(m, feats, tokens, mask1=mask1, mask2=mask2)
    assert abs(loss_full - f32_loss) <= 2e-2 * abs(f32_loss), (loss_full, f32_loss)   # the stated bf16 tolerance, full batch vs f32 oracle
    for name, a, b in zip(orc.PARAM_NAMES, acc, g_full):
        rel = np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)
        assert rel < 2e-2, (name, rel)
    ctx.close()


def test_c3_vgg_stack_at_128_images_vs_oracle():
    w = L.synthetic_vgg_weights(seed=3, bias_std=0.1)
    host = ([L.from_jl(t) for t in w[0]], [t.cpu().numpy() for t in w[1]], (L.from_jl(w[2][0]), w[2][1].cpu().numpy()),
            (L.from_jl(w[3][0]), w[3][1].cpu().numpy()))
    g = torch.Generator(device="cuda")
    g.manual_seed(30)
    imgs = torch.randint(0, 256, (B, 224, 224, 3), generator=g, device="cuda", dtype=torch.uint8)
    ctx = L.Context(8, 8, 8, 17, max_B=2, max_T=1, vgg_dtype=lrcn_amd.LRCN_BF16, max_images=B)
    L.vgg_load(ctx, *w)
    L.vgg_set_wg_cap(ctx, 224)   # what dp.py sets for the C3 bench line
    got = L.from_jl(L.convnet_u8(ctx, imgs))
    assert np.isfinite(got).all()
    pick = [0, 1, 63, 64, 126, 127]   # both ends, and the images either side of the middle of the batch
    x = orc.preprocess_u8(imgs[pick].cpu().numpy(), np.array(L.VGG_MEAN, np.float32))
    ref = orc.vgg_forward(host[0], host[1], host[2], host[3], x)
    for i, n in enumerate(pick):
        err = float(np.abs(got[n] - ref[i]).max() / (np.abs(ref[i]).max() + 1e-30))
        cos = float((got[n] * ref[i]).sum() / (np.linalg.norm(got[n]) * np.linalg.norm(ref[i])))
        assert err <= 3e-2 and cos > 0.999, (n, err, cos)
    ctx.close()


def test_c3_train_step_at_128_rows_is_knets_adam_on_its_own_gradients(c3):
    m, feats, tokens, _, _ = c3
    ctx = L.Context(E, H, H, V, max_B=B, max_T=T, lstm_dtype=lrcn_amd.LRCN_BF16)
    param = L.model_from_arrays(m.p)
    optim = L.initparams(param)
    grads = [L.jl_empty(*t.shape) for t in param]
    for step in (1, 2):   # the second step exercises non-zero moments and the bias corrections at t = 2
        before = [L.from_jl(p).astype(np.float64) for p in param]
        m0 = [L.from_jl(x).astype(np.float64) for x in optim.m]
        v0 = [L.from_jl(x).astype(np.float64) for x in optim.v]
        val = L.train_step(ctx, param, optim, grads, L.to_jl(feats), tokens, pdrop=0.4, seed=100 + step, want_loss=True)
        ctx.sync()
        assert np.isfinite(val) and optim.t == step
        for k, n in enumerate(orc.PARAM_NAMES):
            g = L.from_jl(grads[k]).astype(np.float64)
            m1 = 0.9 * m0[k] + 0.1 * g
            v1 = 0.999 * v0[k] + 0.001 * g * g
            w1 = before[k] - 1e-3 * (m1 / (1 - 0.9 ** step)) / (np.sqrt(v1 / (1 - 0.999 ** step)) + 1e-8)
            np.testing.assert_allclose(L.from_jl(param[k]), w1, rtol=0, atol=5e-6, err_msg="%s step %d" % (n, step))
            # float32 moments: 1 - 0.999f = 0.00100004673 (4.7e-5 off the decimal), 1 - 0.9f = 0.100000024
            np.testing.assert_allclose(L.from_jl(optim.m[k]), m1, rtol=1e-5, atol=1e-6 * np.abs(m1).max(), err_msg="m %s" % n)
            np.testing.assert_allclose(L.from_jl(optim.v[k]), v1, rtol=2e-4, atol=1e-20, err_msg="v %s" % n)
    ctx.close()
