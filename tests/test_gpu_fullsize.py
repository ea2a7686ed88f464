"""GPU: size-independent properties at BASELINE.json's full sizes (configs[2]/[3] shapes: E = H = 1000, V = 7730 / 10640,
B = 128 / 256, bf16), where the CPU oracle would take minutes:
  * near-zero weights: loss = ln V (the deck's epoch-0 points 8.9528 / 9.2723);
  * sharding: lossgradient on 256 rows == sum over 8 row-shards of 32 normalised by the GLOBAL batch (lrcn.jl:564-568) --
    exactly what the data-parallel ranks compute before the all-reduce, here on one GPU;
  * linearity of the frozen extractor's last layer: VGG features of a batch == features of its halves;
  * Adam: one update! moves every parameter by at most lr (|m/(sqrt(v)+eps)| <= 1 after bias correction at t = 1)."""
import math

import numpy as np
import pytest
import torch

import lrcn_amd
from lrcn_amd import lrcn as L

pytestmark = pytest.mark.gpu


def make(V, B, T, seed=42):
    ctx = L.Context(1000, 1000, 1000, V, max_B=B, max_T=T, lstm_dtype=lrcn_amd.LRCN_BF16)
    param = L.initweights(ctx, seed=seed)
    rng = np.random.default_rng(seed)
    feats = (rng.standard_normal((B, 4096)) * 0.01).astype(np.float32)
    pz = 1.0 / np.arange(1, V - 3 + 1)
    tokens = (rng.choice(V - 3, size=(T, B), p=pz / pz.sum()) + 3).astype(np.int32)
    return ctx, param, feats, tokens


@pytest.mark.parametrize("V,B,T,deck", [(7730, 128, 12, 8.9528), (10640, 256, 11, 9.2723)])
def test_initial_loss_is_ln_V(V, B, T, deck):
    ctx, param, feats, tokens = make(V, B, T)
    for p in param:
        p.mul_(1e-3)  # logits ~ 0 -> uniform softmax
    val = L.loss(ctx, param, L.to_jl(feats), tokens)
    assert abs(val - math.log(V)) < 2e-3 and abs(val - deck) < 2e-3
    ctx.close()


def test_eight_shards_of_32_equal_one_batch_of_256():
    V, B, T = 10640, 256, 11
    ctx, param, feats, tokens = make(V, B, T)
    g_full, loss_full = L.lossgradient(ctx, param, L.to_jl(feats), tokens)
    g_full = [L.from_jl(g).astype(np.float64) for g in g_full]
    acc, loss_sum = [np.zeros_like(g) for g in g_full], 0.0
    for r in range(8):
        rows = slice(32 * r, 32 * (r + 1))
        g, l = L.lossgradient(ctx, param, L.to_jl(feats[rows]), tokens[:, rows], norm_B=B)
        loss_sum += l
        for a, x in zip(acc, g):
            a += L.from_jl(x)
    assert abs(loss_sum - loss_full) <= 1e-3 * abs(loss_full)  # bf16 GEMMs with different tile shapes per batch size
    for name, a, b in zip("W1 b1 W2 b2 Wproj Wcnn Wembed Wout bout".split(), acc, g_full):
        rel = np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)
        assert rel < 2e-2, (name, rel)
    ctx.close()


def test_adam_first_step_is_bounded_by_lr():
    ctx, param, feats, tokens = make(10640, 64, 11)
    before = [p.clone() for p in param]
    optim = L.initparams(param)
    grads, _ = L.lossgradient(ctx, param, L.to_jl(feats), tokens)
    L.update(ctx, param, grads, optim)
    ctx.sync()
    for p, q, g in zip(param, before, grads):
        d = (p - q).abs()
        assert float(d.max()) <= 1e-3 * (1 + 1e-4)
        big = g.abs() > 1e-6  # |g| >> eps: the bias-corrected first step is lr * g / (|g| + eps) ~ lr
        if bool(big.any()):
            assert float(d[big].min()) >= 0.9e-3
    ctx.close()


def test_vgg_batch_equals_its_halves():
    ctx = L.Context(8, 8, 8, 17, max_B=2, max_T=1, vgg_dtype=lrcn_amd.LRCN_BF16, max_images=16)
    L.vgg_load(ctx, *L.synthetic_vgg_weights(seed=1))
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    img = torch.randint(0, 256, (16, 224, 224, 3), generator=g, device="cuda", dtype=torch.uint8)
    full = L.from_jl(L.convnet_u8(ctx, img))
    a = L.from_jl(L.convnet_u8(ctx, img[:8].contiguous()))
    b = L.from_jl(L.convnet_u8(ctx, img[8:].contiguous()))
    np.testing.assert_allclose(np.concatenate([a, b]), full, rtol=0, atol=2e-2 * np.abs(full).max())
    ctx.close()
