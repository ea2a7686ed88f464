"""GPU: the data-parallel trainer's single-GPU pipeline (dp.py): VGG forward of step k+1 on a side HIP stream, concurrent
with the LSTM forward/backward + Adam of step k, must give the same training trajectory as the strictly in-order run."""
import numpy as np
import pytest
import torch

import lrcn_amd
from lrcn_amd import dp
from lrcn_amd import lrcn as L

pytestmark = pytest.mark.gpu


def run_steps(monkeypatch, overlap, nsteps=4):
    monkeypatch.setenv("LRCN_OVERLAP_VGG", "1" if overlap else "0")
    E = H = 64
    V, B, T = 300, 4, 5
    ctx = L.Context(E, H, H, V, max_B=B, max_T=T, lstm_dtype=lrcn_amd.LRCN_BF16, vgg_dtype=lrcn_amd.LRCN_BF16, max_images=B)
    L.vgg_load(ctx, *L.synthetic_vgg_weights(seed=1))
    param = L.initweights(ctx, seed=42)
    optim = L.initparams(param)
    tr = dp.DataParallelTrainer(ctx, param, optim, B, 1, 0, pdrop=0.4, seed=7)
    assert (tr._side is not None) == overlap
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    imgs = [torch.randint(0, 256, (B, 224, 224, 3), generator=g, device="cuda", dtype=torch.uint8) for _ in range(nsteps)]
    rng = np.random.default_rng(3)
    toks = [torch.as_tensor(rng.integers(3, V, size=(T, B)).astype(np.int32)).cuda() for _ in range(nsteps)]
    losses = []
    for k in range(nsteps):
        tr.step(imgs[k], toks[k], next_img_u8=imgs[k + 1] if k + 1 < nsteps else None)
        losses.append(tr.loss_value())
    torch.cuda.synchronize()
    out = [L.from_jl(p).copy() for p in param]
    ctx.close()
    return losses, out


def test_side_stream_vgg_overlap_keeps_the_trajectory(monkeypatch):
    la, pa = run_steps(monkeypatch, overlap=True)
    lb, pb = run_steps(monkeypatch, overlap=False)
    assert all(np.isfinite(la)) and la[-1] != la[0]
    np.testing.assert_allclose(la, lb, rtol=1e-5)
    for a, b in zip(pa, pb):
        np.testing.assert_allclose(a, b, rtol=0, atol=2e-4)
