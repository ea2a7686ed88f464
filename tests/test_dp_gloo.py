"""CPU, world_size 2 over gloo: the data-parallel step's sharding + all-reduce + replicated Adam (dp.py) equals the
single-process full-batch step.  The device ops are injected (the oracle stands in for the HIP calls HERE ONLY: this
tests the collective logic, which is what differs between N=1 and N>1)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle as orc


class OracleOps:
    def __init__(self, dims):
        self.dims = dims
        self._loss = 0.0

    def _model(self, param):
        E, H1, H2, V = self.dims
        return orc.Model(E, H1, H2, V, {n: p.numpy() for n, p in zip(orc.PARAM_NAMES, param)})

    def vgg(self, img):
        raise AssertionError("features are given in this test")

    def lossgradient(self, param, feats, tokens, norm_B, pdrop, seed, grads):
        val, g = orc.loss(self._model(param), feats.numpy(), tokens, norm_B=norm_B, want_grad=True)
        self._loss = val
        for n, t in zip(orc.PARAM_NAMES, grads):
            t.copy_(torch.as_tensor(g.p[n]))

    def update(self, param, grads, optim):
        optim.t += 1
        for p, g, m, v in zip(param, grads, optim.m, optim.v):
            w, mm, vv = (np.asfortranarray(a.numpy()) for a in (p, m, v))
            orc.adam(w, np.asfortranarray(g.numpy()), mm, vv, optim.t)
            p.copy_(torch.as_tensor(w)); m.copy_(torch.as_tensor(mm)); v.copy_(torch.as_tensor(vv))

    def last_loss(self):
        return self._loss


class HostAdam:
    def __init__(self, param):
        self.t = 0
        self.m = [torch.zeros_like(p) for p in param]
        self.v = [torch.zeros_like(p) for p in param]


def _worker(rank, world, port, golden, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lrcn_amd import dp
    z = np.load(golden)
    dims = tuple(int(z[k]) for k in ("E", "H1", "H2", "V"))
    param = [torch.as_tensor(np.array(z["p_" + n])) for n in orc.PARAM_NAMES]
    Bg = z["feats"].shape[0]
    rows = dp.shard_rows(Bg, world, rank)
    tr = dp.DataParallelTrainer(None, param, HostAdam(param), Bg, world, rank, pdrop=0.0, ops=OracleOps(dims))
    feats = torch.as_tensor(z["feats"][rows])
    toks = z["tokens"][:, rows]
    losses = []
    for _ in range(2):
        tr.step(None, toks, feats=feats)
        losses.append(tr.loss_value())
    if rank == 0:
        np.savez(out, losses=np.array(losses), **{n: p.numpy() for n, p in zip(orc.PARAM_NAMES, param)})
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_step_equals_full_batch(golden_dir, tmp_path):
    golden = os.path.join(golden_dir, "lstm_mid.npz")
    out = str(tmp_path / "dp.npz")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, golden, out), nprocs=2, join=True)
    got = np.load(out)
    # single-process reference: two full-batch steps with the oracle
    z = np.load(golden)
    dims = tuple(int(z[k]) for k in ("E", "H1", "H2", "V"))
    model = orc.Model(*dims, {n: z["p_" + n] for n in orc.PARAM_NAMES})
    mom = {n: np.zeros_like(model.p[n]) for n in orc.PARAM_NAMES}
    var = {n: np.zeros_like(model.p[n]) for n in orc.PARAM_NAMES}
    ref_losses = []
    for t in (1, 2):
        val, g = orc.loss(model, z["feats"], z["tokens"], want_grad=True)
        ref_losses.append(val)
        for n in orc.PARAM_NAMES:
            orc.adam(model.p[n], g.p[n], mom[n], var[n], t)
    np.testing.assert_allclose(got["losses"], ref_losses, rtol=1e-6)
    for n in orc.PARAM_NAMES:
        np.testing.assert_allclose(got[n], model.p[n], rtol=0, atol=3e-6, err_msg=n)


def test_shard_rows_and_flat_views():
    from lrcn_amd import dp
    assert dp.shard_rows(256, 8, 3) == slice(96, 128)
    with pytest.raises(Exception):
        dp.shard_rows(10, 4, 0)
    flat, views = dp.flat_model_like([(3, 4), (1, 5)], device="cpu")
    assert flat.numel() == 17 and views[0].shape == (3, 4) and views[0].stride() == (1, 3)
    views[1][0, 2] = 7.0
    assert flat[12 + 2] == 7.0
