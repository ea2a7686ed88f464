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


class OracleGroupOps(OracleOps):
    """The same, plus the CPU stand-ins of what the per-group [all-reduce -> Adam] pipeline needs (dp.py
    _reduce_and_update_groups / the bucketed _allreduce_async): "streams" are labels, the gradient-ready "events" are already
    complete (lossgradient is synchronous here), and every call is logged so the test can check the ORDER the trainer drives."""

    def __init__(self, dims):
        super().__init__(dims)
        self.log = []

    def make_streams(self, n):
        self.log.append(("make_streams", n))
        return ["bucket%d" % k for k in range(n)]

    def stream_ctx(self, stream):
        import contextlib
        return contextlib.nullcontext()

    def grad_group_wait(self, group, stream):
        self.log.append(("wait", group, stream))

    def update_group(self, param, grads, optim, group, stream):
        from lrcn_amd import dp
        self.log.append(("adam", group, stream, optim.t))
        for k in dp.GRAD_GROUPS[group]:
            w, mm, vv = (np.asfortranarray(a.numpy()) for a in (param[k], optim.m[k], optim.v[k]))
            orc.adam(w, np.asfortranarray(grads[k].numpy()), mm, vv, optim.t)
            param[k].copy_(torch.as_tensor(w)); optim.m[k].copy_(torch.as_tensor(mm)); optim.v[k].copy_(torch.as_tensor(vv))

    def join(self, streams):
        self.log.append(("join", len(streams)))


class HostAdam:
    def __init__(self, param):
        self.t = 0
        self.m = [torch.zeros_like(p) for p in param]
        self.v = [torch.zeros_like(p) for p in param]


def _worker(rank, world, port, golden, out, mode="plain"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if mode == "bucket_one_adam":
        os.environ["LRCN_DP_GROUP_ADAM"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lrcn_amd import dp
    z = np.load(golden)
    dims = tuple(int(z[k]) for k in ("E", "H1", "H2", "V"))
    param = [torch.as_tensor(np.array(z["p_" + n])) for n in orc.PARAM_NAMES]
    Bg = z["feats"].shape[0]
    rows = dp.shard_rows(Bg, world, rank)
    if mode == "abi_fallback":
        # the C-ABI communicator cannot be set up (here: on rank 1 only): every rank must agree to fall back to torch.distributed
        from lrcn_amd import lrcn as L

        class FailingAbiOps(OracleOps):
            destroyed = False

            def train_step_dp(self, *a, **k):
                raise AssertionError("the fallback must not take the C-ABI step")

            def comm_init(self, world_, rank_, uid):
                assert len(uid) == 128
                if rank_ == 1:
                    raise L.LrcnError("simulated: librccl not loadable")

            def comm_destroy(self):
                self.destroyed = True

        import unittest.mock as mock
        ops = FailingAbiOps(dims)
        with mock.patch.object(L, "comm_unique_id", lambda: bytes(range(128))):
            tr = dp.DataParallelTrainer(None, param, HostAdam(param), Bg, world, rank, pdrop=0.0, ops=ops, backend="abi")
        assert tr.backend == "torch" and ops.destroyed
    else:
        ops = OracleOps(dims) if mode == "plain" else OracleGroupOps(dims)
        tr = dp.DataParallelTrainer(None, param, HostAdam(param), Bg, world, rank, pdrop=0.0, ops=ops)
    assert tr.backend == "torch"
    feats = torch.as_tensor(z["feats"][rows])
    toks = z["tokens"][:, rows]
    losses = []
    for _ in range(2):
        tr.step(None, toks, feats=feats)
        losses.append(tr.loss_value())
    if mode == "group_pipeline":
        # the path N > 1 takes on the GPU: per step, for each gradient group in the order lossgradient finalises them,
        # [wait for the group's event on its stream] -> [all-reduce] -> [Adam of the group with this step's t], then one join
        per_step = [e for e in ops.log if e[0] != "make_streams"]
        assert ops.log[0] == ("make_streams", 5) and len(per_step) == 2 * 11
        for step in range(2):
            ev = per_step[step * 11:(step + 1) * 11]
            for k in range(5):
                assert ev[2 * k] == ("wait", k, "bucket%d" % k) and ev[2 * k + 1] == ("adam", k, "bucket%d" % k, step + 1), ev
            assert ev[10] == ("join", 5)
    elif mode == "bucket_one_adam":
        waits = [e for e in ops.log if e[0] == "wait"]
        assert len(waits) == 2 * 5 and not [e for e in ops.log if e[0] == "adam"]  # bucketed all-reduces, then ONE update
    if rank == 0:
        np.savez(out, losses=np.array(losses), **{n: p.numpy() for n, p in zip(orc.PARAM_NAMES, param)})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["plain", "group_pipeline", "bucket_one_adam", "abi_fallback"])
def test_two_rank_step_equals_full_batch(golden_dir, tmp_path, mode):
    golden = os.path.join(golden_dir, "lstm_mid.npz")
    out = str(tmp_path / "dp.npz")
    port = 29500 + (os.getpid() % 2000) + {"plain": 0, "group_pipeline": 1, "bucket_one_adam": 2, "abi_fallback": 3}[mode]
    mp.spawn(_worker, args=(2, port, golden, out, mode), nprocs=2, join=True)
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
