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


from dp_oracle_ops import HostAdam, OracleGroupOps, OracleOps  # noqa: E402,F401


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
    if mode in ("abi_probe_fail", "abi_init_fail"):
        # the C-ABI communicator cannot be set up: every rank must agree to fall back to torch.distributed, and -- lrcn_comm_init being
        # a collective (ncclCommInitRank) -- no rank may ENTER it unless every rank's local probe passed (a lone rank would block forever)
        from lrcn_amd import lrcn as L

        class FailingAbiOps(OracleOps):
            destroyed = False
            entered_init = False

            def train_step_dp(self, *a, **k):
                raise AssertionError("the fallback must not take the C-ABI step")

            def comm_probe(self):
                if mode == "abi_probe_fail" and rank == 1:
                    return False, "simulated: librccl not loadable"
                return True, ""

            def comm_init(self, world_, rank_, uid):
                assert mode == "abi_init_fail", "entered the collective although a rank's probe failed: the other ranks would hang"
                assert len(uid) == 128 and uid == bytes(range(128))
                self.entered_init = True
                raise L.LrcnError("simulated: ncclCommInitRank failed on every rank")   # a symmetric failure inside the collective

            def comm_destroy(self):
                self.destroyed = True

        import unittest.mock as mock
        ops = FailingAbiOps(dims)
        with mock.patch.object(L, "comm_unique_id", lambda: bytes(range(128))):
            tr = dp.DataParallelTrainer(None, param, HostAdam(param), Bg, world, rank, pdrop=0.0, ops=ops, backend="abi")
        assert tr.backend == "torch" and ops.destroyed and tr.backend_note
        assert ops.entered_init == (mode == "abi_init_fail")
    elif mode == "shard_adam":
        # reduce-scatter -> Adam on this rank's slice of the flat parameter buffer -> all-gather (gloo has no reduce-scatter: dp.py falls
        # back to all-reduce + slice here, which leaves the group padding, the slice arithmetic and the gather to be tested)
        ops = OracleGroupOps(dims)
        before = [p.clone() for p in param]
        tr = dp.DataParallelTrainer(None, param, HostAdam(param), Bg, world, rank, pdrop=0.0, ops=ops, shard_adam=True)
        assert tr.shard and all(torch.equal(a, b) for a, b in zip(before, param))          # re-homed into the flat buffer, values kept
        assert all(p.untyped_storage().data_ptr() == tr.flat_param.untyped_storage().data_ptr() for p in param)
        assert all((b - a) % (4 * world) == 0 for a, b in tr._ranges)
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
    elif mode == "shard_adam":
        flats = [e for e in ops.log if e[0] == "adam_flat"]
        assert len(flats) == 2 * 5 and [e[2] for e in flats] == [1] * 5 + [2] * 5            # five slices per step, with that step's t
        assert sum(e[3] for e in flats[:5]) * world == tr.flat_param.numel()                  # the slices tile the padded buffer
        assert not [e for e in ops.log if e[0] == "adam"]
    elif mode == "bucket_one_adam":
        waits = [e for e in ops.log if e[0] == "wait"]
        assert len(waits) == 2 * 5 and not [e for e in ops.log if e[0] == "adam"]  # bucketed all-reduces, then ONE update
    if rank == 0:
        np.savez(out, losses=np.array(losses), **{n: p.numpy() for n, p in zip(orc.PARAM_NAMES, param)})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["plain", "group_pipeline", "bucket_one_adam", "abi_probe_fail", "abi_init_fail", "shard_adam"])
def test_two_rank_step_equals_full_batch(golden_dir, tmp_path, mode):
    golden = os.path.join(golden_dir, "lstm_mid.npz")
    out = str(tmp_path / "dp.npz")
    port = 29500 + (os.getpid() % 2000) + {"plain": 0, "group_pipeline": 1, "bucket_one_adam": 2, "abi_probe_fail": 3, "abi_init_fail": 4, "shard_adam": 5}[mode]
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
    # group-padded layout of the sharded update: every gradient group (adjacent tensors) starts where the previous one's padding ends, its
    # padded length is a multiple of the alignment, the views tile the unpadded part without overlap, the padding stays zero
    shapes = [(5, 8), (1, 8), (4, 8), (1, 8), (3, 2), (7, 2), (9, 3), (3, 9), (1, 9)]
    for align in (8, 16, 12):
        flat, views, ranges = dp.flat_model_like(shapes, device="cpu", group_align=align)
        assert len(ranges) == len(dp.GRAD_GROUPS) and all((b - a) % align == 0 and b > a for a, b in ranges)
        assert sorted(ranges) == sorted(set(ranges)) and sum(b - a for a, b in ranges) == flat.numel()
        for k, v in enumerate(views):
            v.fill_(k + 1)
        for grp, (a, b) in zip(dp.GRAD_GROUPS, ranges):
            n = sum(int(np.prod(shapes[k])) for k in grp)
            seg = flat[a:b]
            assert set(seg[:n].tolist()) == {float(k + 1) for k in grp} and float(seg[n:].abs().sum()) == 0.0
            assert views[min(grp)].storage_offset() == a
