"""CPU, world_size 2 over gloo: train! / train1 / average_loss over the data-parallel trainer (lrcn_amd/train.py = lrcn.jl:223-246,
330-397, 407-486 with every bucketed batch split by rows over the ranks).  The two-rank job's checkpoint -- parameters AND Adam state --
must equal the single-process one, epoch losses included; with the sharded update too (moments gathered back for the checkpoint).
The device operations are the oracle's stand-ins (tests/dp_oracle_ops.py): this tests the loop, the sharding and the collectives."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dp_oracle_ops import HostAdam, OracleGroupOps
from oracle import oracle as orc

E, H, EPOCHS = 12, 16, 2


def dataset():
    """12 'scenes' x 2 captions of two lengths; feature = one-hot-ish block per scene (as tests/test_gpu_cli.py)."""
    from lrcn_amd import captions as cap
    nouns, verbs = ["dog", "cat", "man", "bird"], ["runs", "sleeps", "jumps"]
    anns, feats = [], {}
    for img in range(48):
        n, v = nouns[img % 4], verbs[(img // 4) % 3]
        f = np.zeros(4096, np.float32)
        f[(img % 4) * 100:(img % 4) * 100 + 50] = 1.0
        f[1000 + ((img // 4) % 3) * 100:1000 + ((img // 4) % 3) * 100 + 50] = 1.0
        feats[img] = f / f.sum()
        anns.append({"image_id": img, "caption": "A %s %s ." % (n, v)})
        anns.append({"image_id": img, "caption": "The %s %s now ." % (n, v)})
    import json
    caps = cap.tokenize_coco(json.dumps({"annotations": anns}))
    vocab = cap.build_vocab([caps], min_count=1) if "min_count" in cap.build_vocab.__code__.co_varnames else cap.build_vocab([caps])
    seq = cap.minibatch(caps, vocab, 8)
    blocks = list(cap.batches(seq[0], seq[1], seq[2], seq[3]))
    return vocab, blocks, feats, seq[3]


def _worker(rank, world, port, out, shard):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from lrcn_amd import dp, formats as fmt, train as trn
    vocab, blocks, feats, Bg = dataset()
    V = len(vocab)
    m = orc.init_weights(E, H, H, V, seed=5)
    param = [torch.as_tensor(np.array(m.p[n])) for n in orc.PARAM_NAMES]
    optim = HostAdam(param)
    tr = dp.DataParallelTrainer(None, param, optim, Bg, world, rank, pdrop=0.0, ops=OracleGroupOps((E, H, H, V)), shard_adam=shard)
    assert tr.shard == shard

    def feats_of(ids):
        return torch.as_tensor(np.stack([feats[i] for i in ids]))

    def save(epoch):
        if rank == 0:
            fmt.save_checkpoint(out % world, [p.numpy() for p in param], vocab,
                                adam={"m": [t.numpy() for t in optim.m], "v": [t.numpy() for t in optim.v], "step": optim.t})

    lines = []
    hist = trn.train(tr, [blocks, blocks[:3]], EPOCHS, seed=11, feats_of=feats_of, save=save, log=lines.append)
    if rank == 0:
        assert len(lines) == EPOCHS and all(ln.startswith("(:epoch, ") for ln in lines), lines
        np.save((out % world) + ".hist.npy", np.asarray(hist))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("shard", [False, True])
def test_two_rank_training_checkpoint_equals_single_process(tmp_path, shard):
    import socket
    from lrcn_amd import formats as fmt
    out = str(tmp_path / "ck_w%d.npz")
    for world in (1, 2):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        if world == 1:
            _worker(0, 1, port, out, False)
        else:
            mp.spawn(_worker, args=(world, port, out, shard), nprocs=world, join=True)
    (m1, v1, a1, _), (m2, v2, a2, _) = fmt.load_checkpoint(out % 1), fmt.load_checkpoint(out % 2)
    assert v1 == v2 and a1["step"] == a2["step"] > 0
    h1, h2 = np.load((out % 1) + ".hist.npy"), np.load((out % 2) + ".hist.npy")
    np.testing.assert_allclose(h1, h2, rtol=1e-6)
    assert h1[-1, 0] < h1[0, 0]                       # it trains
    for a, b in zip(m1, m2):
        np.testing.assert_allclose(a, b, rtol=0, atol=2e-6)
    for k in ("m", "v"):
        for a, b in zip(a1[k], a2[k]):
            np.testing.assert_allclose(a, b, rtol=1e-4, atol=1e-9)
    assert any(np.abs(t).max() > 0 for t in a2["m"])  # the sharded update's moments made it back into the checkpoint


def test_epoch_order_and_sharding_are_rank_independent():
    from lrcn_amd import train as trn
    a, b = trn.epoch_order(17, 5, 3), trn.epoch_order(17, 5, 3)
    assert (a == b).all() and sorted(a) == list(range(17)) and not (a == trn.epoch_order(17, 5, 4)).all()
    ids, toks = list(range(8)), np.arange(24).reshape(3, 8)
    parts = [trn.shard_block((ids, toks), 4, r) for r in range(4)]
    assert sum((p[0] for p in parts), []) == ids
    np.testing.assert_array_equal(np.concatenate([p[1] for p in parts], axis=1), toks)
