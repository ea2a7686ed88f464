#!/usr/bin/env python3
"""Validation aid for a ONE-GPU box: the data-parallel trainer (dp.py, torch backend) with N ranks that SHARE device 0 over gloo (RCCL
refuses two ranks per device), on GIVEN features, pdrop 0, LRCN_DETERMINISTIC=1 -- so that the loss trajectory must equal the one-rank
trajectory to summation-order accuracy.  Only the transport differs from a real N-GPU run.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/fake_multi_check.py [--shard-adam]
    python tools/fake_multi_check.py            # the one-rank reference
Never start it from a process that has touched the GPU (the ranks must be children of a GPU-free parent)."""
import os
import sys

os.environ.setdefault("LRCN_DETERMINISTIC", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import lrcn_amd  # noqa: E402
from lrcn_amd import dp  # noqa: E402
from lrcn_amd import lrcn as L  # noqa: E402

world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
if world > 4:
    raise SystemExit("at most 4 ranks may share one GPU (the pool's limit is 6 processes per device)")
torch.cuda.set_device(0)
if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)
E = H = 1000
V, T, Bg = 10640, 11, int(os.environ.get("FM_BG", "256"))
rows = dp.shard_rows(Bg, world, rank)
B = rows.stop - rows.start
ctx = L.Context(E, H, H, V, max_B=B, max_T=T, lstm_dtype=lrcn_amd.LRCN_BF16)
param = L.initweights(ctx, seed=42)
tr = dp.DataParallelTrainer(ctx, param, L.initparams(param), Bg, world, rank, pdrop=0.0, seed=7, backend="torch",
                            shard_adam="--shard-adam" in sys.argv)
rng = np.random.default_rng(7)
pz = 1.0 / np.arange(1, V - 3 + 1)
losses = []
for k in range(int(os.environ.get('FM_STEPS', '4'))):
    feats = (rng.standard_normal((Bg, 4096)) * 0.5).astype(np.float32)
    toks = (rng.choice(V - 3, size=(T, Bg), p=pz / pz.sum()) + 3).astype(np.int32)
    tr.step(None, torch.as_tensor(np.ascontiguousarray(toks[:, rows])).cuda(), feats=L.to_jl(feats[rows]))
    losses.append(tr.loss_value())
torch.cuda.synchronize()
per = [float(np.abs(L.from_jl(p).astype(np.float64)).sum()) for p in param]
chk = float(sum(float(L.from_jl(p).astype(np.float64).sum()) for p in param))
if rank == 0 and os.environ.get("FM_VERBOSE"):
    print("per-tensor |.| sums " + " ".join("%.6f" % x for x in per), flush=True)
if rank == 0:
    print("world %d %s losses %s  param checksum %.9f" % (world, "sharded" if tr.shard else "replicated", " ".join("%.7f" % x for x in losses), chk), flush=True)
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
