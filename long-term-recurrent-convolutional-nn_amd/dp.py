"""Data-parallel training step: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The reference is single-device; what shards is the batch: each rank runs VGG forward + LSTM
forward/backward on its B_global/N rows, normalising by the GLOBAL batch (lrcn.jl:564-568 uses the global
`batchsize`), then ONE all-reduce(SUM) of the 9 gradient tensors (held in one flat buffer: one collective of
4 B/param) and an identical Adam step everywhere.  VGG is frozen (its output does not depend on the LSTM
parameters), so the all-reduce of step k is overlapped with the VGG forward of step k+1 while keeping exactly
synchronous-SGD semantics: RCCL runs on its own stream, the compute stream waits for it only before Adam.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import lrcn as L


def shard_rows(B_global, world, rank):
    """Contiguous row shard of a global batch (equal-length captions, so every rank sees the same T)."""
    if B_global % world:
        raise L.LrcnError("global batch %d is not divisible by world size %d" % (B_global, world))
    b = B_global // world
    return slice(rank * b, (rank + 1) * b)


def flat_model_like(shapes, device="cuda"):
    """One flat float32 buffer + 9 column-major views into it (so the gradient all-reduce is one collective)."""
    sizes = [int(np.prod(s)) for s in shapes]
    flat = torch.zeros(sum(sizes), device=device, dtype=torch.float32)
    views, off = [], 0
    for s, n in zip(shapes, sizes):
        v = flat[off:off + n].view(*reversed(s)).permute(*reversed(range(len(s))))
        views.append(v)
        off += n
    return flat, views


class HipOps:
    """The device operations a trainer step is made of -- all of them liblrcn_hip calls."""

    def __init__(self, ctx):
        self.ctx = ctx

    def vgg(self, img_u8):
        return L.convnet_u8(self.ctx, img_u8)

    def lossgradient(self, param, feats, tokens, norm_B, pdrop, seed, grads):
        L.lossgradient(self.ctx, param, feats, tokens, norm_B=norm_B, pdrop=pdrop, seed=seed, grads=grads,
                       want_loss=False)

    def update(self, param, grads, optim):
        L.update(self.ctx, param, grads, optim)

    def last_loss(self):
        return L.last_loss(self.ctx)


class DataParallelTrainer:
    """train1's batch loop body (lrcn.jl:369-394) sharded over ranks. world_size 1 = no collective.
    `ops` defaults to the HIP operations; tests of the collective logic on CPU (gloo) inject their own."""

    def __init__(self, ctx, param, optim, B_global, world=1, rank=0, pdrop=0.4, seed=0, group=None, ops=None):
        self.ops = ops if ops is not None else HipOps(ctx)
        self.ctx, self.param, self.optim = ctx, param, optim
        self.B_global, self.world, self.rank = B_global, world, rank
        self.pdrop, self.seed = pdrop, seed
        self.group = group
        self.flat_grads, self.grads = flat_model_like([tuple(t.shape) for t in param], device=param[0].device)
        self.step_no = 0
        self._feats_next = None

    def _allreduce_async(self):
        if self.world == 1:
            return None
        return dist.all_reduce(self.flat_grads, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def vgg(self, img_u8):
        return self.ops.vgg(img_u8)

    def step(self, img_u8, tokens, next_img_u8=None, feats=None):
        """One synchronous-SGD step on this rank's shard.  img_u8: this rank's uint8 crops (or feats given);
        next_img_u8: the NEXT step's crops, whose VGG forward is issued under this step's all-reduce."""
        if feats is None:
            feats = self._feats_next if self._feats_next is not None else self.vgg(img_u8)
        self._feats_next = None
        self.step_no += 1
        # rank-dependent dropout stream: masks differ per shard like rows of one big batch would
        self.ops.lossgradient(self.param, feats, tokens, self.B_global, self.pdrop,
                              (self.seed + self.step_no) * 65536 + self.rank, self.grads)
        work = self._allreduce_async()
        if next_img_u8 is not None:
            self._feats_next = self.vgg(next_img_u8)  # overlaps the all-reduce (frozen VGG)
        if work is not None:
            work.wait()  # compute stream waits for RCCL
        self.ops.update(self.param, self.grads, self.optim)

    def loss_value(self):
        """Global loss of the last step: sum over ranks of the locally normalised partial losses."""
        v = torch.tensor([self.ops.last_loss()], device=self.param[0].device, dtype=torch.float64)
        if self.world > 1:
            dist.all_reduce(v, group=self.group)
        return float(v.item())
