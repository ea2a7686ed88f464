"""train! / train1 / average_loss (lrcn.jl:223-246, 330-397, 407-486) over dp.DataParallelTrainer.

The reference's epoch loop, unchanged in what it computes: batches of equal-length captions (captions.minibatch) visited in a
shuffled order (lrcn.jl:351), `lossgradient` + `update!` per batch (:378-394), then average_loss over the training and the
validation split (:233-234) and the `(:epoch, n, :loss, train, val)` line (:236).  What is new is WHERE it runs: every batch is split
by rows over the ranks of a one-node job (same T on every rank, the GLOBAL batch size as the loss normaliser, gradients summed
over ranks, identical Adam everywhere -- dp.py), and a batch may arrive as image ids whose crops go through the VGG forward on
the device (end-to-end training; the reference only ever trained on precomputed features, lrcn.jl:369-376).

Device-agnostic: the trainer's `ops` do the arithmetic (HipOps = liblrcn_hip; the CPU tests inject stand-ins), `feats_of(ids)`
/ `crops_of(ids)` put a batch's inputs where the ops want them.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import dp


def epoch_order(n_blocks, seed, epoch):
    """shuffle(1:batch_size:length(lengths)) (lrcn.jl:351) -- the same permutation on every rank (Julia's stream is not reproduced)."""
    return np.random.default_rng([int(seed) & 0x7FFFFFFF, int(epoch)]).permutation(n_blocks)


def shard_block(block, world, rank):
    """(image ids, tokens [T][B]) of a full batch -> this rank's contiguous rows of it."""
    ids, toks = block
    rows = dp.shard_rows(len(ids), world, rank)
    return list(ids[rows.start:rows.stop]), np.ascontiguousarray(np.asarray(toks)[:, rows.start:rows.stop])


def batches_per_forward(rows, chunk_images=256):
    """How many consecutive batches of `rows` captions one VGG forward should cover (dp.DataParallelTrainer.step): from 128 rows per GPU
    one (no gain measured), below that as many as fit into `chunk_images` crops -- 8 at 32 rows (one rank of 8 at batch 256), 4 at 64."""
    return max(1, int(chunk_images) // int(rows)) if rows <= 64 else 1


def train1(trainer, blocks, order, feats_of=None, crops_of=None, lookahead=1):
    """One epoch (lrcn.jl:350-396).  feats_of(ids) -> this rank's feature rows on the device, or crops_of(ids) -> uint8 crops
    [len(ids)][224][224][3] (device, or pinned host): then the VGG forward of the NEXT `lookahead` batches (one forward for all of them)
    runs beside the LSTM steps of the current ones.  Returns the number of captions this rank trained on."""
    W, r = trainer.world, trainer.rank
    n = 0
    shards = [shard_block(blocks[k], W, r) for k in order]
    nxt_pos, nxt = 1, None          # position (in `order`) of the first batch whose crops have not been handed to the trainer yet
    for pos, (ids, toks) in enumerate(shards):
        if crops_of is None:
            trainer.step(None, toks, feats=feats_of(ids))
        else:
            nxt_pos = max(nxt_pos, pos + 1)
            if nxt is None and nxt_pos < len(shards):
                nxt = crops_of(sum((s[0] for s in shards[nxt_pos:nxt_pos + lookahead]), []))
            # the first batch's own crops are only read when no earlier step produced its features (the very first step of the epoch)
            cur = crops_of(ids) if (pos == 0 or not trainer._feat_q) else None
            if trainer.step(cur, toks, next_img_u8=nxt):
                nxt_pos += lookahead
                nxt = None
        n += len(ids)
    return n


def average_loss(trainer, blocks, feats_of):
    """average_loss (lrcn.jl:407-486): forward-only NLL, pdrop 0, -total / count with count = sum of B (T+1) over the batches.
    Whole batches are dealt round-robin to the ranks (the parameters are replicated); (total, count) are summed over ranks."""
    W, r = trainer.world, trainer.rank
    total, count = 0.0, 0
    for ids, toks in blocks[r::W]:
        toks = np.asarray(toks)
        T, B = toks.shape
        val = trainer.ops.loss(trainer.param, feats_of(list(ids)), toks)   # -sum logp / (B (T+1))
        total += val * B * (T + 1)
        count += B * (T + 1)
    if W > 1:
        t = torch.tensor([total, float(count)], dtype=torch.float64, device=trainer.param[0].device if dist.get_backend(trainer.group) == "nccl" else "cpu")
        dist.all_reduce(t, group=trainer.group)
        total, count = float(t[0].item()), int(round(float(t[1].item())))
    return total / max(count, 1)


def train(trainer, splits, epochs, seed, feats_of=None, crops_of=None, eval_feats_of=None, save=None, log=print, sync=None, lookahead=1):
    """train! (lrcn.jl:223-246).  splits: [(blocks of the training split), (blocks of the dev split)?], blocks = [(ids, tokens)].
    feats_of / crops_of: the training inputs (one of them); eval_feats_of[i](ids): features of split i for average_loss (default:
    feats_of).  save(epoch): called on every rank after each epoch (rank 0 writes; a sharded update gathers its moments first).
    Returns the list of per-epoch loss tuples."""
    import time
    history = []
    ev = eval_feats_of or [feats_of] * len(splits)
    for epoch in range(1, epochs + 1):
        t0 = time.time()
        n = train1(trainer, splits[0], epoch_order(len(splits[0]), seed, epoch), feats_of=feats_of, crops_of=crops_of, lookahead=lookahead)
        if sync is not None:
            sync()
        dt_s = time.time() - t0
        if save is not None:
            trainer.gather_optim_state()
            save(epoch)
        losses = tuple(average_loss(trainer, blk, f) for blk, f in zip(splits, ev))
        history.append(losses)
        if trainer.rank == 0:
            log("(:epoch, %d, :loss, %s)  [%.0f captions/s on %d rank%s]" % (epoch, ", ".join("%.4f" % v for v in losses),
                                                                           n * trainer.world / max(dt_s, 1e-9), trainer.world, "" if trainer.world == 1 else "s"))
    return history
