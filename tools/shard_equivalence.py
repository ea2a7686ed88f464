#!/usr/bin/env python3
"""Validation aid (one GPU): one training step on a full batch of 256 rows against the same step computed as N row shards with the global
normaliser, gradients summed, one Adam step -- in ONE process, so that only the sharding arithmetic (and the kernel routes the smaller
shards take) differs, not any multi-process plumbing.  Prints the loss before, the relative gradient differences, and the loss after
the update for N = 1, 2, 4, 8.   LRCN_DETERMINISTIC=1 python tools/shard_equivalence.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import lrcn_amd  # noqa: E402
from lrcn_amd import lrcn as L  # noqa: E402

E = H = 1000
V, T, Bg = 10640, 11, 256
rng = np.random.default_rng(7)
feats_np = (rng.standard_normal((Bg, 4096)) * 0.5).astype(np.float32)
pz = 1.0 / np.arange(1, V - 3 + 1)
toks_np = (rng.choice(V - 3, size=(T, Bg), p=pz / pz.sum()) + 3).astype(np.int32)


def run(N):
    B = Bg // N
    ctx = L.Context(E, H, H, V, max_B=B, max_T=T, lstm_dtype=lrcn_amd.LRCN_BF16)
    param = L.initweights(ctx, seed=42)
    optim = L.initparams(param)
    total = L.zeros_like_model(param)
    loss0 = 0.0
    for r in range(N):
        rows = slice(r * B, (r + 1) * B)
        g, val = L.lossgradient(ctx, param, L.to_jl(feats_np[rows]), np.ascontiguousarray(toks_np[:, rows]), norm_B=Bg)
        loss0 += val
        for t, gg in zip(total, g):
            t.add_(gg)
    L.update(ctx, param, total, optim)
    loss1 = 0.0
    for r in range(N):
        rows = slice(r * B, (r + 1) * B)
        loss1 += L.loss(ctx, param, L.to_jl(feats_np[rows]), np.ascontiguousarray(toks_np[:, rows]), norm_B=Bg)
    torch.cuda.synchronize()
    out = [L.from_jl(t).astype(np.float64) for t in total]
    ctx.close()
    return loss0, loss1, out


def main():
    ref = run(1)
    print("N=1  loss before %.9f  after one update %.9f" % ref[:2])
    for N in (2, 4, 8):
        l0, l1, g = run(N)
        rel = [np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300) for a, b in zip(g, ref[2])]
        print("N=%d  loss before %.9f  after one update %.9f   gradient rel. diff vs N=1: %s" % (N, l0, l1, " ".join("%.1e" % x for x in rel)))


if __name__ == "__main__":
    main()
