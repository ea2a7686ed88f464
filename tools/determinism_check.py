#!/usr/bin/env python3
"""Development aid: is lossgradient bit-reproducible call to call (same inputs)?  Prints, per gradient tensor, the number of differing
elements between two calls, for the weight-gradient stream on and off.  usage: tools/determinism_check.py [B=256]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import lrcn_amd  # noqa: E402
from lrcn_amd import lrcn as L  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
V, T = 10640, 11
names = "W1 b1 W2 b2 Wproj Wcnn Wembed Wout bout".split()
for wg in ("1", "0"):
    os.environ["LRCN_WG_STREAM"] = wg
    ctx = L.Context(1000, 1000, 1000, V, max_B=B, max_T=T, lstm_dtype=lrcn_amd.LRCN_BF16)
    param = L.initweights(ctx, seed=42)
    rng = np.random.default_rng(42)
    feats = L.to_jl((rng.standard_normal((B, 4096)) * 0.01).astype(np.float32))
    tokens = rng.integers(3, V, size=(T, B)).astype(np.int32)
    runs = []
    for it in range(3):
        g, l = L.lossgradient(ctx, param, feats, tokens)
        runs.append(([L.from_jl(x).copy() for x in g], l))
    for a in (1, 2):
        diff = {n: int((x != y).sum()) for n, x, y in zip(names, runs[0][0], runs[a][0])}
        print("WG_STREAM=%s run0 vs run%d: loss %r %r; differing elements %s" % (wg, a, runs[0][1], runs[a][1], {k: v for k, v in diff.items() if v}))
    ctx.close()
