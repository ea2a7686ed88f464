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

import torch  # noqa: E402
w = L.synthetic_vgg_weights(seed=3, bias_std=0.1)
g = torch.Generator(device="cuda")
g.manual_seed(5)
imgs = torch.randint(0, 256, (B, 224, 224, 3), generator=g, device="cuda", dtype=torch.uint8)
for vdt, nm in ((lrcn_amd.LRCN_BF16, "bf16"),):
    ctx = L.Context(8, 8, 8, 17, max_B=2, max_T=1, vgg_dtype=vdt, max_images=B)
    L.vgg_load(ctx, *w)
    a = L.from_jl(L.convnet_u8(ctx, imgs)).copy()
    b = L.from_jl(L.convnet_u8(ctx, imgs)).copy()
    print("VGG forward %s, %d images: differing feature elements between two calls: %d" % (nm, B, int((a != b).sum())))
    ctx.close()
