#!/usr/bin/env python3
"""Numerical-health check of the bf16 training path at BASELINE configs[3] dimensions (E = H = 1000, V = 10640, T = 11): the same training
run -- same initial weights, same batches, same device-generated dropout masks -- in fp32 (exact-fp32 MFMA) and in bf16, step by step.
Learnable synthetic data: each of 64 "scenes" has its own feature pattern and its own caption.  Prints the per-step loss of both runs and
their largest relative difference.   usage: tools/bf16_vs_f32_trajectory.py [steps=300] [B=32]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import lrcn_amd  # noqa: E402
from lrcn_amd import dp  # noqa: E402
from lrcn_amd import lrcn as L  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
E = H = 1000
V, T, NS = 10640, 11, 64
rng = np.random.default_rng(1)
scene_feat = np.zeros((NS, 4096), np.float32)
for s in range(NS):
    scene_feat[s, rng.choice(4096, 200, replace=False)] = 1.0 / 200
scene_cap = (rng.integers(3, V, size=(NS, T))).astype(np.int32)
batches = []
for k in range(steps):
    sc = rng.integers(0, NS, size=B)
    batches.append((scene_feat[sc] + rng.normal(0, 2e-4, (B, 4096)).astype(np.float32), np.ascontiguousarray(scene_cap[sc].T)))


def run(dtype):
    ctx = L.Context(E, H, H, V, max_B=B, max_T=T, lstm_dtype=dtype)
    param = L.initweights(ctx, seed=42)
    tr = dp.DataParallelTrainer(ctx, param, L.initparams(param), B, 1, 0, pdrop=0.4, seed=7)
    out = []
    for f, t in batches:
        tr.step(None, torch.as_tensor(t).cuda(), feats=L.to_jl(f))
        out.append(tr.loss_value())
    tr.close()
    ctx.close()
    return np.array(out)


a = run(lrcn_amd.LRCN_F32)
b = run(lrcn_amd.LRCN_BF16)
rel = np.abs(a - b) / np.abs(a)
print("steps %d, B = %d, E = H = %d, V = %d, T = %d, dropout 0.4 (same device-generated masks), Adam 1e-3, fused update on" % (steps, B, E, V, T))
for k in list(range(0, steps, max(1, steps // 15))) + [steps - 1]:
    print("step %4d  fp32 %.6f  bf16 %.6f  rel %.2e" % (k + 1, a[k], b[k], rel[k]))
print("largest relative difference over the run: %.2e at step %d; mean %.2e; final loss fp32 %.4f bf16 %.4f (ln V = %.4f)" % (
    rel.max(), int(rel.argmax()) + 1, rel.mean(), a[-1], b[-1], np.log(V)))
