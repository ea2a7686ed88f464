"""VGG-16 -> fc7 forward throughput (images/s) on one GPU: bf16 against the e4m3 convolution stack (BASELINE config 5).
usage: python tools/vgg_bench.py [N=256] [iters=10]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import lrcn_amd
from lrcn_amd import lrcn as L

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
w = L.synthetic_vgg_weights(seed=1)
img = torch.as_tensor(np.random.default_rng(1234).integers(0, 256, size=(N, 224, 224, 3), dtype=np.uint8)).cuda()
res = {}
for name, dt in (("bf16", lrcn_amd.LRCN_BF16), ("fp8", lrcn_amd.LRCN_FP8)):
    ctx = L.Context(8, 8, 8, 17, max_B=2, max_T=1, vgg_dtype=dt, max_images=N)
    L.vgg_load(ctx, *w)
    if os.environ.get("LRCN_VGG_WG_CAP"):  # the capped persistent grids of the data-parallel step, alone on the chip
        L.vgg_set_wg_cap(ctx, int(os.environ["LRCN_VGG_WG_CAP"]))
    if dt == lrcn_amd.LRCN_FP8:
        L.vgg_calibrate(ctx, img[: min(N, 32)])
    feats = L.jl_empty(N, L.CNNOUT)
    for _ in range(3):
        L.convnet_u8(ctx, img, feats=feats)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        L.convnet_u8(ctx, img, feats=feats)
    torch.cuda.synchronize()
    dt_s = (time.perf_counter() - t0) / iters
    res[name] = (dt_s, L.from_jl(feats).copy())
    print("%s: N=%d %.3f ms/forward  %.0f images/s  (%.0f TFLOP/s on 30.93 GFLOP/image)" % (name, N, dt_s * 1e3, N / dt_s, N * 30.93e9 / dt_s / 1e12))
    ctx.close()
a, b = res["fp8"][1], res["bf16"][1]
cos = (a * b).sum(axis=1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
print("fp8 vs bf16 features: min cosine %.4f, rel L2 %.4f; speedup %.2fx" % (cos.min(), np.linalg.norm(a - b) / np.linalg.norm(b), res["bf16"][0] / res["fp8"][0]))
