"""Development probe: host-side cost of lrcn_upload_crops (hipMemcpyAsync from pinned memory on the copy stream) and of a VGG forward issue."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import lrcn_amd  # noqa: E402
from lrcn_amd import lrcn as L  # noqa: E402

N = 256
ctx = L.Context(8, 8, 8, 17, max_B=2, max_T=1, vgg_dtype=lrcn_amd.LRCN_BF16, max_images=N)
L.vgg_load(ctx, *L.synthetic_vgg_weights(seed=1))
host = [torch.randint(0, 256, (N, 224, 224, 3), dtype=torch.uint8).pin_memory() for _ in range(3)]
dev = host[0].cuda()
feats = L.jl_empty(N, 4096)
for _ in range(3):
    L.convnet_u8(ctx, dev, feats=feats)
torch.cuda.synchronize()
# 1. upload alone, GPU idle
for k in range(4):
    t0 = time.perf_counter()
    s = L.upload_crops(ctx, host[k % 3])
    t1 = time.perf_counter()
    L.upload_wait(ctx)
    t2 = time.perf_counter()
    L.convnet_u8(ctx, s, feats=feats)
    torch.cuda.synchronize()
    print("idle GPU: upload call %.3f ms, copy done after %.3f ms (%.1f GB/s)" % (1e3 * (t1 - t0), 1e3 * (t2 - t0), N * 150528 / (t2 - t0) / 1e9))
# 2. upload while a forward runs
for k in range(4):
    L.convnet_u8(ctx, dev, feats=feats)
    t0 = time.perf_counter()
    s = L.upload_crops(ctx, host[k % 3])
    t1 = time.perf_counter()
    L.upload_wait(ctx)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    L.convnet_u8(ctx, s, feats=feats)
    torch.cuda.synchronize()
    print("busy GPU: upload call %.3f ms, copy done after %.3f ms, forward done after %.3f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t0), 1e3 * (t3 - t0)))
# 3. forward time with and without a concurrent upload
def fwd(with_upload):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(10):
        if with_upload:
            s = L.upload_crops(ctx, host[k % 3])
        L.convnet_u8(ctx, dev, feats=feats)
        if with_upload:
            L.convnet_u8(ctx, s, feats=feats)   # consume
        else:
            L.convnet_u8(ctx, dev, feats=feats)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20
print("forward ms: alone %.3f  with concurrent uploads %.3f" % (fwd(False), fwd(True)))
print("forward ms: alone %.3f  with concurrent uploads %.3f" % (fwd(False), fwd(True)))
# 4. torch's own pinned copy on a side stream for comparison
cs = torch.cuda.Stream()
buf = torch.empty_like(dev)
for k in range(3):
    L.convnet_u8(ctx, dev, feats=feats)
    t0 = time.perf_counter()
    with torch.cuda.stream(cs):
        buf.copy_(host[k % 3], non_blocking=True)
    t1 = time.perf_counter()
    cs.synchronize()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    print("torch copy_ on a side stream, busy GPU: call %.3f ms, done after %.3f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t0)))
