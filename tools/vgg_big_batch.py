import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lrcn_amd
from lrcn_amd import lrcn as L
N = int(sys.argv[1])
w = L.synthetic_vgg_weights(seed=3, bias_std=0.1)
g = torch.Generator(device="cuda"); g.manual_seed(5)
imgs = torch.randint(0, 256, (N, 224, 224, 3), generator=g, device="cuda", dtype=torch.uint8)
ctx = L.Context(8, 8, 8, 17, max_B=2, max_T=1, vgg_dtype=lrcn_amd.LRCN_BF16, max_images=N)
L.vgg_load(ctx, *w)
try:
    big = L.from_jl(L.convnet_u8(ctx, imgs)).copy()
except Exception as e:
    print("N=%d failed: %s" % (N, e)); sys.exit(0)
print("routes", L.debug_route(ctx, 1))
ctx2 = L.Context(8, 8, 8, 17, max_B=2, max_T=1, vgg_dtype=lrcn_amd.LRCN_BF16, max_images=256)
L.vgg_load(ctx2, *w)
worst = 0.0
for s in range(0, N, 256):
    part = L.from_jl(L.convnet_u8(ctx2, imgs[s:s + 256])).copy()
    worst = max(worst, float(np.abs(part - big[s:s + 256]).max() / np.abs(part).max()))
print("N=%d finite=%s max rel diff vs 256-chunks %.3g" % (N, np.isfinite(big).all(), worst))
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): L.convnet_u8(ctx, imgs)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
print("N=%d %.3f ms/forward %.0f img/s" % (N, dt * 1e3, N / dt))
