#!/usr/bin/env python3
"""Kernel-development aid: the fused conv1_1 + conv1_2 + pool launch (lrcn_conv1_fused) against the bf16-emulating oracle, with WHERE the
values differ (tile, position inside the tile, channel) and by how many bf16 steps.   python tools/fused_conv1_probe.py [S N cap gen]..."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def run(S, N, cap, gen):
    os.environ["LRCN_FUSE11_GEN"] = str(gen)
    import lrcn_amd
    from lrcn_amd import lrcn as L
    from oracle import oracle as orc
    rng = np.random.default_rng(S * 10 + N)
    img = rng.integers(0, 256, size=(N, S, S, 3), dtype=np.uint8)
    w11 = (rng.standard_normal((3, 3, 3, 64)) * np.sqrt(2.0 / 27)).astype(np.float32)
    w12 = (rng.standard_normal((3, 3, 64, 64)) * np.sqrt(2.0 / 576)).astype(np.float32)
    b11 = (rng.standard_normal(64) * 20.0).astype(np.float32)
    b12 = (rng.standard_normal(64) * 20.0).astype(np.float32)
    mean = np.array(L.VGG_MEAN, np.float32)
    x = orc.preprocess_u8(img, mean)
    with orc.emulate_bf16():
        emu = orc.pool2(orc.conv3x3(orc.conv3x3(x, w11, b11, relu=True), w12, b12, relu=True))
    ctx = L.Context(8, 8, 8, 17, max_B=2, max_T=1, vgg_dtype=lrcn_amd.LRCN_BF16, max_images=0)
    if cap:
        L.vgg_set_wg_cap(ctx, cap)
    got = L.from_jl(L.conv1_fused(ctx, torch.as_tensor(img).cuda(), mean, L.to_jl(w11), torch.as_tensor(b11).cuda(), L.to_jl(w12),
                                  torch.as_tensor(b12).cuda()))
    ctx.close()
    diff = np.abs(got - emu)
    step = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(emu), 1e-30))) - 7)
    steps = diff / step
    bad = np.argwhere(steps > 1.01)
    print("S=%d N=%d cap=%d gen=%s: identical %.4f, one step %.5f, more %d of %d (max %.1f steps, max diff %.3g of max %.3g)"
          % (S, N, cap, gen, (diff == 0).mean(), ((steps > 0) & (steps <= 1.01)).mean(), len(bad), diff.size, steps.max(), diff.max(), np.abs(emu).max()))
    for i, j, c, n in bad[:12]:
        print("   y(i=%d, j=%d, c=%d, n=%d): got %g emu %g   tile (%d,%d) inside (%d,%d)" % (i, j, c, n, got[i, j, c, n], emu[i, j, c, n], i // 8, j // 8, i % 8, j % 8))


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:]]
    cases = [a[k:k + 4] for k in range(0, len(a), 4)] or [[16, 1, 0, 2], [48, 2, 8, 2], [48, 2, 8, 1], [64, 5, 8, 2], [224, 1, 24, 2], [224, 1, 24, 1]]
    for S, N, cap, gen in cases:
        run(S, N, cap, gen)
