#!/usr/bin/env python3
"""Kernel-development aid: where one output tile of the phase-interleaved convolution kernel spends its time.
  LRCN_STAMPS=1 python tools/tile_stamps.py [N=256] [layer names...]
Reads the per-tile shader-clock stamps gemm8p_tile records (include/lrcn.h lrcn_debug_stamps) and prints, per layer, the median
cycles of: prologue DMA issue, prologue wait (includes the previous tile's store drain: one in-order vmcnt), main loop,
accumulator staging, store issue, and the whole tile (start to start of the same workgroup's next tile).  Needs an MI355X."""
import ctypes as C
import os
import sys

os.environ.setdefault("LRCN_STAMPS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import lrcn_amd  # noqa: E402
from lrcn_amd import lrcn as L  # noqa: E402

LAYERS = {"conv2_2": (112, 128, 128, 1), "conv3_1": (56, 128, 256, 0), "conv3_2": (56, 256, 256, 0), "conv3_3": (56, 256, 256, 1),
          "conv4_1": (28, 256, 512, 0), "conv4_2": (28, 512, 512, 0), "conv4_3": (28, 512, 512, 1), "conv5_1": (14, 512, 512, 0),
          "conv5_3": (14, 512, 512, 1)}


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    names = sys.argv[2:] or list(LAYERS)
    ctx = L.Context(8, 8, 8, 17, max_B=2, max_T=1, lstm_dtype=lrcn_amd.LRCN_BF16, vgg_dtype=lrcn_amd.LRCN_BF16, max_images=1)
    lib = lrcn_amd._lib.lib()
    if os.environ.get("TILE_STAMPS_CAP"):  # persistent walk of a capped grid, as in the two-stream training step (224 there)
        L.vgg_set_wg_cap(ctx, int(os.environ["TILE_STAMPS_CAP"]))
    print("%-8s %-6s %6s | %8s %8s %8s %8s %8s | %8s %8s  %5s" % ("layer", "route", "tiles", "issue", "wait", "mainloop", "stage", "stores",
                                                                  "tile", "K-tile", "GHz"))
    for name in names:
        S, cin, cout, pool = LAYERS[name]
        ms = C.c_double()
        lrcn_amd._lib.check(ctx._h, lib.lrcn_bench_conv(ctx._h, N, S, cin, cout, pool, 3, C.byref(ms)))
        route = L.debug_route(ctx)
        cfg = int(route.split(":")[1]) if route.startswith("8p:") else -1
        if cfg < 0:
            print("%-8s %-6s (not the phase-interleaved kernel)" % (name, route))
            continue
        bm, bn = {0: (256, 256), 1: (256, 128), 2: (512, 128)}[cfg]
        M = N * S * S
        ntiles = -(-M // bm) * -(-cout // bn)
        buf = (C.c_ulonglong * (ntiles * 8))()
        lrcn_amd._lib.check(ctx._h, lib.lrcn_debug_stamps(ctx._h, buf, ntiles * 8))
        st = np.frombuffer(buf, dtype=np.uint64).reshape(ntiles, 8).astype(np.int64)
        grid = min(ntiles, 256)  # uncapped: one workgroup per tile up to ... the kernel is launched with ntiles workgroups; tiles run in waves of CUs
        seg = lambda a, b: np.median(st[:, b] - st[:, a])
        # whole-tile time: consecutive tiles on one CU are not identifiable without a cap; use start-to-stores-issued + the next wait
        tile = np.median(st[:, 5] - st[:, 0])
        kt = 9 * cin // 64
        # clock the chip holds inside the kernel: shader cycles per 10 ns tick of the wall counter, per tile, median
        ticks = (st[:, 6] - st[:, 7]).astype(np.float64)
        ghz = float(np.median((st[:, 5] - st[:, 0]) / np.maximum(ticks, 1.0))) / 10.0
        print("%-8s %-6s %6d | %8.0f %8.0f %8.0f %8.0f %8.0f | %8.0f %8.1f  %5.2f" % (name, route, ntiles, seg(0, 1), seg(1, 2), seg(2, 3), seg(3, 4),
                                                                              seg(4, 5), tile, seg(2, 3) / kt, ghz))
    ctx.close()


if __name__ == "__main__":
    main()
