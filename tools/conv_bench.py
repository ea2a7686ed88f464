#!/usr/bin/env python3
"""Kernel-development aid: TFLOP/s of each VGG conv layer shape (bf16 implicit GEMM) in isolation. Needs an MI355X."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lrcn_amd  # noqa: E402
from lrcn_amd import lrcn as L  # noqa: E402

LAYERS = [("conv1_2", 224, 64, 64, 1), ("conv2_1", 112, 64, 128, 0), ("conv2_2", 112, 128, 128, 1),
          ("conv3_1", 56, 128, 256, 0), ("conv3_2", 56, 256, 256, 0), ("conv3_3", 56, 256, 256, 1),
          ("conv4_1", 28, 256, 512, 0), ("conv4_2", 28, 512, 512, 0), ("conv4_3", 28, 512, 512, 1),
          ("conv5_1", 14, 512, 512, 0), ("conv5_2", 14, 512, 512, 0), ("conv5_3", 14, 512, 512, 1)]


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
    ctx = L.Context(8, 8, 8, 17, max_B=2, max_T=1, lstm_dtype=lrcn_amd.LRCN_BF16, vgg_dtype=lrcn_amd.LRCN_BF16, max_images=1)  # max_images: the VGG stream's split-K workspace
    lib = lrcn_amd._lib.lib()
    if os.environ.get("LRCN_VGG_WG_CAP"):   # the capped persistent grids of the two-stream training step (dp.py sets 224 at 256 rows)
        L.vgg_set_wg_cap(ctx, int(os.environ["LRCN_VGG_WG_CAP"]))
    tot_f = tot_t = 0.0
    for name, S, cin, cout, pool in LAYERS:
        if only and name not in only:
            continue
        ms = C.c_double()
        lrcn_amd._lib.check(ctx._h, lib.lrcn_bench_conv(ctx._h, N, S, cin, cout, pool, 10, C.byref(ms)))
        fl = 2.0 * N * S * S * cout * 9 * cin
        tot_f += fl
        tot_t += ms.value
        print("%-8s S=%3d Cin=%3d Cout=%3d pool=%d  %8.3f ms  %7.1f TF  %s" % (name, S, cin, cout, pool, ms.value, fl / ms.value / 1e9, L.debug_route(ctx)))
    print("sum %.3f ms  %.1f TF" % (tot_t, tot_f / tot_t / 1e9))


if __name__ == "__main__":
    main()
