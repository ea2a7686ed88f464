#!/usr/bin/env python3
"""Kernel-development aid: where a patch of the halo-patch convolution kernel (conv64.hip) spends its time.
  python tools/conv64_stamps.py [N=256] [cap=0]
Per kernel (conv1_1+conv1_2 fused from a VGG forward; conv1_2 and conv2_1 alone through lrcn_bench_conv) and per wave group (0 = waves 0..3,
1 = waves 4..7, one barrier behind): median shader cycles of [first half-taps | mid barrier | second half-taps | epilogue (+ producer) |
end barrier], the whole patch, and the clock held (shader cycles per 10 ns of the wall counter).  Needs an MI355X."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def report(name, lib, ctx, ntiles, skip=0, fused=False):
    import lrcn_amd
    buf = (C.c_ulonglong * (ntiles * 16))()
    lrcn_amd._lib.check(ctx._h, lib.lrcn_debug_stamps(ctx._h, buf, ntiles * 16))
    st = np.frombuffer(buf, dtype=np.uint64).reshape(ntiles, 2, 8).astype(np.int64)[skip:]
    for g in range(2):
        s = st[:, g, :]
        ok = (s[:, 5] > s[:, 0]) & (s[:, 0] > 0)
        s = s[ok]
        seg = lambda a, b: float(np.median(s[:, b] - s[:, a]))
        if fused and os.environ.get("LRCN_FUSE11_GEN", "2") != "1":
            # conv64f.hip: [0] patch start, [1] sub-step 36 of 72, [3] loop done, [4] epilogue + this wave's LDS writes done, [5] barrier passed
            ticks = np.maximum((s[:, 6] - s[:, 7]).astype(np.float64), 1.0)
            ghz = float(np.median((s[:, 5] - s[:, 0]) / ticks)) / 10.0
            print("%-22s group %d | sub-steps 0-35 %6.0f  36-71 %6.0f  epilogue %6.0f  barrier %6.0f | patch %7.0f cycles  %.2f GHz  (%d tiles)"
                  % (name, g, seg(0, 1), seg(1, 3), seg(3, 4), seg(4, 5), seg(0, 5), ghz, len(s)))
            continue
        if fused:   # fine stamps of the FUSE epilogue phase: [3] stores issued, [6] raw window landed, [7] producer's LDS reads landed
            ok2 = (s[:, 7] > s[:, 6]) & (s[:, 6] > s[:, 2])
            t = s[ok2]
            sg = lambda a, b: float(np.median(t[:, b] - t[:, a]))
            # stamp order in the kernel: 0 start, 1 half-taps done, 3 epilogue stores issued, 2 mid barrier passed, 6 raw window landed,
            # 7 producer's LDS reads landed, 4 producer done, 5 end barrier passed
            print("%-22s group %d | half-taps %7.0f  epilogue+stores %6.0f  barrier %6.0f | raw-window wait %6.0f  producer reads %6.0f  producer math+writes %6.0f  "
                  "barrier %6.0f | patch %7.0f cycles (%d tiles)" % (name, g, seg(0, 1), sg(1, 3), sg(3, 2), sg(2, 6), sg(6, 7), sg(7, 4), seg(4, 5), seg(0, 5), len(t)))
            continue
        ticks = np.maximum((s[:, 6] - s[:, 7]).astype(np.float64), 1.0)
        ghz = float(np.median((s[:, 5] - s[:, 0]) / ticks)) / 10.0
        print("%-22s group %d | halves-1 %7.0f  barrier %6.0f  halves-2 %7.0f  epilogue %7.0f  barrier %6.0f | patch %7.0f cycles  %.2f GHz  (%d tiles)"
              % (name, g, seg(0, 1), seg(1, 2), seg(2, 3), seg(3, 4), seg(4, 5), seg(0, 5), ghz, len(s)))


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    cap = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    os.environ["LRCN_STAMPS"] = "f"
    import lrcn_amd
    from lrcn_amd import lrcn as L
    lib = lrcn_amd._lib.lib()
    ctx = L.Context(8, 8, 8, 17, max_B=2, max_T=1, lstm_dtype=lrcn_amd.LRCN_BF16, vgg_dtype=lrcn_amd.LRCN_BF16, max_images=N)
    if cap:
        L.vgg_set_wg_cap(ctx, cap)
    L.vgg_load(ctx, *L.synthetic_vgg_weights(seed=1))
    img = torch.randint(0, 256, (N, 224, 224, 3), device="cuda", dtype=torch.uint8)
    for _ in range(2):
        L.convnet_u8(ctx, img)
    torch.cuda.synchronize()
    # the forward's conv2_1 (also conv64.hip) overwrote nothing: LRCN_STAMPS=f stamps the fused kernel only ... but conv_layer stamps too
    # whenever LRCN_STAMPS is set, so read the fused kernel's stamps from a forward cut short: run the fused launch alone
    os.environ["LRCN_STAMPS"] = "f"
    report("conv1_1+conv1_2 fused*", lib, ctx, N * 196, skip=N * 49, fused=True)
    ms = C.c_double()
    for name, (S, cin, cout, pool) in {"conv1_2 (unfused)": (224, 64, 64, 1), "conv2_1": (112, 64, 128, 0)}.items():
        lrcn_amd._lib.check(ctx._h, lib.lrcn_bench_conv(ctx._h, N, S, cin, cout, pool, 2, C.byref(ms)))
        report(name + " %.3f ms" % ms.value, lib, ctx, N * (S // 16) * (S // 16))
    print("* the fused kernel's stamps are overwritten by conv2_1's for tiles < N * 49 (same buffer): its medians come from the rest")
    ctx.close()


if __name__ == "__main__":
    main()
