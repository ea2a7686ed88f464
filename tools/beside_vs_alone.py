#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV of `bench.py --emulate-world 8`: every kernel of the LSTM chain, averaged separately over its launches
that ran BESIDE a convolution launch of the side-stream VGG forward (any overlap in time) and over those that ran with the chip to
themselves -- what sharing the chip costs each kernel of the chain.   usage: tools/beside_vs_alone.py <kernel_trace.csv> [skip_first_ms]"""
import bisect
import collections
import csv
import sys


def short(n):
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")
    return n[:60]


def is_conv(k):
    if "conv64_kernel" in k or "conv64f_kernel" in k or "img_u8_to_bf16" in k:
        return True
    if "gemm8p_kernel<" in k:
        a = [x.strip() for x in k[k.index("<") + 1:k.index(">")].split(",")]
        return a[4] == "1"
    return False


def main():
    rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
    skip = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 0.0
    t00 = int(rows[0]["Start_Timestamp"])
    conv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if is_conv(r["Kernel_Name"])]
    starts = [c[0] for c in conv]
    agg = collections.defaultdict(lambda: [0, 0.0, 0, 0.0])
    for r in rows:
        k = r["Kernel_Name"]
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if is_conv(k) or s - t00 < skip or "at::" in k or "rocclr_copy" in k:
            continue
        i = bisect.bisect_right(starts, e) - 1
        beside = False
        while i >= 0 and conv[i][0] > s - 3_000_000:   # convolution launches last <= ~2 ms
            if conv[i][1] > s and conv[i][0] < e:
                beside = True
                break
            i -= 1
        a = agg[short(k)]
        if beside:
            a[0] += 1
            a[1] += (e - s) / 1e3
        else:
            a[2] += 1
            a[3] += (e - s) / 1e3
    tot_b = sum(a[1] for a in agg.values())
    tot_a = sum(a[3] for a in agg.values())
    print("%-62s %7s %9s %7s %9s %6s" % ("kernel", "n besid", "avg us", "n alone", "avg us", "ratio"))
    for k, a in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][3])):
        ab = a[1] / a[0] if a[0] else 0.0
        aa = a[3] / a[2] if a[2] else 0.0
        print("%-62s %7d %9.1f %7d %9.1f %6.2f" % (k, a[0], ab, a[2], aa, ab / aa if aa and ab else 0.0))
    print("busy time of the chain's kernels: beside %.1f ms, alone %.1f ms" % (tot_b / 1e3, tot_a / 1e3))


if __name__ == "__main__":
    main()
