#!/usr/bin/env python3
"""Kernel-development aid: print the kernel timeline of the last full training step of a rocprofv3 --kernel-trace CSV
(bench.py run), or the per-step kernel totals (--stats).  Usage: tools/trace_step.py <kernel_trace.csv> [--stats]"""
import collections
import csv
import sys


def short(n):
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")
    return n[:72]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    key = "conv11_kernel" if any("conv11_kernel" in r["Kernel_Name"] for r in rows) else "adam_kernel"
    idx = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
    a, b = idx[-2], idx[-1]
    step = rows[a:b]  # one full step: from one conv1_1 launch (VGG of step k) to the next
    if "--stats" in sys.argv:
        tot = collections.defaultdict(lambda: [0, 0.0])
        for r in step:
            t = tot[short(r["Kernel_Name"])]
            t[0] += 1
            t[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        for k, (n, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
            print("%9.1f us  x%-3d %s" % (us, n, k))
        print("sum %.1f us, wall %.1f us" % (sum(v[1] for v in tot.values()),
                                             (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e3))
        return
    t0 = int(step[0]["Start_Timestamp"])
    for r in step:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("%8.1f +%7.1f  grid %sx%s wg %s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Grid_Size_X"], r["Grid_Size_Y"],
                                                     r["Workgroup_Size_X"], short(r["Kernel_Name"])))


if __name__ == "__main__":
    main()
