#!/usr/bin/env python3
"""Anatomy of ONE steady-state training step of a rocprofv3 --kernel-trace CSV of `bench.py --emulate-world 8` (the per-rank step of an
8-GPU job): every kernel of the step in start order with its hardware queue, duration and the idle gap since the previous kernel ON THE
SAME QUEUE ended, then per queue: busy time, summed gaps, launches.  The step = from the end of one update (prepare_weights_kernel / adam) to
the end of the next.   usage: tools/step_chain.py <kernel_trace.csv> [step index from the end, default 3] [--brief]"""
import collections
import csv
import sys


def short(n):
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")
    return n[:64]


def main():
    rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
    back = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 3
    upd = [i for i, r in enumerate(rows) if "prepare_weights_kernel" in r["Kernel_Name"] or "adam_kernel" in r["Kernel_Name"] or "adam_flat" in r["Kernel_Name"]]
    # one update per step is the LAST update kernel of a burst: keep those followed by > 20 other kernels before the next update
    ends = [i for k, i in enumerate(upd) if k + 1 == len(upd) or upd[k + 1] - i > 20]
    a, b = ends[-back - 1], ends[-back]
    t0 = int(rows[a]["End_Timestamp"])
    t1 = int(rows[b]["End_Timestamp"])
    step = [r for r in rows if t0 <= int(r["Start_Timestamp"]) and int(r["End_Timestamp"]) <= t1 + 1]
    last_end = {}
    per_q = collections.defaultdict(lambda: [0.0, 0.0, 0])
    per_k = collections.defaultdict(lambda: [0, 0.0, 0.0])
    lines = []
    for r in step:
        q, s, e = r["Queue_Id"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - last_end[q]) / 1e3 if q in last_end else (s - t0) / 1e3
        last_end[q] = e
        per_q[q][0] += (e - s) / 1e3
        per_q[q][1] += max(gap, 0.0)
        per_q[q][2] += 1
        k = per_k[(q, short(r["Kernel_Name"]))]
        k[0] += 1
        k[1] += (e - s) / 1e3
        k[2] += max(gap, 0.0)
        lines.append("%8.1f  q%-3s +%7.1f us  gap %6.1f  grid %6s  %s" % ((s - t0) / 1e3, q, (e - s) / 1e3, gap, r["Grid_Size_X"], short(r["Kernel_Name"])))
    if "--brief" not in sys.argv:
        print("\n".join(lines))
    print("step wall %.1f us (update end to update end), %d kernels" % ((t1 - t0) / 1e3, len(step)))
    for q, (busy, gaps, n) in sorted(per_q.items(), key=lambda kv: -kv[1][0]):
        print("queue %s: busy %.1f us, idle gaps between its kernels %.1f us, %d launches" % (q, busy, gaps, n))
    print("per kernel (queue, name): launches, busy us, gap-before us")
    for (q, n), (c, busy, gaps) in sorted(per_k.items(), key=lambda kv: -kv[1][1]):
        print("  q%-3s x%-3d %8.1f %8.1f  %s" % (q, c, busy, gaps, n))


if __name__ == "__main__":
    main()
