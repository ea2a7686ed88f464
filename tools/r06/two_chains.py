#!/usr/bin/env python3
"""Round 6: which of the two chains of the C4 step is the critical one?  From a rocprofv3 --kernel-trace CSV of bench.py: the hardware queue that
carries the convolution launches (VGG side stream) and the one that carries the LSTM chain (main stream); per step (conv64f launch to conv64f
launch): busy time, idle time and the largest gaps of each queue, with the kernel that FOLLOWS each gap.  A queue that is never idle for
longer than a launch hand-over (4-5 us) is the critical chain; gaps of 50+ us are waits on the other stream or on the host.
usage: tools/r06/two_chains.py <kernel_trace.csv> [steps_from_end]"""
import csv
import sys
from collections import defaultdict


def short(n):
    return n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")[:60]


rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 4
marks = [r for r in rows if "conv64f_kernel" in r["Kernel_Name"]]
side_q = marks[-1]["Queue_Id"]
byq = defaultdict(list)
for r in rows:
    byq[r["Queue_Id"]].append(r)
t_a, t_b = int(marks[-1 - nlast]["Start_Timestamp"]), int(marks[-1]["Start_Timestamp"])
print("period: %.3f ms per step over the last %d steps; queues: %s" % ((t_b - t_a) / 1e6 / nlast, nlast, {q: len(v) for q, v in byq.items()}))
for q, ks in byq.items():
    ks = [r for r in ks if t_a <= int(r["Start_Timestamp"]) < t_b]
    if len(ks) < 5:
        continue
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ks)
    gaps = []
    for a, b in zip(ks, ks[1:]):
        g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
        gaps.append((g, short(a["Kernel_Name"]), short(b["Kernel_Name"])))
    span = int(ks[-1]["End_Timestamp"]) - int(ks[0]["Start_Timestamp"])
    small = sum(g for g, _, _ in gaps if 0 < g <= 10000)
    big = [x for x in gaps if x[0] > 10000]
    print("\nqueue %s%s: %d launches per step, busy %.3f ms/step, hand-over gaps (<= 10 us) %.3f ms/step, %d larger gaps totalling %.3f ms/step, overlap(negative gaps) %.3f ms/step"
          % (q, " (VGG side stream)" if q == side_q else "", len(ks) // nlast, busy / 1e6 / nlast, small / 1e6 / nlast, len(big),
             sum(g for g, _, _ in big) / 1e6 / nlast, -sum(g for g, _, _ in gaps if g < 0) / 1e6 / nlast))
    grp = defaultdict(lambda: [0, 0])
    for g, a, b in big:
        grp[(a, b)][0] += 1
        grp[(a, b)][1] += g
    for (a, b), (n, tot) in sorted(grp.items(), key=lambda kv: -kv[1][1])[:14]:
        print("   %5d x %7.1f us avg = %7.3f ms/step  after %-58s before %s" % (n, tot / n / 1e3, tot / 1e6 / nlast, a, b))
