#!/usr/bin/env python3
"""Two-stream anatomy of bench.py from a rocprofv3 --kernel-trace CSV: per step, the period (Adam to Adam) and how long the
LSTM stream's chain (first kernel after the previous Adam .. this Adam) is stretched by sharing the chip with the VGG forward
of the next step.  usage: tools/timeline.py <kernel_trace.csv>"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
adam = [r for r in rows if "adam_kernel" in r["Kernel_Name"]]
mq = adam[0]["Queue_Id"]
ae = [int(r["End_Timestamp"]) for r in adam]
print("adam-to-adam ms:", [round((b - a) / 1e6, 3) for a, b in zip(ae, ae[1:])][-6:])
mk = [r for r in rows if r["Queue_Id"] == mq]
for i in range(max(1, len(ae) - 5), len(ae)):
    ks = [r for r in mk if ae[i - 1] < int(r["Start_Timestamp"]) <= int(adam[i]["Start_Timestamp"])]
    st = min(int(r["Start_Timestamp"]) for r in ks)
    print("LSTM chain %d: span %.3f ms, kernel-busy %.3f ms, %d launches, starts %.3f ms after the previous Adam" % (
        i, (ae[i] - st) / 1e6, sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ks) / 1e6, len(ks), (st - ae[i - 1]) / 1e6))
