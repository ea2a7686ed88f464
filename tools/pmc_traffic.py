#!/usr/bin/env python3
"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; collected SEPARATELY, with --kernel-trace only) of
`bench.py` into the per-launch HBM-side traffic of the dominant kernel family (the 12 implicit-GEMM convolution
launches per step), as /opt/skills/guides/MI355X_MICROARCH.md section HBM prescribes:
  bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024   -- FETCH_SIZE reads exactly half of a wide (16 B/lane) coalesced
  stream on gfx950 (global_load and LDS-DMA alike), WRITE_SIZE is exact for 16-B stores; both are in KiB.
The correction is checked on the same run against a kernel with a known byte count (adam_kernel: 16 B/param read,
12 B/param written).  Usage: tools/pmc_traffic.py <fetch_counter_csv> <write_counter_csv> <out.json> [params]"""
import collections
import csv
import json
import sys


def load(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return d


def main():
    fetch, write, out = sys.argv[1:4]
    nparam = int(sys.argv[4]) if len(sys.argv) > 4 else 39846640
    fe, wr = load(fetch), load(write)
    def is_conv(k):  # the 12 launches conv1_2 .. conv5_3: conv64_kernel (Cin = 64) and the CONV3 instantiations of the GEMMs
        if "conv64_kernel" in k or "conv64f_kernel" in k:  # conv64f_kernel: the fused conv1_1 + conv1_2 launch since round 4
            return True
        if "gemm8p_kernel<" in k or "gemm_glds_kernel<" in k:
            args = k[k.index("<") + 1:k.index(">")].split(",")
            return args[4].strip() == "1"  # AMODE == GEMM_A_CONV3
        return False
    conv = [k for k in fe if is_conv(k)]
    n = sum(len(fe[k]) for k in conv)
    f_kib = sum(sum(fe[k]) for k in conv)
    w_kib = sum(sum(wr[k]) for k in conv)
    res = {
        "kernel_family": "conv64f_kernel + conv64_kernel + gemm8p_kernel<*,CONV3,*> (conv1_2..conv5_3)",
        "launches_counted": n,
        "fetch_size_kib_per_launch": f_kib / n,
        "write_size_kib_per_launch": w_kib / n,
        "traffic_bytes_per_launch": (2.0 * f_kib + w_kib) * 1024.0 / n,
        "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request)",
    }
    adam = [k for k in fe if "adam_kernel" in k]
    if adam:
        a_f = sum(fe[adam[0]]) / len(fe[adam[0]]) * 1024.0
        a_w = sum(wr[adam[0]]) / len(wr[adam[0]]) * 1024.0
        res["calibration_adam"] = {"fetch_x2_bytes": 2 * a_f, "algorithmic_read_bytes": 16.0 * nparam,
                                   "write_bytes": a_w, "algorithmic_write_bytes": 12.0 * nparam}
    res["per_kernel"] = {k[:90]: {"n": len(fe[k]), "fetch_kib_avg": sum(fe[k]) / len(fe[k]),
                                  "write_kib_avg": sum(wr.get(k, [0])) / max(1, len(wr.get(k, [0])))}
                         for k in sorted(fe, key=lambda k: -sum(fe[k]))[:16]}
    # bench.py quotes this file only while the kernel sources are the ones it was measured on (the number went stale silently)
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    res["csrc_digest"] = bench.csrc_digest()
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "per_kernel"}, indent=1))


if __name__ == "__main__":
    main()
