#!/bin/bash
# kernel-trace stats of the C5 caption bench (GPU box)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc5
LRCN_C5_LIGHT=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc5 -o p -- python3 $GRAFT_REPO_ROOT/bench.py --config c5 --steps 30 > $GRAFT_REPO_ROOT/gpurun_out/r06_c5_under_rocprof.json 2> /tmp/pc5.log
cp /tmp/pc5/p_kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/r06_kernel_stats_caption_bench_c5.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("/tmp/pc5/p_kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:24]: print("%-100s %6s calls %9.1f us avg %5.1f%%"%(r['Name'][:100], r['Calls'], float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
PY
