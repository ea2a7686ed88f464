#!/bin/bash
# on the GPU box: sample rocm-smi (power, clocks, temperature) twice a second while bench.py runs; prints a summary.
# usage: tools/power_trace.sh [bench.py arguments...]
out=${GRAFT_REPO_ROOT:-.}/gpurun_out/power_trace.txt
python3 ${GRAFT_REPO_ROOT:-.}/bench.py --steps 2500 --warmup 20 --no-cpu-baseline "$@" > /tmp/pt_bench.json 2>/tmp/pt_bench.err &
pid=$!
sleep 6
: > $out
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showpower --showclocks --showtemp --showperflevel 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (junction|edge)" | sed 's/^/  /' >> $out
  echo "--" >> $out
  sleep 0.5
done
wait $pid
tail -c 300 /tmp/pt_bench.json; echo
python3 - "$out" <<'PY'
import re, sys, statistics
txt = open(sys.argv[1]).read()
pw = [float(x) for x in re.findall(r"Power \(W\):\s*([0-9.]+)", txt)]
sc = [float(x) for x in re.findall(r"sclk clock level:.*?\((\d+)Mhz\)", txt)]
tj = [float(x) for x in re.findall(r"junction\) \(C\):\s*([0-9.]+)", txt)]
for name, v in (("socket power W", pw), ("sclk MHz", sc), ("junction C", tj)):
    if v: print("%s: n=%d min %.0f median %.0f max %.0f" % (name, len(v), min(v), statistics.median(v), max(v)))
PY
rocm-smi --showmaxpower 2>/dev/null | grep -i "power" | head -3
